import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SIXDOF_COMM_FORCE_RCCL"] = "1"
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
    torch.zeros(4, device="cuda").sum().item()
    if len(sys.argv) > 2:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
        dist.init_process_group("nccl", rank=0, world_size=1)
        t = torch.ones(4, device="cuda"); dist.all_reduce(t); print("torch rccl ok", t.sum().item())
from elodin_amd import shard, _lib as L
try:
    c = shard.CapiComm(None, 1, 0, 0)
    table = np.random.default_rng(1).normal(size=(8192, 17))
    got = c.broadcast_table(table, table.shape)
    print("broadcast equal", np.array_equal(got, table))
    rows = np.random.default_rng(2).normal(size=(8192, 12))
    print("gather equal", np.array_equal(c.gather_rows(rows, 8192), rows))
    c.close()
except Exception as e:
    print("FAILED:", repr(e))
for line in open("/proc/self/maps"):
    if "rccl" in line or "amdhip64" in line:
        p = line.split()[-1]
        if p not in globals().setdefault("_seen", set()):
            _seen.add(p); print("mapped:", p)
