#!/usr/bin/env python3
"""Falcon 9 program with ticks_per_launch = 1 (every tick round-trips every column through HBM, the reference's per-tick
column semantics): time per tick and algorithmic HBM GB/s, for the config-5 roofline statement — under each cache policy
of the launch (SIXDOF_STREAMING: 0 = plain loads + plain stores, 1 = plain loads + non-temporal stores, 9 = non-temporal both
ways; csrc/step_kernel.hpp col_ld / col_st<POL> for the program's columns, slab DMA / slab_out for the Body columns) and the
default the library picks from the state size.   python tools/falcon9_k1.py [rollouts ...]"""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from elodin_amd.models import falcon9 as f9

print("Falcon 9 ascent program (f32, fast math), ONE tick per launch: us per tick | algorithmic GB/s | fraction of the 8 TB/s HBM peak")
for n in [int(x) for x in sys.argv[1:]] or (32768, 262144, 1048576):
    row = []
    for policy in ("default", "0", "1", "9"):
        if policy == "default":
            os.environ.pop("SIXDOF_STREAMING", None)
        else:
            os.environ["SIXDOF_STREAMING"] = policy
        ex = f9.AscentExec(np.tile(f9.default_param_row(), (n, 1)), dtype=np.float32, fast_math=True, ticks_per_launch=1)
        tr = ex.program.trace()
        read_b = 4 * (sum(w for _, w in tr.columns) + 7 + 6 + 7)
        written = {t.split("_")[0] for s in tr.pre + tr.post for t in s.written if t[0] == "c"}
        write_b = 4 * (sum(w for k, (nm, w) in enumerate(tr.columns) if f"c{k}" in written) + 7 + 6 + 6 + 6 + 7)
        ex.hip.invoke_batch(20)
        ticks = 200 if n < 1_000_000 else 60
        best = min(ex.hip.invoke_batch(ticks).kernel_device_ms for _ in range(3)) / ticks * 1e3
        gbps = (read_b + write_b) * n / best / 1e3
        row.append(f"policy {policy:>7s}: {best:8.2f} us {gbps:7.1f} GB/s {gbps / 8000:.3f}")
        ex.close()
    print(f"{n:8d} rollouts, {read_b} B read + {write_b} B written per rollout-tick\n    " + "\n    ".join(row))
os.environ.pop("SIXDOF_STREAMING", None)
