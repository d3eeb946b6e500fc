#!/usr/bin/env python3
"""Where a one-tick launch of BASELINE configs[1] (65,536 bodies) spends its time, and the row-block-chain lever.

  python tools/k1_floor.py            # child processes, one per SIXDOF_GRAPH_SPLIT value (the library reads it at load)

Per split S in {1, 2, 4, 8}: the driver's 20-step window (host clock and HIP events), a 4,096-launch batch, the bit pattern
of the state after 256 ticks (must be identical for every S), three interleaved passes."""
import hashlib
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def child():
    import numpy as np
    import torch
    import bench
    n = int(os.environ.get("K1_ENTITIES", "65536"))
    ex, w, eff = bench.make_exec(n, 0, 0, 1, True)
    ex.prepare(20)
    ex.invoke_batch(5)
    out = {"split": int(os.environ.get("SIXDOF_GRAPH_SPLIT", "1")), "n": n}
    host, dev = [], []
    for _ in range(30):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tm = ex.invoke_batch(20)
        host.append((time.perf_counter() - t0) / 20 * 1e6)
        dev.append(tm.kernel_device_ms / 20 * 1e3)
        time.sleep(0.002)              # the idle gap a barrier leaves in front of the timed region
    out["steps20_host_us"] = [round(min(host), 3), round(sorted(host)[len(host) // 2], 3)]
    out["steps20_device_us"] = [round(min(dev), 3), round(sorted(dev)[len(dev) // 2], 3)]
    ex.prepare(4096)
    ex.invoke_batch(256)
    long_ = []
    for _ in range(5):
        tm = ex.invoke_batch(4096)
        long_.append(tm.kernel_device_ms / 4096 * 1e3)
    out["long_batch_us"] = [round(min(long_), 3), round(sorted(long_)[2], 3)]
    ex.close()
    ex, w, eff = bench.make_exec(n, 0, 0, 1, True)       # a fresh world: 256 ticks, the bits
    ex.prepare(64)
    ex.invoke_batch(256)
    ex.download()
    h = hashlib.sha256()
    for f in ("world_pos", "world_vel", "world_accel", "force"):
        h.update(np.ascontiguousarray(getattr(ex, f)).tobytes())
    out["state_sha256_after_256_ticks"] = h.hexdigest()[:16]
    ex.close()
    print(json.dumps(out), flush=True)


def main():
    if os.environ.get("K1_CHILD") == "1":
        return child()
    rows = []
    for n in (65536, 262144):
        for rep in range(3):
            for s in (1, 2, 4, 8):
                env = dict(os.environ, K1_CHILD="1", SIXDOF_GRAPH_SPLIT=str(s), K1_ENTITIES=str(n))
                r = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True, timeout=600)
                line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                if r.returncode != 0 or not line:
                    print(f"split {s} n {n}: FAILED rc {r.returncode}\n{r.stderr[-1500:]}")
                    continue
                rows.append(json.loads(line[-1]))
                print(line[-1], flush=True)
    for n in (65536, 262144):
        shas = {r["state_sha256_after_256_ticks"] for r in rows if r["n"] == n}
        print(f"n = {n}: state after 256 ticks bit-identical across splits: {len(shas) == 1} {sorted(shas)}")


if __name__ == "__main__":
    main()
