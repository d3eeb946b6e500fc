#!/usr/bin/env python3
"""Falcon 9 program with ticks_per_launch = 1 (every tick round-trips every column through HBM, the reference's per-tick
column semantics): time per tick and algorithmic HBM GB/s, for the config-5 roofline statement."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from elodin_amd.models import falcon9 as f9

for n in [int(x) for x in sys.argv[1:]] or (32768, 262144, 1048576):
    ex = f9.AscentExec(np.tile(f9.default_param_row(), (n, 1)), dtype=np.float32, fast_math=True, ticks_per_launch=1)
    widths = dict(ex.program.trace().columns)
    read_b = 4 * (sum(widths.values()) + 7 + 6 + 7)
    written = {t.split("_")[0] for s in ex.program.trace().pre + ex.program.trace().post for t in s.written if t[0] == "c"}
    write_b = 4 * (sum(w for k, (nm, w) in enumerate(ex.program.trace().columns) if f"c{k}" in written) + 7 + 6 + 6 + 6 + 7)
    ex.hip.invoke_batch(20)
    ticks = 200 if n < 1_000_000 else 60
    t = ex.hip.invoke_batch(ticks)
    us = t.kernel_device_ms / ticks * 1e3
    print(f"{n:8d} rollouts: {us:9.2f} us/tick  read {read_b} B + write {write_b} B per rollout-tick -> "
          f"{(read_b + write_b) * n / us / 1e3:7.1f} GB/s algorithmic ({(read_b + write_b) * n / us / 1e3 / 8000:.2f} of 8 TB/s)")
    ex.close()
