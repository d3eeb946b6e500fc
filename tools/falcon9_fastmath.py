#!/usr/bin/env python3
"""f32 Falcon 9 campaign: library transcendentals vs hardware (fast_math) — speed and deviation from the f64 flight."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from elodin_amd.models import falcon9 as f9

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
params = f9.sample_params(n)
ref = f9.AscentExec(params, dtype=np.float64, local_origin=True)
ref.run(f9.ASCENT_TICKS)
a = ref.result
for fast in (False, True):
    ex = f9.AscentExec(params, dtype=np.float32, fast_math=fast)
    ex.hip.invoke_batch(1000)
    t0 = time.perf_counter(); ex.hip.invoke_batch(20000); dt = time.perf_counter() - t0
    ex.run(f9.ASCENT_TICKS - 21000)
    b = ex.result
    rel = np.abs(a - b) / np.maximum(np.abs(a), 1e-9)
    print(f"fast_math={fast}: {dt/20000*1e6:.2f} us/tick/wave; worst rel dev from f64 per metric:",
          {k: f"{rel[:, j].max():.2e}" for j, k in enumerate(f9.METRIC_NAMES)}, "abs MECO t dev max", np.abs(a[:, 3] - b[:, 3]).max())
    ex.close()
