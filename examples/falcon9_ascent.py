#!/usr/bin/env python3
"""examples/falcon9 of the reference — the ascent half — as a GPU Monte-Carlo campaign (BASELINE config 5).

  python examples/falcon9_ascent.py [rollouts] [f32|f64]

Every rollout is one lane of the generated step kernel: plant, hold-down clamp, 1 kHz attitude loop and the 100 Hz
ascent guidance of the flight software run fused, 1000 ticks per launch (elodin_amd/models/falcon9.py).  The plan is
spec.toml's Latin-hypercube table (seed 20170814) through the reference's own sampler."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np  # noqa: E402

from elodin_amd.models import falcon9 as f9  # noqa: E402


def main(n=4096, dtype="f32"):
    params = f9.sample_params(n)
    t0 = time.perf_counter()
    ex = f9.AscentExec(params, dtype=np.float32 if dtype == "f32" else np.float64)
    t1 = time.perf_counter()
    ex.run(f9.ASCENT_TICKS)
    t2 = time.perf_counter()
    res = ex.result
    print(f"{n} rollouts x {f9.ASCENT_TICKS} ticks ({dtype}): build {t1 - t0:.2f} s, flight {t2 - t1:.2f} s "
          f"= {n * f9.ASCENT_TICKS / (t2 - t1):.3e} rollout-steps/s")
    nominal = f9.AscentExec(f9.default_param_row()[None, :], dtype=np.float32 if dtype == "f32" else np.float64)
    nominal.run(f9.ASCENT_TICKS)
    print("calibrated defaults:", {k: round(float(v), 2) for k, v in zip(f9.METRIC_NAMES, nominal.result[0])},
          "(recorded CRS-12: Max-Q T+64 s, MECO T+147 s)")
    for k, name in enumerate(f9.METRIC_NAMES):
        col = res[:, k]
        print(f"  {name:18s} min {col.min():12.2f}  p50 {np.median(col):12.2f}  max {col.max():12.2f}")
    return res


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 4096, sys.argv[2] if len(sys.argv) > 2 else "f32")
