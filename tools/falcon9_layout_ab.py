#!/usr/bin/env python3
"""A/B of the generated-program column layout on the Falcon 9 closed loop (f32, hardware transcendentals):
  SIXDOF_COLUMN_SOA            0 = [n][w] rows (the reference's layout), 1 = element-major [w][n] on the device
  SIXDOF_NO_TRANSIENT_COLUMNS  1 = per-tick (write-before-read) columns kept across ticks like any other
For each combination: one tick per launch at 262,144 rollouts (every tick round-trips the hot columns through HBM) and the
1,000-ticks-per-launch campaign at 32,768 rollouts.   python tools/falcon9_layout_ab.py [--prebuild]
--prebuild (no GPU needed) compiles the four variants into elodin_amd/_jit so the GPU box does not run hipcc."""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
COMBOS = [("0", "1"), ("1", "1"), ("0", "0"), ("1", "0")]          # (soa, no_transient)

if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, str(ROOT))
    import numpy as np
    from elodin_amd import codegen
    from elodin_amd.models import falcon9 as f9
    soa = os.environ["SIXDOF_COLUMN_SOA"] == "1"
    if sys.argv[2] == "prebuild":
        cols = f9.initial_columns(f9.default_param_row()[None, :])
        widths = {k: v.shape[1] for k, v in cols.items()}
        so = codegen.build(f9.build_program(origin=f9.pad_ecef()).trace(widths), "float32", 1, fast_math=True, column_soa=soa)
        print(so.name, codegen.last_resources)
        sys.exit(0)
    n1 = 262144
    ex = f9.AscentExec(np.tile(f9.default_param_row(), (n1, 1)), dtype=np.float32, fast_math=True, ticks_per_launch=1)
    ex.hip.invoke_batch(20)
    t = ex.hip.invoke_batch(200)
    us1 = t.kernel_device_ms / 200 * 1e3
    ex.close()
    ex = f9.AscentExec(f9.sample_params(32768), dtype=np.float32, fast_math=True, ticks_per_launch=1000)
    ex.hip.invoke_batch(1000)
    t = ex.hip.invoke_batch(20000)
    usk = t.kernel_device_ms / 20000 * 1e3
    ex.close()
    print(f"soa={os.environ['SIXDOF_COLUMN_SOA']} transient={'off' if os.environ.get('SIXDOF_NO_TRANSIENT_COLUMNS') == '1' else 'on '}: "
          f"K=1 @262,144 rollouts {us1:8.2f} us/tick | K=1000 @32,768 rollouts {usk:6.3f} us/tick = {32768 / usk * 1e6:.3e} rollout-steps/s")
    sys.exit(0)

mode = "prebuild" if "--prebuild" in sys.argv else "run"
for soa, no_tr in COMBOS:
    env = dict(os.environ, SIXDOF_COLUMN_SOA=soa, SIXDOF_NO_TRANSIENT_COLUMNS=no_tr)
    r = subprocess.run([sys.executable, __file__, "--one", mode], env=env, capture_output=True, text=True)
    print((r.stdout.strip().splitlines() or ["(no output)"])[-1], flush=True)
    if r.returncode != 0:
        print(r.stderr[-1500:])
