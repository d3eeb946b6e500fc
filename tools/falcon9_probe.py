#!/usr/bin/env python3
"""Fly the nominal Falcon 9 ascent on the GPU and print a timeline (development probe)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from elodin_amd.models import falcon9 as f9

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
params = np.tile(f9.default_param_row(), (n, 1))
for dtype, local in ((np.float64, False), (np.float64, True), (np.float32, True)):
    t0 = time.time()
    ex = f9.AscentExec(params, dtype=dtype, local_origin=local, ticks_per_launch=1000)
    print(f"--- dtype={np.dtype(dtype).name} local_origin={local} build {time.time()-t0:.1f}s")
    for step in range(17):
        tt = ex.run(10_000)
        c = ex.column
        print(f"t={ex.hip.tick/1000:6.1f} phase={c('fsw_state')[0,0]:.0f} alt={c('altitude_geodetic')[0,0]/1000:7.2f}km "
              f"v={c('ground_speed')[0,0]:7.1f} m={c('inertia')[0,6]/1000:6.1f}t q={c('qbar')[0,0]/1000:5.1f}kPa "
              f"thr={c('thrust_total')[0,0]/1e6:5.2f}MN u={c('engine_cmd')[0,0]:.3f} mach={c('mach')[0,0]:.2f} "
              f"tvc={np.degrees(c('tvc_state')[0]).round(2)} kern={tt.kernel_device_ms:.1f}ms")
    print(dict(zip(f9.METRIC_NAMES, ex.result[0].round(2))))
    ex.close()
