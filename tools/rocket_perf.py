"""Throughput of the rocket example (tests/rocket_dsl.py: 13 systems | six_dof(RK4) with 3 effectors, a 480 x 3 window per
rocket) as a Monte-Carlo-sized batch of identical rockets: python tools/rocket_perf.py [n] [ticks_per_launch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import elodin_amd as ea
from elodin_amd import _lib as L
from tests import rocket_dsl as R
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
pos, vel, inertia, comps = R.spawn(n)
comps["v_rel_accel_buffer"] = comps["v_rel_accel_buffer"].reshape(n, R.LP_BUFFER_SIZE, 3)
hip = ea.HipExec(pos, vel, inertia, simulation_time_step=R.SIM_TIME_STEP_NS, integrator=L.RK4, effectors=R.program(), columns=comps,
                 ticks_per_launch=k)
hip.invoke_batch(4 * k)
best = min(hip.invoke_batch(16 * k).kernel_device_ms / (16 * k) for _ in range(3))
win_bytes = n * R.LP_BUFFER_SIZE * 3 * 8
print(f"rockets={n} ticks/launch={k}: {best:.4f} ms/tick = {n / best * 1e3:.3e} rocket-ticks/s; window read per tick {win_bytes / 1e6:.0f} MB "
      f"-> {win_bytes / best / 1e6:.0f} GB/s of window traffic alone")
