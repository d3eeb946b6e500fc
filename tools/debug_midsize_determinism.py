import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import elodin_amd as ea
from elodin_amd import workloads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 512
w = workloads.independent_bodies(n)
eff = workloads.gravity_torque_effectors(w["body_torque"])
def run(**kw):
    ex = ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], simulation_time_step=workloads.DT_120HZ, effectors=eff, **kw)
    if kw.get("use_graph"): ex.prepare(ticks)
    ex.run(ticks)
    out = (ex.world_pos.copy(), ex.world_vel.copy(), ex.world_accel.copy(), ex.force.copy())
    ex.close()
    return out
ref = run(ticks_per_launch=64)
for name, kw in (("fused64 again", dict(ticks_per_launch=64)), ("eager K=1", dict(ticks_per_launch=1)), ("graph K=1", dict(ticks_per_launch=1, use_graph=True)),
                 ("graph K=1 again", dict(ticks_per_launch=1, use_graph=True))):
    got = run(**kw)
    bad = [int((~np.isclose(a, b, rtol=0, atol=0, equal_nan=True)).any(axis=1).sum()) for a, b in zip(got, ref)]
    print(f"n={n} {name}: rows differing from fused64 (pos, vel, accel, force) = {bad}")
