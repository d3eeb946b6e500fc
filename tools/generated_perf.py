import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import elodin_amd as ea
from elodin_amd import dsl, workloads
np_ = dsl.np
@dsl.effector
def gravity(force, inertia):
    return force + dsl.SpatialForce(linear=np_.array([0.0, 0.0, -9.81]) * inertia.mass())
@dsl.effector(body_torque=3)
def rcs(force, pos, body_torque):
    return force + dsl.SpatialForce(torque=pos.angular() @ body_torque)
n = 65536
w = workloads.independent_bodies(n)
for label, kw in (("built-in", dict(effectors=workloads.gravity_torque_effectors(w["body_torque"]))),
                  ("generated", dict(effectors=gravity | rcs, columns={"body_torque": w["body_torque"]}))):
    for K in (1, 64):
        ex = ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], simulation_time_step=workloads.DT_120HZ,
                        ticks_per_launch=K, use_graph=True, **kw)
        ex.invoke_batch(512)
        best = min(ex.invoke_batch(4096).kernel_device_ms / 4096 for _ in range(3))
        print(f"{label:10s} K={K:3d}: {best*1e3:.3f} us/tick  {n/best/1e6:.2f} G entity-steps/s")
        ex.close()
