import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import elodin_amd as ea
from elodin_amd import workloads
n = 65536
w = workloads.independent_bodies(n)
eff = workloads.gravity_torque_effectors(w["body_torque"])
for K, ring in ((1, 0), (64, 0), (64, 256), (16, 256), (256, 512)):
    ex = ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], simulation_time_step=workloads.DT_120HZ, effectors=eff,
                    ticks_per_launch=K, use_graph=True)
    if ring: ex.enable_history(ring)
    ticks = 4096
    ex.invoke_batch(256)
    best = min(ex.invoke_batch(ticks).kernel_device_ms / ticks for _ in range(3))
    print(f"K={K} ring={ring}: {best*1e3:.3f} us/tick  {n/best/1e6:.2f} G entity-steps/s  write {200*n/best/1e6:.0f} GB/s")
    ex.close()
