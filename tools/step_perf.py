import sys, os, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import elodin_amd as ea
from elodin_amd import workloads
for n in [int(x) for x in sys.argv[1:]] or (65536, 1 << 22):
    w = workloads.independent_bodies(n)
    eff = workloads.gravity_torque_effectors(w["body_torque"])
    for nt in ("0", "1", "2"):
        os.environ["SIXDOF_STREAMING"] = nt
        ex = ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], simulation_time_step=workloads.DT_120HZ, effectors=eff, use_graph=True)
        reps = 2048 if n == 65536 else 64
        ex.invoke_batch(reps // 8)
        best = min(ex.invoke_batch(reps).kernel_device_ms / reps for _ in range(3))
        print(f"n={n} nt={nt} {best*1e3:.2f} us/launch  {384*n/best/1e6:.0f} GB/s")
        ex.close()
