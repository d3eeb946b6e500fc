"""A/B of whole-library builds on one box: python tools/variant_ab.py <lib.so> [<lib.so> ...]
Each library is measured in its own process (SIXDOF_LIBRARY), the libraries interleaved over ROUNDS rounds so that clock
drift hits all of them alike.  Prints the best K=1 launch (graph replay) and the best K=64 tick per library."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path.insert(0, %r)
import elodin_amd as ea
from elodin_amd import workloads
n = 65536
w = workloads.independent_bodies(n)
eff = workloads.gravity_torque_effectors(w["body_torque"])
out = []
for k, reps in ((1, 4096), (64, 64)):
    ex = ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], simulation_time_step=workloads.DT_120HZ, effectors=eff, use_graph=True, ticks_per_launch=k)
    ex.prepare(reps * k)
    ex.invoke_batch(reps * k // 4)
    best = min(ex.invoke_batch(reps * k).kernel_device_ms / (reps * k) for _ in range(5))
    out.append(best * 1e3)
    ex.close()
print("%%.4f %%.4f" %% tuple(out))
''' % ROOT
libs = sys.argv[1:]
rounds = int(os.environ.get("ROUNDS", "3"))
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, SIXDOF_LIBRARY=os.path.abspath(l))
        o = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
        if o.returncode:
            print(l, "FAILED", o.stderr[-400:])
            continue
        res[l].append(tuple(float(x) for x in o.stdout.split()[-2:]))
for l in libs:
    k1 = [a for a, _ in res[l]]
    k64 = [b for _, b in res[l]]
    print(f"{os.path.basename(l):28s} K=1 us/launch {' '.join(f'{x:.3f}' for x in k1)}  (best {min(k1):.3f})   K=64 us/tick {' '.join(f'{x:.4f}' for x in k64)}  (best {min(k64):.4f})")
