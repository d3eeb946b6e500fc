import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import elodin_amd as ea
from elodin_amd import _lib as L, workloads
from oracle import oracle as orc
from tests import parity
n, ticks = 65536, 10000
w = workloads.independent_bodies(n)
eff = workloads.gravity_torque_effectors(w["body_torque"])
hip = ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], entity_ids=w["entity_ids"], simulation_time_step=workloads.DT_120HZ, effectors=eff, ticks_per_launch=100)
ref = orc.OracleWorld(w["world_pos"], w["world_vel"], w["inertia"], simulation_time_step=workloads.DT_120HZ, ops=parity.to_oracle_ops(eff))
th = len(os.sched_getaffinity(0))
for cp in (1000, 2500, 5000, 10000):
    t=time.time(); hip.run(cp - hip.tick); ref.step(cp - ref.tick, threads=th)
    print(cp, {k: f"{v:.2e}" for k, v in parity.state_errors(hip, ref).items()}, f"{time.time()-t:.1f}s", flush=True)
