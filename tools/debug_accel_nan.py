import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import elodin_amd as ea
from elodin_amd import _lib as L, workloads
from oracle import oracle as orc
from tests import parity
n = 300
w = workloads.independent_bodies(n)
accel = np.zeros((n, 6)); accel[7, 4] = np.nan; accel[100, 1] = np.inf; accel[299, 5] = -np.inf
if len(sys.argv) > 1 and sys.argv[1] == "pair":
    eff = [ea.Effector(L.EFF_ALLPAIRS_GRAVITY_SOFTENED, (1e-3, 1e-2))]
    oops = [(orc.EFF_ALLPAIRS_GRAVITY_SOFTENED, (1e-3, 1e-2), None)]
else:
    eff = workloads.gravity_torque_effectors(w["body_torque"])
    oops = parity.to_oracle_ops(eff)
hip = ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], world_accel=accel, simulation_time_step=workloads.DT_120HZ, effectors=eff)
ref = orc.OracleWorld(w["world_pos"], w["world_vel"], w["inertia"], world_accel=accel, simulation_time_step=workloads.DT_120HZ, ops=oops)
for k in range(3):
    hip.run(1); ref.step(1)
    for f in parity.FIELDS:
        g, r = getattr(hip, f), getattr(ref, f)
        d = np.argwhere(np.isnan(g) != np.isnan(r))
        print(k, f, "n nan rows hip", len(set(np.argwhere(np.isnan(g))[:, 0].tolist())), "ref", len(set(np.argwhere(np.isnan(r))[:, 0].tolist())), "diff", d[:10].tolist())
        for row in sorted(set(d[:, 0].tolist()))[:4]:
            print("   row", row, "hip", g[row], "ref", r[row])
