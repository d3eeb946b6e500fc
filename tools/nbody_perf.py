import sys, time, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import elodin_amd as ea
from elodin_amd import _lib as L
from tests.test_gpu_parity import _plummer, K_SQ, EPS_AU2
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
pos, vel, inertia = _plummer(n)
hip = ea.HipExec(pos, vel, inertia, simulation_time_step=3600.0,
                 effectors=[ea.Effector(L.EFF_ALLPAIRS_GRAVITY_SOFTENED, (K_SQ, EPS_AU2))])
hip.invoke_batch(3)
t = hip.invoke_batch(20)
ms = t.kernel_device_ms / 20
pairs = n * (n - 1) * 3
print(f"n={n} tick {ms:.3f} ms  pair-evals/s {pairs/ms*1e3:.3e}  (x21 instr = {pairs*21/ms*1e3/1e12:.2f} T f64-instr/s; peak 39.3)")
