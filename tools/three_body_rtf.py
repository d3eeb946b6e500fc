import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import elodin_amd as ea
from elodin_amd import _lib as L
G = 6.6743e-11
pos = np.array([[0, 0, 0, 1, 0.8920281421, 0, 0], [0, 0, 0, 1, -0.6628498947, 0, 0], [0, 0, 0, 1, -0.2291782474, 0, 0]], dtype=float)
vel = np.array([[0, 0, 0, 0, 0.9957939373, 0], [0, 0, 0, 0, -1.6191613336, 0], [0, 0, 0, 0, 0.6233673964, 0]], dtype=float)
inertia = np.tile([1 / G, 1 / G, 1 / G, 0, 0, 0, 1 / G], (3, 1))
edges = (np.array([1, 2, 1, 2, 3, 3], dtype=np.uint64), np.array([2, 1, 3, 3, 1, 2], dtype=np.uint64))
for K in (1, 10, 100, 1000):
    ex = ea.HipExec(pos, vel, inertia, entity_ids=[1, 2, 3], simulation_time_step=0.008333333,
                    effectors=[ea.Effector(L.EFF_EDGE_GRAVITY_NEWTON, (G,))], edges=edges, ticks_per_launch=K)
    ex.invoke_batch(1000)
    t = ex.invoke_batch(20000)
    us = t.kernel_invoke_ms / 20000 * 1e3
    print(f"three-body ticks_per_launch={K}: {us:.3f} us/tick wall -> real_time_factor {0.008333333 / (us * 1e-6):.0f}")
