#!/usr/bin/env python3
"""The 35-body world (one wavefront per world, four 34-trip counted loops, 254 registers) at 1, 2 and 4 waves per SIMD: does a second
wave hide the table-load / ds_bpermute latency a trip opens with?  us per tick and pair evaluations per second."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
import elodin_amd as ea
from elodin_amd import _lib as L
from elodin_amd import dsl, workloads
from elodin_amd import stablehlo as sh
from tests.golden import hlo_world_builder as hb

nb = 35
rng = np.random.default_rng(nb)
cpos = np.concatenate([np.tile([0, 0, 0, 1.0], (nb, 1)), rng.normal(size=(nb, 3)) * 3], axis=1)
cvel = np.concatenate([np.zeros((nb, 3)), rng.normal(size=(nb, 3)) * 1e-3], axis=1)
cm = rng.uniform(1e-6, 1e-3, nb)
cin = np.concatenate([np.tile(cm[:, None], (1, 3)), np.zeros((nb, 3)), cm[:, None]], axis=1)
text, slots = hb.nbody_world(nb, 2.9591220828e-4, 1e-6)
system, man = sh.world_system(text, slots, mode="auto")
S = man["rows_per_world"]
for worlds in (512, 1024, 2048, 4096, 8192):
    rows = S * worlds
    w = workloads.independent_bodies(rows)

    def lay(a, fill):
        o = np.tile(np.asarray(fill, dtype=np.float64), (rows, 1))
        for i in range(nb):
            o[i::S] = a[i]
        return o
    cols = {"hlo_tick": np.zeros((rows, 1)), "hlo_simulation_time_step": np.full((rows, 1), 0.5), "hlo_world_pos": lay(cpos, [0, 0, 0, 1.0, 0, 0, 0]),
            "hlo_world_vel": lay(cvel, np.zeros(6)), "hlo_inertia": lay(cin, np.ones(7)), "hlo_world_accel": np.zeros((rows, 6)), "hlo_force": np.zeros((rows, 6))}
    ex = ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], integrator=L.INTEGRATOR_NONE, effectors=dsl.Program([system], dsl.Pipe([]), []),
                    columns=cols, ticks_per_launch=50)
    ex.invoke_batch(50)
    t = min(ex.invoke_batch(200).kernel_device_ms for _ in range(3))
    ex.close()
    print(f"{worlds:5d} worlds ({worlds / 1024:.1f} waves per SIMD): {t / 200 * 1e3:7.2f} us per tick, {4.0 * nb * (nb - 1) * worlds * 200 / (t * 1e-3):.3e} pair evaluations/s")
