#!/usr/bin/env python3
"""The whole-world StableHLO ticks of bench.py's `world_module` leg, a few launches each, for a rocprofv3 --pmc pass
(profiles/collect_world.sh): the three-body world module with one lane per WORLD and with one lane per ENTITY (lane exchange), and
the 10-body solar system in lane mode.  All three run the SAME kernel template (sixdof_step_kernel<double, 2, PipeCustom, ...>), so
each gets its own grid size and writes `grid -> program` next to the counters.  TICKS_PER_LAUNCH is what the summary divides by."""
import json
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
from tests import solar_util as su

TICKS_PER_LAUNCH = 100
keys = {}


def run(system, cols, rows, key):
    w = workloads.independent_bodies(rows)
    ex = ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], integrator=L.INTEGRATOR_NONE, effectors=dsl.Program([system], dsl.Pipe([]), []),
                    columns=cols, ticks_per_launch=TICKS_PER_LAUNCH)
    ex.invoke_batch(4 * TICKS_PER_LAUNCH)
    ex.close()
    keys[str(rows)] = key


g_pos = np.array([0, 0, 0, 1, 0.8920281421, 0, 0, 0, 0, 0, 1, -0.6628498947, 0, 0, 0, 0, 0, 1, -0.2291782474, 0, 0.0])
g_vel = np.array([0, 0, 0, 0, 0.9957939373, 0, 0, 0, 0, 0, -1.6191613336, 0, 0, 0, 0, 0, 0.6233673964, 0.0])
m = 1.0 / 6.6743e-11
g_in = np.tile([m, m, m, 0, 0, 0, m], 3)
text, slots = hb.three_body_world()
system, manifest = sh.world_system(text, slots, mode="world")
worlds = 65536
run(system, {"hlo_tick": np.zeros((worlds, 1)), "hlo_simulation_time_step": np.full((worlds, 1), 0.008333333), "hlo_world_pos": np.tile(g_pos, (worlds, 1)),
             "hlo_world_vel": np.tile(g_vel, (worlds, 1)), "hlo_world_accel": np.zeros((worlds, 18)), "hlo_force": np.zeros((worlds, 18)),
             "hlo_inertia": np.tile(g_in, (worlds, 1))}, worlds, "three_body_world_mode")

lsys, lman = sh.world_system(text, slots, mode="auto")
S = lman["rows_per_world"]
rows = S * 8192                                                    # 32,768 rows: told apart from the other two by the grid


def lay(vals, width, fill):
    a = np.tile(np.asarray(fill, dtype=np.float64), (rows, 1))
    for i in range(3):
        a[i::S] = vals[i * width:(i + 1) * width]
    return a


run(lsys, {"hlo_tick": np.zeros((rows, 1)), "hlo_simulation_time_step": np.full((rows, 1), 0.008333333), "hlo_world_pos": lay(g_pos, 7, [0, 0, 0, 1.0, 0, 0, 0]),
           "hlo_world_vel": lay(g_vel, 6, np.zeros(6)), "hlo_inertia": lay(g_in, 7, np.ones(7)), "hlo_world_accel": np.zeros((rows, 6)),
           "hlo_force": np.zeros((rows, 6))}, rows, "three_body_lane_mode")

_, spos, svel, sin_ = su.load()
nb = spos.shape[0]
ntext, nslots = hb.nbody_world(nb, su.K_SQUARED, su.SOFTENING_AU2)
nsys, nman = sh.world_system(ntext, nslots, mode="auto")
S = nman["rows_per_world"]
rows = S * 8192                                                    # 131,072 rows


def nlay(a, fill):
    o = np.tile(np.asarray(fill, dtype=np.float64), (rows, 1))
    for i in range(nb):
        o[i::S] = a[i]
    return o


run(nsys, {"hlo_tick": np.zeros((rows, 1)), "hlo_simulation_time_step": np.full((rows, 1), su.DT), "hlo_world_pos": nlay(spos, [0, 0, 0, 1.0, 0, 0, 0]),
           "hlo_world_vel": nlay(svel, np.zeros(6)), "hlo_inertia": nlay(sin_, np.ones(7)), "hlo_world_accel": np.zeros((rows, 6)),
           "hlo_force": np.zeros((rows, 6))}, rows, "solar_system_10_bodies_lane_mode")
nb = 35                                                            # a 35-body cluster: a world = one wavefront, four 34-trip counted loops
rng = np.random.default_rng(nb)
cpos = np.concatenate([np.tile([0, 0, 0, 1.0], (nb, 1)), rng.normal(size=(nb, 3)) * 3], axis=1)
cvel = np.concatenate([np.zeros((nb, 3)), rng.normal(size=(nb, 3)) * 1e-3], axis=1)
cm = rng.uniform(1e-6, 1e-3, nb)
cin = np.concatenate([np.tile(cm[:, None], (1, 3)), np.zeros((nb, 3)), cm[:, None]], axis=1)
ctext, cslots = hb.nbody_world(nb, 2.9591220828e-4, 1e-6)
csys, cman = sh.world_system(ctext, cslots, mode="auto")
S = cman["rows_per_world"]
rows = S * 256                                                     # 16,384 rows
spos, svel, sin_ = cpos, cvel, cin
run(csys, {"hlo_tick": np.zeros((rows, 1)), "hlo_simulation_time_step": np.full((rows, 1), 0.5), "hlo_world_pos": nlay(spos, [0, 0, 0, 1.0, 0, 0, 0]),
           "hlo_world_vel": nlay(svel, np.zeros(6)), "hlo_inertia": nlay(sin_, np.ones(7)), "hlo_world_accel": np.zeros((rows, 6)),
           "hlo_force": np.zeros((rows, 6))}, rows, "cluster_35_bodies_lane_mode")
# BASELINE configs[1] as a whole-world module ([n, 7] / [n, 6] tensors, no edges), one lane per entity: the instruction count that
# explains its K = 1 time against the hand-written kernel's (the module's arithmetic is the reference's, operation for operation)
rows = 262144                                                      # a grid of its own
itext, islots = hb.independent_bodies_world(rows)
isys, iman = sh.world_system(itext, islots, mode="lane")
iw = workloads.independent_bodies(rows)
run(isys, {"hlo_tick": np.zeros((rows, 1)), "hlo_simulation_time_step": np.full((rows, 1), workloads.DT_120HZ), "hlo_world_pos": iw["world_pos"].copy(),
           "hlo_world_vel": iw["world_vel"].copy(), "hlo_world_accel": np.zeros((rows, 6)), "hlo_force": np.zeros((rows, 6)), "hlo_inertia": iw["inertia"].copy(),
           "hlo_torque": iw["body_torque"].copy()}, rows, "independent_bodies_lane_mode")
# ... and the same module under world_system(arith="relaxed", one_world=True): finite values assumed, shared reciprocals, contraction
rows = 524288                                                      # a grid of its own
rtext, rslots = hb.independent_bodies_world(rows)
rsys, rman = sh.world_system(rtext, rslots, mode="lane", arith="relaxed", one_world=True)
rw = workloads.independent_bodies(rows)
run(rsys, {"hlo_tick": np.zeros((rows, 1)), "hlo_simulation_time_step": np.full((rows, 1), workloads.DT_120HZ), "hlo_world_pos": rw["world_pos"].copy(),
           "hlo_world_vel": rw["world_vel"].copy(), "hlo_world_accel": np.zeros((rows, 6)), "hlo_force": np.zeros((rows, 6)), "hlo_inertia": rw["inertia"].copy(),
           "hlo_torque": rw["body_torque"].copy()}, rows, "independent_bodies_lane_mode_relaxed")
out = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "gpurun_out" / "world_keys.json"
out.write_text(json.dumps({"ticks_per_launch": TICKS_PER_LAUNCH, "grids": keys}))
print("done", keys)
