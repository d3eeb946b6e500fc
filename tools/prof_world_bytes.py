#!/usr/bin/env python3
"""BASELINE configs[1] as a whole-world StableHLO module (one lane per entity), ONE tick per launch, a few launches — for the two PMC
passes that count its HBM bytes (FETCH_SIZE, WRITE_SIZE: separate rocprofv3 runs):   tools/prof_world_bytes.sh
WORLD_ARITH=relaxed / WORLD_ONE_WORLD=1: the module under stablehlo.world_system(arith="relaxed", one_world=True)."""
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

n = 65536
text, slots = hb.independent_bodies_world(n)
import os
ARITH = os.environ.get("WORLD_ARITH", "reference")
system, manifest = sh.world_system(text, slots, mode="lane", arith=ARITH, one_world=os.environ.get("WORLD_ONE_WORLD", "") == "1")
w = workloads.independent_bodies(n)
cols = {"hlo_tick": np.zeros((n, 1)), "hlo_simulation_time_step": np.full((n, 1), workloads.DT_120HZ), "hlo_world_pos": w["world_pos"].copy(),
        "hlo_world_vel": w["world_vel"].copy(), "hlo_world_accel": np.zeros((n, 6)), "hlo_force": np.zeros((n, 6)), "hlo_inertia": w["inertia"].copy(),
        "hlo_torque": w["body_torque"].copy()}
ex = ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], integrator=L.INTEGRATOR_NONE, effectors=dsl.Program([system], dsl.Pipe([]), []), columns=cols)
ex.invoke_batch(32)
ex.close()
