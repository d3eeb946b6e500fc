#!/usr/bin/env python3
"""BASELINE configs[1] as a whole-world StableHLO module, one lane per entity, 1 tick per launch: eager launches vs graph replay
(what the headline line uses), next to the hand-written kernel on the same world.  us per tick = device time of the batch / ticks."""
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
system, manifest = sh.world_system(text, slots, mode="lane")
w = workloads.independent_bodies(n)
for use_graph in (False, True):
    cols = {"hlo_tick": np.zeros((n, 1)), "hlo_simulation_time_step": np.full((n, 1), workloads.DT_120HZ), "hlo_world_pos": w["world_pos"].copy(),
            "hlo_world_vel": w["world_vel"].copy(), "hlo_world_accel": np.zeros((n, 6)), "hlo_force": np.zeros((n, 6)), "hlo_inertia": w["inertia"].copy(),
            "hlo_torque": w["body_torque"].copy()}
    ex = ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], integrator=L.INTEGRATOR_NONE, effectors=dsl.Program([system], dsl.Pipe([]), []),
                    columns=cols, use_graph=use_graph)
    if use_graph:
        ex.prepare(1024)
    ex.invoke_batch(1024)
    best = min(ex.invoke_batch(1024).kernel_device_ms for _ in range(5))
    t = ex.invoke_batch(1024)
    print(f"whole-world module, use_graph={use_graph}: {best / 1024 * 1e3:.3f} us per tick (best of 5), graph launches {t.graph_launches}/{t.launches}")
    ex.close()
eff = workloads.gravity_torque_effectors(w["body_torque"])
for use_graph in (False, True):
    ex = ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], entity_ids=w["entity_ids"], simulation_time_step=workloads.DT_120HZ, effectors=eff, use_graph=use_graph)
    if use_graph:
        ex.prepare(1024)
    ex.invoke_batch(1024)
    best = min(ex.invoke_batch(1024).kernel_device_ms for _ in range(5))
    print(f"hand-written kernel, use_graph={use_graph}: {best / 1024 * 1e3:.3f} us per tick (best of 5)")
    ex.close()
