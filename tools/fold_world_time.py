#!/usr/bin/env python3
"""Tick time of whole-world StableHLO ticks run as fold stages (stablehlo.world_program): n-body worlds of 80 and 256 bodies, one
world and a Monte-Carlo of them.   python tools/fold_world_time.py   (needs a GPU)"""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import elodin_amd as ea
from elodin_amd import _lib as L
from elodin_amd import stablehlo as sh
from tests.golden import hlo_world_builder as hb

out = {}
for nb, worlds, wave in ((80, 1, True), (80, 64, True), (256, 1, True), (256, 16, True), (256, 1, False), (1024, 1, True)):
    t0 = time.perf_counter()
    text, slots = hb.nbody_world(nb, 2.9591220828e-4, 1e-6)
    prog, manifest, edges = sh.world_program(text, slots, wave_folds=wave)
    rows = nb * worlds
    rng = np.random.default_rng(nb)
    pos = np.concatenate([np.tile([0, 0, 0, 1.0], (rows, 1)), rng.normal(size=(rows, 3)) * 3], axis=1)
    vel = np.concatenate([np.zeros((rows, 3)), rng.normal(size=(rows, 3)) * 1e-3], axis=1)
    m = rng.uniform(1e-6, 1e-3, rows)
    inertia = np.concatenate([np.tile(m[:, None], (1, 3)), np.zeros((rows, 3)), m[:, None]], axis=1)
    cols = {c["column"]: np.zeros((rows, c["width"])) for c in manifest["columns"]}
    cols["hlo_simulation_time_step"][:] = 0.5
    cols["hlo_world_pos"], cols["hlo_world_vel"], cols["hlo_inertia"] = pos, vel, inertia
    ids = np.arange(1, rows + 1, dtype=np.uint64)
    ex = ea.HipExec(np.tile([0, 0, 0, 1.0, 0, 0, 0], (rows, 1)), np.zeros((rows, 6)), np.ones((rows, 7)), entity_ids=ids, integrator=L.INTEGRATOR_NONE,
                    effectors=prog, columns=cols, graph_replicas=(worlds, nb) if worlds > 1 else None,
                    graph_edges=sh.edges_as_entity_ids(edges, ids))
    build_s = time.perf_counter() - t0
    ex.invoke_batch(5)
    tm = ex.invoke_batch(50)
    us = tm.kernel_device_ms / 50 * 1e3
    # the module carries its own tick column, so the launches of a batch are identical: with SIXDOF_FLAG_USE_GRAPH the 13-launch chain
    # of a tick replays from a captured graph like the hand-written kernel's
    ex.set_flags(L.FLAG_USE_GRAPH)
    ex.prepare(64)
    ex.invoke_batch(64)
    tg = ex.invoke_batch(64)
    us_graph, graph_launches = tg.kernel_device_ms / 64 * 1e3, int(tg.graph_launches)
    ex.close()
    out[f"{nb}_bodies_x_{worlds}_worlds" + ("" if wave else "_sequential_fold")] = {"us_per_tick": round(us, 2), "us_per_tick_graph_replay": round(us_graph, 2), "graph_launches_of_64": graph_launches, "launches_per_tick": int(tm.launches // 50), "pair_evals_per_s": round(4.0 * nb * (nb - 1) * worlds / us * 1e6, 1),
                                              "build_seconds_incl_trace_and_hipcc_or_cache": round(build_s, 2)}
print(json.dumps(out, indent=1))
