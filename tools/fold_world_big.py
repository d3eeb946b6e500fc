#!/usr/bin/env python3
"""A whole-world n-body tick of N bodies (default 2,048: a 232 MB module, 4.2 M edges per scan) as the reference would dump it, through
stablehlo.world_program (the scans over the edge slot as fold stages over the implicit complete graph), a few ticks on the GPU against
the CPU oracle's sequential softened fold.   python tools/fold_world_big.py [N] [ticks] [wave|sequential] [reference|relaxed]      (needs a GPU; minutes of host time)"""
import json
import resource
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import elodin_amd as ea
from elodin_amd import _lib as L
from elodin_amd import stablehlo as sh
from oracle import oracle as orc
from tests.golden import hlo_world_builder as hb

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 3
wave = (sys.argv[3] if len(sys.argv) > 3 else "wave") != "sequential"      # a wave per source (default) or the one-lane sequential fold
arith = sys.argv[4] if len(sys.argv) > 4 else "reference"      # world_program(arith=): the reference's arithmetic or dsl.relaxed_arithmetic
K, EPS, DT = 2.9591220828e-4, 1e-6, 0.5
out = {"bodies": nb, "ticks": ticks, "fold": "a wave per source" if wave else "sequential, one lane per source", "arith": arith}
t0 = time.perf_counter()
text, slots = hb.nbody_world(nb, K, EPS)
out["module_text_MB"] = round(len(text) / 1e6, 1)
out["module_seconds"] = round(time.perf_counter() - t0, 1)
t0 = time.perf_counter()
prog, manifest, edges = sh.world_program(text, slots, wave_folds=wave, arith=arith)
del text
out["world_program_seconds"] = round(time.perf_counter() - t0, 1)
out["fold_stages"], out["edges_per_fold"], out["edges"] = manifest["fold_stages"], manifest["edges_per_fold"], [e[0] if isinstance(e[0], str) else "explicit" for e in edges.values()]
rng = np.random.default_rng(nb)
pos = np.concatenate([np.tile([0, 0, 0, 1.0], (nb, 1)), rng.normal(size=(nb, 3)) * 3], axis=1)
vel = np.concatenate([np.zeros((nb, 3)), rng.normal(size=(nb, 3)) * 1e-3], axis=1)
m = rng.uniform(1e-6, 1e-3, nb)
inertia = np.concatenate([np.tile(m[:, None], (1, 3)), np.zeros((nb, 3)), m[:, None]], axis=1)
cols = {c["column"]: np.zeros((nb, c["width"])) for c in manifest["columns"]}
cols["hlo_simulation_time_step"][:] = DT
cols["hlo_world_pos"], cols["hlo_world_vel"], cols["hlo_inertia"] = pos.copy(), vel.copy(), inertia.copy()
t0 = time.perf_counter()
hip = ea.HipExec(np.tile([0, 0, 0, 1.0, 0, 0, 0], (nb, 1)), np.zeros((nb, 6)), np.ones((nb, 7)), integrator=L.INTEGRATOR_NONE, effectors=prog, columns=cols,
                 graph_edges=sh.edges_as_entity_ids(edges, np.arange(1, nb + 1, dtype=np.uint64)))
out["trace_and_build_seconds"] = round(time.perf_counter() - t0, 1)
ref = orc.OracleWorld(pos, vel, inertia, simulation_time_step=DT, ops=[(orc.EFF_ALLPAIRS_GRAVITY_SOFTENED, (K, EPS), None)])
worst = 0.0
for r in range(1, ticks + 1):
    tm = hip.run(1)
    ref.step(1, threads=8)
    for c, a in (("world_pos", ref.world_pos), ("world_vel", ref.world_vel), ("world_accel", ref.world_accel), ("force", ref.force)):
        g = hip._aux["hlo_" + c]
        for sl in ((slice(0, 4), slice(4, 7)) if c == "world_pos" else (slice(0, 3), slice(3, 6))):
            scale = np.maximum(np.max(np.abs(a[:, sl]), axis=1, keepdims=True), 1e-300)
            worst = max(worst, float(np.max(np.abs(g[:, sl] - a[:, sl]) / scale)))
out["max_rel_err_vs_oracle"] = worst
tm = hip.invoke_batch(5)
out["us_per_tick"] = round(tm.kernel_device_ms / 5 * 1e3, 1)
out["tick_column"] = float(hip.download_column("hlo_tick")[0, 0]) if hasattr(hip, "download_column") else None
out["host_maxrss_GB"] = round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, 2)
hip.close()
print(json.dumps(out, indent=1))
sys.exit(0 if worst <= 1e-9 else 1)
