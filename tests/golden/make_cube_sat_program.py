#!/usr/bin/env python3
"""The reference's cube-sat example compiled by THIS repo's front end, frozen for the GPU box (the counterpart of
make_drone_program.py): examples/cube-sat/main.py imported UNMODIFIED under elodin_amd.compat with a zero gravity field in the
place of its EGM08 evaluation (tests/cube_sat_util.py says why and what that leaves pinned), resolved like World.build resolves
it — 11 rows (satellite, Earth, three wheels, six sun sensors), four edge folds as links of one launch chain — and written to
tests/golden/cube_sat_program.json: the generated HIP source (this repo's compiler output), its column table, the spawned
columns and the entity -> row map.   python tests/golden/make_cube_sat_program.py"""
import sys as _sys
_sys.dont_write_bytecode = True      # the reference checkout is read-only: no __pycache__ next to what is imported from it
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.setrecursionlimit(20000)

import numpy as np  # noqa: E402

import elodin_amd.compat as compat  # noqa: E402
from elodin_amd import codegen  # noqa: E402
from tests import test_compat_reference_scripts as T  # noqa: E402

ref, plan, row_of = T.load_cube_sat(compat)
tp = plan["effectors"].trace()
n = len(plan["row_ids"])
cols = plan["columns"]
codegen.build(tp, "float64", plan["integrator"])            # settles on the first variant that fits a wave's registers
doc = {
    "variant": codegen.last_variant[0],
    "source": codegen.generate_variant(tp, codegen.last_variant[0], "float64", plan["integrator"]),
    "columns": [[name, w] for name, w in tp.columns], "mats": {k: list(v) for k, v in tp.table.mats.items()},
    "integrator": plan["integrator"], "simulation_time_step": plan["dt"], "row_of": row_of,
    "entity_ids": [int(e) for e in plan["row_ids"]],
    "body": {k: np.asarray(v, dtype=np.float64).tolist() for k, v in plan["body"].items()},
    "initial": {name: (np.asarray(cols[name], dtype=np.float64).reshape(n, -1) if name in cols else np.zeros((n, w))).tolist()
                for name, w in tp.columns},
    "folds": [s.name for s in tp.fold_stages],
}
out = ROOT / "tests" / "golden" / "cube_sat_program.json"
out.write_text(json.dumps(doc))
print(out, out.stat().st_size, "bytes; variant", doc["variant"] + ";", len(doc["columns"]), "columns,", doc["source"].count("\n"), "source lines, folds", doc["folds"])
