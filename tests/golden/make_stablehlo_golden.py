#!/usr/bin/env python3
"""Golden rows of the reference's StableHLO coverage example (examples/stablehlo/sim.py) from its CI baseline
scripts/ci/baseline/stablehlo/*.csv: ticks 0..100 of the seven float-valued component columns and the int64 bitwise column.  Run in the build container:  python tests/golden/make_stablehlo_golden.py"""
import sys as _sys
_sys.dont_write_bytecode = True      # the reference checkout is read-only: no __pycache__ next to what is imported from it
import csv
import json
from pathlib import Path

SRC = Path("/root/reference/scripts/ci/baseline/stablehlo")
OUT = Path(__file__).with_name("stablehlo.json")
FILES = {"math_state": "math.math_state.csv", "sort_state": "sorter.sort_state.csv", "shape_state": "shaper.shape_state.csv",
         "control_state": "ctrl.control_state.csv", "linalg_state": "linalg.linalg_state.csv",
         "convert_state": "cvt.convert_state.csv", "linalg2_state": "linalg2.linalg2_state.csv",
         "bitwise_state": "bits.bitwise_state.csv"}
doc = {"source": "scripts/ci/baseline/stablehlo", "rows": {}}
for comp, fn in FILES.items():
    rows = list(csv.reader((SRC / fn).open()))[1:]
    doc["rows"][comp] = [[float(x) for x in r[1:]] for r in rows]
doc["simulation_time_step"] = float(list(csv.reader((SRC / "globals.simulation_time_step.csv").open()))[1][1])
OUT.write_text(json.dumps(doc))
print(OUT, OUT.stat().st_size, {k: len(v) for k, v in doc["rows"].items()})
