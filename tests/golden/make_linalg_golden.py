#!/usr/bin/env python3
"""Golden rows of the reference's linalg example (examples/linalg/sim.py) from its CI baseline
scripts/ci/baseline/linalg/*.csv: ticks 0..100 of every component of its six entities (verbatim re-pack, f64 repr kept).
Run in the build container:  python tests/golden/make_linalg_golden.py"""
import sys as _sys
_sys.dont_write_bytecode = True      # the reference checkout is read-only: no __pycache__ next to what is imported from it
import csv
import json
from pathlib import Path

SRC = Path("/root/reference/scripts/ci/baseline/linalg")
OUT = Path(__file__).with_name("linalg.json")
FILES = {"kf3_state": "tracker3.kf3_state.csv", "kf3_cov": "tracker3.kf3_cov.csv", "kf3_info": "tracker3.kf3_info.csv",
         "ekf6_state": "tracker6.ekf6_state.csv", "ekf6_cov": "tracker6.ekf6_cov.csv", "ekf6_info": "tracker6.ekf6_info.csv",
         "sm2_state": "small2.sm2_state.csv", "sm2_cov": "small2.sm2_cov.csv", "mrhs_state": "mat_rhs.mrhs_state.csv",
         "mode_state": "mode_sel.mode_state.csv", "chol_res_norms": "chol_variants.chol_res_norms.csv"}
doc = {"source": "scripts/ci/baseline/linalg", "rows": {}}
for comp, fn in FILES.items():
    rows = list(csv.reader((SRC / fn).open()))[1:]
    doc["rows"][comp] = [[float(x) for x in r[1:]] for r in rows]
doc["simulation_time_step"] = float(list(csv.reader((SRC / "globals.simulation_time_step.csv").open()))[1][1])
OUT.write_text(json.dumps(doc))
print(OUT, OUT.stat().st_size, {k: (len(v), len(v[0])) for k, v in doc["rows"].items()})
