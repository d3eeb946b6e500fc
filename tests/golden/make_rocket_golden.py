#!/usr/bin/env python3
"""Golden rows of the reference's rocket example (examples/rocket/main.py: thirteen @el.map systems piped in front of
six_dof(RK4) with three effectors, a PID loop through a 480 x 3 sample window) from its CI baseline
scripts/ci/baseline/rocket-csv/*.csv: ticks 0..100 of every component column.  The window column is kept as its final
row set only (every row of it is a v_rel_accel sample that the v_rel_accel column already holds tick by tick).
Run in the build container:  python tests/golden/make_rocket_golden.py"""
import sys as _sys
_sys.dont_write_bytecode = True      # the reference checkout is read-only: no __pycache__ next to what is imported from it
import csv
import json
from pathlib import Path

SRC = Path("/root/reference/scripts/ci/baseline/rocket-csv")
OUT = Path(__file__).with_name("rocket.json")
doc = {"source": "scripts/ci/baseline/rocket-csv", "rows": {}}
for f in sorted(SRC.glob("rocket.*.csv")):
    comp = f.name[len("rocket."):-len(".csv")]
    rows = [[float(x) for x in r[1:]] for r in list(csv.reader(f.open()))[1:]]
    if comp == "v_rel_accel_buffer":
        doc["v_rel_accel_buffer_final"] = rows[-1]
        doc["v_rel_accel_buffer_nonzero_rows_per_tick"] = [sum(1 for k in range(0, len(r), 3) if any(r[k:k + 3])) for r in rows]
        continue
    doc["rows"][comp] = rows
doc["tick"] = [int(float(r[1])) for r in list(csv.reader((SRC / "globals.tick.csv").open()))[1:]]
doc["simulation_time_step"] = float(list(csv.reader((SRC / "globals.simulation_time_step.csv").open()))[1][1])
OUT.write_text(json.dumps(doc))
print(OUT, OUT.stat().st_size, {k: (len(v), len(v[0])) for k, v in doc["rows"].items()}, doc["tick"][:3], doc["tick"][-1], doc["simulation_time_step"])
