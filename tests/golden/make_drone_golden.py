#!/usr/bin/env python3
"""Golden rows of the reference's drone example (examples/drone) from its CI baseline scripts/ci/baseline/drone-csv/*.csv:
the 35 recorded rows (ticks 0, 3, 6, ..., 99, 100 — the example commits telemetry every third tick) of every component of the
`drone` entity, verbatim (f64 repr kept).  Run in the build container:  python tests/golden/make_drone_golden.py"""
import sys as _sys
_sys.dont_write_bytecode = True      # the reference checkout is read-only: no __pycache__ next to what is imported from it
import csv
import json
from pathlib import Path

SRC = Path("/root/reference/scripts/ci/baseline/drone-csv")
OUT = Path(__file__).with_name("drone.json")
doc = {"source": "scripts/ci/baseline/drone-csv", "rows": {}}
for f in sorted(SRC.glob("drone.*.csv")):
    rows = list(csv.reader(f.open()))[1:]
    doc["rows"][f.name[len("drone."):-len(".csv")]] = [[float(x) for x in r[1:]] for r in rows]
doc["tick"] = [int(r[1]) for r in list(csv.reader((SRC / "globals.tick.csv").open()))[1:]]
doc["simulation_time_step"] = float(list(csv.reader((SRC / "globals.simulation_time_step.csv").open()))[1][1])
OUT.write_text(json.dumps(doc))
print(OUT, OUT.stat().st_size, len(doc["tick"]), "rows of", len(doc["rows"]), "components; ticks", doc["tick"][:4], "...", doc["tick"][-2:])
