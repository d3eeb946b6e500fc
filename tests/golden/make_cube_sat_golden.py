#!/usr/bin/env python3
"""Every recorded column of the reference's cube-sat example (examples/cube-sat/main.py: MEKF attitude filter, LQR pointing
law, three reaction wheels, six sun sensors, semi-implicit six_dof at 120 Hz) from its CI baseline
scripts/ci/baseline/cube-sat-csv/*.csv: ticks 0..100 of all 11 entities, verbatim (f64 repr kept), keyed entity -> component.
tests/golden/cube_sat.csv (make_golden.py) holds the satellite's Body columns of the same run for the integrator pin; this
file is the whole world, for the closed attitude loop (tests/test_compat_reference_scripts.py).
Run in the build container:  python tests/golden/make_cube_sat_golden.py"""
import sys as _sys
_sys.dont_write_bytecode = True      # the reference checkout is read-only: no __pycache__ next to what is imported from it
import csv
import json
from pathlib import Path

SRC = Path("/root/reference/scripts/ci/baseline/cube-sat-csv")
OUT = Path(__file__).with_name("cube_sat_world.json")
doc = {"source": "scripts/ci/baseline/cube-sat-csv", "entities": {}}
for f in sorted(SRC.glob("*.csv")):
    entity, comp = f.name[:-len(".csv")].split(".", 1)
    if entity == "globals" or comp.endswith("_edge"):
        continue
    rows = list(csv.reader(f.open()))[1:]
    doc["entities"].setdefault(entity, {})[comp] = [[float(x) for x in r[1:]] for r in rows]
doc["tick"] = [int(r[1]) for r in list(csv.reader((SRC / "globals.tick.csv").open()))[1:]]
doc["simulation_time_step"] = float(list(csv.reader((SRC / "globals.simulation_time_step.csv").open()))[1][1])
OUT.write_text(json.dumps(doc, separators=(",", ":")))
print(OUT, OUT.stat().st_size, "bytes;", len(doc["tick"]), "rows;", {e: len(c) for e, c in doc["entities"].items()})
