#!/usr/bin/env python3
"""Fixture for the solar-system sanity check of SURVEY 8(d) config 3: day-0 state of the nine bodies of
examples/n-body/planets_truth.csv (JPL-derived daily ephemeris, AU and AU/day) and their truth positions every 10th day
over the first 840 days (the reference's accuracy report runs 20,000 one-hour ticks = 833 days).  Masses are the
example's BODY_META (examples/n-body/sim.py:57-67).  Run in the build container (needs /root/reference):
    python tests/golden/make_solar_golden.py"""
import sys as _sys
_sys.dont_write_bytecode = True      # the reference checkout is read-only: no __pycache__ next to what is imported from it
import csv
import json
from pathlib import Path

SRC = Path("/root/reference/examples/n-body/planets_truth.csv")
OUT = Path(__file__).with_name("solar_system.json")
MASS = {"mercury": 1.6605e-7, "venus": 2.4478e-6, "earth": 3.0035e-6, "mars": 3.2272e-7, "jupiter": 9.5459e-4,
        "saturn": 2.8588e-4, "uranus": 4.3662e-5, "neptune": 5.1514e-5, "pluto": 6.55e-9}

rows = list(csv.DictReader(SRC.open()))
bodies, pos, vel = [], {}, {}
for r in rows:
    name = r["name"].split(maxsplit=1)[-1].strip().lower()
    if name not in pos:
        bodies.append(name)
        pos[name], vel[name] = [], []
    pos[name].append([float(r["x_au"]), float(r["y_au"]), float(r["z_au"])])
    vel[name].append([float(r["vx_au_per_day"]), float(r["vy_au_per_day"]), float(r["vz_au_per_day"])])
days = list(range(0, 841, 10))
doc = {"source": "examples/n-body/planets_truth.csv", "date0": rows[0]["date"], "bodies": bodies,
       "mass_solar": [MASS[b] for b in bodies], "pos0_au": [pos[b][0] for b in bodies],
       "vel0_au_per_day": [vel[b][0] for b in bodies], "days": days,
       "truth_au": [[pos[b][d] for d in days] for b in bodies]}
OUT.write_text(json.dumps(doc))
print(OUT, OUT.stat().st_size, "bytes", bodies)
