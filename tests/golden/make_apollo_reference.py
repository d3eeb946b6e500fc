#!/usr/bin/env python3
"""Write elodin_amd/data/apollo_reference.csv: the smoothed Apollo 11 descent reference profile that the
reference example shares between its sim and its guidance controller, produced by the reference's OWN code
(examples/apollo-lander/reference.py build_reference(), stdlib-only) from its vendored telemetry CSVs.
Values are written with repr() so they round-trip exactly.  Build container only."""
import sys as _sys
_sys.dont_write_bytecode = True      # the reference checkout is read-only: no __pycache__ next to what is imported from it
import importlib.util
import sys
from pathlib import Path

REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
OUT = Path(__file__).resolve().parents[2] / "elodin_amd" / "data" / "apollo_reference.csv"
spec = importlib.util.spec_from_file_location("apollo_reference", REF / "examples/apollo-lander/reference.py")
mod = importlib.util.module_from_spec(spec)
sys.modules["apollo_reference"] = mod
spec.loader.exec_module(mod)
r = mod.build_reference()
cols = [("time_s", r.time_s), ("altitude_m", r.altitude_m), ("descent_rate_mps", r.descent_rate_mps),
        ("pitch_deg", r.pitch_deg), ("slant_range_m", r.slant_range_m),
        ("horizontal_speed_mps", r.horizontal_speed_mps), ("downrange_m", r.downrange_m)]
with open(OUT, "w") as f:
    f.write(",".join(c for c, _ in cols) + "\n")
    for i in range(len(r.time_s)):
        f.write(",".join(repr(float(v[i])) for _, v in cols) + "\n")
print(f"{len(r.time_s)} rows, t_end = {r.t_end} -> {OUT}")
