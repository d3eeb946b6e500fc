#!/usr/bin/env python3
"""elodin_amd/data/falcon9_crs12_profile.csv: the recorded CRS-12 ascent profile as the Falcon 9 flight software resamples
it at start-up (controller/src/profile.rs), computed by the PRODUCT's restatement (models/falcon9.resample_profile) from the
reference's data file examples/falcon9/data/crs12/stage1_raw.json (the file ELODIN_F9_PROFILE names, main.py:193-199).
Build container only.  tests/test_falcon9_fsw_oracle.py checks the table against the C oracle's own resampling."""
import sys as _sys
_sys.dont_write_bytecode = True      # the reference checkout is read-only: no __pycache__ next to what is imported from it
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")

import numpy as np  # noqa: E402

from elodin_amd.models import falcon9 as f9  # noqa: E402

for mission in ("crs12",):
    raw = json.loads((REF / "examples" / "falcon9" / "data" / mission / "stage1_raw.json").read_text())
    cols = f9.resample_profile(raw["time"], raw["velocity"], raw["altitude"])
    out = ROOT / "elodin_amd" / "data" / f"falcon9_{mission}_profile.csv"
    with open(out, "w") as f:
        f.write("time_s,speed_mps,alt_m,vspeed_mps\n")
        for row in zip(*cols):
            f.write(",".join(repr(float(v)) for v in row) + "\n")
    print(out, len(cols[0]), "rows")
