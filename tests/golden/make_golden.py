#!/usr/bin/env python3
"""Regenerate tests/golden/*.csv from the reference's own regression baselines.

Source (reference checkout, read-only): scripts/ci/baseline/{three-body,ball,cube-sat}-csv/ —
the CSVs `scripts/ci/regress.sh` gates CI on (tolerance 1e-4; they carry 17 significant
digits).  This script only re-packs them: one file per example, wall-clock `time` column
dropped (the reference comparator ignores it too, scripts/ci/compare_baseline_csv.py:21),
columns joined side by side, row r = state after r ticks.  Decimal strings are copied
verbatim so no precision is lost.

Run in the build container only (the GPU box has no /root/reference):
    python tests/golden/make_golden.py [/root/reference]
"""
import sys as _sys
_sys.dont_write_bytecode = True      # the reference checkout is read-only: no __pycache__ next to what is imported from it
import csv
import sys
from pathlib import Path

REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
OUT = Path(__file__).resolve().parent

EXAMPLES = {
    "three_body": ("scripts/ci/baseline/three-body-csv",
                   [f"{e}.{c}" for e in "abc" for c in ("world_pos", "world_vel", "world_accel", "force", "inertia")]
                   + [f"{e}.gravity_edge" for e in ("a_to_b", "b_to_a", "a_to_c", "b_to_c", "c_to_a", "c_to_b")]
                   + ["globals.tick", "globals.simulation_time_step"]),
    "ball": ("scripts/ci/baseline/ball-csv",
             [f"ball.{c}" for c in ("world_pos", "world_vel", "world_accel", "force", "inertia", "wind", "seed")]
             + ["globals.tick", "globals.simulation_time_step"]),
    # SemiImplicit run (examples/cube-sat/main.py:699-710) with non-zero torque and a non-uniform inertia diagonal:
    # the Body columns of the satellite and of the (purely rotating) earth.  Used teacher-forced — the recorded
    # `force` row r is fed to a one-tick step from rows r-1 — so none of the example's effectors is restated.
    # drone-csv is NOT usable that way: it records every 3rd tick and each tick chains three six_dof sub-steps
    # (examples/drone/sim.py:173-208), so the forces between two rows are not in the data.
    "cube_sat": ("scripts/ci/baseline/cube-sat-csv",
                 [f"{e}.{c}" for e in ("ore_sat", "earth")
                  for c in ("world_pos", "world_vel", "world_accel", "force", "inertia")]
                 + ["globals.tick", "globals.simulation_time_step"]),
}


def main():
    for name, (rel, stems) in EXAMPLES.items():
        header, cols, nrows = [], [], None
        for stem in stems:
            with open(REF / rel / f"{stem}.csv", newline="") as f:
                rows = list(csv.reader(f))
            assert rows[0][0] == "time"
            nrows = nrows or len(rows) - 1
            assert len(rows) - 1 == nrows, (stem, len(rows))
            for j, h in enumerate(rows[0][1:], start=1):
                header.append(h)
                cols.append([r[j] for r in rows[1:]])
        with open(OUT / f"{name}.csv", "w", newline="") as f:
            wr = csv.writer(f)
            wr.writerow(["row"] + header)
            for r in range(nrows):
                wr.writerow([r] + [c[r] for c in cols])
        print(f"{name}: {nrows} rows x {len(header)} columns -> {OUT / (name + '.csv')}")


if __name__ == "__main__":
    main()
