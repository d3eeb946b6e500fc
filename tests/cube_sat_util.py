"""The reference's cube-sat example (examples/cube-sat/main.py) against its CI baseline (tests/golden/cube_sat_world.json <-
scripts/ci/baseline/cube-sat-csv, ticks 0..100 of all 11 entities): shared by the CPU walk of the unmodified script
(tests/test_compat_reference_scripts.py) and the GPU run of its generated kernel (tests/test_gpu_cube_sat.py).

What is closed and what is not.  The example's gravity is `elodin.egm08.EGM08(64)`, a degree-64 spherical-harmonic field whose
coefficient tables the reference downloads on first use (python/elodin/egm08.py:27-41): they are not in the checkout and there
is no network here, so the field cannot be evaluated.  Gravity acts on the satellite's TRANSLATION only (a force through the
centre of mass); everything else the example computes — attitude dynamics under the wheels' torques, sun sensors, magnetometer,
gyro (noise keyed on truncated coordinates), the MEKF with its 3 x 3 pseudo-inverses, the pointing law, wheel allocation /
friction / saturation, all folds between satellite, wheels and sensors — reads the translation but does not feed it.  The
tests therefore fly the example with a zero gravity field and put the satellite's position and linear velocity of the
baseline's row t in place before tick t + 1; the attitude loop runs closed over the 100 ticks on its own outputs.  Compared:
every recorded column of every entity except the three the missing field decides (the satellite's linear position,
velocity, acceleration / force)."""
import json
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"
SAT, BODY = "ore_sat", ("world_pos", "world_vel", "world_accel", "force")
ANGULAR = {"world_pos": slice(0, 4), "world_vel": slice(0, 3), "world_accel": slice(0, 3), "force": slice(0, 3)}


def gold():
    return json.loads((GOLDEN / "cube_sat_world.json").read_text())


def put_translation(g, tick, world_pos, world_vel, sat_row):
    """The state tick `tick` starts from: the satellite's orbit position / velocity of baseline row tick - 1."""
    world_pos[sat_row, 4:] = g["entities"][SAT]["world_pos"][tick - 1][4:]
    world_vel[sat_row, 3:] = g["entities"][SAT]["world_vel"][tick - 1][3:]


def errors(g, tick, row_of, body, component, worst):
    """body: name -> [rows, w] array of the five Body columns; component(name) -> [rows, w] array of a plain column or None."""
    for ent, cols in g["entities"].items():
        r = row_of[ent]
        for comp, rows in cols.items():
            ref = np.asarray(rows[tick], dtype=np.float64)
            if comp in body:
                got = np.asarray(body[comp][r], dtype=np.float64)
                if ent == SAT and comp in ANGULAR:
                    got, ref = got[ANGULAR[comp]], ref[ANGULAR[comp]]
            else:
                got = component("P" if comp == "p" else comp)          # the database spells component names in lower case
                if got is None:
                    if comp == "rw_voltage":               # a constant no system reads or writes: not a column of the program
                        continue
                    raise KeyError(comp)
                got = np.asarray(got[r], dtype=np.float64).reshape(-1)
            e = float(np.max(np.abs(got - ref))) / max(float(np.max(np.abs(ref))), 1e-9)
            key = f"{ent}.{comp}"
            worst[key] = max(worst.get(key, 0.0), e)


def verdict(worst, bound=1e-9):
    assert len(worst) == 56, (len(worst), sorted(worst))        # 59 recorded columns - 3 x rw_voltage
    bad = {k: v for k, v in worst.items() if not v < bound}
    assert not bad, bad
