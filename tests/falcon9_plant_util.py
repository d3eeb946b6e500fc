"""Shared by the CPU and GPU plant-parity tests: fixture loading and the comparison against the reference-driven
trajectories of tests/golden/falcon9_plant.json (made by tests/golden/make_falcon9_fixtures.py)."""
import json
from pathlib import Path

import numpy as np

from elodin_amd.models import falcon9 as f9
from tests import falcon9_script as fs

PLANT = json.loads((Path(__file__).parent / "golden" / "falcon9_plant.json").read_text())

# columns whose value is a difference of nearly equal numbers while the vehicle sits on its setpoint, or a force that is
# exactly zero in parts of a window: measured against the actuator's natural scale instead of their own magnitude
FLOORS = {"tvc_cmd": 1e-3, "tvc_state": 1e-3, "rcs_torque_cmd": 1.0, "aero_wrench": 1.0, "engine_wrench": 1.0, "fin_wrench": 1.0,
          "rcs_wrench": 1.0, "qbar": 1e-3, "mach": 1e-6, "fin_state": 1e-3, "fin_cmd": 1e-3, "rcs_levels": 1e-3, "wind_ecef": 1e-3,
          "world_accel": 1e-3, "force": 1.0, "liftoff_time": 1e-3, "axial_specific_force": 1e-3, "thrust_total": 1.0,
          "mdot_total": 1e-3, "engine_spool": 1e-6, "valve_state": 1e-6}
# f32 state (config 5's arithmetic): an f32 ECEF / pad-relative metre resolves ~0.25-0.5 m, so quantities that are small
# differences of large ones get floors at that resolution times their gain instead of their own (tiny) magnitude
# (geodetic altitude: ECEF -> geodetic on f32 coordinates of magnitude 6.4e6 m is good to a few metres: measured 2.7 m)
FLOORS_F32 = dict({k: v * 1e3 for k, v in FLOORS.items()}, altitude_geodetic=500.0, rcs_torque_cmd=2.0e4, ground_speed=1.0,
                  tvc_cmd=2e-2, tvc_state=2e-2, liftoff_time=1.0)
BODY = ("world_pos", "world_vel", "world_accel", "force", "inertia")


def param_row(case):
    """The campaign knobs the reference bakes into its systems (make_engine_dynamics(thrust_scale, isp_scale), ...) are
    per-rollout parameter columns here."""
    c = PLANT[case]["config"]
    row = f9.default_param_row().copy()
    for k in ("thrust_scale", "isp_scale", "ca_scale", "cn_scale"):
        row[f9.P[k]] = c[k]
    init = PLANT[case]["init"]
    row[f9.P["lox_kg"]], row[f9.P["rp1_kg"]] = init["propellant_lox"][0], init["propellant_rp1"][0]
    return row[None, :]


def initial_columns(case):
    """This repo's column dict with every value the reference spawned (fixture `init`)."""
    init = PLANT[case]["init"]
    params = param_row(case)
    cols = f9.initial_columns(params, upper_kg=init["upper_mass"][0])
    for name, v in init.items():
        if name in cols:
            cols[name] = np.asarray(v, dtype=np.float64).reshape(1, -1)
    if "wind_ned" in cols:
        cols["wind_ned"] = np.asarray(PLANT[case]["config"]["wind_ned"], dtype=np.float64).reshape(1, 3)
    return params, cols


def script(case):
    return fs.make_script(case, PLANT[case]["base_attitude"])


def compare(case, tick, get, floors=None):
    """get(name) -> [1, w] current column of this repo's run; returns {column: rel err} against the fixture checkpoint."""
    ref = next(c for c in PLANT[case]["checkpoints"] if c["tick"] == tick)["state"]
    errs = {}
    for name, want in ref.items():
        try:
            got = np.asarray(get(name), dtype=np.float64).reshape(-1)
        except KeyError:
            continue
        want = np.asarray(want, dtype=np.float64)
        if name == "world_pos":
            parts = [(got[:4], want[:4]), (got[4:], want[4:])]
        elif name in ("world_vel", "world_accel", "force", "aero_wrench", "engine_wrench", "fin_wrench", "rcs_wrench"):
            parts = [(got[:3], want[:3]), (got[3:], want[3:])]
        else:
            parts = [(got, want)]
        e = 0.0
        for g, w in parts:
            scale = max(float(np.max(np.abs(w))), (floors or FLOORS).get(name, 1e-300))
            e = max(e, float(np.max(np.abs(g - w))) / scale)
        errs[name] = e
    return errs


def load_program_fixture():
    """tests/golden/falcon9_plant_program.json with its generated sources inflated (make_falcon9_plant_program.py stores them deflated)."""
    import base64
    import zlib
    doc = json.loads((Path(__file__).parent / "golden" / "falcon9_plant_program.json").read_text())
    for k in doc.get("packed", ()):
        doc[k] = zlib.decompress(base64.b64decode(doc[k])).decode()
    return doc
