"""Shared by the CPU and GPU closed-loop parity tests: tests/golden/falcon9_closed_loop.json holds whole Falcon 9 ascents
flown by the REFERENCE's plant, sensors and post_step bridge (examples/falcon9/{sim,sensors,main}.py, imported unmodified
under tests/golden/refshim.py) with the flight software restated in C (oracle/falcon9_fsw.c <- controller/src/*.rs); see
tests/golden/make_falcon9_closed_loop.py.  Nothing of elodin_amd/ is on that side of the comparison."""
import json
from pathlib import Path

import numpy as np

from elodin_amd.models import falcon9 as f9
from tests import falcon9_plant_util as pu

PATH = Path(__file__).parent / "golden" / "falcon9_closed_loop.json"
FLIGHTS = json.loads(PATH.read_text()) if PATH.exists() else {}

# the flight software's private state, as the fixture's `fsw` record names it -> (model column, slice)
FSW_MAP = {"nav_pos": ("nav_pos", slice(0, 3)), "nav_vel": ("nav_vel", slice(0, 3)), "nav_att": ("nav_att", slice(0, 4)),
           "up_pad": ("fsw_frame", slice(0, 3)), "track_dir": ("fsw_frame", slice(3, 6)),
           "initialized": ("nav_aux", slice(0, 1)), "last_gps_count": ("nav_aux", slice(1, 2)), "radar_alt_m": ("nav_aux", slice(3, 4)),
           "phase": ("fsw_state", slice(0, 1)), "phase_t0": ("fsw_state", slice(1, 2)), "purge_until": ("fsw_state", slice(2, 3)),
           "t_liftoff": ("fsw_state", slice(4, 5))}
# small differences of large numbers / quantities that sit at zero: measured against their natural scale
# Measured on the GPU (f64, whole flights, profiles/r03_falcon9_closed_loop.txt): every quantity agrees to <= 1e-10 of its scale
# except three families that are rounding noise by construction, which get their natural scale as the floor —
#   * body rates / angular accelerations while the vehicle SITS on its attitude setpoint (4e-9 rad/s, differing by 4e-14):
#     1e-3 rad/s, a thousandth of what the attitude loop commands in the pitch-over;
#   * the torques the TVC loop produces chasing those 1e-10 rad attitude errors (~1 N m, differing by 1e-6 N m): 1e4 N m, a
#     thousandth of the gimbal's authority (7.6 MN x 20 m x 0.087 rad);
#   * altitudes: ECEF -> geodetic subtracts two 6.4e6 m numbers, one ulp of which is 9.3e-10 m — exactly the difference seen on
#     the pad; 1 km, i.e. a micrometre.
FLOORS = dict(pu.FLOORS, world_vel=(1e-3, 1e-3), world_accel=(1e-3, 1e-3), force=(1e4, 1.0), engine_wrench=(1.0, 1e4),
              altitude_geodetic=1e3, radar_range=1e3, radar_alt_m=1e3, imu_gyro=1e-3, imu_accel=1e-2, nav_vel=1e-3, gps_vel=1e-3, pressure_meas=1.0,
              t_liftoff=1e-3, phase_t0=1e-3, purge_until=1e-3, gps_timer=1e-3, radar_timer=1e-3,
              attitude_setpoint=1.0, nav_att=1.0, engine_cmd=1e-3, ctrl_enable=1.0, valve_cmd=1.0)
SKIP = {"sensor_tick", "fsw"}
# the model samples the IMU and the pressure transducers on the guidance-exchange ticks only (the only ticks their samples
# are consumed on; the noise is keyed by the tick, so those samples equal the reference's): compared there, ticks 1, 11, 21, ...
EXCHANGE_ONLY = {"imu_accel", "imu_gyro", "pressure_meas"}


DETAIL = {}      # name -> (abs error, scale, part) of the last comparison's worst part: what a failing test prints


def param_row(flight):
    row = f9.default_param_row().copy()
    for k, v in flight["context"].items():
        if k in f9.P:
            row[f9.P[k]] = v
    row[f9.P["lox_kg"]], row[f9.P["rp1_kg"]] = flight["lox_kg"], flight["rp1_kg"]
    return row[None, :]


def initial_columns(flight):
    params = param_row(flight)
    cols = f9.initial_columns(params, upper_kg=flight["upper_kg"])
    for name, v in flight["init"].items():
        if name in cols and name != "fsw":
            cols[name] = np.asarray(v, dtype=np.float64).reshape(1, -1)
    return params, cols


def compare(flight, cp, get, origin=None, floors=None):
    """{name: rel err} of this repo's run (get(name) -> [1, w]) against one fixture checkpoint, flight-software state included."""
    floors = floors or FLOORS
    org = np.zeros(3) if origin is None else np.asarray(origin, dtype=np.float64)
    ref = cp["state"]
    errs = {}

    def rel(name, got, want):
        if name == "world_pos":
            parts = [(got[:4], want[:4]), (got[4:], want[4:])]
        elif name in ("world_vel", "world_accel", "force", "aero_wrench", "engine_wrench", "fin_wrench", "rcs_wrench"):
            parts = [(got[:3], want[:3]), (got[3:], want[3:])]
        else:
            parts = [(got, want)]
        e = 0.0
        for k, (g, w) in enumerate(parts):
            fl = floors.get(name, 1e-300)
            fl = fl[k] if isinstance(fl, tuple) else fl
            scale = max(float(np.max(np.abs(w))), fl)
            d = float(np.max(np.abs(g - w)))
            if d / scale > e:
                e = d / scale
                DETAIL[name] = (d, scale, k)
        return e

    for name, want in ref.items():
        if name in SKIP or (name in EXCHANGE_ONLY and cp["tick"] % f9.GUIDANCE_PERIOD_TICKS != 1):
            continue
        try:
            got = np.asarray(get(name), dtype=np.float64).reshape(-1).copy()
        except KeyError:
            continue
        want = np.asarray(want, dtype=np.float64).reshape(-1)
        if name == "world_pos":
            got[4:] += org
        if name == "gps_pos" and np.any(want):
            got += org
        errs[name] = rel(name, got, want)
    for name, want in ref["fsw"].items():
        if name not in FSW_MAP:
            continue
        col, sl = FSW_MAP[name]
        got = np.asarray(get(col), dtype=np.float64).reshape(-1)[sl].copy()
        want = np.asarray(want, dtype=np.float64).reshape(-1)
        if name == "nav_pos" and np.any(want):
            got += org
        errs["fsw." + name] = rel(name, got, want)
    return errs
