"""Shared by the CPU and GPU Apollo parity tests: tests/golden/apollo_reference_runs.json (descents flown by the
reference's own sim.py systems + main.py post_step, see tests/golden/make_apollo_fixtures.py) against this repo's columns."""
import json
from pathlib import Path

import numpy as np

from elodin_amd.models import apollo

RUNS = json.loads((Path(__file__).parent / "golden" / "apollo_reference_runs.json").read_text())
NAMES = sorted(RUNS)
# apollo_state [n,16] slot of each reference component (include/sixdof_apollo.h APOLLO_S_*)
STATE_SLOTS = {"throttle": slice(0, 1), "throttle_cmd": slice(1, 2), "attitude_setpoint": slice(2, 6), "propellant": slice(6, 7),
               "rcs_propellant": slice(7, 8), "thrust": slice(8, 9), "rcs_torque": slice(9, 12), "landed": slice(12, 13),
               "touchdown_speed": slice(13, 14), "touchdown_horizontal_speed": slice(14, 15), "pitch": slice(15, 16)}
# columns that pass through zero during a descent: measured against their natural scale
FLOORS = {"rcs_torque": 1.0, "world_vel": 1e-3, "touchdown_speed": 1e-6, "touchdown_horizontal_speed": 1e-6, "pitch": 1e-3,
          "score": 1e-6, "thrust": 1.0}


def param_table():
    return np.array([[RUNS[name]["params"][k] for k in apollo.PARAM_NAMES] for name in NAMES])


def checkpoint_ticks():
    """Union of every run's checkpoint tick counts, ascending."""
    return sorted({c["ticks_done"] for name in NAMES for c in RUNS[name]["checkpoints"]})


def compare_state(get, ticks_done):
    """get(column) -> [n_runs, w] arrays of this repo's run (world_pos, world_vel, inertia, apollo_state, apollo_guidance,
    apollo_score) after `ticks_done` ticks; -> {component: worst rel err} over the runs that hold a checkpoint there."""
    errs = {}
    for i, name in enumerate(NAMES):
        cp = next((c for c in RUNS[name]["checkpoints"] if c["ticks_done"] == ticks_done), None)
        if cp is None:
            continue
        st = cp["state"]
        mine = {"world_pos": get("world_pos")[i], "world_vel": get("world_vel")[i], "inertia": get("inertia")[i],
                "guidance": get("apollo_guidance")[i][:7], "score": get("apollo_score")[i][:3]}
        for comp, sl in STATE_SLOTS.items():
            mine[comp] = get("apollo_state")[i][sl]
        for comp, got in mine.items():
            want = np.asarray(st[comp], dtype=np.float64)
            got = np.asarray(got, dtype=np.float64)
            parts = ([(got[:4], want[:4]), (got[4:], want[4:])] if comp == "world_pos" else
                     [(got[:3], want[:3]), (got[3:], want[3:])] if comp == "world_vel" else
                     [(got[:1], want[:1]), (got[1:5], want[1:5]), (got[5:6], want[5:6]), (got[6:], want[6:])] if comp == "guidance" else
                     [(got[k:k + 1], want[k:k + 1]) for k in range(3)] if comp == "score" else [(got, want)])
            for g, w in parts:
                scale = max(float(np.max(np.abs(w))), FLOORS.get(comp, 1e-300))
                errs[comp] = max(errs.get(comp, 0.0), float(np.max(np.abs(g - w))) / scale)
    return errs


def compare_results(result_rows):
    """result_rows [n_runs, 12] (models.apollo.RESULT_NAMES) vs the reference's el.monte_carlo.result(...) records."""
    errs = {}
    for i, name in enumerate(NAMES):
        want, row = RUNS[name]["result"], dict(zip(apollo.RESULT_NAMES, result_rows[i]))
        assert bool(row["landed"]) == want["landed"] and bool(row["soft_landing"]) == want["soft_landing"], (name, row, want)
        assert int(row["tick"]) == RUNS[name]["result_end_tick"], (name, row["tick"], RUNS[name]["result_end_tick"])
        for k, v in want.items():
            if isinstance(v, bool):
                continue
            errs[k] = max(errs.get(k, 0.0), abs(row[k] - v) / max(abs(v), 1e-9))
    return errs
