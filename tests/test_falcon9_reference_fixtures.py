"""The Falcon 9 model's physics helpers against known answers produced BY THE REFERENCE'S OWN CODE.

tests/golden/falcon9_helpers.json was written by tests/golden/make_falcon9_fixtures.py, which imports
/root/reference/examples/falcon9/{atmosphere,frames,propulsion,aero,rcs}.py unmodified under a numpy shim for jax and
calls each function on seeded inputs.  Here every helper of elodin_amd/models/falcon9.py must reproduce those outputs to
1e-12 — evaluated twice, as numpy code and as the traced DAG that becomes kernel code (tests/dsl_numpy.py)."""
import json
from pathlib import Path

import numpy as np
import pytest

from elodin_amd.models import falcon9 as f9
from tests import dsl_numpy

FIX = json.loads((Path(__file__).parent / "golden" / "falcon9_helpers.json").read_text())

# fixture name -> callable(xp, *args) of this repo
MINE = {
    "pressure_temperature_at_geopotential": f9.pressure_temperature_at_geopotential,
    "pressure": f9.pressure,
    "density": f9.density,
    "speed_of_sound": f9.speed_of_sound,
    "geodetic_to_ecef": f9.geodetic_to_ecef,
    "ecef_to_geodetic": f9.ecef_to_geodetic,
    "ned_basis": f9.ned_basis,
    "gravity_accel": f9.gravity_accel,
    "frame_accel": f9.frame_accel,
    "apparent_gravity": f9.apparent_gravity,
    "engine_thrust_per_engine": f9.engine_thrust_per_engine,
    "cluster_mdot": f9.cluster_mdot,
    "split_mdot": lambda xp, m: f9.split_mdot(m),
    "actuator_step": f9.actuator_step,
    "actuator_step_limited": lambda xp, x, c, dt, tau, rl, lo, hi: f9.actuator_step(xp, x, c, dt, tau, rate_limit=rl, lo=lo, hi=hi),
    "actuator_step_vec": lambda xp, x, c, dt, tau: f9.actuator_step(xp, x, c, dt, tau, lo=0.0, hi=1.0),
    "stack_mass_props": f9.stack_mass_props,
    "tank_pressure_step": f9.tank_pressure_step,
    "inlet_pressure": f9.inlet_pressure,
    "config_blend": f9.config_blend,
    "plume_dominance": f9.plume_dominance,
    "body_aero_wrench": lambda xp, v, m, q, cg, om, ca, cn: f9.body_aero_wrench(xp, v, m, q, cg, omega_body=om, ca_scale=ca, cn_scale=cn),
    "fin_mix": f9.fin_mix,
    "fin_wrench": f9.fin_wrench,
    "rcs_wrench": f9.rcs_wrench,
    "allocate_torque": f9.allocate_torque,
}


def _flat(x):
    if isinstance(x, (tuple, list)) and x and isinstance(x[0], (tuple, list, np.ndarray)):
        return np.concatenate([_flat(v) for v in x])
    return np.ravel(np.asarray(x, dtype=np.float64))


def _flat_out(x):
    parts = x if isinstance(x, (tuple, list)) else (x,)
    return np.concatenate([np.ravel(np.asarray(p, dtype=np.float64)) for p in parts])


def test_fixture_covers_every_physics_helper_of_the_model():
    assert set(MINE) <= set(FIX)
    assert sum(len(FIX[k]) for k in MINE) >= 300


@pytest.mark.parametrize("name", sorted(MINE))
def test_helper_reproduces_the_reference_functions_output(name):
    fn = MINE[name]
    worst = 0.0
    for case in FIX[name]:
        args = [np.asarray(a, dtype=np.float64) if isinstance(a, list) else float(a) for a in case["args"]]
        want = _flat_out(case["out"])
        got_np = _flat_out(fn(np, *args))
        got_dag = _flat_out(dsl_numpy.trace_eval(fn, *args))
        scale = np.maximum(np.abs(want), 1e-9 * max(1.0, float(np.max(np.abs(want)))))
        for got in (got_np, got_dag):
            assert got.shape == want.shape, (name, got.shape, want.shape)
            worst = max(worst, float(np.max(np.abs(got - want) / scale)))
    print(f"{name}: {len(FIX[name])} cases, worst rel err {worst:.2e}")
    assert worst < 1e-12, (name, worst)
