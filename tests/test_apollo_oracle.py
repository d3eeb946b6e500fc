"""Apollo-lander rollout (BASELINE config 4): CPU-side checks of the restated closed loop.
Parity of this model is UNPINNED by reference data (no golden trajectory exists; the reference's own closed
loop is paced by wall-clock UDP) — these tests pin the physics qualitatively against what the example documents."""
from pathlib import Path

import numpy as np

from elodin_amd import monte_carlo as mc
from elodin_amd.models import apollo
from oracle.apollo import ApolloOracle

PLANS = Path(__file__).resolve().parent / "golden" / "plans"


def test_reference_profile_table():
    ref = apollo.load_reference()
    assert len(ref["time_s"]) == 473 and ref["time_s"][0] == 0.0 and ref["time_s"][-1] == 472.0
    assert np.all(np.diff(ref["time_s"]) == 1.0)
    assert ref["altitude_m"][-1] == 2.4            # extended to footpad contact (reference.py:385-428)
    assert apollo.max_ticks(ref) == 59041          # sim.py:55


def test_nominal_descent_lands_softly():
    ref = apollo.load_reference()
    d = apollo.default_params(ref)
    P = np.array([[d[k] for k in apollo.PARAM_NAMES]])
    o = ApolloOracle(apollo.initial_columns(P, ref), ref, max_ticks=apollo.max_ticks(ref)).step(59041)
    r = dict(zip(apollo.RESULT_NAMES, o.result[0]))
    assert r["landed"] == 1.0 and r["soft_landing"] == 1.0
    assert r["touchdown_speed"] < 1.0              # "the real LM touched down at roughly 0.5 m/s" (main.rs:15)
    assert 100.0 < r["fuel_remaining"] < 1000.0
    assert o.world_pos[0, 6] == 2.40 and np.all(o.world_vel[0] == 0.0)   # ground_contact pins the vehicle
    assert o.tick == 59041


def test_reference_campaign_plan_mostly_lands_softly():
    """The example's own 30-sample LHS plan (spec.toml, seed 19690720) through the restated loop."""
    ref = apollo.load_reference()
    plan = mc.materialize(mc.load_spec(PLANS / "apollo.toml"))
    assert plan.param_names == apollo.PARAM_NAMES
    P = plan.table()
    o = ApolloOracle(apollo.initial_columns(P, ref), ref, max_ticks=apollo.max_ticks(ref)).step(59041, threads=4)
    res = o.result
    assert np.all(res[:, 8] == 1.0)                 # every rollout reaches the surface
    assert np.all(res[:, 0] < 1.0)                  # at ~0.5 m/s vertical (terminal contact rate, main.rs:15)
    assert res[:, 9].mean() >= 0.6                  # most meet the 1 m/s horizontal soft-landing criterion too
    assert np.all(res[:, 2] > 0.0)                  # with fuel left
