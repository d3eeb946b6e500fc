"""The Apollo CPU oracle against descents flown by the REFERENCE'S OWN CODE (tests/golden/apollo_reference_runs.json:
sim.py's systems + main.py's post_step executed on numpy, batched like the server loop — see
tests/golden/make_apollo_fixtures.py for exactly what is reference code and what is restated).  Pins
oracle/apollo_oracle.c — plant, post_step glue, telemetry batching, result record — over four full descents
(the nominal spec.ci.toml run and three rows of the example's own LHS plan); the guidance law is on both sides."""
import numpy as np

from elodin_amd.models import apollo
from oracle.apollo import ApolloOracle
from tests import apollo_fixture_util as fx


def test_fixture_shape():
    assert fx.NAMES == ["nominal", "run_0000000", "run_0000011", "run_0000023"]
    for name in fx.NAMES:
        r = fx.RUNS[name]
        assert r["max_ticks"] == 59041 and r["result"]["landed"] and 50_000 < r["result_end_tick"] < 59_041
        assert (r["result_end_tick"] + 1) % 3 == 0          # post_step only ever sees batch ends (end_tick = 3k - 1)
        assert set(r["params"]) == set(apollo.PARAM_NAMES)


def test_oracle_follows_the_reference_flown_descents():
    ref = apollo.load_reference()
    o = ApolloOracle(apollo.initial_columns(fx.param_table(), ref), ref, max_ticks=apollo.max_ticks(ref), ticks_per_telemetry=3)
    worst, done = {}, 0
    for t in fx.checkpoint_ticks():
        o.step(t - done)
        done = t
        get = lambda k: {"world_pos": o.world_pos, "world_vel": o.world_vel, "inertia": o.inertia, "apollo_state": o.apollo_state,
                         "apollo_guidance": o.guidance, "apollo_score": o.score}[k]
        for k, e in fx.compare_state(get, t).items():
            worst[k] = max(worst.get(k, 0.0), e)
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:5]
    print("apollo oracle vs reference-flown descents, worst per component:", ", ".join(f"{k} {e:.1e}" for k, e in top))
    res = fx.compare_results(o.result)
    print("result records:", {k: f"{e:.1e}" for k, e in res.items()})
    assert max(worst.values()) < 1e-9, top
    assert max(res.values()) < 1e-9, res


def test_post_step_every_tick_is_a_different_flight():
    """Guards the batching: with post_step after every tick (ticks_per_telemetry = 1, what round 1 flew) the guidance
    runs at 24 Hz instead of the reference's effective 8 Hz and the descent is measurably different."""
    ref = apollo.load_reference()
    P = fx.param_table()[:1]
    a = ApolloOracle(apollo.initial_columns(P, ref), ref, max_ticks=apollo.max_ticks(ref), ticks_per_telemetry=3).step(59041)
    b = ApolloOracle(apollo.initial_columns(P, ref), ref, max_ticks=apollo.max_ticks(ref), ticks_per_telemetry=1).step(59041)
    assert a.result[0, 10] != b.result[0, 10] or abs(a.result[0, 2] - b.result[0, 2]) > 1e-3
