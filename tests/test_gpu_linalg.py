"""Small dense linear algebra through the generated gfx950 kernel (elodin_amd/dsl_mat.py: every factorisation unrolled into
registers, one lane per entity): the reference's linalg example as it builds it — six entities, one archetype each, six
systems piped in its order (examples/linalg/sim.py:377-414) — 100 ticks against the rows of its CI baseline
scripts/ci/baseline/linalg/*.csv (tests/golden/linalg.json); and the same filters over thousands of entities against the
numpy walk of the trace."""
import numpy as np
import pytest

from elodin_amd import api as el
from elodin_amd import dsl
from tests import dsl_numpy, linalg_dsl as S
from tests.test_linalg_reference import GOLD, worst_errors

pytestmark = pytest.mark.gpu


def test_linalg_example_on_the_gpu_lands_on_the_reference_baseline_rows():
    w = el.World()
    for ent, comps in S.INITIAL.items():
        w.spawn([el.C(k, v) for k, v in comps.items()], ent)
    pipe = S.SYSTEMS[0]
    for s_ in S.SYSTEMS[1:]:
        pipe = pipe | s_
    exec = w.build(pipe, simulation_rate=120.0)
    worst = {}
    for tick in range(1, 101):
        exec.run(1)
        worst_errors(lambda name: exec.column_array(name)[0], tick, worst)
    print("linalg example on the GPU vs reference baseline (100 ticks), worst per component:", {k: f"{v:.1e}" for k, v in worst.items()})
    assert len(worst) == 11 and max(worst.values()) < 1e-9, worst
    assert np.array_equal(exec.column_array("mode_state")[0], GOLD["mode_state"][100])


@pytest.mark.parametrize("ticks_per_launch", [1, 25])
def test_kalman_filters_over_many_entities(ticks_per_launch):
    """4,097 trackers (a ragged last wave), each a 3-state and a 2-state filter with its own state and covariance, 50 ticks:
    the generated kernel against the numpy walk of the same trace, 1e-9."""
    import elodin_amd as ea
    from elodin_amd import _lib as L, workloads
    n = 4097
    rng = np.random.default_rng(9)
    spd = lambda k: np.stack([(lambda a: a @ a.T + np.eye(k))(rng.normal(size=(k, k))) for _ in range(n)])
    cols = {"kf3_state": rng.normal(size=(n, 3)), "kf3_cov": spd(3), "kf3_info": np.zeros((n, 5)),
            "sm2_state": rng.normal(size=(n, 2)), "sm2_cov": spd(2), "mrhs_state": rng.normal(size=(n, 3, 2))}
    w = workloads.independent_bodies(n)
    prog = dsl.Program([S.mat_rhs_step, S.small2_step, S.kf3_step], dsl.pipe(), [])
    hip = ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], simulation_time_step=1.0 / 120.0, integrator=L.INTEGRATOR_NONE,
                     effectors=prog, columns={k: v.copy() for k, v in cols.items()}, ticks_per_launch=ticks_per_launch)
    tp = prog.trace()
    comps = {name: cols[name].reshape(n, -1).copy() for name, _ in tp.columns}
    pos, vel, acc, inertia = w["world_pos"].copy(), w["world_vel"].copy(), np.zeros((n, 6)), w["inertia"].copy()
    for tick in range(1, 51):
        dsl_numpy.program_tick_systems_only(tp, pos, vel, acc, inertia, comps, tick)
    hip.run(50)
    for name in comps:
        got = np.asarray(hip.component(name), dtype=np.float64).reshape(n, -1)
        scale = np.maximum(np.max(np.abs(comps[name]), axis=1, keepdims=True), 1e-300)
        assert float(np.max(np.abs(got - comps[name]) / scale)) < 1e-9, name
    assert hip.component("kf3_cov").shape == (n, 3, 3)
