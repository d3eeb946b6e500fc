"""Pin the CPU oracle against the reference's OWN golden regression data.

The fixtures are re-packed copies of scripts/ci/baseline/{three-body,ball}-csv (see
tests/golden/make_golden.py).  The reference CI accepts 1e-4; the oracle must reproduce them
to 1e-12 relative, which only the quirk-faithful RK4 does (a textbook RK4 misses by ~1e-3).
"""
import numpy as np
import pytest

from oracle import oracle as orc
from tests import golden_util as gu

G = 6.6743e-11


def _three_body_world(g):
    names = "abc"
    pos = np.stack([g[f"{e}.world_pos"][0] for e in names])
    vel = np.stack([g[f"{e}.world_vel"][0] for e in names])
    inertia = np.stack([g[f"{e}.inertia"][0] for e in names])
    # examples/three-body/main.py:82-89 spawn order; edge columns hold (from id, to id)
    edge_names = ["a_>_b", "b_>_a", "a_>_c", "b_>_c", "c_>_a", "c_>_b"]
    frm = np.array([g[f"{e}.gravity_edge"][0, 0] for e in edge_names], dtype=np.uint64)
    to = np.array([g[f"{e}.gravity_edge"][0, 1] for e in edge_names], dtype=np.uint64)
    body_ids = np.array([1, 2, 3], dtype=np.uint64)  # Globals = 0, then a, b, c (world.rs:193-196)
    src, dst = orc.resolve_edges(body_ids, frm, to)
    dt = float(g["globals.simulation_time_step"][0, 0])
    return orc.OracleWorld(pos, vel, inertia, simulation_time_step=dt,
                           ops=[(orc.EFF_EDGE_GRAVITY_NEWTON, (G,), None)], edges=(src, dst)), (src, dst)


def test_three_body_edges_are_bit_exact():
    g = gu.load("three_body")
    _, (src, dst) = _three_body_world(g)
    # a->b, b->a, a->c, b->c, c->a, c->b as row indices
    assert src.tolist() == [0, 1, 0, 1, 2, 2]
    assert dst.tolist() == [1, 0, 2, 2, 0, 1]
    assert src.dtype == np.uint32


def test_three_body_matches_reference_golden_100_ticks():
    g = gu.load("three_body")
    w, _ = _three_body_world(g)
    assert w._w.simulation_time_step == 0.008333333
    worst = {}
    for r in range(1, 101):
        w.step(1)
        assert w.tick == int(g["globals.tick"][r, 0])
        for i, e in enumerate("abc"):
            for comp, arr in (("world_pos", w.world_pos), ("world_vel", w.world_vel),
                              ("world_accel", w.world_accel), ("force", w.force)):
                err = gu.rel_err(arr[i], g[f"{e}.{comp}"][r])
                worst[comp] = max(worst.get(comp, 0.0), err)
    print("three-body worst rel err:", worst)
    for comp, err in worst.items():
        assert err < 1e-12, (comp, err)
    # with the reference's operation order and no FMA contraction the match is bit-for-bit
    assert all(err == 0.0 for err in worst.values()), worst


def test_textbook_rk4_would_fail_golden():
    """Guards the quirk: advancing stage positions with stage velocities misses G1 by ~1e-3."""
    g = gu.load("three_body")
    names = "abc"
    x = np.stack([g[f"{e}.world_pos"][0, 4:] for e in names])
    v = np.stack([g[f"{e}.world_vel"][0, 3:] for e in names])
    m = np.array([g[f"{e}.inertia"][0, 6] for e in names])
    dt = 0.008333333

    def acc(x):
        a = np.zeros_like(x)
        for i in range(3):
            for j in range(3):
                if i != j:
                    r = x[i] - x[j]
                    a[i] -= G * m[j] * r / np.linalg.norm(r) ** 3
        return a

    for _ in range(100):
        k1v, k1a = v, acc(x)
        k2v, k2a = v + 0.5 * dt * k1a, acc(x + 0.5 * dt * k1v)
        k3v, k3a = v + 0.5 * dt * k2a, acc(x + 0.5 * dt * k2v)
        k4v, k4a = v + dt * k3a, acc(x + dt * k3v)
        x = x + dt / 6 * (k1v + 2 * k2v + 2 * k3v + k4v)
        v = v + dt / 6 * (k1a + 2 * k2a + 2 * k3a + k4a)
    ref = np.stack([g[f"{e}.world_pos"][100, 4:] for e in names])
    assert gu.rel_err(x, ref) > 1e-5


def test_ball_matches_reference_golden_100_ticks():
    g = gu.load("ball")
    wind = g["ball.wind"][1]  # sampled by sample_wind in tick 1, constant afterwards (seed 0)
    assert np.all(g["ball.wind"][1:] == wind)
    w = orc.OracleWorld(g["ball.world_pos"][0], g["ball.world_vel"][0], g["ball.inertia"][0],
                        simulation_time_step=float(g["globals.simulation_time_step"][0, 0]),
                        ops=[(orc.EFF_UNIFORM_GRAVITY, (0.0, 0.0, -9.81), None),
                             # examples/ball/sim.py:100-104: Cd, rho, 2*3.1415*r**2
                             (orc.EFF_BALL_DRAG, (0.5, 1.225, 2 * 3.1415 * 0.2**2), wind[None, :])])
    worst = {}
    for r in range(1, 101):
        # bounce (examples/ball/sim.py:65-73) runs before six_dof; never triggers in 100 ticks
        assert not (max(w.world_pos[0, 6], w.world_vel[0, 5]) < 0.0)
        w.step(1)
        for comp, arr in (("world_pos", w.world_pos), ("world_vel", w.world_vel),
                          ("world_accel", w.world_accel), ("force", w.force)):
            worst[comp] = max(worst.get(comp, 0.0), gu.rel_err(arr[0], g[f"ball.{comp}"][r]))
    print("ball worst rel err:", worst)
    for comp, err in worst.items():
        assert err < 1e-12, (comp, err)
    assert worst["world_pos"] == 0.0 and worst["world_vel"] == 0.0  # bit-exact


@pytest.mark.parametrize("threads", [2, 3])
def test_omp_variant_is_identical(threads):
    rng = np.random.default_rng(5)
    n = 37
    pos = np.concatenate([rng.normal(size=(n, 4)), rng.uniform(-10, 10, (n, 3))], axis=1)
    pos[:, :4] /= np.linalg.norm(pos[:, :4], axis=1, keepdims=True)
    vel = rng.normal(size=(n, 6))
    m = rng.uniform(1, 10, n)
    inertia = np.concatenate([rng.uniform(0.1, 10, (n, 3)) * m[:, None], np.zeros((n, 3)), m[:, None]], axis=1)
    tb = rng.uniform(-1, 1, (n, 3))
    ops = [(orc.EFF_UNIFORM_GRAVITY, (0, 0, -9.81), None), (orc.EFF_BODY_TORQUE, (), tb)]
    a = orc.OracleWorld(pos, vel, inertia, ops=ops).step(20)
    b = orc.OracleWorld(pos, vel, inertia, ops=ops).step(20, threads=threads)
    for f in ("world_pos", "world_vel", "world_accel", "force"):
        assert np.array_equal(getattr(a, f), getattr(b, f))
    assert a.tick == b.tick == 20


def test_solar_system_against_the_ephemeris_truth():
    """examples/n-body on its own truth data (planets_truth.csv excerpt, tests/golden/solar_system.json): sun + nine
    planets, softened all-pairs gravity, RK4 at one tick per hour for 20,000 ticks like the example's accuracy report
    (README: global RMS 0.155 AU, coefficient 0.9848 on its body set).  Physical sanity of the restated fold and RK4."""
    from tests import solar_util as su
    d, pos, vel, inertia = su.load()
    n = pos.shape[0]
    w = orc.OracleWorld(pos, vel, inertia, simulation_time_step=su.DT,
                        ops=[(orc.EFF_ALLPAIRS_GRAVITY_SOFTENED, (su.K_SQUARED, su.SOFTENING_AU2), None)])
    days = [x for x in d["days"] if x * su.TICKS_PER_DAY <= 20_000]
    sim = np.zeros((n - 1, len(days), 3))
    done = 0
    for k, day in enumerate(days):
        w.step(day * su.TICKS_PER_DAY - done)
        done = day * su.TICKS_PER_DAY
        sim[:, k] = w.world_pos[1:, 4:]
    truth = np.array(d["truth_au"])[:, :len(days)]
    rms, coeff, per_body = su.accuracy(sim, truth)
    print("solar system vs truth: global RMS", rms, "AU, coefficient", coeff, dict(zip(d["bodies"], per_body.round(5))))
    # the residual is common to all bodies: the example starts the sun at rest at the origin while the ephemeris is
    # barycentric, so the whole system drifts by the sun's reflex motion
    assert coeff > 0.999 and rms < 5e-3 and per_body.max() - per_body.min() < 1e-4
