"""The reference's own pytest cases for this path (libs/nox-py/python/tests/test_all.py), written against
the mirrored API and executed by the HIP backend."""
import numpy as np
import pytest

import elodin_amd as el
from tests import golden_util as gu
from tests import parity

pytestmark = pytest.mark.gpu


def test_six_dof():  # test_all.py:67-83
    w = el.World()
    w.spawn(el.Body(world_pos=el.SpatialTransform(linear=np.array([0.0, 0.0, 0.0])),
                    world_vel=el.SpatialMotion(linear=np.array([1.0, 0.0, 0.0])),
                    inertia=el.SpatialInertia(1.0)), "e1")
    exec = w.build(el.six_dof(1.0 / 60.0))
    exec.run()
    x = exec.column_array("world_pos")[-1]
    assert np.allclose(x[:4], [0.0, 0.0, 0.0, 1.0])
    assert np.allclose(x[4:], [0.01666667, 0.0, 0.0])
    assert exec.tick == 1


@pytest.mark.parametrize("omega,q", [([0, 0, 1.0], [0.0, 0.0, 0.479425538604203, 0.8775825618903728]),
                                     ([0, 1.0, 0], [0.0, 0.479425538604203, 0.0, 0.8775825618903728]),
                                     ([1.0, 1.0, 0], [0.45936268493243, 0.45936268493243, 0.0, 0.76024459707606])])
def test_six_dof_ang_vel_int(omega, q):  # test_all.py:228-292, "value from Julia and Simulink"
    w = el.World()
    w.spawn(el.Body(world_vel=el.SpatialMotion(angular=np.array(omega)), inertia=el.SpatialInertia(1.0)), "e1")
    exec = w.build(el.six_dof(1.0 / 120.0))
    exec.run(120)
    assert np.isclose(exec.column_array("world_pos")[-1], q + [0.0, 0.0, 0.0], rtol=1e-5).all()


def test_six_dof_force():  # test_all.py:342-364, "values taken from simulink"
    w = el.World()
    w.spawn(el.Body(inertia=el.SpatialInertia(1.0)), "e1")
    exec = w.build(el.six_dof(1.0 / 120.0, el.constant_wrench(force=(1.0, 0.0, 0.0))))
    exec.run(120)
    assert np.isclose(exec.column_array("world_pos")[-1], [0.0, 0.0, 0.0, 1.0, 0.5, 0.0, 0.0], rtol=1e-5).all()


def test_three_body_example_script():
    """examples/three-body/main.py written against the mirror; checked against the reference's golden CSV."""
    G = 6.6743e-11
    w = el.World()
    a = w.spawn(el.Body(world_pos=el.SpatialTransform(linear=[0.8920281421, 0.0, 0.0]),
                        world_vel=el.SpatialMotion(linear=[0.0, 0.9957939373, 0.0]), inertia=el.SpatialInertia(1.0 / G)), name="A")
    b = w.spawn(el.Body(world_pos=el.SpatialTransform(linear=[-0.6628498947, 0.0, 0.0]),
                        world_vel=el.SpatialMotion(linear=[0.0, -1.6191613336, 0.0]), inertia=el.SpatialInertia(1.0 / G)), name="B")
    c = w.spawn(el.Body(world_pos=el.SpatialTransform(linear=[-0.2291782474, 0, 0]),
                        world_vel=el.SpatialMotion(linear=[0, 0.6233673964, 0.0]), inertia=el.SpatialInertia(1.0 / G)), name="C")
    for x, y in ((a, b), (b, a), (a, c), (b, c), (c, a), (c, b)):
        w.spawn(el.GravityEdge(x, y))
    exec = w.build(el.six_dof(sys=el.gravity_newton(G)), simulation_rate=120.0)
    exec.run(100)
    g = gu.load("three_body")
    for i, e in enumerate("abc"):
        assert parity.pos_rel_err(exec.column_array("world_pos")[i:i + 1], g[f"{e}.world_pos"][100][None]) < parity.F64_RTOL
        assert parity.field_rel_err(exec.column_array("world_vel")[i:i + 1, 3:], g[f"{e}.world_vel"][100][None, 3:]) < parity.F64_RTOL
    assert exec.entity_ids().tolist() == [1, 2, 3] and exec.tick == 100
    assert exec.profile()["real_time_factor"] > 0


def test_ball_example_with_telemetry_batches():
    """examples/ball (gravity | drag), telemetry_rate = simulation_rate / 4 -> 4 ticks per launch."""
    g = gu.load("ball")
    w = el.World()
    w.spawn([el.Body(world_pos=el.SpatialTransform(linear=[0.0, 0.0, 6.0])), el.C("wind", g["ball.wind"][1])], name="ball")
    exec = w.build(el.six_dof(sys=el.uniform_gravity() | el.ball_drag("wind")), simulation_rate=120.0, telemetry_rate=30.0)
    exec.run(100)
    assert parity.pos_rel_err(exec.column_array("world_pos"), g["ball.world_pos"][100][None]) < parity.F64_RTOL
    assert parity.field_rel_err(exec.column_array("world_vel")[:, 3:], g["ball.world_vel"][100][None, 3:]) < parity.F64_RTOL


def test_six_dof_runs_on_the_intersection_of_entity_sets():
    """Entities with a world_pos but no Body (static scene objects / truth ghosts, apollo-lander/sim.py:312-332)
    are never touched; joins iterate the id intersection in ascending id order (query.rs:136-208)."""
    from elodin_amd import _lib as L
    from oracle import oracle as orc
    rng = np.random.default_rng(1)
    w = el.World()
    bodies, statics = [], []
    for k in range(9):
        if k % 3 == 1:   # world_pos only
            statics.append(w.spawn(el.C("world_pos", np.concatenate([[0, 0, 0, 1.0], rng.normal(size=3)]))))
        else:
            q = rng.normal(size=4); q /= np.linalg.norm(q)
            bodies.append(w.spawn([el.Body(world_pos=el.SpatialTransform(np.concatenate([q, rng.normal(size=3)])),
                                           world_vel=el.SpatialMotion(rng.normal(size=6)),
                                           inertia=el.SpatialInertia(rng.uniform(1, 5), rng.uniform(1, 3, 3))),
                                   el.C("rcs", rng.uniform(-1, 1, 3))]))
    assert [int(b) for b in bodies] == [1, 3, 4, 6, 7, 9] and [int(s) for s in statics] == [2, 5, 8]
    pos0, pos_ids = w.column("world_pos")
    vel0, vel_ids = w.column("world_vel")
    exec = w.build(el.six_dof(sys=el.uniform_gravity() | el.body_torque("rcs")), simulation_rate=120.0)
    hip = exec._hip
    # integer parity: gather indices == positions of the intersection inside each column
    joined = np.intersect1d(pos_ids, vel_ids)
    assert hip.n == 6 and joined.tolist() == [1, 3, 4, 6, 7, 9]
    assert hip.join_rows("world_pos").tolist() == [int(np.where(pos_ids == j)[0][0]) for j in joined] == [0, 2, 3, 5, 6, 8]
    assert hip.join_rows("world_vel").tolist() == [0, 1, 2, 3, 4, 5] and hip.join_rows("world_vel").dtype == np.uint32
    exec.run(25)
    rows = hip.join_rows("world_pos")
    inertia0, _ = w.column("inertia")
    rcs, _ = w.column("rcs")
    ref = orc.OracleWorld(pos0[rows], vel0, inertia0, simulation_time_step=0.008333333,
                          ops=[(orc.EFF_UNIFORM_GRAVITY, (0, 0, -9.81), None), (orc.EFF_BODY_TORQUE, (), rcs)]).step(25)
    got = exec.column_array("world_pos")
    assert parity.pos_rel_err(got[rows], ref.world_pos) < parity.F64_RTOL
    static_rows = [1, 4, 7]
    assert np.array_equal(got[static_rows], pos0[static_rows])          # untouched, bit for bit
    assert max(parity.field_rel_err(exec.column_array("world_vel")[:, :3], ref.world_vel[:, :3]),
               parity.field_rel_err(exec.column_array("world_vel")[:, 3:], ref.world_vel[:, 3:])) < parity.F64_RTOL


def test_join_in_ascending_id_order_when_components_were_inserted_late():
    """insert() on an existing entity appends its row out of id order; the join still runs in ascending id order."""
    w = el.World()
    a = w.spawn(el.C("world_pos", [0, 0, 0, 1.0, 1, 0, 0]))                     # id 1: pose first ...
    b = w.spawn(el.Body(world_pos=el.SpatialTransform(linear=[2.0, 0, 0]), world_vel=el.SpatialMotion(linear=[0, 1.0, 0])))
    for name, val in (("world_vel", [0, 0, 0, 1.0, 0, 0]), ("inertia", [1, 1, 1, 0, 0, 0, 1.0]),
                      ("force", [0] * 6), ("world_accel", [0] * 6)):
        w.insert(a, el.C(name, val))                                            # ... the rest of the Body later
    exec = w.build(el.six_dof(1.0))
    hip = exec._hip
    assert hip.join_rows("world_pos").tolist() == [0, 1] and hip.join_rows("world_vel").tolist() == [1, 0]
    exec.run(1)
    assert np.allclose(exec.column_array("world_pos")[:, 4:], [[2.0, 0, 0], [2.0, 1.0, 0]])


# ---- pipes of per-entity systems without six_dof (el.map over Body components) ----------------------------------------------

def test_spatial_integration():  # test_all.py:86-114
    from elodin_amd import dsl

    @dsl.system
    def integrate_velocity(world_pos, world_vel):
        linear = world_pos.linear() + world_vel.linear()
        angular = world_pos.angular().integrate_body(world_vel.angular())
        return {"world_pos": dsl.SpatialTransform(linear=linear, angular=angular)}

    w = el.World()
    w.spawn(el.Body(world_pos=el.SpatialTransform(linear=np.array([0.0, 0.0, 0.0])),
                    world_vel=el.SpatialMotion(linear=np.array([1.0, 0.0, 0.0]), angular=np.array([np.pi / 2, 0.0, 0.0])),
                    inertia=el.SpatialInertia(1.0)), "e1")
    exec = w.build(integrate_velocity)
    exec.run()
    exec.run()
    pos = exec.column_array("world_pos")[-1]
    assert (pos[4:] == [2.0, 0.0, 0.0]).all()
    assert np.allclose(pos[:4], np.array([0.97151626, 0.0, 0.0, 0.23697292]))
    assert exec.tick == 2
    # nothing integrates: velocity, acceleration and force columns pass through
    assert np.array_equal(exec.column_array("world_vel")[-1], [np.pi / 2, 0.0, 0.0, 1.0, 0.0, 0.0])
    assert not exec.column_array("world_accel").any() and not exec.column_array("force").any()


def test_spatial_vector_algebra():  # test_all.py:204-225
    from elodin_amd import dsl

    @dsl.system
    def double_vec(world_vel):
        return {"world_vel": world_vel + world_vel}

    w = el.World()
    w.spawn(el.Body(world_vel=el.SpatialMotion(linear=np.array([1.0, 0.0, 0.0]))), "e1")
    exec = w.build(double_vec)
    exec.run()
    assert np.array_equal(exec.column_array("world_vel")[-1], [0.0, 0.0, 0.0, 2.0, 0.0, 0.0])


def test_map_with_cond_over_a_component():  # test_all.py:731-772 (el.map + jax.lax.cond), on Body entities
    from elodin_amd import dsl

    @dsl.system
    def cond_with_map(x):
        result = dsl.lax.cond(x > 5.0, lambda _: x * 2.0, lambda _: x * 10.0, operand=None)
        taken = dsl.lax.cond(x > 5.0, lambda _: 1.0, lambda _: 0.0, operand=None)
        return {"x": result, "branch_taken": taken}

    w = el.World()
    w.spawn([el.Body(), el.C("x", [3.0]), el.C("branch_taken", [0.0])], "e1")
    w.spawn([el.Body(), el.C("x", [10.0]), el.C("branch_taken", [0.0])], "e2")
    exec = w.build(cond_with_map)
    exec.run()
    assert np.allclose(exec.column_array("x")[:, 0], [30.0, 20.0])
    assert np.allclose(exec.column_array("branch_taken")[:, 0], [0.0, 1.0])


def test_basic_system():  # test_all.py:18-64: three piped systems over plain components; each runs on its own query join
    from elodin_amd import dsl

    @dsl.system
    def foo(x):
        return {"x": x * 2}

    @dsl.system
    def bar(x, y):
        return {"x": x * y}

    @dsl.system
    def baz(x, effect):
        return {"x": x + effect}

    w = el.World()
    w.spawn([el.C("x", [1.0]), el.C("y", [500.0])], "e1")
    w.spawn([el.C("x", [15.0]), el.C("y", [500.0]), el.C("effect", [15.0])], "e2")
    exec = w.build(foo | bar | baz)
    hist = {"x": [exec.column_array("x")[:, 0].copy()], "y": [exec.column_array("y")[:, 0].copy()]}
    for _ in range(2):
        exec.run()
        hist["x"].append(exec.column_array("x")[:, 0].copy())
        hist["y"].append(exec.column_array("y")[:, 0].copy())
    assert np.array_equal(np.array(hist["x"])[:, 0], [1.0, 1000.0, 1000000.0])           # e1.x
    assert np.array_equal(np.array(hist["x"])[:, 1], [15.0, 15015.0, 15015015.0])        # e2.x: baz only touches e2
    assert np.array_equal(np.array(hist["y"]), [[500.0, 500.0]] * 3)
    assert np.array_equal(exec.column_array("effect"), [[15.0]])                         # lives on e2 alone


def test_map_over_multiple_entities_and_outputs():  # test_all.py:443-500 (map_seq == map results)
    from elodin_amd import dsl

    @dsl.system
    def compute(x):
        return {"x": x * 2, "y": x + 100.0}

    w = el.World()
    for k, v in enumerate((1.0, 2.0, 3.0)):
        w.spawn([el.C("x", [v]), el.C("y", [0.0])], f"e{k + 1}")
    exec = w.build(compute)
    exec.run()
    exec.run()
    assert np.array_equal(exec.column_array("x")[:, 0], [4.0, 8.0, 12.0])
    assert np.array_equal(exec.column_array("y")[:, 0], [102.0, 104.0, 106.0])


def test_graph():  # test_all.py:117-142: edge_fold over a plain component as a stand-alone system
    from elodin_amd import dsl

    @dsl.graph_fold("e", left=("x",), right=("x",), out="x", init=5.0)
    def fold_test(x, a, b):
        return x + a + b

    w = el.World()
    a = w.spawn(el.C("x", [1.0]), "e1")
    b = w.spawn(el.C("x", [2.0]), "e2")
    c = w.spawn(el.C("x", [2.0]), "e3")
    w.spawn(el.Edge(a, b, component="e"))
    w.spawn(el.Edge(a, c, component="e"))
    w.spawn(el.Edge(b, c, component="e"))
    exec = w.build(fold_test)
    assert np.array_equal(exec.column_array("x")[:, 0], [1.0, 2.0, 2.0])
    exec.run()
    assert np.array_equal(exec.column_array("x")[:, 0], [11.0, 9.0, 2.0])     # e3 has no out-edges: untouched
    exec.run()                                                               # folds read the values from before the system ran
    assert np.array_equal(exec.column_array("x")[:, 0], [5.0 + 11 + 9 + 11 + 2, 5.0 + 9 + 2, 2.0]) and exec.tick == 2


def test_graph_fold_over_vector_components_against_numpy():
    """Random sparse graph, 3-wide accumulator, different left / right queries, several ticks, vs a plain Python fold."""
    from elodin_amd import dsl
    np_ = dsl.np

    @dsl.graph_fold("link", left=("p", "m"), right=("p",), out="f", init=(0.0, 0.0, 0.0))
    def spring(acc, a_p, a_m, b_p):
        r = b_p - a_p
        return acc + r * (a_m / (1.0 + np_.dot(r, r)))

    rng = np.random.default_rng(3)
    n = 500
    w = el.World()
    P, M = rng.normal(size=(n, 3)), rng.uniform(1, 2, size=n)
    ids = [w.spawn([el.C("p", P[i]), el.C("m", [M[i]]), el.C("f", [0.0, 0.0, 0.0])]) for i in range(n)]
    src, dst = rng.integers(0, n - 50, 1500), rng.integers(0, n, 1500)
    for s_, d_ in zip(src, dst):
        w.spawn(el.Edge(ids[s_], ids[d_], component="link"))
    exec = w.build(spring)
    exec.run(3)                                   # p and m never change: every tick recomputes the same fold
    want = np.zeros((n, 3))
    for s_, d_ in zip(src, dst):
        r = P[d_] - P[s_]
        want[s_] = want[s_] + r * (M[s_] / (1.0 + r @ r))
    got = exec.column_array("f")
    assert np.allclose(got, want, rtol=1e-12, atol=1e-14) and np.all(got[n - 50:] == 0.0) and exec.tick == 3


def test_stablehlo_coverage_example_against_the_reference_baseline():
    """examples/stablehlo (sim.py:330-353) as the reference builds it — eight entities with one component each, the
    systems piped in its order — on the GPU, 100 ticks against the rows of scripts/ci/baseline/stablehlo
    (tests/golden/stablehlo.json).  Every system runs on its own single-entity query join.  `math_state`'s baseline predates the current math_step (tests/test_dsl_host.py), so six
    float columns and the int64 bitwise column (exact) are compared with the baseline and math_state with the numpy evaluation of the same trace."""
    import json
    from pathlib import Path
    from elodin_amd import dsl
    from tests import dsl_numpy, stablehlo_dsl as S
    gold = json.loads((Path(__file__).parent / "golden" / "stablehlo.json").read_text())["rows"]
    w = el.World()
    for name, init in S.INITIAL.items():
        w.spawn(el.C(name, init), name)
    pipe = S.SYSTEMS[0]
    for s_ in S.SYSTEMS[1:]:
        pipe = pipe | s_
    exec = w.build(pipe, simulation_rate=120.0)
    tm = dsl.ColumnTable("c", 48, 16, {"math_state": 4})
    t_math = dsl.TracedSystem(S.math_step, tm)
    math_ref = {"math_state": np.array([S.INITIAL["math_state"]])}
    dummy = (np.array([[0, 0, 0, 1.0, 0, 0, 0]]), np.zeros((1, 6)), np.ones((1, 7)))
    worst = {}
    for tick in range(1, 101):
        exec.run(1)
        dsl_numpy._run_systems([t_math], *dummy, math_ref, tm, tick)
        for name, rows in gold.items():
            ref = math_ref["math_state"][0] if name == "math_state" else np.array(rows[tick])
            got = exec.column_array(name)[0]
            worst[name] = max(worst.get(name, 0.0), float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-12))))
    print("stablehlo example on the GPU, worst relative error per component:", worst)
    assert max(worst.values()) < 1e-12 and worst["bitwise_state"] == 0.0, worst


def test_seed():  # test_all.py:145-193: a singleton Seed query read by per-entity systems, jax.random inside a map
    from elodin_amd import dsl

    @dsl.system
    def foo(x):
        return {"x": x * 2}

    @dsl.system
    def bar(x, y):
        return {"x": x * y}

    @dsl.system(singletons=("seed",))                                # s: el.Query[el.Seed] ... s[0]
    def seed_mul(seed, x):
        return {"x": x * seed}

    @dsl.system(singletons=("seed",))
    def seed_sample(seed, x, y):
        key = dsl.random.fold_in(dsl.random.key(seed), x)
        return {"y": y * dsl.random.uniform(key, minval=1.0, maxval=2.0)}

    w = el.World()
    w.spawn(el.C("seed", [2.0]))                                   # Globals(seed=2): lives on its own entity
    w.spawn([el.C("x", [1.0]), el.C("y", [500.0])], "e1")
    w.spawn([el.C("x", [15.0]), el.C("y", [500.0])], "e2")
    exec = w.build(foo | bar | seed_mul | seed_sample)
    exec.run()
    x, y = exec.column_array("x")[:, 0], exec.column_array("y")[:, 0]
    assert np.isclose(x[0], 2000.0) and np.isclose(x[1], 30000.0)
    assert 500.0 <= y[0] <= 1000.0 and 500.0 <= y[1] <= 1000.0 and y[0] != y[1]
    assert np.array_equal(exec.column_array("seed"), [[2.0]])
