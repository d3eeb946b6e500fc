"""libs/nox-py/python/tests/test_all.py as the reference wrote it — @el.system / @el.map / el.Query / el.Archetype /
exec.history — executed by the HIP backend through elodin_amd.frontend.  Differences from the original file are the ones
the frontend documents: `el.np` / `el.lax` / `el.random` inside traced functions instead of jax, and history frames that
are dicts of numpy arrays (polars is not in this image)."""
import typing as ty
from dataclasses import dataclass

import numpy as np
import pytest

import elodin_amd.frontend as el

pytestmark = pytest.mark.gpu

X = ty.Annotated[el.Array, el.Component("x", el.ComponentType.F64)]
Y = ty.Annotated[el.Array, el.Component("y", el.ComponentType.F64)]
Effect = ty.Annotated[el.Array, el.Component("e", el.ComponentType.F64)]
E = ty.Annotated[el.Edge, el.Component("test_edge")]


@dataclass
class Test(el.Archetype):
    __test__ = False
    x: X
    y: Y


@dataclass
class OnlyX(el.Archetype):
    x: X


def frame_equal(df, expected):
    assert set(df) == set(expected) | {"time"}
    for k, v in expected.items():
        assert np.array_equal(df[k], np.asarray(v)), (k, df[k], v)


def test_basic_system():  # test_all.py:18-64
    @el.system
    def foo(x: el.Query[X]) -> el.Query[X]:
        return x.map(X, lambda x: x * 2)

    @el.system
    def bar(q: el.Query[X, Y]) -> el.Query[X]:
        return q.map(X, lambda x, y: x * y)

    @el.map
    def baz(x: X, z: Effect) -> X:
        return x + z

    @dataclass
    class EffectArchetype(el.Archetype):
        e: Effect

    sys = foo.pipe(bar).pipe(baz)
    w = el.World()
    w.spawn(Test(np.array([1.0]), np.array([500.0])), "e1")
    w.spawn([Test(np.array([15.0]), np.array([500.0])), EffectArchetype(np.array([15.0]))], "e2")
    exec = w.build(sys)
    exec.run()
    exec.run()
    df = exec.history(["e1.x", "e2.x", "e1.y", "e2.y"])
    frame_equal(df, {"e1.x": [1.0, 1000.0, 1000000.0], "e2.x": [15.0, 15015.0, 15015015.0],
                     "e1.y": [500.0, 500.0, 500.0], "e2.y": [500.0, 500.0, 500.0]})
    assert df["time"].tolist() == [0.0, exec._dt, 2 * exec._dt]


def test_six_dof():  # test_all.py:67-83
    w = el.World()
    w.spawn(el.Body(world_pos=el.SpatialTransform(linear=np.array([0.0, 0.0, 0.0])),
                    world_vel=el.SpatialMotion(linear=np.array([1.0, 0.0, 0.0])),
                    inertia=el.SpatialInertia(1.0)), "e1")
    sys = el.six_dof(1.0 / 60.0)
    exec = w.build(sys)
    exec.run()
    df = exec.history("e1.world_pos")
    x = df["e1.world_pos"][-1]
    assert np.allclose(x[:4], np.array([0.0, 0.0, 0.0, 1.0]))
    assert np.allclose(x[4:], np.array([0.01666667, 0.0, 0.0]))


def test_spatial_integration():  # test_all.py:86-114
    @el.map
    def integrate_velocity(world_pos: el.WorldPos, world_vel: el.WorldVel) -> el.WorldPos:
        linear = world_pos.linear() + world_vel.linear()
        angular = world_pos.angular().integrate_body(world_vel.angular())
        return el.SpatialTransform(linear=linear, angular=angular)

    w = el.World()
    w.spawn(el.Body(world_pos=el.SpatialTransform(linear=np.array([0.0, 0.0, 0.0])),
                    world_vel=el.SpatialMotion(linear=np.array([1.0, 0.0, 0.0]), angular=np.array([np.pi / 2, 0.0, 0.0])),
                    inertia=el.SpatialInertia(1.0)), "e1")
    exec = w.build(integrate_velocity)
    exec.run()
    exec.run()
    pos = exec.history("e1.world_pos")["e1.world_pos"][-1]
    assert (pos[4:] == [2.0, 0.0, 0.0]).all()
    assert np.allclose(pos[:4], np.array([0.97151626, 0.0, 0.0, 0.23697292]))


def test_graph():  # test_all.py:117-142
    @dataclass
    class EdgeArchetype(el.Archetype):
        edge: E

    @el.system
    def fold_test(graph: el.GraphQuery[E], x: el.Query[X]) -> el.Query[X]:
        return graph.edge_fold(x, x, X, np.array(5.0), lambda x, a, b: x + a + b)

    w = el.World()
    a = w.spawn(OnlyX(np.array([1.0])), "e1")
    b = w.spawn(OnlyX(np.array([2.0])), "e2")
    c = w.spawn(OnlyX(np.array([2.0])), "e3")
    w.spawn(EdgeArchetype(el.Edge(a, b)))
    w.spawn(EdgeArchetype(el.Edge(a, c)))
    w.spawn(EdgeArchetype(el.Edge(b, c)))
    exec = w.build(fold_test)
    exec.run()
    frame_equal(exec.history(["e1.x", "e2.x", "e3.x"]), {"e1.x": [1.0, 11.0], "e2.x": [2.0, 9.0], "e3.x": [2.0, 2.0]})


def test_seed():  # test_all.py:145-193
    @el.system
    def foo(x: el.Query[X]) -> el.Query[X]:
        return x.map(X, lambda x: x * 2)

    @el.system
    def bar(q: el.Query[X, Y]) -> el.Query[X]:
        return q.map(X, lambda x, y: x * y)

    @el.system
    def seed_mul(s: el.Query[el.Seed], q: el.Query[X]) -> el.Query[X]:
        return q.map(X, lambda x: x * s[0])

    @el.system
    def seed_sample(s: el.Query[el.Seed], q: el.Query[X, Y]) -> el.Query[Y]:
        def sample_inner(x, y):
            key = el.random.key(s[0])
            key = el.random.fold_in(key, x)
            scaler = el.random.uniform(key, minval=1.0, maxval=2.0)
            return y * scaler

        return q.map(Y, sample_inner)

    @dataclass
    class Globals(el.Archetype):
        seed: el.Seed

    sys = foo.pipe(bar).pipe(seed_mul).pipe(seed_sample)
    w = el.World()
    w.spawn(Globals(seed=np.array(2)))
    w.spawn(Test(np.array(1.0), np.array(500.0)), "e1")
    w.spawn(Test(np.array(15.0), np.array(500.0)), "e2")
    exec = w.build(sys)
    exec.run()
    df = exec.history(["e1.x", "e2.x", "e1.y", "e2.y"])
    assert np.isclose(df["e1.x"][-1], 2000.0)
    assert np.isclose(df["e2.x"][-1], 30000.0)
    for k in ("e1.y", "e2.y"):
        assert 500.0 <= df[k][-1] <= 1000.0
    # and the draw is jax.random's: uniform(fold_in(key(2), x)) for x = 2000, 30000 (threefry2x32, partitionable)
    from tests import dsl_numpy
    for k, x in (("e1.y", 2000.0), ("e2.y", 30000.0)):
        u = dsl_numpy.trace_eval(lambda np_, xv: el.random.uniform(el.random.fold_in(el.random.key(2), xv), minval=1.0, maxval=2.0), x)
        assert np.isclose(df[k][-1], 500.0 * u, rtol=1e-15)


def test_spatial_vector_algebra():  # test_all.py:204-225
    @el.map
    def double_vec(v: el.WorldVel) -> el.WorldVel:
        return v + v

    w = el.World()
    w.spawn(el.Body(world_vel=el.SpatialMotion(linear=np.array([1.0, 0.0, 0.0]))), "e1")
    exec = w.build(double_vec)
    exec.run()
    frame_equal(exec.history("e1.world_vel"), {"e1.world_vel": [[0.0, 0.0, 0.0, 1.0, 0.0, 0.0], [0.0, 0.0, 0.0, 2.0, 0.0, 0.0]]})


@pytest.mark.parametrize("omega,q", [([0, 0, 1.0], [0.0, 0.0, 0.479425538604203, 0.8775825618903728]),
                                     ([0, 1.0, 0], [0.0, 0.479425538604203, 0.0, 0.8775825618903728]),
                                     ([1.0, 1.0, 0], [0.45936268493243, 0.45936268493243, 0.0, 0.76024459707606])])
def test_six_dof_ang_vel_int(omega, q):  # test_all.py:228-292, "value from Julia and Simulink"
    w = el.World()
    w.spawn(el.Body(world_pos=el.SpatialTransform(linear=np.array([0.0, 0.0, 0.0])),
                    world_vel=el.SpatialMotion(angular=np.array(omega)), inertia=el.SpatialInertia(1.0)), "e1")
    exec = w.build(el.six_dof(1.0 / 120.0))
    exec.run(120)
    df = exec.history("e1.world_pos")
    assert len(df["time"]) == 121                      # one row per telemetry commit, the spawned state first
    assert np.isclose(df["e1.world_pos"][-1], np.array(q + [0.0, 0.0, 0.0]), rtol=1e-5).all()


def test_six_dof_force():  # test_all.py:342-364, "values taken from simulink"
    w = el.World()
    w.spawn(el.Body(world_pos=el.SpatialTransform(linear=np.array([0.0, 0.0, 0.0])),
                    world_vel=el.SpatialMotion(angular=np.array([0.0, 0.0, 0.0])), inertia=el.SpatialInertia(1.0)), "e1")

    @el.map
    def constant_force(_: el.Force) -> el.Force:
        return el.SpatialForce(linear=np.array([1.0, 0.0, 0.0]))

    exec = w.build(el.six_dof(1.0 / 120.0, constant_force))
    exec.run(120)
    df = exec.history(["e1.world_pos", "e1.world_vel", "e1.world_accel"])
    assert np.isclose(df["e1.world_pos"][-1], np.array([0.0, 0.0, 0.0, 1.0, 0.5, 0.0, 0.0]), rtol=1e-5).all()
    assert np.isclose(df["e1.world_vel"][-1], np.array([0.0, 0.0, 0.0, 1.0, 0.0, 0.0]), rtol=1e-5).all()
    assert np.isclose(df["e1.world_accel"][-1], np.array([0.0, 0.0, 0.0, 1.0, 0.0, 0.0]), rtol=1e-5).all()


def test_map_seq_single_entity():  # test_all.py:421-440
    @el.system
    def double_x_seq(q: el.Query[X]) -> el.Query[X]:
        return q.map_seq(X, lambda x: x * 2)

    w = el.World()
    w.spawn(OnlyX(np.array(5.0)), "e1")
    exec = w.build(double_x_seq)
    exec.run()
    exec.run()
    frame_equal(exec.history("e1.x"), {"e1.x": [5.0, 10.0, 20.0]})


def test_map_seq_multiple_entities_and_outputs():  # test_all.py:443-500
    @el.system
    def add_xy_seq(q: el.Query[X, Y]) -> el.Query[X]:
        return q.map_seq(X, lambda x, y: x + y)

    @el.system
    def swap_xy_seq(q: el.Query[X, Y]) -> el.Query[X, Y]:
        return q.map_seq((X, Y), lambda x, y: (y, x))

    for system, expected in ((add_xy_seq, {"e1.x": [1.0, 11.0], "e2.x": [2.0, 22.0], "e3.x": [3.0, 33.0]}),
                             (swap_xy_seq, {"e1.x": [1.0, 10.0], "e1.y": [10.0, 1.0], "e2.x": [2.0, 20.0], "e2.y": [20.0, 2.0]})):
        w = el.World()
        w.spawn(Test(np.array(1.0), np.array(10.0)), "e1")
        w.spawn(Test(np.array(2.0), np.array(20.0)), "e2")
        w.spawn(Test(np.array(3.0), np.array(30.0)), "e3")
        exec = w.build(system)
        exec.run()
        frame_equal(exec.history(list(expected)), expected)


@pytest.mark.parametrize("values", [[(2.0, 3.0)], [(1.0, 5.0), (2.0, 10.0)], [(1.0, 2.0), (3.0, 4.0), (5.0, 6.0)]])
def test_map_vs_map_seq_results_match(values):  # test_all.py:503-576, 629-678
    @el.system
    def compute_with_map(q: el.Query[X, Y]) -> el.Query[X, Y]:
        return q.map((X, Y), lambda x, y: (x * y + 1.0, x * y))

    @el.system
    def compute_with_map_seq(q: el.Query[X, Y]) -> el.Query[X, Y]:
        return q.map_seq((X, Y), lambda x, y: (x * y + 1.0, x * y))

    frames = []
    for system in (compute_with_map, compute_with_map_seq):
        w = el.World()
        names = []
        for k, (x, y) in enumerate(values):
            w.spawn(Test(np.array(x), np.array(y)), f"e{k + 1}")
            names += [f"e{k + 1}.x", f"e{k + 1}.y"]
        exec = w.build(system)
        exec.run()
        frames.append(exec.history(names))
    for k in frames[0]:
        assert np.array_equal(frames[0][k], frames[1][k])
    for k, (x, y) in enumerate(values):
        assert frames[0][f"e{k + 1}.x"].tolist() == [x, x * y + 1.0] and frames[0][f"e{k + 1}.y"].tolist() == [y, x * y]


def test_query_of_a_component_no_entity_has():  # test_all.py:579-626: the reference panics at build; this raises
    Z = ty.Annotated[el.Array, el.Component("z_unused", el.ComponentType.F64)]

    @el.system
    def compute_with_map(q: el.Query[Z]) -> el.Query[Z]:
        return q.map(Z, lambda z: z * 2.0)

    @el.system
    def compute_with_map_seq(q: el.Query[Z]) -> el.Query[Z]:
        return q.map_seq(Z, lambda z: z * 2.0)

    for system in (compute_with_map, compute_with_map_seq):
        w = el.World()
        w.spawn(OnlyX(np.array(1.0)), "e1")
        with pytest.raises(KeyError):
            w.build(system)


@pytest.mark.parametrize("seq", [True, False])
def test_cond_semantics(seq):  # test_all.py:681-772
    BranchTaken = ty.Annotated[el.Array, el.Component("branch_taken", el.ComponentType.F64)]

    @el.system
    def cond_system(q: el.Query[X]) -> el.Query[X, BranchTaken]:
        def conditional_compute(x):
            def true_branch(_):
                return x * 2.0

            def false_branch(_):
                return x * 10.0

            result = el.lax.cond(x > 5.0, true_branch, false_branch, operand=None)
            branch_taken = el.lax.cond(x > 5.0, lambda _: 1.0, lambda _: 0.0, operand=None)
            return result, branch_taken

        return (q.map_seq if seq else q.map)((X, BranchTaken), conditional_compute)

    @dataclass
    class WithBranch(el.Archetype):
        x: X
        branch_taken: BranchTaken

    w = el.World()
    w.spawn(WithBranch(np.array(3.0), np.array(0.0)), "e1")
    w.spawn(WithBranch(np.array(10.0), np.array(0.0)), "e2")
    exec = w.build(cond_system)
    exec.run()
    df = exec.history(["e1.x", "e2.x", "e1.branch_taken", "e2.branch_taken"])
    assert np.isclose(df["e1.x"][-1], 30.0) and np.isclose(df["e2.x"][-1], 20.0)
    assert np.isclose(df["e1.branch_taken"][-1], 0.0) and np.isclose(df["e2.branch_taken"][-1], 1.0)


def test_map_seq_decorators():  # test_all.py:775-860
    @el.map_seq
    def double_x(x: X) -> X:
        return x * 2

    @el.map_seq
    def conditional_double(x: X) -> X:
        return el.lax.cond(x > 5.0, lambda _: x * 2.0, lambda _: x * 10.0, operand=None)

    @el.map_seq
    def compute_xy(x: X, y: Y) -> tuple[X, Y]:
        return x + y, x * y

    w = el.World()
    w.spawn(OnlyX(np.array(5.0)), "e1")
    w.spawn(OnlyX(np.array(7.0)), "e2")
    exec = w.build(double_x)
    exec.run()
    exec.run()
    frame_equal(exec.history(["e1.x", "e2.x"]), {"e1.x": [5.0, 10.0, 20.0], "e2.x": [7.0, 14.0, 28.0]})

    w = el.World()
    for k, v in enumerate((3.0, 10.0, 1.0)):
        w.spawn(OnlyX(np.array(v)), f"e{k + 1}")
    exec = w.build(conditional_double)
    exec.run()
    df = exec.history(["e1.x", "e2.x", "e3.x"])
    assert [df[k][-1] for k in ("e1.x", "e2.x", "e3.x")] == [30.0, 20.0, 10.0]

    w = el.World()
    w.spawn(Test(np.array(2.0), np.array(3.0)), "e1")
    w.spawn(Test(np.array(4.0), np.array(5.0)), "e2")
    exec = w.build(compute_xy)
    exec.run()
    frame_equal(exec.history(["e1.x", "e1.y", "e2.x", "e2.y"]), {"e1.x": [2.0, 5.0], "e1.y": [3.0, 6.0], "e2.x": [4.0, 9.0], "e2.y": [5.0, 20.0]})


def test_three_body_gravity_system():
    """examples/three-body/main.py:40-78 verbatim in shape: a GravityEdge component, a GravityConstraint archetype, the
    gravity system as an edge_fold — against the built-in Newton pair functor on the same world."""
    import elodin_amd as builtin
    G = 6.6743e-11
    GravityEdge = el.Annotated[el.Edge, el.Component("gravity_edge", el.ComponentType.Edge)]

    @el.dataclass
    class GravityConstraint(el.Archetype):
        a: GravityEdge

        def __init__(self, a: el.EntityId, b: el.EntityId):
            self.a = GravityEdge(a, b)

    @el.system
    def gravity(graph: el.GraphQuery[GravityEdge], query: el.Query[el.WorldPos, el.Inertia]) -> el.Query[el.Force]:
        def gravity_fn(force, a_pos, a_inertia, b_pos, b_inertia):
            r = a_pos.linear() - b_pos.linear()
            m = a_inertia.mass()
            M = b_inertia.mass()
            norm = el.np.linalg.norm(r)
            f = G * M * m * r / (norm * norm * norm)
            return el.Force(linear=force.force() - f)

        return graph.edge_fold(left_query=query, right_query=query, return_type=el.Force, init_value=el.Force(), fold_fn=gravity_fn)

    def world(mod):
        w = mod.World()
        ids = []
        for name, p, v in (("A", [0.8822391241, 0, 0], [0, 1.0042424155, 0]), ("B", [-0.6432718586, 0, 0], [0, -1.6491842814, 0]),
                           ("C", [-0.2389672654, 0, 0], [0, 0.6449418659, 0])):
            ids.append(w.spawn(mod.Body(world_pos=mod.SpatialTransform(linear=np.array(p)), world_vel=mod.SpatialMotion(linear=np.array(v)),
                                        inertia=mod.SpatialInertia(1.0 / G)), name))
        return w, ids

    w, (a, b, c) = world(el)
    for s, d in ((a, b), (a, c), (b, c), (b, a), (c, a), (c, b)):
        w.spawn(GravityConstraint(s, d))
    exec = w.build(el.six_dof(sys=gravity), simulation_rate=120.0)
    exec.run(240)
    w2, (a, b, c) = world(builtin)
    for s, d in ((a, b), (a, c), (b, c), (b, a), (c, a), (c, b)):
        w2.spawn(builtin.GravityEdge(s, d))
    ref = w2.build(builtin.six_dof(sys=builtin.gravity_newton(G)), simulation_rate=120.0)
    ref.run(240)
    for col in ("world_pos", "world_vel", "force"):
        assert np.allclose(exec.column_array(col), ref.column_array(col), rtol=1e-10, atol=1e-12), col
    assert len(exec.history("A.world_pos")["time"]) == 241 and not np.allclose(exec.column_array("world_pos")[0, 4:], [0.8822391241, 0, 0])


def test_effector_maps_inside_six_dof():
    """examples/ball/sim.py:57-59,96-116 in the reference's own spelling: @el.map functions over el.Force piped into six_dof."""
    Wind = ty.Annotated[el.Array, el.Component("wind", el.ComponentType(el.PrimitiveType.F64, (3,)))]

    @el.map
    def gravity(f: el.Force, inertia: el.Inertia) -> el.Force:
        return f + el.SpatialForce(linear=inertia.mass() * el.np.array([0.0, 0.0, -9.81]))

    @el.map
    def apply_drag(w: Wind, v: el.WorldVel, f: el.Force) -> el.Force:
        fluid_vel = w - v.linear()
        speed = el.np.linalg.norm(fluid_vel)
        return f + el.SpatialForce(linear=0.5 * 1.225 * 0.5 * 0.25 * speed * fluid_vel)

    @dataclass
    class Windy(el.Archetype):
        wind: Wind

    import elodin_amd as builtin
    w = el.World()
    w.spawn([el.Body(world_pos=el.SpatialTransform(linear=np.array([0.0, 0.0, 6.0])), world_vel=el.SpatialMotion(linear=np.array([1.0, 0.0, 0.0]))),
             Windy(np.array([0.5, -1.0, 0.0]))], "ball")
    exec = w.build(el.six_dof(sys=gravity | apply_drag))
    exec.run(60)
    w2 = builtin.World()
    w2.spawn([builtin.Body(world_pos=builtin.SpatialTransform(linear=[0.0, 0.0, 6.0]), world_vel=builtin.SpatialMotion(linear=[1.0, 0.0, 0.0])),
              builtin.C("wind", [0.5, -1.0, 0.0])], "ball")
    ref = w2.build(builtin.six_dof(sys=builtin.uniform_gravity() | builtin.ball_drag("wind", cd=0.5, rho=1.225, area=0.25)))
    ref.run(60)
    assert np.allclose(exec.column_array("world_pos"), ref.column_array("world_pos"), rtol=1e-12)
    assert np.allclose(exec.history("ball.world_vel")["ball.world_vel"][-1], ref.column_array("world_vel")[0], rtol=1e-12)


def test_n_body_gravity_system():
    """examples/n-body/sim.py:344-369 in the reference's spelling (softened gravity as an edge_fold over a complete set of
    GravityEdge entities, typed fold arguments, `acc + el.SpatialForce(...)`) against the built-in softened pair functor."""
    import elodin_amd as builtin
    K_SQUARED, SOFTENING_AU2 = 2.9591220828559115e-04, 1.0e-8
    GravityEdge = el.Annotated[el.Edge, el.Component("gravity_edge", el.ComponentType.Edge)]

    @el.dataclass
    class GravityConstraint(el.Archetype):
        a: GravityEdge

        def __init__(self, a: el.EntityId, b: el.EntityId):
            self.a = GravityEdge(a, b)

    @el.system
    def gravity(graph: el.GraphQuery[GravityEdge], q: el.Query[el.WorldPos, el.Inertia]) -> el.Query[el.Force]:
        def gravity_fn(acc: el.Force, a_pos: el.WorldPos, a_inertia: el.Inertia, b_pos: el.WorldPos, b_inertia: el.Inertia) -> el.Force:
            r = b_pos.linear() - a_pos.linear()
            dist_sq = el.np.dot(r, r) + SOFTENING_AU2
            inv_dist = el.np.reciprocal(el.np.sqrt(dist_sq))
            inv_dist3 = inv_dist * inv_dist * inv_dist
            scalar = K_SQUARED * a_inertia.mass() * b_inertia.mass() * inv_dist3
            return acc + el.SpatialForce(linear=scalar * r)

        return graph.edge_fold(left_query=q, right_query=q, return_type=el.Force, init_value=el.SpatialForce(), fold_fn=gravity_fn)

    rng = np.random.default_rng(12)
    n = 24
    P, V, M = rng.normal(size=(n, 3)) * 3.0, rng.normal(size=(n, 3)) * 0.01, rng.uniform(1e-6, 1.0, size=n)

    def world(mod, edge):
        w = mod.World()
        ids = [w.spawn(mod.Body(world_pos=mod.SpatialTransform(linear=P[k]), world_vel=mod.SpatialMotion(linear=V[k]),
                                inertia=mod.SpatialInertia(M[k])), f"b{k}") for k in range(n)]
        for i in range(n):
            for j in range(n):
                if i != j:
                    w.spawn(edge(ids[i], ids[j]))
        return w

    exec = world(el, GravityConstraint).build(el.six_dof(sys=gravity, integrator=el.Integrator.Rk4), simulation_rate=60.0, history=False)
    ref = world(builtin, builtin.GravityEdge).build(builtin.six_dof(sys=builtin.gravity_softened(K_SQUARED, SOFTENING_AU2)), simulation_rate=60.0)
    exec.run(50)
    ref.run(50)
    for col in ("world_pos", "world_vel", "force"):
        err = np.abs(exec.column_array(col) - ref.column_array(col)) / np.maximum(np.abs(ref.column_array(col)), 1e-30)
        assert np.nanmax(np.where(np.abs(ref.column_array(col)) > 1e-12, err, 0.0)) < 1e-9, col
    with pytest.raises(RuntimeError):
        exec.history("b0.world_pos")                    # built with history=False


@pytest.mark.parametrize("seed", range(8))
def test_random_worlds_of_plain_components_follow_the_query_join_rules(seed):
    """Components scattered over entities at random, a random pipe of systems each with its own query: a system touches
    exactly the entities that carry every component it reads or writes (query.rs:136-208), reads the values its
    predecessors left, and leaves everything else alone — against a dictionary-based evaluation of those rules."""
    rng = np.random.default_rng(9300 + seed)
    Z = ty.Annotated[el.Array, el.Component("z", el.ComponentType.F64)]
    W = ty.Annotated[el.Array, el.Component("w", el.ComponentType(el.PrimitiveType.F64, (3,)))]

    @el.system
    def s1(q: el.Query[X]) -> el.Query[X]:
        return q.map(X, lambda x: x * 1.5 + 0.125)

    @el.system
    def s2(q: el.Query[X, Y]) -> el.Query[X]:
        return q.map(X, lambda x, y: x + y)

    @el.system
    def s3(q: el.Query[Y, Z]) -> el.Query[Y, Z]:
        return q.map((Y, Z), lambda y, z: (z, y * 0.5))

    @el.system
    def s4(q: el.Query[X, Z]) -> el.Query[Z]:
        return q.map(Z, lambda x, z: el.np.where(x > z, x, z - 1.0))

    @el.system
    def s5(q: el.Query[W, X]) -> el.Query[W]:
        return q.map(W, lambda w, x: w * x + el.np.array([1.0, 0.0, -1.0]))

    @el.map
    def s6(w: W, z: Z) -> Z:
        return z + el.np.sum(w) * 0.25

    py = {"s1": (("x",), ("x",), lambda v: {"x": v["x"] * 1.5 + 0.125}),
          "s2": (("x", "y"), ("x",), lambda v: {"x": v["x"] + v["y"]}),
          "s3": (("y", "z"), ("y", "z"), lambda v: {"y": v["z"], "z": v["y"] * 0.5}),
          "s4": (("x", "z"), ("z",), lambda v: {"z": np.where(v["x"] > v["z"], v["x"], v["z"] - 1.0)}),
          "s5": (("w", "x"), ("w",), lambda v: {"w": v["w"] * v["x"] + np.array([1.0, 0.0, -1.0])}),
          "s6": (("w", "z"), ("z",), lambda v: {"z": v["z"] + np.sum(v["w"]) * 0.25})}
    systems = {"s1": s1, "s2": s2, "s3": s3, "s4": s4, "s5": s5, "s6": s6}
    order = [str(k) for k in rng.permutation(list(systems))[:int(rng.integers(2, 7))]]
    used = sorted({c for k in order for c in py[k][0] + py[k][1]})
    n_entities = int(rng.choice([3, 10, 70, 300]))
    types = {"x": X, "y": Y, "z": Z, "w": W}
    state = {}
    w = el.World()
    names = []
    member = [[c for c in used if rng.random() < 0.7] or [used[0]] for _ in range(n_entities)]
    for k, c in enumerate(used):                           # every component the pipe names lives somewhere
        if not any(c in m for m in member):
            member[k % n_entities].append(c)
    for e in range(n_entities):
        have = member[e]
        vals = {c: (np.round(rng.normal(size=3), 3) if c == "w" else np.array(np.round(rng.normal(), 3))) for c in have}
        w.spawn(el.C(tuple(types[c] for c in have), tuple(vals[c] for c in have)), f"e{e}")
        state[f"e{e}"] = {c: np.array(v, dtype=np.float64) for c, v in vals.items()}
        names.append(f"e{e}")
    pipe = systems[order[0]]
    for k in order[1:]:
        pipe = pipe | systems[k]
    exec = w.build(pipe)
    ticks = 3
    exec.run(ticks)
    for _ in range(ticks):
        for k in order:
            reads, writes, fn = py[k]
            for ent in names:
                comps = state[ent]
                if all(c in comps for c in reads + writes):
                    comps.update({c: np.asarray(v, dtype=np.float64) for c, v in fn({c: comps[c] for c in reads}).items()})
    keys = [f"{ent}.{c}" for ent in names for c in state[ent]]
    df = exec.history(keys)
    for key in keys:
        ent, c = key.split(".")
        assert np.allclose(df[key][-1], state[ent][c], rtol=1e-13, atol=1e-13), (seed, order, key, df[key][-1], state[ent][c])
    assert len(df["time"]) == ticks + 1


def test_plain_component_entities_beside_bodies_run_too():
    """A system over plain components also matches an entity that is no Body while a six_dof stage is in the pipe: the
    reference updates it like any other row of the query.  Here the Body join is the main executor's row set and such
    entities get a lockstepped executor of their own; columns and history read back merged."""
    @el.system
    def count(q: el.Query[X]) -> el.Query[X]:
        return q.map(X, lambda x: x + 1.0)

    @el.map
    def speed(v: el.WorldVel, x: X) -> X:
        return x + el.np.linalg.norm(v.linear())

    def world(stray):
        w = el.World()
        w.spawn([el.Body(world_vel=el.SpatialMotion(linear=np.array([3.0, 4.0, 0.0]))), OnlyX(np.array(1.0))], "ball")
        if stray:
            w.spawn(OnlyX(np.array(10.0)), "globals")
            w.spawn(el.Body(), "rock")                                    # a Body without x: in neither query
        return w

    exec = world(True).build(count | speed | el.six_dof(1 / 120.0))
    exec.run(2)
    df = exec.history(["ball.x", "globals.x", "ball.world_pos", "rock.world_pos"])
    assert df["ball.x"].tolist() == [1.0, 7.0, 13.0] and df["globals.x"].tolist() == [10.0, 11.0, 12.0]
    assert exec.column_array("x")[:, 0].tolist() == [13.0, 12.0] and exec.tick == 2
    assert np.allclose(df["ball.world_pos"][-1][4:], [3.0 * 2 / 120, 4.0 * 2 / 120, 0.0]) and np.all(df["rock.world_pos"][-1][4:] == 0.0)
    exec = world(True).build(speed | el.six_dof(1 / 120.0))                # its query needs WorldVel: the stray entity is not in it
    exec.run(2)
    assert exec.history("ball.x")["ball.x"].tolist() == [1.0, 6.0, 11.0] and exec.column_array("x")[:, 0].tolist() == [11.0, 10.0]
    exec = world(False).build(count | el.six_dof(1 / 120.0))
    exec.run(3)
    assert exec.history("ball.x")["ball.x"].tolist() == [1.0, 2.0, 3.0, 4.0]


@pytest.mark.parametrize("seed", range(6))
def test_random_worlds_mixing_bodies_and_plain_entities(seed):
    """Bodies (some carrying plain components) next to plain-component entities, body-free systems and one reading the
    body state piped around six_dof: plain components follow the query-join rules on every entity, Bodies integrate."""
    rng = np.random.default_rng(9700 + seed)

    @el.system
    def s1(q: el.Query[X]) -> el.Query[X]:
        return q.map(X, lambda x: x * 1.5 + 0.125)

    @el.system
    def s2(q: el.Query[X, Y]) -> el.Query[Y]:
        return q.map(Y, lambda x, y: y - x * 0.25)

    @el.map
    def s3(v: el.WorldVel, y: Y) -> Y:
        return y + v.linear()[0]

    n_entities = int(rng.choice([4, 20, 150]))
    w = el.World()
    state, bodies = {}, {}
    for e in range(n_entities):
        name, arch = f"e{e}", []
        comps = {c: np.array(np.round(rng.normal(), 3)) for c in ("x", "y") if rng.random() < 0.6}
        if rng.random() < 0.5 or e == 0:
            vx = float(np.round(rng.normal(), 3))
            arch.append(el.Body(world_vel=el.SpatialMotion(linear=np.array([vx, 0.0, 0.0]))))
            bodies[name] = vx
        if e == 1:
            comps.setdefault("x", np.array(0.5))
            comps.setdefault("y", np.array(-0.5))
        if comps:
            arch.append(el.C(tuple({"x": X, "y": Y}[c] for c in comps), tuple(comps.values())))
        if not arch:
            arch.append(el.C(X, np.array(1.0)))
            comps = {"x": np.array(1.0)}
        w.spawn(arch, name)
        state[name] = {c: float(v) for c, v in comps.items()}
    exec = w.build(s1 | s2 | el.six_dof(1 / 60.0) | s3)
    ticks = 3
    exec.run(ticks)
    for _ in range(ticks):
        for ent, comps in state.items():
            if "x" in comps:
                comps["x"] = comps["x"] * 1.5 + 0.125
        for ent, comps in state.items():
            if "x" in comps and "y" in comps:
                comps["y"] = comps["y"] - comps["x"] * 0.25
        for ent, comps in state.items():
            if ent in bodies and "y" in comps:
                comps["y"] = comps["y"] + bodies[ent]
    keys = [f"{ent}.{c}" for ent, comps in state.items() for c in comps]
    df = exec.history(keys + [f"{ent}.world_pos" for ent in bodies])
    for key in keys:
        ent, c = key.split(".")
        assert np.isclose(df[key][-1], state[ent][c], rtol=1e-13, atol=1e-13), (seed, key, df[key][-1], state[ent][c])
    for ent, vx in bodies.items():
        assert np.isclose(df[f"{ent}.world_pos"][-1][4], vx * ticks / 60.0, rtol=1e-12, atol=1e-15), ent


def test_effector_reading_a_component_only_some_bodies_carry():
    """examples/ball's drag as the reference spells it, in a world where one ball has no `wind`: apply_drag's query is
    (Wind, WorldVel, Force), so that ball falls under gravity alone while the others feel their wind."""
    import elodin_amd as builtin
    Wind = ty.Annotated[el.Array, el.Component("wind", el.ComponentType(el.PrimitiveType.F64, (3,)))]

    @el.map
    def gravity(f: el.Force, inertia: el.Inertia) -> el.Force:
        return f + el.SpatialForce(linear=inertia.mass() * el.np.array([0.0, 0.0, -9.81]))

    @el.map
    def apply_drag(w: Wind, v: el.WorldVel, f: el.Force) -> el.Force:
        fluid_vel = w - v.linear()
        return f + el.SpatialForce(linear=0.5 * 1.225 * 0.5 * 0.25 * el.np.linalg.norm(fluid_vel) * fluid_vel)

    winds = {"a": [0.5, -1.0, 0.0], "b": None, "c": [0.0, 2.0, 0.1]}
    for sys_ in (lambda: el.six_dof(sys=gravity | apply_drag), lambda: el.six_dof(sys=apply_drag | gravity)):
        w = el.World()
        for k, (name, wind) in enumerate(winds.items()):
            arch = [el.Body(world_pos=el.SpatialTransform(linear=np.array([float(k), 0.0, 6.0])), world_vel=el.SpatialMotion(linear=np.array([1.0, 0.0, 0.0])))]
            if wind is not None:
                arch.append(el.C(Wind, np.array(wind)))
            w.spawn(arch, name)
        exec = w.build(sys_())
        exec.run(40)
        for k, (name, wind) in enumerate(winds.items()):
            w2 = builtin.World()
            arch = [builtin.Body(world_pos=builtin.SpatialTransform(linear=[float(k), 0.0, 6.0]), world_vel=builtin.SpatialMotion(linear=[1.0, 0.0, 0.0]))]
            ops = builtin.uniform_gravity()
            if wind is not None:
                arch.append(builtin.C("wind", wind))
                ops = ops | builtin.ball_drag("wind", cd=0.5, rho=1.225, area=0.25)
            w2.spawn(arch, name)
            ref = w2.build(builtin.six_dof(sys=ops))
            ref.run(40)
            assert np.allclose(exec.column_array("world_pos")[k], ref.column_array("world_pos")[0], rtol=1e-12), name
            assert np.allclose(exec.column_array("force")[k], ref.column_array("force")[0], rtol=1e-11, atol=1e-15), name
        assert exec.column_array("wind").tolist() == [winds["a"], winds["c"]]


@pytest.mark.parametrize("integrator", [el.Integrator.Rk4, el.Integrator.SemiImplicit])
def test_imu_model_reads_world_accel_after_and_in_front_of_six_dof(integrator):
    """An accelerometer / gyro model piped AFTER six_dof (the reference's sensor systems read el.WorldAccel the integrator
    just wrote): specific force in the body frame = q^-1 (a - g).  Checked against the same formula on the world_accel and
    world_pos rows the executor reports for that tick; piping it BEFORE six_dof is refused."""
    Accel = ty.Annotated[el.Array, el.Component("accel_meas", el.ComponentType(el.PrimitiveType.F64, (3,)))]
    Gyro = ty.Annotated[el.Array, el.Component("gyro_meas", el.ComponentType(el.PrimitiveType.F64, (3,)))]
    G = np.array([0.0, 0.0, -9.81])

    @el.map
    def gravity(f: el.Force, inertia: el.Inertia) -> el.Force:
        return f + el.SpatialForce(linear=inertia.mass() * el.np.array(G))

    @el.map
    def thrust(f: el.Force, p: el.WorldPos) -> el.Force:
        return f + el.SpatialForce(linear=p.angular() @ el.np.array([0.0, 0.0, 30.0]), torque=p.angular() @ el.np.array([0.02, 0.0, 0.01]))

    @el.map
    def imu(a: el.WorldAccel, p: el.WorldPos, v: el.WorldVel, _acc: Accel, _gyro: Gyro) -> tuple[Accel, Gyro]:
        q_inv = p.angular().inverse()
        return q_inv @ (a.linear() - el.np.array(G)), q_inv @ v.angular()

    def world():
        w = el.World()
        w.spawn([el.Body(world_pos=el.SpatialTransform(angular=el.Quaternion.from_axis_angle(np.array([1.0, 0.2, 0.0]), 0.3)),
                         world_vel=el.SpatialMotion(angular=np.array([0.1, -0.2, 0.05])), inertia=el.SpatialInertia(2.0, np.array([0.5, 0.6, 0.7]))),
                 el.C((Accel, Gyro), (np.zeros(3), np.zeros(3)))], "probe")
        return w

    exec = world().build(el.six_dof(sys=gravity | thrust, integrator=integrator) | imu)
    exec.run(25)
    df = exec.history(["probe.accel_meas", "probe.gyro_meas", "probe.world_accel", "probe.world_pos", "probe.world_vel"])

    def rot_inv(q, v):                                   # q^-1 v for a unit scalar-last quaternion
        u, w_ = -q[:3], q[3]
        t = 2.0 * np.cross(u, v)
        return v + w_ * t + np.cross(u, t)
    for k in range(1, 26):
        q = df["probe.world_pos"][k][:4]
        q = q / np.linalg.norm(q)
        assert np.allclose(df["probe.accel_meas"][k], rot_inv(q, df["probe.world_accel"][k][3:] - G), rtol=1e-11, atol=1e-12), k
        assert np.allclose(df["probe.gyro_meas"][k], rot_inv(q, df["probe.world_vel"][k][:3]), rtol=1e-11, atol=1e-13), k
    assert np.allclose(np.linalg.norm(df["probe.accel_meas"][-1]), 15.0, rtol=1e-9)      # thrust / mass, whatever the attitude
    # piped IN FRONT of six_dof the same system measures the state the tick starts from and the world_accel column as the
    # previous tick left it (the reference's column semantics; examples/rocket/main.py:452-462 relies on it)
    exec = world().build(imu | el.six_dof(sys=gravity | thrust, integrator=integrator))
    exec.run(25)
    df = exec.history(["probe.accel_meas", "probe.gyro_meas", "probe.world_accel", "probe.world_pos", "probe.world_vel"])
    for k in range(1, 26):
        q = df["probe.world_pos"][k - 1][:4]
        q = q / np.linalg.norm(q)
        assert np.allclose(df["probe.accel_meas"][k], rot_inv(q, df["probe.world_accel"][k - 1][3:] - G), rtol=1e-11, atol=1e-12), k
        assert np.allclose(df["probe.gyro_meas"][k], rot_inv(q, df["probe.world_vel"][k - 1][:3]), rtol=1e-11, atol=1e-13), k


def test_history_rows_follow_the_telemetry_rate():
    """simulation_rate 120 Hz, telemetry_rate 30 Hz: the executor steps four ticks per batch (world_builder.rs:211-243) and
    exec.history has one row per batch — the state after ticks 0, 4, 8, ... — with run(ticks) stopping wherever asked."""
    @el.map
    def count(x: X) -> X:
        return x + 1.0

    w = el.World()
    w.spawn([el.Body(world_vel=el.SpatialMotion(linear=np.array([1.2, 0.0, 0.0]))), OnlyX(np.array(0.0))], "e")
    exec = w.build(count | el.six_dof(), simulation_rate=120.0, telemetry_rate=30.0)
    exec.run(10)                                         # batches of 4, 4 and the remaining 2
    df = exec.history(["e.x", "e.world_pos"])
    assert exec.tick == 10 and df["e.x"].tolist() == [0.0, 4.0, 8.0, 10.0]
    assert np.allclose(df["time"], np.array([0, 4, 8, 10]) * exec._dt) and np.allclose(df["e.world_pos"][:, 4], np.array([0, 4, 8, 10]) * 1.2 * exec._dt)
    with pytest.raises(ValueError):
        w.build(count | el.six_dof(), simulation_rate=120.0, telemetry_rate=50.0)      # not a divisor of the simulation rate
    with pytest.raises(KeyError):
        exec.history("nobody.x")


def test_graph_reversed_and_total_edges():
    """el.GraphQuery[Annotated[E, el.RevEdge]] folds over E's edges reversed (cube-sat's sensor -> satellite folds,
    examples/cube-sat/main.py:137,422; elodin/__init__.py:432-439) and el.GraphQuery[el.TotalEdge] over every ordered
    pair of distinct entities (graph.rs:144-158): same worlds as test_graph, expected values by hand."""
    from typing import Annotated

    @dataclass
    class EdgeArchetype(el.Archetype):
        edge: E

    def world():
        w = el.World()
        a = w.spawn(OnlyX(np.array([1.0])), "e1")
        b = w.spawn(OnlyX(np.array([2.0])), "e2")
        c = w.spawn(OnlyX(np.array([4.0])), "e3")
        w.spawn(EdgeArchetype(el.Edge(a, b)))
        w.spawn(EdgeArchetype(el.Edge(a, c)))
        w.spawn(EdgeArchetype(el.Edge(b, c)))
        return w

    @el.system
    def fold_rev(graph: el.GraphQuery[Annotated[E, el.RevEdge]], x: el.Query[X]) -> el.Query[X]:
        return graph.edge_fold(x, x, X, np.array(5.0), lambda acc, a, b: acc + a + b)

    # reversed edges: b->a, c->a, c->b.  e2: 5 + (2+1) = 8; e3: 5 + (4+1) + (4+2) = 16; e1 has no out-edge now
    exec = world().build(fold_rev)
    exec.run()
    frame_equal(exec.history(["e1.x", "e2.x", "e3.x"]), {"e1.x": [1.0, 1.0], "e2.x": [2.0, 8.0], "e3.x": [4.0, 16.0]})

    @el.system
    def fold_total(graph: el.GraphQuery[el.TotalEdge], x: el.Query[X]) -> el.Query[X]:
        return graph.edge_fold(x, x, X, np.array(0.0), lambda acc, a, b: acc + b)

    # every entity sums the OTHER entities' x (edge entities and Globals carry no x and drop out of the join)
    exec = world().build(fold_total)
    exec.run()
    frame_equal(exec.history(["e1.x", "e2.x", "e3.x"]), {"e1.x": [1.0, 6.0], "e2.x": [2.0, 5.0], "e3.x": [4.0, 3.0]})


def test_cube_sat_sun_sensor_folds_in_the_reference_spelling():
    """examples/cube-sat/main.py:112-146,587-655: `sun_pos | sun_sensor | sun_sensor_value` — a map, a fold over CSSEdge
    (sensor -> satellite, reading the satellite's WorldPos) and a fold over the REVERSED edges (satellite <- sensors), the
    sensors being entities without a Body — piped in front of six_dof, decorators and queries as the reference spells them
    (minus the sensor noise; the reference's norm has no floor, the first readings here can be all zero)."""
    la = el.np.linalg
    SunPos = ty.Annotated[el.Array, el.Component("sun_pos", el.ComponentType(el.PrimitiveType.F64, (3,)))]
    CssReading = ty.Annotated[el.Array, el.Component("css_reading", el.ComponentType(el.PrimitiveType.F64, (3,)))]
    CssValue = ty.Annotated[el.Array, el.Component("css_value", el.ComponentType(el.PrimitiveType.F64, ()))]
    CssFov = ty.Annotated[el.Array, el.Component("css_fov", el.ComponentType(el.PrimitiveType.F64, (1,)))]
    CssNormal = ty.Annotated[el.Array, el.Component("css_normal", el.ComponentType(el.PrimitiveType.F64, (3,)))]
    CSSEdge = ty.Annotated[el.Edge, el.Component("css_edge")]

    @dataclass
    class CSSRel(el.Archetype):
        edge: CSSEdge

    @el.map
    def sun_pos(pos: el.WorldPos) -> SunPos:
        pos = pos.linear()
        return pos / la.norm(pos)

    @el.system
    def sun_sensor(sensor: el.GraphQuery[CSSEdge], css_normal: el.Query[CssNormal, CssFov],
                   sun_pos: el.Query[SunPos, el.WorldPos]) -> el.Query[CssValue]:
        def inner(acc, css_normal, fov, sun_pos, world_pos):
            sun_pos_b = world_pos.angular().inverse() @ sun_pos
            cos = el.np.dot(css_normal, sun_pos_b)
            return acc + el.lax.select(el.np.abs(el.np.arccos(cos)) < fov, cos, 0.0)
        return sensor.edge_fold(css_normal, sun_pos, CssValue, np.array(0.0), inner)

    @el.system
    def sun_sensor_value(graph: el.GraphQuery[ty.Annotated[CSSEdge, el.RevEdge]], css: el.Query[CssValue, CssNormal],
                         sat: el.Query[el.WorldPos]) -> el.Query[CssReading]:
        value = graph.edge_fold(sat, css, CssReading, np.array([0.0, 0.0, 0.0]), lambda acc, _, value, norm: acc + value * norm)
        return value.map(CssReading, lambda x: x / el.np.maximum(la.norm(x), 1e-12))

    rng = np.random.default_rng(12)
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    p0 = np.array([7.0e6, 1.0e6, -2.0e6])
    w = el.World()
    sat = w.spawn([el.Body(world_pos=el.SpatialTransform(angular=el.Quaternion(q), linear=p0),
                           world_vel=el.SpatialMotion(angular=np.array([0.01, 0.02, -0.015]), linear=np.array([0.0, 7.5e3, 0.0]))),
                   el.C((SunPos, CssReading), (np.zeros(3), np.zeros(3)))], "sat")
    normals = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], dtype=float)
    css = [w.spawn([el.C((CssValue, CssFov, CssNormal), (np.array(0.0), np.array([np.pi / 4]), normals[k]))], f"css_{k}") for k in range(6)]
    for c_ in css:
        w.spawn(CSSRel(el.Edge(c_, sat)))
    exec = w.build(sun_pos | sun_sensor | sun_sensor_value | el.six_dof(integrator=el.Integrator.SemiImplicit), simulation_rate=120.0)
    exec.run(12)
    df = exec.history(["sat.world_pos", "sat.css_reading", "sat.sun_pos"] + [f"css_{k}.css_value" for k in range(6)])

    def rot_inv(qv, v):
        u, w_ = -qv[:3], qv[3]
        t = 2.0 * np.cross(u, v)
        return v + w_ * t + np.cross(u, t)
    for k in range(1, 13):
        pos_before = df["sat.world_pos"][k - 1]               # the systems run in front of six_dof: on the pose the tick starts from
        qn = pos_before[:4] / np.linalg.norm(pos_before[:4])
        sun = pos_before[4:] / np.linalg.norm(pos_before[4:])
        sun_b = rot_inv(qn, sun)
        cosines = normals @ sun_b
        values = np.where(np.abs(np.arccos(cosines)) < np.pi / 4, cosines, 0.0)
        got = np.array([df[f"css_{j}.css_value"][k] for j in range(6)])
        assert np.allclose(got, values, rtol=1e-10, atol=1e-13), k
        reading = (values[:, None] * normals).sum(axis=0)
        reading = reading / max(np.linalg.norm(reading), 1e-12)
        assert np.allclose(df["sat.css_reading"][k], reading, rtol=1e-10, atol=1e-13), k
        assert np.allclose(df["sat.sun_pos"][k], sun, rtol=1e-12)
    assert np.abs(df["sat.css_reading"][-1]).max() > 0.1


def test_window_component_through_world_build():
    """A 2-D component (rocket's sample buffer shape) spawned with el.C, pushed and filtered by @el.map systems in front of
    six_dof, through World.build: the exec hands the window back in the reference's order."""
    Sample = ty.Annotated[el.Array, el.Component("sample", el.ComponentType(el.PrimitiveType.F64, (3,)))]
    Buf = ty.Annotated[el.Array, el.Component("sample_buffer", el.ComponentType(el.PrimitiveType.F64, (16, 3)))]
    Filt = ty.Annotated[el.Array, el.Component("sample_filtered", el.ComponentType(el.PrimitiveType.F64, (3,)))]

    @el.map
    def sample(v: el.WorldVel) -> Sample:
        return v.linear()

    @el.map
    def push(a: Sample, buffer: Buf) -> Buf:
        return buffer.push(a)

    @el.map
    def low_pass(s: Buf) -> Filt:
        return s.scan(lambda c, row: (c * 0.5 + row, None), s[0], start=1)

    @el.map
    def gravity(f: el.Force, inertia: el.Inertia) -> el.Force:
        return f + el.SpatialForce(linear=inertia.mass() * el.np.array([0.0, 0.0, -9.81]))

    w = el.World()
    for k in range(3):
        w.spawn([el.Body(world_vel=el.SpatialMotion(linear=np.array([1.0 + k, 0.0, 2.0]))),
                 el.C((Sample, Buf, Filt), (np.zeros(3), np.zeros((16, 3)), np.zeros(3)))], f"b{k}")
    exec = w.build(sample | push | low_pass | el.six_dof(sys=gravity, integrator=el.Integrator.SemiImplicit), simulation_rate=120.0)
    exec.run(20)
    dt = 0.008333333
    logical = np.zeros((3, 16, 3))
    v = np.array([[1.0 + k, 0.0, 2.0] for k in range(3)])
    for _ in range(20):
        logical = np.concatenate([logical[:, 1:], v[:, None, :]], axis=1)      # the systems see the velocity the tick starts from
        v = v + dt * np.array([0.0, 0.0, -9.81])
    c = logical[:, 0]
    for r in range(1, 16):
        c = c * 0.5 + logical[:, r]
    assert np.allclose(exec.column_array("sample_buffer"), logical.reshape(3, -1), rtol=1e-12, atol=1e-15)
    assert np.allclose(exec.column_array("sample_filtered"), c, rtol=1e-12)


def test_query_rs_join_cases_by_name():
    """The join cases libs/nox-py/src/query.rs tests by name (1046-1150), as WORLD behaviour through the HIP backend:
    `cross_archetype_join_shape_intersects_entity_maps` (X on entities {1, 2, 3}, E on {2}: only the entity carrying both is touched),
    `mixed_batch1_join_*_recovers_local_singleton` (a one-entity component joined with a batched one: that one entity),
    `singleton_updates_rebatch_into_broader_world_buffer` (the update of one entity lands in the 3-row column, the other rows keep
    their bits), `filtering_singleton_from_batched_query` (the rows a join leaves out are never written), and
    `batch1_mismatched_joins_stay_structurally_valid` (components on DISJOINT entity sets: an empty join — nothing runs, nothing
    breaks, every value keeps its bits)."""
    @el.map
    def add_effect(x: X, e: Effect) -> X:
        return x + e

    @el.system
    def scale_all(q: el.Query[X]) -> el.Query[X]:
        return q.map(X, lambda x: x * 2.0)

    w = el.World()
    w.spawn(el.C((X,), (np.array(1.0),)), "e1")
    w.spawn(el.C((X, Effect), (np.array(10.0), np.array(0.5))), "e2")
    w.spawn(el.C((X,), (np.array(100.0),)), "e3")
    exec = w.build(add_effect)
    exec.run(3)
    df = exec.history(["e1.x", "e2.x", "e3.x", "e2.e"])
    assert np.array_equal(df["e1.x"], [1.0] * 4) and np.array_equal(df["e3.x"], [100.0] * 4)          # untouched, bit for bit
    assert np.array_equal(df["e2.x"], [10.0, 10.5, 11.0, 11.5]) and np.array_equal(df["e2.e"], [0.5] * 4)

    w = el.World()                                                                                       # the join behind a batched system
    w.spawn(el.C((X,), (np.array(1.0),)), "e1")
    w.spawn(el.C((X, Effect), (np.array(10.0), np.array(0.5))), "e2")
    w.spawn(el.C((X,), (np.array(100.0),)), "e3")
    exec = w.build(scale_all | add_effect)
    exec.run(2)
    df = exec.history(["e1.x", "e2.x", "e3.x"])
    assert np.array_equal(df["e1.x"], [1.0, 2.0, 4.0]) and np.array_equal(df["e3.x"], [100.0, 200.0, 400.0])
    assert np.array_equal(df["e2.x"], [10.0, 20.5, 41.5])

    w = el.World()                                                                                       # disjoint entity sets: an empty join
    w.spawn(el.C((X,), (np.array(1.0),)), "e1")
    w.spawn(el.C((Effect,), (np.array(0.5),)), "e2")
    exec = w.build(scale_all | add_effect)
    exec.run(2)
    df = exec.history(["e1.x", "e2.e"])
    assert np.array_equal(df["e1.x"], [1.0, 2.0, 4.0]) and np.array_equal(df["e2.e"], [0.5] * 3)


def test_graph_rs_single_edge_groups_by_name():
    """libs/nox-py/src/graph.rs:529-566 as world behaviour: `singleton_edge_groups_recover_local_singletons` — a graph of ONE edge
    (one source, out-degree 1: the reference's group is a local singleton) — and `multi_source_single_edge_groups_rebatch_on_append`
    — two sources with one edge each (a [2, 1, 3] gather in the reference).  fold = init + (from + to) per edge; the result replaces
    the SOURCE rows only, entities without an out-edge keep their bits."""
    @dataclass
    class EdgeArchetype(el.Archetype):
        edge: E

    @el.system
    def fold_test(graph: el.GraphQuery[E], x: el.Query[X]) -> el.Query[X]:
        return graph.edge_fold(x, x, X, np.array(5.0), lambda acc, a, b: acc + a + b)

    w = el.World()
    a = w.spawn(OnlyX(np.array([1.0])), "e1")
    b = w.spawn(OnlyX(np.array([2.0])), "e2")
    w.spawn(OnlyX(np.array([7.0])), "e3")
    w.spawn(EdgeArchetype(el.Edge(a, b)))
    exec = w.build(fold_test)
    exec.run(2)
    frame_equal(exec.history(["e1.x", "e2.x", "e3.x"]), {"e1.x": [1.0, 8.0, 15.0], "e2.x": [2.0, 2.0, 2.0], "e3.x": [7.0, 7.0, 7.0]})

    w = el.World()
    a = w.spawn(OnlyX(np.array([1.0])), "e1")
    b = w.spawn(OnlyX(np.array([2.0])), "e2")
    w.spawn(OnlyX(np.array([7.0])), "e3")
    w.spawn(EdgeArchetype(el.Edge(a, b)))
    w.spawn(EdgeArchetype(el.Edge(b, a)))
    exec = w.build(fold_test)
    exec.run(2)
    # every fold of a tick sees the values from before the tick: 5 + 1 + 2 on both sources, then 5 + 8 + 8
    frame_equal(exec.history(["e1.x", "e2.x", "e3.x"]), {"e1.x": [1.0, 8.0, 21.0], "e2.x": [2.0, 8.0, 21.0], "e3.x": [7.0, 7.0, 7.0]})


def test_external_control_waiting():  # test_all.py:381-418
    """Exec.run with an external-control component (metadata {"external_control": "true"}) nobody writes: the ticks run, the system
    sees the spawned value."""
    ExternalControl = ty.Annotated[el.Array, el.Component("external_control", el.ComponentType.F64, metadata={"external_control": "true"})]

    @el.map
    def use_external_control(x: X, ext: ExternalControl) -> X:
        return x + ext

    @dataclass
    class TestWithExternal(el.Archetype):
        __test__ = False
        x: X
        external_control: ExternalControl

    w = el.World()
    w.spawn(TestWithExternal(np.array(1.0), np.array(0.0)), "e1")
    exec = w.build(use_external_control)
    exec.run(3)
    df = exec.history("e1.x")
    assert len(df["e1.x"]) >= 3
    assert np.isclose(df["e1.x"][-1], 1.0)
