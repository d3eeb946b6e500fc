"""The reference-shaped decorator surface (elodin_amd/frontend.py) on the host: what each decorated function lowers to,
the archetype / component helpers, and — through the numpy interpreter of traced programs (tests/dsl_numpy.py) — that
the lowered systems compute what the reference's tests expect.  The GPU run of the same scripts is tests/test_gpu_frontend.py."""
import typing as ty
from dataclasses import dataclass

import numpy as np
import pytest

import elodin_amd.frontend as el
from elodin_amd import api, dsl
from tests import dsl_numpy

X = ty.Annotated[el.Array, el.Component("x", el.ComponentType.F64)]
Y = ty.Annotated[el.Array, el.Component("y", el.ComponentType.F64)]
Effect = ty.Annotated[el.Array, el.Component("e", el.ComponentType.F64)]
E = ty.Annotated[el.Edge, el.Component("test_edge")]


def test_archetype_name():  # test_all.py:195-202
    @dataclass
    class TestArchetype(el.Archetype):
        x: X

    assert TestArchetype.archetype_name() == "test_archetype"
    assert el.Body.archetype_name() == "body"


def test_component_metadata():
    assert el.Component.name(X) == "x" and el.Component.id(el.WorldPos) == "world_pos"
    assert el.Component.of(el.Seed).ty.ty is el.PrimitiveType.U64
    assert el.Component.of(el.WorldVel).metadata["element_names"].split(",")[3:] == ["x", "y", "z"]
    V3 = ty.Annotated[el.Array, el.Component("wind", el.ComponentType(el.PrimitiveType.F64, (3,)))]
    assert el.Component.of(V3).ty.width == 3 and el.ComponentType.Edge.shape == (2,)
    with pytest.raises(TypeError):
        el.Component.name(float)


def test_archetypes_flatten_to_columns_and_edges():
    @dataclass
    class Test(el.Archetype):
        x: X
        y: Y

    @dataclass
    class EdgeArchetype(el.Archetype):
        edge: E

    assert {k: v.tolist() for k, v in Test(np.array([1.0]), np.array(500.0)).components().items()} == {"x": [1.0], "y": [500.0]}
    body = el.Body(world_pos=el.SpatialTransform(linear=np.array([1.0, 2.0, 3.0])), inertia=el.SpatialInertia(2.0))
    assert body.components()["world_pos"].tolist() == [0, 0, 0, 1, 1, 2, 3] and body.components()["inertia"].tolist() == [2, 2, 2, 0, 0, 0, 2]
    assert [c.name_ for c in body.component_data()] == ["world_pos", "world_vel", "inertia", "force", "world_accel"]
    e = EdgeArchetype(el.Edge(3, 4))
    assert e.components() == {} and e.edges() == [api.GravityEdge(3, 4, "test_edge")]
    c = el.C((X, Y), (np.array(1.0), np.array(2.0)))
    assert {k: v.tolist() for k, v in c.components().items()} == {"x": [1.0], "y": [2.0]}
    w = el.World()
    a = w.spawn(Test(np.array(1.0), np.array(2.0)), "e1")
    b = w.spawn([Test(np.array(3.0), np.array(4.0)), el.C(Effect, np.array(5.0))], "e2")
    w.spawn(EdgeArchetype(el.Edge(a, b)))
    assert w.column("x")[0].tolist() == [[1.0], [3.0]] and w.column("e")[1].tolist() == [int(b)]
    assert w._edges == {"test_edge": [(int(a), int(b))]} and w.entity_len == 4


def test_spatial_values_are_host_or_symbolic():
    host = el.SpatialForce(linear=np.array([1.0, 0.0, 0.0]))
    assert isinstance(host, api.SpatialForce) and isinstance(host, el.SpatialForce)
    sym = el.SpatialForce(linear=dsl.Vec([dsl.leaf("a"), 0.0, 0.0]))
    assert isinstance(sym, dsl.SpatialForce) and isinstance(sym, el.SpatialForce) and not isinstance(sym, api.SpatialForce)
    assert isinstance(el.Force(linear=np.zeros(3)), api.SpatialForce)            # the Annotated alias constructs too
    q = el.Quaternion.from_axis_angle(np.array([0.0, 0.0, 1.0]), dsl.leaf("angle"))
    assert isinstance(q, dsl.Quaternion)
    assert np.allclose(el.Quaternion.from_axis_angle([0, 0, 2.0], np.pi).vector(), [0, 0, 1, 0])
    tr = el.SpatialTransform(angular=el.Quaternion.identity(), linear=dsl.Vec([dsl.leaf("a"), 1.0, 2.0]))
    assert isinstance(tr, dsl.SpatialTransform) and tr.angular().vector().e[3].is_const(1.0)
    inertia = el.SpatialInertia(dsl.leaf("m"))
    assert isinstance(inertia, dsl.SpatialInertia) and inertia.inertia_diag().e[0] is inertia.mass()


def test_what_decorated_functions_lower_to():
    @el.system
    def foo(x: el.Query[X]) -> el.Query[X]:
        return x.map(X, lambda x: x * 2)

    @el.map
    def baz(x: X, z: Effect) -> X:
        return x + z

    @el.map_seq
    def both(x: X, y: Y) -> tuple[X, Y]:
        return x + y, x * y

    @el.map
    def gravity(f: el.Force, inertia: el.Inertia) -> el.Force:
        return f + el.SpatialForce(linear=inertia.mass() * el.np.array([0.0, 0.0, -9.81]))

    @el.map
    def constant_force(_: el.Force) -> el.Force:         # test_all.py:353-356: a host-built constant
        return el.SpatialForce(linear=np.array([1.0, 0.0, 0.0]))

    @el.system
    def fold_test(graph: el.GraphQuery[E], x: el.Query[X]) -> el.Query[X]:
        return graph.edge_fold(x, x, X, np.array(5.0), lambda x, a, b: x + a + b)

    @el.system
    def pull(graph: el.GraphQuery[E], q: el.Query[el.WorldPos, el.Inertia]) -> el.Query[el.Force]:
        def fn(acc, a_pos, a_inertia, b_pos, b_inertia):
            r = b_pos.linear() - a_pos.linear()
            return acc + el.SpatialForce(linear=r * (a_inertia.mass() * b_inertia.mass()))
        return graph.edge_fold(q, q, el.Force, el.SpatialForce(), fn)

    @el.system
    def seed_mul(s: el.Query[el.Seed], q: el.Query[X]) -> el.Query[X]:
        return q.map(X, lambda x: x * s[0])

    assert isinstance(foo, dsl.System) and foo.params == ["x"] and foo.__name__ == "foo"
    assert isinstance(baz, dsl.System) and baz.params == ["x", "e"] and baz.widths == {"x": 1, "e": 1}
    assert isinstance(both, dsl.System)
    assert isinstance(gravity, dsl.Effector) and gravity.params == ["force", "inertia"]
    assert isinstance(constant_force, dsl.Effector)
    assert isinstance(fold_test, dsl.GraphFold) and (fold_test.left, fold_test.right, fold_test.out, fold_test.init) == (("x",), ("x",), "x", (5.0,))
    assert isinstance(pull, dsl.EdgeFold) and pull.edge_component == "test_edge" and len(pull.trace().outputs) == 6
    assert seed_mul.singletons == ("seed",) and foo.singletons == ()
    # piping: systems around six_dof, effectors inside it
    stages = foo.pipe(baz) | el.six_dof(1 / 120.0, gravity | constant_force) | both
    assert [type(i).__name__ for i in stages.items] == ["System", "System", "System", "System"]
    pipe = stages.items[2].effectors.trace()
    assert [e.value for e in pipe.linear.e[:2]] == [1.0, 0.0] and pipe.columns == []
    # a map writing a plain component piped among the force effectors (examples/drone/sim.py:193): a stage list that six_dof
    # splits — with the semi-implicit integrator (one evaluation of the pipe per step) the map runs in front of the forces
    mixed = gravity | foo | constant_force
    assert isinstance(mixed, dsl.Stages) and [type(i).__name__ for i in mixed.items] == ["Effector", "System", "Effector"]
    six = el.six_dof(1 / 120.0, mixed, integrator=el.Integrator.SemiImplicit)
    assert six.stage_systems == [foo] and len(six.effectors.effectors) == 2
    with pytest.raises(NotImplementedError, match="semi-implicit"):
        el.six_dof(1 / 120.0, mixed)
    with pytest.raises(TypeError):
        el.World().build(gravity)
    with pytest.raises(TypeError):
        @el.system
        def untyped(q):
            return q
    from typing import Annotated
    assert el.GraphQuery[el.TotalEdge].edge_component == "*total_edge"                       # graph.rs:144-158
    assert el.GraphQuery[Annotated[E, el.RevEdge]].edge_component == "test_edge~rev"        # elodin/__init__.py:432-439
    w = el.World()
    a, b, c = (w.spawn(el.C(X, np.array([float(k)]))) for k in range(3))

    @dataclass
    class _EdgeArch(el.Archetype):
        edge: E
    w.spawn(_EdgeArch(el.Edge(a, b)))
    w.spawn(_EdgeArch(el.Edge(a, c)))
    assert [list(map(int, v)) for v in w.edge_pairs("test_edge")] == [[1, 1], [2, 3]]
    assert [list(map(int, v)) for v in w.edge_pairs("test_edge~rev")] == [[2, 3], [1, 1]]
    frm, to = w.edge_pairs("*total_edge")
    n = w.entity_len
    assert len(frm) == n * (n - 1) and not np.any(frm == to) and (int(frm[0]), int(to[0])) == (0, 1)
    with pytest.raises(Exception, match="multiple inputs"):
        @el.system
        def bad_index(q: el.Query[X, Y]) -> el.Query[X]:
            return q.map(X, lambda x, y: x * q[0])


def _run(systems, comps, ticks):
    """The lowered systems on the host: traced exactly as World.build traces them, evaluated by the numpy interpreter."""
    n = len(next(iter(comps.values())))
    comps = {k: np.array(v, dtype=np.float64).reshape(n, -1) for k, v in comps.items()}
    tp = dsl.Program(list(systems), dsl.Pipe([]), []).trace({k: v.shape[1] for k, v in comps.items()})
    pos, vel, inertia = np.tile([0.0, 0, 0, 1, 0, 0, 0], (n, 1)), np.zeros((n, 6)), np.tile([1.0, 1, 1, 0, 0, 0, 1], (n, 1))
    hist = [{k: v.copy() for k, v in comps.items()}]
    for t in range(ticks):
        dsl_numpy._run_systems(tp.pre, pos, vel, inertia, comps, tp.table, t + 1)
        hist.append({k: v.copy() for k, v in comps.items()})
    return hist, pos, vel


def test_basic_system_values():  # test_all.py:18-64, entity e2 (the one carrying every component)
    @el.system
    def foo(x: el.Query[X]) -> el.Query[X]:
        return x.map(X, lambda x: x * 2)

    @el.system
    def bar(q: el.Query[X, Y]) -> el.Query[X]:
        return q.map(X, lambda x, y: x * y)

    @el.map
    def baz(x: X, z: Effect) -> X:
        return x + z

    hist, _, _ = _run(foo.pipe(bar).pipe(baz).items, {"x": [15.0], "y": [500.0], "e": [15.0]}, 2)
    assert [h["x"][0, 0] for h in hist] == [15.0, 15015.0, 15015015.0] and [h["y"][0, 0] for h in hist] == [500.0] * 3


def test_map_and_map_seq_agree_with_cond():  # test_all.py:629-772
    Branch = ty.Annotated[el.Array, el.Component("branch_taken", el.ComponentType.F64)]

    def conditional_compute(x):
        result = el.lax.cond(x > 5.0, lambda _: x * 2.0, lambda _: x * 10.0, operand=None)
        return result, el.lax.cond(x > 5.0, lambda _: 1.0, lambda _: 0.0, operand=None)

    @el.system
    def with_map(q: el.Query[X]) -> el.Query[X, Branch]:
        return q.map((X, Branch), conditional_compute)

    @el.system
    def with_map_seq(q: el.Query[X]) -> el.Query[X, Branch]:
        return q.map_seq((X, Branch), conditional_compute)

    a, _, _ = _run([with_map], {"x": [3.0, 10.0], "branch_taken": [0.0, 0.0]}, 1)
    b, _, _ = _run([with_map_seq], {"x": [3.0, 10.0], "branch_taken": [0.0, 0.0]}, 1)
    assert a[-1]["x"][:, 0].tolist() == b[-1]["x"][:, 0].tolist() == [30.0, 20.0]
    assert a[-1]["branch_taken"][:, 0].tolist() == b[-1]["branch_taken"][:, 0].tolist() == [0.0, 1.0]


def test_spatial_map_values():  # test_all.py:86-114 and 204-225 on the interpreter
    @el.map
    def integrate_velocity(world_pos: el.WorldPos, world_vel: el.WorldVel) -> el.WorldPos:
        linear = world_pos.linear() + world_vel.linear()
        angular = world_pos.angular().integrate_body(world_vel.angular())
        return el.SpatialTransform(linear=linear, angular=angular)

    @el.map
    def double_vec(v: el.WorldVel) -> el.WorldVel:
        return v + v

    tp = dsl.Program([integrate_velocity], dsl.Pipe([]), []).trace({})
    pos, vel = np.array([[0.0, 0, 0, 1, 0, 0, 0]]), np.array([[np.pi / 2, 0, 0, 1.0, 0, 0]])
    inertia = np.array([[1.0, 1, 1, 0, 0, 0, 1]])
    for t in range(2):
        dsl_numpy._run_systems(tp.pre, pos, vel, inertia, {}, tp.table, t + 1)
    assert pos[0, 4:].tolist() == [2.0, 0.0, 0.0] and np.allclose(pos[0, :4], [0.97151626, 0.0, 0.0, 0.23697292])
    tp = dsl.Program([double_vec], dsl.Pipe([]), []).trace({})
    dsl_numpy._run_systems(tp.pre, pos, vel, inertia, {}, tp.table, 1)
    assert vel[0].tolist() == [np.pi, 0, 0, 2.0, 0, 0]


def test_world_accel_after_six_dof_is_this_ticks_and_in_front_of_it_the_previous_ticks():
    Meas = ty.Annotated[el.Array, el.Component("meas", el.ComponentType(el.PrimitiveType.F64, (3,)))]

    @el.map
    def imu(a: el.WorldAccel, p: el.WorldPos, _m: Meas) -> Meas:
        return p.angular().inverse() @ a.linear()

    assert isinstance(imu, dsl.System) and imu.params == ["world_accel", "world_pos", "meas"]
    tp = dsl.Program([], dsl.Pipe([]), [imu]).trace({"meas": 3})
    pos = np.array([[0.0, 0.0, np.sin(0.25), np.cos(0.25), 0, 0, 0]])
    vel, inertia, accel = np.zeros((1, 6)), np.array([[1.0, 1, 1, 0, 0, 0, 1]]), np.array([[0.0, 0, 0, 1.0, 2.0, 3.0]])
    comps = {"meas": np.zeros((1, 3))}
    dsl_numpy._run_systems(tp.post, pos, vel, inertia, comps, tp.table, 1, accel)
    c, s_ = np.cos(0.5), np.sin(0.5)
    assert np.allclose(comps["meas"][0], [c * 1.0 + s_ * 2.0, -s_ * 1.0 + c * 2.0, 3.0])
    # piped IN FRONT of six_dof the same system reads the column as the previous tick left it (examples/rocket/main.py:452-462
    # does exactly that); the program is flagged so the kernel loads the column at launch start
    tp_pre = dsl.Program([imu], dsl.Pipe([]), []).trace({"meas": 3})
    assert tp_pre.pre_reads_accel and not tp.pre_reads_accel
    comps = {"meas": np.zeros((1, 3))}
    dsl_numpy._run_systems(tp_pre.pre, pos, vel, inertia, comps, tp_pre.table, 1, accel)
    assert np.allclose(comps["meas"][0], [c * 1.0 + s_ * 2.0, -s_ * 1.0 + c * 2.0, 3.0])

    def peek(f: el.Force, m: Meas) -> Meas:
        return m + f.force()
    with pytest.raises(TypeError, match="effectors inside six_dof"):
        el.map(peek)


def test_two_dimensional_components_are_windows():
    """examples/rocket/main.py:91-98,464-472: a component declared with a 2-D shape reaches @el.map functions as a dsl.Window
    (push / row access / scan), not as 1,440 registers."""
    Sample = ty.Annotated[el.Array, el.Component("sample", el.ComponentType(el.PrimitiveType.F64, (3,)))]
    Buf = ty.Annotated[el.Array, el.Component("sample_buffer", el.ComponentType(el.PrimitiveType.F64, (16, 3)))]
    Filt = ty.Annotated[el.Array, el.Component("sample_filtered", el.ComponentType(el.PrimitiveType.F64, (3,)))]

    @el.map
    def push(a: Sample, buffer: Buf) -> Buf:
        return buffer.push(a)

    @el.map
    def low_pass(s: Buf) -> Filt:
        return s.scan(lambda c, row: (c * 0.5 + row, None), s[0], start=1)

    assert push.widths["sample_buffer"] == (16, 3) and low_pass.widths["sample_buffer"] == (16, 3)
    tp = dsl.Program([push, low_pass], dsl.Pipe([]), []).trace({"sample": 3, "sample_buffer": 48, "sample_filtered": 3})
    assert tp.windows == {"sample_buffer": (tp.windows["sample_buffer"][0], 16, 3)}
    n = 2
    comps = {"sample": np.array([[1.0, 2.0, 3.0], [-1.0, 0.5, 4.0]]), "sample_buffer": np.zeros((n, 48)), "sample_buffer#head": np.zeros((n, 1)),
             "sample_filtered": np.zeros((n, 3))}
    pos = np.tile([0.0, 0, 0, 1, 0, 0, 0], (n, 1))
    logical = np.zeros((n, 16, 3))
    for tick in range(1, 21):
        dsl_numpy._run_systems(tp.pre, pos, np.zeros((n, 6)), np.ones((n, 7)), comps, tp.table, tick)
        logical = np.concatenate([logical[:, 1:], comps["sample"][:, None, :]], axis=1)
        c = logical[:, 0]
        for r in range(1, 16):
            c = c * 0.5 + logical[:, r]
        assert np.array_equal(dsl_numpy.window_rows(comps, "sample_buffer", 16, 3), logical)
        assert np.allclose(comps["sample_filtered"], c, rtol=1e-15)
