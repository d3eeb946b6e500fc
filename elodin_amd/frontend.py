"""The reference's decorator surface for the six_dof path, lowered onto this backend.

`import elodin_amd.frontend as el` gives a sim script the names it uses with `import elodin as el`
(libs/nox-py/python/elodin/__init__.py:160-400 system / Query / map / map_seq, :408-557 GraphQuery.edge_fold, :563-669
Archetype / C / Body and the well-known component types, elodin.pyi:173-183,417-443 ComponentType / Edge / Component):

    X = ty.Annotated[el.Array, el.Component("x", el.ComponentType.F64)]

    @el.system
    def bar(q: el.Query[X, Y]) -> el.Query[X]:
        return q.map(X, lambda x, y: x * y)

    @el.map
    def gravity(f: el.Force, inertia: el.Inertia) -> el.Force:
        return f + el.SpatialForce(linear=inertia.mass() * el.np.array([0.0, 0.0, -9.81]))

    exec = w.build(foo.pipe(bar) | el.six_dof(sys=gravity))

Nothing here computes.  A decorated function is called once on symbols (elodin_amd.dsl) to learn what it is, and becomes
 * a dsl.System      — per-entity map over component columns, compiled into the step kernel around six_dof,
 * a dsl.Effector    — a map whose output is el.Force, compiled into the RK4 stage loop as a six_dof effector,
 * a dsl.EdgeFold    — edge_fold over Query[WorldPos, Inertia] returning el.Force (the pair kernels), or
 * a dsl.GraphFold   — edge_fold over plain components (a stand-alone generated kernel).
Inside the functions `el.np`, `el.lax` and `el.random` stand where the reference's scripts use jax.numpy, jax.lax and
jax.random.  `map_seq` is `map`: one lane per entity, so `lax.cond` already runs one branch per entity and the
results are identical (test_all.py:503-578 asserts exactly that of the reference)."""
from __future__ import annotations

import dataclasses
import enum
import inspect
import re
import typing
from dataclasses import dataclass  # noqa: F401  (scripts write @el.dataclass)
from typing import Annotated, Any, List, Optional, Sequence, Tuple  # noqa: F401

import numpy as _np

from . import api as _api
from . import dsl as _dsl
from . import monte_carlo  # noqa: F401  (el.monte_carlo.Param / params_spec / params / result / port)
from .api import EntityId, Integrator, skew  # noqa: F401
from ._lib import BackendError  # noqa: F401

np, lax, random = _dsl.np, _dsl.lax, _dsl.random
Array = _np.ndarray          # stands where the reference's annotations say jax.Array
table = _dsl.HostTable       # `jnp.asarray(big_host_array)` of a reference script: a constant table traced code gathers rows from


# ---- component metadata (elodin.pyi:165-183,424-443) ---------------------------------------------------------------

class PrimitiveType(enum.Enum):
    F64 = "f64"; F32 = "f32"; U64 = "u64"; U32 = "u32"; U16 = "u16"; U8 = "u8"      # noqa: E702
    I64 = "i64"; I32 = "i32"; I16 = "i16"; I8 = "i8"; Bool = "bool"                # noqa: E702


class ComponentType:
    """Element type + shape.  Columns on this backend are floating point (the executor's dtype); integer components such
    as el.Seed ride in them exactly up to 2**53."""

    def __init__(self, ty: PrimitiveType, shape: Sequence[int] = ()):
        self.ty, self.shape = ty, tuple(int(s) for s in shape)

    @property
    def width(self) -> int:
        return int(_np.prod(self.shape)) if self.shape else 1

    def __repr__(self): return f"ComponentType({self.ty.name}, {self.shape})"


ComponentType.U64 = ComponentType(PrimitiveType.U64)
ComponentType.F64 = ComponentType(PrimitiveType.F64)
ComponentType.F32 = ComponentType(PrimitiveType.F32)
ComponentType.Edge = ComponentType(PrimitiveType.U64, (2,))
ComponentType.Quaternion = ComponentType(PrimitiveType.F64, (4,))
ComponentType.SpatialPosF64 = ComponentType(PrimitiveType.F64, (7,))
ComponentType.SpatialMotionF64 = ComponentType(PrimitiveType.F64, (6,))


COMPONENT_METADATA: dict = {}      # component name -> the metadata it was declared with (the commit path reads `external_control`)


class Component:
    def __init__(self, name: str, ty: Optional[ComponentType] = None, asset: bool = False, metadata: Optional[dict] = None):
        self.name_, self.ty, self.asset, self.metadata = name, ty, asset, dict(metadata or {})
        if self.metadata:
            COMPONENT_METADATA.setdefault(name, {}).update(self.metadata)

    @staticmethod
    def of(component: Any) -> "Component":
        for m in getattr(component, "__metadata__", ()):
            if isinstance(m, Component):
                return m
        raise TypeError(f"{component!r} is not an Annotated component type")

    @staticmethod
    def name(component: Any) -> str:
        return Component.of(component).name_

    id = name   # the deprecated spelling


def _origin(component: Any):
    return getattr(component, "__origin__", None)


def _window_shape(component: Any) -> Optional[Tuple[int, int]]:
    """(rows, width) of a component declared with a 2-D shape: `el.ComponentType(F64, (480, 3))` (examples/rocket/main.py:91-98)
    is kept in HBM as a window (dsl.Window); a small one — `(3, 3)`, a filter covariance (examples/linalg/sim.py:33-36), up to
    36 values — is a register matrix (dsl_mat.Mat).  The tracer decides by size (dsl._is_matrix_shape)."""
    c = Component.of(component)
    if c.ty is not None and len(c.ty.shape) == 2:
        return int(c.ty.shape[0]), int(c.ty.shape[1])
    return None


def _width(component: Any) -> Optional[int]:
    c = Component.of(component)
    if c.ty is not None:
        return c.ty.width
    return {SpatialTransform: 7, SpatialMotion: 6, SpatialForce: 6, SpatialInertia: 7, Quaternion: 4}.get(_origin(component))


# ---- spatial values: numpy-backed when spawning, symbolic inside a traced function -------------------------------------

from . import dsl_mat as _dsl_mat  # noqa: E402

_SYMBOLIC = (_dsl.Expr, _dsl.Vec, _dsl.Quaternion, _dsl.SpatialTransform, _dsl.SpatialMotion, _dsl.SpatialForce,
             _dsl.SpatialInertia, _dsl_mat.Mat)


def _symbolic(v) -> bool:
    return isinstance(v, _SYMBOLIC)


def _lift(v):
    """A host value met inside a traced expression becomes a constant of the trace."""
    if _symbolic(v) or v is None:
        return v
    vec = lambda a: _dsl.Vec([float(x) for x in a])
    if isinstance(v, _api.Quaternion):
        return _dsl.Quaternion(vec(v.arr))
    if isinstance(v, _api.SpatialTransform):
        return _dsl.SpatialTransform(_dsl.Quaternion(vec(v.arr[:4])), vec(v.arr[4:]))
    if isinstance(v, _api.SpatialMotion):
        return _dsl.SpatialMotion(vec(v.arr[:3]), vec(v.arr[3:]))
    if isinstance(v, _api.SpatialForce):
        return _dsl.SpatialForce(vec(v.arr[:3]), vec(v.arr[3:]))
    if isinstance(v, _api.SpatialInertia):
        return _dsl.SpatialInertia(vec(v.arr[:3]), float(v.arr[6]))
    if isinstance(v, (_np.ndarray, list, tuple)):
        a = _np.asarray(v, dtype=_np.float64)
        return float(a) if a.ndim == 0 else _dsl.Vec([float(x) for x in a.reshape(-1)])
    return v


class _DualMeta(type):
    def __call__(cls, *args, **kw):
        if any(_symbolic(v) for v in list(args) + list(kw.values())):
            return cls._traced(*[_lift(a) for a in args], **{k: _lift(v) for k, v in kw.items()})
        return cls._host(*args, **kw)

    def __instancecheck__(cls, obj):
        return isinstance(obj, (cls._host, cls._traced_cls))


def _dual(name, host, traced_cls, traced=None, **statics):
    ns = {"_host": host, "_traced_cls": traced_cls, "_traced": staticmethod(traced or traced_cls), "__doc__": host.__doc__}
    ns.update({k: staticmethod(v) for k, v in statics.items()})
    return _DualMeta(name, (), ns)


def _q_from_axis_angle(axis, angle):
    if _symbolic(axis) or _symbolic(angle):
        return _dsl.Quaternion.from_axis_angle(_lift(axis), _lift(angle))
    return _api.Quaternion.from_axis_angle(axis, angle)


def _traced_inertia(mass, inertia=None):     # spatial.rs:392-405: inertia defaults to ones(3) * mass
    mass = _lift(mass)
    return _dsl.SpatialInertia(_lift(inertia) if inertia is not None else _dsl.Vec([mass, mass, mass]), mass)


def _q_from_array(arr):
    return _dsl.Quaternion(_lift(arr)) if _symbolic(arr) else _api.Quaternion(arr)


Quaternion = _dual("Quaternion", _api.Quaternion, _dsl.Quaternion, identity=_api.Quaternion.identity,
                   from_axis_angle=_q_from_axis_angle, from_array=_q_from_array)
SpatialTransform = _dual("SpatialTransform", _api.SpatialTransform, _dsl.SpatialTransform)
SpatialMotion = _dual("SpatialMotion", _api.SpatialMotion, _dsl.SpatialMotion)
SpatialForce = _dual("SpatialForce", _api.SpatialForce, _dsl.SpatialForce)
SpatialInertia = _dual("SpatialInertia", _api.SpatialInertia, _dsl.SpatialInertia, traced=_traced_inertia)


def Edge(left, right):
    """el.Edge(left, right) (graph.rs:17-41); the edge component's name comes from the archetype field it is stored in."""
    return _api.GravityEdge(int(left), int(right))


Edge.__metadata__ = ()

WorldPos = Annotated[SpatialTransform, Component("world_pos", metadata={"element_names": "q0,q1,q2,q3,x,y,z", "priority": 5})]
WorldVel = Annotated[SpatialMotion, Component("world_vel", metadata={"element_names": "ωx,ωy,ωz,x,y,z", "priority": 5})]
WorldAccel = Annotated[SpatialMotion, Component("world_accel", metadata={"element_names": "αx,αy,αz,x,y,z", "priority": 5})]
Force = Annotated[SpatialForce, Component("force", metadata={"element_names": "τx,τy,τz,x,y,z", "priority": 5})]
Inertia = Annotated[SpatialInertia, Component("inertia", metadata={"priority": 5})]
Seed = Annotated[Array, Component("seed", ComponentType.U64, metadata={"priority": 5})]
SimulationTick = Annotated[Array, Component("tick", ComponentType.F64, metadata={"priority": 7})]
SimulationTimeStep = Annotated[Array, Component("simulation_time_step", ComponentType.F64, metadata={"priority": 8})]


# ---- archetypes (__init__.py:560-669) ---------------------------------------------------------------------------------

_snake = re.compile(r"(?<!^)(?=[A-Z])")


def _host_rows(value) -> _np.ndarray:
    return _np.atleast_1d(_np.asarray(value.arr if hasattr(value, "arr") else value, dtype=_np.float64)).reshape(-1)


def _class_hints(cls) -> dict:
    try:
        return typing.get_type_hints(cls, include_extras=True)
    except Exception:            # a class body naming types of an enclosing function: the annotations are the objects themselves
        return {k: v for c in reversed(cls.__mro__) for k, v in getattr(c, "__annotations__", {}).items()}


class Archetype:
    """Base of user archetypes: a dataclass whose fields are annotated with component types."""

    @classmethod
    def archetype_name(cls) -> str:
        return _snake.sub("_", cls.__name__).lower()

    def component_data(self) -> List[Component]:
        return [Component.of(h) for h in _class_hints(type(self)).values()]

    def _fields(self):
        for attr, hint in _class_hints(type(self)).items():
            yield Component.name(hint), getattr(self, attr)

    def components(self):
        return {name: _host_rows(v) for name, v in self._fields() if not isinstance(v, _api.GravityEdge)}

    def edges(self):
        return [_api.GravityEdge(v.a, v.b, name) for name, v in self._fields() if isinstance(v, _api.GravityEdge)]


class C(Archetype):
    """el.C(X, value) / el.C((X, Y), (x, y)): components without declaring an archetype (__init__.py:643-661)."""

    def __init__(self, tys, values):
        if not isinstance(tys, tuple):
            tys, values = (tys,), (values,)
        self._items = [(Component.name(t), v) for t, v in zip(tys, values)]

    def component_data(self): return [Component(n) for n, _ in self._items]
    def _fields(self): return iter(self._items)


@dataclass
class Body(Archetype):
    """six_dof.rs:152-159 / __init__.py:663-669: identity pose, zero velocity, unit mass by default."""
    world_pos: WorldPos = dataclasses.field(default_factory=_api.SpatialTransform)
    world_vel: WorldVel = dataclasses.field(default_factory=_api.SpatialMotion)
    inertia: Inertia = dataclasses.field(default_factory=lambda: _api.SpatialInertia(1.0))
    force: Force = dataclasses.field(default_factory=_api.SpatialForce)
    world_accel: WorldAccel = dataclasses.field(default_factory=_api.SpatialMotion)


# ---- queries ----------------------------------------------------------------------------------------------------------

class _QueryType:
    """`el.Query[X, Y]` as an annotation."""

    def __init__(self, components: Tuple[Any, ...]):
        self.components = tuple(components)
        self.names = [Component.name(c) for c in self.components]


class RevEdge: ...


class TotalEdge: ...


class _GraphQueryType:
    """`el.GraphQuery[E]`, `el.GraphQuery[Annotated[E, el.RevEdge]]` (E's edges reversed) or `el.GraphQuery[el.TotalEdge]`
    (every ordered pair of distinct entities) as an annotation (elodin/__init__.py:427-439, graph.rs:113-158)."""

    def __init__(self, edge: Any):
        if edge is TotalEdge:
            self.edge_component = _api.World.TOTAL_EDGE
            return
        meta = getattr(edge, "__metadata__", ())
        name = Component.name(edge)
        self.edge_component = name + _api.World.REV_SUFFIX if (len(meta) > 1 and meta[1] is RevEdge) else name


class _Misuse(Exception):
    """A misuse of the decorator surface reported with the reference's own message (never deferred, see _Deferred)."""


class Query:
    """The values of one query inside a system being traced: one symbol (or symbolic spatial value) per component, the
    entity axis implicit — every entity of the query's join is one lane of the generated kernel."""

    def __class_getitem__(cls, item):
        item = item if isinstance(item, tuple) else (item,)
        return _QueryType(tuple(x for it in item for x in (it if isinstance(it, tuple) else (it,))))

    def __init__(self, components: Sequence[Any], values: Sequence[Any], indexed: Optional[set] = None):
        self.components, self.values = list(components), list(values)
        self._indexed = indexed if indexed is not None else set()

    @property
    def names(self): return [Component.name(c) for c in self.components]

    def map(self, out_tps, f) -> "Query":
        outs = out_tps if isinstance(out_tps, tuple) else (out_tps,)
        res = f(*self.values)
        res = tuple(res) if isinstance(res, (tuple, list)) and len(outs) > 1 else (res,)
        if len(res) != len(outs):
            raise TypeError(f"map function returned {len(res)} values for {len(outs)} output components")
        return Query(outs, [_lift(r) for r in res], self._indexed)

    map_seq = map

    def join(self, other: "Query") -> "Query":
        return Query(self.components + other.components, self.values + other.values, self._indexed)

    def __getitem__(self, index: int):
        if len(self.values) > 1:
            raise _Misuse("Cannot index into a query with multiple inputs")
        if index != 0:
            raise IndexError("only q[0] — the value of a one-entity component such as el.Seed — is available: the entity "
                             "axis of a query is the kernel's lane axis")
        self._indexed.add(self.names[0])
        return self.values[0]


class _Fold:
    """What `graph.edge_fold(...)` returns while a system is traced."""

    def __init__(self, edge_component, left, right, out, init, fn):
        self.edge_component, self.left, self.right, self.out, self.init, self.fn = edge_component, left, right, out, init, fn
        self.then = []          # maps chained on the fold's result: [(out component, fn)]
        self.out_type = None

    def map(self, out_tp, f) -> "_Fold":
        """`graph.edge_fold(...).map(T, f)` (examples/cube-sat/main.py:141-146): a per-entity map over the fold's output
        component, run right behind the fold."""
        self.then.append((out_tp, f))
        return self


class GraphQuery:
    def __class_getitem__(cls, item):
        return _GraphQueryType(item)

    def __init__(self, edge_component: str):
        self.edge_component = edge_component

    def edge_fold(self, left_query: Query, right_query: Query, return_type, init_value, fold_fn) -> _Fold:
        f = _Fold(self.edge_component, left_query.names, right_query.names, Component.name(return_type), init_value, fold_fn)
        f.out_type = return_type
        f.types = {n: c for q in (left_query, right_query) for n, c in zip(q.names, q.components)}
        return f


# ---- system / map ------------------------------------------------------------------------------------------------------

_BODY = ("world_pos", "world_vel", "inertia")


def _probe_value(component):
    """A symbol of the right shape for the decoration-time call (only the system's kind and its singleton queries are
    read off that call; the real trace happens at World.build with the columns' actual widths)."""
    name, w = Component.name(component), _width(component)
    if name in _BODY:
        return dict(zip(_BODY, _dsl._body_symbols()))[name]
    if name == "world_accel":
        return _dsl.SpatialMotion(_dsl.Vec([_dsl.leaf("aa" + c) for c in "xyz"]), _dsl.Vec([_dsl.leaf("al" + c) for c in "xyz"]))
    if name == "force":
        return _dsl.SpatialForce(_dsl.Vec([_dsl.leaf(f"acc{k}") for k in range(3)]), _dsl.Vec([_dsl.leaf(f"acc{k}") for k in range(3, 6)]))
    if name == "tick":
        return _dsl.leaf("tick")
    if _window_shape(component) is not None:
        rows, width = _window_shape(component)
        if _dsl._is_matrix_shape((rows, width)):      # a small matrix (a 3 x 3 covariance): registers (dsl_mat.Mat), not a window
            from . import dsl_mat
            return dsl_mat.Mat([[_dsl.leaf(f"probe:{name}:{i}_{j}") for j in range(width)] for i in range(rows)])
        return _dsl.Window(name, 0, rows, width, _dsl.leaf(f"probe:{name}:head"), 0)
    v = _dsl.Vec([_dsl.leaf(f"probe:{name}:{k}") for k in range(w or 1)])
    return v if len(v) > 1 else v[0]


def _annotations(func):
    try:
        hints = typing.get_type_hints(func, include_extras=True)
    except Exception:                                # annotations that only resolve in the defining scope
        hints = {}
    sig = inspect.signature(func)
    params = [(n, hints.get(n, p.annotation)) for n, p in sig.parameters.items()]
    return params, hints.get("return", sig.return_annotation)


def _typed(component, v):
    """A user component annotated with a spatial type (`Annotated[el.SpatialForce, el.Component("body_thrust")]`,
    `Annotated[el.Quaternion, el.Component("attitude_target")]`, examples/drone): its column is the flat vector, the function
    sees the typed value (spatial.rs layouts: [torque, force], [angular, linear], [x, y, z, w], [q, p])."""
    name = Component.name(component)
    if name in _BODY + ("force", "world_accel", "tick") or not isinstance(v, _dsl.Vec) or type(v) is not _dsl.Vec:
        return v
    o = _origin(component)
    if o is SpatialForce and len(v) == 6:
        return _dsl.SpatialForce(torque=v[:3], linear=v[3:])
    if o is SpatialMotion and len(v) == 6:
        return _dsl.SpatialMotion(v[:3], v[3:])
    if o is Quaternion and len(v) == 4:
        return _dsl.Quaternion(v)
    if o is SpatialTransform and len(v) == 7:
        return _dsl.SpatialTransform(_dsl.Quaternion(v[:4]), v[4:])
    return v


def _untyped(v):
    """The flat column value of what a system returned for a plain component (the inverse of _typed)."""
    if isinstance(v, _dsl.SpatialForce):
        return _dsl.np.concatenate([v.torque(), v.force()])
    if isinstance(v, _dsl.SpatialMotion):
        return _dsl.np.concatenate([v.angular(), v.linear()])
    if isinstance(v, _dsl.Quaternion):
        return v.vector()
    if isinstance(v, _dsl.SpatialTransform):
        return _dsl.np.concatenate([v.angular().vector(), v.linear()])
    return v


class _Deferred:
    """A decorated function whose decoration-time call on symbols failed (its body reads something that does not exist yet:
    examples/drone reads `Config.GLOBAL`, set by main.py after the modules with the @el.map functions were imported).  The
    reference traces at build time, so such scripts work there; here the function is lowered on first USE instead — when it
    is piped, handed to six_dof or built."""

    def __init__(self, func, error):
        self.func, self.error, self._made = func, error, None
        self.__name__ = getattr(func, "__name__", "system")

    def resolve(self):
        if self._made is None:
            self._made = _system_now(self.func)
        return self._made

    def __or__(self, other): return self.resolve() | _resolved(other)
    def __ror__(self, other): return _resolved(other) | self.resolve()
    def pipe(self, other): return self | other


def _resolved(x):
    """x with every deferred system inside it lowered."""
    if isinstance(x, _Deferred):
        return x.resolve()
    if isinstance(x, _dsl.Stages):
        return _dsl.Stages([_resolved(i) for i in x.items])
    if isinstance(x, (list, tuple)):
        return type(x)(_resolved(i) for i in x)
    return x


def system(func):
    """@el.system (__init__.py:160-185): parameters annotated el.Query[...] / el.GraphQuery[...]; returns the query that
    holds the written components.  Lowered now (one call on symbols); if that call fails, on first use (_Deferred)."""
    try:
        return _system_now(func)
    except (TypeError, IndexError, _Misuse):
        raise                      # a misuse the decorator reports (bad annotations, reading `force` outside six_dof ...)
    except Exception as e:  # noqa: BLE001
        return _Deferred(func, e)


def _system_now(func):
    params, _ret = _annotations(func)
    for pname, ann in params:
        if not isinstance(ann, (_QueryType, _GraphQueryType)):
            raise TypeError(f"system {func.__name__}: parameter {pname} must be annotated el.Query[...] or el.GraphQuery[...]")
    q_components = list(dict.fromkeys(c for _, a in params if isinstance(a, _QueryType) for c in a.components))
    by_name = {Component.name(c): c for c in q_components}

    def call(values: dict, indexed: set):
        args = [Query(a.components, [_typed(c, values[n]) for c, n in zip(a.components, a.names)], indexed) if isinstance(a, _QueryType)
                else GraphQuery(a.edge_component) for _, a in params]
        with _dsl.tracing():
            return func(*args)

    unreadable = "force is a stage value of the integrator: readable by effectors inside six_dof(sys=...) only"
    indexed: set = set()
    probe = call({n: _probe_value(c) for n, c in by_name.items()}, indexed)
    name = getattr(func, "__name__", "system")
    if isinstance(probe, _Fold):
        return _lower_fold(probe, name)
    if not isinstance(probe, Query):
        raise TypeError(f"system {name} must return a query (q.map(...)) or graph.edge_fold(...)")
    out_names = probe.names
    widths = {n: (_window_shape(c) or w) for n, c in {**by_name, **{Component.name(c): c for c in probe.components}}.items()
              if (w := _width(c)) is not None and n not in _BODY + ("force", "tick", "world_accel")}

    if "world_accel" in out_names:
        raise TypeError(f"system {name}: force / world_accel are produced inside six_dof — return el.Force alone and pass "
                        "the system as six_dof(sys=...)")
    if "force" in out_names:                         # `-> el.Force`: an effector of six_dof (six_dof.rs:161-203 `sys`)
        k_force = out_names.index("force")
        def effector_fn(**cols):             # the pipe tracer hands every plain column over as a Vec; shape-() ones are scalars
            cols = {n: (v[0] if isinstance(v, _dsl.Vec) and type(v) is _dsl.Vec and len(v) == 1 else v) for n, v in cols.items()}
            return call(cols, set()).values[k_force]
        eff = _dsl.Effector(effector_fn, widths)
        eff.params, eff.__name__ = list(by_name), name
        if len(out_names) == 1:
            return eff
        # `-> tuple[el.Force, Radius]` (examples/cube-sat/main.py:516-527): the force is the effector; the plain components are a
        # map of their own over the same query, which may not look at the force (a stage value, gone when the step is over)
        plain_names = [n for n in out_names if n != "force"]
        def plain_fn(**cols):
            cols = dict(cols)
            cols["force"] = _probe_value(by_name["force"]) if "force" in by_name else None
            out = call(cols, set())
            vals = {n: (v if n in _BODY else _untyped(v)) for n, v in zip(out.names, out.values) if n != "force"}
            flat = [e for v in vals.values() for e in (v.e if isinstance(v, _dsl.Vec) else [_dsl._lift(v)]) if isinstance(e, _dsl.Expr)]
            if any(l.startswith("acc") and l[3:].isdigit() for l in _dsl._leaves_of(flat)):
                raise TypeError(f"system {name}: {unreadable}")
            return vals
        s2 = _dsl.System(plain_fn, widths, 1, tuple(sorted(indexed)))
        s2.params, s2.__name__, s2.aliases = [n for n in by_name if n != "force"], name + "_components", False
        return _dsl.Stages([eff, s2])
    if "force" in by_name:
        raise TypeError(f"system {name}: {unreadable}")

    def system_fn(**cols):
        out = call(cols, set())
        return {n: (v if n in _BODY else _untyped(v)) for n, v in zip(out.names, out.values)}
    s = _dsl.System(system_fn, widths, 1, tuple(sorted(indexed)))
    s.params, s.__name__ = list(by_name), name
    s.aliases = False            # component names are exact here: a user component may be called `accel`, `pos`, `vel`
    return s


def _fold_init(v) -> list:
    """The fold's initial value as numbers: arrays, 0-d arrays written while tracing (`np.array(0.0)`), spatial zeros."""
    if hasattr(v, "arr"):                      # api.SpatialForce() / SpatialMotion(): [angular, linear]
        return [float(x) for x in _np.asarray(v.arr, dtype=_np.float64).reshape(-1)]
    v = _untyped(v)
    if isinstance(v, _dsl.Expr):
        v = _dsl.Vec([v])
    if isinstance(v, _dsl.Vec):
        if not all(_dsl._lift(e).is_const() for e in v.e):
            raise TypeError("an edge_fold's initial value must be constant")
        return [float(_dsl._lift(e).value) for e in v.e]
    return [float(x) for x in _np.atleast_1d(_np.asarray(v, dtype=_np.float64)).reshape(-1)]


def _lower_fold(fold: _Fold, name: str):
    body_pair = ["world_pos", "inertia"]
    if fold.out == "force" and fold.left == body_pair and fold.right == body_pair:
        init = fold.init.arr if hasattr(fold.init, "arr") else None
        if init is None or _np.any(init != 0.0):
            raise ValueError("an edge_fold returning el.Force starts from el.SpatialForce() (zero)")
        fn = fold.fn
        ef = _dsl.EdgeFold(lambda acc, a_pos, a_inertia, b_pos, b_inertia: fn(acc, a_pos, a_inertia, b_pos, b_inertia),
                           fold.edge_component)
        ef.__name__ = name
        return ef
    init = _fold_init(fold.init)
    fn, n_args = fold.fn, 1 + len(fold.left) + len(fold.right)
    types = getattr(fold, "types", {})

    def spatial(cname, v):      # Body components and spatially typed user components reach the fold function as the reference's types
        if cname == "world_pos":
            return _dsl.SpatialTransform(_dsl.Quaternion(_dsl.Vec(v.e[:4])), _dsl.Vec(v.e[4:]))
        if cname == "world_vel":
            return _dsl.SpatialMotion(_dsl.Vec(v.e[:3]), _dsl.Vec(v.e[3:]))
        if cname == "inertia":
            return _dsl.SpatialInertia(_dsl.Vec(v.e[:3]), v.e[6])
        if cname == "force":
            return _dsl.SpatialForce(torque=_dsl.Vec(v.e[:3]), linear=_dsl.Vec(v.e[3:]))
        return _typed(types[cname], v) if cname in types else v

    def fixed_arity(*args):
        names = (fold.out,) + tuple(fold.left) + tuple(fold.right)
        typed = [spatial(n, a) for n, a in zip(names, args)]
        if fold.out != "force" and fold.out_type is not None:
            typed[0] = _typed(fold.out_type, args[0])
        return _untyped(fn(*typed))
    fixed_arity.__signature__ = inspect.Signature([inspect.Parameter(f"a{k}", inspect.Parameter.POSITIONAL_ONLY) for k in range(n_args)])
    if fold.out == "force":
        # `-> el.Query[el.Force]` folded over other components (examples/cube-sat/main.py:492-504: the satellite's force is the sum
        # of its wheels' torques).  Force is a stage value of the integrator, a fold reads other entities' rows: the fold runs as
        # a stand-alone fold into a hidden column in front of the force evaluation, and an effector puts that value in the
        # place of the force on the rows that have out-edges (the others keep theirs) — World.build makes the two columns
        hidden, flag = f"fold_force:{name}", f"fold_src:{name}"
        gf = _dsl.GraphFold(fixed_arity, fold.edge_component, fold.left, fold.right, hidden, init)
        gf.__name__, gf.hidden_force = name, (hidden, flag)

        def put(**cols):
            force, h, src = cols["force"], cols[hidden], cols[flag]
            m = (src[0] if isinstance(src, _dsl.Vec) else src) > 0.5
            pick = lambda a, b: _dsl.Vec([_dsl.Expr("select", (m, _dsl._lift(x), _dsl._lift(y))) for x, y in zip(a.e, b.e)])
            zero = _dsl.Vec([0.0, 0.0, 0.0])
            out = _dsl.SpatialForce(linear=pick(_dsl.Vec(h.e[3:]), force._f), _tw=pick(_dsl.Vec(h.e[:3]), force._tw), _tb=pick(zero, force._tb))
            out._q = force._q
            return out
        eff = _dsl.Effector(put, {hidden: 6, flag: 1})
        eff.params, eff.__name__ = ["force", hidden, flag], name + "_put"
        if fold.then:
            raise NotImplementedError("edge_fold(...).map(...) on a fold returning el.Force")
        return _dsl.Stages([gf, eff])
    gf = _dsl.GraphFold(fixed_arity, fold.edge_component, fold.left, fold.right, fold.out, init)
    gf.__name__ = name
    if not fold.then:
        return gf
    stages = [gf]                                     # fold | map | map ...: a pipe of its own (dsl.Stages)
    in_tp = fold.out_type
    def make(out_tp, f):
        def mapped(q):
            return q.map(out_tp, f)
        return mapped
    for k, (out_tp, f) in enumerate(fold.then):
        mapped = make(out_tp, f)
        mapped.__name__ = f"{name}_map{k}"
        mapped.__annotations__ = {"q": _QueryType((in_tp,))}
        stages.append(system(mapped))
        in_tp = out_tp
    return _dsl.Stages(stages)


def _map(func, seq: bool):
    params, ret = _annotations(func)
    if typing.get_origin(ret) is tuple:
        ret = tuple(typing.get_args(ret))
    query_t = _QueryType(tuple(a for _, a in params))

    def inner(q):
        return (q.map_seq if seq else q.map)(ret, func)
    inner.__name__ = getattr(func, "__name__", "map")
    inner.__annotations__ = {"q": query_t}
    return system(inner)


def map(func):          # noqa: A001  (the reference's name)
    """@el.map (__init__.py:360-374): parameters and return annotated with component types."""
    return _map(func, False)


def map_seq(func):
    """@el.map_seq (__init__.py:377-396)."""
    return _map(func, True)


# `a.pipe(b)` beside `a | b` (system.rs:1001-1011)
def _pipe(self, other):
    return self | other


for _cls in (_dsl.System, _dsl.Stages, _dsl.Effector, _dsl.Pipe, _dsl.GraphFold, _api.System):
    _cls.pipe = _pipe


def six_dof(time_step: Optional[float] = None, sys=None, integrator: Integrator = Integrator.Rk4):
    """elodin.six_dof (lib.rs:106-127): `sys` is what @el.map / @el.system made of functions returning el.Force —
    effectors piped with `|`, optionally closed by one edge_fold system."""
    sys = _resolved(sys)
    if isinstance(sys, _dsl.Stages):                 # `gravity | drag` of two effector-kind systems
        sys = sys.items
    plain = []
    if isinstance(sys, (list, tuple)):
        # maps and folds that write plain components piped among the force effectors (examples/drone/sim.py:193 `gravity | drag |
        # motor_thrust_response | body_thrust | apply_body_forces`; examples/cube-sat/main.py:699-710 runs its whole flight
        # software inside six_dof): with the semi-implicit integrator the pipe is evaluated once per step on the state the
        # step starts from, so they run in front of the force evaluation in pipe order (nothing they read is a force).  What
        # that reordering must not change — a force effector reading a component that a LATER plain system writes — is checked
        # when the program is traced (dsl.TracedProgram, `pipe_index`)
        flat = []
        def walk(items):
            for it in items:
                walk(it.items) if isinstance(it, _dsl.Stages) else flat.append(it)
        walk(sys)
        import copy
        for k, it in enumerate(flat):
            if getattr(it, "pipe_index", None) is not None:      # the same system object piped twice: each place its own index
                flat[k] = it = copy.copy(it)
            it.pipe_index = k
        plain = [s for s in flat if isinstance(s, (_dsl.System, _dsl.GraphFold))]
        sys = [s for s in flat if not isinstance(s, (_dsl.System, _dsl.GraphFold))]
        if plain and integrator != Integrator.SemiImplicit:
            raise NotImplementedError("maps writing plain components inside six_dof(sys=...) are supported with the "
                                      "semi-implicit integrator only (RK4 would evaluate them once per stage)")
        folds = [s for s in sys if isinstance(s, _dsl.EdgeFold)]
        effs = [s for s in sys if not isinstance(s, _dsl.EdgeFold)]
        if folds and effs:
            raise TypeError("a user edge_fold cannot be piped with generated effectors: fold it in its own six_dof(sys=...)")
        sys = folds[0] if folds else _dsl.pipe(*effs)
    out = _api.six_dof(time_step, sys, integrator)
    out.stage_systems = plain
    return out


_SYSTEM_KINDS = (_dsl.System, _dsl.Stages, _dsl.Effector, _dsl.Pipe, _dsl.EdgeFold, _dsl.GraphFold, _api.System, _Deferred)


class _SystemMeta(type):
    def __instancecheck__(cls, x): return isinstance(x, _SYSTEM_KINDS)


class System(metaclass=_SystemMeta):
    """el.System: the type of everything the decorators and `|` produce — for `-> el.System` / `el.System | None` annotations
    (examples/falcon9/sim.py:1476) and isinstance checks."""


# ---- world --------------------------------------------------------------------------------------------------------------

class World(_api.World):
    """el.World(): spawn archetypes, build.  `exec.history([...])` works as in the reference (one row per telemetry
    commit); pass history=False to build() for long runs that only read the final columns."""

    def insert(self, eid, archetypes) -> None:
        if not isinstance(archetypes, (list, tuple)):
            archetypes = [archetypes]
        flat = []
        for a in archetypes:
            flat.append(a)
            flat.extend(a.edges() if isinstance(a, Archetype) else [])
        super().insert(eid, flat)

    HISTORY_AUTO_LIMIT = 32 << 20        # bytes of component data per telemetry sample up to which history defaults to on

    def build(self, system, simulation_rate: float = 120.0, telemetry_rate: Optional[float] = None, device: int = 0,
              backend: str = "hip", history: Optional[bool] = None, _dry: bool = False):
        """history=None: record telemetry samples for exec.history() unless one sample of this world exceeds
        HISTORY_AUTO_LIMIT (then a warning says so); True / False force it."""
        system = _resolved(system)
        if isinstance(system, _dsl.Effector):
            raise TypeError("a system returning el.Force is a six_dof effector: build(el.six_dof(sys=...))")
        if _dry:
            return super().build(system, simulation_rate=simulation_rate, telemetry_rate=telemetry_rate, device=device, backend=backend, _dry=True)
        ex = super().build(system, simulation_rate=simulation_rate, telemetry_rate=telemetry_rate, device=device, backend=backend)
        if history is None:
            sample = sum(self.column(c)[0].nbytes for c in self._components)
            history = sample <= self.HISTORY_AUTO_LIMIT
            if not history:
                import warnings
                warnings.warn(f"exec.history() is off: one telemetry sample of this world is {sample >> 20} MiB "
                              "(pass history=True to record anyway, history=False to silence this)", RuntimeWarning, stacklevel=2)
        if history:
            _api.record_history(ex, self)
        return ex


WorldBuilder = World
