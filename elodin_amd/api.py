"""Host-side mirror of the reference's Python surface for the six_dof path.

Same names and argument meaning as `elodin` (libs/nox-py/python/elodin/__init__.py,
elodin.pyi) for the pieces this backend covers, so a sim script / test for this path reads like the
reference's:

    w = World()
    a = w.spawn(Body(world_pos=SpatialTransform(linear=[0.89, 0, 0]), inertia=SpatialInertia(1 / G)), name="A")
    w.spawn(GravityEdge(a, b)) ...
    sys = six_dof(sys=gravity_newton(G))
    exec = w.build(sys, simulation_rate=120.0)
    exec.run(100)
    exec.column_array("world_pos")

What differs, on purpose: `sys` is a pipe of built-in effector descriptors (include/sixdof_hip.h
sixdof_effector_kind) instead of arbitrary JAX, and `build` returns an executor bound to the HIP
backend (there is no other backend; no GPU -> BackendError).  Column storage follows
libs/nox-py/src/world.rs:23-45,193-229: per component a row-major buffer + entity ids in spawn order,
ids sequential from `entity_len`, entity 0 = "Globals" holding tick and simulation_time_step.
"""
from __future__ import annotations

import enum
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import os
import numpy as np

from . import _lib as L
from .exec import Effector, HipExec, TickTimings


# ---- spatial value types (libs/nox-py/src/spatial.rs:21-449) --------------------------------------------

class Quaternion:
    """Scalar-last [x, y, z, w] (libs/nox/src/quaternion.rs:90-102)."""

    def __init__(self, arr=(0.0, 0.0, 0.0, 1.0)):
        self.arr = np.asarray(arr, dtype=np.float64).reshape(4)

    @staticmethod
    def identity() -> "Quaternion":
        return Quaternion()

    @staticmethod
    def from_axis_angle(axis, angle) -> "Quaternion":
        axis = np.asarray(axis, dtype=np.float64)
        axis = axis / np.sqrt(np.dot(axis, axis))
        half = float(angle) / 2.0
        return Quaternion(np.concatenate([axis * np.sin(half), [np.cos(half)]]))

    def vector(self) -> np.ndarray:
        return self.arr

    # host-side algebra (spawn-time data: a tilted motor axis, an initial attitude) — the traced twin is dsl.Quaternion
    __array_ufunc__ = None                                     # `q @ ndarray` / `ndarray * q` must not broadcast over the object

    def inverse(self) -> "Quaternion":                         # quaternion.rs:152-155: conj / |q|^2
        x, y, z, w = self.arr
        return Quaternion(np.array([-x, -y, -z, w]) / float(np.dot(self.arr, self.arr)))

    def normalize(self) -> "Quaternion":                       # quaternion.rs:147-149
        return Quaternion(self.arr / np.sqrt(np.dot(self.arr, self.arr)))

    def __mul__(self, o):                                      # Hamilton product, quaternion.rs:268-281
        if not isinstance(o, Quaternion):
            return NotImplemented
        l, r = self.arr, o.arr
        return Quaternion([l[3] * r[0] + l[0] * r[3] + l[1] * r[2] - l[2] * r[1],
                           l[3] * r[1] - l[0] * r[2] + l[1] * r[3] + l[2] * r[0],
                           l[3] * r[2] + l[0] * r[1] - l[1] * r[0] + l[2] * r[3],
                           l[3] * r[3] - l[0] * r[0] - l[1] * r[1] - l[2] * r[2]])

    def __matmul__(self, v):                                   # q (x) (v, 0) (x) q^-1, quaternion.rs:283-305
        if hasattr(v, "arr") and not isinstance(v, Quaternion):            # a spatial value: both halves
            a = v.arr
            return type(v)(arr=np.concatenate([self @ a[:3], self @ a[3:]]))
        if not isinstance(v, np.ndarray) and not isinstance(v, (list, tuple)):
            return NotImplemented
        v = np.asarray(v, dtype=np.float64).reshape(3)
        u = self.arr[:3]
        t = np.cross(u, v) * (2.0 / float(np.dot(self.arr, self.arr)))
        return v + t * self.arr[3] + np.cross(u, t)


class SpatialTransform:
    def __init__(self, arr=None, angular: Optional[Quaternion] = None, linear=None):
        if arr is not None:
            self.arr = np.asarray(arr, dtype=np.float64).reshape(7)
        else:
            q = (angular or Quaternion.identity()).vector()
            p = np.zeros(3) if linear is None else np.asarray(linear, dtype=np.float64)
            self.arr = np.concatenate([q, p])

    def linear(self): return self.arr[4:]
    def angular(self): return Quaternion(self.arr[:4])


class _Spatial6:
    def __init__(self, arr=None, first=None, linear=None):
        if arr is not None:
            self.arr = np.asarray(arr, dtype=np.float64).reshape(6)
        else:
            a = np.zeros(3) if first is None else np.asarray(first, dtype=np.float64)
            b = np.zeros(3) if linear is None else np.asarray(linear, dtype=np.float64)
            self.arr = np.concatenate([a, b])

    def linear(self): return self.arr[3:]


class SpatialMotion(_Spatial6):
    """el.SpatialMotion(angular=None, linear=None) (libs/nox-py/src/spatial.rs:121-133), positional too: `SpatialMotion(w, v)`
    (examples/cube-sat/main.py:541).  One six-element argument is taken as the whole [angular, linear] array."""

    def __init__(self, angular=None, linear=None, arr=None):
        if arr is None and linear is None and angular is not None and np.size(angular) == 6:
            arr, angular = angular, None
        super().__init__(arr, angular, linear)

    def angular(self): return self.arr[:3]


class SpatialForce(_Spatial6):
    def __init__(self, arr=None, torque=None, linear=None):
        super().__init__(arr, torque, linear)

    def torque(self): return self.arr[:3]
    def force(self): return self.arr[3:]


class SpatialInertia:
    """[Ixx, Iyy, Izz, px, py, pz, m]; inertia defaults to ones(3) * mass (spatial.rs:392-405)."""

    def __init__(self, mass, inertia=None):
        mass = float(mass)
        diag = np.ones(3) * mass if inertia is None else np.asarray(inertia, dtype=np.float64)
        self.arr = np.concatenate([diag, np.zeros(3), [mass]])

    def mass(self): return self.arr[6]
    def inertia_diag(self): return self.arr[:3]


class EntityId(int):
    pass


class Integrator(enum.Enum):  # integrator/mod.rs:7-10
    Rk4 = L.RK4
    SemiImplicit = L.SEMI_IMPLICIT


# ---- archetypes ---------------------------------------------------------------------------------------------

@dataclass
class Body:
    """six_dof.rs:152-159 / __init__.py:663-669: identity pose, zero velocity, unit mass by default."""
    world_pos: SpatialTransform = field(default_factory=SpatialTransform)
    world_vel: SpatialMotion = field(default_factory=SpatialMotion)
    inertia: SpatialInertia = field(default_factory=lambda: SpatialInertia(1.0))
    force: SpatialForce = field(default_factory=SpatialForce)
    world_accel: SpatialMotion = field(default_factory=SpatialMotion)

    def components(self):
        return {"world_pos": self.world_pos.arr, "world_vel": self.world_vel.arr, "inertia": self.inertia.arr,
                "force": self.force.arr, "world_accel": self.world_accel.arr}


@dataclass
class C:
    """el.C(Component, value): one extra component on the entity being spawned (f64 vector)."""
    name: str
    value: Sequence[float]

    def components(self):
        return {self.name: np.atleast_1d(np.asarray(self.value, dtype=np.float64))}


@dataclass
class GravityEdge:
    """An Edge component entity (el.Edge(from, to), graph.rs:17-41) — consumes an entity id like any spawn."""
    a: int
    b: int
    component: str = "gravity_edge"


def skew(v) -> np.ndarray:
    """el.skew (libs/nox-py/python/elodin/__init__.py): the cross-product matrix of a 3-vector."""
    x, y, z = (float(c) for c in np.asarray(v, dtype=np.float64).reshape(3))
    return np.array([[0.0, -z, y], [z, 0.0, -x], [-y, x, 0.0]])


Edge = GravityEdge      # el.Edge(a, b) under an archetype's own component name: Edge(a, b, component="e")


# ---- systems --------------------------------------------------------------------------------------------------

@dataclass
class Effectors:
    """A pipe of built-in effector descriptors; compose with `|` like reference systems (system.rs:1001-1011)."""
    ops: List[Effector] = field(default_factory=list)
    edge_component: Optional[str] = None

    def __or__(self, other) -> "Effectors":
        from . import dsl as _dsl
        if isinstance(other, _dsl.EdgeFold):     # user-written fold function closes the pipe
            return Effectors(self.ops + [other], other.edge_component)
        return Effectors(self.ops + other.ops, other.edge_component or self.edge_component)


def uniform_gravity(g=(0.0, 0.0, -9.81)) -> Effectors:          # examples/ball/sim.py:57-59
    return Effectors([Effector(L.EFF_UNIFORM_GRAVITY, tuple(g))])


def constant_wrench(torque=(0, 0, 0), force=(0, 0, 0)) -> Effectors:   # test_all.py:353-356
    return Effectors([Effector(L.EFF_CONST_WRENCH, tuple(torque) + tuple(force))])


def body_torque(component: str) -> Effectors:                     # apollo-lander/sim.py:396-398
    return Effectors([Effector(L.EFF_BODY_TORQUE, (), aux_name=component)])


def body_force(component: str) -> Effectors:                      # apollo-lander/sim.py:391-394
    return Effectors([Effector(L.EFF_BODY_FORCE, (), aux_name=component)])


def world_torque(component: str) -> Effectors:                    # force + SpatialForce(torque=<component column>)
    return Effectors([Effector(L.EFF_WORLD_TORQUE, (), aux_name=component)])


def world_force(component: str) -> Effectors:                     # force + SpatialForce(linear=<component column>)
    return Effectors([Effector(L.EFF_WORLD_FORCE, (), aux_name=component)])


def ball_drag(wind_component: str, cd=0.5, rho=1.225, area=2 * 3.1415 * 0.2**2) -> Effectors:  # ball/sim.py:96-116
    return Effectors([Effector(L.EFF_BALL_DRAG, (cd, rho, area), aux_name=wind_component)])


def gravity_newton(G: float, edge_component: str = "gravity_edge") -> Effectors:   # three-body/main.py:56-78
    return Effectors([Effector(L.EFF_EDGE_GRAVITY_NEWTON, (G,))], edge_component)


def gravity_softened(K: float, eps: float, edge_component: Optional[str] = "gravity_edge") -> Effectors:
    """examples/n-body/sim.py:344-369 over explicit edges; edge_component=None = complete graph, tiled kernel."""
    kind = L.EFF_EDGE_GRAVITY_SOFTENED if edge_component else L.EFF_ALLPAIRS_GRAVITY_SOFTENED
    return Effectors([Effector(kind, (K, eps))], edge_component)


@dataclass
class System:
    time_step: Optional[float]
    effectors: object   # Effectors (built-in ops) or dsl.Pipe (generated)
    integrator: Integrator

    # `pre_system | six_dof(...) | post_system` with systems written against elodin_amd.dsl
    def __or__(self, other):
        from . import dsl as _dsl
        return _dsl.Stages([self]) | other

    def __ror__(self, other):
        from . import dsl as _dsl
        return _dsl.Stages([other]) | self


def six_dof(time_step: Optional[float] = None, sys=None, integrator: Integrator = Integrator.Rk4) -> System:
    """elodin.six_dof(time_step=None, sys=None, integrator=Integrator.Rk4) — lib.rs:106-127, six_dof.rs:161-203.
    `sys` is a pipe of built-in effector descriptors (`uniform_gravity() | ...`) or of user-written effectors
    traced by elodin_amd.dsl (`dsl.pipe(f, g)` / `f | g`), which are compiled into the step kernel at build()."""
    if not isinstance(integrator, Integrator):
        raise TypeError("integrator must be an Integrator")
    from . import dsl as _dsl
    if isinstance(sys, _dsl.Effector):
        sys = _dsl.pipe(sys)
    if isinstance(sys, _dsl.EdgeFold):
        sys = Effectors([sys], sys.edge_component)
    return System(time_step, sys if sys is not None else Effectors(), integrator)


# ---- world ----------------------------------------------------------------------------------------------------

class World:
    """Python face of the C++ column store (csrc/world.cpp, the analogue of libs/nox-py/src/world.rs)."""

    def __init__(self):
        import ctypes as C
        self._lib = L.lib()
        self._w = C.c_void_p(self._lib.sixdof_world_create())
        self._names: Dict[int, str] = {0: "Globals"}
        self.entity_ids_by_name: Dict[str, int] = {}      # spawn(..., id="ore_sat"): the entity's database name -> its id
        self._edges: Dict[str, List[tuple]] = {}
        self._components: List[str] = []

    def __del__(self):
        try:
            if getattr(self, "_w", None) and self._w.value:
                self._lib.sixdof_world_destroy(self._w)
        except Exception:
            pass

    @property
    def entity_len(self) -> int:
        return int(self._lib.sixdof_world_entity_len(self._w))

    def spawn(self, archetypes, name: Optional[str] = None, id: Optional[str] = None) -> EntityId:   # noqa: A002 (the reference's keyword)
        """WorldBuilder.spawn(spawnable, name=None, id=None) (world_builder.rs:268-310); `id` is the entity's snake_case name in
        the database (the prefix of its CSV columns), kept in `entity_ids_by_name`."""
        eid = EntityId(self._lib.sixdof_world_spawn(self._w))
        self.insert(eid, archetypes)
        if name is not None:
            self._names[int(eid)] = name
        if id is not None:
            self.entity_ids_by_name[id] = int(eid)
        return eid

    def insert(self, eid: EntityId, archetypes) -> None:
        import ctypes as C
        if not isinstance(archetypes, (list, tuple)):
            archetypes = [archetypes]
        for arch in archetypes:
            if isinstance(arch, GravityEdge):
                self._edges.setdefault(arch.component, []).append((int(arch.a), int(arch.b)))
                continue
            for cname, value in arch.components().items():
                value = np.ascontiguousarray(value, dtype=np.float64)   # lib.rs:64-75: C-contiguous rows
                if value.ndim == 2:        # a matrix component ((3, 3) covariance, (480, 3) sample window): its row-major bytes
                    value = value.reshape(-1)
                dims = (C.c_uint64 * 2)(value.shape[0] if value.ndim else 1, 0)
                rc = self._lib.sixdof_world_insert(self._w, int(eid), cname.encode(), L.PRIM_F64, dims, 1,
                                                   value.ctypes.data, value.nbytes)
                if rc == L.ERR_VALUE_SIZE_MISMATCH:
                    raise ValueError(f"component {cname}: value size mismatch")   # Error::ValueSizeMismatch
                if rc != L.OK:
                    raise ValueError(self._lib.sixdof_world_last_error(self._w).decode())
                if cname not in self._components:
                    self._components.append(cname)

    def column(self, name: str):
        """(rows [n,w] float64 copy, entity ids [n] uint64) of a component; KeyError = Error::ComponentNotFound."""
        import ctypes as C
        c = L.Column()
        if self._lib.sixdof_world_column(self._w, L.component_id(name), C.byref(c)) != L.OK:
            raise KeyError(name)
        n, w = int(c.n_rows), int(c.dims[0]) if c.ndim else 1
        rows = np.ctypeslib.as_array(C.cast(c.host_ptr, C.POINTER(C.c_double)), shape=(n, w)).copy() if n else np.zeros((0, w))
        ids = np.ctypeslib.as_array(c.entity_ids, shape=(n,)).copy() if n else np.zeros(0, dtype=np.uint64)
        return rows, ids.astype(np.uint64)

    TOTAL_EDGE, REV_SUFFIX = "*total_edge", "~rev"   # edge-component spellings of el.TotalEdge / Annotated[E, el.RevEdge]

    def edge_pairs(self, edge_component: str):
        """(from ids, to ids) of a GraphQuery's edges, spawn order (graph.rs:113-175).  `E~rev` = every edge of component E
        reversed (GraphQueryInner.from_builder(builder, edge_id, reverse=True), elodin/__init__.py:432-439); `*total_edge`
        = every ordered pair of distinct entities of the world, ascending (graph.rs:144-158, GraphQuery<TotalEdge>) —
        endpoints that do not carry the queried components drop out when the fold joins them."""
        if edge_component == self.TOTAL_EDGE:
            n = int(self.entity_len)
            a = np.repeat(np.arange(n, dtype=np.uint64), n)
            b = np.tile(np.arange(n, dtype=np.uint64), n)
            keep = a != b
            return a[keep], b[keep]
        reverse = edge_component.endswith(self.REV_SUFFIX)
        name = edge_component[: -len(self.REV_SUFFIX)] if reverse else edge_component
        pairs = self._edges.get(name)
        if pairs is None:
            raise KeyError(name)
        frm = np.array([a for a, _ in pairs], dtype=np.uint64)
        to = np.array([b for _, b in pairs], dtype=np.uint64)
        return (to, frm) if reverse else (frm, to)

    def generated_sources(self, system, simulation_rate: float = 120.0, dtype: str = "float64") -> Dict[str, str]:
        """The HIP sources World.build would generate for `system` (kernel name -> text), without touching a device: what a
        user script compiles to.  Two scripts that build the same program produce the same text (tests compare a reference
        script imported under elodin_amd.compat with its respelling this way).  Built-in effector ops generate nothing."""
        plan = self.build(system, simulation_rate=simulation_rate, _dry=True)
        from . import codegen, dsl as _dsl
        out = {}
        effs = plan["effectors"]
        if isinstance(effs, _dsl.Effector):
            effs = _dsl.pipe(effs)
        if isinstance(effs, (_dsl.Pipe, _dsl.Program)):
            cols = plan["columns"] or {}
            widths = {k: (tuple(int(x) for x in np.shape(v)[1:]) if np.ndim(v) == 3 else int(np.atleast_2d(np.asarray(v)).shape[-1]))
                      for k, v in cols.items() if v is not None}
            out["step"] = codegen.generate_source(effs.trace(widths), dtype, plan["integrator"])
        else:
            for e in effs:
                if isinstance(e, _dsl.EdgeFold):
                    small = int(self.entity_len) <= 256 and os.environ.get("SIXDOF_PAIR_SMALL", "")[:1] != "0"     # as exec.HipExec does
                    out["pair"] = codegen.generate_pair_source(e.trace(), integrator=plan["integrator"], small=small)
        return out

    def build(self, system: System, simulation_rate: float = 120.0, telemetry_rate: Optional[float] = None,
              device: int = 0, backend: str = "hip", _dry: bool = False) -> "Exec":
        """World.build (world_builder.rs:1737-1780): validate rates, fix globals, bind the backend."""
        import os
        backend = os.environ.get("ELODIN_BACKEND", backend)    # same override as world_builder.rs:248
        if backend != "hip":
            raise ValueError(f"unknown backend {backend!r}: this package provides 'hip' only")
        if telemetry_rate is not None and telemetry_rate <= 0.0:
            raise ValueError(f"telemetry_rate must be > 0 Hz, got {telemetry_rate}")
        # validate_rates + set_globals in the C++ world (world_builder.rs:211-243)
        if self._lib.sixdof_world_set_rates(self._w, float(simulation_rate), float(telemetry_rate or 0.0)) != L.OK:
            raise ValueError(self._lib.sixdof_world_last_error(self._w).decode())
        ticks_per_telemetry = int(self._lib.sixdof_world_ticks_per_telemetry(self._w))
        dt = float(self._lib.sixdof_world_time_step(self._w))
        from . import dsl as _dsl
        if isinstance(system, _dsl.GraphFold):     # a stand-alone edge_fold system over plain components (test_all.py:117-142)
            from .graph_exec import GraphFoldExec
            edges = self.edge_pairs(system.edge_component)
            names = dict.fromkeys(system.left + system.right + (system.out,))
            return GraphFoldExec(system, {n: self.column(n) for n in names}, edges, device=device)
        # one build = one trace: whatever trace the system's effector pipe kept from an earlier build is dropped here (its functions
        # may close over values that changed since), everything below shares the fresh one and hands it to the executor as it is
        eff0 = getattr(system, "effectors", None)
        if isinstance(eff0, (_dsl.Pipe, _dsl.Program)) and not isinstance(eff0, _dsl.FrozenProgram) and getattr(eff0, "_traced", None) is not None:
            eff0._traced = None
        program_stages = None
        substeps = 1
        if isinstance(system, _dsl.System):
            system = _dsl.Stages([system])
        if isinstance(system, System) and getattr(system, "stage_systems", None):
            system = _dsl.Stages([system])       # six_dof(sys=...) alone, with maps / folds among its effectors: a program too
        if isinstance(system, _dsl.Stages):      # pre | six_dof(effectors) | post  -> one generated program
            six = [k for k, it in enumerate(system.items) if isinstance(it, System)]
            if len(six) > 1:
                # `pre | (six_dof | post) x k` with the SAME six_dof and post objects repeated (examples/drone/sim.py:173-208
                # `inner_loop`): k integrator sub-steps per tick = k executor ticks per world tick (dsl.Program.substeps)
                seg = system.items[six[0]:six[1]]
                k = len(six)
                tail = system.items[six[0]:]
                if len(tail) != k * len(seg) or any(a is not b for j in range(k) for a, b in zip(seg, tail[j * len(seg):(j + 1) * len(seg)])):
                    raise ValueError("a system pipe with several six_dof(...) stages must repeat one `six_dof | post` segment")
                substeps = k
                system = _dsl.Stages(system.items[:six[0]] + seg)
                six = six[:1]
            if not six:     # `w.build(sys)` with per-entity systems only (test_all.py:86-114): no integration stage
                program_stages = (system.items, [])
                system = System(None, Effectors(), Integrator.Rk4)
                system.no_six_dof = True
            else:
                program_stages = (system.items[:six[0]], system.items[six[0] + 1:])
                system = system.items[six[0]]
                if substeps > 1:      # the systems in front run on the first sub-step of every tick; all see the world tick
                    import copy
                    def at_rate(s_, every):
                        c = copy.copy(s_)
                        c.every, c.phase, c.tick_substeps = every, 1 % every, substeps
                        return c
                    program_stages = ([at_rate(s_, substeps) for s_ in program_stages[0]], [at_rate(s_, 1) for s_ in program_stages[1]])
                extra = list(getattr(system, "stage_systems", []) or [])
                if extra:             # maps among the force effectors (frontend.six_dof): every step, behind the systems in front
                    import copy
                    def stepwise(s_):
                        c = copy.copy(s_)
                        c.tick_substeps = substeps
                        return c
                    program_stages = (program_stages[0] + [stepwise(s_) for s_ in extra], program_stages[1])
            for it in program_stages[0] + program_stages[1]:
                if not isinstance(it, (_dsl.System, _dsl.GraphFold)):
                    raise TypeError("systems piped around six_dof must be elodin_amd.dsl systems (or stand-alone edge folds)")
        if program_stages is None and isinstance(system.effectors, _dsl.Pipe) and "world_pos" in self._components:
            # generated effectors reading a component that only some Bodies carry: the program path knows presence masks
            body_ids = self.column("world_pos")[1]
            for name, _w in system.effectors.trace().columns:
                if name in self._components and not np.all(np.isin(body_ids, self.column(name)[1])):
                    program_stages = ([], [])
                    break
        synthesized_body = None
        if "world_pos" not in self._components and program_stages is not None and getattr(system, "no_six_dof", False):
            # a world of plain components (no Body anywhere): the row set is every entity carrying a component the systems
            # touch; the executor still wants Body columns, so identity ones stand in (never read by the systems)
            widths0 = {name: int(self.column(name)[0].shape[1]) for name in self._components}
            plain = [[it for it in part if not isinstance(it, _dsl.GraphFold)] for part in program_stages]
            used = [n for n, _ in _dsl.Program(plain[0], _dsl.Pipe([]), plain[1]).trace(widths0).columns]
            used += [n for it in program_stages[0] + program_stages[1] if isinstance(it, _dsl.GraphFold)
                     for n in it.left + it.right + (it.out,) if n in self._components]
            univ = np.unique(np.concatenate([self.column(n)[1] for n in used] or [np.zeros(0, np.uint64)])).astype(np.uint64)
            nb = len(univ)
            synthesized_body = {"world_pos": np.tile([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0], (nb, 1)), "world_vel": np.zeros((nb, 6)),
                                "inertia": np.tile([1.0, 1.0, 1.0, 0.0, 0.0, 0.0, 1.0], (nb, 1)), "world_accel": np.zeros((nb, 6)),
                                "force": np.zeros((nb, 6))}
        if synthesized_body is not None:
            pos, ids = synthesized_body["world_pos"], univ
            body = {k: (synthesized_body[k], univ) for k in ("world_vel", "inertia", "world_accel", "force")}
        else:
            pos, ids = self.column("world_pos")
            body = {k: self.column(k) for k in ("world_vel", "inertia", "world_accel", "force")}
        # components may live on different entity sets (a scene object with a world_pos but no Body):
        # six_dof runs on the intersection (query.rs:136-208); each column keeps its own id vector
        column_ids = {"world_pos": ids, **{k: v[1] for k, v in body.items()}}
        effs = []
        extra_columns = None
        self_partial = {}      # components living on fewer entities than the row set: (rows in the set, own rows, n, original)
        if program_stages is not None:
            eff_pipe = system.effectors if isinstance(system.effectors, _dsl.Pipe) else _dsl.Pipe([])
            if not isinstance(system.effectors, _dsl.Pipe) and system.effectors.ops:
                raise TypeError("inside a generated program the six_dof effectors must be dsl effectors")
            widths = {name: int(self.column(name)[0].shape[1]) for name in self._components}
            # a fold returning el.Force (frontend._lower_fold): its result column and the mark of the rows it is folded onto
            hidden = {}
            for it in program_stages[0] + program_stages[1]:
                if isinstance(it, _dsl.GraphFold) and getattr(it, "hidden_force", None):
                    hidden[it.hidden_force[0]], hidden[it.hidden_force[1]] = ("value", it), ("mark", it)
                    widths[it.hidden_force[0]], widths[it.hidden_force[1]] = 6, 1
            # the executor's row set: the Body join (ascending id unless the Body columns already coincide, query.rs:673,702)
            row_ids = ids
            for v in column_ids.values():
                if not np.array_equal(v, row_ids):
                    row_ids = np.intersect1d(row_ids, v)
            # stand-alone folds inside the pipe (graph.rs:239-361): their edges as row pairs of the row set, spawn order; an
            # edge whose endpoint is no row cannot join the fold's queries
            folds = [it for it in program_stages[0] + program_stages[1] if isinstance(it, _dsl.GraphFold)]
            fold_rows = None
            body_rows = None       # rows of the executor that are real Bodies, when it also holds stand-in rows
            if folds:
                # an edge joins a fold when its source carries the left query + the output and its target the right query
                # (query.rs:136-208); the Body columns count as components of the Body entities
                body_set = set(int(e) for e in row_ids)
                members = {n: set(int(x) for x in self.column(n)[1]) for it in folds for n in it.left + it.right + (it.out,)
                           if n in self._components}
                def carries(e, names):
                    return all((int(e) in body_set) if n in ("world_pos", "world_vel", "inertia")
                               else int(e) in members.get(n, ()) for n in names)
                fold_pairs = {}
                for it in folds:
                    frm, to = self.edge_pairs(it.edge_component)
                    out_names = () if it.out in hidden else (it.out,)
                    keep = [k for k in range(len(frm)) if carries(frm[k], it.left + out_names) and carries(to[k], it.right)]
                    fold_pairs[it.edge_component] = ([int(frm[k]) for k in keep], [int(to[k]) for k in keep])
                extra = sorted({e for f_, t_ in fold_pairs.values() for e in f_ + t_} - body_set)
                if extra and synthesized_body is None:
                    # folds between Bodies and plain entities (cube-sat's sensors -> satellite): the plain entities become
                    # rows of the executor too, with stand-in Body values nothing reads (systems that touch the Body are
                    # masked to the real ones through `has:world_pos`); six_dof integrates the stand-ins at rest
                    ext_ids = np.union1d(row_ids, np.array(extra, dtype=np.uint64)).astype(np.uint64)
                    at_real = np.searchsorted(ext_ids, row_ids)
                    stand_in = {"world_pos": [0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0], "world_vel": [0.0] * 6, "inertia": [1.0, 1.0, 1.0, 0.0, 0.0, 0.0, 1.0],
                                "world_accel": [0.0] * 6, "force": [0.0] * 6}
                    def extend(arr, own_ids, fill):
                        out = np.tile(np.array(fill, dtype=np.float64), (len(ext_ids), 1))
                        sel = np.isin(own_ids, row_ids)
                        out[np.searchsorted(ext_ids, own_ids[sel])] = arr[sel]
                        return out
                    pos = extend(pos, ids, stand_in["world_pos"])
                    body = {k: (extend(v[0], v[1], stand_in[k]), ext_ids) for k, v in body.items()}
                    ids = ext_ids
                    column_ids = {"world_pos": ids, **{k: ids for k in body}}
                    body_rows, row_ids = at_real, ext_ids
                where_row = {int(e): k for k, e in enumerate(row_ids)}
                fold_rows = {name: ([where_row[a] for a in f_], [where_row[b] for b in t_]) for name, (f_, t_) in fold_pairs.items()}
            probe = _dsl.Program(program_stages[0], eff_pipe, program_stages[1]).trace(widths, fold_edges=fold_rows)
            partial = [n for n, _ in probe.columns if "#fold" not in n and not n.endswith("#head") and n not in hidden
                       and not np.all(np.isin(row_ids, self.column(n)[1]))]
            if body_rows is not None:
                partial.append("world_pos")
            written = {probe.table.cols[int(t[1:].split("_")[0])][0] for s_ in probe.pre + probe.post for t in s_.written if t[0] == "c"}
            # a component living on exactly ONE entity and only read is a singleton query (`el.Query[el.Seed]`, `s[0]`,
            # system.rs:12-23 entity axis elided): every row sees that one value
            declared = {n for s_ in program_stages[0] + program_stages[1] for n in getattr(s_, "singletons", ())}
            singletons = {n for n in partial if n in declared and len(self.column(n)[1]) == 1 and n not in written}
            partial = [n for n in partial if n not in singletons]
            # The executor's rows are the Body join.  A system whose query names no Body component also runs, in the
            # reference, on matching entities OUTSIDE that join (a Globals-like entity carrying only plain components).
            # Those entities and systems form an independent little world (systems are per-entity maps; the only thing
            # shared across entities, a singleton, is read-only): it gets an executor of its own, stepped in lockstep.
            side_systems, side_entities = [], set()
            if not getattr(system, "no_six_dof", False):
                body_names = set(_dsl._BODY_NAMES)
                for s_, t_ in zip(program_stages[0] + program_stages[1], probe.pre + probe.post):
                    if isinstance(s_, _dsl.GraphFold):
                        continue
                    touched = set(s_.params) | {probe.table.cols[int(t[1:].split("_")[0])][0] for t in t_.written if t[0] == "c"}
                    touched = {n for n in touched if not n.startswith("has:") and not n.endswith("#head")}
                    if touched & body_names or not touched:
                        continue
                    members = None
                    for n in touched - singletons:
                        members = self.column(n)[1] if members is None else np.intersect1d(members, self.column(n)[1])
                    stray = np.setdiff1d(members, row_ids) if members is not None else []
                    if len(stray):
                        if touched & singletons:
                            raise NotImplementedError(f"system {s_.__name__} reads a one-entity component and matches entities that "
                                                      "are not Bodies while six_dof is in the pipe")
                        side_systems.append(s_)
                        side_entities.update(int(e) for e in stray)
            effs = _dsl.Program(program_stages[0], eff_pipe, program_stages[1], substeps=substeps)
            extra_columns = {}
            for name, w_ in effs.trace(widths, partial, fold_edges=fold_rows).columns:
                if name == "has:world_pos":                      # which rows are real Bodies (the others are stand-ins)
                    mask = np.zeros((len(row_ids), 1))
                    mask[body_rows] = 1.0
                    extra_columns[name] = mask
                    column_ids[name] = row_ids
                    continue
                if name in hidden:
                    kind, it = hidden[name]
                    col = np.zeros((len(row_ids), w_))
                    if kind == "mark":
                        col[sorted(set(fold_rows[it.edge_component][0]))] = 1.0
                    extra_columns[name] = col
                    column_ids[name] = row_ids
                    continue
                if name.startswith("has:") or "#fold" in name or name.endswith("#head"):   # presence columns / fold scratch rows /
                    continue                                                                # window heads: made below / by HipExec
                arr, aids = self.column(name)
                if name in singletons:
                    extra_columns[name] = np.tile(arr[0], (len(row_ids), 1))
                    column_ids[name] = row_ids
                    self_partial[name] = (np.zeros(1, dtype=np.int64), np.zeros(1, dtype=np.int64), 1, arr)
                elif name in partial:      # densify onto the row set + presence column (the system's query join mask)
                    where = {int(e): k for k, e in enumerate(row_ids)}
                    sel = np.array([k for k, e in enumerate(aids) if int(e) in where], dtype=np.int64)
                    at = np.array([where[int(aids[k])] for k in sel], dtype=np.int64)
                    dense, has = np.zeros((len(row_ids), w_)), np.zeros((len(row_ids), 1))
                    dense[at], has[at] = arr[sel], 1.0
                    extra_columns[name], extra_columns["has:" + name] = dense, has
                    column_ids[name] = column_ids["has:" + name] = row_ids
                    self_partial[name] = (at, sel, len(aids), arr)
                else:
                    column_ids[name] = aids
                    extra_columns[name] = arr
        elif isinstance(system.effectors, _dsl.Pipe):
            effs = system.effectors
            extra_columns = {}
            for name, _w in system.effectors.trace().columns:
                arr, aids = self.column(name)
                column_ids[name] = aids
                extra_columns[name] = arr
        else:
            for e in system.effectors.ops:
                if not isinstance(e, _dsl.EdgeFold) and e.aux_name is not None:
                    arr, aids = self.column(e.aux_name)
                    column_ids[e.aux_name] = aids
                    e = Effector(e.kind, e.p, e.aux_name, arr)
                effs.append(e)
        same = all(np.array_equal(v, ids) for v in column_ids.values())
        edges = None
        if program_stages is None and not isinstance(system.effectors, _dsl.Pipe) and system.effectors.edge_component:
            edges = self.edge_pairs(system.effectors.edge_component)
            if system.effectors.edge_component == self.TOTAL_EDGE:   # only pairs of Body entities can join the fold's queries
                body_ids = set(int(e) for e in ids)
                for v in column_ids.values():
                    body_ids &= set(int(e) for e in v)
                keep = np.array([int(a) in body_ids and int(b) in body_ids for a, b in zip(*edges)], dtype=bool)
                edges = (edges[0][keep], edges[1][keep])
        if _dry:       # generated_sources(): everything resolved, nothing bound
            return dict(effectors=effs, columns=extra_columns, edges=edges, dt=dt, time_step=system.time_step, row_ids=np.asarray(ids).copy(),
                        names=dict(self.entity_ids_by_name),
                        substeps=substeps if program_stages is not None else 1, body=dict(world_pos=pos, **{k: v[0] for k, v in body.items()}),
                        integrator=L.INTEGRATOR_NONE if getattr(system, "no_six_dof", False) else system.integrator.value)
        hip = HipExec(pos, body["world_vel"][0], body["inertia"][0], world_accel=body["world_accel"][0],
                      force=body["force"][0], entity_ids=ids, simulation_time_step=dt, time_step=system.time_step,
                      integrator=L.INTEGRATOR_NONE if getattr(system, "no_six_dof", False) else system.integrator.value,
                      effectors=effs, edges=edges,
                      ticks_per_launch=ticks_per_telemetry * (substeps if program_stages is not None else 1), device=device,
                      column_entity_ids=None if same else column_ids, columns=extra_columns,
                      reuse_trace=True)      # traced above, in THIS build, with the presence masks and fold rows the executor cannot know
        ex = Exec(hip, self, ticks_per_telemetry, dt)
        ex._substeps = substeps if program_stages is not None else 1
        ex._partial = self_partial
        ex._body_rows = body_rows if program_stages is not None else None
        if program_stages is not None and side_systems:
            sub = World()
            while sub.entity_len <= max(side_entities):      # same entity ids as in this world
                sub._lib.sixdof_world_spawn(sub._w)
            touched = sorted({n for s_ in side_systems for n in s_.params} | {n for n, _ in probe.columns if not n.startswith("has:")})
            for name in touched:
                if name not in self._components:
                    continue
                arr, aids = self.column(name)
                for row, e in zip(arr, aids):
                    if int(e) in side_entities:
                        sub.insert(EntityId(int(e)), C(name, row))
            sub._names = {k: v for k, v in self._names.items() if k in side_entities}
            side = sub.build(_dsl.Stages(side_systems), simulation_rate=simulation_rate, telemetry_rate=telemetry_rate, device=device)
            ex._side = side
        return ex


class _History:
    """The rows `exec.history` reads back in the reference (a DB archive, exec.rs:189-213): every component column after
    each telemetry commit, starting with the spawned state."""

    def __init__(self, ex, world: "World", components: Sequence[str], dt: float):
        self._ex, self._world, self._dt = ex, world, dt
        self._components = list(components)
        self.ticks: List[int] = []
        self.rows: Dict[str, List[np.ndarray]] = {c: [] for c in self._components}
        self.sample()

    def sample(self) -> None:
        self.ticks.append(int(self._ex.tick))
        for c in self._components:
            self.rows[c].append(np.array(self._ex.column_array(c), dtype=np.float64))

    def frame(self, wanted) -> Dict[str, np.ndarray]:
        wanted = [wanted] if isinstance(wanted, str) else list(wanted)
        by_name = {v: k for k, v in self._world._names.items()}
        out = {"time": np.asarray(self.ticks, dtype=np.float64) * self._dt}
        for key in wanted:
            ent, _, comp = key.partition(".")
            if ent not in by_name or comp not in self.rows:
                raise KeyError(key)
            at = np.nonzero(self._ex.column_ids(comp) == by_name[ent])[0]
            if not len(at):
                raise KeyError(key)
            series = np.stack([r[at[0]] for r in self.rows[comp]])
            out[key] = series[:, 0] if series.ndim == 2 and series.shape[1] == 1 else series
        return out


def record_history(ex, world: "World") -> None:
    """Start the telemetry log behind exec.history() on a freshly built executor."""
    names = []
    for c in ("world_pos", "world_vel", "world_accel", "force", "inertia") + tuple(world._components):
        try:
            ex.column_array(c)
        except KeyError:
            continue
        if c not in names:
            names.append(c)
    dt = getattr(ex, "_dt", None) or float(world._lib.sixdof_world_time_step(world._w))
    if not hasattr(ex, "_world"):        # GraphFoldExec: wrap its run
        ex._world, run = world, ex.run
        def logged_run(ticks: int = 1):
            for _ in range(int(ticks)):
                run(1)
                ex._history.sample()
        ex.run = logged_run
        ex.history = lambda components: ex._history.frame(components)
    ex._history = _History(ex, world, names, dt)


class Exec:
    """PyExec (exec.rs:95-240): run(ticks), column access, profile."""

    def __init__(self, hip: HipExec, world: World, ticks_per_telemetry: int, dt: float):
        self._hip, self._world, self._tpt, self._dt = hip, world, ticks_per_telemetry, dt
        self._last = TickTimings()

    def run(self, ticks: int = 1) -> None:
        log = getattr(self, "_history", None)
        if log is None:
            self._advance(ticks)
            return
        done = 0                                   # exec.rs:110-172: batches of ticks_per_telemetry, a commit after each
        while done < ticks:
            step = min(self._tpt, ticks - done)
            self._advance(step)
            log.sample()
            done += step

    def _advance(self, ticks: int) -> None:
        self._last = self._hip.run(ticks * getattr(self, "_substeps", 1))      # k integrator sub-steps per world tick
        side = getattr(self, "_side", None)        # plain-component entities beside the Bodies: same number of ticks
        if side is not None:
            side.run(ticks)

    def history(self, components):
        """exec.history("e1.x") / exec.history(["e1.x", "e2.x"]) (exec.rs:189-213): {"time": seconds, "<entity>.<component>":
        one row per telemetry commit, the spawned state first}.  Needs build(..., history=True) of elodin_amd.frontend."""
        if getattr(self, "_history", None) is None:
            raise RuntimeError("history is not being recorded: build with elodin_amd.frontend.World (history=True)")
        return self._history.frame(components)

    def column_ids(self, name: str) -> np.ndarray:
        """Entity id of each row of column_array(name)."""
        n = len(self._main_column_array(name))
        if name in self._world._components:
            ids = self._world.column(name)[1]
            if len(ids) == n:
                return ids
        return self._hip.entity_ids

    @property
    def tick(self) -> int:
        return self._hip.tick // getattr(self, "_substeps", 1)

    def column_array(self, name: str) -> np.ndarray:
        out = self._main_column_array(name)
        side = getattr(self, "_side", None)
        if side is not None and name in side._world._components:
            ids = self.column_ids(name)
            if len(ids) == len(out):
                out = np.array(out)
                where = {int(e): r for r, e in enumerate(ids)}
                for row, e in zip(side.column_array(name), side.column_ids(name)):
                    out[where[int(e)]] = row
        return out

    def _main_column_array(self, name: str) -> np.ndarray:
        cols = {"world_pos": self._hip.world_pos, "world_vel": self._hip.world_vel, "world_accel": self._hip.world_accel,
                "force": self._hip.force, "inertia": self._hip.inertia}
        if name in cols:
            rows = getattr(self, "_body_rows", None)       # the executor also holds stand-in rows for plain entities
            return cols[name] if rows is None else cols[name][rows]
        if name in getattr(self._hip, "_windows", {}):     # a window component: the reference's order, flattened like its column
            w = self._hip.component(name)
            return w.reshape(w.shape[0], -1)
        if name in getattr(self, "_partial", {}):     # component on fewer entities than the row set: its own rows, own order
            at, sel, n_own, original = self._partial[name]
            out = np.array(original, dtype=np.float64)
            out[sel] = self._hip._aux[name][at]
            return out
        if name in self._hip._aux:
            return self._hip._aux[name]
        raise KeyError(name)

    def component(self, name: str) -> np.ndarray:
        """Any bound component column after the last run (program columns are downloaded with the Body columns)."""
        return self.column_array(name)

    def entity_ids(self) -> np.ndarray:
        return self._hip.entity_ids

    def profile(self) -> Dict[str, float]:
        t = self._hip.last_timings()   # metric names of profile.rs:14-59
        tick_ms = t.kernel_invoke_ms / max(1, t.ticks)
        rtf = (self._dt * 1e3) / tick_ms if tick_ms > 0 else float("inf")   # real_time_factor, profile.rs:55
        return {"kernel_invoke": t.kernel_invoke_ms, "h2d_upload": t.h2d_upload_ms, "d2h_download": t.d2h_download_ms,
                "kernel_device": t.kernel_device_ms, "tick": tick_ms, "real_time_factor": rtf,
                "launches": float(t.launches)}
