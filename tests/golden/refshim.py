"""Test-only shims that let the REFERENCE's own Python example code run on plain numpy in the build container.

Used by the fixture generators in this directory (make_falcon9_fixtures.py, make_apollo_fixtures.py) to produce golden
vectors from the reference's functions themselves — examples/falcon9/{atmosphere,aero,propulsion,frames,rcs,sim}.py,
examples/apollo-lander/sim.py — which are written against `jax.numpy` and the compiled `elodin` wheel, neither of which
exists here.  Nothing below is product code and nothing under elodin_amd/ imports it; it never travels to
the GPU box as anything but a dormant file (the generators need /root/reference).

* `jax` / `jax.numpy` / `jax.random` / `jax.lax`: numpy with an ndarray subclass that has `.at[i].set/.add`, a loop
  `vmap`, and `jax.random.key / fold_in / normal` as JAX's default generator computes them: threefry2x32 in the
  partitionable layout (jax >= 0.5 default; the reference pins jax 0.10.0, libs/nox-py/pyproject.toml:14), 64 random bits
  per float64 sample, `sqrt(2) * erfinv(uniform(nextafter(-1, 0), 1))`.  JAX itself is not installed; the restatement
  follows jax/_src/prng.py (`threefry_2x32`, `threefry_fold_in`, `_threefry_random_bits_partitionable`) and
  jax/_src/random.py (`_uniform`, `_normal_real`), and is anchored on the one reference-held draw there is: the wind row
  of scripts/ci/baseline/ball-csv (`random.normal(random.key(seed), shape=(3,))`, tests/test_refshim_random.py).
* `elodin`: decorators that hand back the undecorated function in a callable, `|`-pipeable wrapper (`el.map`,
  `el.system`, `el.six_dof` remembering its effectors), a World that remembers what was spawned, a tiny
  `el.monte_carlo` (params context in, result(...) out), inert declarations for everything else,
  and the spatial types `Quaternion / SpatialTransform / SpatialMotion / SpatialForce / SpatialInertia` whose arithmetic
  is NOT restated here but delegated to the pinned C oracle (oracle/sixdof_oracle.c: orc_quat_mul, orc_quat_inverse,
  orc_quat_rotate, orc_quat_from_axis_angle, orc_transform_add_motion — K1-K8 + golden CSVs pin those), mirroring
  libs/nox-py/src/spatial.rs:21-107,121-176,190-262,276-379,392-449.
"""
from __future__ import annotations

import sys as _sys
_sys.dont_write_bytecode = True      # the reference checkout is read-only: no __pycache__ next to what is imported from it

import sys
import types

import numpy as np

from oracle import oracle as orc


# ---- jax.numpy over numpy ------------------------------------------------------------------------------------------

class _At:
    def __init__(self, arr):
        self._a = arr

    def __getitem__(self, idx):
        return _AtIdx(self._a, idx)


class _AtIdx:
    def __init__(self, arr, idx):
        self._a, self._i = arr, idx

    def set(self, v):
        out = np.array(self._a, dtype=np.float64, copy=True).view(JArray)
        out[self._i] = v
        return out

    def add(self, v):
        out = np.array(self._a, dtype=np.float64, copy=True).view(JArray)
        out[self._i] = out[self._i] + v
        return out


class JArray(np.ndarray):
    """ndarray with jax's functional-update surface."""

    @property
    def at(self):
        return _At(self)


def _wrap(x):
    if isinstance(x, np.ndarray) and not isinstance(x, JArray):
        return x.view(JArray)
    if isinstance(x, tuple):
        return tuple(_wrap(v) for v in x)
    return x


def _unwrap_args(a):
    return a


class _JnpModule(types.ModuleType):
    """`jax.numpy`: every attribute is numpy's, results re-viewed as JArray; f64 everywhere (jax_enable_x64)."""

    def __init__(self):
        super().__init__("jax.numpy")
        self.ndarray = JArray
        self.float64, self.float32, self.int32, self.int64 = np.float64, np.float32, np.int32, np.int64
        self.pi, self.inf, self.nan, self.newaxis = np.pi, np.inf, np.nan, np.newaxis
        self.linalg = _Wrapped(np.linalg)

    def __getattr__(self, name):
        fn = getattr(np, name)
        if not callable(fn) or isinstance(fn, type):
            return fn

        def call(*a, **k):
            return _wrap(fn(*a, **k))
        call.__name__ = name
        return call

    @staticmethod
    def _dtype(x, dtype):
        # data that already IS an integer numpy array keeps its type, like jax (`jnp.asarray(np.linspace(..., dtype=np.int32))`,
        # examples/monte-carlo/sim.py:65: row indices); everything else is float64 (jax_enable_x64, Python lists of numbers)
        if dtype is None and isinstance(x, (np.ndarray, np.generic)) and np.asarray(x).dtype.kind in "iub":
            return np.asarray(x).dtype
        return np.float64 if dtype is None else dtype

    def array(self, x, dtype=None):
        return np.array(x, dtype=self._dtype(x, dtype)).view(JArray)

    def asarray(self, x, dtype=None):
        return np.asarray(x, dtype=self._dtype(x, dtype)).view(JArray)

    def zeros(self, shape, dtype=None):
        return np.zeros(shape, dtype=np.float64 if dtype is None else dtype).view(JArray)

    def ones(self, shape, dtype=None):
        return np.ones(shape, dtype=np.float64 if dtype is None else dtype).view(JArray)

    def full(self, shape, v, dtype=None):
        return np.full(shape, v, dtype=np.float64 if dtype is None else dtype).view(JArray)

    def arange(self, *a, **k):
        return np.arange(*a, **k).astype(np.float64).view(JArray)


class _Wrapped:
    def __init__(self, mod):
        self._m = mod

    def __getattr__(self, name):
        fn = getattr(self._m, name)

        def call(*a, **k):
            return _wrap(fn(*a, **k))
        return call


def _vmap(fn):
    """jax.vmap over axis 0 of every argument, stacking every output."""
    def mapped(*args):
        outs = [fn(*[a[i] for a in args]) for i in range(len(args[0]))]
        if isinstance(outs[0], tuple):
            return tuple(np.stack([np.asarray(o[k]) for o in outs]).view(JArray) for k in range(len(outs[0])))
        return np.stack([np.asarray(o) for o in outs]).view(JArray)
    return mapped


# ---- jax.random (threefry2x32, partitionable layout) -------------------------------------------------------------------

def _threefry2x32(k0, k1, x0, x1):
    """jax/_src/prng.py `_threefry2x32_lowering`: 20 rounds in five groups of four, key schedule injected after each."""
    u32 = np.uint32
    k0, k1 = u32(k0), u32(k1)
    x0, x1 = np.asarray(x0, dtype=u32).copy(), np.asarray(x1, dtype=u32).copy()
    ks = (k0, k1, k0 ^ k1 ^ u32(0x1BD11BDA))
    rotations = ((13, 15, 26, 6), (17, 29, 16, 24))
    with np.errstate(over="ignore"):
        x0 += ks[0]
        x1 += ks[1]
        for g in range(5):
            for r in rotations[g % 2]:
                x0 += x1
                x1 = (x1 << u32(r)) | (x1 >> u32(32 - r))
                x1 ^= x0
            x0 += ks[(g + 1) % 3]
            x1 += ks[(g + 2) % 3] + u32(g + 1)
    return x0, x1


def _rng_key(seed):
    """`random.key(seed)` / threefry_seed: the 64-bit seed split into (high, low) uint32 words."""
    seed = int(seed)
    return (np.uint32((seed >> 32) & 0xFFFFFFFF), np.uint32(seed & 0xFFFFFFFF))


def _rng_fold_in(key, data):
    """threefry_fold_in: threefry_2x32(key, threefry_seed(uint32(data))) = block (0, data) under `key`."""
    d = int(np.asarray(data).astype(np.int64)) & 0xFFFFFFFF
    x0, x1 = _threefry2x32(key[0], key[1], np.array([0], dtype=np.uint32), np.array([d], dtype=np.uint32))
    return (np.uint32(x0[0]), np.uint32(x1[0]))


def _rng_normal(key, shape=(), dtype=np.float64):
    """_normal_real for float64: counters (hi = 0, lo = i) -> 64 bits each -> mantissa = bits >> 12 -> [1, 2) - 1 ->
    uniform on [nextafter(-1, 0), 1) -> sqrt(2) * erfinv."""
    from scipy.special import erfinv
    n = int(np.prod(shape)) if shape else 1
    hi, lo = _threefry2x32(key[0], key[1], np.zeros(n, dtype=np.uint32), np.arange(n, dtype=np.uint32))
    bits = (hi.astype(np.uint64) << np.uint64(32)) | lo.astype(np.uint64)
    floats = ((bits >> np.uint64(12)) | np.uint64(0x3FF0000000000000)).view(np.float64) - 1.0
    lo_ = np.nextafter(-1.0, 0.0)
    u = np.maximum(lo_, floats * (1.0 - lo_) + lo_)
    z = np.sqrt(2.0) * erfinv(u)
    return (z.reshape(shape) if shape else np.float64(z[0])).view(JArray) if shape else np.float64(z[0])


def _make_jax():
    jax = types.ModuleType("jax")
    jnp = _JnpModule()
    jax.numpy = jnp
    jax.Array = np.ndarray
    jax.vmap = _vmap
    jax.config = types.SimpleNamespace(update=lambda *a, **k: None)
    rnd = types.ModuleType("jax.random")
    rnd.key = _rng_key
    rnd.PRNGKey = _rng_key
    rnd.fold_in = _rng_fold_in
    rnd.normal = _rng_normal
    jax.random = rnd
    lax = types.ModuleType("jax.lax")
    lax.cond = lambda pred, t, f, *ops: (t(*ops) if bool(pred) else f(*ops))
    lax.select = lambda p, a, b: _wrap(np.where(p, a, b))
    jax.lax = lax
    return jax, jnp, rnd, lax


# ---- elodin over the pinned oracle ----------------------------------------------------------------------------------------

def _v(x, n=None):
    a = np.asarray(x, dtype=np.float64).reshape(-1)
    assert n is None or a.size == n, (a.shape, n)
    return a


class Quaternion:                                        # libs/nox-py/src/spatial.rs:276-379, scalar-last [i,j,k,w]
    def __init__(self, arr):
        self._q = _v(arr, 4).copy()

    @staticmethod
    def identity():
        return Quaternion([0.0, 0.0, 0.0, 1.0])

    @staticmethod
    def from_array(arr):
        return Quaternion(arr)

    @staticmethod
    def from_axis_angle(axis, angle):                     # quaternion.rs:157-169
        return Quaternion(orc.quat_from_axis_angle(_v(axis, 3), float(angle)))

    def vector(self):
        return self._q.copy().view(JArray)

    def inverse(self):                                    # quaternion.rs:141-155: conj / |q|^2
        return Quaternion(orc.quat_inverse(self._q))

    def normalize(self):
        return Quaternion(orc.quat_normalize(self._q))

    def integrate_body(self, delta):                      # quaternion.rs:176-182
        return Quaternion(orc.quat_integrate_body(self._q, _v(delta, 3)))

    def __mul__(self, other):                             # Hamilton product, quaternion.rs:268-281
        if isinstance(other, Quaternion):
            return Quaternion(orc.quat_mul(self._q, other._q))
        return NotImplemented

    def __matmul__(self, other):                          # q (x) (v,0) (x) q^-1, quaternion.rs:283-305; spatial.rs:571-593
        if isinstance(other, SpatialMotion):
            return SpatialMotion(angular=self @ other.angular(), linear=self @ other.linear())
        if isinstance(other, SpatialForce):
            return SpatialForce(torque=self @ other.torque(), linear=self @ other.force())
        return orc.quat_rotate(self._q, _v(other, 3)).view(JArray)


class SpatialTransform:                                   # spatial.rs:21-107: [q(4), p(3)]
    def __init__(self, arr=None, angular=None, linear=None):
        if arr is not None:
            a = _v(arr, 7)
            self._q, self._p = Quaternion(a[:4]), a[4:].copy()
        else:
            self._q = Quaternion.identity() if angular is None else (angular if isinstance(angular, Quaternion) else Quaternion(angular))
            self._p = np.zeros(3) if linear is None else _v(linear, 3).copy()

    def angular(self):
        return self._q

    def linear(self):
        return self._p.copy().view(JArray)

    def asarray(self):
        return np.concatenate([self._q._q, self._p])

    def __add__(self, m):                                 # spatial.rs:530-549
        assert isinstance(m, SpatialMotion)
        return SpatialTransform(orc.transform_add_motion(self.asarray(), m.asarray()))


class SpatialMotion:                                      # spatial.rs:121-176: [omega(3), v(3)]
    def __init__(self, angular=None, linear=None):
        self._w = np.zeros(3) if angular is None else _v(angular, 3).copy()
        self._l = np.zeros(3) if linear is None else _v(linear, 3).copy()

    def angular(self):
        return self._w.copy().view(JArray)

    def linear(self):
        return self._l.copy().view(JArray)

    def asarray(self):
        return np.concatenate([self._w, self._l])

    def __add__(self, o):
        return SpatialMotion(self._w + o._w, self._l + o._l)


class SpatialForce:                                       # spatial.rs:190-262: [tau(3), f(3)]
    def __init__(self, arr=None, torque=None, linear=None):
        if arr is not None:
            a = _v(arr, 6)
            torque, linear = a[:3], a[3:]
        self._t = np.zeros(3) if torque is None else _v(torque, 3).copy()
        self._f = np.zeros(3) if linear is None else _v(linear, 3).copy()

    def torque(self):
        return self._t.copy().view(JArray)

    def force(self):
        return self._f.copy().view(JArray)

    linear = force

    def asarray(self):
        return np.concatenate([self._t, self._f])

    def __add__(self, o):
        return SpatialForce(torque=self._t + o._t, linear=self._f + o._f)


class SpatialInertia:                                     # spatial.rs:392-449: inertia defaults to ones(3) * mass
    def __init__(self, mass, inertia=None):
        self._m = float(np.asarray(mass).reshape(-1)[0])
        self._i = np.ones(3) * self._m if inertia is None else _v(inertia, 3).copy()

    def mass(self):
        return self._m

    def inertia_diag(self):
        return self._i.copy().view(JArray)

    def asarray(self):
        return np.concatenate([self._i, np.zeros(3), [self._m]])


class _Query:
    """el.Query stand-in for @el.system bodies: `q[0]` and `q.map(out_types, fn)` over ONE entity's values."""

    def __init__(self, *values):
        self._v = values

    def __class_getitem__(cls, item):
        return cls

    def __getitem__(self, i):
        return self._v[i]

    def map(self, out_types, fn):
        return fn(*self._v)


class _Inert:
    """Anything declarative (ComponentType, PrimitiveType.F64, Integrator.SemiImplicit, Archetype, s10 recipes ...)."""

    def __init__(self, *a, **k):
        self.args, self.kw = a, k

    def __call__(self, *a, **k):
        return _Inert(*a, **k)

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Inert(name)

    def __getitem__(self, item):
        return self

    def __or__(self, other):
        return self

    __ror__ = __or__

    def __mro_entries__(self, bases):
        return (object,)


class Component:
    """el.Component(name, type, metadata=...): only the name matters here."""

    def __init__(self, name, ty=None, metadata=None, **k):
        self.name, self.metadata = name, metadata or {}


def component_name(annot):
    """`ty.Annotated[jax.Array, el.Component("thrust", ...)]` -> "thrust"."""
    for m in getattr(annot, "__metadata__", ()):
        if isinstance(m, Component):
            return m.name
    raise KeyError(f"no el.Component in {annot!r}")


class _Sys:
    """What @el.map / @el.system hand back: the undecorated function, callable, and pipeable with `|`."""

    def __init__(self, fn):
        self.fn, self.__name__ = fn, getattr(fn, "__name__", "system")

    def __call__(self, *a, **k):
        return self.fn(*a, **k)

    def __or__(self, other):
        return _Pipe(_flatten(self) + _flatten(other))

    def __ror__(self, other):
        return _Pipe(_flatten(other) + _flatten(self))


class _SixDof(_Sys):
    """el.six_dof(time_step=None, sys=None, integrator=...) (six_dof.rs:161-203): remembered, run by the generator."""

    def __init__(self, time_step=None, sys=None, integrator=None):
        self.time_step, self.integrator, self.__name__ = time_step, integrator, "six_dof"
        self.effectors = _flatten(sys) if sys is not None else []
        self.fn = None


class _Pipe(_Sys):
    def __init__(self, systems):
        self.systems, self.fn, self.__name__ = list(systems), None, "pipe"

    def names(self):
        return [s.__name__ for s in self.systems]

    def __getitem__(self, name):
        return next(s for s in self.systems if s.__name__ == name)


def _flatten(x):
    if x is None or isinstance(x, _Inert):
        return []
    if isinstance(x, _Pipe):
        return list(x.systems)
    return [x]


class _Body:
    def __init__(self, world_pos=None, world_vel=None, inertia=None, **k):
        self.world_pos = world_pos if world_pos is not None else SpatialTransform()
        self.world_vel = world_vel if world_vel is not None else SpatialMotion()
        self.inertia = inertia if inertia is not None else SpatialInertia(1.0)


class World:
    """el.World: remembers what was spawned (component name -> initial value per entity), ignores the rest."""

    def __init__(self, *a, **k):
        self.entities = {}

    def spawn(self, components, name=None, **k):
        comps = {}
        for c in (components if isinstance(components, (list, tuple)) else [components]):
            if isinstance(c, _Body):
                comps.update(world_pos=c.world_pos, world_vel=c.world_vel, inertia=c.inertia)
            elif isinstance(c, tuple) and len(c) == 2 and isinstance(c[0], str):
                comps[c[0]] = c[1]
        self.entities[name or f"entity{len(self.entities)}"] = comps
        return len(self.entities)

    def __getattr__(self, name):      # insert / recipe / schematic / run / ...: nothing to do without a runtime
        if name.startswith("__"):
            raise AttributeError(name)
        return lambda *a, **k: None


def _C(annot, value):
    """el.C(ComponentType, value) -> (component name, value)."""
    try:
        return (component_name(annot), value)
    except KeyError:
        return ("?", value)


class _McParams(dict):
    db_path = None


def _make_monte_carlo():
    mc = types.ModuleType("elodin.monte_carlo")
    mc.CONTEXT = {}       # the current rollout's parameter overrides (set by the generator before importing main.py)
    mc.RESULTS = []       # what the sim handed to el.monte_carlo.result(...)

    class Param:
        def __init__(self, ty=float, default=None, min=None, max=None, **k):
            self.ty, self.default, self.min, self.max = ty, default, min, max

    def params_spec(**spec):
        return dict(spec)

    def params(spec=None):
        out = _McParams({k: p.default for k, p in (spec or {}).items()})
        out.update(mc.CONTEXT)          # defaults overlaid by the run's context, like the reference
        return out

    mc.Param, mc.Params, mc.params_spec, mc.params = Param, _McParams, params_spec, params
    mc.port = lambda name, default=0: default
    mc.result = lambda **kw: mc.RESULTS.append(dict(kw))
    return mc


def _make_elodin(jnp):
    el = types.ModuleType("elodin")
    el.__path__ = []   # so `import elodin.x` style probes fail cleanly
    wrap = lambda fn=None, **k: (_Sys(fn) if fn is not None else (lambda f: _Sys(f)))
    el.map = wrap
    el.map_seq = wrap
    el.system = wrap
    import dataclasses
    el.dataclass = dataclasses.dataclass
    el.Query = _Query
    el.GraphQuery = _Query
    for name in ("Quaternion", "SpatialTransform", "SpatialMotion", "SpatialForce", "SpatialInertia"):
        setattr(el, name, globals()[name])
    # well-known component aliases: used as annotations, and (rarely) as constructors of the underlying spatial type
    el.WorldPos = SpatialTransform
    el.WorldVel = SpatialMotion
    el.WorldAccel = SpatialMotion
    el.Force = SpatialForce
    el.Inertia = SpatialInertia
    el.Component, el.World, el.Body, el.C, el.six_dof = Component, World, _Body, _C, _SixDof
    for name in ("ComponentType", "PrimitiveType", "Integrator", "System", "Archetype", "StepContext",
                 "SimulationTick", "SimulationTimeStep", "Seed", "Edge", "Time", "Panel", "Mesh", "Material",
                 "Shape", "Color", "Glb", "Scene", "Line3d", "BodyAxes", "VectorArrow", "s10"):
        setattr(el, name, _Inert(name))
    el.monte_carlo = _make_monte_carlo()
    el.linear = lambda v: SpatialMotion(linear=v)          # legacy helpers some examples use in spawn code
    el.angular = lambda v: SpatialMotion(angular=v)
    return el


def install(example_dir: str):
    """Inject the shims and put the reference example on sys.path.  Returns (jax, jnp, el)."""
    jax, jnp, rnd, lax = _make_jax()
    sys.modules["jax"] = jax
    sys.modules["jax.numpy"] = jnp
    sys.modules["jax.random"] = rnd
    sys.modules["jax.lax"] = lax
    el = _make_elodin(jnp)
    sys.modules["elodin"] = el
    sys.modules["elodin.monte_carlo"] = el.monte_carlo
    if example_dir not in sys.path:
        sys.path.insert(0, example_dir)
    return jax, jnp, el
