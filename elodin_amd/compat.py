"""Run a reference script UNMODIFIED: `import elodin as el`, `import jax`, `from jax import numpy as jnp`, `jax.lax`,
`jax.random`, `jax.scipy.linalg` resolve to this package's front-end.

    import elodin_amd.compat as compat
    compat.install()                 # before the script is imported
    import sim                        # e.g. /root/reference/examples/ball/sim.py, untouched
    exec = sim.world().build(sim.system())

The reference's Python surface (libs/nox-py/python/elodin/__init__.py:160-557: decorators, queries, archetypes, spatial
types) is `elodin_amd.frontend`; what a script additionally needs is `jax.numpy`.  A reference script uses it in two places
with two meanings: at module level and while spawning, on concrete numbers (`jnp.array([0.0, 0.0, 6.0])` is data); inside
`@el.map` / `@el.system` functions, on traced values.  The shim keeps that split: every `jnp.<fn>` call goes to the tracer
(`elodin_amd.dsl.np`) when user code is being traced (dsl.TRACING) or an argument is a traced value, and to plain numpy
otherwise; host arrays captured by traced code (module-level constants such as a filter's F matrix) enter the trace as
constants (dsl._host).  Nothing here computes on the device or restates an algorithm — it is name resolution.

What the shim is not: JAX.  `jax.jit`, `jax.grad`, `jax.vmap` over traced code, device arrays, pytrees are not provided;
a script that uses them fails with AttributeError / NotImplementedError naming the missing piece.  `World.run(...)` (the
reference's blocking entry point that also launches the editor) builds the executor and steps `max_ticks` ticks on the GPU
when `max_ticks` is given; `compat.install(run="record")` makes it only remember its arguments (`world.compat_run`), which is
what tests on a machine without a GPU use to look at the program a script builds.
"""
from __future__ import annotations

import sys
import types

import numpy as _np

from . import dsl as _dsl
from . import dsl_mat as _mat
from . import frontend as _fe

_RUN_MODE = ["execute"]


def _symbolic(x) -> bool:
    if isinstance(x, (_dsl.Expr, _dsl.Vec, _dsl.Quaternion, _dsl.SpatialTransform, _dsl.SpatialMotion, _dsl.SpatialForce,
                      _dsl.SpatialInertia, _dsl.Window)):
        return True
    if isinstance(x, (list, tuple)):
        return any(_symbolic(v) for v in x)
    if isinstance(x, dict):
        return any(_symbolic(v) for v in x.values())
    return False


def _to_trace(x):
    """Host data handed to a tracer function: arrays become constant vectors / matrices, scalars stay."""
    if isinstance(x, _np.ndarray):
        return _dsl._host(x) if x.ndim else float(x)
    if isinstance(x, _np.generic):
        return float(x)
    return x


_MISSING = object()      # "attribute absent" sentinel: None is a legitimate value (newaxis)


class _Dispatch(types.ModuleType):
    """A module whose functions exist twice: `traced` (elodin_amd.dsl.*) and `host` (numpy.*)."""

    def __init__(self, name, traced, host, extra=None):
        super().__init__(name)
        self.__dict__["_traced"], self.__dict__["_host"] = traced, host
        self.__dict__.update(extra or {})

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        traced, host = self.__dict__["_traced"], self.__dict__["_host"]
        if name == "newaxis":
            return None                                  # numpy's / jnp's newaxis IS None
        t = getattr(traced, name, _MISSING) if traced is not None else _MISSING
        h = getattr(host, name, _MISSING) if host is not None else _MISSING
        t = None if t is _MISSING else t
        h = None if h is _MISSING else h
        if t is None and h is None:
            raise AttributeError(f"{self.__name__}.{name} is not provided by elodin_amd.compat")
        if h is not None and not callable(h):
            if isinstance(h, (int, float, type(None))):
                return h                                 # constants: pi, inf, e, newaxis
            raise AttributeError(f"{self.__name__}.{name} is not provided by elodin_amd.compat")   # sub-packages: fft, ma, ...
        if isinstance(h, type) and t is None:
            return h                                     # numpy types used as annotations / constructors

        def call(*args, **kw):
            tracing = _dsl.TRACING[0] > 0 or _symbolic(args) or _symbolic(kw)
            if tracing and t is not None:
                kw.pop("dtype", None) if name not in ("array", "zeros", "ones", "asarray", "arange", "eye", "full") else None
                try:
                    return t(*[_to_trace(a) for a in args], **{k: _to_trace(v) for k, v in kw.items()})
                except TypeError as err:
                    if h is not None and not (_symbolic(args) or _symbolic(kw)):
                        # concrete data only, in a shape the tracer has no value for (a 4-D coefficient grid assembled inside a
                        # decorated function, examples/rocket/main.py:236-252): plain numpy — it is a constant of the program
                        return h(*args, **kw)
                    if "unexpected keyword" in str(err):      # say which keyword of which function, not the tracer's internals
                        raise NotImplementedError(f"{self.__name__}.{name}({', '.join(kw)}=...) on traced values: {err} "
                                                  f"— not provided by elodin_amd.compat") from err
                    raise
            if h is None:
                if t is None:
                    raise AttributeError(f"{self.__name__}.{name}")
                return t(*args, **kw)
            if _symbolic(args) or _symbolic(kw):      # numpy would wrap traced values into an object array
                raise NotImplementedError(f"{self.__name__}.{name} on traced values is not provided by elodin_amd.compat")
            return h(*args, **kw)
        call.__name__ = name
        return call


class _ScalarType:
    """jnp.float64 / jnp.int32 ...: a dtype (`dtype=jnp.int64`, `x.astype(jnp.int32)`) and a constructor (`jnp.int32(x)`), on
    data and on traced values (integer types truncate toward zero; values stay in the executor's float type)."""

    def __init__(self, np_type):
        self.np_type, self.dtype, self.__name__ = np_type, _np.dtype(np_type), np_type.__name__

    def __call__(self, x):
        if _symbolic(x):
            x = _to_trace(x)
            return _dsl.np.int32(x) if self.dtype.kind in "iu" else x
        return self.np_type(x)

    def __eq__(self, other):
        return other is self or other == self.np_type or other == self.dtype
    def __hash__(self): return hash(self.dtype)
    def __repr__(self): return f"jnp.{self.__name__}"


class _HostArray(_np.ndarray):
    """What jnp.array / jnp.asarray give on concrete data: a numpy array that can also be indexed by a TRACED value —
    `T_B[i]` with `T_B` a module-level table and `i` computed inside a decorated function (examples/falcon9/atmosphere.py:64)
    is a gather from a constant table, which the tracer spells as a select chain."""

    def __getitem__(self, idx):
        if _symbolic(idx) or (isinstance(idx, tuple) and any(_symbolic(k) for k in idx)):
            # a long table (examples/monte-carlo/sim.py:84-97: 262,144 rows) is a gather from device memory, a short one a
            # select chain in registers
            if self.ndim <= 2 and self.shape[0] >= _dsl.GATHER_MIN_ROWS and (not isinstance(idx, tuple) or not _symbolic(idx[1:])):
                return _dsl.HostTable(_np.asarray(self))[idx]
            return _dsl._host(_np.asarray(self))[idx]
        return super().__getitem__(idx)


def _host_array(x, dtype=None):
    """jnp.array / jnp.asarray on concrete data under jax_enable_x64 (the reference's setting): Python floats are float64,
    Python ints int64 — numpy's own rules."""
    a = _np.asarray(x)
    a = a.astype(dtype) if dtype is not None else a
    return a.view(_HostArray) if a.ndim else a


class _HostNumpy:
    """numpy with jax.numpy's float64 / functional-update defaults for the handful of constructors scripts call on data."""

    def __getattr__(self, name):
        return getattr(_np, name)

    @staticmethod
    def array(x, dtype=None): return _host_array(x, dtype)
    @staticmethod
    def asarray(x, dtype=None): return _host_array(x, dtype)
    @staticmethod
    def zeros(shape, dtype=None): return _np.zeros(shape, dtype=dtype or _np.float64)
    @staticmethod
    def ones(shape, dtype=None): return _np.ones(shape, dtype=dtype or _np.float64)
    @staticmethod
    def eye(n, m=None, dtype=None): return _np.eye(n, m, dtype=dtype or _np.float64)
    @staticmethod
    def identity(n, dtype=None): return _np.identity(n, dtype=dtype or _np.float64)
    @staticmethod
    def concat(parts, axis=0): return _np.concatenate(parts, axis=axis)


class _TracedScipyLinalg:
    """jax.scipy.linalg calls met in the reference's examples (examples/linalg/sim.py:343-345)."""

    @staticmethod
    def cholesky(a, lower=False): return _mat.cholesky(a, lower=lower)
    @staticmethod
    def solve(a, b, **kw): return _mat.solve(a, b)
    @staticmethod
    def solve_triangular(a, b, trans=0, lower=False, unit_diagonal=False, **kw):
        return _mat.solve_triangular(a, b, lower=lower, trans=trans, unit_diagonal=unit_diagonal)
    @staticmethod
    def inv(a): return _mat.inv(a)
    @staticmethod
    def det(a): return _mat.det(a)


class _TracedScipyNdimage:
    """jax.scipy.ndimage.map_coordinates over a constant grid at traced coordinates (examples/rocket/main.py:9,368)."""

    @staticmethod
    def map_coordinates(input, coordinates, order, mode="constant", cval=0.0):      # noqa: A002  (jax's argument name)
        return _dsl.map_coordinates(input, coordinates, order, mode=mode, cval=cval)


class _TracedScipySpecial:
    """jax.scipy.special calls met in the reference's examples (examples/stablehlo/sim.py:158)."""
    erfc = staticmethod(_dsl.np.erfc)

    @staticmethod
    def erf(x): return 1.0 - _dsl.np.erfc(x)


def _make_jax():
    import scipy.linalg as _sla
    jax = types.ModuleType("jax")
    host = _HostNumpy()
    jnp = _Dispatch("jax.numpy", _dsl.np, host)
    la = _Dispatch("jax.numpy.linalg", _dsl.np.linalg, _np.linalg)
    jnp.__dict__["linalg"] = la
    for tname in ("float64", "float32", "int64", "int32", "uint64", "uint32", "bool_"):
        jnp.__dict__[tname] = _ScalarType(getattr(_np, tname))
    jnp.__dict__["ndarray"] = _np.ndarray
    lax = _Dispatch("jax.lax", _dsl.lax, None)
    rnd = _Dispatch("jax.random", _dsl.random, None, {"PRNGKey": _dsl.random.key})
    jsl = _Dispatch("jax.scipy.linalg", _TracedScipyLinalg, _sla)
    import scipy.special as _ssp
    jsp = _Dispatch("jax.scipy.special", _TracedScipySpecial, _ssp)
    import scipy.ndimage as _snd
    jnd = _Dispatch("jax.scipy.ndimage", _TracedScipyNdimage, _snd)
    jscipy = types.ModuleType("jax.scipy")
    jscipy.__path__ = []
    jscipy.linalg, jscipy.special, jscipy.ndimage = jsl, jsp, jnd
    jax.numpy, jax.lax, jax.random, jax.scipy = jnp, lax, rnd, jscipy
    jax.__path__ = []                                   # a package: `from jax.typing import ArrayLike`
    jax.Array = _np.ndarray
    jtyping = types.ModuleType("jax.typing")
    jtyping.ArrayLike = _np.ndarray
    jax.typing = jtyping
    jax.config = types.SimpleNamespace(update=lambda *a, **k: None)

    def _unsupported(what):
        def f(*a, **k):
            raise NotImplementedError(f"jax.{what} is not provided by elodin_amd.compat (the front-end traces per-entity code; "
                                      "there is no JAX underneath)")
        return f
    jax.jit = lambda f=None, **k: (f if f is not None else (lambda g: g))      # a no-op: everything traced is compiled anyway
    jax.grad, jax.pmap = _unsupported("grad"), _unsupported("pmap")

    def vmap(f, in_axes=0, out_axes=0):
        """jax.vmap over the leading axis of small static arrays inside per-entity code (four landing legs, examples/falcon9/sim.py:794):
        unrolled — f per slice, results stacked.  in_axes: 0 or None per argument."""
        if out_axes != 0:
            raise NotImplementedError("jax.vmap(out_axes != 0) is not provided by elodin_amd.compat")

        def mapped(*args):
            axes = in_axes if isinstance(in_axes, (tuple, list)) else (in_axes,) * len(args)
            if any(a not in (0, None) for a in axes):
                raise NotImplementedError("jax.vmap: in_axes must be 0 or None")
            n = {len(a) for a, ax in zip(args, axes) if ax == 0}
            if len(n) != 1:
                raise ValueError("jax.vmap: mapped arguments differ in length")
            outs = [f(*[(a[i] if ax == 0 else a) for a, ax in zip(args, axes)]) for i in range(n.pop())]
            stack = lambda items: (jnp.stack(items) if _symbolic(items) else _np.stack([_np.asarray(x) for x in items]))
            if isinstance(outs[0], tuple):
                return tuple(stack([o[k] for o in outs]) for k in range(len(outs[0])))
            return stack(outs)
        return mapped
    jax.vmap = vmap
    return {"jax": jax, "jax.numpy": jnp, "jax.numpy.linalg": la, "jax.lax": lax, "jax.random": rnd, "jax.scipy": jscipy,
            "jax.scipy.linalg": jsl, "jax.scipy.special": jsp, "jax.scipy.ndimage": jnd, "jax.typing": jtyping}


class _Inert:
    """Editor / recipe declarations (el.Panel, el.s10.PyRecipe ...): accepted and ignored."""

    def __init__(self, *a, **k): pass
    def __call__(self, *a, **k): return _Inert()
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Inert()


class StepContext:
    """el.StepContext (elodin.pyi:25-170) over an executor's host columns: what pre_step / post_step callbacks read and write.
    The reference's callbacks talk to the database the world commits to; here the committed state IS the executor's host
    columns (refreshed after every batch), and a write lands in them and is uploaded before the next batch runs
    (copy_db_to_world, impeller2_server.rs:607-640)."""

    def __init__(self, ex, world, dt: float, start_timestamp: int = 0, sink=None):
        """sink: an elodin_amd.telemetry.Sink attached to `ex` — reads then come from the pairs' time series (with the
        reference's timestamp semantics) and writes are pushes into them, brought into the world by copy_db_to_world before
        the next batch; without one (a vectorised campaign: thousands of contexts) reads and writes go to the executor's
        host columns directly."""
        self._ex, self._world, self._dt, self._t0 = ex, world, dt, int(start_timestamp or 0)
        self._tick = 0
        self._dirty = False
        self._sink = sink
        self._now = None                                 # the loop sets the batch-end timestamp for post_step
        names = dict(world._names)
        self._entity = {}
        for eid, name in names.items():
            self._entity[name] = eid
            self._entity[_snake(name)] = eid
        self._entity.update(world.entity_ids_by_name)
        self._canon = dict(names)
        self._canon.update({eid: nm for nm, eid in world.entity_ids_by_name.items()})

    @property
    def tick(self) -> int: return int(self._tick)

    @property
    def timestamp(self) -> int:                          # microseconds since the epoch: start + tick * time step
        return self._now if self._now is not None else self._t0 + int(round(self._tick * self._dt * 1e6))

    def _pair(self, pair_name: str) -> str:
        """The name the sink knows the pair under: the entity as spawned (`name=` or `id=`), any accepted spelling."""
        entity, _, comp = pair_name.rpartition(".")
        eid = self._entity.get(entity)
        if eid is None:
            raise RuntimeError(f"component {pair_name!r} does not exist: no entity named {entity!r}")
        canon = self._canon.get(eid, entity)
        return f"{canon}.{comp}"

    def _locate(self, pair_name: str):
        entity, _, comp = pair_name.rpartition(".")
        if not entity or entity not in self._entity:
            raise RuntimeError(f"component {pair_name!r} does not exist: no entity named {entity!r}")
        eid = self._entity[entity]
        cache = self._ex.__dict__.setdefault("_ctx_row_of", {})      # component -> {entity id: row}; entity sets never change
        try:
            col = self._ex.column_array(comp)
            if comp not in cache:
                cache[comp] = {int(e): k for k, e in enumerate(self._ex.column_ids(comp))}
        except KeyError:
            # a component no system reads or writes (examples/monte-carlo's `target`) is not bound to the executor: it keeps the
            # value it was spawned with, which the world still holds
            static = self._ex.__dict__.setdefault("_ctx_static", {})
            if comp not in static:
                try:
                    rows, ids = self._world.column(comp)
                except KeyError:
                    raise RuntimeError(f"component {pair_name!r} does not exist") from None
                static[comp] = rows
                cache[comp] = {int(e): k for k, e in enumerate(ids)}
            col = static[comp]
        row = cache[comp].get(int(eid))
        if row is None:
            raise RuntimeError(f"component {pair_name!r} does not exist: entity {entity!r} does not carry {comp!r}")
        return comp, col, row

    def read_component(self, pair_name: str, timestamp=None):
        if self._sink is not None:
            name = self._pair(pair_name)
            return (self._sink.latest(name) if timestamp is None else self._sink.at(name, int(timestamp)))[1]
        _, col, row = self._locate(pair_name)
        return _np.array(col[row], dtype=_np.float64)

    def write_component(self, pair_name: str, data, timestamp=None) -> None:
        if self._sink is not None:       # a push into the pair's series (TimeTravel if older than its last sample); the world sees
            self._sink.push(self._pair(pair_name), data, self.timestamp if timestamp is None else int(timestamp))     # it at the next copy_db_to_world
            return
        comp, col, row = self._locate(pair_name)
        data = _np.asarray(data, dtype=_np.float64).reshape(-1)
        if data.size != col.shape[1]:
            raise ValueError(f"component {pair_name!r}: {data.size} values for a component of {col.shape[1]}")
        if comp in self._ex.__dict__.get("_ctx_static", {}):      # nothing on the device reads it: the host copy is the component
            col[row] = data
            return
        target = self._ex._main_column_array(comp)       # the executor's own host array (a view the upload reads)
        if target.shape != col.shape or getattr(self._ex, "_body_rows", None) is not None or comp in getattr(self._ex, "_partial", {}):
            raise NotImplementedError(f"StepContext.write_component({pair_name!r}): writes into a joined / partial column are not provided")
        target[row] = data
        self._dirty = True

    def component_batch_operation(self, reads=(), writes=None, **kw):
        out = {name: self.read_component(name) for name in (reads or ())}
        for name, data in (writes or {}).items():
            self.write_component(name, data)
        return out

    def truncate(self):
        if self._sink is None:
            raise NotImplementedError("StepContext.truncate needs the commit sink (World.run attaches one)")
        self._sink.truncate()
    def read_msg(self, *a, **k): raise NotImplementedError("StepContext.read_msg is not provided by elodin_amd.compat (no message log)")
    def stop_recipes(self): return None


def _snake(name: str) -> str:
    out = []
    for ch in str(name).strip():
        out.append("_" if ch in " -" else ch.lower())
    return "".join(out)


def run_stepwise(ex, world, simulation_rate, telemetry_rate, max_ticks, pre_step, post_step, is_canceled=None, start_timestamp=None):
    """The reference's server loop (impeller2_server.rs:553-678) around an executor, with its hand-off (elodin_amd.telemetry.Sink
    = the pairs' time series): per batch of ticks_per_telemetry ticks
        pre_step(first tick of the batch) -> copy_db_to_world (what callbacks wrote reaches the device) -> the batch ->
        commit_world_head at the batch's END timestamp -> post_step(LAST tick of the batch)."""
    from . import telemetry
    from . import frontend as _fe2
    tpt = max(1, int(round(simulation_rate / telemetry_rate))) if telemetry_rate else 1
    dt = 1.0 / float(simulation_rate)
    t0 = int(start_timestamp or 0)
    stamp = lambda k: t0 + int(round(k * dt * 1e6))
    external = tuple(n for n, md in _fe2.COMPONENT_METADATA.items() if str(md.get("external_control", "")).lower() == "true")
    sink = telemetry.Sink.attach(ex, world, t0, external=external)
    ctx = StepContext(ex, world, dt, t0, sink=sink)
    tick = ex.tick
    limit = int(max_ticks) if max_ticks else None
    while limit is None or tick < limit:
        if is_canceled is not None and is_canceled():
            break
        batch = tpt if limit is None else max(1, min(tpt, limit - tick))
        ctx._tick, ctx._now = tick, None
        if pre_step is not None:
            pre_step(tick, ctx)
        sink.copy_to_world()
        ex.run(batch)
        tick = ex.tick
        end_tick = tick - 1                      # the world now reflects the last tick of the batch
        sink.commit(stamp(end_tick))
        ctx._tick, ctx._now = end_tick, stamp(end_tick)
        if post_step is not None:
            post_step(end_tick, ctx)
    sink.copy_to_world()
    ex.compat_sink = sink
    return ctx


def _make_elodin():
    el = types.ModuleType("elodin")
    el.__path__ = []
    for name in dir(_fe):
        if not name.startswith("_"):
            setattr(el, name, getattr(_fe, name))
    import typing
    el.Annotated = typing.Annotated
    el.skew = lambda v: (_mat.skew(_to_trace(v)) if (_dsl.TRACING[0] > 0 or _symbolic(v)) else
                         _np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]]))

    class World(_fe.World):
        """el.World with the reference's entry points that have no meaning without its editor / database accepted:
        `schematic(...)`, `recipe(...)` are ignored, `run(system, ...)` builds and steps (see the module docstring)."""

        def schematic(self, *a, **k): return None
        def recipe(self, *a, **k): return None
        def glb(self, *a, **k): return None
        def sensor_camera(self, *a, **k): return None      # a rendered camera of the editor (examples/sensor-camera)

        def run(self, system, simulation_rate: float = 120.0, generate_real_time: bool = False, telemetry_rate=None,
                default_playback_speed: float = 1.0, max_ticks=None, optimize: bool = False, is_canceled=None, pre_step=None,
                post_step=None, db_path=None, interactive: bool = True, start_timestamp=None, log_level=None,
                backend: str = "cranelift"):
            """World.run with the reference's signature, in the reference's positional order (elodin/__init__.py:673-690).
            Editor / database arguments (generate_real_time, default_playback_speed, db_path, interactive, log_level, optimize,
            backend) have nothing to act on here and are recorded only.  `pre_step(tick, ctx)` / `post_step(end_tick, ctx)` run
            on the server loop's cadence (impeller2_server.rs:553-678): per batch of ticks_per_telemetry ticks — pre_step with the
            batch's first tick, the world's columns refreshed from what the callbacks wrote, the batch, the commit, post_step
            with the batch's LAST tick — through a StepContext over the executor's host columns."""
            self.compat_run = dict(system=system, simulation_rate=simulation_rate, max_ticks=max_ticks, telemetry_rate=telemetry_rate,
                                   pre_step=pre_step, post_step=post_step, is_canceled=is_canceled, start_timestamp=start_timestamp,
                                   ignored=dict(generate_real_time=generate_real_time, default_playback_speed=default_playback_speed,
                                                optimize=optimize, db_path=db_path, interactive=interactive, log_level=log_level,
                                                backend=backend))
            if _RUN_MODE[0] == "record":
                return None
            ex = self.build(system, simulation_rate=simulation_rate, telemetry_rate=telemetry_rate)
            self.compat_exec = ex
            if pre_step is None and post_step is None and is_canceled is None:
                if max_ticks:
                    ex.run(int(max_ticks))
                return ex
            if not max_ticks and is_canceled is None:
                raise ValueError("World.run with step callbacks needs max_ticks or is_canceled here (there is no editor to stop the loop)")
            run_stepwise(ex, self, simulation_rate, telemetry_rate, max_ticks, pre_step, post_step, is_canceled, start_timestamp)
            return ex
    el.World = el.WorldBuilder = World

    def _get_cache_dir():                      # lib.rs:130-142: ProjectDirs("systems", "elodin", "elodin-cli").cache_dir()
        import os
        return os.path.join(os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache"), "elodin-cli")
    el._get_cache_dir = _get_cache_dir
    native = types.ModuleType("elodin.elodin")      # the reference's extension module: same names (`from elodin.elodin import Quaternion`)
    native.__dict__.update({k: v for k, v in el.__dict__.items() if not k.startswith("__")})
    el.elodin = native
    egm = types.ModuleType("elodin.egm08")

    class EGM08:
        """elodin.egm08.EGM08 (python/elodin/egm08.py:17-216) as a declaration: the reference downloads its degree-2190 coefficient
        tables on first use (egm08.py:27-41) and evaluates the spherical-harmonic series with jax scans.  Neither the tables nor
        a network exist here, so a script that constructs the model imports; evaluating the field is refused."""

        def __init__(self, max_degree, cache_directory=""):
            self.max_degree, self.cache_directory = max_degree, cache_directory

        def compute_field(self, x, y, z, mass):
            raise NotImplementedError("elodin.egm08.EGM08.compute_field is not provided by elodin_amd.compat: the EGM2008 coefficient "
                                      "tables (C_normal.npy / S_normal.npy) are fetched from the network by the reference")
    egm.EGM08 = EGM08
    el.egm08 = egm
    for name in ("Panel", "Mesh", "Material", "Shape", "Color", "Glb", "Scene", "Line3d", "BodyAxes", "VectorArrow", "s10",
                 "StepContext", "Time"):
        if not hasattr(el, name):
            setattr(el, name, _Inert())
    return el


_INSTALLED = {}


def install(run: str = "execute", inert=()) -> None:
    """Make `import elodin`, `import jax`, `from jax import numpy, lax, random` resolve to this package.  Refuses to shadow a
    real JAX / elodin that is already imported.  `inert` names modules a script imports for branches that are not taken here
    (examples/drone/main.py:4 imports polars for its --telemetry export): absent ones become empty modules."""
    if run not in ("execute", "record"):
        raise ValueError("run must be 'execute' or 'record'")
    _RUN_MODE[0] = run
    for name in inert:
        if name not in sys.modules:
            try:
                __import__(name)
            except ImportError:
                if name == "polars":         # not an empty stand-in: the pandas-backed subset (compat_polars)
                    from . import compat_polars
                    _INSTALLED[name] = sys.modules[name] = compat_polars.module()
                else:
                    _INSTALLED[name] = sys.modules[name] = types.ModuleType(name)
    if "elodin" in _INSTALLED:
        return
    for name in ("jax", "elodin"):
        if name in sys.modules and getattr(sys.modules[name], "__file__", None):
            raise RuntimeError(f"a real `{name}` is already imported; elodin_amd.compat will not shadow it")
    import typing
    if not hasattr(typing, "Self"):          # the reference targets Python >= 3.11 (examples/drone/config.py:65 `ty.Self`)
        try:
            import typing_extensions
            typing.Self = typing_extensions.Self
        except ImportError:
            pass
    mods = _make_jax()
    if "polars" not in sys.modules:      # host-side table preparation of example scripts: a pandas-backed subset when polars is absent
        try:
            __import__("polars")
        except ImportError:
            from . import compat_polars
            mods["polars"] = compat_polars.module()
    mods["elodin"] = _make_elodin()
    mods["elodin.elodin"], mods["elodin.egm08"] = mods["elodin"].elodin, mods["elodin"].egm08
    sys.modules.update(mods)
    _INSTALLED.update(mods)


def uninstall() -> None:
    for name, mod in list(_INSTALLED.items()):
        if sys.modules.get(name) is mod:
            del sys.modules[name]
    _INSTALLED.clear()


def main(argv=None) -> None:
    """`python -m elodin_amd.compat script.py [args...]`: run a reference sim script on this backend as it is."""
    import argparse
    import os
    import runpy
    ap = argparse.ArgumentParser(prog="python -m elodin_amd.compat",
                                 description="Run an elodin sim script unmodified: `import elodin`, `jax`, `jax.numpy` ... resolve to "
                                             "elodin_amd's front end; world.run / world.build compile for and step on the GPU.")
    ap.add_argument("script")
    ap.add_argument("args", nargs=argparse.REMAINDER)
    ap.add_argument("--record", action="store_true", help="world.run(...) only records its arguments (no GPU needed)")
    ns = ap.parse_args(argv)
    install(run="record" if ns.record else "execute")
    sys.argv = [ns.script, *ns.args]
    sys.path.insert(0, os.path.dirname(os.path.abspath(ns.script)))
    runpy.run_path(ns.script, run_name="__main__")


if __name__ == "__main__":
    main()
