"""Effector front-end: a small jax.numpy-shaped tracer for per-entity effectors.

The reference lets `six_dof(sys=...)` take arbitrary JAX functions over components
(`@el.map def gravity(f: el.Force, inertia: el.Inertia) -> el.Force`, examples/ball/sim.py:57-59) and
JIT-compiles the traced graph (libs/nox-py/src/system.rs:784-900, cranelift_compile.rs:13-162).  This module
is the MI355X counterpart for the per-entity case: effector functions written against `dsl.np` (a subset
of jax.numpy) and the spatial types below are traced into a scalar expression DAG, which
elodin_amd/codegen.py turns into the effector stage of the fused HIP step kernel.

    from elodin_amd import dsl
    np = dsl.np

    @dsl.effector
    def lunar_gravity(force, inertia, vel):                       # apollo-lander/sim.py:380-389
        v_h_sq = np.sum(vel.linear()[:2] ** 2)
        g_eff = np.maximum(1.622 - v_h_sq / 1_737_400.0, 0.0)
        return force + dsl.SpatialForce(linear=np.array([0.0, 0.0, -1.0]) * g_eff * inertia.mass())

    sys = six_dof(sys=dsl.pipe(lunar_gravity, apply_thrust))      # effectors compose like `a | b`

Arguments are bound BY NAME: `force`, `pos`/`world_pos`, `vel`/`world_vel`, `inertia`; any other name is a
per-entity component column of width 1..3 spawned on the Body entities (el.C(name, value)).
Python numbers are constants baked into the generated code, like the reference's JIT bakes closure
constants.  Tracing is eager and shape-static: vectors are tuples of scalar nodes.
"""
from __future__ import annotations

import inspect
import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple, Union

Number = Union[int, float]


class Expr:
    """One scalar node of the DAG."""
    __slots__ = ("op", "args", "value", "name")
    _interned: Dict[tuple, "Expr"] = {}

    def __new__(cls, op: str, args: tuple = (), value=None, name: Optional[str] = None):
        key = (op, tuple(id(a) for a in args), value, name)
        hit = cls._interned.get(key)
        if hit is not None:
            return hit
        self = object.__new__(cls)
        self.op, self.args, self.value, self.name = op, args, value, name
        cls._interned[key] = self
        return self

    # ---- arithmetic -------------------------------------------------------------------------------------------
    def __add__(self, o): return NotImplemented if isinstance(o, Vec) else _bin("add", self, o)
    def __radd__(self, o): return _bin("add", o, self)
    def __sub__(self, o): return NotImplemented if isinstance(o, Vec) else _bin("sub", self, o)
    def __rsub__(self, o): return _bin("sub", o, self)
    def __mul__(self, o): return NotImplemented if isinstance(o, Vec) else _bin("mul", self, o)
    def __rmul__(self, o): return _bin("mul", o, self)
    def __truediv__(self, o): return NotImplemented if isinstance(o, Vec) else _bin("div", self, o)
    def __rtruediv__(self, o): return _bin("div", o, self)
    def __neg__(self): return Expr("neg", (self,))
    def __pow__(self, k):
        if isinstance(k, int) and k >= 1:      # jnp integer_pow: repeated multiply
            r = self
            for _ in range(k - 1):
                r = r * self
            return r
        raise TypeError("only small positive integer powers are supported")
    def __lt__(self, o): return _bin("lt", self, o)
    def __le__(self, o): return _bin("le", self, o)
    def __gt__(self, o): return _bin("lt", o, self)
    def __ge__(self, o): return _bin("le", o, self)
    def __bool__(self):
        raise TypeError("traced values have no truth value; use dsl.np.where / logical_and")
    __hash__ = object.__hash__

    def is_const(self, v=None) -> bool:
        return self.op == "const" and (v is None or self.value == v)


def const(v: Number) -> Expr:
    return Expr("const", (), float(v))


def _lift(x) -> Expr:
    if isinstance(x, Expr):
        return x
    if isinstance(x, (int, float)):
        return const(x)
    raise TypeError(f"cannot use {type(x).__name__} in a traced effector")


def _bin(op: str, a, b) -> Expr:
    a, b = _lift(a), _lift(b)
    # constant folding (Python float arithmetic = IEEE double, what a JIT would bake) and the identities
    # x+0, x-0, 0+x, x*1, 1*x, x/1 (exact up to the sign of zero)
    if a.op == "const" and b.op == "const" and op in ("add", "sub", "mul", "div"):
        x, y = a.value, b.value
        if not (op == "div" and y == 0.0):
            return const({"add": x + y, "sub": x - y, "mul": x * y, "div": x / y if y else 0.0}[op])
    if op == "add" and b.is_const(0.0): return a
    if op == "add" and a.is_const(0.0): return b
    if op == "sub" and b.is_const(0.0): return a
    if op == "mul" and b.is_const(1.0): return a
    if op == "mul" and a.is_const(1.0): return b
    if op == "div" and b.is_const(1.0): return a
    return Expr(op, (a, b))


class Vec:
    """Fixed-length vector of scalar nodes (jnp 1-D array of static shape)."""

    def __init__(self, elems: Sequence):
        self.e: Tuple[Expr, ...] = tuple(_lift(x) for x in elems)

    def __len__(self): return len(self.e)
    def __iter__(self): return iter(self.e)
    def __getitem__(self, i):
        r = self.e[i]
        return Vec(r) if isinstance(i, slice) else r
    def _zip(self, o, f):
        if isinstance(o, Vec):
            if len(o) != len(self):
                raise ValueError("shape mismatch")
            return Vec([f(a, b) for a, b in zip(self.e, o.e)])
        return Vec([f(a, o) for a in self.e])
    def __add__(self, o): return self._zip(o, lambda a, b: a + b)
    def __radd__(self, o): return self._zip(o, lambda a, b: b + a)
    def __sub__(self, o): return self._zip(o, lambda a, b: a - b)
    def __rsub__(self, o): return self._zip(o, lambda a, b: b - a)
    def __mul__(self, o): return self._zip(o, lambda a, b: a * b)
    def __rmul__(self, o): return self._zip(o, lambda a, b: b * a)
    def __truediv__(self, o): return self._zip(o, lambda a, b: a / b)
    def __neg__(self): return Vec([-a for a in self.e])
    def __pow__(self, k): return Vec([a ** k for a in self.e])


def _unary(op):
    def f(x):
        if isinstance(x, Vec):
            return Vec([Expr(op, (a,)) for a in x.e])
        return Expr(op, (_lift(x),))
    return f


class _Np:
    """The jax.numpy subset effectors in the reference's examples use."""
    pi = math.pi
    sqrt, abs, sin, cos, tan, exp, log, arccos, arcsin = (_unary(k) for k in
                                                          ("sqrt", "abs", "sin", "cos", "tan", "exp", "log", "acos", "asin"))

    @staticmethod
    def array(x, dtype=None): return Vec(list(x))
    @staticmethod
    def zeros(n, dtype=None): return Vec([0.0] * int(n))
    @staticmethod
    def sum(v: Vec):
        acc = v.e[0]
        for a in v.e[1:]:
            acc = acc + a
        return acc
    @staticmethod
    def dot(a: Vec, b: Vec): return _Np.sum(a * b)
    @staticmethod
    def cross(a: Vec, b: Vec):
        return Vec([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]])
    @staticmethod
    def maximum(a, b): return _zipv(a, b, lambda x, y: Expr("max", (_lift(x), _lift(y))))
    @staticmethod
    def minimum(a, b): return _zipv(a, b, lambda x, y: Expr("min", (_lift(x), _lift(y))))
    @staticmethod
    def clip(x, lo, hi): return _Np.minimum(_Np.maximum(x, lo), hi)
    @staticmethod
    def where(c, a, b):
        if isinstance(a, Vec) or isinstance(b, Vec):
            n = len(a) if isinstance(a, Vec) else len(b)
            av = a if isinstance(a, Vec) else Vec([a] * n)
            bv = b if isinstance(b, Vec) else Vec([b] * n)
            cv = c.e if isinstance(c, Vec) else [c] * n
            return Vec([Expr("select", (_lift(k), x, y)) for k, x, y in zip(cv, av.e, bv.e)])
        return Expr("select", (_lift(c), _lift(a), _lift(b)))
    @staticmethod
    def logical_and(a, b): return Expr("and", (_lift(a), _lift(b)))
    @staticmethod
    def logical_or(a, b): return Expr("or", (_lift(a), _lift(b)))
    @staticmethod
    def logical_not(a): return Expr("not", (_lift(a),))
    @staticmethod
    def arctan2(y, x): return Expr("atan2", (_lift(y), _lift(x)))
    @staticmethod
    def hypot(x, y): return Expr("hypot", (_lift(x), _lift(y)))
    @staticmethod
    def deg2rad(x): return x * (math.pi / 180.0)
    @staticmethod
    def rad2deg(x): return x * (180.0 / math.pi)

    class linalg:
        @staticmethod
        def norm(v: Vec): return _Np.sqrt(_Np.sum(v * v))   # jnp.linalg.norm, ord=None


def _zipv(a, b, f):
    if isinstance(a, Vec) or isinstance(b, Vec):
        n = len(a) if isinstance(a, Vec) else len(b)
        av = a.e if isinstance(a, Vec) else [a] * n
        bv = b.e if isinstance(b, Vec) else [b] * n
        return Vec([f(x, y) for x, y in zip(av, bv)])
    return f(a, b)


np = _Np


# ---- spatial types (thin mirrors of libs/nox-py/src/spatial.rs wrappers) ---------------------------------------

class Quaternion:
    """Scalar-last [x,y,z,w]; assumed unit (the stage attitude is renormalised every stage)."""

    def __init__(self, v: Vec):
        self.v = v

    def vector(self) -> Vec: return self.v
    def inverse(self) -> "Quaternion":
        return Quaternion(Vec([-self.v[0], -self.v[1], -self.v[2], self.v[3]]))
    def __matmul__(self, x: Vec) -> Vec:
        """q @ v: rotate a 3-vector (quaternion.rs:283-305), as v + w t + u x t with t = 2 u x v."""
        u = Vec(self.v.e[:3])
        t = np.cross(u, x) * 2.0
        return x + t * self.v[3] + np.cross(u, t)
    def __mul__(self, o: "Quaternion") -> "Quaternion":
        l, r = self.v, o.v
        return Quaternion(Vec([l[3] * r[0] + l[0] * r[3] + l[1] * r[2] - l[2] * r[1],
                               l[3] * r[1] - l[0] * r[2] + l[1] * r[3] + l[2] * r[0],
                               l[3] * r[2] + l[0] * r[1] - l[1] * r[0] + l[2] * r[3],
                               l[3] * r[3] - l[0] * r[0] - l[1] * r[1] - l[2] * r[2]]))


class SpatialTransform:
    def __init__(self, q: Quaternion, p: Vec): self._q, self._p = q, p
    def angular(self): return self._q
    def linear(self): return self._p


class SpatialMotion:
    def __init__(self, ang: Vec, lin: Vec): self._a, self._l = ang, lin
    def angular(self): return self._a
    def linear(self): return self._l


class SpatialInertia:
    def __init__(self, diag: Vec, mass: Expr): self._d, self._m = diag, mass
    def mass(self): return self._m
    def inertia_diag(self): return self._d


class SpatialForce:
    def __init__(self, torque: Optional[Vec] = None, linear: Optional[Vec] = None):
        self._t = torque if torque is not None else Vec([0.0, 0.0, 0.0])
        self._f = linear if linear is not None else Vec([0.0, 0.0, 0.0])
    def torque(self): return self._t
    def force(self): return self._f
    def __add__(self, o: "SpatialForce"): return SpatialForce(self._t + o._t, self._f + o._f)


# ---- tracing -----------------------------------------------------------------------------------------------------

_FIXED = {"force", "pos", "world_pos", "vel", "world_vel", "inertia"}


class Effector:
    def __init__(self, fn: Callable, widths: Optional[Dict[str, int]] = None):
        self.fn = fn
        self.params = list(inspect.signature(fn).parameters)
        self.widths = dict(widths or {})
        self.__name__ = getattr(fn, "__name__", "effector")

    def __or__(self, other): return pipe(self, other)


def effector(fn=None, **widths):
    """Decorator. Keyword arguments give the row width of component columns the function reads
    (`@dsl.effector(thrust=1, rcs_torque=3)`); width defaults to 3."""
    if fn is None:
        return lambda f: Effector(f, widths)
    return Effector(fn, widths)


def leaf(name: str) -> Expr:
    return Expr("leaf", (), None, name)


class TracedPipe:
    """Result of tracing: output wrench nodes, the component columns read, and dependency flags."""

    def __init__(self, effectors: Sequence[Effector]):
        self.effectors = list(effectors)
        self.columns: List[Tuple[str, int]] = []      # (component name, width) in first-use order
        q = Quaternion(Vec([leaf(f"q{c}") for c in "ijkw"]))
        pos = SpatialTransform(q, Vec([leaf(f"p{c}") for c in "xyz"]))
        vel = SpatialMotion(Vec([leaf(f"w{c}") for c in "xyz"]), Vec([leaf(f"v{c}") for c in "xyz"]))
        inertia = SpatialInertia(Vec([leaf(f"I{c}") for c in "xyz"]), leaf("mass"))
        force = SpatialForce()                          # clear_forces: the pipe starts from zero (six_dof.rs:148-150)
        for eff in self.effectors:
            kwargs = {}
            for name in eff.params:
                if name == "force":
                    kwargs[name] = force
                elif name in ("pos", "world_pos"):
                    kwargs[name] = pos
                elif name in ("vel", "world_vel"):
                    kwargs[name] = vel
                elif name == "inertia":
                    kwargs[name] = inertia
                else:
                    w = int(eff.widths.get(name, 3))
                    if not 1 <= w <= 3:
                        raise ValueError(f"component {name}: width must be 1..3")
                    known = dict(self.columns)
                    if name in known and known[name] != w:
                        raise ValueError(f"component {name}: conflicting widths")
                    if name not in known:
                        self.columns.append((name, w))
                    slot = [c for c, _ in self.columns].index(name)
                    kwargs[name] = Vec([leaf(f"aux{slot}_{k}") for k in range(w)])
            out = eff.fn(**kwargs)
            if not isinstance(out, SpatialForce):
                raise TypeError(f"effector {eff.__name__} must return a dsl.SpatialForce")
            force = out
        if len(self.columns) > 4:
            raise ValueError("a generated pipe can read at most 4 component columns")
        self.torque, self.linear = force.torque(), force.force()
        self.outputs: List[Expr] = list(self.torque.e) + list(self.linear.e)
        deps = set()
        seen = set()

        def walk(e: Expr):
            if id(e) in seen:
                return
            seen.add(id(e))
            if e.op == "leaf":
                deps.add(e.name)
            for a in e.args:
                walk(a)
        for o in self.outputs:
            walk(o)
        self.leaves = deps
        self.reads_velocity = any(n[0] in "wv" and len(n) == 2 for n in deps)
        self.world_torque = not all(t.is_const(0.0) for t in self.torque.e)


def pipe(*effectors: Effector) -> "Pipe":
    flat: List[Effector] = []
    for e in effectors:
        flat.extend(e.effectors if isinstance(e, Pipe) else [e])
    return Pipe(flat)


class Pipe:
    """An ordered effector pipe (`a | b | c`), traced lazily."""

    def __init__(self, effectors: Sequence[Effector]):
        self.effectors = list(effectors)
        self._traced: Optional[TracedPipe] = None

    def __or__(self, other): return pipe(self, other)

    def trace(self) -> TracedPipe:
        if self._traced is None:
            self._traced = TracedPipe(self.effectors)
        return self._traced
