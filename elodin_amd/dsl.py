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

import numpy as _numpy          # host arrays met inside traced code (module-level constants of a user script); `np` below is _Np
from typing import Callable, Dict, List, Optional, Sequence, Tuple, Union

Number = Union[int, float]


class Expr:
    """One scalar node of the DAG."""
    __slots__ = ("op", "args", "value", "name", "seq")
    _interned: Dict[tuple, "Expr"] = {}
    _count = [0]      # creation counter: `seq` orders nodes the way the user's program created them (arguments first)

    def __new__(cls, op: str, args: tuple = (), value=None, name: Optional[str] = None):
        key = (op, tuple(id(a) for a in args), value, name)
        hit = cls._interned.get(key)
        if hit is not None:
            return hit
        self = object.__new__(cls)
        self.op, self.args, self.value, self.name = op, args, value, name
        self.seq = cls._count[0]
        cls._count[0] += 1
        cls._interned[key] = self
        return self

    @classmethod
    def fresh(cls) -> None:
        """Start of a top-level trace: forget the hash-consing table and restart the creation counter.  A program's nodes are
        then its own — the text generated for it (emission follows `seq`) does not depend on what else the process traced
        before, so equal programs give equal sources and the JIT cache hits whatever the order.  Nodes of earlier traces
        stay valid; they are just no longer shared."""
        cls._interned = {}
        cls._count[0] = 0
        _Lax._loop_ids[0] = 0      # loop-carried leaves are named after a counter: it restarts too, or the second trace of the
        #                            same program in one process would name them differently (and miss the JIT cache)

    # numpy must not broadcast a traced value into an object array: `ndarray * traced` falls through to the reflected method
    __array_ufunc__ = None

    # ---- arithmetic -------------------------------------------------------------------------------------------
    def __add__(self, o): return NotImplemented if isinstance(o, Vec) else (_host(o) + self if _is_host_array(o) else _bin("add", self, o))
    def __radd__(self, o): return _host(o) + self if _is_host_array(o) else _bin("add", o, self)
    def __sub__(self, o): return NotImplemented if isinstance(o, Vec) else (-(_host(o) - self) if _is_host_array(o) else _bin("sub", self, o))
    def __rsub__(self, o): return _host(o) - self if _is_host_array(o) else _bin("sub", o, self)
    def __mul__(self, o): return NotImplemented if isinstance(o, Vec) else (_host(o) * self if _is_host_array(o) else _bin("mul", self, o))
    def __rmul__(self, o): return _host(o) * self if _is_host_array(o) else _bin("mul", o, self)
    def __truediv__(self, o): return NotImplemented if isinstance(o, Vec) else (self * (1.0 / _host(o)) if _is_host_array(o) else _bin("div", self, o))
    def __rtruediv__(self, o): return _host(o) / self if _is_host_array(o) else _bin("div", o, self)
    def __neg__(self): return const(-self.value) if self.op == "const" else Expr("neg", (self,))
    def __pow__(self, k):
        if isinstance(k, int) and k >= 1:      # jnp integer_pow: repeated multiply
            r = self
            for _ in range(k - 1):
                r = r * self
            return r
        if isinstance(k, int) and k == 0:
            return const(1.0)
        return _Np.power(self, k)              # jnp.power for everything else
    def __rpow__(self, b): return _Np.power(b, self)
    def __mod__(self, o): return Expr("mod", (self, _lift(o)))
    def __and__(self, o): return Expr("and", (self, _lift(o)))
    def __rand__(self, o): return Expr("and", (_lift(o), self))
    def __or__(self, o): return Expr("or", (self, _lift(o)))
    def __ror__(self, o): return Expr("or", (_lift(o), self))
    def __invert__(self): return Expr("not", (self,))
    def __eq__(self, o):              # `tick % 9 == 0` in user code (examples/drone/sensors.py:187): a traced comparison
        if isinstance(o, (Expr, int, float, _numpy.generic)):
            return Expr("eq", (self, _lift(o)))
        return NotImplemented
    def __ne__(self, o):
        if isinstance(o, (Expr, int, float, _numpy.generic)):
            return Expr("not", (Expr("eq", (self, _lift(o))),))
        return NotImplemented
    def __lt__(self, o): return _bin("lt", self, o)
    def __le__(self, o): return _bin("le", self, o)
    def __gt__(self, o): return _bin("lt", o, self)
    def __ge__(self, o): return _bin("le", o, self)
    def astype(self, dtype):
        """x.astype(int) truncates toward zero (kept as a float: columns compute in the executor's dtype); float types pass."""
        name = getattr(dtype, "__name__", str(dtype))
        return _un("trunc", self) if "int" in name else self
    dtype = _numpy.float64
    shape = ()
    def __getitem__(self, k):
        """A component of shape (1,) is held as its one value: `mdot[0]` (examples/falcon9/sim.py:445) is that value."""
        if isinstance(k, Expr):          # a traced index into the one element: jax clamps it, there is nothing else to read
            return self                  # (tested before the membership check below: `expr in tuple` would call bool() on a traced ==)
        if any(k is v for v in (Ellipsis,)) or (isinstance(k, tuple) and k == ()) or (isinstance(k, int) and k in (0, -1)) \
                or (isinstance(k, slice) and k == slice(None)):
            return self
        raise IndexError("a scalar traced value has one element")
    def reshape(self, *shape):                        # `jnp.asarray(x[0]).reshape(())` (examples/falcon9/sim.py:1288): still the one value
        shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape
        if tuple(shape) in ((), (1,), (-1,)):
            return self if tuple(shape) == () else Vec([self])
        raise ValueError(f"cannot reshape a scalar traced value to {tuple(shape)}")
    def all(self, axis=None): return self           # a 0-d comparison: `(x < fov).all()`
    def any(self, axis=None): return self

    def __bool__(self):
        raise TypeError("traced values have no truth value; use dsl.np.where / logical_and")
    __hash__ = object.__hash__

    def is_const(self, v=None) -> bool:
        return self.op == "const" and (v is None or self.value == v)


# Campaign vectorisation (elodin_amd/vectorize.py): a sim script bakes a Monte-Carlo parameter into its traced code as a plain
# Python float (`mass = float(params.get("mass"))`, examples/monte-carlo/sim.py:72-75).  While a campaign's program is traced the
# parameters hold SENTINEL values; a constant that equals one is that parameter and becomes a per-run column `mc:<name>` of the
# executor (one row per run) instead of a literal.
PARAM_SENTINELS: Dict[float, str] = {}
_ACTIVE_TABLE: List[Optional["ColumnTable"]] = [None]      # the column table of the system / pipe being traced


def const(v: Number) -> Expr:
    if PARAM_SENTINELS and _ACTIVE_TABLE[0] is not None:
        name = PARAM_SENTINELS.get(float(v))
        if name is not None:
            return _ACTIVE_TABLE[0].symbols("mc:" + name, 1, 1)[0]
    return Expr("const", (), float(v))


def _lift(x) -> Expr:
    if isinstance(x, Expr):
        return x
    if isinstance(x, (int, float)):
        return const(x)
    if isinstance(x, (_numpy.generic, _numpy.ndarray)) and _numpy.ndim(x) == 0:
        return const(float(x))
    raise TypeError(f"cannot use {type(x).__name__} in a traced effector")


def _nonneg(e: "Expr", depth: int = 0) -> bool:
    """Provably >= 0 from the expression's shape alone (a clipped / counted / floored index): such an index needs no
    negative-index normalisation, and the gather keeps the select chain it always had."""
    if depth > 16:
        return False
    if e.op == "const":
        return e.value >= 0.0
    if e.op == "leaf":
        return e.name in ("tick", "dt", "dt_g")                     # the tick counter (u64) and the time steps
    if e.op in ("lt", "le", "eq", "ne", "and", "or", "not", "abs"):
        return True                                                   # comparisons are 0 / 1
    if e.op == "max":
        return any(_nonneg(a, depth + 1) for a in e.args)
    if e.op in ("add", "mul", "min"):
        return all(_nonneg(a, depth + 1) for a in e.args)
    if e.op in ("floor", "trunc", "ceil", "rint", "sqrt"):
        return _nonneg(e.args[0], depth + 1)
    if e.op == "select":
        return _nonneg(e.args[1], depth + 1) and _nonneg(e.args[2], depth + 1)
    return False


def _normalise_index(idx: "Expr", n: int) -> "Expr":
    """jax's treatment of a dynamic index: a negative one counts from the end (idx + n) before anything clamps or drops it."""
    return idx if _nonneg(idx) else Expr("select", (idx < 0.0, idx + float(n), idx))


def _dynamic_index(items, idx: "Expr"):
    """items[idx] for a traced idx: selects on idx == k over scalars, vectors or rows.  Like jax's gather the index is
    normalised first (a negative idx counts from the end: x[-1] is the last element) and THEN clamped into [0, n)."""
    n = len(items)
    idx = _normalise_index(idx, n)
    out = items[n - 1]
    for k in range(n - 2, -1, -1):
        cond = idx < (k + 0.5)           # idx <= k (integral values): rows above the last match fall through, clamping both ends
        out = _tree_select(cond, items[k], out)
    return out


# ---- constant tables in device memory: a traced gather --------------------------------------------------------------------
# `table[idx, 0]` with `table` a host array of thousands of rows and `idx` computed per entity (examples/monte-carlo/sim.py:84-97:
# a 262,144-row drag table; an EGM-style coefficient table is the same thing) cannot be a select chain.  The table becomes a
# `__device__ const` array of the generated translation unit — in HBM once, shared by every entity — and the read a per-lane
# global load.  jax's gather semantics: a negative index counts from the end, then the index is clamped into [0, rows).
GATHER_MIN_ROWS = 33                        # shorter tables stay select chains (registers, no memory access)
_GATHER_TABLES: Dict[str, "_numpy.ndarray"] = {}      # content key -> [rows, cols] float64


def _gather_key(table) -> str:
    import hashlib
    t = _numpy.ascontiguousarray(table, dtype=_numpy.float64)
    t = t.reshape(t.shape[0], -1)
    key = "g" + hashlib.sha256(t.tobytes() + repr(t.shape).encode()).hexdigest()[:16]
    _GATHER_TABLES.setdefault(key, t)
    return key


def gather(table, row, col: int = 0) -> "Expr":
    """table[row, col] for a traced `row` (integral values) over a constant host table [rows] or [rows, cols]."""
    key = _gather_key(table)
    t = _GATHER_TABLES[key]
    n, w = t.shape
    col = int(col) + (w if int(col) < 0 else 0)
    if not 0 <= col < w:
        raise IndexError(f"column {col} of a table with {w} columns")
    row = _lift(row)
    if row.op == "const":
        i = int(row.value)
        i = i + n if i < 0 else i
        return const(float(t[min(max(i, 0), n - 1), col]))
    return Expr("gather", (row,), (key, col, n, w))


class HostTable:
    """A constant host array indexable by traced rows: `tab[i]`, `tab[i, c]`, `tab[rows_vec, c]` (each element a gather)."""

    def __init__(self, array):
        self.array = _numpy.ascontiguousarray(array, dtype=_numpy.float64)
        if self.array.ndim not in (1, 2):
            raise TypeError("a gather table has one or two dimensions")
        self.shape = self.array.shape

    def __len__(self): return self.shape[0]

    def __getitem__(self, idx):
        row, col = (idx if isinstance(idx, tuple) else (idx, None))
        if isinstance(idx, tuple) and len(idx) != 2:
            raise IndexError("a gather table is indexed [row] or [row, column]")
        two_d = self.array.ndim == 2
        if two_d and col is None:                      # a whole row of a 2-D table
            cols = list(range(self.shape[1]))
        elif two_d and isinstance(col, slice):
            cols = list(range(self.shape[1]))[col]
        elif two_d:
            cols = int(col)
        else:
            if col is not None:
                raise IndexError("too many indices for a 1-D table")
            cols = 0
        rows = row.e if isinstance(row, Vec) else ([_lift(x) for x in row] if isinstance(row, (list, tuple)) else None)

        def one(r):
            return Vec([gather(self.array, r, c) for c in cols]) if isinstance(cols, list) else gather(self.array, r, cols)
        if rows is None:
            return one(row)
        out = [one(r) for r in rows]
        return out if isinstance(cols, list) else Vec(out)


def map_coordinates(grid, coordinates, order: int = 1, mode: str = "constant", cval: float = 0.0):
    """jax.scipy.ndimage.map_coordinates(input, coordinates, order, mode, cval) for a CONSTANT n-d `grid` and one traced
    coordinate per axis (examples/rocket/main.py:368: a 3 x 5 x 4 aerodynamic table interpolated at (Mach, fin deflection,
    angle of attack)).  order 0 (nearest sample) or 1 (multilinear); modes "nearest" (indices clamped to the grid) and
    "constant" (samples outside it read `cval`).  Same arithmetic, in the same order, as jax's _map_coordinates: per axis
    lower = floor(c), weights (1 - (c - lower), c - lower); contributions grid[corner] * (w0 * w1 * ...) summed over
    itertools.product of the per-axis (index, weight) pairs, first axis slowest.  Corner values come from the flattened grid
    by a traced linear index: a select chain for small grids, a gather from device memory for larger ones."""
    import itertools

    def concrete(x):      # the grid may arrive as the tracer's own constant containers (compat hands host arrays over that way)
        if isinstance(x, Expr):
            if x.op != "const":
                raise NotImplementedError("map_coordinates: the sampled grid must be a constant")
            return x.value
        if isinstance(x, Vec):
            return [concrete(e) for e in x.e]
        if isinstance(x, (list, tuple)) or type(x).__name__ in ("Mat", "Batch"):
            return [concrete(e) for e in x]
        return x
    g = _numpy.ascontiguousarray(concrete(grid), dtype=_numpy.float64)
    coords = list(coordinates.e) if isinstance(coordinates, Vec) else list(coordinates)
    if len(coords) != g.ndim:
        raise ValueError("coordinates must be a sequence of length input.ndim")
    if order not in (0, 1):
        raise NotImplementedError("map_coordinates: order must be 0 or 1")
    if mode not in ("nearest", "constant"):
        raise NotImplementedError(f"map_coordinates: mode {mode!r} is not provided (nearest, constant)")
    flat = g.reshape(-1)
    strides = [int(_numpy.prod(g.shape[a + 1:])) for a in range(g.ndim)]
    per_axis = []
    for c, size in zip(coords, g.shape):
        c = _lift(c)
        if order == 0:
            items = [(_un("rint", c), const(1.0))]        # round half to even, like jnp.round
        else:
            lower = _un("floor", c)
            upper_weight = c - lower
            items = [(lower, 1.0 - upper_weight), (lower + 1.0, upper_weight)]
        fixed = []
        for index, weight in items:
            inside = (index > -0.5) & (index < size - 0.5) if mode == "constant" else None
            fixed.append((_Np.clip(index, 0.0, float(size - 1)), weight, inside))
        per_axis.append(fixed)
    table = HostTable(flat) if flat.size >= GATHER_MIN_ROWS else Vec([float(v) for v in flat])
    total = None
    for items in itertools.product(*per_axis):
        lin = None
        for (index, _, _), stride in zip(items, strides):
            term = index * float(stride)
            lin = term if lin is None else lin + term
        value = table[lin]
        if mode == "constant":
            ok = None
            for _, _, inside in items:
                ok = inside if ok is None else (ok & inside)
            value = _Np.where(ok, value, float(cval))
        weight = None
        for _, w, _ in items:
            weight = w if weight is None else weight * w
        contribution = value * weight
        total = contribution if total is None else total + contribution
    return total


def _is_host_array(o) -> bool:
    return isinstance(o, _numpy.ndarray) and o.ndim >= 1


def _host(o):
    """A host array met inside traced code (a module-level constant of a user script) as a traced constant: 1-D -> Vec,
    2-D -> a matrix (dsl_mat.Mat)."""
    if isinstance(o, _numpy.ndarray):
        if o.ndim == 1:
            return Vec([float(x) for x in o])
        if o.ndim == 2:
            from . import dsl_mat
            return dsl_mat.Mat([[float(x) for x in r] for r in o])
        if o.ndim == 0:
            return const(float(o))
        if o.ndim == 3:
            from . import dsl_mat
            return dsl_mat.Batch([_host(m) for m in o])
        raise TypeError("arrays of more than three dimensions cannot enter traced code")
    return o


# > 0 while user code is being called on symbols (frontend decorators at decoration time, the tracers at build time): a
# script-compatibility layer (elodin_amd/compat.py) then hands out traced values where plain numpy would do outside
TRACING = [0]


class _Tracing:
    """`with tracing(table):` — user code is being called on symbols; `table` (when given) is where columns it implies live."""
    def __init__(self, table=None): self.table, self.saved = table, None
    def __enter__(self):
        TRACING[0] += 1
        self.saved, _ACTIVE_TABLE[0] = _ACTIVE_TABLE[0], (self.table if self.table is not None else _ACTIVE_TABLE[0])
    def __exit__(self, *a):
        TRACING[0] -= 1
        _ACTIVE_TABLE[0] = self.saved


tracing = _Tracing


def _const_tree(e: "Expr") -> bool:
    """const, or a select whose both branches are const trees (e.g. a time constant picked by regime)."""
    return e.op == "const" or (e.op == "select" and _const_tree(e.args[1]) and _const_tree(e.args[2]))


def _map_const_tree(e: "Expr", f) -> "Expr":
    if e.op == "const":
        return f(e)
    return Expr("select", (e.args[0], _map_const_tree(e.args[1], f), _map_const_tree(e.args[2], f)))


_FOLD1 = {"sqrt": math.sqrt, "abs": abs, "sin": math.sin, "cos": math.cos, "tan": math.tan, "exp": math.exp, "log": math.log,
          "acos": math.acos, "asin": math.asin, "floor": math.floor, "ceil": math.ceil, "trunc": math.trunc}


def _un(op: str, x) -> "Expr":
    """Unary node with constant folding, also through selects of constants: exp(-dt / where(c, tau1, tau2)) becomes
    where(c, k1, k2) — what a JIT's constant propagation does with a regime-dependent time constant."""
    x = _lift(x)
    if x.op == "const" and op in _FOLD1:
        try:
            return const(_FOLD1[op](x.value))
        except (ValueError, OverflowError):
            pass
    elif x.op == "select" and op in _FOLD1 and _const_tree(x):
        return _map_const_tree(x, lambda c: _un(op, c))
    return Expr(op, (x,))


# RELAXED ARITHMETIC (opt-in, `with relaxed_arithmetic():` around a trace — stablehlo.world_system(arith="relaxed")).  The default
# DAG is the user's program operation for operation: every node a correctly rounded IEEE value, `0 * x` kept because x may be
# Inf / NaN.  Relaxed trades that for the forms a hand-written kernel uses and stays inside BASELINE's 1e-9: it ASSUMES FINITE
# VALUES (`0 * x` is 0, which with x + 0 = x removes the zero halves of `q (x) [v, 0]`), turns `a / d` into `a * (1 / d)` so that
# the quotients of one denominator (a quaternion normalised, inverted, a wrench over one mass) share ONE division through the
# hash-consing of nodes, and lets the compiler contract `a * b + c` (System.fp_contract -> codegen).  Results differ from the
# reference's in the last bits (<= a few ulp per operation), not bit for bit.
_RELAXED = [False]


class relaxed_arithmetic:
    def __enter__(self):
        self.saved, _RELAXED[0] = _RELAXED[0], True
    def __exit__(self, *a):
        _RELAXED[0] = self.saved


def _sum_of_squares(n: "Expr"):
    """[t0, t1, ...] when n = (((t0 * t0) + t1 * t1) + t2 * t2) ... (signs of the t stripped), else None."""
    terms = []
    while n.op == "add" and len(n.args) == 2:
        terms.append(n.args[1])
        n = n.args[0]
    terms.append(n)
    out = []
    for t in reversed(terms):
        if t.op != "mul" or t.args[0] is not t.args[1]:
            return None
        x = t.args[0]
        out.append(x.args[0] if x.op == "neg" else x)
    return out if len(out) >= 2 else None


def _unit_norm_squared(n: "Expr") -> bool:
    """Relaxed arithmetic only: n is the squared norm of a vector that was just normalised — every component x_i * r with ONE
    r = 1 / sqrt(S) and S the very node that sums the squares of those x_i (hash-consing makes that an identity test).  Then
    n = 1 to rounding (a few 1e-16): the division by it is dropped, as the hand-written kernel drops it."""
    comps = _sum_of_squares(n)
    if comps is None:
        return False
    xs, r = [], None
    for c in comps:
        if c.op != "mul":
            return False
        a, b = c.args
        cand = b if (b.op == "div" and b.args[0].is_const(1.0)) else (a if (a.op == "div" and a.args[0].is_const(1.0)) else None)
        if cand is None or (r is not None and cand is not r):
            return False
        r = cand
        xs.append(a if cand is b else b)
    if r.args[1].op != "sqrt":
        return False
    base = _sum_of_squares(r.args[1].args[0])
    return base is not None and len(base) == len(xs) and all(p is q for p, q in zip(base, xs))


def _bin(op: str, a, b) -> Expr:
    a, b = _lift(a), _lift(b)
    if _RELAXED[0]:
        if op == "mul" and (a.is_const(0.0) or b.is_const(0.0)):
            return const(0.0)
        if op == "div" and a.op != "const":
            if b.op == "const" and b.value not in (0.0,) and b.value == b.value and abs(b.value) != float("inf"):
                return _bin("mul", a, const(1.0 / b.value))
            if b.op != "const":
                if _unit_norm_squared(b):      # a / |q^|^2 for q^ = q / |q|: the reference's `inverse` of a normalised quaternion (quaternion.rs:141-155)
                    return a
                return _bin("mul", a, Expr("div", (const(1.0), b)))
    if op in ("add", "sub", "mul", "div"):
        if a.op == "const" and b.op == "select" and _const_tree(b):
            return _map_const_tree(b, lambda c: _bin(op, a, c))
        if b.op == "const" and a.op == "select" and _const_tree(a):
            return _map_const_tree(a, lambda c: _bin(op, c, b))
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

    __array_ufunc__ = None

    def __init__(self, elems: Sequence):
        self.e: Tuple[Expr, ...] = tuple(_lift(x) for x in elems)

    def __len__(self): return len(self.e)
    def __iter__(self): return iter(self.e)
    dtype = _numpy.float64

    def astype(self, dtype): return Vec([a.astype(dtype) for a in self.e])

    def __getitem__(self, i):
        if isinstance(i, tuple):            # v[:, None] / v[None, :]: a column / row matrix (numpy's newaxis)
            from . import dsl_mat
            if len(i) == 2 and i[1] is None:
                return dsl_mat.Mat([[x] for x in Vec.__getitem__(self, i[0]).e])
            if len(i) == 2 and i[0] is None:
                return dsl_mat.Mat([list(Vec.__getitem__(self, i[1]).e)])
            raise IndexError(i)
        if isinstance(i, Expr):             # a TRACED index into a vector: a select chain, clamped like jax's gather
            return _dynamic_index([x for x in self.e], i)
        if isinstance(i, (list, _numpy.ndarray)):      # a constant index list: `torque[AXIS_OF_THRUSTER]` (examples/apollo-lander/sim.py:451)
            return Vec([self.e[int(k)] for k in _numpy.asarray(i).reshape(-1)])
        r = self.e[i]
        return Vec(r) if isinstance(i, slice) else r
    def _zip(self, o, f):
        if _is_host_array(o):
            o = _host(o)
        if isinstance(o, list) and o and isinstance(o[0], Vec):      # a matrix operand: numpy's row broadcast, done by the matrix
            return o._zip(self, lambda m, v: f(v, m))
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
    def __rtruediv__(self, o): return self._zip(o, lambda a, b: b / a)
    def __neg__(self): return Vec([-a for a in self.e])
    def __pow__(self, k): return Vec([a ** k for a in self.e])
    def __mod__(self, o): return self._zip(o, lambda a, b: a % b)
    def set(self, i: int, value) -> "Vec":
        """x.at[i].set(value) of jax: a copy with element i replaced."""
        e = list(self.e)
        e[i] = _lift(value)
        return Vec(e)
    @property
    def at(self): return _VecAt(self)
    @property
    def T(self): return self                       # a 1-D array is its own transpose
    @property
    def shape(self): return (len(self.e),)
    def transpose(self, *axes): return self
    def dot(self, o): return self @ o
    def flatten(self): return self
    ravel = flatten
    def reshape(self, *shape):
        from . import dsl_mat
        return dsl_mat.reshape(self, shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape)
    def __matmul__(self, o):
        from . import dsl_mat
        return dsl_mat.matmul(self, o)
    def __rmatmul__(self, o):
        from . import dsl_mat
        return dsl_mat.matmul(o, self)
    def __lt__(self, o): return self._zip(o, lambda a, b: a < b)
    def __le__(self, o): return self._zip(o, lambda a, b: a <= b)
    def __gt__(self, o): return self._zip(o, lambda a, b: a > b)
    def __ge__(self, o): return self._zip(o, lambda a, b: a >= b)
    def __and__(self, o): return self._zip(o, lambda a, b: a & b)
    def __rand__(self, o): return self._zip(o, lambda a, b: b & a)
    def __or__(self, o): return self._zip(o, lambda a, b: a | b)
    def __ror__(self, o): return self._zip(o, lambda a, b: b | a)
    def __invert__(self): return Vec([~a for a in self.e])


class _VecAt:
    """`v.at[i].set(x)` / `.add(x)` of jax for a static index or slice — or a TRACED index (`result.at[idx].set(1)`,
    examples/linalg/sim.py:371-372): every element becomes a select on `idx == k` after jax's normalisation of a negative index
    (idx + n); an index still outside [0, n) leaves the vector unchanged (jax's default scatter mode drops it)."""

    def __init__(self, v): self.v = v
    def __getitem__(self, idx): return _VecAtIdx(self.v, idx)


class _VecAtIdx:
    def __init__(self, v, idx): self.v, self.idx = v, idx

    def _apply(self, value, combine):
        e = list(self.v.e)
        if isinstance(self.idx, (Expr,)):
            val = _lift(value)
            idx = _normalise_index(self.idx, len(e))     # x.at[-1] is the last element, like jax
            return Vec([Expr("select", (Expr("eq", (idx, const(float(k)))), combine(x, val), x)) for k, x in enumerate(e)])
        ks = list(range(len(e)))[self.idx] if isinstance(self.idx, slice) else [range(len(e))[int(self.idx)]]
        vals = list(value.e) if isinstance(value, Vec) else [_lift(value)] * len(ks)
        if len(vals) != len(ks):
            raise ValueError("at[...]: value does not fit the slice")
        for k, x in zip(ks, vals):
            e[k] = combine(e[k], x)
        return Vec(e)

    def set(self, value): return self._apply(value, lambda old, new: new)
    def add(self, value): return self._apply(value, lambda old, new: old + new)


def _unary(op):
    def f(x):
        if isinstance(x, Vec):
            return Vec([_un(op, a) for a in x.e])
        if isinstance(x, list) and x and isinstance(x[0], Vec):          # a matrix (dsl_mat.Mat): element-wise
            from . import dsl_mat
            return dsl_mat.Mat([[_un(op, a) for a in r.e] for r in x])
        return _un(op, x)
    return f


class _Np:
    """The jax.numpy subset effectors in the reference's examples use."""
    pi = math.pi
    sqrt, abs, sin, cos, tan, exp, log, arccos, arcsin = (_unary(k) for k in
                                                          ("sqrt", "abs", "sin", "cos", "tan", "exp", "log", "acos", "asin"))
    log1p, expm1, cbrt, floor, ceil, trunc, sinh, cosh, erfc = (_unary(k) for k in
                                                                ("log1p", "expm1", "cbrt", "floor", "ceil", "trunc", "sinh", "cosh", "erfc"))
    round = rint = _unary("rint")          # jnp.round: half to even
    isfinite = _unary("isfinite")

    @staticmethod
    def remainder(x, y): return _zipv(x, y, lambda a, b: Expr("mod", (_lift(a), _lift(b))))   # sign of the divisor, like jnp
    mod = remainder

    @staticmethod
    def sort(v: "Vec") -> "Vec":
        """jnp.sort of a fixed-length vector: an odd-even transposition network of min / max (n rounds, data-independent)."""
        e = list(v.e)
        n = len(e)
        for rnd in range(n):
            for i in range(rnd % 2, n - 1, 2):
                lo, hi = Expr("min", (e[i], e[i + 1])), Expr("max", (e[i], e[i + 1]))
                e[i], e[i + 1] = lo, hi
        return Vec(e)

    @staticmethod
    def max(v: "Vec"):
        acc = v.e[0]
        for a in v.e[1:]:
            acc = Expr("max", (acc, a))
        return acc

    @staticmethod
    def min(v: "Vec"):
        acc = v.e[0]
        for a in v.e[1:]:
            acc = Expr("min", (acc, a))
        return acc

    @staticmethod
    def flip(v: "Vec") -> "Vec": return Vec(list(reversed(v.e)))
    @staticmethod
    def arange(n, dtype=None) -> "Vec": return Vec([float(k) for k in range(int(n))])
    @staticmethod
    def outer(a: "Vec", b: "Vec"):
        from . import dsl_mat
        return dsl_mat.Mat([[x * y for y in b.e] for x in a.e])
    @staticmethod
    def matvec(rows, v: "Vec") -> "Vec": return Vec([_Np.dot(r, v) for r in rows])    # `mat @ v` for a list of rows

    @staticmethod
    def array(x, dtype=None):
        """jnp.array of a flat sequence -> Vec; of a sequence of rows -> a matrix (dsl_mat.Mat)."""
        from . import dsl_mat
        if isinstance(x, (Vec, dsl_mat.Mat, Expr)):
            return x
        if isinstance(x, (int, float, bool)):        # a 0-d array: `np.array(0.0)` as a fold's initial value
            return _lift(float(x))
        x = list(x)
        if x and isinstance(x[0], (list, tuple, Vec)):
            return dsl_mat.Mat(x)
        return Vec(x)
    @staticmethod
    def broadcast_to(x, shape):
        """jnp.broadcast_to of a scalar / vector to (n,) or (rows, n)."""
        from . import dsl_mat
        shape = tuple(int(k) for k in (shape if hasattr(shape, "__len__") else (shape,)))
        x = _host(x)
        row = x if isinstance(x, Vec) else Vec([x] * shape[-1])
        if len(row) != shape[-1]:
            if len(row) != 1:
                raise ValueError(f"cannot broadcast a vector of {len(row)} to {shape}")
            row = Vec([row.e[0]] * shape[-1])
        if len(shape) == 1:
            return row
        if len(shape) == 2:
            return dsl_mat.Mat([Vec(list(row.e)) for _ in range(shape[0])])
        raise NotImplementedError("broadcast_to beyond two dimensions")
    @staticmethod
    def zeros(n, dtype=None, shape=None):
        from . import dsl_mat
        n = shape if shape is not None else n
        if isinstance(n, (tuple, list)):
            return dsl_mat.zeros2(*n) if len(n) == 2 else Vec([0.0] * int(n[0]))
        return Vec([0.0] * int(n))
    @staticmethod
    def eye(n, m=None, dtype=None):
        from . import dsl_mat
        return dsl_mat.eye(n, m)
    identity = eye
    @staticmethod
    def diag(x):
        from . import dsl_mat
        return dsl_mat.diag(x)
    @staticmethod
    def block(blocks):
        from . import dsl_mat
        return dsl_mat.block(blocks)
    @staticmethod
    def transpose(x, axes=None): return x.T
    @staticmethod
    def swapaxes(x, a, b):
        x = _host(x)
        return x.mT if hasattr(x, "mT") else x.T
    @staticmethod
    def trace(x):
        from . import dsl_mat
        return dsl_mat.trace(x)
    @staticmethod
    def matmul(a, b):
        from . import dsl_mat
        return dsl_mat.matmul(a, b)
    @staticmethod
    def reshape(x, shape):
        from . import dsl_mat
        return dsl_mat.reshape(x, shape)
    @staticmethod
    def concat(parts, axis=0): return _Np.concatenate(parts, axis)
    @staticmethod
    def nan_to_num(x, nan=0.0, posinf=None, neginf=None):
        """jnp.nan_to_num: NaN -> `nan`, +-inf -> the largest finite values (or the given ones)."""
        big = 1.7976931348623157e308
        hi, lo = (big if posinf is None else posinf), (-big if neginf is None else neginf)
        def f(a):
            a = _lift(a)
            finite = Expr("isfinite", (a,))
            is_nan = Expr("not", (Expr("eq", (a, a)),))
            return Expr("select", (finite, a, Expr("select", (is_nan, const(nan), Expr("select", (a > 0.0, const(hi), const(lo)))))))
        if isinstance(x, Vec):
            return Vec([f(a) for a in x.e])
        if isinstance(x, list) and x and isinstance(x[0], Vec):
            from . import dsl_mat
            return dsl_mat.Mat([[f(a) for a in r.e] for r in x])
        return f(x)
    @staticmethod
    def positive(x): return x
    @staticmethod
    def linspace(a, b, n): return Vec([a + (b - a) * k / (n - 1) for k in range(int(n))])
    @staticmethod
    def sum(v, axis=None):
        v = _host(v)
        if isinstance(v, list) and v and isinstance(v[0], Vec):       # a matrix: all elements, or along an axis
            if axis is None:
                return _Np.sum(Vec([e for r in v for e in r.e]))
            rows = v if axis in (1, -1) else v.T
            return Vec([_Np.sum(r) for r in rows])
        acc = v.e[0]
        for a in v.e[1:]:
            acc = acc + a
        return acc
    @staticmethod
    def dot(a, b):
        if isinstance(a, Vec) and isinstance(b, Vec):
            return _Np.sum(a * b)
        from . import dsl_mat
        return dsl_mat.matmul(a, b)
    @staticmethod
    def cross(a, b):
        a, b = _host(a), _host(b)
        am, bm = (isinstance(x, list) and x and isinstance(x[0], Vec) for x in (a, b))
        if am or bm:                        # [n, 3] x [n, 3] (or x [3]): row by row, like jnp.cross over the last axis
            from . import dsl_mat
            n = len(a) if am else len(b)
            return dsl_mat.Mat([_Np.cross(a[i] if am else a, b[i] if bm else b) for i in range(n)])
        return Vec([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]])
    @staticmethod
    def maximum(a, b): return _zipv(a, b, lambda x, y: Expr("max", (_lift(x), _lift(y))))
    @staticmethod
    def minimum(a, b): return _zipv(a, b, lambda x, y: Expr("min", (_lift(x), _lift(y))))
    @staticmethod
    def clip(x, lo=None, hi=None, min=None, max=None):       # noqa: A002  (jnp.clip's keyword names)
        lo, hi = (min if lo is None else lo), (max if hi is None else hi)
        if lo is not None:
            x = _Np.maximum(x, lo)
        return x if hi is None else _Np.minimum(x, hi)
    @staticmethod
    def where(c, a, b):
        if (isinstance(a, list) and a and isinstance(a[0], Vec)) or (isinstance(b, list) and b and isinstance(b[0], Vec)):
            from . import dsl_mat
            m = a if isinstance(a, list) else b
            am = a if isinstance(a, list) else [Vec([a] * len(m[0]))] * len(m)
            bm = b if isinstance(b, list) else [Vec([b] * len(m[0]))] * len(m)
            cm = c if isinstance(c, list) else [c] * len(m)
            return dsl_mat.Mat([_Np.where(ck, x, y) for ck, x, y in zip(cm, am, bm)])
        if isinstance(a, Vec) or isinstance(b, Vec) or isinstance(c, Vec):      # scalars broadcast against the vector among them
            n = len(next(v for v in (a, b, c) if isinstance(v, Vec)))
            av = a if isinstance(a, Vec) else Vec([a] * n)
            bv = b if isinstance(b, Vec) else Vec([b] * n)
            cv = c.e if isinstance(c, Vec) else [c] * n
            if len(cv) == 1 and n > 1:        # a [rows, 1] condition against [rows, n] values, row by row
                cv = list(cv) * n
            if not (len(cv) == len(av) == len(bv)):
                raise ValueError("where: shape mismatch")
            return Vec([Expr("select", (_lift(k), x, y)) for k, x, y in zip(cv, av.e, bv.e)])
        return Expr("select", (_lift(c), _lift(a), _lift(b)))
    @staticmethod
    def equal(a, b): return _zipv(a, b, lambda x, y: Expr("eq", (_lift(x), _lift(y))))
    @staticmethod
    def logical_and(a, b): return _zipv(a, b, lambda x, y: Expr("and", (_lift(x), _lift(y))))
    @staticmethod
    def logical_or(a, b): return _zipv(a, b, lambda x, y: Expr("or", (_lift(x), _lift(y))))
    @staticmethod
    def logical_not(a): return Vec([Expr("not", (x,)) for x in a.e]) if isinstance(a, Vec) else Expr("not", (_lift(a),))
    # integer components (I64 flags / counters) are integral values in the executor's float columns: bitwise ops act on them
    @staticmethod
    def bitwise_xor(a, b): return _zipv(a, b, lambda x, y: Expr("bxor", (_lift(x), _lift(y))))
    @staticmethod
    def bitwise_or(a, b): return _zipv(a, b, lambda x, y: Expr("bor", (_lift(x), _lift(y))))
    @staticmethod
    def bitwise_and(a, b): return _zipv(a, b, lambda x, y: Expr("band", (_lift(x), _lift(y))))
    @staticmethod
    def left_shift(a, b): return _zipv(a, b, lambda x, y: Expr("shl", (_lift(x), _lift(y))))
    @staticmethod
    def right_shift(a, b): return _zipv(a, b, lambda x, y: Expr("shr", (_lift(x), _lift(y))))      # non-negative operands
    @staticmethod
    def arctan2(y, x): return _zipv(y, x, lambda a, b: Expr("atan2", (_lift(a), _lift(b))))
    @staticmethod
    def hypot(x, y): return _zipv(x, y, lambda a, b: Expr("hypot", (_lift(a), _lift(b))))
    @staticmethod
    def arctan(x): return _zipv(x, 1.0, lambda a, b: Expr("atan2", (_lift(a), _lift(b))))   # atan2(x, 1) == atan(x)
    @staticmethod
    def power(x, y): return _zipv(x, y, lambda a, b: Expr("pow", (_lift(a), _lift(b))))
    @staticmethod
    def concatenate(parts, axis=0):
        parts = list(parts)
        if parts and isinstance(parts[0], WindowSlice) and axis == 0:
            # `jnp.concatenate((buffer[1:], row.reshape(1, w)))` (examples/rocket/main.py:449): drop the oldest row, append one = push
            w = parts[0]
            rest = [r for p_ in parts[1:] for r in (list(p_) if isinstance(p_, list) else [p_])]
            if w.start == 1 and w.stop == w.window.rows and len(rest) == 1 and isinstance(rest[0], Vec):
                return w.window.push(rest[0])
            raise NotImplementedError("concatenate over window rows: only `concatenate((window[1:], one_row))` (a push) is provided")
        if any(isinstance(p_, LazyRows) for p_ in parts) and axis == 0:
            return LazyRows([q for p_ in parts for q in (p_.parts if isinstance(p_, LazyRows) else [list(p_)])])
        parts = [_host(p_) for p_ in parts]
        if parts and isinstance(parts[0], list) and parts[0] and isinstance(parts[0][0], Vec):     # matrices
            from . import dsl_mat
            if axis == 0:
                return dsl_mat.Mat([r for m in parts for r in m])
            return dsl_mat.Mat([[e for m in parts for e in m[i].e] for i in range(len(parts[0]))])
        out = []
        for v in parts:
            out.extend(v.e if isinstance(v, Vec) else [_lift(v)])
        return Vec(out)
    @staticmethod
    def ones(n, dtype=None):
        from . import dsl_mat
        if isinstance(n, (tuple, list)):
            return dsl_mat.Mat([[1.0] * int(n[1]) for _ in range(int(n[0]))]) if len(n) == 2 else Vec([1.0] * int(n[0]))
        return Vec([1.0] * int(n))
    # ---- the remaining array-construction / cast / reduction calls the reference's scripts make on this path ----------
    @staticmethod
    def asarray(x, dtype=None):
        """jnp.asarray: sequences become vectors, traced values and scalars pass through."""
        if isinstance(x, (Vec, Expr)) or not hasattr(x, "__len__"):
            return x if isinstance(x, (Vec, Expr)) else _lift(x)
        return Vec(list(x))
    @staticmethod
    def float64(x): return x if isinstance(x, (Vec, Expr)) else _lift(x)      # columns already compute in the executor's dtype
    float32 = float64
    @staticmethod
    def int32(x): return _Np.trunc(x if isinstance(x, (Vec, Expr)) else _lift(x))   # astype(int): toward zero, kept as a float
    int64 = int32
    @staticmethod
    def stack(parts, axis=0):
        """jnp.stack of scalars -> a vector; of vectors -> a matrix as a list of rows (usable with matvec / outer results)."""
        parts = [_host(p_) for p_ in parts]
        if parts and isinstance(parts[0], Vec):
            from . import dsl_mat
            m = dsl_mat.Mat(parts)
            return m.T if axis in (1, -1) else m
        return Vec(parts)
    @staticmethod
    def zeros_like(x): return Vec([0.0] * len(x)) if isinstance(x, Vec) else const(0.0)
    @staticmethod
    def ones_like(x): return Vec([1.0] * len(x)) if isinstance(x, Vec) else const(1.0)
    @staticmethod
    def full(n, value, dtype=None): return Vec([value] * int(n[0] if hasattr(n, "__len__") else n))
    @staticmethod
    def reciprocal(x): return 1.0 / x
    @staticmethod
    def square(x): return x * x
    @staticmethod
    def negative(x): return -x
    @staticmethod
    def mean(v: Vec): return _Np.sum(v) / float(len(v))
    @staticmethod
    def degrees(x): return _Np.rad2deg(x)
    @staticmethod
    def radians(x): return _Np.deg2rad(x)
    @staticmethod
    def searchsorted(a, v, side="left"):
        """jnp.searchsorted over a CONSTANT sorted table: the count of entries below (side='left') / not above (side='right') v."""
        v = _lift(v)
        table = [float(t.value) if isinstance(t, Expr) and t.op == "const" else float(t) for t in a]
        hits = [(_Np.where(v > t, 1.0, 0.0) if side == "left" else _Np.where(v >= t, 1.0, 0.0)) for t in table]
        return _Np.sum(Vec(hits)) if hits else const(0.0)
    @staticmethod
    def interp(x, xp, fp):
        """jnp.interp(x, xp, fp) with CONSTANT tables (atmosphere / thrust-curve lookups, e.g. examples/rocket/main.py:356-375)."""
        def table(t):
            t = t.e if isinstance(t, Vec) else t
            out = []
            for v in t:
                if isinstance(v, Expr):
                    if v.op != "const":
                        raise TypeError("interp tables must be constants")
                    v = v.value
                out.append(float(v))
            return tuple(out)
        xs, fs = table(xp), table(fp)
        if isinstance(x, Vec):                      # element-wise over a vector of abscissae
            return Vec([_Np.interp(a, xs, fs) for a in x.e])
        if len(xs) != len(fs) or len(xs) < 2 or any(b < a for a, b in zip(xs, xs[1:])):
            raise ValueError("interp tables must be equally long (>= 2) with non-decreasing xp")
        return Expr("interp", (_lift(x),), (xs, fs))
    @staticmethod
    def sign(x): return _Np.where(x > 0.0, 1.0, _Np.where(x < 0.0, -1.0, 0.0))
    @staticmethod
    def tanh(x):
        em = _Np.expm1(_Np.abs(x) * -2.0)        # in (-1, 0]: saturates to +-1 for large |x|, keeps the relative accuracy near 0
        r = -em / (em + 2.0)
        return _Np.where(x < 0.0, -r, r)
    @staticmethod
    def deg2rad(x): return x * (math.pi / 180.0)
    @staticmethod
    def rad2deg(x): return x * (180.0 / math.pi)

    class linalg:
        """jax.numpy.linalg for per-entity matrices (elodin_amd/dsl_mat.py: every factorisation unrolled into the kernel)."""
        @staticmethod
        def norm(v, ord=None, axis=None):                 # jnp.linalg.norm, ord=None: 2-norm of a vector, Frobenius of a matrix
            v = _host(v)
            if isinstance(v, Vec):
                return _Np.sqrt(_Np.sum(v * v))
            from . import dsl_mat
            return dsl_mat.fro_norm(v)
        @staticmethod
        def det(rows):
            """Determinant: cofactor form for 2x2 / 3x3 (no pivoting needed), LU with partial pivoting beyond."""
            if len(rows) == 2:
                return rows[0][0] * rows[1][1] - rows[0][1] * rows[1][0]
            if len(rows) == 3:
                return _Np.dot(rows[0], _Np.cross(rows[1], rows[2]))
            from . import dsl_mat
            return dsl_mat.det(rows)
        @staticmethod
        def solve(a, b):
            from . import dsl_mat
            return dsl_mat.solve(a, b)
        @staticmethod
        def inv(a):
            from . import dsl_mat
            return dsl_mat.inv(a)
        @staticmethod
        def pinv(a, rcond=None, rtol=None, hermitian=False):
            from . import dsl_mat
            return dsl_mat.pinv(a, rcond if rcond is not None else rtol)
        @staticmethod
        def slogdet(a):
            from . import dsl_mat
            return dsl_mat.slogdet(a)
        @staticmethod
        def cholesky(a, upper=False):
            from . import dsl_mat
            return dsl_mat.cholesky(a, lower=not upper)
        @staticmethod
        def qr(a, mode="reduced"):
            from . import dsl_mat
            return dsl_mat.qr(a)
        @staticmethod
        def svd(a, full_matrices=True, compute_uv=True):
            from . import dsl_mat
            return dsl_mat.svd(a, full_matrices, compute_uv)
        @staticmethod
        def eigh(a, UPLO=None, symmetrize_input=True):
            from . import dsl_mat
            return dsl_mat.eigh(a)
        @staticmethod
        def eigvalsh(a, UPLO=None):
            from . import dsl_mat
            return dsl_mat.eigvalsh(a)
        @staticmethod
        def matrix_transpose(a): return a.T


def _zipv(a, b, f):
    if isinstance(a, Vec) or isinstance(b, Vec):
        n = len(a) if isinstance(a, Vec) else len(b)
        av = a.e if isinstance(a, Vec) else [a] * n
        bv = b.e if isinstance(b, Vec) else [b] * n
        return Vec([f(x, y) for x, y in zip(av, bv)])
    return f(a, b)


# numpy 2 / jax.numpy spellings of the same functions
_Np.atan2, _Np.atan, _Np.asin, _Np.acos = _Np.arctan2, _Np.arctan, _Np.arcsin, _Np.arccos
_Np.pow, _Np.absolute = _Np.power, _Np.abs
_Np.multiply = staticmethod(lambda a, b: a * b)
_Np.add = staticmethod(lambda a, b: a + b)
_Np.subtract = staticmethod(lambda a, b: a - b)
_Np.divide = staticmethod(lambda a, b: a / b)

np = _Np


def _from_host_value(x):
    """A host spatial value (elodin_amd.api: `.arr`) or array met in traced code, as its traced twin."""
    arr = getattr(x, "arr", None)
    if arr is None:
        return _host(x) if _is_host_array(x) else x
    vec = lambda a_: Vec([float(v) for v in a_])
    kind = type(x).__name__
    if kind == "Quaternion":
        return Quaternion(vec(arr))
    if kind == "SpatialMotion":
        return SpatialMotion(vec(arr[:3]), vec(arr[3:]))
    if kind == "SpatialForce":
        return SpatialForce(torque=vec(arr[:3]), linear=vec(arr[3:]))
    if kind == "SpatialTransform":
        return SpatialTransform(Quaternion(vec(arr[:4])), vec(arr[4:]))
    if kind == "SpatialInertia":
        return SpatialInertia(vec(arr[:3]), const(float(arr[6])))
    return x


def _tree_select(c, a, b):
    """Element-wise select over the value kinds systems exchange: scalars, Vec, spatial types, tuples / dicts of those."""
    a, b = _from_host_value(a), _from_host_value(b)
    if isinstance(a, list) and a and isinstance(a[0], Vec):       # a matrix: before the generic sequence case, keeps its type
        return _Np.where(c, a, b)
    if isinstance(a, (tuple, list)):
        return type(a)(_tree_select(c, x, y) for x, y in zip(a, b))
    if isinstance(a, dict):
        return {k: _tree_select(c, a[k], b[k]) for k in a}
    if isinstance(a, SpatialMotion):
        return SpatialMotion(_Np.where(c, a.angular(), b.angular()), _Np.where(c, a.linear(), b.linear()))
    if isinstance(a, SpatialTransform):
        return SpatialTransform(Quaternion(_Np.where(c, a.angular().vector(), b.angular().vector())),
                                _Np.where(c, a.linear(), b.linear()))
    if isinstance(a, Quaternion):
        return Quaternion(_Np.where(c, a.vector(), b.vector()))
    if isinstance(a, SpatialInertia):
        return SpatialInertia(_Np.where(c, a.inertia_diag(), b.inertia_diag()), _Np.where(c, a.mass(), b.mass()))
    if isinstance(a, SpatialForce) or isinstance(b, SpatialForce):
        r = SpatialForce(linear=_Np.where(c, a._f, b._f), _tw=_Np.where(c, a._tw, b._tw), _tb=_Np.where(c, a._tb, b._tb))
        r._q = a._q or b._q
        return r
    return _Np.where(c, a, b)


class _Lax:
    """The jax.lax control-flow calls the reference's examples use inside per-entity code.  Per lane there is no branch
    to take: both sides are traced and the result is selected, which is also what a vmapped `lax.cond` lowers to."""

    @staticmethod
    def cond(pred, true_fun, false_fun, *operands, operand=None):
        """jax.lax.cond(pred, true_fun, false_fun, *operands) — e.g. examples/ball/sim.py:66-71 (the bounce)."""
        args = operands if operands else (operand,)
        return _tree_select(pred, true_fun(*args), false_fun(*args))

    @staticmethod
    def select(pred, on_true, on_false):
        return _tree_select(pred, on_true, on_false)

    @staticmethod
    def convert_element_type(x, dtype):
        """lax.convert_element_type: to an integer type truncates toward zero (values stay in the executor's float type)."""
        name = getattr(dtype, "__name__", str(dtype))
        f = (lambda e: _un("trunc", _lift(e))) if "int" in name else (lambda e: e)
        return Vec([f(e) for e in x.e]) if isinstance(x, Vec) else f(x)

    @staticmethod
    def branch_cond(pred, true_fun, false_fun, *operands):
        """`cond` with a REAL branch in the kernel (not in jax.lax): `true_fun(*operands)` is evaluated only by waves with a
        lane whose `pred` holds, instead of by every lane on every tick with the result selected away.  For rare, expensive
        branches — a sensor that draws its noise on one tick in forty (threefry + erfinv per sample), a relight sequence.
        Built on the loop machinery: a `while_loop` of at most one iteration whose carried values are the operands, so
        everything in `true_fun` that depends on an operand stays inside the branch (what does not is hoisted in front of it
        like any loop-invariant code).  `false_fun(*operands)` supplies the values of the lanes that do not branch and is
        evaluated by all: keep it cheap.  Both must return the same structure."""
        ops_flat, ops_rebuild = _flatten(list(operands))
        out_false, out_rebuild = _flatten(false_fun(*operands))
        n_ops = len(ops_flat)
        limit = _Np.where(pred, 1.0, 0.0)

        def cond(c):
            return c[0] < limit

        def body(c):
            out_true, _ = _flatten(true_fun(*ops_rebuild(list(c[1:1 + n_ops]))))
            if len(out_true) != len(out_false):
                raise TypeError("branch_cond: true_fun and false_fun must return the same structure")
            return [c[0] + 1.0] + list(c[1:1 + n_ops]) + list(out_true)
        res = _Lax.while_loop(cond, body, [const(0.0)] + list(ops_flat) + list(out_false), max_iter=1)
        return out_rebuild(list(res[1 + n_ops:]))

    @staticmethod
    def switch(index, branches, *operands):
        """jax.lax.switch: index clamped to [0, len(branches) - 1]."""
        out = branches[0](*operands)
        for k in range(1, len(branches)):
            out = _tree_select(_lift(index) >= (k - 0.5), branches[k](*operands), out)
        return out

    @staticmethod
    def fori_loop(lower: int, upper: int, body_fun, init_val):
        """jax.lax.fori_loop with STATIC bounds: unrolled at trace time (the loop index is a Python int)."""
        if not (isinstance(lower, int) and isinstance(upper, int)):
            raise TypeError("fori_loop bounds must be Python ints (the loop is unrolled while tracing)")
        val = init_val
        for i in range(lower, upper):
            val = body_fun(i, val)
        return val

    @staticmethod
    def scan(f, init, xs=None, length: Optional[int] = None):
        """jax.lax.scan(f, init, xs) -> (carry, ys) over a STATIC-length leading axis, unrolled at trace time — the shape
        the reference's own `edge_fold` lowers to (vmap of scan over the gathered out-edges,
        libs/nox-py/python/elodin/__init__.py:524-544) and what a filter bank over a short sample window looks like.
        `xs`: a Vec (one scalar per step), a list / tuple of per-step values (rows), or a tuple / dict of those (every
        leaf indexed by the step); None with `length` = scan over range(length).  `ys` come back stacked: scalars as one
        Vec, Vec rows as a list of rows (`np.stack`-able), tuples / dicts leaf by leaf; a step returning None for y is fine."""
        def n_steps(x):
            if isinstance(x, Vec):
                return len(x)
            if isinstance(x, dict):
                return n_steps(next(iter(x.values())))
            if isinstance(x, tuple) and x and isinstance(x[0], (Vec, list, tuple, dict)):
                return n_steps(x[0])
            return len(x)

        def at(x, i):
            if isinstance(x, Vec):
                return x[i]
            if isinstance(x, dict):
                return {k: at(v, i) for k, v in x.items()}
            if isinstance(x, tuple):          # a tuple of scanned operands, like jax pytrees
                return tuple(at(v, i) for v in x)
            return x[i]                        # list of per-step rows
        if isinstance(xs, WindowSlice):
            # a REAL loop over the window's rows in the kernel; the last step's output rides along in the carry, the first is
            # the loop's first iteration written out (emitted only if something reads it): LazyRows
            if len(xs) == 0:
                return init, LazyRows([])
            carry1, y_first = f(init, xs.window[xs.start])
            yf, y_rebuild = _flatten(y_first)

            def step(c, row):
                nc, y = f(c[0], row)
                return (nc, y), None
            last_carry, y_last = xs.window.scan(step, (init, y_rebuild([const(0.0)] * len(yf))), start=xs.start, stop=xs.stop)
            return last_carry, LazyRows.of_scan(y_first, y_last, len(xs))
        L = int(length) if xs is None else n_steps(xs)
        if length is not None and xs is not None and int(length) != L:
            raise ValueError("scan: length disagrees with the leading axis of xs")
        carry, ys = init, []
        for i in range(L):
            carry, y = f(carry, None if xs is None else at(xs, i))
            ys.append(y)

        def stack(vals):
            v0 = vals[0] if vals else None
            if v0 is None:
                return None
            if isinstance(v0, dict):
                return {k: stack([v[k] for v in vals]) for k in v0}
            if isinstance(v0, tuple):
                return tuple(stack([v[j] for v in vals]) for j in range(len(v0)))
            if isinstance(v0, Vec):
                return list(vals)             # [L] rows of width w
            return Vec([_lift(v) for v in vals])
        return carry, stack(ys)

    _loop_ids = [0]

    @staticmethod
    def while_loop(cond_fun, body_fun, init_val, max_iter: int = 1_000_000, counted: Optional[Tuple[int, ...]] = None):
        """jax.lax.while_loop with a data-dependent trip count (e.g. examples/stablehlo/sim.py:223, or a flight
        computer propagating a ballistic arc until it meets the ground): becomes a real loop in the generated kernel,
        lanes leave it independently.  `init_val`: a scalar, a Vec, or a tuple / list of those.  `max_iter` bounds the
        loop (a lane whose condition never turns false stops there)."""
        flat, rebuild = _flatten(init_val)
        lid = _Lax._loop_ids[0]
        _Lax._loop_ids[0] += 1
        names = tuple(f"lv{lid}_{j}" for j in range(len(flat)))
        carried = rebuild([leaf(nm) for nm in names])
        cond = _lift(cond_fun(carried))
        body, _ = _flatten(body_fun(carried))
        if len(body) != len(flat):
            raise TypeError("while_loop body must return the structure of init_val")
        # counted=(start, stop[, unroll]): the loop runs exactly stop - start times and the condition agrees (Window.scan's counter; a
        # statically counted stablehlo.while) — the code generator emits a plain counted loop it can unroll (by 8, or `unroll`) and
        # software-pipeline
        node = Expr("while", tuple(_lift(x) for x in flat),
                    (names, cond, tuple(_lift(b) for b in body), int(max_iter)) + ((tuple(counted),) if counted else ()))
        return rebuild([Expr("while_out", (node,), j) for j in range(len(flat))])

    @staticmethod
    def max(a, b): return _Np.maximum(a, b)
    @staticmethod
    def min(a, b): return _Np.minimum(a, b)
    @staticmethod
    def rsqrt(x): return 1.0 / _Np.sqrt(x)
    @staticmethod
    def shift_right_logical(a, b): return _Np.right_shift(a, b)
    @staticmethod
    def shift_left(a, b): return _Np.left_shift(a, b)


def _flatten(tree):
    """(flat list of scalar nodes, rebuild(list) -> same structure) for scalars, Vec and tuples / lists of those."""
    if isinstance(tree, (tuple, list)):
        parts = [_flatten(t) for t in tree]
        sizes = [len(p[0]) for p in parts]

        def rebuild(xs, parts=parts, sizes=sizes, kind=type(tree)):
            out, k = [], 0
            for (_, rb), sz in zip(parts, sizes):
                out.append(rb(xs[k:k + sz]))
                k += sz
            return kind(out)
        return [x for p in parts for x in p[0]], rebuild
    if isinstance(tree, Vec):
        return list(tree.e), (lambda xs: Vec(xs))
    return [_lift(tree)], (lambda xs: xs[0])


lax = _Lax


class _Key:
    """A jax.random key: two uint32 words (held exactly in f64 nodes)."""

    def __init__(self, k0, k1): self.k0, self.k1 = _lift(k0), _lift(k1)


def _threefry2x32_host(k0: int, k1: int, x0: int, x1: int):
    """threefry2x32 on Python ints (the block function behind jax.random; the generated kernels carry the same rounds as
    m_threefry, codegen.py): 20 rounds in five groups of four, a key-schedule word injected after each group."""
    m = 0xFFFFFFFF
    ks = (k0 & m, k1 & m, (k0 ^ k1 ^ 0x1BD11BDA) & m)
    rot = ((13, 15, 26, 6), (17, 29, 16, 24))
    x0, x1 = (x0 + ks[0]) & m, (x1 + ks[1]) & m
    for g in range(5):
        for r in rot[g % 2]:
            x0 = (x0 + x1) & m
            x1 = ((x1 << r) | (x1 >> (32 - r))) & m
            x1 ^= x0
        x0 = (x0 + ks[(g + 1) % 3]) & m
        x1 = (x1 + ks[(g + 2) % 3] + g + 1) & m
    return x0, x1


class _Random:
    """jax.random for per-entity code, bit-compatible with JAX's default threefry2x32 generator in its partitionable
    layout (the default of the reference's JAX): examples/ball/sim.py:92-94 `random.normal(random.key(seed), shape=(3,))`
    reproduces the wind row of the reference's golden CSV.  x64 semantics (64 random bits per sample): the generated code
    evaluates everything between the key words and the finished sample in double, also inside a float32 program, which
    therefore draws the same noise as its float64 twin, rounded (seeds and fold_in data must be exact in the program's
    dtype on entry: below 2**24 for float32; the tick is taken as the integer it is)."""

    @staticmethod
    def key(seed) -> _Key:
        """threefry_seed on an int64 seed: words (seed >>> 32, seed & 0xffffffff) of its two's complement — a negative seed
        (examples/cube-sat/main.py:173 seeds with a truncated ECEF coordinate) has the high word 0xffffffff."""
        seed = _lift(seed)
        hi = _un("floor", seed / 4294967296.0)
        lo = seed - hi * 4294967296.0
        if seed.op == "const":
            return _Key(float(int(hi.value) & 0xFFFFFFFF) if hi.op == "const" else hi, lo)
        return _Key(hi - _un("floor", hi / 4294967296.0) * 4294967296.0, lo)

    @staticmethod
    def _threefry(key: _Key, c0, c1):
        args = (key.k0, key.k1, _lift(c0), _lift(c1))
        if all(a.op == "const" for a in args):       # a constant key folded with a constant (fold_in(key(seed), salt)): done here
            x0, x1 = _threefry2x32_host(*[int(a.value) for a in args])
            return const(float(x0)), const(float(x1))
        return Expr("threefry", args, 0), Expr("threefry", args, 1)

    @staticmethod
    def fold_in(key: _Key, data) -> _Key:
        x0, x1 = _Random._threefry(key, 0.0, data)          # threefry_2x32(key, [0, uint32(data)])
        return _Key(x0, x1)

    @staticmethod
    def _unit(key: _Key, n: int):
        """n samples in [0, 1): 64 bits each, counter i as (hi = 0, lo = i); mantissa = bits >> 12."""
        out = []
        for i in range(n):
            hi, lo = _Random._threefry(key, 0.0, float(i))
            out.append((hi * 1048576.0 + _un("floor", lo / 4096.0)) * 2.220446049250313e-16)
        return out

    @staticmethod
    def uniform(key: _Key, shape=(), minval=0.0, maxval=1.0):
        n = int(shape[0]) if shape else 1
        u = [np.maximum(minval, x * (maxval - minval) + minval) for x in _Random._unit(key, n)]
        return Vec(u) if shape else u[0]

    @staticmethod
    def normal(key: _Key, shape=()):
        lo = -0.9999999999999999                             # nextafter(-1, 0)
        n = int(shape[0]) if shape else 1
        u = [np.maximum(lo, x * (1.0 - lo) + lo) for x in _Random._unit(key, n)]
        z = [_un("erfinv", x) * 1.4142135623730951 for x in u]
        return Vec(z) if shape else z[0]


random = _Random


# ---- spatial types (thin mirrors of libs/nox-py/src/spatial.rs wrappers) ---------------------------------------

class RotVec(Vec):
    """`stage_attitude @ x`: remembers x so a torque built this way can be handed to calc_accel in the body frame
    (alpha = q * ((q^-1 * tau) / I): q^-1 * (q * x) = x) instead of being rotated out and straight back."""

    def __init__(self, elems, body: Vec):
        super().__init__(elems)
        self.body = body


class Quaternion:
    """Scalar-last [x,y,z,w].  `unit` marks values known to have norm 1 by construction (the stage attitude, which the
    kernel renormalises every stage; normalize(), from_axis_angle(), identity(), and inverses / products of those): only
    for them may `q @ v` and `inverse()` drop the division by |q|^2 the reference always performs
    (quaternion.rs:152-155,283-305 — q @ v is scale-invariant there).  Anything else — a quaternion built from an array,
    raw spawn data a system sees on tick 0 — keeps it."""

    __array_ufunc__ = None

    def __init__(self, v: Vec, stage_attitude: bool = False, unit: bool = False):
        self.v = _host(v) if _is_host_array(v) else v
        self.stage_attitude = stage_attitude
        self.unit = bool(unit or stage_attitude)

    @staticmethod
    def _of(o) -> "Quaternion":
        """A host quaternion (api.Quaternion: `.arr`) met in traced code becomes a constant of the trace."""
        if isinstance(o, Quaternion):
            return o
        arr = getattr(o, "arr", None)
        if arr is not None and len(arr) == 4:
            return Quaternion(Vec([float(x) for x in arr]))
        raise TypeError(f"expected a quaternion, got {type(o).__name__}")

    def vector(self) -> Vec: return self.v
    def inverse(self) -> "Quaternion":
        """conj / |q|^2 (quaternion.rs:152-155).  The stage attitude inside six_dof is renormalised by the kernel before
        the effectors see it, so there the division is by 1 and is dropped."""
        c = Vec([-self.v[0], -self.v[1], -self.v[2], self.v[3]])
        return Quaternion(c, unit=True) if self.unit else Quaternion(c / np.dot(self.v, self.v))
    def conjugate(self) -> "Quaternion":
        return Quaternion(Vec([-self.v[0], -self.v[1], -self.v[2], self.v[3]]), unit=self.unit)
    def normalize(self) -> "Quaternion":                     # quaternion.rs:147-149
        return Quaternion(self.v / np.linalg.norm(self.v), unit=True)
    @staticmethod
    def identity() -> "Quaternion":
        return Quaternion(Vec([0.0, 0.0, 0.0, 1.0]), unit=True)
    @staticmethod
    def from_axis_angle(axis, angle) -> "Quaternion":         # quaternion.rs:158-171 (the axis is normalised)
        axis = axis if isinstance(axis, Vec) else Vec(list(axis))
        axis = axis / np.linalg.norm(axis)
        half = _lift(angle) / 2.0
        return Quaternion(np.concatenate([axis * np.sin(half), np.cos(half)]), unit=True)
    def integrate_body(self, body_delta: Vec) -> "Quaternion":   # quaternion.rs:176-182: q + q (x) (delta/2, 0), normalised
        half = Quaternion(np.concatenate([body_delta / 2.0, 0.0]))
        return Quaternion(self.v + (self * half).v).normalize()
    def __add__(self, o: "Quaternion") -> "Quaternion":
        return Quaternion(self.v + o.v)
    def __matmul__(self, x: Vec) -> Vec:
        """q @ v: rotate a 3-vector, (q (x) (v,0) (x) q^-1).xyz with q^-1 = conj / |q|^2 (quaternion.rs:283-305), written
        as v + w t + u x t with t = (2 / |q|^2) u x v — the 1 / |q|^2 is dropped only for quaternions known to be unit."""
        if isinstance(x, SpatialForce):                    # q @ SpatialForce: both halves (spatial.rs:571-593)
            return SpatialForce(torque=self @ x.torque(), linear=self @ x.force())
        if isinstance(x, SpatialMotion):
            return SpatialMotion(self @ x.angular(), self @ x.linear())
        x = _host(x) if _is_host_array(x) else (Vec(list(x)) if isinstance(x, (list, tuple)) else x)
        u = Vec(self.v.e[:3])
        t = np.cross(u, x) * (2.0 if self.unit else 2.0 / np.dot(self.v, self.v))
        r = x + t * self.v[3] + np.cross(u, t)
        return RotVec(r.e, x) if self.stage_attitude else r
    def __rmul__(self, o) -> "Quaternion":                # host quaternion * traced quaternion
        return Quaternion._of(o) * self
    def __mul__(self, o: "Quaternion") -> "Quaternion":
        o = Quaternion._of(o)
        l, r = self.v, o.v
        return Quaternion(Vec([l[3] * r[0] + l[0] * r[3] + l[1] * r[2] - l[2] * r[1],
                               l[3] * r[1] - l[0] * r[2] + l[1] * r[3] + l[2] * r[0],
                               l[3] * r[2] + l[0] * r[1] - l[1] * r[0] + l[2] * r[3],
                               l[3] * r[3] - l[0] * r[0] - l[1] * r[1] - l[2] * r[2]]), unit=self.unit and o.unit)


class SpatialTransform:
    def __init__(self, q: Optional[Quaternion] = None, p: Optional[Vec] = None, *, angular=None, linear=None):
        """el.SpatialTransform(angular=identity, linear=0) (libs/nox-py/src/spatial.rs:21-44), also positional (q, p)."""
        q = angular if q is None else q
        p = linear if p is None else p
        self._q = q if q is not None else Quaternion(Vec([0.0, 0.0, 0.0, 1.0]))
        self._p = p if p is not None else Vec([0.0, 0.0, 0.0])
    def angular(self): return self._q
    def linear(self): return self._p
    def __add__(self, m: "SpatialMotion") -> "SpatialTransform":
        """SpatialTransform + SpatialMotion (spatial.rs:530-549): q' = normalize(q + (w/2, 0) (x) q), p' = p + v."""
        half = Quaternion(np.concatenate([m.angular() / 2.0, 0.0]))
        return SpatialTransform(Quaternion(self._q.v + (half * self._q).v).normalize(), self._p + m.linear())


class SpatialMotion:
    def __init__(self, ang: Optional[Vec] = None, lin: Optional[Vec] = None, *, angular=None, linear=None):
        """el.SpatialMotion(angular=0, linear=0) (libs/nox-py/src/spatial.rs:121-140), also positional (ang, lin)."""
        ang = angular if ang is None else ang
        lin = linear if lin is None else lin
        self._a = ang if ang is not None else Vec([0.0, 0.0, 0.0])
        self._l = lin if lin is not None else Vec([0.0, 0.0, 0.0])
    def angular(self): return self._a
    def linear(self): return self._l
    def __add__(self, o: "SpatialMotion"): return SpatialMotion(self._a + o._a, self._l + o._l)
    def __mul__(self, k): return SpatialMotion(self._a * k, self._l * k)
    __rmul__ = __mul__


class SpatialInertia:
    def __init__(self, diag: Vec, mass: Expr): self._d, self._m = diag, mass
    def mass(self): return self._m
    def inertia_diag(self): return self._d


class SpatialForce:
    """[torque, force], world frame.  A torque given as `pos.angular() @ x` is additionally tracked in the body
    frame (`_tb`), so the generated kernel can skip the world-frame round trip."""

    def __init__(self, torque: Optional[Vec] = None, linear: Optional[Vec] = None, _tw=None, _tb=None):
        zero = Vec([0.0, 0.0, 0.0])
        self._f = linear if linear is not None else zero
        if _tw is not None or _tb is not None:
            self._tw, self._tb = (_tw if _tw is not None else zero), (_tb if _tb is not None else zero)
        elif isinstance(torque, RotVec):
            self._tw, self._tb = zero, torque.body
        else:
            self._tw, self._tb = (torque if torque is not None else zero), zero
        self._q = None   # stage attitude, set by the tracer when a body-frame part exists
    def torque(self):
        if all(e.is_const(0.0) for e in self._tb.e):
            return self._tw
        if self._q is None:
            raise TypeError("reading back a body-frame torque needs the stage attitude")
        return self._tw + Vec((self._q @ self._tb).e)
    def force(self): return self._f
    def __add__(self, o: "SpatialForce"):
        r = SpatialForce(linear=self._f + o._f, _tw=self._tw + o._tw, _tb=self._tb + o._tb)
        r._q = self._q or o._q
        return r


# ---- tracing -----------------------------------------------------------------------------------------------------

_BODY_NAMES = {"force", "pos", "world_pos", "vel", "world_vel", "inertia", "tick", "accel", "world_accel"}


class Effector:
    def __init__(self, fn: Callable, widths: Optional[Dict[str, int]] = None):
        self.fn = fn
        self.params = list(inspect.signature(fn).parameters)
        self.widths = dict(widths or {})
        self.__name__ = getattr(fn, "__name__", "effector")

    def __or__(self, other): return pipe(self, other)


def effector(fn=None, **widths):
    """Decorator. Keyword arguments give the row width of component columns the function reads
    (`@dsl.effector(thrust=1, rcs_torque=3)`); otherwise the width comes from the bound column (World.build)
    or defaults to 3."""
    if fn is None:
        return lambda f: Effector(f, widths)
    return Effector(fn, widths)


def leaf(name: str) -> Expr:
    return Expr("leaf", (), None, name)


def _body_symbols(stage: bool = False):
    q = Quaternion(Vec([leaf(f"q{c}") for c in "ijkw"]), stage_attitude=stage)
    pos = SpatialTransform(q, Vec([leaf(f"p{c}") for c in "xyz"]))
    vel = SpatialMotion(Vec([leaf(f"w{c}") for c in "xyz"]), Vec([leaf(f"v{c}") for c in "xyz"]))
    inertia = SpatialInertia(Vec([leaf(f"I{c}") for c in "xyz"]), leaf("mass"))
    return pos, vel, inertia


def _leaves_of(outputs: Sequence[Expr]) -> set:
    deps, seen = set(), set()

    def walk(e: Expr):
        if id(e) in seen:
            return
        seen.add(id(e))
        if e.op == "leaf":
            deps.add(e.name)
        if e.op == "while":              # free variables of the loop's condition and body, minus the carried ones
            names, cond, body = e.value[:3]
            inner = _leaves_of([cond, *body]) - set(names)
            deps.update(inner)
        for a in e.args:
            walk(a)
    for o in outputs:
        walk(o)
    return deps


class Window:
    """A WIDE component — `rows` x `width` values per entity, e.g. the 4 s sample buffer of the reference's rocket example
    (`el.ComponentType(F64, (480, 3))`, examples/rocket/main.py:91-98) — that stays in HBM instead of the register file.
    Declared by giving the system a (rows, width) tuple for the component: `@dsl.system(v_rel_accel_buffer=(480, 3))`.

    The generated kernel keeps it as a RING in the component's own column (`[n, rows*width]`, the reference's row layout) with
    a per-entity head (hidden 1-wide component `<name>#head` = physical index of the oldest row): `push` overwrites the oldest
    row and advances the head — 3 stores per tick where `concatenate((buffer[1:], row))` moves the whole buffer; the host
    un-rotates on download (HipExec.column), so callers always see the reference's order, oldest row first.

    Reads: `w[i]` (static row, negative from the end) -> Vec of `width`; `w.row(i)` with a traced index (inside loops);
    `w.scan(f, init, start, stop)` = jax.lax.scan over rows start..stop-1 as a REAL loop (carry only — what an IIR filter over
    the window needs); `w.push(row)` -> the value to return for the component."""

    def __init__(self, name: str, slot: int, rows: int, width: int, head: "Expr", version: int):
        self.name, self.slot, self.rows, self.width, self.head, self.version = name, slot, rows, width, head, version

    def __len__(self): return self.rows

    def row(self, index) -> "Vec":
        idx = _lift(index)
        return Vec([Expr("wload", (self.head, idx), (self.slot, self.rows, self.width, j, self.version)) for j in range(self.width)])

    def __getitem__(self, i):
        if isinstance(i, slice):              # `buffer[1:]`, `signal[2:]`, `signal[0:1]` of a reference script: rows, not yet read
            start, stop, step = i.indices(self.rows)
            if step != 1:
                raise TypeError("window[a:b:c]: only unit steps")
            return WindowSlice(self, start, max(start, stop))
        if not isinstance(i, int):
            raise TypeError("window[i] takes a Python int (use .row(index) for a traced index, .scan(...) for a loop)")
        if not -self.rows <= i < self.rows:
            raise IndexError(i)
        return self.row(float(i % self.rows))

    @property
    def shape(self): return (self.rows, self.width)

    def push(self, row) -> "WindowPush":
        row = row if isinstance(row, Vec) else Vec([row])
        if len(row) != self.width:
            raise ValueError(f"window {self.name}: a row has {self.width} values, got {len(row)}")
        return WindowPush(self, row)

    def scan(self, f, init, start: int = 0, stop: Optional[int] = None):
        """carry = init; for r in range(start, stop): carry, _ = f(carry, window[r]) -> carry.  A real loop in the kernel."""
        stop = self.rows if stop is None else int(stop)
        start = int(start)
        if not 0 <= start <= stop <= self.rows:
            raise IndexError((start, stop))
        flat, rebuild = _flatten(init)

        def cond(c):
            return c[0] < float(stop)

        def body(c):
            out = f(rebuild(list(c[1:])), self.row(c[0]))
            new_carry = out[0] if isinstance(out, tuple) and len(out) == 2 else out
            nf, _ = _flatten(new_carry)
            if len(nf) != len(flat):
                raise TypeError("window.scan: the step must return (carry, y) with the carry structure of init")
            return [c[0] + 1.0] + list(nf)
        res = _Lax.while_loop(cond, body, [const(float(start))] + list(flat), max_iter=stop - start + 1, counted=(start, stop))
        return rebuild(list(res[1:]))


class WindowPush:
    """What a system returns for a window component after `window.push(row)`."""

    def __init__(self, window: Window, row: "Vec"):
        self.window, self.row = window, row


class WindowSlice:
    """`window[a:b]`: rows a..b-1 of a window, not read yet.  What a reference script does with it decides what is generated:
    `concatenate((buffer[1:], row))` is a push (Window.push); `lax.scan(f, init, signal[2:])` a loop over those rows in the
    kernel (Window.scan) whose stacked outputs stay lazy (LazyRows); indexing reads rows."""

    def __init__(self, window: Window, start: int, stop: int):
        self.window, self.start, self.stop = window, int(start), int(stop)

    def __len__(self): return self.stop - self.start

    @property
    def shape(self): return (len(self), self.window.width)

    def __getitem__(self, i):
        if isinstance(i, slice):
            a, b, step = i.indices(len(self))
            if step != 1:
                raise TypeError("window[a:b][c:d:e]: only unit steps")
            return WindowSlice(self.window, self.start + a, self.start + max(a, b))
        if not -len(self) <= i < len(self):
            raise IndexError(i)
        return self.window[self.start + (i % len(self))]


class LazyRows:
    """The stacked per-step outputs of a scan over a window (478 rows of the rocket example's filtered signal,
    examples/rocket/main.py:187-195), of which a script uses the ends: a sequence of `parts` — explicit rows, or (first row,
    last row, count) of a scan's outputs.  `rows[-1]`, `rows[0]`, `rows[0:1]`, `rows[-1:]`, len, and concatenation are what exist;
    anything in the middle of a scan's outputs was never computed and says so."""

    def __init__(self, parts): self.parts = [p for p in parts if (p[2] if isinstance(p, tuple) else len(p)) > 0]

    @staticmethod
    def of_scan(first, last, count): return LazyRows([(first, last, int(count))])

    def __len__(self): return sum(p[2] if isinstance(p, tuple) else len(p) for p in self.parts)

    def _at(self, i):
        n = len(self)
        if not -n <= i < n:
            raise IndexError(i)
        i %= n
        for p in self.parts:
            m = p[2] if isinstance(p, tuple) else len(p)
            if i < m:
                if not isinstance(p, tuple):
                    return p[i]
                if i == 0:
                    return p[0]
                if i == m - 1:
                    return p[1]
                raise NotImplementedError("only the first and the last output of a scan over a window are available")
            i -= m
        raise IndexError(i)

    def __getitem__(self, i):
        if isinstance(i, slice):
            a, b, step = i.indices(len(self))
            if step != 1:
                raise TypeError("only unit steps")
            if (a, b) == (0, len(self)):
                return self
            return LazyRows([[self._at(k) for k in range(a, b)]]) if b - a <= 2 else self._middle(a, b)
        return self._at(i)

    def _middle(self, a, b):
        raise NotImplementedError("a slice of more than two rows out of a scan's stacked outputs is not provided")


# A component declared with a 2-D shape is a MATRIX held in registers (dsl_mat.Mat: a 3 x 3 / 6 x 6 filter covariance,
# examples/linalg/sim.py:33-58) when it has at most _MAT_AUTO_ELEMS values, otherwise a WINDOW kept in HBM (dsl.Window: the
# rocket example's 480 x 3 sample buffer).  `dsl.matrix(r, c)` / `dsl.window(r, c)` as the declared shape say it explicitly
# (a register matrix may have up to _MAT_MAX_ELEMS values).
_MAT_AUTO_ELEMS = 36
_MAT_MAX_ELEMS = 64


class matrix(tuple):
    """`@dsl.system(p=dsl.matrix(8, 8))`: a 2-D component as a register matrix whatever its size (<= 64 values)."""
    def __new__(cls, rows, cols): return tuple.__new__(cls, (int(rows), int(cols)))


class window(tuple):
    """`@dsl.system(buf=dsl.window(16, 3))`: a 2-D component as a memory-resident window whatever its size."""
    def __new__(cls, rows, cols): return tuple.__new__(cls, (int(rows), int(cols)))


def _is_matrix_shape(d) -> bool:
    if isinstance(d, window) or not isinstance(d, (tuple, list)) or len(d) != 2:
        return False
    n = int(d[0]) * int(d[1])
    if isinstance(d, matrix):
        if n > _MAT_MAX_ELEMS:
            raise ValueError(f"a register matrix holds at most {_MAT_MAX_ELEMS} values, got {tuple(d)}")
        return True
    return n <= _MAT_AUTO_ELEMS


class ColumnTable:
    """Component columns used by generated code, in first-use order: name -> (slot, width)."""

    def __init__(self, prefix: str, limit: int, max_width: int, known: Optional[Dict[str, int]] = None):
        self.prefix, self.limit, self.max_width = prefix, limit, max_width
        self.known = dict(known or {})
        self.cols: List[Tuple[str, int]] = []
        self.windows: Dict[str, List[int]] = {}      # name -> [slot, rows, width, version]
        self.mats: Dict[str, Tuple[int, int]] = {}    # name -> (rows, cols) of a small 2-D component held in registers

    def window(self, name: str, rows: int, width: int) -> Window:
        """The memory-resident component `name` ([rows, width] per entity) at its current version, plus its hidden head."""
        rows, width = int(rows), int(width)
        if rows < 2 or width < 1:
            raise ValueError(f"window {name}: need rows >= 2 and width >= 1")
        have = dict(self.cols)
        if name in have and name not in self.windows:
            raise ValueError(f"component {name} is already used as a register column")
        if name not in self.windows:
            if len(self.cols) >= self.limit:
                raise ValueError(f"generated code can use at most {self.limit} component columns")
            self.cols.append((name, rows * width))
            self.windows[name] = [len(self.cols) - 1, rows, width, 0]
        slot, r0, w0, version = self.windows[name]
        if (r0, w0) != (rows, width):
            raise ValueError(f"window {name}: conflicting shapes {(r0, w0)} / {(rows, width)}")
        head = self.symbols(name + "#head", 1, 1)[0]
        return Window(name, slot, rows, width, head, version)

    def bump(self, name: str):
        self.windows[name][3] += 1

    def matrix(self, name: str, rows: int, cols: int):
        """A SMALL 2-D component (`el.ComponentType(F64, (3, 3))`: a filter covariance, examples/linalg/sim.py:33-36): a
        register column of rows * cols values in the reference's row-major order, handed to the system as a dsl_mat.Mat."""
        from . import dsl_mat
        rows, cols = int(rows), int(cols)
        have = self.mats.get(name)
        if have is not None and have != (rows, cols):
            raise ValueError(f"matrix component {name}: conflicting shapes {have} / {(rows, cols)}")
        self.mats[name] = (rows, cols)
        return dsl_mat.reshape(self.symbols(name, rows * cols, rows * cols), (rows, cols))

    def symbols(self, name: str, declared: Optional[int], default: int) -> Vec:
        if name in self.windows:
            raise ValueError(f"component {name} is a window: give the system its (rows, width) shape")
        w = int(declared if declared is not None else self.known.get(name, default))
        if not 1 <= w <= self.max_width:
            raise ValueError(f"component {name}: width must be 1..{self.max_width} (a wider one is a window: declare its "
                             "(rows, width) shape)")
        have = dict(self.cols)
        if name in have and have[name] != w:
            raise ValueError(f"component {name}: conflicting widths {have[name]} / {w}")
        if name not in have:
            if len(self.cols) >= self.limit:
                raise ValueError(f"generated code can use at most {self.limit} component columns")
            self.cols.append((name, w))
        slot = [c for c, _ in self.cols].index(name)
        return Vec([leaf(f"{self.prefix}{slot}_{k}") for k in range(w)])


class TracedPipe:
    """Result of tracing an effector pipe: output wrench nodes, the component columns read, dependency flags."""

    def __init__(self, effectors: Sequence[Effector], table: Optional[ColumnTable] = None,
                 widths: Optional[Dict[str, int]] = None, partial: Sequence[str] = ()):
        self.effectors = list(effectors)
        self.table = table or ColumnTable("aux", 4, 3, widths)
        pos, vel, inertia = _body_symbols(stage=True)
        force = SpatialForce()                          # clear_forces: the pipe starts from zero (six_dof.rs:148-150)
        force._q = pos.angular()
        self.leaves_upto = []
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
                    kwargs[name] = self.table.symbols(name, eff.widths.get(name), 3)
            with tracing(self.table if self.table.prefix == "c" else None):      # a program's table holds `mc:` parameter columns too
                out = eff.fn(**kwargs)
            if not isinstance(out, SpatialForce):
                raise TypeError(f"effector {eff.__name__} must return a dsl.SpatialForce")
            if out._q is None:
                out._q = pos.angular()
            # an effector is a map over ITS query (query.rs:136-208): an entity lacking one of the components it reads
            # keeps the force its predecessors left.  Such columns arrive densified with a presence column `has:<name>`.
            need = [n for n in eff.params if n in partial]
            if need:
                mask = None
                for n in dict.fromkeys(need):
                    h = self.table.symbols("has:" + n, 1, 1)[0] > 0.5
                    mask = h if mask is None else (mask & h)
                pick = lambda a, b: Vec([Expr("select", (mask, _lift(x), _lift(y))) for x, y in zip(a.e, b.e)])
                kept = SpatialForce(linear=pick(out._f, force._f), _tw=pick(out._tw, force._tw), _tb=pick(out._tb, force._tb))
                kept._q = out._q
                out = kept
            force = out
            if getattr(eff, "pipe_index", None) is not None:      # what the force depends on up to here (TracedProgram's order check)
                self.leaves_upto.append((eff.pipe_index, _leaves_of(list(force._tw.e) + list(force._f.e) + list(force._tb.e))))
        self.torque_world, self.torque_body, self.linear = force._tw, force._tb, force.force()
        # outputs: world torque (3), force (3), body-frame torque (3)
        self.outputs: List[Expr] = list(self.torque_world.e) + list(self.linear.e) + list(self.torque_body.e)
        self.leaves = _leaves_of(self.outputs)
        self.reads_velocity = any(n[0] in "wv" and len(n) == 2 for n in self.leaves)
        self.world_torque = not all(t.is_const(0.0) for t in self.torque_world.e)
        self.body_torque = not all(t.is_const(0.0) for t in self.torque_body.e)

    @property
    def columns(self) -> List[Tuple[str, int]]:
        return self.table.cols


def pipe(*effectors: Effector) -> "Pipe":
    flat: List[Effector] = []
    for e in effectors:
        flat.extend(e.effectors if isinstance(e, Pipe) else [e])
    if any(isinstance(e, (System, Stages, GraphFold)) for e in flat):
        # `gravity | drag | motor_response | apply_forces` (examples/drone/sim.py:193): force effectors piped with maps that
        # write plain components.  Kept as a flat stage list; six_dof(sys=...) splits it (frontend.six_dof)
        items = []
        for e in flat:
            items.extend(e.items if isinstance(e, Stages) else [e])
        return Stages(items)
    for e in flat:
        if not isinstance(e, Effector):
            raise TypeError(f"an effector pipe holds functions returning a force; {getattr(e, '__name__', e)!r} is a "
                            f"{type(e).__name__} (pipe it around six_dof instead)")
    return Pipe(flat)


class Pipe:
    """An ordered effector pipe (`a | b | c`), traced lazily."""

    def __init__(self, effectors: Sequence[Effector]):
        self.effectors = list(effectors)
        self._traced: Optional[TracedPipe] = None

    def __or__(self, other): return pipe(self, other)

    def trace(self, widths: Optional[Dict[str, int]] = None) -> TracedPipe:
        if self._traced is None:
            Expr.fresh()
            self._traced = TracedPipe(self.effectors, widths=widths)
        return self._traced


# ---- GraphQuery.edge_fold with a user-written fold function ------------------------------------------------------

class _EdgeTransform:
    """WorldPos of one edge endpoint.  The pair kernels stage the three integrator-stage positions of every body
    (csrc/pair_kernel.hpp: pack), not attitudes, so only `.linear()` is available inside a fold."""

    def __init__(self, p: Vec): self._p = p
    def linear(self): return self._p
    def angular(self): raise TypeError("edge_fold functions can read pos.linear() only")


class _EdgeInertia:
    def __init__(self, m: Expr): self._m = m
    def mass(self): return self._m
    def inertia_diag(self): raise TypeError("edge_fold functions can read inertia.mass() only")


class EdgeFold:
    """`graph.edge_fold(query, query, el.Force, el.SpatialForce(), fn)` (graph.rs:177-282, used by
    examples/three-body/main.py:56-78 and examples/n-body/sim.py:344-369) with left/right queries
    `Query[WorldPos, Inertia]`: fn(acc, a_pos, a_inertia, b_pos, b_inertia) -> SpatialForce, folded from zero over
    each source's out-edges in spawn order; the result REPLACES Force on source rows."""

    def __init__(self, fn: Callable, edge_component: str = "gravity_edge"):
        self.fn = fn
        self.edge_component = edge_component
        self.__name__ = getattr(fn, "__name__", "edge_fold")
        if len(inspect.signature(fn).parameters) != 5:
            raise TypeError("edge_fold function must take (acc, a_pos, a_inertia, b_pos, b_inertia)")
        self._traced = None

    def trace(self) -> "TracedFold":
        if self._traced is None:
            Expr.fresh()
            self._traced = TracedFold(self)
        return self._traced


def edge_fold(fn=None, edge_component: str = "gravity_edge"):
    if fn is None:
        return lambda f: EdgeFold(f, edge_component)
    return EdgeFold(fn, edge_component)


class TracedFold:
    LEAVES = ["acc0", "acc1", "acc2", "acc3", "acc4", "acc5", "ax", "ay", "az", "ma", "bx", "by", "bz", "mb"]

    def __init__(self, fold: EdgeFold):
        self.fold = fold
        acc = SpatialForce(torque=Vec([leaf(f"acc{k}") for k in range(3)]), linear=Vec([leaf(f"acc{k}") for k in range(3, 6)]))
        a = (_EdgeTransform(Vec([leaf("a" + c) for c in "xyz"])), _EdgeInertia(leaf("ma")))
        b = (_EdgeTransform(Vec([leaf("b" + c) for c in "xyz"])), _EdgeInertia(leaf("mb")))
        with tracing():
            out = fold.fn(acc, a[0], a[1], b[0], b[1])
        if not isinstance(out, SpatialForce):
            raise TypeError(f"edge_fold function {fold.__name__} must return a dsl.SpatialForce")
        if not all(e.is_const(0.0) for e in out._tb.e):
            raise TypeError("edge_fold functions cannot produce body-frame torques")
        self.outputs: List[Expr] = list(out._tw.e) + list(out.force().e)   # new acc: [tau(3), f(3)]
        self.leaves = _leaves_of(self.outputs)


class GraphFold:
    """`graph.edge_fold(left_query, right_query, return_type, init_value, fn)` as a STAND-ALONE system over arbitrary
    components (libs/nox-py/python/elodin/__init__.py:454-557, e.g. test_all.py:117-142): for every entity with out-edges,
    acc = init; for each out-edge in spawn order: acc = fn(acc, *left components of the source, *right components of the
    target); the result replaces the `out` component on source rows.  All folds read the component values from BEFORE the
    system ran (the reference's arrays are immutable)."""

    def __init__(self, fn: Callable, edge_component: str, left: Sequence[str], right: Sequence[str], out: str, init):
        self.fn, self.edge_component = fn, edge_component
        self.left, self.right, self.out = tuple(left), tuple(right), out
        self.init = tuple(float(v) for v in (init if isinstance(init, (list, tuple)) else [init]))
        self.__name__ = getattr(fn, "__name__", "graph_fold")
        if len(inspect.signature(fn).parameters) != 1 + len(self.left) + len(self.right):
            raise TypeError("fold function must take (acc, *left components, *right components)")

    def trace(self, widths: Dict[str, int]) -> "TracedGraphFold":
        Expr.fresh()
        return TracedGraphFold(self, widths)


def graph_fold(edge_component: str, left: Sequence[str], right: Sequence[str], out: str, init=0.0):
    return lambda fn: GraphFold(fn, edge_component, left, right, out, init)


class TracedGraphFold:
    def __init__(self, fold: GraphFold, widths: Dict[str, int]):
        self.fold = fold
        self.widths = {n: int(widths[n]) for n in dict.fromkeys(fold.left + fold.right + (fold.out,))}
        w_out = self.widths[fold.out]
        if len(fold.init) != w_out:
            raise ValueError(f"init has {len(fold.init)} values, component {fold.out} has width {w_out}")
        sym = lambda prefix, w: (lambda v: v if w > 1 else v[0])(Vec([leaf(f"{prefix}_{k}") for k in range(w)]))
        acc = sym("acc", w_out)
        args = [sym(f"a{i}", self.widths[n]) for i, n in enumerate(fold.left)]
        args += [sym(f"b{i}", self.widths[n]) for i, n in enumerate(fold.right)]
        with tracing():
            out = fold.fn(acc, *args)
        out = out if isinstance(out, Vec) else Vec([out])
        if len(out) != w_out:
            raise ValueError(f"fold function returned {len(out)} values for component {fold.out} of width {w_out}")
        self.outputs: List[Expr] = list(out.e)


# ---- systems piped around six_dof ----------------------------------------------------------------------------------

class System:
    """A per-entity system outside six_dof (`@el.map` in the reference, e.g. apollo-lander/sim.py:334-378,400-431):
    reads components by parameter name (plus `pos`, `vel`, `inertia`, `tick`), returns {component: new value}.
    `every=n` runs it only on ticks with `tick % n == phase` (wave-uniform branch; `tick` counts completed ticks, 1-based) and,
    if `also_at` is given, on that tick too.  That is how the reference's post_step cadence is written as a system: the
    server loop runs `ticks_per_telemetry` ticks per batch and calls post_step(end_tick = ticks completed - 1) after each
    batch, cutting the last batch at max_ticks (impeller2_server.rs:553-678) -> `every=ticks_per_telemetry, also_at=max_ticks`,
    and a `tick % 5 == 0` test inside post_step becomes `every=15, phase=6` for a 3-tick batch."""

    def __init__(self, fn: Callable, widths: Optional[Dict[str, int]] = None, every: int = 1, singletons: Sequence[str] = (),
                 phase: int = 0, also_at: Optional[int] = None):
        self.fn = fn
        self.params = list(inspect.signature(fn).parameters)
        self.widths = dict(widths or {})
        self.every = int(every)
        self.phase = int(phase) % max(self.every, 1)
        self.also_at = None if also_at is None else int(also_at)
        # components queried on their own (`s: el.Query[el.Seed]` ... `s[0]`): a one-entity column every row may read
        self.singletons = tuple(singletons)
        self.__name__ = getattr(fn, "__name__", "system")


    def __or__(self, other): return Stages([self]) | other
    def __ror__(self, other): return Stages([other]) | self


class Stages:
    """`a | b | six_dof(...) | c`: the reference's system pipe (system.rs:1001-1011), kept as a flat list."""

    def __init__(self, items): self.items = list(items)
    def __or__(self, other):
        return Stages(self.items + (other.items if isinstance(other, Stages) else [other]))
    def __ror__(self, other):
        return Stages((other.items if isinstance(other, Stages) else [other]) + self.items)


def system(fn=None, every: int = 1, singletons: Sequence[str] = (), phase: int = 0, also_at: Optional[int] = None, **widths):
    if fn is None:
        return lambda f: System(f, widths, every, singletons, phase, also_at)
    return System(fn, widths, every, singletons, phase, also_at)


class TracedSystem:
    """One system as (target, expression) assignments over the register file / body state."""

    def __init__(self, sys_: System, table: ColumnTable, partial: Sequence[str] = (), after_six_dof: bool = False):
        self.name, self.every, self.phase, self.also_at = sys_.__name__, sys_.every, sys_.phase, sys_.also_at
        self.reads_accel = False
        self.body_free = bool(getattr(sys_, "body_free", False))     # the system declares it touches no Body column (checked by codegen)
        self.float32_refused = list(getattr(sys_, "float32_refused", ()) or ())     # stablehlo.float32_hazards: integer work exact in f64 only
        self.fp_contract = bool(getattr(sys_, "fp_contract", False))     # traced under relaxed_arithmetic: the build may contract a * b + c
        self.uniform = tuple(getattr(sys_, "uniform", ()) or ())        # columns the caller promises hold ONE value in every row
        pos, vel, inertia = _body_symbols()
        kwargs = {}

        # `pos` / `vel` / `accel` are this DSL's short spellings of world_pos / world_vel / world_accel; systems lowered from the
        # reference's decorator surface (frontend) name components exactly, and a user component may be called `accel`
        # (examples/drone/sensors.py: the IMU's accelerometer)
        short = getattr(sys_, "aliases", True)
        POS = ("pos", "world_pos") if short else ("world_pos",)
        VEL = ("vel", "world_vel") if short else ("world_vel",)
        ACC = ("accel", "world_accel") if short else ("world_accel",)

        def declared(name):
            return sys_.widths.get(name, table.known.get(name))

        def shape_of(name):
            d = declared(name)
            return tuple(int(x) for x in d) if isinstance(d, (tuple, list)) else None
        for name in sys_.params:
            if name in ACC:
                # after six_dof: the acceleration it just produced; before: the column as the previous tick left it
                # (what a reference system piped in front of six_dof reads, e.g. examples/rocket/main.py:452-462)
                self.reads_accel = True
                kwargs[name] = SpatialMotion(Vec([leaf("aa" + c) for c in "xyz"]), Vec([leaf("al" + c) for c in "xyz"]))
                continue
            if name in POS:
                kwargs[name] = pos
            elif name in VEL:
                kwargs[name] = vel
            elif name == "inertia":
                kwargs[name] = inertia
            elif name == "tick":
                k = getattr(sys_, "tick_substeps", 1)
                kwargs[name] = leaf("tick") if k <= 1 else _un("floor", (leaf("tick") - 1.0) / float(k)) + 1.0
            elif name == "force":
                raise TypeError("systems outside six_dof cannot read `force`")
            elif _is_matrix_shape(declared(name)) and name not in table.windows:
                kwargs[name] = table.matrix(name, *shape_of(name))      # a small matrix: registers, not a window
            elif shape_of(name) is not None:
                if name in partial:
                    raise TypeError(f"system {self.name}: window component {name} must live on every row of the executor")
                kwargs[name] = table.window(name, *shape_of(name))
            else:
                v = table.symbols(name, sys_.widths.get(name), 1)
                kwargs[name] = v if len(v) > 1 else v[0]
        with tracing(table):
            out = sys_.fn(**kwargs)
        if not isinstance(out, dict):
            raise TypeError(f"system {self.name} must return a dict {{component: value}}")
        self.assign: List[Tuple[str, Expr]] = []      # (leaf name written, value)
        self.writes_inertia = False
        touched: List[str] = []                        # pseudo-leaves `win<slot>`: loads of a pushed window go stale
        for cname, val in out.items():
            if isinstance(val, (Window, WindowPush)):
                if isinstance(val, Window):
                    raise TypeError(f"system {self.name}: return window.push(row) for {cname} (a window cannot be rewritten whole)")
                w = val.window
                if w.name != cname or cname in partial:
                    raise TypeError(f"system {self.name}: {cname} must be returned as ITS OWN window's push, on every row")
                if w.version != table.windows[cname][3]:
                    raise TypeError(f"system {self.name}: window {cname} was pushed twice from the same handle")
                # the new row replaces the oldest (physical row `head`), then the head moves on: stores first, head last
                for j, e in enumerate(val.row.e):
                    self.assign.append((f"wst{w.slot}_{j}", e))
                nxt = w.head + 1.0
                self.assign.append((w.head.name, Expr("select", (nxt < float(w.rows), nxt, const(0.0)))))
                table.bump(cname)
                touched.append(f"win{w.slot}")
                continue
            if cname in POS:
                if not isinstance(val, SpatialTransform):
                    raise TypeError("world_pos must be a dsl.SpatialTransform")
                for c, e in zip("ijkw", val.angular().vector().e):
                    self.assign.append((f"q{c}", e))
                for c, e in zip("xyz", val.linear().e):
                    self.assign.append((f"p{c}", e))
            elif cname in VEL:
                for c, e in zip("xyz", val.angular().e):
                    self.assign.append((f"w{c}", e))
                for c, e in zip("xyz", val.linear().e):
                    self.assign.append((f"v{c}", e))
            elif cname == "inertia":
                self.writes_inertia = True
                for c, e in zip("xyz", val.inertia_diag().e):
                    self.assign.append((f"I{c}", e))
                self.assign.append(("mass", _lift(val.mass())))
            else:
                if isinstance(val, list) and val and isinstance(val[0], Vec):      # a matrix (dsl_mat.Mat / list of rows): row-major
                    table.mats.setdefault(cname, (len(val), len(val[0])))
                    val = Vec([e for row in val for e in row.e])
                v = val if isinstance(val, Vec) else Vec([val])
                declared = sys_.widths.get(cname, len(v))
                if isinstance(declared, (tuple, list)):
                    declared = int(declared[0]) * int(declared[1])
                cur = table.symbols(cname, declared, len(v))
                if len(cur) != len(v):
                    raise ValueError(f"system {self.name}: component {cname} has width {len(cur)}, got {len(v)} values")
                for k, e in enumerate(v.e):
                    self.assign.append((cur[k].name, e))
        # query join (query.rs:136-208): a system runs on the entities that HAVE every component it reads or writes.
        # Components living on fewer entities than the executor's row set arrive densified with a presence column
        # `has:<name>`; the system's writes are predicated on all of them.
        need = [n for n in list(sys_.params) + list(out.keys()) if n in partial]
        # the Body columns themselves may be partial: an executor whose rows also hold non-Body entities (folds between
        # Bodies and plain entities) carries stand-in Body values on those rows, and a system that touches the Body runs on
        # Bodies only
        body_names = (_BODY_NAMES if short else _BODY_NAMES - {"pos", "vel", "accel"}) - {"tick", "force"}
        if "world_pos" in partial and any(n in body_names for n in list(sys_.params) + list(out.keys())):
            need.append("world_pos")
        if need:
            mask = None
            for n in dict.fromkeys(need):
                h = table.symbols("has:" + n, 1, 1)[0] > 0.5
                mask = h if mask is None else (mask & h)
            self.assign = [(t, Expr("select", (mask, e, leaf(t)))) for t, e in self.assign]
        self.written = [t for t, _ in self.assign] + touched


class Program:
    """`pre systems | six_dof(effectors) | post systems` — one whole tick, like the reference's compiled pipe.

    `pre` / `post` may also hold stand-alone folds (`dsl.GraphFold`, the reference's `graph.edge_fold` over arbitrary
    components, graph.rs:239-361): a fold reads OTHER entities' rows, so every lane must have finished the systems in front
    of it — the tick then runs as a chain of launches (systems | fold | systems | ... | six_dof | ...), all generated into one
    translation unit and driven by one launch call, the columns staying in HBM between the links (codegen.py).  The edges of
    each fold's edge component are given as ROW pairs of the executor's row set when tracing (`fold_edges`, spawn order;
    HipExec resolves them from entity ids) and are baked into the generated code, like the reference bakes its gather
    indices into the compiled tick."""

    def __init__(self, pre: Sequence, effectors: Pipe, post: Sequence, substeps: int = 1):
        """substeps = k > 1: the reference pipe was `pre | (six_dof | post) x k` (examples/drone/sim.py:173-208: three
        integrator sub-steps per tick).  The program then runs k executor ticks per world tick — `pre` on the first of each
        k (the caller hands them over with `every=k, phase=1`), `six_dof | post` on every one — and user code that reads the
        tick sees the WORLD tick, floor((t - 1) / k) + 1."""
        self.pre, self.effectors, self.post = list(pre), effectors, list(post)
        self.substeps = int(substeps)
        self._traced = None

    @property
    def folds(self) -> List["GraphFold"]:
        return [s for s in self.pre + self.post if isinstance(s, GraphFold)]

    def trace(self, widths: Optional[Dict[str, int]] = None, partial: Sequence[str] = (), fold_edges=None,
              fold_replicas: Optional[Tuple[int, int]] = None) -> "TracedProgram":
        """fold_replicas=(count, stride): the executor holds `count` copies of one small world, `stride` rows each (a
        Monte-Carlo batch of graph worlds); `fold_edges` then describe replica 0 only (rows < stride) and every replica folds
        over the same template, shifted by its base row — one baked CSR for the whole batch."""
        if self._traced is None:
            Expr.fresh()
            self._traced = TracedProgram(self, widths, partial, fold_edges, fold_replicas)
        return self._traced


class FrozenProgram(Program):
    """A program whose generated HIP source already exists (codegen.generate_source's text, produced where the user code
    could be traced) with the column table it was generated for — e.g. the reference's drone example, which can only be
    imported where the reference checkout is: its kernel source is committed as a fixture and run where the GPU is
    (tests/golden/make_drone_program.py).  HipExec compiles and binds it like a program it traced itself."""

    def __init__(self, source: Optional[str], columns: Sequence[Tuple[str, int]], mats: Optional[Dict[str, Tuple[int, int]]] = None,
                 substeps: int = 1, column_soa: bool = False, windows: Optional[Dict[str, Tuple[int, int, int]]] = None,
                 prebuilt_so: Optional[str] = None):
        """column_soa: the text was generated with element-major program columns (codegen.generate_source(column_soa=True));
        the executor lays the columns out the way the text expects, whatever its row count.
        prebuilt_so: the shared object itself already exists (`python -m elodin_amd.stablehlo ... -o pipe.so`): nothing is
        generated or compiled, the executor installs it with sixdof_set_custom_pipe as a host without Python would."""
        super().__init__([], Pipe([]), [], substeps=substeps)
        import types
        table = types.SimpleNamespace(mats={k: tuple(v) for k, v in (mats or {}).items()}, windows={})
        # windows: name -> (slot, rows, width) of the wide components the text keeps in HBM as rings (entity-major layout)
        self._traced = types.SimpleNamespace(frozen_source=source, prebuilt_so=prebuilt_so, columns=[(str(n), int(w)) for n, w in columns],
                                             windows={str(k): tuple(int(x) for x in v) for k, v in (windows or {}).items()},
                                             table=table, fold_stages=[], pre=[], post=[], column_soa=bool(column_soa))

    def trace(self, *a, **k):
        return self._traced


class TracedFoldStage:
    """A stand-alone fold as one link of a program's launch chain: the traced fold function, where its components live
    (program column slots, or the Body columns) and its edges as CSR over the executor's rows."""

    BODY_WIDTH = {"world_pos": 7, "world_vel": 6, "inertia": 7}

    def __init__(self, fold: GraphFold, table: ColumnTable, index: int, edges, partial: Sequence[str] = (),
                 replicas: Optional[Tuple[int, int]] = None):
        self.name, self.index = fold.__name__, index
        self.replicas = (int(replicas[0]), int(replicas[1])) if replicas else None
        names = list(dict.fromkeys(fold.left + fold.right + (fold.out,)))
        if fold.out in self.BODY_WIDTH:
            raise TypeError(f"fold {self.name}: a stand-alone fold writes a plain component (use an edge_fold effector for Force)")
        # components living on fewer entities than the row set arrive densified; the caller keeps only the edges whose
        # endpoints carry what the fold's queries name (the reference's query join, query.rs:136-208)
        widths = {}
        for n in names:
            widths[n] = self.BODY_WIDTH[n] if n in self.BODY_WIDTH else len(table.symbols(n, None, 1))
        self.traced = TracedGraphFold(fold, widths)
        self.widths = widths
        slot_of = lambda n: None if n in self.BODY_WIDTH else [c for c, _ in table.cols].index(n)
        self.left = [(n, slot_of(n), widths[n]) for n in fold.left]
        self.right = [(n, slot_of(n), widths[n]) for n in fold.right]
        self.out = (fold.out, slot_of(fold.out), widths[fold.out])
        # every fold reads the values from BEFORE it ran: results go to a scratch column first, then are committed
        self.scratch_name = f"{fold.out}#fold{index}"
        table.symbols(self.scratch_name, widths[fold.out], widths[fold.out])
        self.scratch_slot = slot_of(self.scratch_name)
        if edges is None:
            raise ValueError(f"fold {self.name}: no edges given for edge component {fold.edge_component!r} (fold_edges)")
        self.complete = 0
        if len(edges) == 2 and isinstance(edges[0], str) and edges[0] == "complete":
            # the COMPLETE graph over rows 0..n-1 of a world (every source folds every other row in ascending order — the spawn order
            # of examples/n-body/sim.py:330-338): nothing is baked, the kernel forms the target of slot s as s + (s >= source), so
            # the graph may have any number of edges
            n_ = int(edges[1])
            if self.replicas and n_ > self.replicas[1]:
                raise ValueError(f"fold {self.name}: a complete graph of {n_} rows does not fit a replica of {self.replicas[1]} rows")
            self.complete = n_
            self.src_rows = list(range(n_))
            self._n_edges = n_ * (n_ - 1)
            self.written = [f"c{self.out[1]}_{k}" for k in range(self.out[2])] + [f"c{self.scratch_slot}_{k}" for k in range(self.out[2])]
            self.every, self.phase, self.also_at, self.reads_accel, self.writes_inertia = 1, 0, None, False, False
            return
        src = [int(x) for x in edges[0]]
        dst = [int(x) for x in edges[1]]
        if len(src) != len(dst):
            raise ValueError("fold edges: from / to lengths differ")
        if self.replicas and any(not 0 <= r < self.replicas[1] for r in src + dst):
            raise ValueError(f"fold {self.name}: with fold_replicas the edges describe replica 0 (rows 0..{self.replicas[1] - 1})")
        if len(src) > 65536:
            raise ValueError(f"fold {self.name}: {len(src)} edges — folds inside a program bake their edges into the generated "
                             "code (<= 65,536); run a larger graph as a stand-alone fold (World.build(fold))")
        by_src: Dict[int, List[int]] = {}                            # spawn order kept inside a source (graph.rs:113-175)
        for a, b in zip(src, dst):
            by_src.setdefault(a, []).append(b)
        self.src_rows = sorted(by_src)
        self.row_start, self.dst = [0], []
        for r in self.src_rows:
            self.dst += by_src[r]
            self.row_start.append(len(self.dst))
        self.written = [f"c{self.out[1]}_{k}" for k in range(self.out[2])] + [f"c{self.scratch_slot}_{k}" for k in range(self.out[2])]
        self.every, self.phase, self.also_at, self.reads_accel, self.writes_inertia = 1, 0, None, False, False


    # a complete graph's CSR, made on demand (the numpy walker of the tests reads it; the generated kernel does not)
    def __getattr__(self, name):
        if name in ("row_start", "dst") and self.__dict__.get("complete"):
            n_ = self.complete
            if n_ > 1024:
                raise ValueError(f"fold {self.name}: the explicit edge list of a {n_}-row complete graph is not materialised")
            self.__dict__["row_start"] = [s_ * (n_ - 1) for s_ in range(n_ + 1)]
            self.__dict__["dst"] = [t for s_ in range(n_) for t in range(n_) if t != s_]
            return self.__dict__[name]
        raise AttributeError(name)


MAX_PROGRAM_COLUMNS = 128      # = csrc/kernels.hpp kMaxModelCols (two kernarg pointers per column)


class TracedProgram:
    def __init__(self, prog: Program, widths: Optional[Dict[str, int]] = None, partial: Sequence[str] = (), fold_edges=None,
                 fold_replicas: Optional[Tuple[int, int]] = None):
        self.table = ColumnTable("c", MAX_PROGRAM_COLUMNS, _MAT_MAX_ELEMS, widths)
        self.partial = tuple(partial)
        fold_edges = fold_edges or {}
        n_folds = [0]

        def trace_item(s, after):
            if isinstance(s, GraphFold):
                n_folds[0] += 1
                return TracedFoldStage(s, self.table, n_folds[0] - 1, fold_edges.get(s.edge_component), self.partial, fold_replicas)
            return TracedSystem(s, self.table, self.partial, after_six_dof=after)
        self.pre = [trace_item(s, False) for s in prog.pre]
        self.pipe = TracedPipe(prog.effectors.effectors, table=self.table, partial=self.partial)
        self.post = [trace_item(s, True) for s in prog.post]
        self.fold_stages = [s for s in self.pre + self.post if isinstance(s, TracedFoldStage)]
        # every system declares itself free of Body state and nothing else is in the pipe: a systems-only executor of this program
        # may leave the Body slabs alone (NoModel::kBodyDead, csrc/effectors.hpp)
        self.body_free = (bool(self.pre + self.post) and not self.fold_stages and not prog.effectors.effectors
                          and all(getattr(s, "body_free", False) for s in self.pre + self.post))
        self.float32_refused = [r for s in self.pre + self.post for r in getattr(s, "float32_refused", ())]
        # relaxed arithmetic is a property of the whole object (one contraction pragma, one reciprocal flavour): a program either
        # holds only systems traced under relaxed_arithmetic or none (fold stages follow their systems)
        relaxed_ = [bool(getattr(s, "fp_contract", False)) for s in self.pre + self.post if not isinstance(s, TracedFoldStage)]
        if any(relaxed_) and not all(relaxed_):
            raise ValueError("a program mixes systems traced under relaxed arithmetic with reference-arithmetic systems: build them as separate "
                             "programs, or trace all of them the same way")
        self.fp_contract = any(relaxed_)
        self.writes_inertia = any(s.writes_inertia for s in self.pre + self.post)
        # maps / folds that stood among the force effectors inside six_dof(sys=...) run in front of the force evaluation
        # (frontend.six_dof).  That is the reference's order unless an effector reads what a system BEHIND it in the pipe writes
        # (it would see the old value there, the new one here)
        for s, ts in zip(prog.pre, self.pre):
            j = getattr(s, "pipe_index", None)
            for i, leaves in self.pipe.leaves_upto if j is not None else ():
                clash = set(ts.written) & leaves if j > i else None
                if clash:
                    cols = sorted({self.table.cols[int(t[1:].split("_")[0])][0] for t in clash if t[0] == "c"})
                    raise NotImplementedError(f"six_dof(sys=...): {ts.name} writes {cols}, which a force effector in front of it reads")
        # world_accel read in front of six_dof = the previous tick's; a system BEHIND a post fold runs in a launch of its own
        # and finds this tick's there, loaded the same way
        seen_fold = False
        behind = []
        for s in self.post:
            seen_fold = seen_fold or isinstance(s, TracedFoldStage)
            if seen_fold and not isinstance(s, TracedFoldStage):
                behind.append(s)
        self.pre_reads_accel = any(s.reads_accel for s in self.pre) or any(s.reads_accel for s in behind)
        self.windows = {name: tuple(v[:3]) for name, v in self.table.windows.items()}     # name -> (slot, rows, width)
        if self.fold_stages and self.windows:
            raise TypeError("a program with stand-alone folds cannot also hold window components")
        written = set()
        for s in self.pre + self.post:
            written.update(t for t in s.written if t[0] == "c")
        self.written_slots = sorted({int(t[1:].split("_")[0]) for t in written})
        self.reads_velocity = self.pipe.reads_velocity
        self.world_torque = self.pipe.world_torque
        # WAVE-UNIFORM columns (System.uniform; stablehlo.world_system(one_world=True)): the caller promises that every row of the
        # executor holds the same value there (the Globals of ONE world, replicated per row) — the kernel then reads the first row of
        # its wavefront's block instead of 64 rows (codegen._emit_pipe_struct).  Only when every system of a one-kernel program agrees.
        systems_ = self.pre + self.post
        declared = [set(getattr(s, "uniform", ()) or ()) for s in systems_]
        names_ = [c for c, _ in self.table.cols]
        self.uniform_slots = (sorted(names_.index(c) for c in set.intersection(*declared) if c in names_)
                              if declared and not self.fold_stages and not self.windows else [])

    @property
    def columns(self) -> List[Tuple[str, int]]:
        return self.table.cols
