"""The random-program generator of the fuzz tests, and an INDEPENDENT evaluation of what it generates.

`Gen(seed, backend)` draws expressions over a backend namespace.  With TRACED (elodin_amd.dsl) they are nodes of a program for
the tracer / code generator / kernel; with TWIN they are plain numpy closures built by this file alone — nothing of
elodin_amd is involved in evaluating them.  The same seed draws the same expression in both (the random choices do not depend
on the backend), so `twin_program(seed)` is the numpy twin of `make_program(seed)`: a bug in the tracer (elodin_amd/dsl.py) —
which the walker of the traced DAG (tests/dsl_numpy.py) inherits — makes the two disagree
(tests/test_fuzz_twin.py::test_a_seeded_tracer_bug_is_caught)."""
import types

import numpy as np
import scipy.special

from elodin_amd import dsl

TRACED = types.SimpleNamespace(np=dsl.np, lax=dsl.lax, Vec=dsl.Vec, const=dsl.const)


# ---- the numpy twin backend: values are closures env -> ndarray ------------------------------------------------------------

class N:
    """One value of the twin: `f(env)` -> float64 array [rows] (or bool array for comparisons)."""
    __array_ufunc__ = None

    def __init__(self, f): self.f = f
    def __call__(self, env): return self.f(env)
    def __add__(self, o): return _b(self, o, np.add)
    def __radd__(self, o): return _b(o, self, np.add)
    def __sub__(self, o): return _b(self, o, np.subtract)
    def __rsub__(self, o): return _b(o, self, np.subtract)
    def __mul__(self, o): return _b(self, o, np.multiply)
    def __rmul__(self, o): return _b(o, self, np.multiply)
    def __truediv__(self, o): return _b(self, o, np.divide)
    def __rtruediv__(self, o): return _b(o, self, np.divide)
    def __neg__(self): return _u(self, np.negative)
    def __pow__(self, k): return _u(self, lambda a: a ** k)
    def __gt__(self, o): return _b(self, o, np.greater)
    def __lt__(self, o): return _b(self, o, np.less)
    def __ge__(self, o): return _b(self, o, np.greater_equal)
    def __le__(self, o): return _b(self, o, np.less_equal)


def _n(x):
    return x if isinstance(x, N) else N(lambda env, v=float(x): np.full(env["#rows"], v))


def _b(a, b, op):
    a, b = _n(a), _n(b)
    return N(lambda env: op(a(env), b(env)))


def _u(a, op):
    a = _n(a)
    return N(lambda env: op(a(env)))


class NVec:
    """A short vector of twin values (dsl.Vec's counterpart): element-wise arithmetic, indexing."""

    def __init__(self, elems): self.e = [_n(x) for x in elems]
    def __len__(self): return len(self.e)
    def __iter__(self): return iter(self.e)
    def __getitem__(self, i): return NVec(self.e[i]) if isinstance(i, slice) else self.e[i]
    def _z(self, o, f): return NVec([f(a, b) for a, b in zip(self.e, o.e if isinstance(o, NVec) else [o] * len(self.e))])
    def __add__(self, o): return self._z(o, lambda a, b: a + b)
    def __sub__(self, o): return self._z(o, lambda a, b: a - b)
    def __mul__(self, o): return self._z(o, lambda a, b: a * b)
    def __rmul__(self, o): return self._z(o, lambda a, b: b * a)


def _stack(v, env):
    return np.stack([x(env) for x in v.e], axis=0)


class _TwinNp:
    abs = staticmethod(lambda a: _u(a, np.abs))
    sin = staticmethod(lambda a: _u(a, np.sin))
    cos = staticmethod(lambda a: _u(a, np.cos))
    tan = staticmethod(lambda a: _u(a, np.tan))
    tanh = staticmethod(lambda a: _u(a, np.tanh))
    sqrt = staticmethod(lambda a: _u(a, np.sqrt))
    exp = staticmethod(lambda a: _u(a, np.exp))
    log = staticmethod(lambda a: _u(a, np.log))
    log1p = staticmethod(lambda a: _u(a, np.log1p))
    expm1 = staticmethod(lambda a: _u(a, np.expm1))
    cbrt = staticmethod(lambda a: _u(a, np.cbrt))
    sinh = staticmethod(lambda a: _u(a, np.sinh))
    cosh = staticmethod(lambda a: _u(a, np.cosh))
    erfc = staticmethod(lambda a: _u(a, scipy.special.erfc))
    arccos = staticmethod(lambda a: _u(a, np.arccos))
    arcsin = staticmethod(lambda a: _u(a, np.arcsin))
    arctan = staticmethod(lambda a: _u(a, np.arctan))
    sign = staticmethod(lambda a: _u(a, np.sign))
    floor = staticmethod(lambda a: _u(a, np.floor))
    ceil = staticmethod(lambda a: _u(a, np.ceil))
    trunc = staticmethod(lambda a: _u(a, np.trunc))
    rint = staticmethod(lambda a: _u(a, np.rint))                       # round half to even, like jnp.rint
    maximum = staticmethod(lambda a, b: _b(a, b, np.maximum))
    minimum = staticmethod(lambda a, b: _b(a, b, np.minimum))
    hypot = staticmethod(lambda a, b: _b(a, b, np.hypot))
    arctan2 = staticmethod(lambda a, b: _b(a, b, np.arctan2))
    remainder = staticmethod(lambda a, b: _b(a, b, np.remainder))       # sign of the divisor, like jnp.remainder
    power = staticmethod(lambda a, b: _b(a, b, np.power))
    logical_and = staticmethod(lambda a, b: _b(a, b, np.logical_and))
    logical_or = staticmethod(lambda a, b: _b(a, b, np.logical_or))
    logical_not = staticmethod(lambda a: _u(a, np.logical_not))

    @staticmethod
    def clip(a, lo, hi): return _u(a, lambda x: np.clip(x, lo, hi))

    @staticmethod
    def where(c, a, b):
        c, a, b = _n(c), _n(a), _n(b)
        return N(lambda env: np.where(c(env), a(env), b(env)))

    @staticmethod
    def dot(u, v): return N(lambda env: np.sum(_stack(u, env) * _stack(v, env), axis=0))

    @staticmethod
    def cross(u, v):
        return NVec([N(lambda env, k=k: np.cross(_stack(u, env), _stack(v, env), axis=0)[k]) for k in range(3)])

    @staticmethod
    def sum(v): return N(lambda env: np.sum(_stack(v, env), axis=0))

    @staticmethod
    def max(v): return N(lambda env: np.max(_stack(v, env), axis=0))

    @staticmethod
    def min(v): return N(lambda env: np.min(_stack(v, env), axis=0))

    @staticmethod
    def sort(v): return NVec([N(lambda env, k=k: np.sort(_stack(v, env), axis=0)[k]) for k in range(len(v))])

    @staticmethod
    def interp(x, xs, fs): return _u(x, lambda a: np.interp(a, np.array(xs), np.array(fs)))

    class linalg:
        @staticmethod
        def norm(v): return N(lambda env: np.sqrt(np.sum(_stack(v, env) ** 2, axis=0)))


class _TwinLax:
    @staticmethod
    def cond(pred, true_fun, false_fun, *operands, operand=None):
        args = operands if operands else (operand,)
        return _TwinNp.where(pred, true_fun(*args), false_fun(*args))

    @staticmethod
    def select(pred, a, b): return _TwinNp.where(pred, a, b)

    @staticmethod
    def switch(index, branches, *operands):                            # jax clamps the index into range
        index, outs = _n(index), [_n(br(*operands)) for br in branches]
        return N(lambda env: np.choose(np.clip(index(env).astype(int), 0, len(outs) - 1), [o(env) for o in outs]))

    @staticmethod
    def fori_loop(lower, upper, body, init):
        v = init
        for i in range(lower, upper):
            v = body(i, v)
        return v


TWIN = types.SimpleNamespace(np=_TwinNp, lax=_TwinLax, Vec=NVec, const=_n)


class Gen:
    """Random expressions whose values stay O(1) (every partial function is fed through a guard), so the comparison
    measures the generated code and not the conditioning of the expression."""

    def __init__(self, seed, backend=None):
        """backend: the namespace the expressions are built in — TRACED (elodin_amd.dsl: nodes of a program) or TWIN (plain
        numpy closures, tests/fuzz_gen.py): the same seed draws the same expression in either."""
        self.rng = np.random.default_rng(seed)
        self.b = backend or TRACED

    def pick(self, xs):
        return xs[int(self.rng.integers(len(xs)))]

    def const(self):
        return float(np.round(self.rng.uniform(-2.0, 2.0), 3))

    def scalar(self, leaves, depth):
        r = self.rng
        if depth == 0 or r.random() < 0.12:
            return self.pick(leaves) if r.random() < 0.8 else self.b.const(self.const())
        s = lambda: self.scalar(leaves, depth - 1)
        kind = self.pick(["bin", "bin", "un", "un", "sel", "vec", "pow", "cmpmix", "lax"])
        if kind == "bin":
            a, b = s(), s()
            return self.pick([lambda: a + b, lambda: a - b, lambda: a * b, lambda: a / (self.b.np.abs(b) + 0.5),
                              lambda: self.b.np.maximum(a, b), lambda: self.b.np.minimum(a, b), lambda: self.b.np.hypot(a, b),
                              lambda: self.b.np.arctan2(a, b + 2.5 * self.b.np.sign(b) + 0.1), lambda: a * self.const() + b,
                              lambda: self.b.np.remainder(a, self.b.np.abs(b) + 0.7)])()
        if kind == "un":
            a = s()
            return self.pick([lambda: self.b.np.sin(a), lambda: self.b.np.cos(a), lambda: self.b.np.tanh(a), lambda: self.b.np.sqrt(self.b.np.abs(a) + 0.1),
                              lambda: self.b.np.exp(self.b.np.clip(a, -3.0, 2.0)), lambda: self.b.np.log(self.b.np.abs(a) + 0.5), lambda: self.b.np.abs(a) - 0.3,
                              lambda: self.b.np.arccos(self.b.np.clip(a, -0.95, 0.95)), lambda: self.b.np.arcsin(self.b.np.clip(a * 0.5, -0.95, 0.95)),
                              lambda: self.b.np.arctan(a), lambda: self.b.np.tan(self.b.np.clip(a, -1.2, 1.2)), lambda: self.b.np.log1p(self.b.np.abs(a)),
                              lambda: self.b.np.expm1(self.b.np.clip(a, -2.0, 1.0)), lambda: self.b.np.cbrt(self.b.np.abs(a) + 0.2), lambda: self.b.np.sinh(self.b.np.clip(a, -2.0, 2.0)),
                              lambda: self.b.np.cosh(self.b.np.clip(a, -2.0, 2.0)), lambda: self.b.np.erfc(a), lambda: -a, lambda: self.b.np.sign(a) * 0.5 + a,
                              lambda: self.b.np.clip(a, -0.7, 0.9), lambda: a ** 2, lambda: a ** 3])()
        if kind == "sel":
            c = self.cond(leaves, depth - 1)
            return self.b.np.where(c, s(), s())
        if kind == "lax":
            c, a, b = self.cond(leaves, depth - 1), s(), s()
            return self.pick([lambda: self.b.lax.cond(c, lambda _: a * 2.0, lambda _: b - 1.0, operand=None),
                              lambda: self.b.lax.select(c, a, b),
                              lambda: self.b.lax.switch(self.b.np.floor(self.b.np.clip(a, 0.0, 2.9)), [lambda: a, lambda: b, lambda: a * b]),
                              lambda: self.b.lax.fori_loop(0, 3, lambda i, v: v * 0.5 + self.b.np.sin(v + float(i)), a)])()
        if kind == "pow":
            return self.b.np.power(self.b.np.abs(s()) + 0.5, self.pick([-2.0, -1.5, -0.5, 0.5, 1.5, 2.0, 3.0, self.const()]))
        if kind == "cmpmix":       # staircase functions of an exactly representable argument
            leaf = self.pick(leaves)
            return self.pick([self.b.np.floor, self.b.np.ceil, self.b.np.trunc, self.b.np.rint])(leaf * 4.0) * 0.25 + s()
        u, v = self.vec3(leaves, depth - 1), self.vec3(leaves, depth - 1)
        return self.pick([lambda: self.b.np.dot(u, v), lambda: self.b.np.linalg.norm(u), lambda: self.b.np.cross(u, v)[int(self.rng.integers(3))],
                          lambda: self.b.np.sum(u * v + u), lambda: self.b.np.max(u) - self.b.np.min(v), lambda: self.b.np.sort(u)[1],
                          lambda: self.b.np.interp(self.b.np.clip(u[0], -1.0, 1.0), [-1.0, -0.2, 0.3, 1.0], [0.5, -1.0, 2.0, 0.25])])()

    def vec3(self, leaves, depth):
        return self.b.Vec([self.scalar(leaves, depth) for _ in range(3)])

    def cond(self, leaves, depth):
        a, b = self.scalar(leaves, depth), self.scalar(leaves, depth)
        c = self.pick([lambda: a > b, lambda: a < b + 0.25, lambda: a >= -b])()
        if self.rng.random() < 0.3:
            d = self.scalar(leaves, depth) > 0.1
            c = self.pick([self.b.np.logical_and, self.b.np.logical_or])(c, d) if self.rng.random() < 0.7 else self.b.np.logical_not(c)
        return c



def make_program(seed, depth=4):
    g = Gen(seed, TRACED)
    np_ = dsl.np

    @dsl.system(x=8, y=8, a=16)
    def sys_a(x, y, a):
        leaves = list(x.e) + list(y.e)
        return {"a": dsl.Vec([g.scalar(leaves, depth) for _ in range(16)])}

    @dsl.system(x=8, a=16, b=16)
    def sys_b(x, a, b, tick):
        leaves = list(x.e) + list(a.e) + [np_.sin(tick * 0.37)]
        out = dsl.Vec([g.scalar(leaves, depth) for _ in range(16)])
        return {"b": out, "x": x * 0.5 + dsl.Vec([np_.tanh(e) for e in out.e[:8]])}     # x rewritten: later reads must see it

    @dsl.system(every=2, x=8, a=16, b=16, c=8)
    def sys_c(x, a, b, c):
        leaves = list(x.e) + list(a.e[:4]) + list(b.e[:4]) + list(c.e[:2])
        return {"c": dsl.Vec([g.scalar(leaves, depth - 1) for _ in range(8)])}
    return dsl.Program([sys_a, sys_b], dsl.Pipe([]), [sys_c])


def columns(seed, n):
    rng = np.random.default_rng(1000 + seed)
    quant = lambda a: np.round(a * 64.0) / 64.0           # exactly representable in f32 too
    return {"x": quant(rng.uniform(-1.5, 1.5, (n, 8))), "y": quant(rng.uniform(-1.5, 1.5, (n, 8))),
            "a": np.zeros((n, 16)), "b": np.zeros((n, 16)), "c": np.zeros((n, 8))}


def twin_run(seed, cols, ticks, depth=4):
    """make_program(seed)'s three systems written out in plain numpy — the same draws from the same seed, the same pipe
    semantics (a system reads what its predecessors wrote this tick and computes all its outputs before writing any; sys_c runs
    on even ticks) — stepped `ticks` ticks over copies of `cols`.  No elodin_amd code evaluates anything here."""
    g = Gen(seed, TWIN)
    col = lambda name, k: N(lambda env: env[name][:, k])
    vec = lambda name, w: NVec([col(name, k) for k in range(w)])
    x, y, a, b, c = vec("x", 8), vec("y", 8), vec("a", 16), vec("b", 16), vec("c", 8)
    tick = N(lambda env: np.full(env["#rows"], float(env["#tick"])))
    # drawn in the order the tracer calls the systems: sys_a, sys_b, sys_c
    out_a = [g.scalar(list(x.e) + list(y.e), depth) for _ in range(16)]
    out_b = [g.scalar(list(x.e) + list(a.e) + [_TwinNp.sin(tick * 0.37)], depth) for _ in range(16)]
    new_x = [x.e[k] * 0.5 + _TwinNp.tanh(out_b[k]) for k in range(8)]
    out_c = [g.scalar(list(x.e) + list(a.e[:4]) + list(b.e[:4]) + list(c.e[:2]), depth - 1) for _ in range(8)]
    env = {k: np.array(v, dtype=np.float64) for k, v in cols.items()}
    env["#rows"] = len(cols["x"])
    with np.errstate(all="ignore"):
        for t in range(1, ticks + 1):
            env["#tick"] = t
            env["a"] = np.stack([_n(e)(env) for e in out_a], axis=1)
            vb = np.stack([_n(e)(env) for e in out_b], axis=1)
            vx = np.stack([_n(e)(env) for e in new_x], axis=1)
            env["b"], env["x"] = vb, vx
            if t % 2 == 0:
                env["c"] = np.stack([_n(e)(env) for e in out_c], axis=1)
    return {k: env[k] for k in cols}
