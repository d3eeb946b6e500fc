"""Small dense matrices for traced user code: `Mat` and the `jax.numpy.linalg` / `jax.scipy.linalg` calls the reference's
estimators make per entity (examples/linalg/sim.py — solve, inv, cholesky, qr, det, slogdet, svd, eigh, norm;
examples/cube-sat/main.py:262-291 — a 6-state MEKF update with `pinv`; examples/drone/mekf.py:183 — `inv`).

The reference lowers these to LAPACK custom calls inside its compiled tick (libs/cranelift-mlir: `lapack_dgetrf_ffi`,
`lapack_dpotrf_ffi`, `lapack_dgesdd_ffi`, ...).  Here a matrix of a traced system is a grid of scalar expression nodes — one
lane is one entity, an n x n matrix with n <= 8 lives in that lane's registers — and every factorisation is UNROLLED at
trace time into straight-line, branch-free arithmetic on those nodes (pivoting is a cascade of compare-and-swap selects),
which the code generator then emits into the fused step kernel like any other expression.  No contraction dimension is
shared between lanes, so this is VALU work, not MFMA work: MI355X-first here means no per-entity library call, no memory
traffic, no divergence.

Algorithms (results agree with LAPACK's to rounding; where a factorisation is only defined up to signs or order the
convention of jax.numpy is kept):
  solve / inv / det / slogdet   LU with partial pivoting (dgetrf's pivot rule: largest magnitude in the column, first wins)
  cholesky                      Cholesky-Banachiewicz, lower (upper = transpose), like dpotrf('L')
  qr                            Householder reflections with dgeqrf's sign convention (R's diagonal opposes the column's head)
  eigh / eigvalsh               cyclic Jacobi, a fixed number of sweeps, eigenvalues ascending
  svd / pinv                    one-sided (Hestenes) Jacobi, singular values descending; pinv cuts at rcond * s_max with
                                jax's default rcond = 10 * max(m, n) * eps
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

from . import dsl as _d

MAX_ELEMS = 64          # the most values a register matrix may hold (dsl._MAT_MAX_ELEMS; larger 2-D components are dsl.Window)
EPS = 2.220446049250313e-16


def jacobi_sweeps(n: int) -> int:
    """Sweeps of cyclic Jacobi for an n x n problem.  Convergence is quadratic once the off-diagonal mass is small (each sweep
    roughly squares it: 1e-1, 1e-2, 1e-4, 1e-8, 1e-16), so 4 + ceil(log2 n) sweeps reach rounding level from any start
    (tests/test_linalg_reference.py checks eigenvalues / singular values / reconstructions against LAPACK at 1e-11 for
    n = 2..8); every sweep is n (n - 1) / 2 unrolled rotations, so sweeps are not free."""
    return 4 + max(1, math.ceil(math.log2(max(n, 2))))


def _s(x):
    return _d._lift(x)


class _MatAt:
    """`m.at[i, j].set(v)` / `.add(v)` with static indices or slices (jax's functional update)."""

    def __init__(self, m):
        self.m = m

    def __getitem__(self, idx):
        return _MatAtIdx(self.m, idx)


class _MatAtIdx:
    def __init__(self, m, idx):
        self.m, self.idx = m, idx

    def _apply(self, value, combine):
        rows, cols = self.m._ranges(self.idx)
        g = [list(r.e) for r in self.m]
        if isinstance(value, Mat):
            if value.shape != (len(rows), len(cols)):
                raise ValueError(f"at[...]: shape {value.shape} does not fit a {len(rows)} x {len(cols)} block")
            src = [[value[i].e[j] for j in range(len(cols))] for i in range(len(rows))]
        elif isinstance(value, _d.Vec):
            if len(rows) == 1 and len(value) == len(cols):
                src = [list(value.e)]
            elif len(cols) == 1 and len(value) == len(rows):
                src = [[e] for e in value.e]
            else:
                raise ValueError("at[...]: vector does not fit the block")
        else:
            src = [[_s(value)] * len(cols) for _ in rows]
        for a, i in enumerate(rows):
            for b, j in enumerate(cols):
                g[i][j] = combine(g[i][j], src[a][b])
        return Mat(g)

    def set(self, value):
        return self._apply(value, lambda old, new: new)

    def add(self, value):
        return self._apply(value, lambda old, new: old + new)


class Mat(list):
    """A static-shape matrix of scalar nodes: a list of `Vec` rows (so code written against lists of rows keeps working)
    with jax.numpy's array surface for 2-D values."""

    __array_ufunc__ = None

    def __init__(self, rows: Sequence):
        rs = [r if isinstance(r, _d.Vec) else _d.Vec(list(r)) for r in rows]
        if not rs or any(len(r) != len(rs[0]) for r in rs):
            raise ValueError("a matrix needs rows of equal length")
        super().__init__(rs)

    # ---- shape ----------------------------------------------------------------------------------------------------
    @property
    def shape(self) -> Tuple[int, int]:
        return (len(self), len(self[0]))

    @property
    def T(self) -> "Mat":
        r, c = self.shape
        return Mat([[list.__getitem__(self, i).e[j] for i in range(r)] for j in range(c)])

    def transpose(self, *axes):
        return self.T

    def flatten(self) -> "_d.Vec":
        return _d.Vec([e for row in self for e in row.e])

    ravel = flatten

    def reshape(self, *shape):
        shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape
        return reshape(self.flatten(), shape)

    def col(self, j: int) -> "_d.Vec":
        return _d.Vec([row.e[j] for row in self])

    @property
    def at(self):
        return _MatAt(self)

    def _ranges(self, idx):
        r, c = self.shape
        if not isinstance(idx, tuple):
            idx = (idx, slice(None))
        ri, ci = idx
        rows = list(range(r))[ri] if isinstance(ri, slice) else [range(r)[int(ri)]]
        cols = list(range(c))[ci] if isinstance(ci, slice) else [range(c)[int(ci)]]
        return rows, cols

    dtype = _d._numpy.float64

    def astype(self, dtype):
        return Mat([r.astype(dtype) for r in self])

    def __getitem__(self, idx):
        if isinstance(idx, _d.Expr):        # a TRACED row index (a flight plan's `points[time.astype(int)]`): clamped select chain
            return _d._dynamic_index([list.__getitem__(self, k) for k in range(len(self))], idx)
        if isinstance(idx, int):
            return list.__getitem__(self, idx)
        if isinstance(idx, slice):
            return Mat(list.__getitem__(self, idx))
        ri, ci = idx
        rows, cols = self._ranges(idx)
        if isinstance(ri, int) and isinstance(ci, int):
            return list.__getitem__(self, rows[0]).e[cols[0]]
        if isinstance(ri, int):
            return _d.Vec([list.__getitem__(self, rows[0]).e[j] for j in cols])
        if isinstance(ci, int):
            return _d.Vec([list.__getitem__(self, i).e[cols[0]] for i in rows])
        return Mat([[list.__getitem__(self, i).e[j] for j in cols] for i in rows])

    # ---- arithmetic ------------------------------------------------------------------------------------------------
    def _zip(self, o, f):
        o = _d._host(o)
        if isinstance(o, Mat) and o.shape == self.shape:
            return Mat([[f(a, b) for a, b in zip(ra.e, rb.e)] for ra, rb in zip(self, o)])
        if isinstance(o, Mat) and o.shape != self.shape:      # numpy broadcasting of a column / row matrix
            (r, c), (ro, co) = self.shape, o.shape
            if co == 1 and ro == r:
                return Mat([[f(a, rb.e[0]) for a in ra.e] for ra, rb in zip(self, o)])
            if c == 1 and ro == r:
                return Mat([[f(ra.e[0], b) for b in rb.e] for ra, rb in zip(self, o)])
            if ro == 1 and co == c:
                return Mat([[f(a, b) for a, b in zip(ra.e, o[0].e)] for ra in self])
            if r == 1 and co == c:
                return Mat([[f(a, b) for a, b in zip(self[0].e, rb.e)] for rb in o])
            raise ValueError(f"shape mismatch {self.shape} / {o.shape}")
        if isinstance(o, _d.Vec):      # broadcasting a row vector over the rows, like numpy
            if len(o) != self.shape[1]:
                raise ValueError("shape mismatch")
            return Mat([[f(a, b) for a, b in zip(ra.e, o.e)] for ra in self])
        return Mat([[f(a, o) for a in ra.e] for ra in self])

    def __add__(self, o): return self._zip(o, lambda a, b: a + b)
    def __radd__(self, o): return self._zip(o, lambda a, b: b + a)
    def __sub__(self, o): return self._zip(o, lambda a, b: a - b)
    def __rsub__(self, o): return self._zip(o, lambda a, b: b - a)
    def __mul__(self, o): return self._zip(o, lambda a, b: a * b)
    def __rmul__(self, o): return self._zip(o, lambda a, b: b * a)
    def __truediv__(self, o): return self._zip(o, lambda a, b: a / b)
    def __neg__(self): return Mat([[-a for a in r.e] for r in self])
    def __lt__(self, o): return self._zip(o, lambda a, b: a < b)
    def __le__(self, o): return self._zip(o, lambda a, b: a <= b)
    def __gt__(self, o): return self._zip(o, lambda a, b: a > b)
    def __ge__(self, o): return self._zip(o, lambda a, b: a >= b)
    def __and__(self, o): return self._zip(o, lambda a, b: a & b)
    def __or__(self, o): return self._zip(o, lambda a, b: a | b)
    def __invert__(self): return Mat([[~a for a in r.e] for r in self])
    def __rtruediv__(self, o): return self._zip(o, lambda a, b: b / a)
    def __pow__(self, k): return Mat([[a ** k for a in r.e] for r in self])
    def __iadd__(self, o): return self.__add__(o)          # list.__iadd__ would extend the row list
    def __imul__(self, o): return self.__mul__(o)

    def __matmul__(self, o):
        return matmul(self, o)

    def __rmatmul__(self, o):
        return matmul(o, self)

    def dot(self, o):
        return matmul(self, o)


class Batch(list):
    """A stack of equal-shape matrices (a [b, r, c] array): what `jnp.linalg.cholesky` of a batch returns
    (examples/linalg/sim.py:344-350).  Element-wise arithmetic, `@` and the last-two-axes transpose map over the stack."""
    __array_ufunc__ = None

    def __init__(self, mats):
        super().__init__([as_mat(m) for m in mats])

    def _zip(self, o, f):
        o = _d._host(o)
        if isinstance(o, Batch):
            if len(o) != len(self):
                raise ValueError("batch size mismatch")
            return Batch([f(a, b) for a, b in zip(self, o)])
        return Batch([f(a, o) for a in self])

    def __add__(self, o): return self._zip(o, lambda a, b: a + b)
    def __radd__(self, o): return self._zip(o, lambda a, b: b + a)
    def __sub__(self, o): return self._zip(o, lambda a, b: a - b)
    def __rsub__(self, o): return self._zip(o, lambda a, b: b - a)
    def __mul__(self, o): return self._zip(o, lambda a, b: a * b)
    def __rmul__(self, o): return self._zip(o, lambda a, b: b * a)
    def __matmul__(self, o): return self._zip(o, lambda a, b: a @ b)
    def __rmatmul__(self, o): return self._zip(o, lambda a, b: b @ a)

    @property
    def mT(self): return Batch([m.T for m in self])


def _dot(a: Sequence, b: Sequence):
    """sum_k a_k b_k in index order, structural zeros skipped (a product with a constant 0 adds nothing to a finite sum: the
    reference's compiled code multiplies them out, which differs only for non-finite operands)."""
    acc = None
    for x, y in zip(a, b):
        if (x.op == "const" and x.value == 0.0) or (y.op == "const" and y.value == 0.0):
            continue
        t = x * y
        acc = t if acc is None else acc + t
    return acc if acc is not None else _d.const(0.0)


def matmul(a, b):
    """`a @ b` for Mat / Vec operands (numpy's rules: vectors are promoted and the added axis dropped again)."""
    a, b = _d._host(a), _d._host(b)
    if isinstance(a, Mat) and isinstance(b, Mat):
        if a.shape[1] != b.shape[0]:
            raise ValueError(f"matmul: {a.shape} @ {b.shape}")
        bt = b.T
        return Mat([[_dot(ra.e, cb.e) for cb in bt] for ra in a])
    if isinstance(a, Mat) and isinstance(b, _d.Vec):
        if a.shape[1] != len(b):
            raise ValueError(f"matmul: {a.shape} @ ({len(b)},)")
        return _d.Vec([_dot(ra.e, b.e) for ra in a])
    if isinstance(a, _d.Vec) and isinstance(b, Mat):
        if b.shape[0] != len(a):
            raise ValueError(f"matmul: ({len(a)},) @ {b.shape}")
        return _d.Vec([_dot(a.e, cb.e) for cb in b.T])
    if isinstance(a, _d.Vec) and isinstance(b, _d.Vec):
        return _dot(a.e, b.e)
    if isinstance(a, list) and a and isinstance(a[0], _d.Vec):        # a plain list of rows (np.stack / np.outer before Mat)
        return matmul(Mat(a), b)
    if isinstance(b, list) and b and isinstance(b[0], _d.Vec):
        return matmul(a, Mat(b))
    raise TypeError(f"matmul: unsupported operands {type(a).__name__} @ {type(b).__name__}")


def as_mat(x) -> Mat:
    x = _d._host(x)
    if isinstance(x, Mat):
        return x
    if isinstance(x, (list, tuple)) and x and isinstance(x[0], (_d.Vec, list, tuple)):
        return Mat(x)
    raise TypeError("expected a matrix (Mat or a list of rows)")


def reshape(v, shape):
    shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    flat = v.flatten() if isinstance(v, Mat) else v
    n = len(flat)
    if len(shape) == 1:
        if shape[0] not in (-1, n):
            raise ValueError("reshape: size mismatch")
        return flat
    r, c = shape
    r = n // c if r == -1 else r
    c = n // r if c == -1 else c
    if r * c != n:
        raise ValueError("reshape: size mismatch")
    return Mat([flat.e[i * c:(i + 1) * c] for i in range(r)])


def eye(n, m=None) -> Mat:
    m = n if m is None else m
    return Mat([[1.0 if i == j else 0.0 for j in range(int(m))] for i in range(int(n))])


def zeros2(r, c) -> Mat:
    return Mat([[0.0] * int(c) for _ in range(int(r))])


def diag(x):
    if isinstance(x, Mat) or (isinstance(x, list) and x and isinstance(x[0], _d.Vec)):
        m = as_mat(x)
        return _d.Vec([m[i].e[i] for i in range(min(m.shape))])
    n = len(x)
    return Mat([[x.e[i] if i == j else 0.0 for j in range(n)] for i in range(n)])


def block(blocks) -> Mat:
    """jnp.block of a list of rows of blocks (Mat / Vec-as-row / scalars), or of one row of blocks."""
    if blocks and not isinstance(blocks[0], (list, tuple)) or (blocks and isinstance(blocks[0], Mat)):
        blocks = [blocks]
    out_rows: List[list] = []
    for brow in blocks:
        mats = [b if isinstance(b, Mat) else (Mat([b]) if isinstance(b, _d.Vec) else Mat([[b]])) for b in brow]
        h = mats[0].shape[0]
        if any(m.shape[0] != h for m in mats):
            raise ValueError("block: blocks of one row need equal heights")
        for i in range(h):
            out_rows.append([e for m in mats for e in m[i].e])
    return Mat(out_rows)


def skew(v) -> Mat:
    """el.skew: the cross-product matrix [v]x (libs/nox-py/python/elodin/__init__.py skew)."""
    x, y, z = v.e
    return Mat([[0.0, -z, y], [z, 0.0, -x], [-y, x, 0.0]])


def trace(m) -> "_d.Expr":
    m = as_mat(m)
    acc = m[0].e[0]
    for i in range(1, min(m.shape)):
        acc = acc + m[i].e[i]
    return acc


def fro_norm(m) -> "_d.Expr":
    acc = None
    if isinstance(m, Batch):       # norm of the whole [b, r, c] array
        for mat in m:
            for r in mat:
                for e in r.e:
                    acc = e * e if acc is None else acc + e * e
        return _d._un("sqrt", acc)
    for r in as_mat(m):
        for e in r.e:
            acc = e * e if acc is None else acc + e * e
    return _d._un("sqrt", acc)


# ---- LU -----------------------------------------------------------------------------------------------------------------

def _lu(a: Mat, rhs: List[List["_d.Expr"]]):
    """Gaussian elimination with partial pivoting on copies of `a` and of the right-hand-side columns `rhs` (a list of rows).
    The pivot of column k is brought up by a cascade of compare-and-swap selects over the rows below (strictly larger
    magnitude swaps: the first of equal candidates stays, dgetrf's rule; the rows left behind end up in another order than
    LAPACK's single swap leaves them, which changes no arithmetic — every later pivot is again the largest of the same set).
    Returns (U rows, transformed rhs rows, permutation sign)."""
    n = a.shape[0]
    if a.shape[1] != n:
        raise ValueError("square matrix expected")
    u = [list(r.e) for r in a]
    b = [list(r) for r in rhs]
    sign = _d.const(1.0)
    where = _d._Np.where
    for k in range(n):
        for i in range(k + 1, n):
            swap = _d._Np.abs(u[i][k]) > _d._Np.abs(u[k][k])
            for j in range(k, n):
                hi, lo = where(swap, u[i][j], u[k][j]), where(swap, u[k][j], u[i][j])
                u[k][j], u[i][j] = hi, lo
            for j in range(len(b[0]) if b else 0):
                hi, lo = where(swap, b[i][j], b[k][j]), where(swap, b[k][j], b[i][j])
                b[k][j], b[i][j] = hi, lo
            sign = where(swap, -sign, sign)
        for i in range(k + 1, n):
            f = u[i][k] / u[k][k]
            for j in range(k + 1, n):
                u[i][j] = u[i][j] - f * u[k][j]
            u[i][k] = _d.const(0.0)
            for j in range(len(b[0]) if b else 0):
                b[i][j] = b[i][j] - f * b[k][j]
    return u, b, sign


def solve(a, b):
    """jnp.linalg.solve(a, b): b a vector or a matrix of right-hand-side columns."""
    a = as_mat(a)
    vec = isinstance(b, _d.Vec)
    rhs = [[e] for e in b.e] if vec else [list(r.e) for r in as_mat(b)]
    n = a.shape[0]
    if len(rhs) != n:
        raise ValueError("solve: shapes do not match")
    u, y, _ = _lu(a, rhs)
    m = len(rhs[0])
    x = [[None] * m for _ in range(n)]
    for j in range(m):
        for i in range(n - 1, -1, -1):
            acc = y[i][j]
            for k in range(i + 1, n):
                acc = acc - u[i][k] * x[k][j]
            x[i][j] = acc / u[i][i]
    return _d.Vec([r[0] for r in x]) if vec else Mat(x)


def solve_triangular(a, b, lower: bool = False, trans=0, unit_diagonal: bool = False):
    """jax.scipy.linalg.solve_triangular(a, b, lower=...): forward / back substitution (dtrsm's order: row by row, the dot over the
    already solved entries first, then the division); only the named triangle of `a` is read."""
    a = as_mat(a)
    if trans not in (0, "N", 1, "T"):
        raise NotImplementedError("solve_triangular: trans must be 0 / 'N' / 1 / 'T'")
    if trans in (1, "T"):
        a, lower = a.T, not lower
    vec = isinstance(b, _d.Vec)
    rhs = [[e] for e in b.e] if vec else [list(r.e) for r in as_mat(b)]
    n = a.shape[0]
    if len(rhs) != n:
        raise ValueError("solve_triangular: shapes do not match")
    m = len(rhs[0])
    x = [[None] * m for _ in range(n)]
    for j in range(m):
        for i in (range(n) if lower else range(n - 1, -1, -1)):
            acc = _s(rhs[i][j])
            for k in (range(i) if lower else range(i + 1, n)):
                acc = acc - a[i].e[k] * x[k][j]
            x[i][j] = acc if unit_diagonal else acc / a[i].e[i]
    return _d.Vec([r[0] for r in x]) if vec else Mat(x)


def inv(a) -> Mat:
    a = as_mat(a)
    return solve(a, eye(a.shape[0]))


def det(a):
    a = as_mat(a)
    u, _, sign = _lu(a, [[] for _ in range(a.shape[0])])
    acc = sign
    for i in range(a.shape[0]):
        acc = acc * u[i][i]
    return acc


def slogdet(a):
    a = as_mat(a)
    u, _, sign = _lu(a, [[] for _ in range(a.shape[0])])
    logabs = None
    for i in range(a.shape[0]):
        d = u[i][i]
        sign = _d._Np.where(d < 0.0, -sign, sign)
        t = _d._Np.log(_d._Np.abs(d))
        logabs = t if logabs is None else logabs + t
    return sign, logabs


# ---- Cholesky / QR ----------------------------------------------------------------------------------------------------------

def cholesky(a, lower: bool = True, upper: bool = False) -> Mat:
    """Lower-triangular L with L L^T = a (jnp.linalg.cholesky; jax.scipy.linalg.cholesky(a, lower=False) returns L^T)."""
    a = _d._host(a)
    if isinstance(a, Batch):
        return Batch([cholesky(m, lower, upper) for m in a])
    a = as_mat(a)
    n = a.shape[0]
    L = [[_d.const(0.0)] * n for _ in range(n)]
    for j in range(n):
        acc = a[j].e[j]
        for k in range(j):
            acc = acc - L[j][k] * L[j][k]
        L[j][j] = _d._un("sqrt", acc)
        for i in range(j + 1, n):
            acc = a[i].e[j]
            for k in range(j):
                acc = acc - L[i][k] * L[j][k]
            L[i][j] = acc / L[j][j]
    m = Mat(L)
    return m.T if (upper or not lower) else m


def qr(a) -> Tuple[Mat, Mat]:
    """Reduced QR by Householder reflections, dgeqrf's convention: beta = -sign(x_0) |x|, so R's diagonal opposes the head of
    the column it came from; Q is accumulated explicitly (dorgqr)."""
    a = as_mat(a)
    m, n = a.shape
    r = [list(row.e) for row in a]
    q = [[_d.const(1.0 if i == j else 0.0) for j in range(m)] for i in range(m)]
    where, sqrt, absf = _d._Np.where, (lambda x: _d._un("sqrt", x)), _d._Np.abs
    for k in range(min(m - 1, n)):
        x = [r[i][k] for i in range(k, m)]
        tail = None
        for e in x[1:]:
            tail = e * e if tail is None else tail + e * e
        norm = sqrt(x[0] * x[0] + tail)
        beta = where(x[0] >= 0.0, -norm, norm)
        degenerate = tail <= 0.0                     # nothing below the diagonal: H = I (dlarfg's tau = 0)
        v0 = x[0] - beta
        v0s = where(degenerate, 1.0, v0)
        v = [_d.const(1.0)] + [e / v0s for e in x[1:]]
        tau = where(degenerate, 0.0, (beta - x[0]) / where(degenerate, 1.0, beta))
        for j in range(k, n):                         # R <- (I - tau v v^T) R
            s = None
            for i in range(k, m):
                t = v[i - k] * r[i][j]
                s = t if s is None else s + t
            s = s * tau
            for i in range(k, m):
                r[i][j] = r[i][j] - v[i - k] * s
        for i in range(m):                            # Q <- Q (I - tau v v^T)
            s = None
            for j in range(k, m):
                t = q[i][j] * v[j - k]
                s = t if s is None else s + t
            s = s * tau
            for j in range(k, m):
                q[i][j] = q[i][j] - s * v[j - k]
        for i in range(k + 1, m):
            r[i][k] = _d.const(0.0)
    kk = min(m, n)
    return Mat([row[:kk] for row in q]), Mat([r[i][:n] for i in range(kk)])


# ---- Jacobi: eigh, svd, pinv -----------------------------------------------------------------------------------------------

def _sort_network(keys: list, cols: List[list], descending: bool):
    """Odd-even transposition sort of `keys` carrying the columns of each matrix in `cols` along (data-independent)."""
    n = len(keys)
    where = _d._Np.where
    for rnd in range(n):
        for i in range(rnd % 2, n - 1, 2):
            swap = (keys[i] < keys[i + 1]) if descending else (keys[i] > keys[i + 1])
            keys[i], keys[i + 1] = where(swap, keys[i + 1], keys[i]), where(swap, keys[i], keys[i + 1])
            for m in cols:
                for row in m:
                    row[i], row[i + 1] = where(swap, row[i + 1], row[i]), where(swap, row[i], row[i + 1])


def eigh(a, sweeps: int = 0):
    """jnp.linalg.eigh of a symmetric matrix: (eigenvalues ascending, eigenvectors as columns).  Cyclic Jacobi: every
    rotation annihilates one off-diagonal pair; rotations on an already-zero pair are the identity (t = 0)."""
    a = as_mat(a)
    n = a.shape[0]
    sweeps = sweeps or jacobi_sweeps(n)
    s = [[(a[i].e[j] + a[j].e[i]) * 0.5 if i != j else a[i].e[i] for j in range(n)] for i in range(n)]   # like jax: symmetrize_input
    v = [[_d.const(1.0 if i == j else 0.0) for j in range(n)] for i in range(n)]
    where, sqrt, absf = _d._Np.where, (lambda x: _d._un("sqrt", x)), _d._Np.abs
    for _ in range(sweeps):
        for p in range(n - 1):
            for q in range(p + 1, n):
                apq = s[p][q]
                zero = absf(apq) <= 1e-300
                theta = (s[q][q] - s[p][p]) / (2.0 * where(zero, 1.0, apq))
                t = where(theta >= 0.0, 1.0, -1.0) / (absf(theta) + sqrt(theta * theta + 1.0))
                t = where(zero, 0.0, t)
                c = 1.0 / sqrt(t * t + 1.0)
                sn = t * c
                s[p][p], s[q][q] = s[p][p] - t * apq, s[q][q] + t * apq
                s[p][q] = s[q][p] = _d.const(0.0)
                for k in range(n):
                    if k != p and k != q:
                        akp, akq = s[k][p], s[k][q]
                        s[k][p] = s[p][k] = c * akp - sn * akq
                        s[k][q] = s[q][k] = sn * akp + c * akq
                for k in range(n):
                    vkp, vkq = v[k][p], v[k][q]
                    v[k][p], v[k][q] = c * vkp - sn * vkq, sn * vkp + c * vkq
    w = [s[i][i] for i in range(n)]
    _sort_network(w, [v], descending=False)
    return _d.Vec(w), Mat(v)


def eigvalsh(a):
    return eigh(a)[0]


def svd(a, full_matrices: bool = True, compute_uv: bool = True, sweeps: int = 0):
    """jnp.linalg.svd(a) -> (u, s, vh), singular values descending, for m >= n (square in every call site of the reference's
    examples).  One-sided Jacobi: columns of a working copy are rotated pairwise until orthogonal; their norms are the
    singular values, the normalised columns U, the accumulated rotations V."""
    a = as_mat(a)
    m, n = a.shape
    if m < n:
        u, s, vh = svd(a.T, full_matrices, True, sweeps)
        return (vh.T, s, u.T) if compute_uv else s
    sweeps = sweeps or jacobi_sweeps(n)
    w = [list(r.e) for r in a]                      # m x n, columns get orthogonalised
    v = [[_d.const(1.0 if i == j else 0.0) for j in range(n)] for i in range(n)]
    where, sqrt, absf = _d._Np.where, (lambda x: _d._un("sqrt", x)), _d._Np.abs
    for _ in range(sweeps):
        for p in range(n - 1):
            for q in range(p + 1, n):
                alpha = beta = gamma = None
                for i in range(m):
                    ap, aq = w[i][p], w[i][q]
                    alpha = ap * ap if alpha is None else alpha + ap * ap
                    beta = aq * aq if beta is None else beta + aq * aq
                    gamma = ap * aq if gamma is None else gamma + ap * aq
                zero = absf(gamma) <= 1e-300
                zeta = (beta - alpha) / (2.0 * where(zero, 1.0, gamma))
                t = where(zeta >= 0.0, 1.0, -1.0) / (absf(zeta) + sqrt(zeta * zeta + 1.0))
                t = where(zero, 0.0, t)
                c = 1.0 / sqrt(t * t + 1.0)
                sn = t * c
                for i in range(m):
                    ap, aq = w[i][p], w[i][q]
                    w[i][p], w[i][q] = c * ap - sn * aq, sn * ap + c * aq
                for i in range(n):
                    vp, vq = v[i][p], v[i][q]
                    v[i][p], v[i][q] = c * vp - sn * vq, sn * vp + c * vq
    sig = []
    for j in range(n):
        acc = None
        for i in range(m):
            acc = w[i][j] * w[i][j] if acc is None else acc + w[i][j] * w[i][j]
        sig.append(sqrt(acc))
    _sort_network(sig, [w, v], descending=True)
    if not compute_uv:
        return _d.Vec(sig)
    u = [[w[i][j] / where(sig[j] > 0.0, sig[j], 1.0) for j in range(n)] for i in range(m)]
    return Mat(u), _d.Vec(sig), Mat(v).T


def pinv(a, rcond=None) -> Mat:
    """jnp.linalg.pinv: V diag(1 / s_i where s_i > rcond * s_max, else 0) U^T, rcond = 10 * max(m, n) * eps by default."""
    a = as_mat(a)
    m, n = a.shape
    u, s, vh = svd(a)
    rc = 10.0 * max(m, n) * EPS if rcond is None else rcond
    cutoff = s.e[0] * rc
    s_inv = _d.Vec([_d._Np.where(e > cutoff, 1.0 / _d._Np.where(e > cutoff, e, 1.0), 0.0) for e in s.e])
    return (vh.T * s_inv) @ u.T                       # scale V's columns, then times U^T
