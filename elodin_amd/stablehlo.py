"""StableHLO text -> this package's scalar DAG (SURVEY §8 f1 in its literal form): the MLIR module the reference dumps before its
JIT (`ELODIN_CRANELIFT_DEBUG_DIR/stablehlo.mlir`, libs/nox-py/src/cranelift_compile.rs:47-68), or the text of any
`jax.jit(f).lower(...).as_text()`, read by the textual form the reference's own parser reads (libs/cranelift-mlir/src/parser.rs) and
turned into code of the SAME generated kernel the tracer feeds (elodin_amd.dsl -> codegen.py).  Three ways in:

* `system(text, inputs, outputs)` — a PER-ENTITY function: `@main`'s tensor arguments are one entity's components, every tensor a
  small static array whose elements become nodes.
* `world_system(text, slots, mode=...)` / `compile_world` / `python -m elodin_amd.stablehlo tick.mlir --slots slots.json -o pipe.so` —
  a WHOLE-WORLD tick, what the reference actually hands a backend: `@main` over entity-batched `[N, w]` columns.  mode "lane": the
  entity axis becomes the executor's rows (_LaneEval follows it through every statement; no `[N, ...]` tensor is ever materialised,
  65,536 bodies trace like twelve; constant-index gathers ALONG the entity axis — joins, an edge_fold's targets — become reads of the
  other lanes of the world's wavefront, dsl op `lane_read`, with a world laid out as `rows_per_world` consecutive rows; what no lane
  exchange covers is refused by reason); mode "world": the whole world in one lane,
  rows = independent worlds (edge folds, joins, reductions over the world are index arithmetic inside a lane); "auto" tries the
  former and falls back to the latter, recording why.  INTEGRATION.md §2b has the host side.
* `load_world(pipe.so)` — the CLI's object + manifest back as a program an executor installs as it is.

What is read: element-wise arithmetic / transcendentals / comparisons / bit operations, constant (decimal, hex bit patterns of floats,
hex byte blobs), iota, convert, bitcast_convert, select, clamp, broadcast_in_dim, reshape, transpose, slice, concatenate, reverse,
pad, dot_general (batching + contracting dims), reduce (`applies` form and reducer regions), while (unrolled when its trip count is
known while tracing, else a real loop), case, func.call, dynamic_slice, dynamic_update_slice, gather (the index-clamping general form),
sort (1-D), cholesky, triangular_solve, scatter, real_dynamic_slice, reduce_window, select_and_scatter, batch_norm_inference, map, the
LAPACK FFI custom calls jax.numpy.linalg lowers to on CPU (dpotrf, dtrsm, dgetrf with LAPACK's pivots and row order, dgesv, dgesdd,
dgeqrf / dorgqr, dsyevd), chlo.{erf_inv, square, acos, asin, sinh, cosh, erfc, ...}.  INTEGERS have the machine's semantics: results of
add / subtract / multiply / negate / shifts / convert wrap to the declared width for types of 32 bits or fewer, `ui64` is exact as two
uint32 words (U64) — jax.random's threefry rounds and its bits -> mantissa -> bitcast construction run bit for bit — and `i64` is one
node, exact below 2^53 (ticks, counters, indices, seeds; no wrap at 2^63).  `convolution` (N-D, strides, padding, dilations, feature
groups) is unrolled into multiply-adds and `rng` is the reference's deterministic fill.  Not read: fft (complex tensors), batch-grouped
convolutions, the remaining LAPACK calls — an unsupported op says which.

Pinned on the reference's own tests: libs/cranelift-mlir/tests/ops.rs (198 inline modules with asserted outputs ->
tests/golden/stablehlo_ops.json; the 22 not extracted are listed there with the reason), the world-tick fragments of
test_gather_3body / test_dynamic_ops_3body / test_while_dyn_slice / test_closed_call / test_threefry / test_threefry_e2e /
test_uniform_pipeline / test_sret_large.rs (-> tests/golden/stablehlo_world_fragments.json, 24 cases, integers compared exactly), and G1's 100
three-body ticks and G2's 100 ball ticks through assembled whole-world modules (tests/golden/hlo_world_builder.py) — CPU walker:
tests/test_stablehlo_ingest.py, tests/test_stablehlo_world.py; generated kernel: tests/test_gpu_stablehlo.py,
tests/test_gpu_stablehlo_world.py.
"""
from __future__ import annotations

import itertools
import re
from pathlib import Path
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import dsl as _dsl
from .dsl import Expr

_np = _dsl.np


# ---- types and values ---------------------------------------------------------------------------------------------------------

class TensorType:
    def __init__(self, shape: Tuple[int, ...], dtype: str):
        self.shape, self.dtype = tuple(int(s) for s in shape), dtype

    @property
    def size(self) -> int: return int(np.prod(self.shape)) if self.shape else 1
    def __repr__(self): return f"tensor<{'x'.join(map(str, self.shape + (self.dtype,)))}>"


def parse_type(text: str) -> TensorType:
    m = re.fullmatch(r"\s*tensor<([^>]*)>\s*", text)
    if not m:
        raise ValueError(f"not a ranked tensor type: {text!r}")
    parts = m.group(1).split("x")
    return TensorType(tuple(int(p) for p in parts[:-1]), parts[-1].strip())


class Sym:
    """A tensor of scalar nodes: an object ndarray of dsl.Expr (booleans for i1; U64 word pairs for ui64) + its element type.

    In entity-parallel ("lane") evaluation a Sym also says how its STORED array relates to the tensor's true shape `tshape`:
    along `eaxis` (the entity axis, true size N) and along every axis of `uni` (broadcast axes: all entries equal) the stored size
    is 1."""

    def __init__(self, arr, dtype: str, eaxis: Optional[int] = None, uni=frozenset(), tshape: Optional[Tuple[int, ...]] = None):
        self.a = arr if isinstance(arr, np.ndarray) and arr.dtype == object else _obj(arr, dtype)
        self.dtype = dtype
        self.eaxis, self.uni = eaxis, frozenset(uni)
        self.tshape = tuple(tshape) if tshape is not None else tuple(self.a.shape)

    @property
    def shape(self): return self.a.shape
    def is_int(self): return self.dtype[0] in "iu" and self.dtype != "i1"
    def is_bool(self): return self.dtype == "i1"


def _obj(values, dtype: str = "f64") -> np.ndarray:
    v = np.asarray(values, dtype=object) if not isinstance(values, np.ndarray) else values
    if dtype == "ui64":
        lift = lambda x: x if isinstance(x, U64) else (U64.from_float(x) if isinstance(x, Expr) else U64.from_int(int(x)))
    else:
        lift = lambda x: x if isinstance(x, Expr) else _dsl.const(float(x))
    out = np.empty(v.shape, dtype=object)
    for idx in np.ndindex(v.shape):
        out[idx] = lift(v[idx])
    if v.shape == ():
        x = v[()] if isinstance(values, np.ndarray) else values
        out = np.empty((), dtype=object)
        out[()] = lift(x)
    return out


def _emap(f: Callable, *arrs) -> np.ndarray:
    arrs = np.broadcast_arrays(*[a for a in arrs])
    out = np.empty(arrs[0].shape, dtype=object)
    for idx in np.ndindex(arrs[0].shape):
        out[idx] = f(*[a[idx] for a in arrs])
    if arrs[0].shape == ():
        out[()] = f(*[a[()] for a in arrs])
    return out


# ---- integer semantics -----------------------------------------------------------------------------------------------------------
# Integer tensors are integral values in float nodes (exact to 2^53 in a float64 program).  What the reference's lowering does
# with them it does in machine integers, so: results of add / subtract / multiply / negate / shift_left / convert WRAP to the
# declared width for every type of 32 bits or fewer (mod 2^bits, two's complement for signed types; a 32 x 32 product is formed
# from 16-bit halves so no intermediate leaves the exact range); `ui64` elements are TWO uint32 words (U64 below) and every
# operation on them is exact — jax.random builds its 52 random mantissa bits that way, (hi << 32 | lo) >> 12 | 0x3FF0..., then
# bitcast_convert (libs/cranelift-mlir/tests/test_uniform_pipeline.rs); `i64` stays one node: exact while |value| < 2^53 (ticks,
# counters, indices, seeds), no wrap at 2^63 — the one documented gap.

_TWO32 = 4294967296.0
_INV32 = 1.0 / _TWO32


def _bits(dtype: str) -> int:
    return int(re.sub(r"\D", "", dtype) or 64)


_CONST_FN1 = {"neg": lambda x: -x, "floor": np.floor, "ceil": np.ceil, "trunc": np.trunc, "abs": abs, "rint": np.rint, "not": lambda x: not x,
              "sqrt": lambda x: float(np.sqrt(x))}
_CONST_FN2 = {"add": lambda a, b: a + b, "sub": lambda a, b: a - b, "mul": lambda a, b: a * b, "div": lambda a, b: a / b if b else None,
              "max": max, "min": min, "lt": lambda a, b: a < b, "le": lambda a, b: a <= b, "eq": lambda a, b: a == b,
              "and": lambda a, b: bool(a) and bool(b), "or": lambda a, b: bool(a) or bool(b),
              "mod": lambda a, b: float(np.mod(a, b)) if b else None,
              "bxor": lambda a, b: float(int(a) ^ int(b)), "bor": lambda a, b: float(int(a) | int(b)), "band": lambda a, b: float(int(a) & int(b)),
              "shl": lambda a, b: float(int(a) << int(b)), "shr": lambda a, b: float((int(a) & 0xFFFFFFFFFFFFFFFF) >> int(b))}
_const_memo: Dict[int, tuple] = {}      # id(Expr) -> (Expr kept alive, value | None)


def _try_const(e):
    """The Python value (float / bool) of a node that depends on no leaf, else None.  Iterative, memoised per node."""
    if not isinstance(e, Expr):
        return e if isinstance(e, (bool, int, float)) else None
    hit = _const_memo.get(id(e))
    if hit is not None:
        return hit[1]
    stack = [e]
    while stack:
        x = stack[-1]
        if id(x) in _const_memo:
            stack.pop()
            continue
        if x.op == "const":
            _const_memo[id(x)] = (x, x.value)
            stack.pop()
            continue
        if x.op not in _CONST_FN1 and x.op not in _CONST_FN2 and x.op != "select":
            _const_memo[id(x)] = (x, None)
            stack.pop()
            continue
        todo = [a for a in x.args if id(a) not in _const_memo]
        if todo:
            stack.extend(todo)
            continue
        vals = [_const_memo[id(a)][1] for a in x.args]
        stack.pop()
        if x.op == "select":
            v = None if vals[0] is None else (vals[1] if vals[0] else vals[2])
        elif any(v is None for v in vals):
            v = None
        else:
            try:
                v = _CONST_FN1[x.op](*vals) if x.op in _CONST_FN1 else _CONST_FN2[x.op](*vals)
                v = bool(v) if isinstance(v, (bool, np.bool_)) else (None if v is None else float(v))
            except (ValueError, OverflowError, ZeroDivisionError):
                v = None
        _const_memo[id(x)] = (x, v)
    return _const_memo[id(e)][1]


def _fold(e):
    """A float node replaced by its constant when it has one (keeps unrolled loop counters and their index arithmetic out of the DAG)."""
    if isinstance(e, Expr) and e.op != "const":
        v = _try_const(e)
        if v is not None and not isinstance(v, bool):
            return _dsl.const(v)
    return e


def _floor_mod(x, m: float):
    """x mod m for an integral x and a power of two m (exact: the reciprocal of m is exact, the product below |x|)."""
    return _fold(x - _np.floor(x * (1.0 / m)) * m)


def _wrap(x, dtype: str):
    """An integral value folded into the range of an integer type of 32 bits or fewer (two's complement)."""
    if dtype[0] not in "iu" or dtype == "i1":
        return x
    bits = _bits(dtype)
    if bits >= 64:
        return x
    m = float(2 ** bits)
    if dtype[0] == "u":
        return _floor_mod(x, m)
    return _fold(_floor_mod(x + m / 2, m) - m / 2)


def _mul_lo32(a, b):
    """Low 32 bits of the product of two values in [0, 2^32): 16-bit halves, every intermediate below 2^49."""
    a1 = _np.floor(a * (1.0 / 65536.0))
    a0 = a - a1 * 65536.0
    t = _floor_mod(a1 * b, _TWO32)
    return _floor_mod(t * 65536.0 + a0 * b, _TWO32)


def _shl32(x, k, bits: int = 32):
    """(x << k) mod 2^bits for x in [0, 2^bits), 0 <= k < bits: the bits that would leave the word are masked off FIRST, so the
    shifted value never exceeds the exact range of the float carrier."""
    kc = _try_const(k)
    if kc is not None:
        kc = int(kc)
        return x if kc == 0 else _fold(_np.left_shift(_np.bitwise_and(x, float((1 << (bits - kc)) - 1)), float(kc)))
    mask = _np.left_shift(1.0, float(bits) - k) - 1.0
    return _np.left_shift(_np.bitwise_and(x, mask), k)


def _where(c, a, b):
    """Element select that folds a constant condition and reaches into ui64 elements."""
    k = _try_const(c)
    if k is not None:
        return a if k else b
    if isinstance(a, U64) or isinstance(b, U64):
        a, b = U64.of(a), U64.of(b)
        return U64(_np.where(c, a.hi, b.hi), _np.where(c, a.lo, b.lo))
    if a is b:
        return a
    return _np.where(c, a, b)


class NotEntityParallel(NotImplementedError):
    """A whole-world module moves data between entities in a way one-lane-per-entity evaluation cannot follow."""


class _KindsChanged(Exception):
    pass


class U64:
    """One ui64 element: two uint32 words, each an integral float node."""
    __slots__ = ("hi", "lo")

    def __init__(self, hi, lo):
        self.hi, self.lo = _fold(_dsl._lift(hi)), _fold(_dsl._lift(lo))

    @staticmethod
    def of(v) -> "U64":
        return v if isinstance(v, U64) else U64.from_float(v)

    @staticmethod
    def from_int(v: int) -> "U64":
        v &= 0xFFFFFFFFFFFFFFFF
        return U64(float(v >> 32), float(v & 0xFFFFFFFF))

    @staticmethod
    def from_float(x) -> "U64":
        """An integral float (possibly negative: two's complement) as words."""
        x = _dsl._lift(x)
        k = _try_const(x)
        if k is not None:
            return U64.from_int(int(k))
        hi = _np.floor(x * _INV32)
        return U64(_floor_mod(hi, _TWO32), x - hi * _TWO32)

    def to_float(self):
        return _fold(self.hi * _TWO32 + self.lo)

    # element-wise operations -------------------------------------------------------------------------------------------------
    def bitwise(self, o: "U64", op) -> "U64": return U64(op(self.hi, o.hi), op(self.lo, o.lo))
    def invert(self) -> "U64": return U64(4294967295.0 - self.hi, 4294967295.0 - self.lo)

    def add(self, o: "U64") -> "U64":
        lo = self.lo + o.lo
        carry = _np.floor(lo * _INV32)
        return U64(_floor_mod(self.hi + o.hi + carry, _TWO32), lo - carry * _TWO32)

    def sub(self, o: "U64") -> "U64":
        lo = self.lo - o.lo
        borrow = _np.floor(lo * _INV32)                        # 0 or -1
        return U64(_floor_mod(self.hi - o.hi + borrow, _TWO32), lo - borrow * _TWO32)

    def mul(self, o: "U64") -> "U64":
        a1, b1 = _np.floor(self.lo * (1.0 / 65536.0)), _np.floor(o.lo * (1.0 / 65536.0))
        a0, b0 = self.lo - a1 * 65536.0, o.lo - b1 * 65536.0
        p0, p1, p2 = a0 * b0, a1 * b0 + a0 * b1, a1 * b1           # < 2^32, < 2^33, < 2^32
        p1h = _np.floor(p1 * (1.0 / 65536.0))
        lo = p0 + (p1 - p1h * 65536.0) * 65536.0                    # < 2^33
        carry = _np.floor(lo * _INV32)
        hi = p2 + p1h + carry + _mul_lo32(self.hi, o.lo) + _mul_lo32(self.lo, o.hi)
        return U64(_floor_mod(hi, _TWO32), lo - carry * _TWO32)

    def shl(self, s) -> "U64":
        k = _try_const(s)
        if k is not None:
            k = int(k)
            if k >= 64:
                return U64(0.0, 0.0)
            if k == 0:
                return self
            if k >= 32:
                return U64(_shl32(self.lo, float(k - 32)), 0.0)
            return U64(_shl32(self.hi, float(k)) + _np.right_shift(self.lo, float(32 - k)),
                       _shl32(self.lo, float(k)))
        big = _dsl._lift(s) >= 32.0
        sb, ss = _np.clip(s - 32.0, 0.0, 31.0), _np.clip(s, 0.0, 31.0)
        hi_small = _shl32(self.hi, ss) + _np.where(_np.equal(ss, 0.0), 0.0, _np.right_shift(self.lo, 32.0 - ss))
        out = U64(_np.where(big, _shl32(self.lo, sb), hi_small),
                  _np.where(big, 0.0, _shl32(self.lo, ss)))
        return _where(_dsl._lift(s) >= 64.0, U64(0.0, 0.0), out)

    def shr(self, s) -> "U64":
        k = _try_const(s)
        if k is not None:
            k = int(k)
            if k >= 64:
                return U64(0.0, 0.0)
            if k == 0:
                return self
            if k >= 32:
                return U64(0.0, _np.right_shift(self.hi, float(k - 32)))
            return U64(_np.right_shift(self.hi, float(k)),
                       _np.right_shift(self.lo, float(k)) + _shl32(self.hi, float(32 - k)))
        big = _dsl._lift(s) >= 32.0
        sb, ss = _np.clip(s - 32.0, 0.0, 31.0), _np.clip(s, 0.0, 31.0)
        lo_small = _np.right_shift(self.lo, ss) + _np.where(_np.equal(ss, 0.0), 0.0, _shl32(self.hi, 32.0 - ss))
        out = U64(_np.where(big, 0.0, _np.right_shift(self.hi, ss)), _np.where(big, _np.right_shift(self.hi, sb), lo_small))
        return _where(_dsl._lift(s) >= 64.0, U64(0.0, 0.0), out)

    def compare(self, o: "U64", d: str):
        eq = _np.logical_and(_np.equal(self.hi, o.hi), _np.equal(self.lo, o.lo))
        lt = _np.logical_or(self.hi < o.hi, _np.logical_and(_np.equal(self.hi, o.hi), self.lo < o.lo))
        return {"EQ": eq, "NE": _np.logical_not(eq), "LT": lt, "LE": _np.logical_or(lt, eq),
                "GT": _np.logical_not(_np.logical_or(lt, eq)), "GE": _np.logical_not(lt)}[d]


def _u64_binary(short: str):
    if short in ("and", "or", "xor"):
        f = {"and": _np.bitwise_and, "or": _np.bitwise_or, "xor": _np.bitwise_xor}[short]
        return lambda x, y: U64.of(x).bitwise(U64.of(y), f)
    if short == "add":
        return lambda x, y: U64.of(x).add(U64.of(y))
    if short == "subtract":
        return lambda x, y: U64.of(x).sub(U64.of(y))
    if short == "multiply":
        return lambda x, y: U64.of(x).mul(U64.of(y))
    if short == "shift_left":
        return lambda x, y: U64.of(x).shl(U64.of(y).lo if isinstance(y, U64) else y)      # amounts of 2^32 and beyond do not occur
    if short == "shift_right_logical":
        return lambda x, y: U64.of(x).shr(U64.of(y).lo if isinstance(y, U64) else y)
    if short in ("maximum", "minimum"):
        return lambda x, y: _where(U64.of(x).compare(U64.of(y), "LT" if short == "minimum" else "GT"), U64.of(x), U64.of(y))
    raise NotImplementedError(f"stablehlo.{short} on ui64 tensors")


# ---- parsing --------------------------------------------------------------------------------------------------------------------

class Op:
    def __init__(self, results, name, text, regions=None, region_args=None):
        self.results, self.name, self.text = results, name, text
        self.regions: List[List["Op"]] = regions or []
        self.region_args: List[List[Tuple[str, TensorType]]] = region_args or []


class Func:
    def __init__(self, name, args, result_types, body):
        self.name, self.args, self.result_types, self.body = name, args, result_types, body


_HEAD = re.compile(r"^(?:(%[\w#.]+(?::\d+)?(?:\s*,\s*%[\w#.]+)*)\s*=\s*)?(\"?[\w.]+\"?)(.*)$", re.S)


def _split_top(text: str, sep: str = ",") -> List[str]:
    parts, depth, cur = [], 0, ""
    for ch in text:
        if ch in "([{<":
            depth += 1
        elif ch in ")]}>":
            depth -= 1
        if ch == sep and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur)
    return [p.strip() for p in parts]


def _typed_args(text: str) -> List[Tuple[str, TensorType]]:
    out = []
    for p in _split_top(text):
        if not p:
            continue
        name, _, ty = p.partition(":")
        ty = re.sub(r"\s*\{.*\}\s*$", "", ty.strip())          # argument attributes
        out.append((name.strip(), parse_type(ty)))
    return out


class _Lines:
    def __init__(self, text):
        raw = [ln.strip() for ln in text.splitlines()]
        raw = [ln for ln in raw if ln and not ln.startswith("//")]
        # logical lines: a statement (or a function header) may wrap — while its parentheses / brackets are open, or it ends
        # in `,` / `:` — but a line that OPENS a region (`({`, `{`) is complete
        self.lines, cur = [], ""
        for ln in raw:
            cur = (cur + " " + ln) if cur else ln
            open_ = cur.count("(") - cur.count(")") + cur.count("[") - cur.count("]")
            if cur.count("<{") > cur.count("}>"):          # inside a generic op's property dictionary `<{ ... }>`: not a region
                continue
            if cur.endswith(("({", "{")) or cur.startswith("^bb") or (open_ <= 0 and not cur.endswith((",", ":"))):
                self.lines.append(cur)
                cur = ""
        if cur:
            self.lines.append(cur)
        self.i = 0

    def peek(self): return self.lines[self.i] if self.i < len(self.lines) else None
    def next(self):
        ln = self.lines[self.i]
        self.i += 1
        return ln


def _parse_block(L: _Lines) -> Tuple[List[Op], str]:
    """Ops until a line that closes the block; returns (ops, the closing line)."""
    ops: List[Op] = []
    while True:
        ln = L.next()
        if ln.startswith("}"):
            return ops, ln
        if ln.startswith("^bb"):                              # block header inside a generic region: handled by the caller
            ops.append(Op([], "^bb", ln))
            continue
        while L.peek() is not None and L.peek()[0] in "(:-" and not ln.rstrip().endswith("({"):      # a signature on its own line
            ln = ln + " " + L.next()
        m = _HEAD.match(ln)
        if not m:
            raise ValueError(f"cannot parse statement: {ln!r}")
        res, name, rest = m.group(1), m.group(2).strip('"'), m.group(3)
        results = []
        if res:
            for r in _split_top(res):
                base, _, count = r.partition(":")
                results += [f"{base}#{k}" for k in range(int(count))] if count else [base]
                if count:
                    results.insert(len(results) - int(count), base)      # %0:2 also answers to %0 (= #0)
        op = Op(results, name, rest.strip())
        text = rest
        # regions
        if name == "stablehlo.while":
            while L.peek() is not None and re.match(r"^(\}\s*)?(cond|do)\s*\{", L.peek()):
                L.next()
                body, closing = _parse_block(L)
                op.regions.append(body)
                while re.match(r"^\}\s*do\s*\{", closing):
                    body, closing = _parse_block(L)
                    op.regions.append(body)
        elif text.rstrip().endswith("({"):
            while True:
                body, closing = _parse_block(L)
                args = []
                if body and body[0].name == "^bb":
                    hdr = body.pop(0).text
                    args = _typed_args(hdr[hdr.index("(") + 1: hdr.rindex(")")])
                op.regions.append(body)
                op.region_args.append(args)
                if re.match(r"^\}\s*,\s*\{", closing):
                    continue
                op.text = text + " " + closing
                break
        elif L.peek() is not None and L.peek().startswith("reducer("):
            hdr = L.next()
            args = _typed_args(hdr[hdr.index("(") + 1: hdr.rindex(")")])
            body, _ = _parse_block(L)
            op.regions.append(body)
            op.region_args.append(args)
        ops.append(op)


def parse_module(text: str) -> Dict[str, Func]:
    L = _Lines(text)
    funcs: Dict[str, Func] = {}
    while L.peek() is not None:
        ln = L.next()
        m = re.match(r"^func\.func\s+(?:public\s+|private\s+)?@([\w.$-]+)\(", ln)
        if not m or not ln.endswith("{"):
            continue
        depth, k0 = 0, m.end() - 1
        for k in range(k0, len(ln)):                       # the balanced argument list
            depth += ln[k] == "("
            depth -= ln[k] == ")"
            if depth == 0:
                break
        args_text, tail = ln[k0 + 1:k], ln[k + 1:-1].strip()
        tail = re.sub(r"attributes\s*\{.*\}\s*$", "", tail).strip()
        res = tail[2:].strip() if tail.startswith("->") else ""
        if res.startswith("("):
            res = res[1:res.rindex(")")]
        rtypes = [parse_type(re.sub(r"\s*\{.*\}\s*$", "", r)) for r in _split_top(res)] if res else []
        body, _ = _parse_block(L)
        funcs[m.group(1)] = Func(m.group(1), _typed_args(args_text), rtypes, body)
    if "main" not in funcs:
        raise ValueError("module has no @main")
    return funcs


# ---- evaluation ---------------------------------------------------------------------------------------------------------------

_UNARY = {"negate": lambda x: -x, "sqrt": _np.sqrt, "rsqrt": lambda x: 1.0 / _np.sqrt(x), "exponential": _np.exp, "log": _np.log,
          "sine": _np.sin, "cosine": _np.cos, "tan": _np.tan, "tanh": _np.tanh, "abs": _np.abs, "sign": _np.sign, "floor": _np.floor,
          "ceil": _np.ceil, "log_plus_one": _np.log1p, "exponential_minus_one": _np.expm1, "expm1": _np.expm1, "cbrt": _np.cbrt,
          "round_nearest_even": _np.rint, "atan": _np.arctan}
_CHLO = {"erf_inv": lambda x: Expr("erfinv", (_dsl._lift(x),)), "square": lambda x: x * x, "acos": _np.arccos, "asin": _np.arcsin,
         "sinh": _np.sinh, "cosh": _np.cosh, "erfc": _np.erfc, "atan": _np.arctan, "tan": _np.tan, "erf": lambda x: 1.0 - _np.erfc(x)}
_CMP = {"EQ": lambda a, b: _np.equal(a, b), "NE": lambda a, b: _np.logical_not(_np.equal(a, b)), "LT": lambda a, b: a < b,
        "LE": lambda a, b: a <= b, "GT": lambda a, b: a > b, "GE": lambda a, b: a >= b}


def _trunc_div(a, b):                              # integer division: toward zero; by zero -> 0 like the reference's lowering
    return _np.where(_np.equal(b, 0.0), 0.0, _np.trunc(a / _np.where(_np.equal(b, 0.0), 1.0, b)))      # (ops.rs test_div_i64_mem)


def _rem(a, b):                                    # stablehlo.remainder: the sign of the DIVIDEND (C's fmod / %)
    r = _np.remainder(_np.abs(a), _np.abs(b))      # both non-negative: the floor-mod node IS fmod (exact, whatever the ratio)
    return _np.where(_dsl._lift(a) < 0.0, -r, r)


class _Eval:
    def __init__(self, funcs: Dict[str, Func]):
        self.funcs = funcs
        self._rt_override: Optional[List[TensorType]] = None      # lane mode: the STORED result types of the next _op call
        self._ints_override: Dict[str, List[int]] = {}            # lane mode: attribute lists with the entity axis at size 1
        self._slice_override = None

    # -- plumbing --
    def call(self, fn: Func, args: Sequence[Sym]) -> List[Sym]:
        env = {name: a for (name, _), a in zip(fn.args, args)}
        return self.block(fn.body, env)

    def block(self, ops: List[Op], env: Dict[str, Sym]) -> List[Sym]:
        env = dict(env)
        for op in ops:
            if op.name in ("return", "stablehlo.return", "func.return"):
                names = _split_top(op.text.split(":")[0]) if op.text.strip() else []
                return [env[n] for n in names]
            outs = self.op(op, env)
            if len(op.results) == 1:
                env[op.results[0]] = outs[0]
            else:
                plain = [r for r in op.results if "#" in r]
                for r, o in zip(plain, outs):
                    env[r] = o
                if plain:
                    env[plain[0].split("#")[0]] = outs[0]
        return []

    def _operands(self, text: str, env) -> List[Sym]:
        return [env[n] for n in re.findall(r"%[\w#.]+", text)]

    @staticmethod
    def _result_types(text: str) -> List[TensorType]:
        sig = text[text.rindex(":") + 1:] if ":" in text else ""
        # the type signature is what follows the LAST top-level colon
        depth, cut = 0, None
        for k, ch in enumerate(text):
            if ch in "([{<":
                depth += 1
            elif ch in ")]}>":
                depth -= 1
            elif ch == ":" and depth == 0:
                cut = k
        sig = text[cut + 1:].strip() if cut is not None else ""
        if cut is None and "->" in text:                       # custom_call: `{attributes} -> (types)` without a colon
            sig = "->" + text.rsplit("->", 1)[1]
        if "->" in sig:
            sig = sig.split("->")[-1].strip()
        if sig.startswith("("):
            sig = sig[1:-1]
        return [parse_type(t) for t in _split_top(sig) if t.startswith("tensor<")]

    def _ints(self, text: str, key: str) -> List[int]:
        if key in self._ints_override:
            return list(self._ints_override[key])
        m = re.search(re.escape(key) + r"\s*=\s*(?:array<i64(?::\s*([^>]*))?>|\[([^\]]*)\])", text)
        if not m:
            return []
        body = m.group(1) if m.group(1) is not None else (m.group(2) or "")
        return [int(x) for x in body.replace(" ", "").split(",") if x]

    # -- ops --
    def op(self, op: Op, env) -> List[Sym]:
        return self._op(op, env)

    def _op(self, op: Op, env) -> List[Sym]:
        name, text = op.name, op.text
        short = name.split(".", 1)[1] if "." in name else name
        if name in ("call", "func.call"):
            m = re.match(r"\s*@([\w.$-]+)\((.*?)\)\s*:", text)
            return self.call(self.funcs[m.group(1)], [env[n] for n in re.findall(r"%[\w#.]+", m.group(2))])
        rts = self._result_types(text)
        if self._rt_override is not None:
            rts, self._rt_override = self._rt_override, None
        rt = rts[0] if rts else None
        if short == "constant":
            return [self._constant(text, rt)]
        if short == "iota":
            dim = int(re.search(r"dim\s*=\s*(\d+)", text).group(1))
            idx = np.indices(rt.shape)[dim] if rt.shape else np.zeros(())
            return [Sym(_obj(idx.astype(float), rt.dtype), rt.dtype)]
        args_text = text
        for r in (op.regions and [""] or []):
            pass
        head = text.split(":")[0] if name.startswith("chlo") or short in _UNARY else text
        if short == "while":
            return self._while(op, text, env)
        xs = self._operands(self._operand_text(text), env)
        if name.startswith("chlo."):
            return [Sym(_emap(_CHLO[short], xs[0].a), rt.dtype)]
        if short in _UNARY:
            x = xs[0]
            if x.dtype == "ui64":
                if short != "negate":
                    raise NotImplementedError(f"stablehlo.{short} on ui64 tensors")
                return [Sym(_emap(lambda v: U64(0.0, 0.0).sub(v), x.a), x.dtype)]
            if short == "abs" and x.is_int():
                return [Sym(_emap(_np.abs, x.a), x.dtype)]
            if short == "negate" and x.is_int():
                return [Sym(_emap(lambda v: _wrap(-v, x.dtype), x.a), x.dtype)]
            return [Sym(_emap(_UNARY[short], x.a), x.dtype)]
        if short == "not":
            x = xs[0]
            if x.is_bool():
                return [Sym(_emap(_np.logical_not, x.a), "i1")]
            if x.dtype == "ui64":
                return [Sym(_emap(lambda v: v.invert(), x.a), x.dtype)]
            bits = _bits(x.dtype)
            return [Sym(_emap(lambda v: (2.0 ** bits - 1.0) - v if x.dtype[0] == "u" else -v - 1.0, x.a), x.dtype)]
        if short == "bitcast_convert":
            return [self._bitcast(xs[0], rt)]
        if short in ("popcnt", "count_leading_zeros"):
            raise NotImplementedError(f"StableHLO op {name} is not provided by elodin_amd.stablehlo")
        if short == "is_finite":
            return [Sym(_emap(lambda v: Expr("isfinite", (_dsl._lift(v),)), xs[0].a), "i1")]
        if short in ("add", "subtract", "multiply", "divide", "maximum", "minimum", "power", "atan2", "remainder", "and", "or", "xor",
                     "shift_left", "shift_right_logical", "shift_right_arithmetic"):
            a, b = xs[0], xs[1]
            return [Sym(_emap(self._binary(short, a), a.a, b.a), a.dtype)]
        if short == "compare":
            d = re.search(r"\b(EQ|NE|LT|LE|GT|GE)\b", text).group(1)
            if xs[0].dtype == "ui64":
                return [Sym(_emap(lambda x, y: U64.of(x).compare(U64.of(y), d), xs[0].a, xs[1].a), "i1")]
            return [Sym(_emap(_CMP[d], xs[0].a, xs[1].a), "i1")]
        if short == "select":
            c, a, b = xs
            return [Sym(_emap(_where, c.a, a.a, b.a), a.dtype)]
        if short == "clamp":
            lo, x, hi = xs
            if x.dtype == "ui64":
                raise NotImplementedError("stablehlo.clamp on ui64 tensors")
            return [Sym(_emap(lambda l, v, h: _np.minimum(_np.maximum(v, l), h), lo.a, x.a, hi.a), x.dtype)]
        if short == "convert":
            return [self._convert(xs[0], rt.dtype)]
        if short == "broadcast_in_dim":
            dims = self._ints(text, "dims") or self._ints(text, "broadcast_dimensions")
            x = xs[0]
            shape = [1] * len(rt.shape)
            for src, dst in enumerate(dims):
                shape[dst] = x.shape[src]
            return [Sym(np.broadcast_to(x.a.reshape(shape), rt.shape).copy(), x.dtype)]
        if short == "reshape":
            return [Sym(xs[0].a.reshape(rt.shape), xs[0].dtype)]
        if short == "transpose":
            perm = self._ints(text, "dims") or self._ints(text, "permutation")
            return [Sym(np.transpose(xs[0].a, perm).copy(), xs[0].dtype)]
        if short == "reverse":
            dims = self._ints(text, "dims") or self._ints(text, "dimensions")
            return [Sym(np.flip(xs[0].a, axis=tuple(dims)).copy(), xs[0].dtype)]
        if short == "slice":
            sl = [slice(*t) for t in (self._slice_override or self._slice_ranges(text))]
            return [Sym(xs[0].a[tuple(sl)].copy(), xs[0].dtype)]
        if short == "concatenate":
            dim = int(re.search(r"dim(?:ension)?\s*=\s*(\d+)", text).group(1))
            return [Sym(np.concatenate([x.a for x in xs], axis=dim), xs[0].dtype)]
        if short == "dot_general":
            return [self._dot_general(xs[0], xs[1], text, rt)]
        if short == "reduce":
            return self._reduce(op, xs, text, rts, env)
        if short == "while":
            return self._while(op, text, env)
        if short == "case":
            return self._case(op, xs[0], env)
        if short == "dynamic_slice":
            sizes = self._ints(text, "sizes") or self._ints(text, "slice_sizes")
            return [self._dynamic_slice(xs[0], xs[1:], sizes)]
        if short == "dynamic_update_slice":
            return [self._dynamic_update_slice(xs[0], xs[1], xs[2:])]
        if short == "gather":
            return [self._gather(xs[0], xs[1], text, rt)]
        if short == "sort":
            return [self._sort(op, xs[0], text)]
        if short == "cholesky":
            from . import dsl_mat
            lower = "lower = true" in text
            rows = [_dsl.Vec(list(r)) for r in xs[0].a]
            Lm = dsl_mat.cholesky(dsl_mat.Mat(rows), lower=lower)
            return [Sym(np.array([[e for e in r.e] for r in Lm], dtype=object), xs[0].dtype)]
        if short == "custom_call":
            return self._custom_call(text, xs, rts)
        if short == "triangular_solve":
            from . import dsl_mat
            if "left_side = true" not in text:
                raise NotImplementedError("stablehlo.triangular_solve: left_side = false")
            lower, unit = "lower = true" in text, "unit_diagonal = true" in text
            trans = 1 if re.search(r"transpose_a\s*=\s*#stablehlo<transpose\s+(TRANSPOSE|ADJOINT)>", text) else 0
            A = dsl_mat.Mat([_dsl.Vec(list(r)) for r in xs[0].a])
            B = xs[1].a
            cols = []
            for j in range(B.shape[1]):
                cols.append(dsl_mat.solve_triangular(A, _dsl.Vec(list(B[:, j])), lower=lower, trans=trans, unit_diagonal=unit))
            return [Sym(np.array([[cols[j].e[i] for j in range(B.shape[1])] for i in range(B.shape[0])], dtype=object), xs[1].dtype)]
        if short == "scatter":
            return self._scatter(op, xs, text, env)
        if short == "real_dynamic_slice":
            return [self._real_dynamic_slice(xs[0], xs[1], xs[3], rt)]
        if short == "reduce_window":
            return self._reduce_window(op, xs, text, rts, env)
        if short == "select_and_scatter":
            return [self._select_and_scatter(op, xs, text, env)]
        if short == "pad":
            return [self._pad(xs[0], xs[1], text, rt)]
        if short == "batch_norm_inference":      # (x - mean) / sqrt(variance + epsilon) * scale + offset along feature_index
            x, scale, offset, mean, var = xs
            eps = float(re.search(r"epsilon\s*=\s*([-+0-9.eE]+)", text).group(1))
            feat = int(re.search(r"feature_index\s*=\s*(\d+)", text).group(1))
            shp = [1] * x.a.ndim
            shp[feat] = x.shape[feat]
            b = lambda v: v.a.reshape(shp)
            return [Sym(_emap(lambda xv, s_, o, m_, v_: (xv - m_) / _np.sqrt(v_ + eps) * s_ + o, x.a, b(scale), b(offset), b(mean), b(var)), x.dtype)]
        if short == "convolution":
            return [self._convolution(xs[0], xs[1], text, rt)]
        if short == "rng":
            # the reference's stablehlo.rng is a DETERMINISTIC fill, not a generator (libs/cranelift-mlir/src/tensor_rt.rs:2103-2140,
            # ARCHITECTURE.md:916; jax.random lowers to threefry instead, which is read bit for bit): UNIFORM = n values linearly spaced
            # from a to b inclusive (0.5 of the way for n = 1), NORMAL = the midpoints (i + 0.5) / n of the same span.  Same values here.
            a, b = _dsl._lift(xs[0].a.reshape(-1)[0]), _dsl._lift(xs[1].a.reshape(-1)[0])
            normal = re.search(r"rng_distribution\s+NORMAL", text) is not None
            n = rt.size
            ts = [((i + 0.5) / n if normal else (i / (n - 1) if n > 1 else 0.5)) for i in range(n)]
            out = np.empty(n, dtype=object)
            for i, t in enumerate(ts):
                out[i] = a + t * (b - a)
            return [Sym(out.reshape(rt.shape), rt.dtype)]
        if short == "map":
            region = self._region_fn(op, 0, env)
            out = np.empty(xs[0].shape, dtype=object)
            for idx in (np.ndindex(xs[0].shape) if xs[0].shape else [()]):
                out[idx] = region(*[Sym(_obj_scalar(x.a[idx]), x.dtype) for x in xs])[0].a[()]
            return [Sym(out, rt.dtype)]
        raise NotImplementedError(f"StableHLO op {name} is not provided by elodin_amd.stablehlo")

    def _convolution(self, lhs: Sym, rhs: Sym, text: str, rt: TensorType) -> Sym:
        """stablehlo.convolution over static shapes, unrolled into multiply-adds: N spatial dimensions, window strides, padding, lhs /
        rhs dilation, feature groups, any dimension_numbers `#stablehlo.conv<[b, 0, f]x[0, i, o]->[b, 0, f]>`; batch_group_count = 1.  A cross-correlation (the kernel is not flipped); a window position that falls into the padding or between the dilated
        input samples contributes nothing.  (libs/cranelift-mlir/ARCHITECTURE.md:910; the reference's own known answer for it,
        ops.rs:4211-4230, is #[ignore]d there — it passes here.)"""
        # generic form: dimension_numbers = #stablehlo.conv<[b, 0, f]x[0, i, o]->[b, 0, f]>, window_strides = array<i64: 1>, padding = dense<...>;
        # pretty form (what jax dumps): dim_numbers = [b, 0, f]x[0, i, o]->[b, 0, f], window = {stride = [1], pad = [[0, 0]], lhs_dilate = [1], ...}
        m = re.search(r"(?:#stablehlo\.conv<\s*(?:raw\s*)?|dim_numbers\s*=\s*)\[([^\]]*)\]\s*x\s*\[([^\]]*)\]\s*->\s*\[([^\]]*)\]", text)
        if not m:
            raise NotImplementedError("stablehlo.convolution without [...]x[...]->[...] dimension numbers")
        ld, kd, od = ([t.strip() for t in g.split(",")] for g in m.groups())
        ns = len(ld) - 2
        g = re.search(r"batch_group_count\s*=\s*(\d+)", text)
        if g and int(g.group(1)) != 1:
            raise NotImplementedError(f"stablehlo.convolution with batch_group_count = {g.group(1)}")
        g = re.search(r"feature_group_count\s*=\s*(\d+)", text)
        groups = int(g.group(1)) if g else 1
        if re.search(r"window_reversal\s*=\s*(?:array<i1:[^>]*true|dense<[^>]*true)", text):
            raise NotImplementedError("stablehlo.convolution with window_reversal")
        win = re.search(r"window\s*=\s*\{(.*?)\}\s*(?:\{|:)", text, re.S)
        if win:
            def wlist(key, default):
                g = re.search(key + r"\s*=\s*\[([^\]]*)\]", win.group(1))
                return [int(v) for v in g.group(1).split(",")] if g and g.group(1).strip() else [default] * ns
            stride, ldil, rdil = wlist("stride", 1), wlist("lhs_dilate", 1), wlist("rhs_dilate", 1)
            g = re.search(r"pad\s*=\s*\[((?:\s*\[[^\]]*\]\s*,?)*)\]", win.group(1))
            vals = [int(v) for v in re.findall(r"-?\d+", g.group(1))] if g else []
            pad = [(vals[2 * d], vals[2 * d + 1]) for d in range(ns)] if vals else [(0, 0)] * ns
            if re.search(r"reverse\s*=\s*\[[^\]]*true", win.group(1)):
                raise NotImplementedError("stablehlo.convolution with window reversal")
        else:
            stride = self._window_attr(text, "window_strides", ns, 1)
            ldil = self._window_attr(text, "lhs_dilation", ns, 1)
            rdil = self._window_attr(text, "rhs_dilation", ns, 1)
            pad = self._padding_attr(text, ns)
        sp = [str(k) for k in range(ns)]
        lb, lf, ls = ld.index("b"), ld.index("f"), [ld.index(k) for k in sp]
        ki, ko, ks = kd.index("i"), kd.index("o"), [kd.index(k) for k in sp]
        ob, of, os_ = od.index("b"), od.index("f"), [od.index(k) for k in sp]
        B, Cin, Cout = lhs.shape[lb], lhs.shape[lf], rhs.shape[ko]
        if rhs.shape[ki] * groups != Cin or Cout % groups:
            raise ValueError("stablehlo.convolution: the kernel's input-feature size times feature_group_count differs from the operand's")
        cin_g, cout_g = Cin // groups, Cout // groups        # feature groups (depthwise: groups = Cin): output feature o reads group o // cout_g
        in_sz = [lhs.shape[a] for a in ls]
        k_sz = [rhs.shape[a] for a in ks]
        out_sz = [rt.shape[a] for a in os_]
        out = np.empty(rt.shape, dtype=object)
        for b in range(B):
            for o in range(Cout):
                for opos in (np.ndindex(*out_sz) if ns else [()]):
                    acc = None
                    for kpos in (np.ndindex(*k_sz) if ns else [()]):
                        src, ok = [], True
                        for d in range(ns):      # position in the dilated + padded input -> the input sample it is, if any
                            p = opos[d] * stride[d] + kpos[d] * rdil[d] - pad[d][0]
                            if p < 0 or p % ldil[d] or p // ldil[d] >= in_sz[d]:
                                ok = False
                                break
                            src.append(p // ldil[d])
                        if not ok:
                            continue
                        for c in range(cin_g):
                            li = [0] * lhs.a.ndim
                            li[lb], li[lf] = b, (o // cout_g) * cin_g + c
                            ri = [0] * rhs.a.ndim
                            ri[ki], ri[ko] = c, o
                            for d in range(ns):
                                li[ls[d]], ri[ks[d]] = src[d], kpos[d]
                            term = _dsl._lift(lhs.a[tuple(li)]) * _dsl._lift(rhs.a[tuple(ri)])
                            acc = term if acc is None else acc + term
                    oi = [0] * len(rt.shape)
                    oi[ob], oi[of] = b, o
                    for d in range(ns):
                        oi[os_[d]] = opos[d]
                    out[tuple(oi)] = acc if acc is not None else _dsl._lift(0.0)
        return Sym(out, lhs.dtype)

    @staticmethod
    def _slice_ranges(text: str) -> List[Tuple[int, int, int]]:
        m = re.search(r"\[([^\]]*)\]\s*:", text)
        out = []
        for part in m.group(1).split(","):
            p = [int(v) for v in part.strip().split(":")]
            out.append((p[0], p[1], p[2] if len(p) > 2 else 1))
        return out

    @staticmethod
    def _operand_text(text: str) -> str:
        """The part of a statement that names its operands: everything before the type signature / attribute dictionary."""
        depth, cut = 0, len(text)
        for k, ch in enumerate(text):
            if ch in "([<":
                depth += 1
            elif ch in ")]>":
                depth -= 1
            elif ch == "{" and depth == 0:
                cut = min(cut, k)
                break
            elif ch == ":" and depth == 0:
                cut = k
                break
        return text[:cut]

    def _binary(self, short, a: Sym):
        """The element function of a binary op on tensors of `a`'s type, integer results wrapped to the type's width.
        INTEGER tensors are carried as integral floats and their semantics (truncating division, remainders, wrap-around, shifts)
        are spelled with float divisions that must be EXACT: relaxed arithmetic (dsl.relaxed_arithmetic: a / b as a * (1 / b),
        49 * (1 / 49) = 0.9999999999999999) is suspended inside them."""
        f = self._binary_any(short, a)
        if not (a.is_int() or a.is_bool()):
            return f

        def exact(x, y):
            saved, _dsl._RELAXED[0] = _dsl._RELAXED[0], False
            try:
                return f(x, y)
            finally:
                _dsl._RELAXED[0] = saved
        return exact

    def _binary_any(self, short, a: Sym):
        if a.dtype == "ui64":
            return _u64_binary(short)
        f = self._binary_raw(short, a)
        if not a.is_int() or _bits(a.dtype) >= 64:
            return f
        dt, bits = a.dtype, _bits(a.dtype)
        uns = "ui" + str(bits)
        if short in ("add", "subtract"):
            return lambda x, y: _wrap(f(x, y), dt)
        if short == "multiply":
            if bits < 32:
                return lambda x, y: _wrap(f(x, y), dt)
            def mul(x, y):
                for p, q in ((x, y), (y, x)):
                    k = _try_const(p)
                    if k is not None and abs(k) < 1048576.0:              # a small constant factor: the product stays exact
                        return _wrap(f(x, y), dt)
                return _wrap(_mul_lo32(_wrap(x, uns), _wrap(y, uns)), dt)
            return mul
        if short == "shift_left":
            return lambda x, y: _where(_dsl._lift(y) >= float(bits), _dsl.const(0.0), _wrap(_shl32(_wrap(x, uns), _np.clip(y, 0.0, float(bits - 1)), bits), dt))
        if short == "shift_right_logical":
            return lambda x, y: _where(_dsl._lift(y) >= float(bits), _dsl.const(0.0), _wrap(_np.right_shift(_wrap(x, uns), _np.clip(y, 0.0, float(bits - 1))), dt))
        return f

    def _binary_raw(self, short, a: Sym):
        if short == "add":
            return lambda x, y: x + y
        if short == "subtract":
            return lambda x, y: x - y
        if short == "multiply":
            return (lambda x, y: _np.logical_and(x, y)) if a.is_bool() else (lambda x, y: x * y)
        if short == "divide":
            return _trunc_div if a.is_int() else (lambda x, y: x / y)
        if short == "maximum":      # max(-inf, x) = x: the init of jnp.max / argmax / a padded max reduce_window never reaches the kernel
            return (lambda x, y: _np.logical_or(x, y)) if a.is_bool() else (
                lambda x, y: y if _try_const(x) == float("-inf") else (x if _try_const(y) == float("-inf") else _np.maximum(x, y)))
        if short == "minimum":
            return (lambda x, y: _np.logical_and(x, y)) if a.is_bool() else (
                lambda x, y: y if _try_const(x) == float("inf") else (x if _try_const(y) == float("inf") else _np.minimum(x, y)))
        if short == "power":
            return _np.power
        if short == "atan2":
            return _np.arctan2
        if short == "remainder":
            return _rem
        if short in ("and", "or", "xor"):
            if a.is_bool():
                return {"and": _np.logical_and, "or": _np.logical_or,
                        "xor": lambda x, y: _np.logical_and(_np.logical_or(x, y), _np.logical_not(_np.logical_and(x, y)))}[short]
            return {"and": _np.bitwise_and, "or": _np.bitwise_or, "xor": _np.bitwise_xor}[short]
        bits = float(int(re.sub(r"\D", "", a.dtype) or 64))
        if short == "shift_left":        # a shift by the width or more gives 0 (StableHLO), and the result wraps to the width
            return lambda x, y: _np.where(y >= bits, 0.0, _np.remainder(_np.left_shift(x, _np.minimum(y, bits - 1.0)), 2.0 ** bits)
                                          if a.dtype[0] == "u" else _np.left_shift(x, _np.minimum(y, bits - 1.0)))
        if short == "shift_right_logical":
            return lambda x, y: _np.where(y >= bits, 0.0, _np.right_shift(x, _np.minimum(y, bits - 1.0)))
        if short == "shift_right_arithmetic":
            return lambda x, y: _np.floor(x / _np.power(2.0, _np.minimum(y, bits - 1.0)))
        raise NotImplementedError(short)

    _NP_OF = {"f64": np.float64, "f32": np.float32, "f16": np.float16, "i64": np.int64, "i32": np.int32, "i16": np.int16, "i8": np.int8,
              "ui64": np.uint64, "ui32": np.uint32, "ui16": np.uint16, "ui8": np.uint8, "i1": np.uint8}

    @staticmethod
    def _hex_float(word: str, dtype: str) -> float:
        """A bare hex literal of a FLOAT type is the value's bit pattern (MLIR prints inf, nan and some finite values that way; the
        reference decodes it with f64::from_bits, libs/cranelift-mlir/src/parser.rs:740-744)."""
        import struct
        bits = int(word, 16)
        if dtype == "f64":
            return struct.unpack("<d", struct.pack("<Q", bits & 0xFFFFFFFFFFFFFFFF))[0]
        if dtype == "f32":
            return float(struct.unpack("<f", struct.pack("<I", bits & 0xFFFFFFFF))[0])
        if dtype == "f16":
            return float(np.array([bits & 0xFFFF], dtype=np.uint16).view(np.float16)[0])
        if dtype == "bf16":
            return float(struct.unpack("<f", struct.pack("<I", (bits & 0xFFFF) << 16))[0])
        raise NotImplementedError(f"hex literal of element type {dtype}")

    @classmethod
    def _constant(cls, text: str, rt: TensorType) -> Sym:
        m = re.search(r"dense<(.*)>\s*:", text, re.S)
        body = m.group(1).strip()
        is_float = rt.dtype[0] in "fb"
        if body.startswith('"0x'):                 # the raw little-endian bytes of the whole tensor (how large constants are printed)
            raw = bytes.fromhex(body.strip('"')[2:])
            if rt.dtype == "bf16":
                vals = [cls._hex_float(raw[k + 1:k + 2].hex() + raw[k:k + 1].hex(), "bf16") for k in range(0, len(raw), 2)]
            else:
                vals = np.frombuffer(raw, dtype=np.dtype(cls._NP_OF[rt.dtype]).newbyteorder("<")).tolist()
            if len(vals) == 1 and rt.size > 1:
                vals = vals * rt.size
        else:
            body = body.replace("true", "1").replace("false", "0")
            words = re.findall(r"-?(?:0x[0-9a-fA-F]+|[\d.]+(?:[eE][-+]?\d+)?|inf|nan)", body)
            vals = []
            for w in words:
                if w.lower().startswith(("0x", "-0x")):
                    vals.append(cls._hex_float(w, rt.dtype) if is_float else int(w, 16))
                elif is_float or any(c in w for c in ".eEn"):
                    vals.append(float(w))
                else:
                    vals.append(int(w))                      # integer literals stay Python ints: ui64 keeps all 64 bits
        if rt.dtype[0] in "iu" and rt.dtype not in ("i1", "ui64"):
            bits = _bits(rt.dtype)
            fold = (lambda v: v % (1 << bits)) if rt.dtype[0] == "u" else (lambda v: (v + (1 << (bits - 1))) % (1 << bits) - (1 << (bits - 1)))
            vals = [fold(int(v)) for v in vals]              # `dense<-1> : tensor<ui32>` is 0xFFFFFFFF
        flat = np.empty(rt.size, dtype=object)
        flat[:] = (list(vals) * rt.size) if len(vals) == 1 else list(vals)
        arr = flat.reshape(rt.shape)
        if rt.dtype == "i1":
            out = np.empty(rt.shape, dtype=object)
            for idx in (np.ndindex(rt.shape) if rt.shape else [()]):
                out[idx] = _dsl.const(float(arr[idx])) > 0.5
            return Sym(out, "i1")
        return Sym(_obj(arr, rt.dtype), rt.dtype)

    @staticmethod
    def _convert(x: Sym, to: str) -> Sym:
        if x.is_bool():
            if to == "i1":
                return x
            if to == "ui64":
                return Sym(_emap(lambda c: U64(0.0, _np.where(c, 1.0, 0.0)), x.a), to)
            return Sym(_emap(lambda c: _np.where(c, 1.0, 0.0), x.a), to)
        if x.dtype == "ui64":
            if to == "ui64":
                return x
            if to == "i1":
                return Sym(_emap(lambda v: _np.logical_not(_np.logical_and(_np.equal(v.hi, 0.0), _np.equal(v.lo, 0.0))), x.a), "i1")
            if to[0] in "iu" and _bits(to) <= 32:            # truncation keeps the low word
                return Sym(_emap(lambda v: _wrap(v.lo, to), x.a), to)
            return Sym(_emap(lambda v: v.to_float(), x.a), to)        # i64 / floats: hi * 2^32 + lo (rounded beyond 2^53)
        if to == "i1":
            return Sym(_emap(lambda v: _np.logical_not(_np.equal(v, 0.0)), x.a), "i1")
        if to == "ui64":
            src = x.a if x.is_int() else _emap(_np.trunc, x.a)
            return Sym(_emap(U64.from_float, src), to)
        if to[0] in "iu":
            src = x.a if x.is_int() else _emap(_np.trunc, x.a)
            return Sym(_emap(lambda v: _wrap(v, to), src), to)
        return Sym(x.a, to)

    @staticmethod
    def _bitcast(x: Sym, rt: TensorType) -> Sym:
        """stablehlo.bitcast_convert between equal-width types (how jax.random turns mantissa bits into a float)."""
        frm, to = x.dtype, rt.dtype
        if frm == to:
            return x
        if frm == "ui64" and to == "f64":
            return Sym(_emap(lambda v: Expr("bits2f", (v.hi, v.lo)), x.a), to)
        if frm == "f64" and to == "ui64":
            return Sym(_emap(lambda v: U64(Expr("fbits", (_dsl._lift(v),), 1), Expr("fbits", (_dsl._lift(v),), 0)), x.a), to)
        if frm in ("ui32", "i32") and to == "f32":
            return Sym(_emap(lambda v: Expr("bits2f32", (_wrap(v, "ui32"),)), x.a), to)
        if frm == "f32" and to in ("ui32", "i32"):
            return Sym(_emap(lambda v: _wrap(Expr("f32bits", (_dsl._lift(v),)), to), x.a), to)
        if frm[0] in "iu" and to[0] in "iu" and _bits(frm) == _bits(to) and _bits(to) <= 32:      # a reinterpretation of sign
            return Sym(_emap(lambda v: _wrap(v, to), x.a), to)
        raise NotImplementedError(f"stablehlo.bitcast_convert {frm} -> {to}")

    def _dot_general(self, a: Sym, b: Sym, text: str, rt: TensorType) -> Sym:
        def pair(key):
            m = re.search(key + r"\s*=\s*\[([^\]]*)\]\s*x\s*\[([^\]]*)\]", text)
            if not m:
                return [], []
            f = lambda s: [int(v) for v in s.replace(" ", "").split(",") if v]
            return f(m.group(1)), f(m.group(2))
        ba, bb = pair("batching_dims")
        ca, cb = pair("contracting_dims")
        fa = [d for d in range(a.a.ndim) if d not in ba + ca]
        fb = [d for d in range(b.a.ndim) if d not in bb + cb]
        out = np.empty(rt.shape, dtype=object)
        for idx in (np.ndindex(rt.shape) if rt.shape else [()]):
            bi, ai, bj = idx[:len(ba)], idx[len(ba):len(ba) + len(fa)], idx[len(ba) + len(fa):]
            acc = None
            for k in itertools.product(*[range(a.shape[d]) for d in ca]):
                ia, ib = [0] * a.a.ndim, [0] * b.a.ndim
                for d, v in zip(ba, bi):
                    ia[d] = v
                for d, v in zip(bb, bi):
                    ib[d] = v
                for d, v in zip(fa, ai):
                    ia[d] = v
                for d, v in zip(fb, bj):
                    ib[d] = v
                for d, v in zip(ca, k):
                    ia[d] = v
                for d, v in zip(cb, k):
                    ib[d] = v
                term = a.a[tuple(ia)] * b.a[tuple(ib)]
                acc = term if acc is None else acc + term
            out[idx] = acc if acc is not None else _dsl.const(0.0)
        return Sym(out, a.dtype)

    def _region_fn(self, op: Op, k: int, env):
        ops, args = op.regions[k], op.region_args[k] if k < len(op.region_args) else []
        def f(*vals):
            e = dict(env)
            for (name, ty), v in zip(args, vals):
                e[name] = v if isinstance(v, Sym) else Sym(_obj(v), ty.dtype)
            return self.block(ops, e)
        return f

    def _reduce(self, op: Op, xs, text, rts, env) -> List[Sym]:
        dims = self._ints(text, "dimensions")
        n = len(xs) // 2
        operands, inits = xs[:n], xs[n:]
        m = re.search(r"applies\s+([\w.]+)", text)
        if m:
            short = m.group(1).split(".", 1)[1]
            fn = self._binary(short, operands[0])
            combine = lambda acc, vals: [fn(acc[0], vals[0])]
        else:
            region = self._region_fn(op, 0, env)
            def combine(acc, vals):
                outs = region(*[Sym(_obj_scalar(v), o.dtype) for v, o in zip(acc, operands)], *[Sym(_obj_scalar(v), o.dtype) for v, o in zip(vals, operands)])
                return [o.a[()] for o in outs]
        keep = [d for d in range(operands[0].a.ndim) if d not in dims]
        out_shape = tuple(operands[0].shape[d] for d in keep)
        outs = [np.empty(out_shape, dtype=object) for _ in operands]
        for idx in (np.ndindex(out_shape) if out_shape else [()]):
            acc = [i.a[()] for i in inits]
            for k in itertools.product(*[range(operands[0].shape[d]) for d in dims]):
                full = [0] * operands[0].a.ndim
                for d, v in zip(keep, idx):
                    full[d] = v
                for d, v in zip(dims, k):
                    full[d] = v
                acc = combine(acc, [o.a[tuple(full)] for o in operands])
            for o, v in zip(outs, acc):
                o[idx] = v
        return [Sym(o, x.dtype) for o, x in zip(outs, operands)]

    UNROLL_MAX_TRIPS = 64          # a while whose trip count is known while tracing is unrolled up to this many iterations ...
    UNROLL_MAX_NODES = 200_000     # ... as long as the unrolled body stays below this many new nodes
    # ... unless the evaluator can keep it a COUNTED loop at no cost (_LaneEval: an edge_fold's scan over the edge slot whose targets are
    # exchange reads — the slot's source table is indexed by the counter, op lane_read_dyn) and the unrolled form is large: a world of 20
    # or 35 bodies unrolls to 9,100 / 13,000 instructions per tick, more than the 64 KB instruction cache holds (58 / 40 us per tick,
    # fetch-bound); as four 19- / 34-trip loops it is ~700 instructions
    ROLL = False
    ROLL_MIN_TRIPS = 12
    ROLL_MIN_NODES = 300
    ROLL_UNROLL = 1

    @staticmethod
    def _flatten_elems(x: Sym) -> list:
        """The float nodes that carry a tensor across a loop boundary: i1 as 0 / 1, ui64 as its two words."""
        out = []
        for v in x.a.reshape(-1):
            if x.dtype == "i1":
                out.append(_np.where(v, 1.0, 0.0))
            elif isinstance(v, U64):
                out += [v.hi, v.lo]
            else:
                out.append(v)
        return out

    def _while(self, op: Op, text: str, env) -> List[Sym]:
        m = re.match(r"\s*\((.*?)\)\s*:", text, re.S)
        binds = [p.split("=") for p in _split_top(m.group(1))]
        names = [b[0].strip() for b in binds]
        inits = [env[b[1].strip()] for b in binds]

        # -- static trip count: the reference's edge_fold is `lax.scan` over a source's out-edges, i.e. a while whose counter
        #    starts at a constant and is compared with a constant (libs/cranelift-mlir/tests/test_while_dyn_slice.rs); unrolled,
        #    its dynamic_slice / dynamic_update_slice by the counter are plain static slices
        state, trips, start_nodes, static_trips = list(inits), 0, Expr._count[0], None
        saved = self._loop_counters()
        while trips <= self.UNROLL_MAX_TRIPS and Expr._count[0] - start_nodes <= self.UNROLL_MAX_NODES:
            e = dict(env)
            e.update(zip(names, state))
            c = _try_const(self.block(op.regions[0], e)[0].a[()])
            if c is None:
                break                      # data-dependent condition: a real loop (from the ORIGINAL state: nothing unrolled is kept)
            if not c:
                static_trips = trips
                break
            outs = self.block(op.regions[1], e)
            state = [self._relabel(o, s_) for o, s_ in zip(outs, state)]
            trips += 1
        if static_trips is not None:
            if not (self.ROLL and static_trips >= self.ROLL_MIN_TRIPS and self._unrolled_size(state, inits, self.ROLL_MIN_NODES) >= self.ROLL_MIN_NODES):
                return state
            unrolled = self._loop_counters()
            self._loop_counters(saved)     # the unrolled nodes are dropped: what they counted is not part of the program
            try:
                return self._while_loop(op, env, names, inits, static_trips, saved)
            except NotEntityParallel:      # a carried value the loop form cannot hold (its layout changes inside the body): unrolled it is
                self._loop_counters(unrolled)
                return state
        return self._while_loop(op, env, names, inits, None, saved)

    def _while_loop(self, op: Op, env, names, inits, static_trips, saved) -> List[Sym]:
        """The while as a loop of the generated kernel: data-dependent (static_trips None), or counted."""

        shapes, dtypes = [x.shape for x in inits], [x.dtype for x in inits]
        kinds = [(x.eaxis, x.uni, x.tshape) for x in inits]

        # carried values the body hands back untouched (a scan's per-source rows, the stacked target rows it slices by the counter) are
        # not loop state: inside the loop they are the outer nodes themselves, so a dynamic_slice of them sees what they are made of
        invariant = [False] * len(inits)
        try:
            probe_in = [[_dsl.leaf(f"__probe{k}_{j}") for j in range(len(self._flatten_elems(x)))] for k, x in enumerate(inits)]
            pe, pk = dict(env), 0
            for nm, x, leaves_ in zip(names, inits, probe_in):
                size = int(np.prod(x.shape)) if x.shape else 1
                arr = np.empty(size, dtype=object)
                step = 2 if x.dtype == "ui64" else 1
                for j in range(size):
                    arr[j] = U64(leaves_[2 * j], leaves_[2 * j + 1]) if step == 2 else ((leaves_[j] > 0.5) if x.dtype == "i1" else leaves_[j])
                pe[nm] = Sym(arr.reshape(x.shape), x.dtype, x.eaxis, x.uni, x.tshape)
            probe_out = self.block(op.regions[1], pe)
            for k, (o, x) in enumerate(zip(probe_out, inits)):
                if x.dtype == "i1" or o.shape != x.shape:
                    continue
                got = self._flatten_elems(o)
                invariant[k] = len(got) == len(probe_in[k]) and all(a is b for a, b in zip(got, probe_in[k]))
        except (NotEntityParallel, _KindsChanged, NotImplementedError, TypeError, ValueError, KeyError):
            invariant = [False] * len(inits)
        self._loop_counters(saved)
        variant = [k for k in range(len(inits)) if not invariant[k]]
        if not variant:
            return list(inits)
        flat = [v for k in variant for v in self._flatten_elems(inits[k])]

        def rebuild(vals):
            out, k = {}, 0
            for idx_, (nm, shp, dt, (ea, un, ts)) in enumerate(zip(names, shapes, dtypes, kinds)):
                if invariant[idx_]:
                    out[nm] = inits[idx_]
                    continue
                size = int(np.prod(shp)) if shp else 1
                arr = np.empty(size, dtype=object)
                for j in range(size):
                    if dt == "ui64":
                        arr[j] = U64(vals[k], vals[k + 1])
                        k += 2
                    else:
                        arr[j] = (vals[k] > 0.5) if dt == "i1" else vals[k]
                        k += 1
                out[nm] = Sym(arr.reshape(shp), dt, ea, un, ts)
            return out

        def cond(c):
            e = dict(env)
            e.update(rebuild(list(c.e) if isinstance(c, _dsl.Vec) else [c]))
            return self.block(op.regions[0], e)[0].a[()]

        def body(c):
            e = dict(env)
            e.update(rebuild(list(c.e) if isinstance(c, _dsl.Vec) else [c]))
            outs = self.block(op.regions[1], e)
            changed = False
            for k, (o, shp) in enumerate(zip(outs, shapes)):
                if invariant[k]:
                    continue
                if o.shape != shp:
                    raise NotEntityParallel(f"a value carried by stablehlo.while changes its stored shape {shp} -> {o.shape}")
                ea, un, ts = kinds[k]
                ea2 = ea if ea is not None else o.eaxis            # a broadcast init that the body makes per-entity
                un2 = frozenset(d for d in (un & o.uni) if d != ea2)
                if (ea2, un2) != (ea, un):
                    kinds[k], changed = (ea2, un2, ts), True
            if changed:
                raise _KindsChanged()
            return _dsl.Vec([v for k in variant for v in self._flatten_elems(outs[k])])
        # (, ROLL_UNROLL = 1): not unrolled again — measured on the 20-body world, unroll 1 / 2 / 4 / 8 = 30.9 / 32.1 / 29.7 / 32.1 us per
        # tick (one wave per SIMD: the tick is ~13,000 issued instructions either way), and 4 / 8 spill the 35-body world
        counted = (0, int(static_trips), int(self.ROLL_UNROLL)) if static_trips is not None else None
        for _ in range(4):
            try:
                res = _dsl.lax.while_loop(cond, body, _dsl.Vec(flat), **({"counted": counted, "max_iter": counted[1] + 1} if counted else {}))
                break
            except _KindsChanged:
                continue
        else:
            raise NotEntityParallel("the carried values of a stablehlo.while do not settle on an entity layout")
        final = rebuild(list(res.e))
        return [final[nm] for nm in names]

    def _unrolled_size(self, state: List[Sym], inits: List[Sym], enough: int) -> int:
        """Nodes the unrolled loop adds between its initial values and its results (counted by reachability, not by creation: nodes
        are hash-consed, a second loop over the same values creates none); stops counting at `enough`."""
        stop = {id(v) for x in inits for v in self._flatten_elems(x) if isinstance(v, Expr)}
        seen, todo = set(), [v for x in state for v in self._flatten_elems(x) if isinstance(v, Expr)]
        while todo and len(seen) < enough:
            x = todo.pop()
            if id(x) in seen or id(x) in stop or x.op in ("const", "leaf"):
                continue
            seen.add(id(x))
            todo.extend(a for a in x.args if isinstance(a, Expr))
        return len(seen)

    def _loop_counters(self, restore=None):
        """Bookkeeping a discarded trial evaluation of a loop body must not leave behind (the lane evaluator's exchange count)."""
        return None

    @staticmethod
    def _relabel(o: Sym, like: Sym) -> Sym:
        return o

    def _case(self, op: Op, index: Sym, env) -> List[Sym]:
        branches = [self.block(r, env) for r in op.regions]
        n = len(branches)
        idx = index.a[()]
        if isinstance(idx, U64):
            idx = idx.to_float()
        k = _try_const(idx)
        if k is not None:
            k = int(k)
            return branches[k if 0 <= k < n else n - 1]
        idx = _np.where(_np.logical_or(idx < 0.0, idx > float(n - 1)), float(n - 1), idx)      # out of range: the last branch
        outs = []
        for k in range(len(branches[0])):
            syms = [branches[j][k] for j in range(n)]
            syms = self._agree(syms, "stablehlo.case")
            pick = syms[n - 1].a
            for j in range(n - 2, -1, -1):
                pick = _emap(lambda cur, alt, j=j: _where(_np.equal(idx, float(j)), alt, cur), pick, syms[j].a)
            outs.append(Sym(pick, syms[0].dtype, syms[0].eaxis, syms[0].uni, syms[0].tshape))
        return outs

    def _agree(self, syms: List[Sym], what: str) -> List[Sym]:
        """Results of alternative regions brought to one entity layout."""
        return syms

    @staticmethod
    def _pick(cands: List[np.ndarray], idx) -> np.ndarray:
        """cands[idx] element-wise for a traced (already clamped) idx."""
        if len(cands) == 1:
            return cands[0]
        k = _try_const(idx)
        if k is not None:
            return cands[int(min(max(k, 0), len(cands) - 1))]
        out = cands[-1]
        for k in range(len(cands) - 2, -1, -1):
            out = _emap(lambda cur, alt, k=k: _where(idx < (k + 0.5), alt, cur), out, cands[k])
        return out

    @staticmethod
    def _index(s: Sym):
        """A rank-0 index operand as one float node."""
        v = s.a.reshape(-1)[0]
        return v.to_float() if isinstance(v, U64) else v

    def _dynamic_slice(self, x: Sym, starts: List[Sym], sizes: List[int]) -> Sym:
        cur = x.a
        for d, (s, size) in enumerate(zip(starts, sizes)):
            hi = cur.shape[d] - size
            idx = _fold(_np.clip(self._index(s), 0.0, float(hi)))        # StableHLO clamps the start so the slice fits
            cands = [np.take(cur, range(k, k + size), axis=d) for k in range(hi + 1)]
            cur = self._pick(cands, idx)
        return Sym(cur, x.dtype)

    def _dynamic_update_slice(self, x: Sym, upd: Sym, starts: List[Sym]) -> Sym:
        idxs = [_fold(_np.clip(self._index(s), 0.0, float(x.shape[d] - upd.shape[d]))) for d, s in enumerate(starts)]
        out = np.empty(x.shape, dtype=object)
        for pos in (np.ndindex(x.shape) if x.shape else [()]):
            val = x.a[pos]
            # the update element that lands on `pos`, if any: pos - start in [0, upd.shape)
            for off in (np.ndindex(upd.shape) if upd.shape else [()]):
                start = [p - o for p, o in zip(pos, off)]
                if any(s_ < 0 or s_ > x.shape[d] - upd.shape[d] for d, s_ in enumerate(start)):
                    continue
                hit = None
                for d, s_ in enumerate(start):
                    c = _np.equal(idxs[d], float(s_))
                    hit = c if hit is None else _np.logical_and(hit, c)
                val = upd.a[off] if hit is None else _where(hit, upd.a[off], val)
            out[pos] = val
        return Sym(out, x.dtype)

    def _gather(self, operand: Sym, indices: Sym, text: str, rt: TensorType) -> Sym:
        g = lambda key: self._ints(text, key)
        offset_dims, collapsed, start_map = g("offset_dims"), g("collapsed_slice_dims"), g("start_index_map")
        op_batch, idx_batch = g("operand_batching_dims"), g("start_indices_batching_dims")
        ivd = int(re.search(r"index_vector_dim\s*=\s*(\d+)", text).group(1))
        slice_sizes = g("slice_sizes")
        batch_dims = [d for d in range(len(rt.shape)) if d not in offset_dims]
        kept_operand_dims = [d for d in range(operand.a.ndim) if d not in collapsed and d not in op_batch]
        out = np.empty(rt.shape, dtype=object)
        for pos in (np.ndindex(rt.shape) if rt.shape else [()]):
            bidx = [pos[d] for d in batch_dims]
            # the start vector of this batch position
            sel = list(bidx)
            if ivd < indices.a.ndim:
                sel.insert(ivd, slice(None))
                vec = list(np.atleast_1d(indices.a[tuple(sel)]))
            else:
                vec = [indices.a[tuple(sel)]]
            full = [None] * operand.a.ndim                           # per operand dim: a static int or a traced index
            vec = [(v.to_float() if isinstance(v, U64) else v) for v in vec]
            for k, d in enumerate(start_map):
                full[d] = _fold(_np.clip(vec[k], 0.0, float(operand.shape[d] - slice_sizes[d])))
            for ob, ib in zip(op_batch, idx_batch):
                pos_in_idx = [d for d in range(indices.a.ndim) if d != ivd]
                full[ob] = bidx[pos_in_idx.index(ib)]
            for k, d in enumerate(kept_operand_dims):
                off = pos[offset_dims[k]]
                full[d] = off if full[d] is None else _fold(full[d] + float(off))
            for d in range(operand.a.ndim):
                if full[d] is None:
                    full[d] = 0
            cur = operand.a
            for d in range(operand.a.ndim):                           # peel one dimension at a time
                ix = full[d]
                if isinstance(ix, (int, np.integer)):
                    cur = cur[int(ix)]
                elif isinstance(ix, Expr) and ix.op == "const":
                    cur = cur[int(ix.value)]
                else:
                    cands = [np.asarray(cur[k], dtype=object) if cur.ndim > 1 else _scalar(cur[k]) for k in range(cur.shape[0])]
                    cur = self._pick(cands, ix)
            out[pos] = cur[()] if isinstance(cur, np.ndarray) else cur
        return Sym(out, operand.dtype)

    def _pad(self, x: Sym, value: Sym, text: str, rt: TensorType) -> Sym:
        """stablehlo.pad: edge padding low / high (negative = crop) and interior padding with one value (jnp.pad, shifted windows)."""
        g = lambda key: self._ints(text, key)
        low, high = g("low") or g("edge_padding_low"), g("high") or g("edge_padding_high")
        interior = g("interior") or g("interior_padding") or [0] * x.a.ndim
        out = np.empty(rt.shape, dtype=object)
        pv = value.a.reshape(-1)[0]
        for pos in (np.ndindex(rt.shape) if rt.shape else [()]):
            src, inside = [], True
            for d, p_ in enumerate(pos):
                q = p_ - low[d]
                step = interior[d] + 1
                if q < 0 or q % step or q // step >= x.shape[d]:
                    inside = False
                    break
                src.append(q // step)
            out[pos] = x.a[tuple(src)] if inside else pv
        return Sym(out, x.dtype)

    @staticmethod
    def _window_attr(text: str, key: str, rank: int, default: int) -> List[int]:
        """window_dimensions / window_strides / ... in either spelling: `array<i64: 2, 2>` or `dense<[2, 2]> : tensor<2xi64>` (a
        splat `dense<1>` repeats)."""
        m = re.search(re.escape(key) + r"\s*=\s*(?:array<i64(?::\s*([^>]*))?>|dense<\[?([^\]>]*)\]?>)", text)
        if not m:
            return [default] * rank
        body = m.group(1) if m.group(1) is not None else (m.group(2) or "")
        vals = [int(x) for x in body.replace(" ", "").split(",") if x]
        return vals * rank if len(vals) == 1 and rank > 1 else (vals or [default] * rank)

    @staticmethod
    def _padding_attr(text: str, rank: int) -> List[Tuple[int, int]]:
        m = re.search(r"padding\s*=\s*dense<([^>]*)>", text)
        if not m:
            return [(0, 0)] * rank
        vals = [int(x) for x in re.findall(r"-?\d+", m.group(1))]
        if len(vals) == 1:
            return [(vals[0], vals[0])] * rank
        return [(vals[2 * d], vals[2 * d + 1]) for d in range(rank)]

    def _scatter(self, op: Op, xs: List[Sym], text: str, env) -> List[Sym]:
        """stablehlo.scatter, one operand: every update element lands on operand[start + window offset] through the update
        region — with traced start indices as a select per (operand position, update element) that can meet; an update whose
        window leaves the operand is skipped (scatter does not clamp).  `x.at[i].set(v)` / `.add(v)` lower to this."""
        if len(xs) != 3:
            raise NotImplementedError("stablehlo.scatter with several operands")
        operand, indices, updates = xs
        g = lambda key: self._ints(text, key)
        uwd, iwd, sdod = g("update_window_dims"), g("inserted_window_dims"), g("scatter_dims_to_operand_dims")
        ob, ib = g("input_batching_dims"), g("scatter_indices_batching_dims")
        if ob or ib:
            raise NotImplementedError("stablehlo.scatter with batching dimensions")
        m = re.search(r"index_vector_dim\s*=\s*(\d+)", text)
        ivd = int(m.group(1)) if m else (indices.a.ndim - 1 if indices.a.ndim else 0)
        region = self._region_fn(op, 0, env)
        scatter_dims = [d for d in range(updates.a.ndim) if d not in uwd]           # update dims that walk the index batch
        window_operand_dims = [d for d in range(operand.a.ndim) if d not in iwd]    # operand dims the update window spans
        out = operand.a.copy()
        for upos in (np.ndindex(updates.shape) if updates.shape else [()]):
            batch = [upos[d] for d in scatter_dims]
            sel = list(batch)
            if ivd < indices.a.ndim:
                sel.insert(ivd, slice(None))
                vec = list(np.atleast_1d(indices.a[tuple(sel)]))
            else:
                vec = [indices.a[tuple(sel)] if sel else indices.a[()]]
            start = [None] * operand.a.ndim
            for k, d in enumerate(sdod):
                start[d] = vec[k]
            offs = [0] * operand.a.ndim
            for k, d in enumerate(window_operand_dims):
                offs[d] = upos[uwd[k]]
            upd = updates.a[upos]
            for pos in (np.ndindex(operand.shape) if operand.shape else [()]):
                hit, possible = None, True
                for d in range(operand.a.ndim):
                    want = pos[d] - offs[d]                    # the start index that would put this update element on `pos`
                    if start[d] is None:
                        possible = possible and want == 0
                        continue
                    window = max((upos[uwd[k]] for k, dd in enumerate(window_operand_dims) if dd == d), default=0)
                    size = (updates.shape[uwd[window_operand_dims.index(d)]] if d in window_operand_dims else 1)
                    if want < 0 or want > operand.shape[d] - size:    # the whole window would not fit: such an update is skipped
                        possible = False
                        break
                    c = _np.equal(start[d], float(want))
                    hit = c if hit is None else _np.logical_and(hit, c)
                if not possible:
                    continue
                new = region(Sym(_obj_scalar(out[pos]), operand.dtype), Sym(_obj_scalar(upd), updates.dtype))[0].a[()]
                out[pos] = new if hit is None else _np.where(hit, new, out[pos])
        return [Sym(out, operand.dtype)]

    def _real_dynamic_slice(self, x: Sym, start: Sym, strides: Sym, rt: TensorType) -> Sym:
        """operand[start + i * stride] for the STATIC result shape (start / limit / strides are tensors; the limit only
        restates the shape)."""
        out = np.empty(rt.shape, dtype=object)
        for pos in (np.ndindex(rt.shape) if rt.shape else [()]):
            cur = x.a
            for d in range(x.a.ndim):
                ix = start.a.reshape(-1)[d] + float(pos[d]) * strides.a.reshape(-1)[d]
                if isinstance(ix, Expr) and ix.op == "const":
                    cur = cur[int(ix.value)]
                elif not isinstance(ix, Expr):
                    cur = cur[int(ix)]
                else:
                    cands = [np.asarray(cur[k], dtype=object) if cur.ndim > 1 else _scalar(cur[k]) for k in range(cur.shape[0])]
                    cur = self._pick(cands, _np.clip(ix, 0.0, float(len(cands) - 1)))
            out[pos] = cur[()] if isinstance(cur, np.ndarray) else cur
        return Sym(out, x.dtype)

    def _windows(self, shape, text: str):
        """(output shape, for each output position the list of (operand position or None for padding / dilation holes))."""
        rank = len(shape)
        wd = self._window_attr(text, "window_dimensions", rank, 1)
        ws = self._window_attr(text, "window_strides", rank, 1)
        bd = self._window_attr(text, "base_dilations", rank, 1)
        wdil = self._window_attr(text, "window_dilations", rank, 1)
        pad = self._padding_attr(text, rank)
        dilated = [(n - 1) * b + 1 if n > 0 else 0 for n, b in zip(shape, bd)]
        padded = [n + lo + hi for n, (lo, hi) in zip(dilated, pad)]
        span = [(w - 1) * dl + 1 for w, dl in zip(wd, wdil)]
        out_shape = tuple(max((p - sp) // st + 1, 0) if p >= sp else 0 for p, sp, st in zip(padded, span, ws))

        def elems(opos):
            res = []
            for w in itertools.product(*[range(k) for k in wd]):
                src = []
                for d in range(rank):
                    q = opos[d] * ws[d] + w[d] * wdil[d] - pad[d][0]          # position in the dilated operand
                    if q < 0 or q >= dilated[d] or q % bd[d] != 0:
                        src = None
                        break
                    src.append(q // bd[d])
                res.append(tuple(src) if src is not None else None)
            return res
        return out_shape, elems

    def _reduce_window(self, op: Op, xs: List[Sym], text: str, rts, env) -> List[Sym]:
        n = len(xs) // 2
        operands, inits = xs[:n], xs[n:]
        region = self._region_fn(op, 0, env)
        out_shape, elems = self._windows(operands[0].shape, text)
        outs = [np.empty(out_shape, dtype=object) for _ in operands]
        for opos in (np.ndindex(out_shape) if out_shape else [()]):
            acc = [i.a[()] for i in inits]
            for src in elems(opos):
                vals = [i.a[()] for i in inits] if src is None else [o.a[src] for o in operands]      # padding reads the init value
                res = region(*[Sym(_obj_scalar(v), o.dtype) for v, o in zip(acc, operands)], *[Sym(_obj_scalar(v), o.dtype) for v, o in zip(vals, operands)])
                acc = [r.a[()] for r in res]
            for o, v in zip(outs, acc):
                o[opos] = v
        return [Sym(o, x.dtype) for o, x in zip(outs, operands)]

    def _select_and_scatter(self, op: Op, xs: List[Sym], text: str, env) -> Sym:
        """Per window of `operand`: the element the select region keeps (the running choice survives while select(choice,
        candidate) holds) receives that window's `source` value through the scatter region; everything starts at `init`."""
        operand, source, init = xs
        select, scatter = self._region_fn(op, 0, env), self._region_fn(op, 1, env)
        out_shape, elems = self._windows(operand.shape, text)
        if tuple(out_shape) != tuple(source.shape):
            raise ValueError(f"select_and_scatter: {len(out_shape)}-d windows give {out_shape}, source is {source.shape}")
        out = np.empty(operand.shape, dtype=object)
        for pos in (np.ndindex(operand.shape) if operand.shape else [()]):
            out[pos] = init.a[()]
        for wpos in (np.ndindex(out_shape) if out_shape else [()]):
            cand = [e for e in elems(wpos) if e is not None]
            if not cand:
                continue
            cur = operand.a[cand[0]]
            chosen = [True]                                   # per candidate: is it the selected element (Python bool or traced)
            for e in cand[1:]:
                keep = select(Sym(_obj_scalar(cur), operand.dtype), Sym(_obj_scalar(operand.a[e]), operand.dtype))[0].a[()]
                chosen = [(_np.logical_and(c, keep) if c is not True else keep) for c in chosen] + [_np.logical_not(keep)]
                cur = _np.where(keep, cur, operand.a[e])
            for e, c in zip(cand, chosen):
                new = scatter(Sym(_obj_scalar(out[e]), operand.dtype), Sym(_obj_scalar(source.a[wpos]), source.dtype))[0].a[()]
                out[e] = new if c is True else _np.where(c, new, out[e])
        return Sym(out, operand.dtype)

    @staticmethod
    def _getrf(a: np.ndarray):
        """dgetrf on an object matrix: (LU in LAPACK's layout, ipiv 1-based, info) — per column the FIRST entry of largest
        magnitude on or below the diagonal is swapped up (one swap, as LAPACK leaves the rows), multipliers stored below the
        diagonal; pivot choice and swap are selects.  info = 1-based index of the first exactly-zero pivot, else 0."""
        n = a.shape[0]
        u = [[a[i, j] for j in range(n)] for i in range(n)]
        ipiv, info = [], _dsl.const(0.0)
        for k in range(n):
            p, best = _dsl.const(float(k)), _np.abs(u[k][k])
            for i in range(k + 1, n):
                better = _np.abs(u[i][k]) > best
                p, best = _np.where(better, float(i), p), _np.where(better, _np.abs(u[i][k]), best)
            ipiv.append(p + 1.0)
            old_k = list(u[k])
            for j in range(n):                                     # row k <- row p
                v = old_k[j]
                for i in range(k + 1, n):
                    v = _np.where(_np.equal(p, float(i)), u[i][j], v)
                u[k][j] = v
            for i in range(k + 1, n):                              # row p <- old row k
                hit = _np.equal(p, float(i))
                for j in range(n):
                    u[i][j] = _np.where(hit, old_k[j], u[i][j])
            info = _np.where(_np.logical_and(_np.equal(info, 0.0), _np.equal(u[k][k], 0.0)), float(k + 1), info)
            for i in range(k + 1, n):
                f = u[i][k] / u[k][k]
                u[i][k] = f
                for j in range(k + 1, n):
                    u[i][j] = u[i][j] - f * u[k][j]
        return np.array(u, dtype=object), np.array(ipiv, dtype=object), _scalar(info)

    @staticmethod
    def _geqrf(a: np.ndarray):
        """dgeqrf (unblocked dgeqr2 + dlarfg): R on and above the diagonal, the Householder vectors (v[0] = 1 implied) below it,
        tau per reflector — H = I - tau v v^T, beta = -sign(alpha) |x|, tau = (beta - alpha) / beta, v = x / (alpha - beta); a
        column that is already zero below the diagonal gets tau = 0."""
        m, n = a.shape
        r = [[a[i, j] for j in range(n)] for i in range(m)]
        taus = []
        for k in range(min(m, n)):
            alpha = r[k][k]
            xnorm2 = None
            for i in range(k + 1, m):
                xnorm2 = r[i][k] * r[i][k] if xnorm2 is None else xnorm2 + r[i][k] * r[i][k]
            if xnorm2 is None:                                   # last row: H = I
                taus.append(_dsl.const(0.0))
                continue
            trivial = _np.equal(xnorm2, 0.0)
            beta = -_np.where(alpha >= 0.0, 1.0, -1.0) * _np.sqrt(alpha * alpha + xnorm2)
            safe_beta = _np.where(trivial, 1.0, beta)
            tau = _np.where(trivial, 0.0, (safe_beta - alpha) / safe_beta)
            scale = 1.0 / _np.where(trivial, 1.0, alpha - safe_beta)
            v = [_dsl.const(1.0)] + [r[i][k] * scale for i in range(k + 1, m)]
            r[k][k] = _np.where(trivial, alpha, beta)
            for i in range(k + 1, m):
                r[i][k] = _np.where(trivial, r[i][k], v[i - k])
            for j in range(k + 1, n):                            # apply H to the trailing columns: c -= tau v (v . c)
                dot = r[k][j]
                for i in range(k + 1, m):
                    dot = dot + v[i - k] * r[i][j]
                for i in range(k, m):
                    r[i][j] = r[i][j] - tau * v[i - k] * dot
            taus.append(tau)
        return np.array(r, dtype=object), np.array(taus, dtype=object)

    @staticmethod
    def _orgqr(a: np.ndarray, tau: np.ndarray) -> np.ndarray:
        """dorgqr: the first n columns of Q = H_1 H_2 ... H_k from dgeqrf's reflectors (applied to the identity, last first)."""
        m, n = a.shape
        k = tau.shape[0]
        q = [[_dsl.const(1.0 if i == j else 0.0) for j in range(n)] for i in range(m)]
        for kk in range(k - 1, -1, -1):
            v = [_dsl.const(1.0)] + [a[i, kk] for i in range(kk + 1, m)]
            for j in range(n):
                dot = None
                for i in range(kk, m):
                    t = v[i - kk] * q[i][j]
                    dot = t if dot is None else dot + t
                for i in range(kk, m):
                    q[i][j] = q[i][j] - tau[kk] * v[i - kk] * dot
        return np.array(q, dtype=object)

    def _custom_call(self, text: str, xs: List[Sym], rts) -> List[Sym]:
        """The LAPACK FFI calls jax.numpy.linalg lowers to on CPU, for the factorisations elodin_amd.dsl_mat unrolls: dpotrf
        (Cholesky; the other triangle is zero) and dtrsm (triangular solve, left side)."""
        from . import dsl_mat
        target = re.search(r"@([\w.]+)", text).group(1)
        attr = lambda key, default=None: (lambda m: int(m.group(1)) if m else default)(re.search(key + r"\s*=\s*(\d+)\s*:\s*ui8", text))
        if target == "lapack_dpotrf_ffi" and xs[0].a.ndim == 2:
            lower = attr("uplo", 76) == 76
            n = xs[0].shape[0]
            Lm = dsl_mat.cholesky(dsl_mat.Mat([_dsl.Vec(list(r)) for r in xs[0].a]), lower=lower)
            out = _obj(np.zeros((n, n)))                    # the reference's dpotrf hands back the factor alone: other triangle zero
            for i in range(n):                              # (libs/cranelift-mlir/tests/ops.rs test_lapack_cholesky_3x3)
                for j in range(n):
                    if (j <= i) if lower else (j >= i):
                        out[i, j] = Lm[i].e[j]
            return [Sym(out, xs[0].dtype), Sym(_obj(np.zeros(())), "i32")][:max(1, len(rts))]
        if target == "lapack_dtrsm_ffi" and xs[0].a.ndim == 2 and attr("side", 76) == 76:
            lower, unit, trans = attr("uplo", 76) == 76, attr("diag", 78) == 85, 0 if attr("trans_x", 78) == 78 else 1
            A, B = dsl_mat.Mat([_dsl.Vec(list(r)) for r in xs[0].a]), xs[1].a
            cols = [dsl_mat.solve_triangular(A, _dsl.Vec(list(B[:, j])), lower=lower, trans=trans, unit_diagonal=unit) for j in range(B.shape[1])]
            return [Sym(np.array([[cols[j].e[i] for j in range(B.shape[1])] for i in range(B.shape[0])], dtype=object), xs[1].dtype)]
        if target == "lapack_dgetrf_ffi" and xs[0].a.ndim == 2 and xs[0].shape[0] == xs[0].shape[1]:
            lu, ipiv, info = self._getrf(xs[0].a)
            return [Sym(lu, xs[0].dtype), Sym(ipiv, "i32"), Sym(info, "i32")][:max(1, len(rts))]
        if target == "lapack_dgesv_ffi" and xs[0].a.ndim == 2 and xs[1].a.ndim == 2:
            # the solution of A X = B by the pivoted elimination above + substitution (dsl_mat.solve: same pivots, same sums)
            X = dsl_mat.solve(dsl_mat.Mat([_dsl.Vec(list(r)) for r in xs[0].a]), dsl_mat.Mat([_dsl.Vec(list(r)) for r in xs[1].a]))
            _, _, info = self._getrf(xs[0].a)
            return [Sym(np.array([[e for e in r.e] for r in X], dtype=object), xs[1].dtype), Sym(info, "i32")][:max(1, len(rts))]
        if target == "lapack_dgesdd_ffi" and xs[0].a.ndim == 2:
            # (A as LAPACK leaves it: unspecified for jobz = 'A', handed back unchanged; s descending; U; V^T; info = 0) by one-sided
            # Jacobi (dsl_mat.svd).  Singular vectors are determined up to a sign per pair, so U / V^T may differ from LAPACK's by it.
            u, sv, vh = dsl_mat.svd(dsl_mat.Mat([_dsl.Vec(list(r)) for r in xs[0].a]))
            mat = lambda m: np.array([[e for e in r.e] for r in m], dtype=object)
            return [Sym(xs[0].a.copy(), xs[0].dtype), Sym(np.array(list(sv.e), dtype=object), xs[0].dtype), Sym(mat(u), xs[0].dtype),
                    Sym(mat(vh), xs[0].dtype), Sym(_obj(np.zeros(())), "i32")][:max(1, len(rts))]
        if target == "lapack_dgeqrf_ffi" and xs[0].a.ndim == 2:
            qr_, tau = self._geqrf(xs[0].a)
            return [Sym(qr_, xs[0].dtype), Sym(tau, xs[0].dtype), Sym(_obj(np.zeros(())), "i32")][:max(1, len(rts))]
        if target == "lapack_dorgqr_ffi" and xs[0].a.ndim == 2:
            return [Sym(self._orgqr(xs[0].a, xs[1].a), xs[0].dtype), Sym(_obj(np.zeros(())), "i32")][:max(1, len(rts))]
        if target == "lapack_dsyevd_ffi" and xs[0].a.ndim == 2:
            # (eigenvectors as columns, eigenvalues ascending, info = 0) of the symmetric matrix whose `uplo` triangle is given, by
            # cyclic Jacobi (dsl_mat.eigh); an eigenvector is determined up to its sign, so columns may differ from LAPACK's by it
            lower = attr("uplo", 76) == 76
            n = xs[0].shape[0]
            sym = [[xs[0].a[max(i, j), min(i, j)] if lower else xs[0].a[min(i, j), max(i, j)] for j in range(n)] for i in range(n)]
            w, v = dsl_mat.eigh(dsl_mat.Mat([_dsl.Vec(r) for r in sym]))
            return [Sym(np.array([[e for e in r.e] for r in v], dtype=object), xs[0].dtype), Sym(np.array(list(w.e), dtype=object), xs[0].dtype),
                    Sym(_obj(np.zeros(())), "i32")][:max(1, len(rts))]
        raise NotImplementedError(f"stablehlo.custom_call @{target} is not provided by elodin_amd.stablehlo")

    def _sort(self, op: Op, x: Sym, text: str) -> Sym:
        if x.a.ndim != 1:
            raise NotImplementedError("stablehlo.sort: one-dimensional operands only")
        body = " ".join(o.text for o in op.regions[0])
        descending = bool(re.search(r"\bGT\b|\bGE\b", body))
        v = _np.sort(_dsl.Vec(list(x.a)))
        elems = list(v.e)[::-1] if descending else list(v.e)
        return Sym(np.array(elems, dtype=object), x.dtype)


class _LaneEval(_Eval):
    """ENTITY-PARALLEL evaluation of a whole-world module: the world's `[N, w]` columns arrive with their entity axis marked
    (Sym.eaxis) and stored at size 1 — one lane's row — and every statement is carried out on that one row, which is exact as long
    as the statement treats the entity axis as a pure batch axis (what `jax.vmap` over a query produces:
    libs/nox-py/python/elodin/__init__.py:212-253).  Every rule below derives where the entity axis of a result lies from where it
    lies in the operands, and REFUSES (NotEntityParallel) a statement that would move data between entities: a constant-index
    gather over the entity axis (an edge_fold's targets), a slice / reduce / contraction / concatenation along it.  A world whose
    tick is refused can still run with one lane per WHOLE world (world_system(mode="world"))."""

    CAP = 1 << 16              # elements a lazily-broadcast (uniform) tensor may take when a statement needs it in full

    def __init__(self, funcs, n_entities: int, stride: Optional[int] = None):
        """stride: rows per world when the executor lays worlds out as `stride` consecutive rows (a power of two <= 64, >= N): lets a
        constant-index gather ALONG the entity axis (a join, an edge_fold's targets) become an exchange inside the wavefront —
        lane i of a world reads what lane j of the same world holds (dsl op `lane_read`, one ds_bpermute per 32-bit half).  None:
        such gathers are refused."""
        super().__init__(funcs)
        self.N = int(n_entities)
        self.S = int(stride) if stride else None
        self.exchanges = 0            # lane_read nodes made: the manifest says whether the row layout matters

    # -- the exchange inside a world --
    ROLL = True

    def _loop_counters(self, restore=None):
        if restore is not None:
            self.exchanges = restore
        return self.exchanges

    def _pick(self, cands: List[np.ndarray], idx) -> np.ndarray:
        """cands[idx] for a traced idx (dynamic_slice by a loop counter).  Element by element: candidates that are exchange reads of ONE
        value — a scan over the edge slot slicing the stacked target rows — become one read whose source table is indexed by `idx`
        (lane_read_dyn); anything else is the plain chain of selects."""
        if len(cands) == 1 or self.S is None or _try_const(idx) is not None:
            return _Eval._pick(cands, idx)
        out = np.empty(cands[0].shape, dtype=object)
        for pos in (np.ndindex(cands[0].shape) if cands[0].shape else [()]):
            col = [c[pos] for c in cands]
            m = self._pick_reads(col, idx)
            if m is None:
                m = col[-1]
                for k in range(len(col) - 2, -1, -1):
                    m = _where(idx < (k + 0.5), col[k], m)
            out[pos] = m
        return out

    def _pick_reads(self, col, idx):
        if all(c is col[0] for c in col):
            return col[0]
        if all(isinstance(c, U64) for c in col):
            hi, lo = self._pick_reads([c.hi for c in col], idx), self._pick_reads([c.lo for c in col], idx)
            return None if hi is None or lo is None else U64(hi, lo)
        if any(not isinstance(c, Expr) for c in col):
            return None
        k0 = _try_const(col[0])
        if k0 is not None and not isinstance(k0, bool) and all(_try_const(c) == k0 for c in col):
            return col[0]
        ident = tuple(range(self.S))
        base, tables = None, []
        for c in col:
            b, t = (c.args[0], tuple(c.value[1])) if c.op == "lane_read" else (c, ident)
            if base is None:
                base = b
            elif b is not base:
                return None
            tables.append(t)
        if all(t == tables[0] for t in tables):
            return col[0]
        self.exchanges += 1
        return Expr("lane_read_dyn", (base, _dsl._lift(idx)), (self.S, tuple(tables)))

    def _lane_read(self, v, table: Tuple[int, ...]):
        """What entity table[i] of this lane's world holds of `v`, seen from entity i."""
        if isinstance(v, U64):
            return U64(self._lane_read(v.hi, table), self._lane_read(v.lo, table))
        v = _dsl._lift(v)
        if _try_const(v) is not None or all(j == i for i, j in enumerate(table)):
            return v
        if v.op == "lane_read":                      # a read of a read: entity i <- t2[i] <- t1[t2[i]]
            _, t1 = v.value
            return self._lane_read(v.args[0], tuple(t1[j] for j in table))
        self.exchanges += 1
        return Expr("lane_read", (v,), (self.S, tuple(int(j) for j in table)))

    def _exchange_gather(self, operand: Sym, indices: Sym, oe: int, ivd: int, rt: TensorType) -> Sym:
        """operand[..., j_b, ...] for constant j_b along the entity axis: every lane gets entity j_b's slice of its own world."""
        if self.S is None:
            raise NotEntityParallel("stablehlo.gather reads other entities' rows of a per-entity tensor (a join or an edge_fold's targets): "
                                    "the tick exchanges data between entities")
        operand = self._mat(operand)
        base = np.squeeze(operand.a, axis=oe)                      # one lane's slice, the entity axis gone
        if operand.dtype == "i1":
            base = _emap(lambda c: _np.where(c, 1.0, 0.0), base)
        idx = self._mat(indices).a
        batch_shape = tuple(d for k, d in enumerate(idx.shape) if k != ivd) if ivd < idx.ndim else idx.shape
        flat = np.moveaxis(idx, ivd, -1).reshape(-1) if ivd < idx.ndim else idx.reshape(-1)
        rows = []
        for v in flat:
            j = _try_const(v.to_float() if isinstance(v, U64) else v)
            if j is None:
                raise NotEntityParallel("stablehlo.gather along the entity axis with a traced (per-tick) index")
            j = int(min(max(j, 0), self.N - 1))                    # gather clamps
            rows.append(_emap(lambda e, j=j: self._lane_read(e, (j,) * self.S), base))
        out = np.stack(rows, axis=0).reshape(batch_shape + base.shape) if rows else np.empty(rt.shape, dtype=object)
        if operand.dtype == "i1":
            out = _emap(lambda e: _dsl._lift(e) > 0.5, out)
        if tuple(out.shape) != tuple(rt.shape):
            raise NotEntityParallel(f"stablehlo.gather along the entity axis: result layout {out.shape} is not the module's {rt.shape}")
        return self._annot(Sym(out, operand.dtype), rt.shape, None, ())

    def _merge_rows(self, full: np.ndarray, dim: int):
        """An [..., N, ...] tensor whose row s along `dim` holds, element by element, `lane_read(v, entity j_s)` of ONE node v (or one
        constant): it is the per-entity tensor whose lane i reads entity j_i — the rows folded back onto the entity axis.  -> the
        stored [..., 1, ...] array, or None when some element is not of that form."""
        moved = np.moveaxis(full, dim, 0)
        out = np.empty((1,) + moved.shape[1:], dtype=object)
        pad = tuple(range(self.N, self.S))

        def merge(col):
            if all(isinstance(c, U64) for c in col):
                hi, lo = merge([c.hi for c in col]), merge([c.lo for c in col])
                return None if hi is None or lo is None else U64(hi, lo)
            if any(not isinstance(c, Expr) for c in col):
                return None
            k0 = _try_const(col[0])
            if k0 is not None and not isinstance(k0, bool) and all(_try_const(c) == k0 for c in col):
                return col[0]
            base, table = None, []
            for c in col:
                if c.op != "lane_read" or len(set(c.value[1])) != 1:
                    return None
                if base is None:
                    base = c.args[0]
                elif c.args[0] is not base:
                    return None
                table.append(c.value[1][0])
            return self._lane_read(base, tuple(table) + pad)
        for pos in (np.ndindex(moved.shape[1:]) if moved.ndim > 1 else [()]):
            m = merge([moved[(s_,) + pos] for s_ in range(self.N)])
            if m is None:
                return None
            out[(0,) + pos] = m
        return np.moveaxis(out, 0, dim)

    # -- helpers --
    @staticmethod
    def _kind(x: Sym, d: int) -> str:
        return "E" if x.eaxis == d else ("U" if d in x.uni else "F")

    def _mat(self, x: Sym, axes=None) -> Sym:
        """`x` with its lazily-broadcast axes (all of them, or those in `axes`) stored in full."""
        axes = set(x.uni) if axes is None else (set(x.uni) & set(axes))
        if not axes:
            return x
        shape = list(x.a.shape)
        for d in axes:
            shape[d] = x.tshape[d]
        if int(np.prod(shape)) > self.CAP:
            raise NotEntityParallel(f"a broadcast tensor {x.tshape} is consumed element by element along a broadcast axis")
        return Sym(np.broadcast_to(x.a, shape).copy(), x.dtype, x.eaxis, frozenset(x.uni) - axes, x.tshape)

    @staticmethod
    def _stored(tshape, eaxis, uni) -> Tuple[int, ...]:
        return tuple(1 if (d == eaxis or d in uni) else s for d, s in enumerate(tshape))

    def _annot(self, out: Sym, tshape, eaxis, uni) -> Sym:
        uni = frozenset(d for d in uni if tshape[d] > 1 and d != eaxis)
        want = self._stored(tshape, eaxis, uni)
        if tuple(out.a.shape) != want:
            raise NotEntityParallel(f"internal: stored shape {out.a.shape} of a {tuple(tshape)} tensor (entity axis {eaxis}, broadcast axes {sorted(uni)})")
        out.tshape, out.eaxis, out.uni = tuple(tshape), eaxis, uni
        return out

    def _join(self, xs: List[Sym], what: str, skip: Sequence[int] = ()):
        """Per axis of equally ranked operands: the common layout.  -> (operands, eaxis, uni)"""
        rank = len(xs[0].tshape)
        eaxis, uni, xs = None, set(), list(xs)
        for d in range(rank):
            if d in skip:
                continue
            kinds = [self._kind(x, d) for x in xs]
            if "E" in kinds:
                for x, k in zip(xs, kinds):
                    if k == "F" and x.tshape[d] > 1:
                        raise NotEntityParallel(f"{what}: a per-entity tensor meets one that differs along the entity axis without being a column of the world")
                eaxis = d
            elif "U" in kinds and "F" in kinds:
                if all(x.tshape[d] == 1 for x, k in zip(xs, kinds) if k == "F"):
                    uni.add(d)                         # size-1 operands broadcast like anywhere else
            elif "U" in kinds:
                uni.add(d)
        return xs, eaxis, uni

    # -- the statement dispatcher --
    _EW = {"add", "subtract", "multiply", "divide", "maximum", "minimum", "power", "atan2", "remainder", "and", "or", "xor", "shift_left",
           "shift_right_logical", "shift_right_arithmetic", "compare", "select", "clamp", "convert", "bitcast_convert", "not", "is_finite"} | set(_UNARY)
    _BATCHED_LEADING = {"cholesky", "triangular_solve", "custom_call"}

    def op(self, op: Op, env) -> List[Sym]:
        name, text = op.name, op.text
        short = name.split(".", 1)[1] if "." in name else name
        if name in ("call", "func.call") or short == "while":
            return self._op(op, env)
        rts = self._result_types(text)
        if short == "constant":
            return [self._lane_constant(text, rts[0])]
        if short == "iota":
            if rts[0].size > self.CAP:
                raise NotEntityParallel(f"stablehlo.iota of {rts[0]}: an index over the entities")
            return self._op(op, env)
        xs = self._operands(self._operand_text(text), env)
        if not any(x.eaxis is not None or x.uni for x in xs):
            if rts and max(t.size for t in rts) > self.CAP and short not in ("broadcast_in_dim",):
                raise NotEntityParallel(f"{name}: a {rts[0]} tensor that is not a column of the world")
            if short != "broadcast_in_dim" and not (short == "concatenate" and self.S):      # (a stack of exchange reads may fold back onto the entity axis)
                return self._op(op, env)
        h = getattr(self, "_lane_" + short, None)
        if name.startswith("chlo.") or short in self._EW:
            h = self._lane_elementwise
        if short in self._BATCHED_LEADING:
            h = self._lane_batched_leading
        if h is None:
            raise NotEntityParallel(f"StableHLO op {name} on per-entity tensors is not provided by the entity-parallel front end")
        outs = h(op, xs, rts, env)
        return outs if isinstance(outs, list) else [outs]

    def _lane_constant(self, text: str, rt: TensorType) -> Sym:
        body = re.search(r"dense<(.*)>\s*:", text, re.S).group(1)
        splat = "[" not in body and not body.strip().startswith('"0x')
        lazy = {d for d, s_ in enumerate(rt.shape) if s_ == self.N and s_ > 1} if splat else set()
        if not lazy:
            if rt.size > self.CAP:
                raise NotEntityParallel(f"a constant {rt}: per-entity data baked into the module")
            return self._constant(text, rt)
        out = self._constant(text, TensorType(self._stored(rt.shape, None, lazy), rt.dtype))
        return self._annot(out, rt.shape, None, lazy)

    def _lane_elementwise(self, op, xs, rts, env):
        ranked = [x for x in xs if len(x.tshape) == len(rts[0].shape)] if rts[0].shape else []
        eaxis, uni = None, set()
        if ranked:
            _, eaxis, uni = self._join(ranked, op.name)
        outs = self._op(op, env)
        return [self._annot(o, rt.shape, eaxis, uni) for o, rt in zip(outs, rts)]

    def _lane_broadcast_in_dim(self, op, xs, rts, env):
        x, rt = xs[0], rts[0]
        dims = self._ints(op.text, "dims") or self._ints(op.text, "broadcast_dimensions")
        src_of = {dst: src for src, dst in enumerate(dims)}
        stored, eaxis, uni = [], None, set()
        for r, size in enumerate(rt.shape):
            k = src_of.get(r)
            kd = self._kind(x, k) if k is not None else None
            if kd == "E":
                eaxis = r
                stored.append(1)
            elif kd == "U" or ((k is None or x.tshape[k] != size) and size == self.N and size > 1):
                uni.add(r)                                  # a new (or stretched) axis as long as the entity axis stays a broadcast
                stored.append(1)
            else:
                stored.append(size)
        shape_in = [1] * len(rt.shape)
        for src, dst in enumerate(dims):
            shape_in[dst] = x.a.shape[src]
        arr = np.broadcast_to(x.a.reshape(shape_in), stored).copy()
        return self._annot(Sym(arr, x.dtype), rt.shape, eaxis, uni)

    @staticmethod
    def _reshape_groups(a: Sequence[int], b: Sequence[int]):
        """Axis groups of a reshape a -> b: [(axes of a, axes of b)] with equal products (size-1 axes ride along)."""
        groups, i, j = [], 0, 0
        while i < len(a) or j < len(b):
            gi, gj, pa, pb = [], [], 1, 1
            if i < len(a):
                gi.append(i); pa *= a[i]; i += 1
            if j < len(b):
                gj.append(j); pb *= b[j]; j += 1
            while pa != pb:
                if pa < pb and i < len(a):
                    gi.append(i); pa *= a[i]; i += 1
                elif j < len(b):
                    gj.append(j); pb *= b[j]; j += 1
                else:
                    raise ValueError("reshape sizes do not match")
            groups.append((gi, gj))
        return groups

    def _lane_reshape(self, op, xs, rts, env):
        x, rt = xs[0], rts[0]
        special = ({x.eaxis} if x.eaxis is not None else set()) | set(x.uni)
        mapping = {}
        for gi, gj in self._reshape_groups(x.tshape, rt.shape):
            mine = [d for d in gi if d in special]
            if not mine:
                continue
            big_i = [d for d in gi if x.tshape[d] > 1]
            big_j = [d for d in gj if rt.shape[d] > 1]
            if len(big_i) == 1 and len(big_j) == 1:
                mapping[big_i[0]] = big_j[0]
            else:
                bad = [d for d in mine if d == x.eaxis]
                if bad:
                    raise NotEntityParallel("stablehlo.reshape merges the entity axis with another axis")
                x = self._mat(x, mine)
        eaxis = mapping.get(x.eaxis) if x.eaxis is not None else None
        uni = {mapping[d] for d in x.uni if d in mapping}
        arr = x.a.reshape(self._stored(rt.shape, eaxis, uni))
        return self._annot(Sym(arr, x.dtype), rt.shape, eaxis, uni)

    def _lane_transpose(self, op, xs, rts, env):
        x = xs[0]
        perm = self._ints(op.text, "dims") or self._ints(op.text, "permutation")
        inv = {src: dst for dst, src in enumerate(perm)}
        return self._annot(Sym(np.transpose(x.a, perm).copy(), x.dtype), rts[0].shape,
                           inv.get(x.eaxis) if x.eaxis is not None else None, {inv[d] for d in x.uni})

    def _lane_reverse(self, op, xs, rts, env):
        x = xs[0]
        dims = self._ints(op.text, "dims") or self._ints(op.text, "dimensions")
        if x.eaxis in dims:
            raise NotEntityParallel("stablehlo.reverse along the entity axis")
        x = self._mat(x, dims)
        return self._annot(Sym(np.flip(x.a, axis=tuple(dims)).copy(), x.dtype), rts[0].shape, x.eaxis, x.uni)

    def _lane_slice(self, op, xs, rts, env):
        x = xs[0]
        ranges, sl, uni = self._slice_ranges(op.text), [], set()
        for d, (lo, hi, st) in enumerate(ranges):
            k = self._kind(x, d)
            if k == "E":
                if (lo, hi, st) != (0, x.tshape[d], 1):
                    raise NotEntityParallel("stablehlo.slice takes some entities' rows out of a column")
                sl.append(slice(0, 1))
            elif k == "U":
                sl.append(slice(0, 1))
                uni.add(d)
            else:
                sl.append(slice(lo, hi, st))
        return self._annot(Sym(x.a[tuple(sl)].copy(), x.dtype), rts[0].shape, x.eaxis, uni)

    def _lane_concatenate(self, op, xs, rts, env):
        dim = int(re.search(r"dim(?:ension)?\s*=\s*(\d+)", op.text).group(1))
        if any(x.eaxis == dim for x in xs):
            raise NotEntityParallel("stablehlo.concatenate along the entity axis (entity sets joined into one column)")
        if self.S and all(x.eaxis is None for x in xs) and rts[0].shape[dim] == self.N and self.N > 1 and xs[0].dtype != "i1":
            # per-source rows gathered from other entities, stacked back along the source axis (graph.rs:187-235): if every row is
            # an exchange read of one value, the stack IS a per-entity tensor again (lane i reads entity j_i)
            full = np.concatenate([self._mat(x).a for x in xs], axis=dim)
            merged = self._merge_rows(full, dim)
            if merged is not None:
                return self._annot(Sym(merged, xs[0].dtype), rts[0].shape, dim, ())
        xs = [self._mat(x, [dim]) for x in xs]
        xs, eaxis, uni = self._join(xs, "stablehlo.concatenate", skip=[dim])
        full = [d for d in range(len(rts[0].shape)) if d != dim and d != eaxis and d not in uni]
        xs = [self._mat(x, full) for x in xs]
        return self._annot(Sym(np.concatenate([x.a for x in xs], axis=dim), xs[0].dtype), rts[0].shape, eaxis, uni)

    def _lane_dot_general(self, op, xs, rts, env):
        a, b = xs
        def pair(key):
            m = re.search(key + r"\s*=\s*\[([^\]]*)\]\s*x\s*\[([^\]]*)\]", op.text)
            f = lambda s_: [int(v) for v in s_.replace(" ", "").split(",") if v]
            return (f(m.group(1)), f(m.group(2))) if m else ([], [])
        (ba, bb), (ca, cb) = pair("batching_dims"), pair("contracting_dims")
        if a.eaxis in ca or b.eaxis in cb:
            raise NotEntityParallel("stablehlo.dot_general contracts over the entity axis")
        a, b = self._mat(a, ca), self._mat(b, cb)
        kinds = []
        for da, db in zip(ba, bb):
            ka, kb = self._kind(a, da), self._kind(b, db)
            if "E" in (ka, kb):
                if "F" in (ka, kb) and max(a.tshape[da], b.tshape[db]) > 1:
                    raise NotEntityParallel("stablehlo.dot_general batches a per-entity tensor against per-entity data that is not a column")
                kinds.append("E")
            elif ka == kb == "U":
                kinds.append("U")
            else:
                a, b = self._mat(a, [da]), self._mat(b, [db])
                kinds.append("F")
        fa = [d for d in range(len(a.tshape)) if d not in ba + ca]
        fb = [d for d in range(len(b.tshape)) if d not in bb + cb]
        kinds += [self._kind(a, d) for d in fa] + [self._kind(b, d) for d in fb]
        es = [r for r, k in enumerate(kinds) if k == "E"]
        if len(es) > 1:
            raise NotEntityParallel("stablehlo.dot_general forms an entity-by-entity product")
        eaxis, uni = (es[0] if es else None), {r for r, k in enumerate(kinds) if k == "U"}
        rt = rts[0]
        out = self._dot_general(a, b, op.text, TensorType(self._stored(rt.shape, eaxis, uni), rt.dtype))
        return self._annot(out, rt.shape, eaxis, uni)

    def _lane_reduce(self, op, xs, rts, env):
        dims = self._ints(op.text, "dimensions")
        n = len(xs) // 2
        if any(x.eaxis in dims for x in xs[:n]):
            raise NotEntityParallel("stablehlo.reduce over the entity axis (a sum over the world)")
        for k in range(n):
            xs[k] = self._mat(xs[k], dims)
        ranked, eaxis, uni = self._join(xs[:n], "stablehlo.reduce", skip=dims)
        keep = [d for d in range(len(xs[0].tshape)) if d not in dims]
        full = [d for d in keep if d != eaxis and d not in uni]
        xs[:n] = [self._mat(x, full) for x in xs[:n]]
        outs = self._reduce(op, xs, op.text, rts, env)
        pos = {d: r for r, d in enumerate(keep)}
        return [self._annot(o, rt.shape, pos.get(eaxis) if eaxis is not None else None, {pos[d] for d in uni}) for o, rt in zip(outs, rts)]

    def _lane_dynamic_slice(self, op, xs, rts, env):
        x, starts = xs[0], xs[1:]
        sizes = self._ints(op.text, "sizes") or self._ints(op.text, "slice_sizes")
        sizes2, uni = list(sizes), set()
        for d in range(len(x.tshape)):
            k = self._kind(x, d)
            if k == "E":
                if sizes[d] != x.tshape[d]:
                    raise NotEntityParallel("stablehlo.dynamic_slice takes some entities' rows out of a column")
                sizes2[d] = 1
            elif k == "U":
                sizes2[d] = 1
                uni.add(d)
        if any(s_.eaxis is not None for s_ in starts):
            raise NotEntityParallel("stablehlo.dynamic_slice with a per-entity start index")
        out = self._dynamic_slice(x, starts, sizes2)
        return self._annot(out, rts[0].shape, x.eaxis, uni)

    def _lane_dynamic_update_slice(self, op, xs, rts, env):
        x, upd, starts = xs[0], xs[1], xs[2:]
        eaxis, uni = None, set()
        for d in range(len(x.tshape)):
            kx, ku = self._kind(x, d), self._kind(upd, d)
            if "E" in (kx, ku):
                if upd.tshape[d] != x.tshape[d] or ("F" in (kx, ku) and x.tshape[d] > 1):
                    raise NotEntityParallel("stablehlo.dynamic_update_slice writes some entities' rows of a column (a partial query's update_var)")
                eaxis = d
            elif kx == ku == "U" and upd.tshape[d] == x.tshape[d]:
                uni.add(d)
            else:
                x, upd = self._mat(x, [d]), self._mat(upd, [d])
        out = self._dynamic_update_slice(x, upd, starts)
        return self._annot(out, rts[0].shape, eaxis, uni)

    def _lane_gather(self, op, xs, rts, env):
        operand, indices = xs
        g = lambda key: _Eval._ints(self, op.text, key)
        offset_dims, collapsed, start_map = g("offset_dims"), g("collapsed_slice_dims"), g("start_index_map")
        op_batch, idx_batch = g("operand_batching_dims"), g("start_indices_batching_dims")
        ivd = int(re.search(r"index_vector_dim\s*=\s*(\d+)", op.text).group(1))
        slice_sizes = g("slice_sizes")
        rt = rts[0]
        batch_dims = [d for d in range(len(rt.shape)) if d not in offset_dims]
        idx_dims = [d for d in range(len(indices.tshape)) if d != ivd]
        kept = [d for d in range(len(operand.tshape)) if d not in collapsed and d not in op_batch]
        indices = self._mat(indices)
        operand = self._mat(operand)
        eaxis = None
        oe, ie = operand.eaxis, indices.eaxis
        if ie is not None and ie == ivd:
            raise NotEntityParallel("stablehlo.gather whose index vector runs over the entities")
        if oe is not None:
            if oe in op_batch:                                  # vmap of an indexed read: both sides batched over the entities
                if ie is None or idx_batch[op_batch.index(oe)] != ie:
                    raise NotEntityParallel("stablehlo.gather batches the entity axis of its operand against non-entity indices")
                eaxis = batch_dims[idx_dims.index(ie)]
            elif oe in kept and slice_sizes[oe] == operand.tshape[oe] and oe not in start_map and ie is None:
                eaxis = offset_dims[kept.index(oe)]             # whole columns picked by a shared index: the entity axis rides along
                slice_sizes = list(slice_sizes)
                slice_sizes[oe] = 1
            elif oe in collapsed and start_map == [oe] and ie is None and all(slice_sizes[d] == operand.tshape[d] for d in kept):
                if offset_dims != list(range(len(rt.shape) - len(offset_dims), len(rt.shape))):
                    raise NotEntityParallel("stablehlo.gather along the entity axis whose offset dimensions are not the trailing ones")
                return self._exchange_gather(operand, indices, oe, ivd, rt)
            else:
                raise NotEntityParallel("stablehlo.gather reads other entities' rows of a per-entity tensor (a join or an edge_fold's targets): "
                                        "the tick exchanges data between entities")
        elif ie is not None:                                     # a per-entity index into a table every entity shares
            eaxis = batch_dims[idx_dims.index(ie)]
        self._ints_override = {"slice_sizes": list(slice_sizes)}
        try:
            out = self._gather(operand, indices, op.text, TensorType(self._stored(rt.shape, eaxis, ()), rt.dtype))
        finally:
            self._ints_override = {}
        return self._annot(out, rt.shape, eaxis, ())

    def _lane_batched_leading(self, op, xs, rts, env):
        """cholesky / triangular_solve / the LAPACK calls on `[N, ...]` operands: the entity axis is their leading batch axis."""
        ranked = [x for x in xs if x.tshape]
        if any(x.eaxis not in (0, None) for x in ranked) or not all(x.eaxis == 0 or x.tshape[0] != self.N for x in ranked):
            raise NotEntityParallel(f"{op.name}: the entity axis is not the leading batch axis of every operand")
        names = re.findall(r"%[\w#.]+", self._operand_text(op.text))
        env2 = dict(env)
        for nm, x in zip(names, xs):
            x = self._mat(x)
            env2[nm] = Sym(x.a.reshape(x.a.shape[1:]), x.dtype) if x.eaxis == 0 else x
        self._rt_override = [TensorType(rt.shape[1:], rt.dtype) if (rt.shape and rt.shape[0] == self.N) else rt for rt in rts]
        outs = self._op(op, env2)
        res = []
        for o, rt in zip(outs, rts):
            if rt.shape and rt.shape[0] == self.N:
                res.append(self._annot(Sym(o.a.reshape((1,) + o.a.shape), o.dtype), rt.shape, 0, ()))
            else:
                res.append(o)
        return res

    def _agree(self, syms: List[Sym], what: str) -> List[Sym]:
        if not syms[0].tshape:
            return syms
        syms, eaxis, uni = self._join(syms, what)
        full = [d for d in range(len(syms[0].tshape)) if d != eaxis and d not in uni]
        syms = [self._mat(x, full) for x in syms]
        for x in syms:
            x.eaxis, x.uni = eaxis, frozenset(uni)
        return syms


def _scalar(v) -> np.ndarray:
    a = np.empty((), dtype=object)
    a[()] = v
    return a


def _obj_scalar(v) -> np.ndarray:
    return _scalar(v if isinstance(v, Expr) else _dsl.const(float(v)))


# ---- the front-end entry points ----------------------------------------------------------------------------------------------

def _column_values(o: Sym) -> list:
    """A result tensor as the float values a component column holds: i1 as 0 / 1, ui64 as hi * 2^32 + lo (exact to 2^53)."""
    return [(_np.where(e, 1.0, 0.0) if o.dtype == "i1" else (e.to_float() if isinstance(e, U64) else e)) for e in o.a.reshape(-1)]


def trace(text: str, inputs: Sequence) -> List[Sym]:
    """Evaluate @main on symbolic inputs (one per argument: a Sym, a dsl.Vec / Expr, or nested lists of those / numbers)."""
    funcs = parse_module(text)
    main = funcs["main"]
    if len(inputs) != len(main.args):
        raise ValueError(f"@main takes {len(main.args)} arguments, {len(inputs)} given")
    args = []
    for (name, ty), v in zip(main.args, inputs):
        if isinstance(v, Sym):
            args.append(v)
            continue
        elems = list(v.e) if isinstance(v, _dsl.Vec) else ([v] if isinstance(v, Expr) else list(np.asarray(v, dtype=object).reshape(-1)))
        if len(elems) != ty.size:
            raise ValueError(f"argument {name}: {ty} has {ty.size} elements, {len(elems)} given")
        arr = np.empty(ty.shape, dtype=object)
        if ty.shape == ():
            arr[()] = elems[0]
        else:
            arr.reshape(-1)[:] = elems
        if ty.dtype == "i1":
            arr = _emap(lambda e: e > 0.5, arr)
        elif ty.dtype == "ui64":
            arr = _emap(U64.of, arr)
        args.append(Sym(arr, ty.dtype))
    return _Eval(funcs).call(main, args)


def system(text: str, inputs: Sequence[str], outputs: Sequence[str], name: str = "stablehlo_main", every: int = 1) -> "_dsl.System":
    """@main as a per-entity system: argument k reads component inputs[k] (its tensor flattened row-major), result k is written
    to component outputs[k].  Booleans are written as 0 / 1."""
    funcs = parse_module(text)
    main = funcs["main"]
    if len(inputs) != len(main.args) or len(outputs) != len(main.result_types):
        raise ValueError("one component name per @main argument and per result")
    widths = {n: ty.size for n, (_, ty) in zip(inputs, main.args)}
    widths.update({n: ty.size for n, ty in zip(outputs, main.result_types)})
    params = list(dict.fromkeys(list(inputs) + [o for o in outputs]))

    def fn(**cols):
        syms = []
        for cname in inputs:
            v = cols[cname]
            syms.append(v if isinstance(v, _dsl.Vec) else _dsl.Vec([v]))
        outs = trace(text, syms)
        res = {}
        for cname, o in zip(outputs, outs):
            res[cname] = _dsl.Vec(_column_values(o))
        return res
    fn.__name__ = name
    import inspect
    fn.__signature__ = inspect.Signature([inspect.Parameter(p, inspect.Parameter.KEYWORD_ONLY) for p in params])
    out = _dsl.system(fn, every=every, **widths)
    out.float32_refused = float32_hazards(funcs)      # a float32 build of a program holding this system is refused (codegen._build)
    return out


# ---- whole-world ticks (what the reference hands a backend: libs/nox-py/src/cranelift_compile.rs:47-68) ----------------------------

class Slot:
    """One argument / result of a world tick's @main: ExecSlotMetadata (libs/nox-py/src/exec.rs:17-22) plus a readable name."""

    def __init__(self, component, shape: Sequence[int], entity_axis_elided: bool, component_id: Optional[int] = None):
        self.component, self.shape, self.elided = str(component), tuple(int(s) for s in shape), bool(entity_axis_elided)
        self.component_id = component_id

    @property
    def column(self) -> str:
        return "hlo_" + re.sub(r"\W", "_", self.component)

    @staticmethod
    def of(x) -> "Slot":
        if isinstance(x, Slot):
            return x
        if isinstance(x, dict):
            cid = x.get("component_id")
            return Slot(x.get("component", x.get("name", f"c{cid}")), x.get("shape", []), x.get("entity_axis_elided", False), cid)
        return Slot(*x)


# ---- worlds of MORE than one wavefront whose entities exchange data: the edge_fold becomes fold stages ---------------------------
# The reference's `edge_fold` (libs/nox-py/src/graph.rs:177-361, python twin elodin/__init__.py:454-557) reaches a backend as, per
# source, constant-index gathers of its targets' rows ([e, w] each), stacked to [N, e, w], transposed to [e, N, w] and consumed by a
# `while` over the edge slot that slices row `i` and calls the fold body (libs/cranelift-mlir/tests/test_gather_3body.rs,
# test_dynamic_ops_3body.rs, test_while_dyn_slice.rs).  Up to 64 entities the reads are lane exchanges inside the wavefront
# (_LaneEval with a stride).  Beyond that an entity's targets live in other wavefronts, so the exchange goes through memory and
# a launch boundary: the loop leaves the per-entity kernel and becomes a fold STAGE of the program (dsl.GraphFold — one lane
# per source folds its out-edges in slot order over CSR, the design of csrc/pair_kernel.hpp's edge kernel), with the per-entity
# statements in front of it and behind it as systems of their own (world_program).

class _Nbr(Sym):
    """Rows of ONE per-entity tensor `base` ([N, w]) taken at other entities, on their way into an edge_fold's scan — never
    materialised.  kind / true shape / table:  "rows" (k, w): rows `table` (k ints) | "rows3" (1, k, w) | "table" (N, e, w):
    table[s] = the e target rows of source s, slot order | "tableT" (e, N, w)."""

    def __init__(self, base: Sym, table, tshape, kind: str):
        self.a = np.empty((0,), dtype=object)
        self.dtype, self.eaxis, self.uni, self.tshape = base.dtype, None, frozenset(), tuple(tshape)
        self.base, self.table, self.kind = base, table, kind


class _FoldRequest:
    """One scan over the edge slot, lifted out of the tick: acc' = f(acc, own values, target values) over `table`."""

    def __init__(self, index, width, init, outputs, acc_leaves, own_exprs, own_leaves, nb_exprs, nb_leaves, table):
        self.index, self.width, self.init, self.outputs = index, width, init, outputs
        self.acc_leaves, self.own_exprs, self.own_leaves = acc_leaves, own_exprs, own_leaves
        self.nb_exprs, self.nb_leaves, self.table = nb_exprs, nb_leaves, table


def _substitute(exprs: Sequence[Expr], mapping: Dict[int, Expr]) -> List[Expr]:
    """The DAGs `exprs` with the nodes in `mapping` (by id) replaced; everything above them rebuilt (hash-consed)."""
    memo: Dict[int, Expr] = {}

    def go(x):
        if not isinstance(x, Expr):
            return x
        if id(x) in mapping:
            return mapping[id(x)]
        if id(x) in memo:
            return memo[id(x)]
        if x.op in ("while", "while_out", "lane_read", "lane_read_dyn", "wload"):
            raise NotEntityParallel(f"an edge_fold body holds a {x.op} node: only straight-line fold functions leave the tick as fold stages")
        args = tuple(go(a) for a in x.args)
        out = x if all(a is b for a, b in zip(args, x.args)) else Expr(x.op, args, x.value, x.name)
        memo[id(x)] = out
        return out
    return [go(e) for e in exprs]


class _FoldEval(_LaneEval):
    """_LaneEval for a world larger than a wavefront (no stride): constant-index gathers along the entity axis stay LAZY (_Nbr),
    and the `while` that consumes them is lifted out as a _FoldRequest; its results read the fold stage's output column."""

    def __init__(self, funcs, n_entities: int, fold_out_leaves):
        super().__init__(funcs, n_entities, None)
        self.fold_out_leaves = fold_out_leaves        # (fold index, width) -> the leaves of that fold's output column
        self.requests: List[_FoldRequest] = []
        self._nb_reads: Dict[Tuple[int, int], Expr] = {}      # (id(base Sym), feature) -> placeholder leaf, inside one fold body
        self._nb_base: Dict[int, Sym] = {}
        self._nb_table = None

    # -- lazy neighbour rows --
    def _exchange_gather(self, operand: Sym, indices: Sym, oe: int, ivd: int, rt: TensorType) -> Sym:
        operand = self._mat(operand)
        if oe != 0 or len(operand.tshape) != 2 or operand.dtype != "f64":
            raise NotEntityParallel("a constant-index gather along the entity axis of a tensor that is not an [N, w] f64 column value")
        idx = self._mat(indices).a
        flat = np.moveaxis(idx, ivd, -1).reshape(-1) if ivd < idx.ndim else idx.reshape(-1)
        rows = []
        for v in flat:
            j = _try_const(v.to_float() if isinstance(v, U64) else v)
            if j is None:
                raise NotEntityParallel("stablehlo.gather along the entity axis with a traced (per-tick) index")
            rows.append(int(min(max(j, 0), self.N - 1)))
        if tuple(rt.shape) != (len(rows), operand.tshape[1]):
            raise NotEntityParallel(f"stablehlo.gather along the entity axis: result {rt} is not [rows, w]")
        return _Nbr(operand, rows, rt.shape, "rows")

    def op(self, op: Op, env) -> List[Sym]:
        name = op.name
        short = name.split(".", 1)[1] if "." in name else name
        if name not in ("call", "func.call") and short not in ("while", "constant", "iota", "return"):
            xs = self._operands(self._operand_text(op.text), env)
            if any(isinstance(x, _Nbr) for x in xs):
                return [self._nbr_op(short, op, xs)]
        return super().op(op, env)

    def _nbr_op(self, short: str, op: Op, xs: List[Sym]) -> Sym:
        rt = self._result_types(op.text)[0]
        x = xs[0]
        w = x.base.tshape[1] if isinstance(x, _Nbr) else None
        if short == "reshape" and x.kind == "rows" and tuple(rt.shape) == (1,) + x.tshape:
            return _Nbr(x.base, x.table, rt.shape, "rows3")
        if short == "concatenate" and all(isinstance(y, _Nbr) and y.base is x.base and y.kind == x.kind for y in xs) and self._ints(op.text, "dim") in ([0], []) \
                and len(xs) == self.N:
            if x.kind == "rows3" and len({len(y.table) for y in xs}) == 1:
                return _Nbr(x.base, [list(y.table) for y in xs], rt.shape, "table")
            if x.kind == "rows" and all(len(y.table) == 1 for y in xs):
                if [y.table[0] for y in xs] == list(range(self.N)):
                    return x.base                    # every source's OWN row, stacked in source order: the column value itself
                raise NotEntityParallel("rows of a per-entity tensor permuted across the world (a join): not an edge_fold's source rows")
        if short == "transpose" and x.kind == "table" and self._ints(op.text, "dims") == [1, 0, 2]:
            return _Nbr(x.base, x.table, rt.shape, "tableT")
        if short == "dynamic_slice" and x.kind == "tableT":
            sizes = self._ints(op.text, "sizes") or self._ints(op.text, "slice_sizes")
            starts = xs[1:]
            if sizes == [1, self.N, w] and all(_try_const(self._index(s_)) == 0 for s_ in starts[1:]) and self._nb_table is not None:
                if self._nb_table[0] is None:
                    self._nb_table[0] = x.table
                elif self._nb_table[0] is not x.table and self._nb_table[0] != x.table:
                    raise NotEntityParallel("one scan slices neighbour rows of two different edge sets")
                self._nb_table[1].append(self._index(starts[0]))
                self._nb_base[id(x.base)] = x.base
                arr = np.empty((1, 1, w), dtype=object)
                t_ = list(self._nb_base).index(id(x.base))
                for j in range(w):
                    if (id(x.base), j) not in self._nb_reads:
                        self._nb_reads[(id(x.base), j)] = _dsl.leaf(f"__nb{len(self.requests)}_{t_}_{j}")
                    arr[0, 0, j] = self._nb_reads[(id(x.base), j)]
                return self._annot(Sym(arr, x.dtype), rt.shape, 1, ())
        raise NotEntityParallel(f"stablehlo.{short} on rows gathered from other entities: in a world of more than 64 entities only an edge_fold's "
                                "scan (stack, transpose, slice by the counter, fold body) may consume them")

    # -- the scan itself --
    def _while(self, op: Op, text: str, env) -> List[Sym]:
        m = re.match(r"\s*\((.*?)\)\s*:", text, re.S)
        binds = [p_.split("=") for p_ in _split_top(m.group(1))]
        names = [b[0].strip() for b in binds]
        inits = [env[b[1].strip()] for b in binds]
        if not any(isinstance(x, _Nbr) for x in inits):
            return super()._while(op, text, env)
        return self._fold_while(op, env, names, inits)

    def _placeholder(self, k: int, x: Sym) -> Sym:
        """A carried value as fresh leaves, per-entity along its N-sized axis when it has one."""
        if x.dtype in ("ui64", "i1"):
            raise NotEntityParallel(f"an edge_fold scan carries a {x.dtype} value")
        eaxis = x.eaxis if x.eaxis is not None else next((d for d, s_ in enumerate(x.tshape) if s_ == self.N and self.N > 1), None)
        stored = self._stored(x.tshape, eaxis, ())
        arr = np.empty(int(np.prod(stored)) if stored else 1, dtype=object)
        for j in range(arr.size):
            arr[j] = _dsl.leaf(f"__fc{len(self.requests)}_{k}_{j}")
        return Sym(arr.reshape(stored), x.dtype, eaxis, (), x.tshape)

    def _fold_while(self, op: Op, env, names, inits) -> List[Sym]:
        f_idx = len(self.requests)
        ph = [x if isinstance(x, _Nbr) else self._placeholder(k, x) for k, x in enumerate(inits)]

        def run_body(ph_):
            saved = (self._nb_reads, self._nb_base, self._nb_table)
            self._nb_reads, self._nb_base, self._nb_table = {}, {}, [None, []]
            try:
                e2 = dict(env)
                e2.update(zip(names, ph_))
                outs_ = self.block(op.regions[1], e2)
                cond_ = self.block(op.regions[0], e2)[0].a[()]
                return outs_, cond_, self._nb_reads, self._nb_base, self._nb_table
            finally:
                self._nb_reads, self._nb_base, self._nb_table = saved
        # carried values the body hands back untouched (the source's own rows) are not state: inside the body they are the outer nodes
        outs, _, _, _, _ = run_body(ph)
        for k, (x, p_, o) in enumerate(zip(inits, ph, outs)):
            if isinstance(x, _Nbr) or isinstance(o, _Nbr):
                continue
            got, was = self._flatten_elems(o), self._flatten_elems(p_)
            if len(got) == len(was) and all(a_ is b_ for a_, b_ in zip(got, was)) and not (not x.tshape and x.is_int()):
                ph[k] = x
        outs, cond, nb_reads, nb_base, (table, counters) = run_body(ph)
        if table is None:
            raise NotEntityParallel("a while carries gathered neighbour rows but never slices them by its counter")
        e_slots = len(table[0])
        # which carried value is the counter: the one the condition reads
        def leaves_of(x, acc):
            todo, seen = [x], set()
            while todo:
                y = todo.pop()
                if not isinstance(y, Expr) or id(y) in seen:
                    continue
                seen.add(id(y))
                if y.op == "leaf":
                    acc.add(y.name)
                todo.extend(y.args)
            return acc
        cond_leaves = leaves_of(cond if isinstance(cond, Expr) else _dsl._lift(cond), set())
        counter = [k for k, x in enumerate(ph) if not isinstance(x, _Nbr) and x.a.size == 1 and not x.tshape and x.a.reshape(-1)[0].name in cond_leaves]
        if len(counter) != 1 or not inits[counter[0]].is_int():
            raise NotEntityParallel("an edge_fold scan whose condition is not a test of one integer counter")
        c = counter[0]
        c_leaf = ph[c].a.reshape(-1)[0]
        if any(ix is not c_leaf for ix in counters):
            raise NotEntityParallel("an edge_fold scan slices its neighbour rows by something that is not its counter")
        # trip count: the counter starts at a constant, goes up by one, and the condition holds exactly while it is below the slot count
        c0 = _try_const(inits[c].a.reshape(-1)[0])
        def with_counter(region, value):
            e3 = dict(env)
            e3.update(zip(names, ph))
            e3[names[c]] = Sym(np.array(_dsl.const(float(value)), dtype=object).reshape(()), inits[c].dtype)
            saved_ = (self._nb_reads, self._nb_base, self._nb_table)
            self._nb_reads, self._nb_base, self._nb_table = {}, {}, [None, []]
            try:
                return self.block(region, e3)
            finally:
                self._nb_reads, self._nb_base, self._nb_table = saved_
        if c0 is None or _try_const(with_counter(op.regions[1], c0)[c].a.reshape(-1)[0]) != c0 + 1 \
                or _try_const(with_counter(op.regions[0], c0 + e_slots - 1)[0].a[()]) is not True \
                or _try_const(with_counter(op.regions[0], c0 + e_slots)[0].a[()]) is not False:
            raise NotEntityParallel(f"an edge_fold scan that does not run its counter from a constant over the {e_slots} edge slots")
        variant, acc_leaves, init_vals, out_exprs, layouts = [], [], [], [], {}
        for k, (x, p_, o) in enumerate(zip(inits, ph, outs)):
            if isinstance(x, _Nbr) or k == c:
                if isinstance(x, _Nbr) and o is not x:
                    raise NotEntityParallel("an edge_fold scan rewrites the neighbour rows it carries")
                continue
            if p_ is x:
                if o is not x and not (len(self._flatten_elems(o)) == len(self._flatten_elems(x)) and all(a is b for a, b in zip(self._flatten_elems(o), self._flatten_elems(x)))):
                    raise NotEntityParallel("internal: a carried value classified as untouched is rewritten by the scan")
                continue                                   # handed back untouched: not state
            o = self._mat(o)
            got, was = self._flatten_elems(o), self._flatten_elems(p_)
            if o.eaxis != p_.eaxis or tuple(o.a.shape) != tuple(p_.a.shape):
                raise NotEntityParallel(f"a value carried by an edge_fold scan changes its entity layout ({p_.tshape})")
            init = x.a
            if p_.eaxis is not None and init.ndim == len(p_.a.shape) and init.shape[p_.eaxis] != 1:      # stored in full along the entity axis
                first = np.take(init, [0], axis=p_.eaxis)
                if not all(a_ is b_ or _try_const(a_) == _try_const(b_) is not None
                           for a_, b_ in zip(np.broadcast_to(first, init.shape).reshape(-1), init.reshape(-1))):
                    raise NotEntityParallel("an edge_fold scan whose initial accumulator differs from entity to entity")
                init = first
            init = np.broadcast_to(init, p_.a.shape)
            vals = [_try_const(v) for v in init.reshape(-1)]
            if any(v is None or isinstance(v, bool) for v in vals):
                raise NotEntityParallel("an edge_fold scan whose initial accumulator is not a constant (the reference folds from el.Force() / a literal)")
            variant.append(k)
            layouts[k] = (p_.a.shape, p_.dtype, p_.eaxis, p_.tshape)
            acc_leaves += was
            init_vals += [float(v) for v in vals]
            out_exprs += [_dsl._lift(v) for v in got]
        if not variant:
            raise NotEntityParallel("an edge_fold scan without an accumulator")
        if any(c_leaf.name in leaves_of(e_, set()) for e_ in out_exprs):
            raise NotEntityParallel("an edge_fold body reads the edge slot's number")
        # what the body reads besides its accumulator and the targets' rows: values of the source's own lane, computed in front of the
        # loop — the maximal sub-DAGs that do not depend on a placeholder
        inner_names = {v.name for v in acc_leaves} | {v.name for v in nb_reads.values()}
        dep: Dict[int, bool] = {}
        def depends(x):
            if not isinstance(x, Expr):
                return False
            if id(x) in dep:
                return dep[id(x)]
            r = (x.op == "leaf" and x.name in inner_names) or any(depends(a) for a in x.args)
            dep[id(x)] = r
            return r
        own: List[Expr] = []
        seen = set()
        def frontier(x):
            if not isinstance(x, Expr) or id(x) in seen:
                return
            seen.add(id(x))
            if not depends(x):
                if x.op != "const":
                    own.append(x)
                return
            for a in x.args:
                frontier(a)
        for e_ in out_exprs:
            frontier(e_)
        own_leaves = [_dsl.leaf(f"__own{f_idx}_{j}") for j in range(len(own))]
        nb_keys = sorted(nb_reads, key=lambda kj: (list(nb_base).index(kj[0]), kj[1]))
        used_nb = [kj for kj in nb_keys if any(nb_reads[kj].name in leaves_of(e_, set()) for e_ in out_exprs)]
        nb_exprs = [_dsl._lift(nb_base[b].a.reshape(-1)[j]) for b, j in used_nb]
        req = _FoldRequest(f_idx, len(acc_leaves), init_vals, out_exprs, acc_leaves, own, own_leaves, nb_exprs, [nb_reads[kj] for kj in used_nb], table)
        self.requests.append(req)
        out_leaves = list(self.fold_out_leaves(f_idx, req.width))
        results, k0 = [], 0
        for k, x in enumerate(inits):
            if isinstance(x, _Nbr):
                results.append(x)
            elif k == c:
                results.append(Sym(np.array(_dsl.const(float(c0 + e_slots)), dtype=object).reshape(()), x.dtype))
            elif k in layouts:
                shp, dt, ea, ts = layouts[k]
                n_ = int(np.prod(shp)) if shp else 1
                arr = np.empty(n_, dtype=object)
                arr[:] = out_leaves[k0:k0 + n_]
                k0 += n_
                results.append(Sym(arr.reshape(shp), dt, ea, (), ts))
            else:
                results.append(x)
        return results


def world_program(text: str, slots: Sequence, out_slots: Optional[Sequence] = None, name: str = "world_tick", wave_folds: bool = True,
                  arith: str = "reference"):
    """A whole-world tick whose entities exchange data across MORE than a wavefront (an edge_fold over a world of more than 64
    entities) as a PROGRAM: per-entity systems with the tick's scans over the edge slot between them as fold stages.
    -> (dsl.Program, manifest, graph_edges)

    The program's columns are the world's slots (`hlo_<component>`, one row per entity; singleton slots replicated per row) plus,
    per scan k, three scratch columns the host provides zero-filled: `hlo_fold<k>_own` (what the fold body reads of the source's
    own lane), `hlo_fold<k>_nbr` (what it reads of a target) and `hlo_fold<k>_out` (the accumulator it leaves).  One tick =
    system 0 | fold 0 | system 1 | ... | fold K-1 | system K: system k recomputes from the world's columns and the earlier folds'
    outputs what fold k needs and stores it; the last one computes the tick's results.  `graph_edges` = {edge component of fold
    k: (source rows, target rows)} in slot order — rows of ONE world; hand them to the executor as entity ids of its rows
    (HipExec graph_edges=, with graph_replicas=(worlds, N) for a Monte-Carlo of such worlds).
    arith="relaxed": the systems AND the fold bodies are traced under dsl.relaxed_arithmetic (world_system's switch: finite values
    assumed, one reciprocal per denominator, a * b + c contracted) — inside 1e-9 of the reference, not its last bits."""
    if arith not in ("reference", "relaxed"):
        raise ValueError("arith must be 'reference' or 'relaxed'")
    funcs = parse_module(text)
    main = funcs["main"]
    ins = [Slot.of(x) for x in slots]
    outs = [Slot.of(x) for x in out_slots] if out_slots is not None else list(ins)
    if len(ins) != len(main.args) or len(outs) != len(main.result_types):
        raise ValueError(f"@main has {len(main.args)} arguments / {len(main.result_types)} results; {len(ins)} / {len(outs)} slots given")
    for k, (s_, ty) in enumerate(zip(outs, main.result_types)):
        if not s_.shape and ty.shape:
            outs[k] = Slot(s_.component, ty.shape, False, s_.component_id)
    counts = {s_.shape[0] for s_ in ins + outs if not s_.elided and s_.shape}
    if len(counts) != 1:
        raise NotEntityParallel("the batched slots do not share one entity count (components on different entity sets)")
    n_entities = counts.pop()
    width = lambda s_: int(np.prod(s_.shape[1:] if not s_.elided else s_.shape)) if (s_.shape[1:] if not s_.elided else s_.shape) else 1
    widths = {}
    for s_ in ins + outs:
        if widths.setdefault(s_.column, width(s_)) != width(s_):
            raise ValueError(f"component {s_.component} appears with two shapes")
    world_cols = list(dict.fromkeys([s_.column for s_ in ins] + [s_.column for s_ in outs]))

    def evaluate(cols, fold_out_leaves):
        """One entity-parallel evaluation of @main over the leaves `cols`: -> (evaluator with its fold requests, {result column: Vec})."""
        args = []
        for s_, (_, ty) in zip(ins, main.args):
            v = cols[s_.column]
            elems = list(v.e) if isinstance(v, _dsl.Vec) else [v]
            stored = ((1,) + tuple(ty.shape[1:])) if not s_.elided else tuple(ty.shape)
            arr = np.empty(len(elems), dtype=object)
            arr[:] = elems
            arr = arr.reshape(stored)
            if ty.dtype in ("i1", "ui64"):
                raise NotEntityParallel(f"slot {s_.component}: {ty.dtype} columns are not provided for worlds larger than a wavefront")
            args.append(Sym(arr, ty.dtype, None if s_.elided else 0, (), ty.shape))
        ev = _FoldEval(funcs, n_entities, fold_out_leaves)
        res = {}
        if arith == "relaxed":
            with _dsl.relaxed_arithmetic():
                results = ev.call(main, args)
        else:
            results = ev.call(main, args)
        for s_, o in zip(outs, results):
            if isinstance(o, _Nbr):
                raise NotEntityParallel(f"result {s_.component} is a stack of other entities' rows")
            o = ev._mat(o, [d for d in o.uni if d != 0] if not s_.elided else None)
            if s_.elided and o.eaxis is not None:
                raise NotEntityParallel(f"the singleton component {s_.component} would become per-entity")
            if not s_.elided and o.eaxis not in (0, None):
                raise NotEntityParallel(f"result {s_.component}: the entity axis is not its leading axis")
            vals = _column_values(o)
            src = cols.get(s_.column)
            src = (list(src.e) if isinstance(src, _dsl.Vec) else [src]) if src is not None else None
            if src is not None and len(src) == len(vals) and all(a is b for a, b in zip(src, vals)):
                continue
            res[s_.column] = _dsl.Vec(vals)
        return ev, res

    # ---- discovery: how many scans, how wide their accumulators, what they read (leaves of this pass never reach the program) ----
    probe_cols = {c: (lambda v: v if len(v) > 1 else v[0])(_dsl.Vec([_dsl.leaf(f"__w_{c}_{j}") for j in range(widths[c])])) for c in world_cols}
    probe, _ = evaluate(probe_cols, lambda k, w: [_dsl.leaf(f"__fo{k}_{j}") for j in range(w)])
    if not probe.requests:
        raise NotEntityParallel("this tick has no scan over gathered neighbour rows: it is entity-parallel (world_system) or not an edge_fold world")
    folds, graph_edges = [], {}
    for r in probe.requests:
        own_c, nbr_c, out_c = f"hlo_fold{r.index}_own", f"hlo_fold{r.index}_nbr", f"hlo_fold{r.index}_out"
        widths[own_c], widths[nbr_c], widths[out_c] = max(1, len(r.own_exprs)), max(1, len(r.nb_exprs)), r.width
        too_wide = [c for c in (own_c, nbr_c, out_c) if widths[c] > _dsl._MAT_MAX_ELEMS]
        if too_wide:
            raise NotImplementedError(f"fold {r.index}: scratch columns {too_wide} are wider than {_dsl._MAT_MAX_ELEMS} values")

        def make_fn(r=r):
            def fold_fn(acc, own, nbr):
                vec = lambda v: list(v.e) if isinstance(v, _dsl.Vec) else [v]
                mapping = {id(a): b for a, b in zip(r.acc_leaves, vec(acc))}
                mapping.update({id(a): b for a, b in zip(r.own_exprs, vec(own))})
                mapping.update({id(a): b for a, b in zip(r.nb_leaves, vec(nbr))})
                return _dsl.Vec(_substitute(r.outputs, mapping))
            fold_fn.__name__ = f"{name}_fold{r.index}"
            return fold_fn
        edge_c = f"hlo_fold{r.index}_edges"
        folds.append(_dsl.GraphFold(make_fn(), edge_c, (own_c,), (nbr_c,), out_c, list(r.init)))
        folds[-1].gather_batch = 4 if len(r.table[0]) >= 4 else 1      # a long scan is a chain of dependent gathers: fetch four targets per round trip (codegen._emit_fold_stage)
        # ... and a scan of a wavefront's worth of edges or more, when it is a plain sum, is folded by a whole WAVE per source (partial
        # sums per lane, a fixed shuffle tree: another association of the same sum, ~1e-16 x sqrt(degree)); wave_folds=False keeps
        # the one-lane sequential fold, bit for bit the reference's order
        folds[-1].wave_fold = bool(wave_folds) and len(r.table[0]) >= 64
        folds[-1].direct_out = True      # the accumulator column is none of the columns the scan reads: no scratch-then-commit launch
        if all(len(r.table[s_]) == n_entities - 1 and r.table[s_] == [t for t in range(n_entities) if t != s_] for s_ in range(n_entities)):
            # every source folds every other entity in ascending order (examples/n-body/sim.py:330-338): the complete graph — said,
            # not listed, so its n (n - 1) edges need not fit the 65,536 a fold stage bakes
            graph_edges[edge_c] = ("complete", n_entities)
        else:
            graph_edges[edge_c] = ([s_ for s_ in range(n_entities) for _ in r.table[s_]], [t for s_ in range(n_entities) for t in r.table[s_]])
    n_folds = len(folds)
    all_cols = world_cols + [c for r in probe.requests for c in (f"hlo_fold{r.index}_own", f"hlo_fold{r.index}_nbr", f"hlo_fold{r.index}_out")]

    # ---- the systems: ONE evaluation over the program's own leaves when system 0 is traced, every later system reads its share ----
    cache: Dict[str, object] = {}

    def phases(cols):
        vec = lambda v: list(v.e) if isinstance(v, _dsl.Vec) else [v]
        ev, res = evaluate(cols, lambda k, w: vec(cols[f"hlo_fold{k}_out"]))
        if len(ev.requests) != n_folds:
            raise NotEntityParallel("internal: the traced evaluation found another number of scans than the discovery pass")
        out = []
        for r in ev.requests:
            pad = lambda xs, c: _dsl.Vec(list(xs) + [_dsl.const(0.0)] * (widths[c] - len(xs)))
            out.append({f"hlo_fold{r.index}_own": pad(r.own_exprs, f"hlo_fold{r.index}_own"), f"hlo_fold{r.index}_nbr": pad(r.nb_exprs, f"hlo_fold{r.index}_nbr")})
        out.append(res)
        return out

    def make_system(k):
        def fn(**cols):
            if k == 0 or "phases" not in cache:
                cache["phases"] = phases(cols)
            return cache["phases"][k]
        fn.__name__ = f"{name}_{k}"
        import inspect
        fn.__signature__ = inspect.Signature([inspect.Parameter(p_, inspect.Parameter.KEYWORD_ONLY) for p_ in all_cols])
        system_ = _dsl.system(fn, **{c: widths[c] for c in all_cols})
        system_.float32_refused = ["a whole-world tick with fold stages is a float64 program (its fold kernels gather doubles)"]
        system_.body_free = True          # every slot of the world is a column of the program: no link of the chain touches a Body column
        system_.fp_contract = arith == "relaxed"
        return system_
    pre = []
    for k in range(n_folds):
        pre += [make_system(k), folds[k]]
    pre.append(make_system(n_folds))
    prog = _dsl.Program(pre, _dsl.Pipe([]), [])
    manifest = {"mode": "folds", "rows": "entities", "entities_per_world": n_entities, "rows_per_world": n_entities, "fold_stages": n_folds,
                **({"arith": "relaxed"} if arith == "relaxed" else {}),
                "edges_per_fold": [sum(len(t_) for t_ in r.table) for r in probe.requests],
                "columns": [{"column": c, "width": widths[c],
                             "component": next((s_.component for s_ in ins + outs if s_.column == c), None),
                             "component_id": next((s_.component_id for s_ in ins + outs if s_.column == c), None),
                             "entity_axis_elided": next((s_.elided for s_ in ins + outs if s_.column == c), False),
                             "scratch": c not in world_cols} for c in all_cols]}
    return prog, manifest, graph_edges


def edges_as_entity_ids(graph_edges: dict, entity_ids) -> dict:
    """world_program's edges (ROWS of one world) as what HipExec(graph_edges=) takes: entity ids of the executor's rows; the complete
    graph's marker passes through."""
    ids = np.asarray(entity_ids)
    return {k: (v if isinstance(v[0], str) else (ids[np.asarray(v[0], dtype=np.int64)], ids[np.asarray(v[1], dtype=np.int64)])) for k, v in graph_edges.items()}


def slots_from_metadata(doc: dict):
    """(argument slots, result slots) from a JSON document: either the reference's ExecMetadata as serde writes it —
    {"arg_ids": [...], "ret_ids": [...], "arg_slots": [{"component_id", "shape", "entity_axis_elided"}]}, optionally with
    "names": {"<id>": "world_pos"} — or the plain {"inputs": [{"component", "shape", "entity_axis_elided"}], "outputs": [...]}."""
    if "arg_slots" in doc:
        names = {int(k): v for k, v in doc.get("names", {}).items()}
        args = [Slot(names.get(int(s_["component_id"]), f"c{s_['component_id']}"), s_["shape"], s_["entity_axis_elided"], int(s_["component_id"]))
                for s_ in doc["arg_slots"]]
        seen, uniq = set(), []
        for a in args:                                    # CraneliftExec::new keeps the first slot of a repeated id (cranelift_exec.rs:66-72)
            if a.component_id not in seen:
                seen.add(a.component_id)
                uniq.append(a)
        by_id = {a.component_id: a for a in uniq}
        rets = [by_id.get(int(i)) or Slot(names.get(int(i), f"c{i}"), [], False, int(i)) for i in doc.get("ret_ids", [])]
        return uniq, rets
    ins = [Slot.of(x) for x in doc["inputs"]]
    outs = [Slot.of(x) for x in doc["outputs"]] if "outputs" in doc else list(ins)
    return ins, outs


def world_system(text: str, slots: Sequence, out_slots: Optional[Sequence] = None, mode: str = "auto", name: str = "world_tick",
                 every: int = 1, arith: str = "reference", one_world: bool = False):
    """A whole-world StableHLO tick (entity-batched `[N, w]` arguments) as ONE system of the generated kernel.  -> (system, manifest)

    slots / out_slots: per @main argument / result a Slot (or (component, shape, entity_axis_elided), or ExecSlotMetadata dicts);
    results default to the arguments' slots in order.  mode:
      "lane"  — one lane per ENTITY: the entity axis of every batched slot becomes the executor's row axis, singleton slots
                (entity axis elided: Globals) are replicated per row; statements that move data between entities are refused;
      "world" — one lane per WORLD: every slot is flattened into one row of a column, the executor's rows are independent worlds
                (a Monte-Carlo of small worlds — edge folds, joins and everything else are just index arithmetic inside a lane);
      "auto"  — "lane" when the tick is entity-parallel, else "world".
    The manifest says which it became and lists the program's columns in binding order.
    arith: "reference" — the module's arithmetic operation for operation (bit for bit the oracle's on the golden worlds);
           "relaxed"   — dsl.relaxed_arithmetic: finite values assumed (`0 * x` = 0, so a read that only feeds one disappears),
                         one division per denominator, `a * b + c` contracted: inside 1e-9, not the reference's last bits.
    one_world (lane mode): the caller promises that the executor's rows are the entities of ONE world — the singleton slots
           (Globals: tick, dt), replicated per row, then hold one value in every row and the kernel reads them once per
           wavefront (16 B per entity-tick less at configs[1]).  Not for a Monte-Carlo of worlds stacked in one executor."""
    if arith not in ("reference", "relaxed"):
        raise ValueError("arith must be 'reference' or 'relaxed'")
    funcs = parse_module(text)
    main = funcs["main"]
    ins = [Slot.of(x) for x in slots]
    outs = [Slot.of(x) for x in out_slots] if out_slots is not None else list(ins)
    if len(ins) != len(main.args) or len(outs) != len(main.result_types):
        raise ValueError(f"@main has {len(main.args)} arguments / {len(main.result_types)} results; {len(ins)} / {len(outs)} slots given")
    for s_, (_, ty) in zip(ins, main.args):
        if tuple(s_.shape) != tuple(ty.shape):
            raise ValueError(f"slot {s_.component}: shape {s_.shape} but @main takes {ty}")
    for k, (s_, ty) in enumerate(zip(outs, main.result_types)):
        if not s_.shape and ty.shape:
            outs[k] = Slot(s_.component, ty.shape, False, s_.component_id)       # a result-only component: shape from the module
        elif tuple(outs[k].shape) != tuple(ty.shape):
            raise ValueError(f"result slot {s_.component}: shape {s_.shape} but @main returns {ty}")
    counts = {s_.shape[0] for s_ in ins + outs if not s_.elided and s_.shape}
    n_entities = counts.pop() if len(counts) == 1 else None
    stride = None
    if n_entities and 1 < n_entities <= 64:
        stride = 1 << (n_entities - 1).bit_length()          # rows per world should the tick exchange data between its entities (<= one wavefront)
    used = {"exchanges": 0}
    hazards = float32_hazards(funcs)
    # integer components of the world (a tick counter) live in columns of the program's element type: in a float32 build they are
    # exact below 2^24 = 16,777,216 — said in the manifest, not refused (38 hours of 120 Hz ticks)
    int_cols = sorted({s_.column for s_, (_, ty) in zip(ins, main.args) if ty.dtype[0] in "iu" and ty.dtype != "i1" and _bits(ty.dtype) >= 32})

    def build(lane: bool):
        def width(s_: Slot) -> int:
            shp = s_.shape[1:] if (lane and not s_.elided) else s_.shape
            return int(np.prod(shp)) if shp else 1
        widths = {}
        for s_ in ins + outs:
            if widths.setdefault(s_.column, width(s_)) != width(s_):
                raise ValueError(f"component {s_.component} appears with two shapes")
        too_wide = [c for c, w in widths.items() if w > _dsl._MAT_MAX_ELEMS]
        if too_wide:
            raise NotImplementedError(f"columns {too_wide} are wider than {_dsl._MAT_MAX_ELEMS} values: this world is too large for one lane per world")
        params = list(dict.fromkeys([s_.column for s_ in ins] + [s_.column for s_ in outs]))

        def fn(**cols):
            args = []
            for s_, (_, ty) in zip(ins, main.args):
                v = cols[s_.column]
                elems = list(v.e) if isinstance(v, _dsl.Vec) else [v]
                batched = lane and not s_.elided
                stored = ((1,) + tuple(ty.shape[1:])) if batched else tuple(ty.shape)
                arr = np.empty(len(elems), dtype=object)
                arr[:] = elems
                arr = arr.reshape(stored)
                if ty.dtype == "i1":
                    arr = _emap(lambda e: e > 0.5, arr)
                elif ty.dtype == "ui64":
                    arr = _emap(U64.of, arr)
                args.append(Sym(arr, ty.dtype, 0 if batched else None, (), ty.shape))
            ev = _LaneEval(funcs, n_entities, stride) if lane else _Eval(funcs)
            res = {}
            if arith == "relaxed":
                with _dsl.relaxed_arithmetic():
                    results = ev.call(main, args)
            else:
                results = ev.call(main, args)
            for s_, o in zip(outs, results):
                if lane:
                    o = ev._mat(o, [d for d in o.uni if d != 0] if not s_.elided else None)
                    if s_.elided and o.eaxis is not None:
                        raise NotEntityParallel(f"the singleton component {s_.component} would become per-entity")
                    if not s_.elided and o.eaxis not in (0, None):
                        raise NotEntityParallel(f"result {s_.component}: the entity axis is not its leading axis")
                    if not s_.elided and o.eaxis is None and 0 not in o.uni and o.tshape and o.tshape[0] > 1:
                        raise NotEntityParallel(f"result {s_.component} is per-entity data that does not come from a column of the world")
                vals = _column_values(o)
                src = cols.get(s_.column)
                src = (list(src.e) if isinstance(src, _dsl.Vec) else [src]) if src is not None else None
                if src is not None and len(src) == len(vals) and all(a is b for a, b in zip(src, vals)):
                    continue                          # the tick hands the column back untouched (inertia, a parameter column): no store
                res[s_.column] = _dsl.Vec(vals)
            # the reads the kernel executes: lane_read nodes the stored results reach (the evaluator builds many more that fold away —
            # rows gathered one source at a time and merged back, carried tensors a loop never uses)
            seen, live, todo = set(), 0, [e for v in res.values() for e in v.e if isinstance(e, Expr)]
            while todo:
                x = todo.pop()
                if id(x) in seen:
                    continue
                seen.add(id(x))
                live += x.op in ("lane_read", "lane_read_dyn")
                todo.extend(a for a in x.args if isinstance(a, Expr))
                if x.op == "while":                  # ... and what its condition and body read
                    todo.extend([x.value[1], *x.value[2]])
            used["exchanges"] = live
            return res
        fn.__name__ = name
        import inspect
        fn.__signature__ = inspect.Signature([inspect.Parameter(p, inspect.Parameter.KEYWORD_ONLY) for p in params])
        system_ = _dsl.system(fn, every=every, **widths)
        system_.body_free = True          # every slot of the world is a column of the program: no Body column is read or written
        system_.float32_refused = hazards     # dsl / codegen refuse a float32 build of a program that holds this system
        system_.fp_contract = arith == "relaxed"
        if one_world and lane:
            system_.uniform = tuple(s_.column for s_ in ins + outs if s_.elided)
        manifest = {"mode": "lane" if lane else "world", "rows": "entities" if lane else "worlds",
                    **({"float32_refused": hazards} if hazards else {}),
                    **({"arith": "relaxed"} if arith == "relaxed" else {}),
                    **({"one_world": True} if (one_world and lane) else {}),
                    **({"float32_integer_columns": int_cols} if int_cols else {}),
                    "entities_per_world": n_entities,
                    "columns": [{"column": c, "width": widths[c],
                                 "component": next(s_.component for s_ in ins + outs if s_.column == c),
                                 "component_id": next(s_.component_id for s_ in ins + outs if s_.column == c),
                                 "shape": list(next(s_.shape for s_ in ins + outs if s_.column == c)),
                                 "entity_axis_elided": next(s_.elided for s_ in ins + outs if s_.column == c),
                                 "argument": next((k for k, s_ in enumerate(ins) if s_.column == c), None),
                                 "result": next((k for k, s_ in enumerate(outs) if s_.column == c), None)} for c in params]}
        return system_, manifest, widths

    if mode not in ("auto", "lane", "world"):
        raise ValueError("mode must be 'auto', 'lane' or 'world'")
    if mode in ("auto", "lane"):
        try:
            if n_entities is None:
                raise NotEntityParallel("the batched slots do not share one entity count (components on different entity sets)")
            system_, manifest, widths = build(True)
            _dsl.Program([system_], _dsl.Pipe([]), []).trace(widths)      # refusals surface while tracing: find out now
            if used["exchanges"]:
                # the tick reads other entities of the same world (joins, an edge_fold's targets): a world must be `stride`
                # consecutive rows of the executor — its N entities, then stride - N rows of padding — so that it never straddles a
                # wavefront and the reads are lane exchanges (dsl op lane_read)
                manifest["rows_per_world"] = stride
                manifest["exchange_reads"] = used["exchanges"]
            return system_, manifest
        except NotEntityParallel as e:
            if mode == "lane":
                raise
            reason = str(e)
    system_, manifest, _ = build(False)
    if mode == "auto":
        manifest["lane_refused"] = reason
    return system_, manifest


_F32_BIT_OPS = ("bitcast_convert", "shift_left", "shift_right_logical", "shift_right_arithmetic", "popcnt", "count_leading_zeros",
                "rng_bit_generator")
_F32_EXACT = float(1 << 24)      # integers above 2^24 are not all representable in an f32 lane


def float32_hazards(funcs: Dict[str, Func]) -> List[str]:
    """Why a module must not be built with dtype float32, or [] when it may.  The evaluator carries every integer tensor as an
    integral FLOAT of the program's element type (i32 / ui32 PRNG words, ui64 as two 32-bit halves, bitcasts, shifts, _mul_lo32):
    exact in f64, where 2^53 covers 32-bit words and their partial products — in an f32 program anything at or above 2^24 rounds,
    and the float <-> bits nodes (m_bits2f / m_fbits) exist for doubles only.  So an f32 build is refused for a module that
    (a) reinterprets, shifts or counts bits, (b) combines integers with and / or / xor, (c) multiplies 32-bit-or-wider integers, or
    (d) holds an integer constant of 2^24 or more; integer COLUMNS of @main (a tick counter) are reported by world_system in the
    manifest (`float32_integer_columns`: exact below 16,777,216), small loop counters and gather indices pass."""
    why: List[str] = []

    def is_wide_int(ty: TensorType) -> bool:
        return ty.dtype[0] in "iu" and ty.dtype != "i1" and _bits(ty.dtype) >= 32

    def walk(ops: List[Op], where: str):
        for op in ops:
            short = op.name.strip('"').split(".")[-1]
            try:
                rts = _Eval._result_types(op.text)
            except Exception:  # noqa: BLE001 - an op this scan cannot type is typed (or refused) by the evaluator itself
                rts = []
            ints = [t for t in rts if is_wide_int(t)]
            if short in _F32_BIT_OPS:
                why.append(f"@{where}: stablehlo.{short} works on bit patterns")
            elif short in ("and", "or", "xor", "not") and ints:
                why.append(f"@{where}: stablehlo.{short} on {ints[0]}")
            elif short == "multiply" and ints:
                why.append(f"@{where}: integer multiply on {ints[0]} (products pass 2^24)")
            elif short == "constant" and ints:
                try:
                    vals = np.asarray(_Eval._constant(op.text, rts[0]).a, dtype=object).ravel()
                    big = [v for v in vals if not isinstance(v, U64) and abs(float(_try_const(v) if _try_const(v) is not None else 0.0)) >= _F32_EXACT]
                    if big or any(isinstance(v, U64) for v in vals):
                        why.append(f"@{where}: integer constant of 2^24 or more ({rts[0]})")
                except Exception:  # noqa: BLE001
                    why.append(f"@{where}: integer constant {rts[0]} could not be bounded")
            for region in op.regions:
                walk(region, where)

    for name, fn in funcs.items():
        walk(fn.body, name)
    return list(dict.fromkeys(why))


def compile_world(text: str, slots_doc: dict, out: Optional[str] = None, mode: str = "auto", dtype: str = "float64", fast_math: bool = False,
                  wave_folds: bool = True, arith: str = "reference", one_world: bool = False):
    """The build-time step a host (`WorldExec::Hip`) runs once per world: module text + slot metadata -> the
    shared object `sixdof_set_custom_pipe` installs, and the manifest of its columns.  -> (path of the .so, manifest)"""
    import json
    import shutil
    import time
    from . import codegen
    t0 = time.perf_counter()
    ins, outs = slots_from_metadata(slots_doc)
    folds = mode == "folds"
    if not folds:
        try:
            system_, manifest = world_system(text, ins, outs, mode=mode, arith=arith, one_world=one_world)
        except (NotEntityParallel, NotImplementedError) as refused:
            # a world of more than a wavefront whose entities exchange data fits neither one lane per entity (the exchange leaves
            # the wavefront) nor one lane per world (too wide): its scans over the edge slot become fold stages (world_program)
            if mode != "auto":
                raise
            try:
                built = world_program(text, ins, outs, wave_folds=wave_folds, arith=arith)
            except NotEntityParallel:
                raise refused from None
            folds = True
    if folds:
        if dtype != "float64":
            raise NotImplementedError("whole-world ticks with fold stages are float64 (the fold kernels gather doubles)")
        if one_world:
            raise NotImplementedError("one_world applies to one-kernel ticks (mode lane), not to fold stages")
        prog, manifest, edges = built if mode == "auto" else world_program(text, ins, outs, wave_folds=wave_folds, arith=arith)
        manifest["folds"] = "a wave per source (lane partials + shuffle tree) for additive scans of 64 edges or more" if wave_folds else "sequential, one lane per source"
        n_world = int(manifest["entities_per_world"])
        rows = int(slots_doc.get("rows", 0)) or n_world
        if rows % n_world:
            raise ValueError(f"{rows} rows are not a whole number of {n_world}-entity worlds")
        widths = {c["column"]: c["width"] for c in manifest["columns"]}
        tp = prog.trace(widths, fold_edges=edges, fold_replicas=(rows // n_world, n_world) if rows > n_world else None)
        t1 = time.perf_counter()
        so = codegen.build(tp, dtype, 2, fast_math=False)
        t2 = time.perf_counter()
        known = {c["column"] for c in manifest["columns"]}
        manifest["columns"] += [{"column": n_, "width": w_, "component": None, "component_id": None, "entity_axis_elided": False, "scratch": True}
                                for n_, w_ in tp.columns if n_ not in known]          # the folds' commit buffers
        order = [n_ for n_, _ in tp.columns]
        manifest["columns"] = sorted(manifest["columns"], key=lambda c: order.index(c["column"]))
        manifest.update({"integrator": "none", "dtype": dtype, "column_layout": "rows", "row_count": rows,      # the fold kernels are generated for this many rows
                         "build": {"trace_ms": round((t1 - t0) * 1e3, 1), "compile_ms": round((t2 - t1) * 1e3, 1), "resources": dict(codegen.last_resources)}})
        if out:
            shutil.copyfile(so, out)
            Path(str(out) + ".json").write_text(json.dumps(manifest, indent=1))
            so = Path(out)
        return so, manifest
    if dtype == "float32" and manifest.get("float32_refused"):
        raise NotImplementedError("this module cannot be built with dtype float32 (integer tensors are carried as floats: exact in f64 only): "
                                  + "; ".join(manifest["float32_refused"][:4]))
    widths = {c["column"]: c["width"] for c in manifest["columns"]}
    tp = _dsl.Program([system_], _dsl.Pipe([]), []).trace(widths)
    t1 = time.perf_counter()
    rows = int(slots_doc.get("rows", 0))
    soa = bool(rows >= codegen.COLUMN_SOA_MIN_ROWS)
    so = codegen.build(tp, dtype, 2, fast_math=fast_math, column_soa=soa)          # 2 = SIXDOF_INTEGRATOR_NONE: the module IS the tick
    t2 = time.perf_counter()
    order = [n for n, _ in tp.columns]
    manifest["columns"] = sorted(manifest["columns"], key=lambda c: order.index(c["column"]))     # = the aux_component_ids order
    manifest.update({"integrator": "none", "dtype": dtype, "column_layout": "element-major" if soa else "rows",
                     "build": {"trace_ms": round((t1 - t0) * 1e3, 1), "compile_ms": round((t2 - t1) * 1e3, 1),
                               "resources": dict(codegen.last_resources)}})
    if out:
        shutil.copyfile(so, out)
        Path(str(out) + ".json").write_text(json.dumps(manifest, indent=1))
        so = Path(out)
    return so, manifest


def load_world(so_path: str):
    """What compile_world / the CLI wrote, as a program an executor can bind: (dsl.FrozenProgram, manifest).  Nothing is traced,
    generated or compiled — the object is installed as it is (sixdof_set_custom_pipe)."""
    import json
    manifest = json.loads(Path(str(so_path) + ".json").read_text())
    prog = _dsl.FrozenProgram(None, [(c["column"], c["width"]) for c in manifest["columns"]],
                              column_soa=manifest.get("column_layout") == "element-major", prebuilt_so=str(so_path))
    prog._traced.rows_multiple = int(manifest.get("rows_per_world", 1))      # exec.HipExec refuses a row count that splits a world
    prog._traced.exact_rows = int(manifest.get("row_count", 0))              # ... and, for an object with fold stages, any count but the one it was generated for
    return prog, manifest


_NP_DTYPE = {"f64": "float64", "f32": "float32", "i64": "int64", "i32": "int32", "ui64": "uint64", "ui32": "uint32", "i1": "bool", "ui8": "uint8", "i8": "int8"}


def checkpoint(debug_dir: str, mode: str = "auto", device: int = 0, rtol: float = 1e-9) -> dict:
    """The reference's first-tick parity harness (libs/cranelift-mlir/tests/checkpoint_test.rs; producer cranelift_compile.rs:70-153,
    cranelift_exec.rs:199-267), with this backend in Cranelift's place.  `ELODIN_CRANELIFT_DEBUG_DIR=<dir>` makes a reference run dump
    `stablehlo.mlir`, `input_<i>.bin` (the raw column bytes of @main's arguments) and `xla_output_<i>.bin` / `cranelift_output_<i>.bin`
    (the first tick's results).  This function compiles that module for gfx950, runs ONE tick on the same inputs, writes
    `hip_output_<i>.bin` next to them and compares: integers bit for bit, floats to `rtol` of each output's largest magnitude.
    -> {"mode", "outputs": [{"index", "max_rel_err" | "equal", "against"}], "ok"}"""
    import json
    d = Path(debug_dir)
    text = (d / "stablehlo.mlir").read_text()
    main = parse_module(text)["main"]
    ins = []
    for k, (_, ty) in enumerate(main.args):
        raw = np.fromfile(d / f"input_{k}.bin", dtype=_NP_DTYPE[ty.dtype])
        if raw.size != ty.size:
            raise ValueError(f"input_{k}.bin holds {raw.size} elements, @main's argument {k} is {ty}")
        ins.append(raw.reshape(ty.shape))
    meta = json.loads((d / "slots.json").read_text()) if (d / "slots.json").exists() else None
    if meta is not None:
        arg_slots, ret_slots = slots_from_metadata(meta)
    else:       # no ExecMetadata beside the dump: the modal leading size of the rank >= 2 arguments is taken for the entity count,
        #             and checkpoint.json (cranelift_exec.rs:211-243) says which result is which argument's component
        lead = [ty.shape[0] for _, ty in main.args if len(ty.shape) >= 2]
        n = max(set(lead), key=lead.count) if lead else None
        ck = json.loads((d / "checkpoint.json").read_text()) if (d / "checkpoint.json").exists() else {}
        in_ids = {int(e["index"]): int(e["component_id"]) for e in ck.get("inputs", [])}
        out_ids = {int(e["index"]): int(e["component_id"]) for e in ck.get("outputs", [])} if len(ck.get("outputs", [])) == len(main.result_types) else {}
        name = lambda ids, k, stem: f"c{ids[k]}" if k in ids else f"{stem}{k}"
        arg_slots = [Slot(name(in_ids, k, "arg"), ty.shape, not (len(ty.shape) >= 2 and ty.shape[0] == n)) for k, (_, ty) in enumerate(main.args)]
        ret_slots = [Slot(name(out_ids, k, "ret"), ty.shape, not (len(ty.shape) >= 2 and ty.shape[0] == n)) for k, ty in enumerate(main.result_types)]
    system_, manifest = world_system(text, arg_slots, ret_slots, mode=mode)
    lane = manifest["mode"] == "lane"
    n_ent = manifest["entities_per_world"] if lane else 1
    rows = manifest.get("rows_per_world", n_ent) if lane else 1
    cols = {}
    for s_, a in zip(arg_slots, ins):
        v = np.asarray(a, dtype=np.float64)
        if lane and not s_.elided:
            cols[s_.column] = np.zeros((rows, v.size // n_ent))
            cols[s_.column][:n_ent] = v.reshape(n_ent, -1)           # the world's entities, then padding rows
        else:
            cols[s_.column] = np.tile(v.reshape(1, -1), (rows, 1))
    for s_ in ret_slots:
        w = next(c["width"] for c in manifest["columns"] if c["column"] == s_.column)
        cols.setdefault(s_.column, np.zeros((rows, w)))
    from . import _lib as L
    from .exec import HipExec
    ident = np.tile([0, 0, 0, 1.0, 0, 0, 0], (rows, 1))
    hip = HipExec(ident, np.zeros((rows, 6)), np.ones((rows, 7)), integrator=L.INTEGRATOR_NONE, effectors=_dsl.Program([system_], _dsl.Pipe([]), []),
                  columns=cols, device=device)
    try:
        hip.run(1)
        report = {"mode": manifest["mode"], "outputs": [], "ok": True}
        for k, (s_, ty) in enumerate(zip(ret_slots, main.result_types)):
            got = np.asarray(hip._aux[s_.column], dtype=np.float64)
            got = (got[:n_ent].reshape(ty.shape) if (lane and not s_.elided) else got[0].reshape(ty.shape))
            np.ascontiguousarray(got.astype(_NP_DTYPE[ty.dtype])).tofile(d / f"hip_output_{k}.bin")
            for ref_name in ("xla_output", "cranelift_output"):
                f_ = d / f"{ref_name}_{k}.bin"
                if not f_.exists():
                    continue
                want = np.fromfile(f_, dtype=_NP_DTYPE[ty.dtype]).reshape(ty.shape)
                if ty.dtype[0] in "iu":
                    rec = {"index": k, "against": ref_name, "equal": bool(np.array_equal(got.astype(want.dtype), want))}
                    report["ok"] &= rec["equal"]
                else:
                    scale = max(float(np.max(np.abs(want))) if want.size else 0.0, 1e-300)
                    err = float(np.max(np.abs(got - want)) / scale) if want.size else 0.0
                    rec = {"index": k, "against": ref_name, "max_rel_err": err}
                    report["ok"] &= bool(err <= rtol or not np.isfinite(want).all())
                report["outputs"].append(rec)
    finally:
        hip.close()
    (d / "hip_checkpoint.json").write_text(json.dumps(report, indent=1))
    return report


def _main(argv=None) -> int:
    """python -m elodin_amd.stablehlo module.mlir --slots slots.json -o pipe.so [--mode auto|lane|world|folds] [--dtype float64|float32]"""
    import argparse
    import json
    ap = argparse.ArgumentParser(prog="python -m elodin_amd.stablehlo", description=_main.__doc__)
    ap.add_argument("module", nargs="?", help="StableHLO text (ELODIN_CRANELIFT_DEBUG_DIR/stablehlo.mlir, libs/nox-py/src/cranelift_compile.rs:58-60)")
    ap.add_argument("--slots", help="ExecMetadata JSON (libs/nox-py/src/exec.rs:17-29) or {inputs, outputs}")
    ap.add_argument("-o", "--out", help="shared object to write (its manifest goes to <out>.json)")
    ap.add_argument("--checkpoint", metavar="DIR", help="a reference debug dump (ELODIN_CRANELIFT_DEBUG_DIR): run its first tick on the GPU and compare "
                                                        "with xla_output_<i>.bin / cranelift_output_<i>.bin (needs a GPU)")
    ap.add_argument("--mode", default="auto", choices=("auto", "lane", "world", "folds"))
    ap.add_argument("--dtype", default="float64", choices=("float64", "float32"))
    ap.add_argument("--fast-math", action="store_true")
    ap.add_argument("--sequential-folds", action="store_true", help="fold stages (--mode folds) with one lane per source in slot order: the reference's "
                                                                     "association bit for bit, N - 1 dependent trips per scan (default: a wave per source for long additive scans)")
    ap.add_argument("--arith", default="reference", choices=("reference", "relaxed"),
                    help="relaxed: finite values assumed (0 * x = 0), one division per denominator, a * b + c contracted — inside 1e-9 of the "
                         "reference instead of its last bits, about half the instructions (world_system)")
    ap.add_argument("--one-world", action="store_true", help="lane mode: the executor's rows are ONE world, its Globals are read once per wavefront")
    a = ap.parse_args(argv)
    if a.checkpoint:
        rep = checkpoint(a.checkpoint, a.mode)
        print(json.dumps(rep))
        return 0 if rep["ok"] else 1
    if not (a.module and a.slots and a.out):
        ap.error("module, --slots and -o are required (or --checkpoint DIR)")
    so, manifest = compile_world(Path(a.module).read_text(), json.loads(Path(a.slots).read_text()), a.out, a.mode, a.dtype, a.fast_math,
                                 wave_folds=not a.sequential_folds, arith=a.arith, one_world=a.one_world)
    print(json.dumps({"object": str(so), "mode": manifest["mode"], "rows": manifest["rows"], "columns": [c["column"] for c in manifest["columns"]],
                      "build": manifest["build"], **({k: manifest[k] for k in ("arith", "one_world") if k in manifest}), **({"lane_refused": manifest["lane_refused"]} if "lane_refused" in manifest else {})}))
    return 0


if __name__ == "__main__":
    raise SystemExit(_main())
