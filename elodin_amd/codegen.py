"""Code generator: traced effector pipe (elodin_amd/dsl.py) -> HIP source -> gfx950 shared object.

The generated translation unit defines `PipeCustom` (same interface as the built-in pipes in
csrc/effectors.hpp) whose `apply` is the user's effector code as straight-line scalar arithmetic, and
instantiates csrc/step_kernel.hpp's fused six_dof kernel with it for the requested dtype / integrator
(both cache-policy variants).  Built with hipcc (the same toolchain as the library; present on the GPU
box too) into elodin_amd/_jit/pipe_<hash>.so and handed to the C ABI with sixdof_set_custom_pipe —
the MI355X analogue of the reference's build-time JIT (cranelift_compile.rs:13-162; its
`build_time_ms` is the cost this step corresponds to).
"""
from __future__ import annotations

import hashlib
import os
import subprocess
from pathlib import Path
from typing import Optional, Dict, List, Sequence

from . import dsl

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
JIT_DIR = PKG / "_jit"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

_LEAF_CPP = {
    "qi": "b.q.i", "qj": "b.q.j", "qk": "b.q.k", "qw": "b.q.w", "px": "b.p.x", "py": "b.p.y", "pz": "b.p.z",
    "wx": "b.v.ang.x", "wy": "b.v.ang.y", "wz": "b.v.ang.z", "vx": "b.v.lin.x", "vy": "b.v.lin.y", "vz": "b.v.lin.z",
    "Ix": "b.I.x", "Iy": "b.I.y", "Iz": "b.I.z", "mass": "b.mass",
}
_BIN = {"add": "+", "sub": "-", "mul": "*", "div": "/"}
_FN1 = {"sqrt": "m_sqrt", "abs": "m_abs", "sin": "m_sin", "cos": "m_cos", "tan": "m_tan", "exp": "m_exp",
        "log": "m_log", "acos": "m_acos", "asin": "m_asin", "log1p": "m_log1p", "expm1": "m_expm1", "cbrt": "m_cbrt",
        "floor": "m_floor", "ceil": "m_ceil", "trunc": "m_trunc", "rint": "m_rint", "sinh": "m_sinh", "cosh": "m_cosh",
        "erfc": "m_erfc", "isfinite": "m_isfinite", "erfinv": "m_erfinv"}
_FN2 = {"max": "m_max", "min": "m_min", "atan2": "m_atan2", "hypot": "m_hypot", "pow": "m_pow", "mod": "m_mod",
        "bxor": "m_bxor", "bor": "m_bor", "band": "m_band", "shl": "m_shl", "shr": "m_shr", "bits2f": "m_bits2f"}
_FN1.update({"bits2f32": "m_bits2f32", "f32bits": "m_f32bits"})
_BOOL_OPS = {"lt", "le", "eq", "and", "or", "not", "isfinite"}


def _literal(v: float) -> str:
    if v != v:            # a StableHLO module may carry them: `dense<0x7FF8000000000000>` (jnp.linalg.solve's failure value), the
        return "T(__builtin_nan(\"\"))"      # -inf init of a max reduction the front end could not fold away
    if v in (float("inf"), float("-inf")):
        return "T(__builtin_inf())" if v > 0 else "T(-__builtin_inf())"
    return f"T({v!r})"


# leaf -> C++ lvalue, per context
_APPLY_LEAVES = dict(_LEAF_CPP)                                   # effector stage: the stage body `b`
_SYSTEM_LEAVES = {"qi": "q.i", "qj": "q.j", "qk": "q.k", "qw": "q.w", "px": "p.x", "py": "p.y", "pz": "p.z",
                  "wx": "v.ang.x", "wy": "v.ang.y", "wz": "v.ang.z", "vx": "v.lin.x", "vy": "v.lin.y", "vz": "v.lin.z",
                  "Ix": "I.x", "Iy": "I.y", "Iz": "I.z", "mass": "mass", "tick": "T(tick)",
                  # world_accel of the tick just integrated: post systems only (an IMU model after six_dof)
                  "aax": "accel.ang.x", "aay": "accel.ang.y", "aaz": "accel.ang.z", "alx": "accel.lin.x", "aly": "accel.lin.y", "alz": "accel.lin.z"}


def _leaf_ref(name: str, table: Dict[str, str]) -> str:
    if name in table:
        return table[name]
    if name.startswith("aux"):
        slot, k = name[3:].split("_")
        return f"aux[{slot}].{'xyz'[int(k)]}"
    if name.startswith("c"):
        slot, k = name[1:].split("_")
        return f"r.c{slot}[{k}]"
    if name.startswith("wst"):      # window push: row `head` (the oldest) of the ring; `@w_act@`: store only from lanes that own a row
        slot, j = (int(x) for x in name[3:].split("_"))
        _, width, head_slot = _WINDOWS[slot]
        return f"@w_act@W{slot}[(size_t)(static_cast<int>(r.c{head_slot}[0]) * {width} + {j}) * w_n]"
    raise KeyError(name)


_TABLES: Dict[tuple, str] = {}    # (xp, fp) -> C++ symbol stem, filled while emitting one translation unit
_GATHERS: Dict[str, str] = {}     # dsl gather-table key -> C++ symbol of its __device__ const array, same lifetime
_LANE_TABLES: Dict[tuple, str] = {}    # (stride, source entity per entity index) of a lane_read in a world of 32 / 64 rows -> symbol, same lifetime
_WINDOWS: Dict[int, tuple] = {}    # window slot -> (rows, width, head slot) of the program being emitted (dsl.Window)
# Device layout of a window column by executor size, fixed when the program is built (HipExec knows its row count; the object
# says which one it was built for, bit 30 of sixdof_custom_column_widths): from this many entities on it is ELEMENT-major, [rows*width][n] — a wave reads 512 contiguous bytes per element (65,536 rockets:
# 0.221 -> 0.158 ms per tick, 4.8 TB/s of window traffic); below, each entity's window is contiguous like in the reference's
# column — a small batch is a chain of dependent loads per lane and the neighbouring elements it finds in cache matter more
# (8,192 rockets: 0.070 ms per tick against 0.088 element-major).
WINDOW_SOA_MIN_ROWS = 32768
_WINDOW_SOA = [False]                # layout of the program being emitted
# Register columns of a program on the device: [n][w] (the reference's rows: a lane reads its w values at a stride of w
# elements, so every load instruction of a wave touches w times the cache lines it uses) or ELEMENT-MAJOR [w][n] (a wave's
# load of element j is 64 consecutive values: two full 128-byte lines per f32 instruction).  The generator owns this layout
# (include/sixdof_hip.h: the object exports it, bit 29 of its column widths); executors of COLUMN_SOA_MIN_ROWS rows or more
# without entity-set joins / folds use the element-major one (exec.py transposes at upload / download).
COLUMN_SOA_MIN_ROWS = 32768
_COLUMN_SOA = [False]


def _col_ptr(k: int, w: int, row: str, const: bool = True) -> str:
    """`g` such that element j of this lane's row of program column k is `g[_col_idx(j)]`."""
    ty = "const T*" if const else "T*"
    cast = f"static_cast<{ty}>(P.model_cols[{k}])"
    return f"{ty} g = {cast} + (size_t){row}" + ("" if _COLUMN_SOA[0] else f" * {w}") + ";"


def _col_idx(j: int) -> str:
    return f"(size_t){j} * P.n" if _COLUMN_SOA[0] else str(j)



class _Emitter:
    """Straight-line C++ for lists of (lvalue, Expr).  Every output of a block is computed into a temporary before any
    is written (a system may read a component it also writes); common sub-expressions are emitted once — also ACROSS the
    blocks of one function as long as none of the leaves they read has been written in between (two systems that both
    convert the same position to geodetic coordinates share the conversion)."""

    def __init__(self, leaves: Dict[str, str]):
        self.leaves = leaves
        self.names: Dict[int, str] = {}      # id(Expr) -> C++ temporary holding it
        self.keep: Dict[int, dsl.Expr] = {}   # keeps the Exprs alive so ids stay unique
        self.deps: Dict[int, frozenset] = {}
        self.wide: Dict[int, bool] = {}
        self.n = 0

    # jax.random's threefry words are uint32 values and its samples carry 52 random bits: everything between the key
    # words and the finished sample is evaluated in double, whatever the program's dtype — a float32 program casts the
    # seed / tick in, and the sample out (and so draws the SAME noise as its float64 twin, rounded).
    _WIDE_ARITH = {"add", "sub", "mul", "div", "neg", "floor", "max", "min", "erfinv"}

    def _is_wide(self, e: dsl.Expr) -> bool:
        w = self.wide.get(id(e))
        if w is None:
            if e.op == "threefry":
                w = True
            elif e.op in self._WIDE_ARITH:
                kinds = [("const" if a.op == "const" else self._is_wide(a)) for a in e.args]
                w = any(k is True for k in kinds) and all(k is True or k == "const" for k in kinds)
            else:
                w = False
            self.wide[id(e)] = w
            self.keep[id(e)] = e
        return w

    def _deps(self, e: dsl.Expr) -> frozenset:
        d = self.deps.get(id(e))
        if d is None:
            if e.op == "leaf":
                d = frozenset((e.name,))
            else:
                d = frozenset().union(*[self._deps(a) for a in e.args]) if e.args else frozenset()
                if e.op == "wload":      # memory-resident: goes stale when its window is pushed (pseudo-leaf win<slot>)
                    d = d | frozenset((f"win{e.value[0]}",))
                if e.op == "while":      # plus what the condition / body read from outside the loop
                    names, cond, body = e.value[:3]
                    inner = frozenset().union(self._deps(cond), *[self._deps(b) for b in body])
                    d = d | (inner - frozenset(names))
            self.deps[id(e)] = d
            self.keep[id(e)] = e
        return d

    def block(self, assign, indent: str = "        ", written: Sequence[str] = (), scoped: bool = False) -> List[str]:
        """assign: [(lvalue, Expr)]; written: the leaf names those lvalues correspond to (invalidates dependants);
        scoped: the block sits inside a conditional — temporaries created in it must not be reused outside."""
        lines: List[str] = []
        before = set(self.names)
        outer = {"parent": None, "names": self.names, "lines": lines, "vars": {}, "varset": frozenset(), "indent": indent}
        guards: Dict[int, tuple] = {}      # id(select) -> (guarded arm index, member node ids, selects of the group): _plan_guards
        fused: Dict[int, int] = {}         # id(add / sub) -> which argument is the product folded into it (_FUSE_FMA)

        def rhs_of(e: dsl.Expr, a: List[str]) -> str:
            if e.op == "div":
                return f"m_div({a[0]}, {a[1]})"
            if e.op in _BIN:
                return f"{a[0]} {_BIN[e.op]} {a[1]}"
            if e.op == "neg":
                return f"-{a[0]}"
            if e.op in _FN1:
                return f"{_FN1[e.op]}({a[0]})"
            if e.op in _FN2:
                return f"{_FN2[e.op]}({a[0]}, {a[1]})"
            if e.op == "select":
                return f"{a[0]} ? {a[1]} : {a[2]}"
            if e.op == "interp":
                stem = _TABLES.setdefault(e.value, f"tab{len(_TABLES)}")
                xs = e.value[0]
                step = (xs[-1] - xs[0]) / (len(xs) - 1)
                uniform = len(xs) > 32 and step > 0 and all(abs((b - a_) - step) <= 1e-9 * step for a_, b in zip(xs, xs[1:]))
                if uniform:   # evenly spaced long table: index by division instead of bisecting through memory
                    return f"m_interp_uniform<T, {len(xs)}>({a[0]}, {stem}_x, {stem}_f, T({1.0 / step!r}))"
                return f"m_interp<T, {len(xs)}>({a[0]}, {stem}_x, {stem}_f)"
            if e.op == "gather":    # constant table in device memory (dsl.gather): jax's index normalisation + clamp in m_gather
                key, col, n, w = e.value
                stem = _GATHERS.setdefault(key, f"gtab{len(_GATHERS)}")
                return f"m_gather<T>({stem}, {a[0]}, {n}, {w}, {col})"
            if e.op == "wload":     # logical row a[1] of the ring whose oldest row sits at physical row a[0]
                slot, rows, width, j, _ = e.value
                return f"W{slot}[(size_t)(((static_cast<int>({a[0]}) + static_cast<int>({a[1]})) % {rows}) * {width} + {j}) * w_n]"
            if e.op == "threefry":
                return f"m_threefry({a[0]}, {a[1]}, {a[2]}, {a[3]}, {int(e.value)})"
            if e.op == "lane_read":     # the value another entity of THIS lane's world holds (whole-world StableHLO ticks: a join / an
                # edge_fold's targets).  A world is `stride` consecutive rows, stride a power of two <= 64 dividing the 64-lane wave,
                # so the exchange never leaves the wavefront: one ds_bpermute per 32-bit half.  The source entity per entity index
                # is a compile-time table packed four bits each into one 64-bit constant.
                stride, table = e.value
                if len(set(table)) == 1:
                    src = f"static_cast<int>((threadIdx.x & ~{stride - 1}u) + {table[0]}u)"
                elif stride > 16:       # worlds of 32 or 64 rows: a byte per entity in constant memory (one cached load per read)
                    stem = _LANE_TABLES.setdefault((int(stride), tuple(int(j) for j in table)), f"ltab{len(_LANE_TABLES)}")
                    src = f"static_cast<int>((threadIdx.x & ~{stride - 1}u) + {stem}[threadIdx.x & {stride - 1}u])"
                else:
                    packed = sum((int(j) & 15) << (4 * i) for i, j in enumerate(table))
                    src = (f"static_cast<int>((threadIdx.x & ~{stride - 1}u) + static_cast<unsigned>(({packed}ull >> ((threadIdx.x & {stride - 1}u) * 4u)) & 15ull))")
                return f"__shfl({a[0]}, {src}, 64)"
            if e.op == "lane_read_dyn":     # the same exchange with the source table picked by a traced index (a scan's counter over the
                # edge slot: stablehlo.py _LaneEval._pick): slot-major bytes in constant memory
                stride, tables = e.value
                flat = tuple(int(j) for t in tables for j in t)
                stem = _LANE_TABLES.setdefault((int(stride), flat), f"ltab{len(_LANE_TABLES)}")
                return (f"__shfl({a[0]}, static_cast<int>((threadIdx.x & ~{stride - 1}u) + {stem}[static_cast<int>({a[1]}) * {stride} + "
                        f"static_cast<int>(threadIdx.x & {stride - 1}u)]), 64)")
            if e.op == "fbits":     # one 32-bit word of a double's bit pattern (stablehlo.bitcast_convert f64 -> ui64): 1 = high
                return f"m_fbits({a[0]}, {int(e.value)})"
            if e.op == "lt":
                return f"{a[0]} < {a[1]}"
            if e.op == "le":
                return f"{a[0]} <= {a[1]}"
            if e.op == "eq":
                return f"{a[0]} == {a[1]}"
            if e.op == "and":
                return f"{a[0]} && {a[1]}"
            if e.op == "or":
                return f"{a[0]} || {a[1]}"
            if e.op == "not":
                return f"!{a[0]}"
            raise ValueError(f"unsupported op {e.op}")

        def emit_loop(node: dsl.Expr, sc) -> List[str]:
            """A data-dependent loop (dsl.lax.while_loop): carried values live in mutable locals of the scope the loop
            belongs to; everything in the condition / body that does not depend on them is hoisted out."""
            names, cond, body, max_iter = node.value[:4]
            counted = node.value[4] if len(node.value) > 4 else None
            inits = [ref(x, sc) for x in node.args]
            cvars = {}
            for nm, init in zip(names, inits):
                cvars[nm] = f"{nm}_{self.n}"
                self.n += 1
                sc["lines"].append(f"{sc['indent']}T {cvars[nm]} = {init};")
            inner = {"parent": sc, "names": {}, "lines": [], "vars": cvars, "varset": frozenset(names), "indent": sc["indent"] + "    "}
            if counted is None:
                # a data-dependent loop leaves per lane (`if (!(c)) break`): an exchange that depends on its carried values would run
                # with some lanes of the world gone.  (One that does not depend on them is hoisted out of the loop by ref().)
                seen_, stack_ = set(), [cond, *body]
                while stack_:
                    x_ = stack_.pop()
                    if id(x_) in seen_ or x_.op in ("const", "leaf"):
                        continue
                    seen_.add(id(x_))
                    if x_.op in _WAVE_WIDE and (self._deps(x_) & frozenset(names)):
                        raise NotImplementedError("a lane exchange (lane_read) inside a data-dependent while loop depends on the loop's carried "
                                                  "values: lanes that have left the loop would be read; only statically counted loops may exchange")
                    stack_.extend(x_.args)
                c = ref(cond, inner)
                inner["lines"].append(f"{inner['indent']}if (!({c})) break;")
            new = [ref(b, inner) for b in body]
            tmp = []
            for k, v in enumerate(new):
                tmp.append(f"nx{self.n}")
                self.n += 1
                inner["lines"].append(f"{inner['indent']}const T {tmp[-1]} = {v};")
            for nm, t_ in zip(names, tmp):
                inner["lines"].append(f"{inner['indent']}{cvars[nm]} = {t_};")
            if counted is not None:   # static trip count, no early exit: let the compiler overlap the loads of several rows
                # (counted[2], optional: the unroll factor — 1 for a loop that was kept a loop BECAUSE its unrolled form is too large,
                # stablehlo.py _Eval.ROLL)
                sc["lines"].append(f"{sc['indent']}#pragma unroll {int(counted[2]) if len(counted) > 2 else 8}")
                sc["lines"].append(f"{sc['indent']}for (int it_{self.n} = {counted[0]}; it_{self.n} < {counted[1]}; it_{self.n}++) {{")
            else:
                sc["lines"].append(f"{sc['indent']}for (int it_{self.n} = 0; it_{self.n} < {max_iter}; it_{self.n}++) {{")
            self.n += 1
            sc["lines"].extend(inner["lines"])
            sc["lines"].append(f"{sc['indent']}}}")
            return [cvars[nm] for nm in names]

        def ref(e: dsl.Expr, sc=outer) -> str:
            if e.op == "const":
                return _literal(e.value)
            if e.op == "leaf":
                s_ = sc
                while s_ is not None:
                    if e.name in s_["vars"]:
                        return s_["vars"][e.name]
                    s_ = s_["parent"]
                return _leaf_ref(e.name, self.leaves)
            # a node that does not depend on this loop's carried values (or, in a guarded region, is not one of its members)
            # belongs to the enclosing scope
            while sc["parent"] is not None and not ((self._deps(e) & sc["varset"]) or id(e) in sc.get("members", ())):
                sc = sc["parent"]
            s_ = sc
            while s_ is not None:                      # visible in this scope or any enclosing one
                if id(e) in s_["names"]:
                    return s_["names"][id(e)]
                s_ = s_["parent"]
            if e.op == "while_out":
                node = e.args[0]
                key = ("loop", id(node))
                if key not in sc["names"]:
                    sc["names"][key] = emit_loop(node, sc)
                    self._deps(node)
                name = sc["names"][key][e.value]
                sc["names"][id(e)] = name
                self._deps(e)
                return name
            if e.op == "select" and id(e) in guards:
                # GUARDED SELECT (opt-in, _GUARD_SELECTS): `where(c, expensive, cheap)` whose expensive arm nobody else needs —
                # a sensor's noise draw behind its sample-time test, as a script written for JAX spells a cadence.  The arm's
                # nodes are emitted inside `if (any lane of the wave wants them)`; the value is the same select.
                arm, members, group = guards[id(e)]
                cast = lambda x, scope: (f"T({ref(x, scope)})" if x.op != "const" and self._is_wide(x) else ref(x, scope))
                c_ref = ref(e.args[0], sc)
                want = c_ref if arm == 1 else f"!({c_ref})"
                others = [cast(g.args[3 - arm], sc) for g in group]          # the cheap sides, outside
                inner = {"parent": sc, "names": {}, "lines": [], "vars": {}, "varset": frozenset(), "indent": sc["indent"] + "    ",
                         "members": members}
                mine = [cast(g.args[arm], inner) for g in group]             # the expensive sides: their nodes land in `inner`
                names = []
                for g, other in zip(group, others):
                    names.append(f"t{self.n}")
                    self.n += 1
                    sc["names"][id(g)] = names[-1]
                    self._deps(g)
                    sc["lines"].append(f"{sc['indent']}T {names[-1]} = {other};")
                sc["lines"].append(f"{sc['indent']}if (__any({want})) {{")
                sc["lines"].extend(inner["lines"])
                for nm, m in zip(names, mine):
                    sc["lines"].append(f"{inner['indent']}{nm} = ({want}) ? {m} : {nm};")
                sc["lines"].append(f"{sc['indent']}}}")
                return sc["names"][id(e)]
            wide = self._is_wide(e)
            if _RELAXED_EMIT[0] and not wide and e.op == "div" and e.args[0].is_const(1.0):
                # RELAXED ARITHMETIC (dsl.relaxed_arithmetic): the shared reciprocal of a denominator is v_rcp_f64 + two Newton steps
                # (7 instructions against the 11 of an IEEE divide, 1 ulp), and 1 / sqrt(s) the hardware seed + one cubic correction
                # (spatial.hpp rsqrt_pos: 5 against a library sqrt and a divide)
                d = e.args[1]
                if d.op == "sqrt" and not self._is_wide(d.args[0]):
                    rhs = f"m_rsqrt_relaxed({ref(d.args[0], sc)})"
                else:
                    rhs = f"m_rcp_relaxed({f'T({ref(d, sc)})' if d.op != 'const' and self._is_wide(d) else ref(d, sc)})"
                name = f"t{self.n}"
                self.n += 1
                sc["names"][id(e)] = name
                self._deps(e)
                sc["lines"].append(f"{sc['indent']}const T {name} = {rhs};")
                return name
            if id(e) in fused:
                # FUSED MULTIPLY-ADD (fast-math builds, _FUSE_FMA): a product whose only use is this sum never becomes a value of
                # its own.  Decided here, per node, so every copy of the tick body the compiler makes rounds the same way (the
                # compiler's own "fast" contraction fuses whatever lands in one basic block: kernels.hpp).
                k = fused[id(e)]
                arg = lambda x: (f"T({ref(x, sc)})" if x.op != "const" and self._is_wide(x) else ref(x, sc))
                p_, q_, o_ = arg(e.args[k].args[0]), arg(e.args[k].args[1]), arg(e.args[1 - k])
                name = f"t{self.n}"
                self.n += 1
                sc["names"][id(e)] = name
                self._deps(e)
                rhs = (f"m_fma({p_}, {q_}, {o_})" if e.op == "add" else
                       (f"m_fma({p_}, {q_}, -{o_})" if k == 0 else f"m_fma(-{p_}, {q_}, {o_})"))
                sc["lines"].append(f"{sc['indent']}const T {name} = {rhs};")
                return name
            a = []
            for x in e.args:
                if wide:        # double island: literals as doubles, state-typed operands cast in
                    if x.op == "leaf" and x.name == "tick" and self.leaves.get("tick") == "T(tick)":
                        a.append("double(tick)")                      # the integer itself, not its float32 rounding
                    else:
                        a.append(repr(float(x.value)) if x.op == "const" else (ref(x, sc) if self._is_wide(x) else f"double({ref(x, sc)})"))
                else:
                    a.append(f"T({ref(x, sc)})" if x.op != "const" and self._is_wide(x) else ref(x, sc))
            name = f"t{self.n}"
            self.n += 1
            sc["names"][id(e)] = name
            self._deps(e)
            ctype = "bool" if e.op in _BOOL_OPS else ("double" if wide else "T")
            sc["lines"].append(f"{sc['indent']}const {ctype} {name} = {rhs_of(e, a)};")
            return name

        # PROGRAM ORDER.  Emitting on demand, depth-first from the outputs, postpones every value to its first use: in an
        # unrolled Jacobi SVD the V-accumulation is needed only when the decomposition is finished, so all 150 rotations'
        # (c, s) pairs stayed live until then — 443 live values at the peak of a 6-state EKF step, ~1,000 VGPR spills.  So the
        # nodes the outputs need are emitted in the order the user's program created them (`Expr.seq`; arguments always
        # precede their users), which is the order a person would have written the code in: the same step then peaks at a
        # few matrices' worth of registers.  (Nodes inside loop bodies keep their own scopes and are emitted with their loop.)
        need, stack = {}, ([e for _, e in assign] if (_EMIT_ORDER[0] in ("program", "pressure") or _GUARD_SELECTS[0] or _FUSE_FMA[0]) else [])
        while stack:
            x = stack.pop()
            if id(x) in need or x.op in ("const", "leaf"):
                continue
            if x.op == "while":            # emitted through its while_out users; its initial values are ordinary nodes
                stack.extend(x.args)
                continue
            need[id(x)] = x
            stack.extend(x.args)
        if _GUARD_SELECTS[0]:
            guards.update(_plan_guards(need, [e for _, e in assign], set(self.names)))
        guarded = set().union(*[g[1] for g in guards.values()]) if guards else set()
        folded = set()
        if _FUSE_FMA[0]:
            n_users: Dict[int, int] = {}
            for n_ in need.values():
                for a_ in n_.args:
                    n_users[id(a_)] = n_users.get(id(a_), 0) + 1
            root_ids = {id(e) for _, e in assign}
            for n_ in sorted(need.values(), key=lambda v: v.seq):
                if n_.op in ("add", "sub") and not self._is_wide(n_):
                    for k in (1, 0):
                        m = n_.args[k]
                        if (m.op == "mul" and n_users.get(id(m)) == 1 and id(m) in need and id(m) not in root_ids and id(m) not in folded
                                and id(m) not in self.names and not self._is_wide(m)):
                            fused[id(n_)] = k
                            folded.add(id(m))
                            break
        if _ESTIMATE[0] is not None and _EMIT_ORDER[0] != "pressure" and need:
            _pressure_order(need, [e for _, e in assign])          # for its estimates only
        if _EMIT_ORDER[0] == "pressure":
            ordered = _pressure_order(need, [e for _, e in assign])
        else:
            ordered = sorted(need.values(), key=lambda n_: n_.seq) if _EMIT_ORDER[0] == "program" else ()
        for x in ordered:
            if id(x) not in guarded and id(x) not in folded:
                ref(x)
        outs = []
        for lv, e in assign:
            o = f"o{self.n}"
            self.n += 1
            lines.append(f"{indent}const T {o} = {('T(' + ref(e) + ')') if e.op not in ('const', 'leaf') and self._is_wide(e) else ref(e)};")
            outs.append((lv, o))
        for lv, o in outs:
            if lv.startswith("@w_act@"):
                lines.append(f"{indent}if (w_act) {lv[7:]} = {o};")
            else:
                lines.append(f"{indent}{lv} = {o};")
        wr = set(written)
        for k in list(self.names):
            dk = self.deps.get(k[1] if isinstance(k, tuple) else k, frozenset())
            if (scoped and k not in before) or (wr and dk & wr):
                del self.names[k]
        return lines


def _pressure_order(need: Dict[int, "dsl.Expr"], roots: Sequence["dsl.Expr"]) -> List["dsl.Expr"]:
    """REGISTER-PRESSURE ORDER.  Greedy list scheduling of a block's nodes: among the nodes whose arguments exist, the one that
    ends the most live ranges comes next (the last user of an argument frees its register); ties go to the node the DEMAND order
    (depth-first from the outputs) would reach first, so nothing that ends no range is started before something needs it.
    What it is for: a whole-world tick (elodin_amd/stablehlo.py, one lane = one world) is four RK4 stages whose stage vectors
    are combined by left-associated sums written after the last stage.  In creation order every stage velocity is live from its
    stage to the end (peak 200 live values for the three-body world, 225 spilled registers); on demand the velocity sum forms
    stage by stage but all four stage accelerations wait for theirs (135 / 10 spills); here each partial sum is formed the moment
    its operands exist (115)."""
    def deps(x):
        src = x.args[0].args if x.op == "while_out" else x.args
        return [a for a in src if id(a) in need]
    users: Dict[int, int] = {}
    consumers: Dict[int, list] = {}
    pending: Dict[int, int] = {}
    for i, x in need.items():
        d = {id(a) for a in deps(x)}
        pending[i] = len(d)
        for a in d:
            users[a] = users.get(a, 0) + 1
            consumers.setdefault(a, []).append(i)
    for r in roots:
        if id(r) in need:
            users[id(r)] = users.get(id(r), 0) + 1                      # a block output stays live to the end
    # demand positions (iterative depth-first post-order from the outputs)
    pos, seen = {}, set()
    for r in roots:
        stack = [(r, 0)]
        while stack:
            n_, k = stack.pop()
            if id(n_) in seen or id(n_) not in need:
                continue
            d = deps(n_)
            if k < len(d):
                stack.append((n_, k + 1))
                stack.append((d[k], 0))
            else:
                seen.add(id(n_))
                pos[id(n_)] = len(pos)
    for i in need:
        pos.setdefault(i, len(pos))
    if _ESTIMATE[0] is not None:       # build(): how many values each emission order keeps live at its worst in this block
        by_seq = sorted(need.values(), key=lambda n_: n_.seq)
        by_pos = sorted(need.values(), key=lambda n_: pos[id(n_)])
        for name_, order_ in (("program", by_seq), ("demand", by_pos)):
            _ESTIMATE[0][name_] = max(_ESTIMATE[0].get(name_, 0), _peak_live(order_, deps, dict(users)))

    def score(i):
        return sum(1 for a in {id(a) for a in deps(need[i])} if users[a] == 1)
    ready = {i for i, p_ in pending.items() if p_ == 0}
    order: List["dsl.Expr"] = []
    while ready:
        best = max(ready, key=lambda i: (score(i), -pos[i]))
        ready.discard(best)
        x = need[best]
        order.append(x)
        for a in {id(a) for a in deps(x)}:
            users[a] -= 1
        for c in consumers.get(best, ()):
            pending[c] -= 1
            if pending[c] == 0:
                ready.add(c)
    if _ESTIMATE[0] is not None:
        users2: Dict[int, int] = {}
        for i, x in need.items():
            for a in {id(a) for a in deps(x)}:
                users2[a] = users2.get(a, 0) + 1
        for r in roots:
            if id(r) in need:
                users2[id(r)] = users2.get(id(r), 0) + 1
        _ESTIMATE[0]["pressure"] = max(_ESTIMATE[0].get("pressure", 0), _peak_live(order, deps, users2))
    return order


_ESTIMATE: List[Optional[Dict[str, int]]] = [None]
_EXPECT_SCRATCH = [False]      # the variant being compiled keeps state in private memory by design ("memory")


def _peak_live(order, deps, users: Dict[int, int]) -> int:
    """Largest number of simultaneously live node values when a block's nodes are emitted in `order` (users: remaining-use counts,
    consumed here)."""
    live = peak = 0
    for x in order:
        live += 1
        peak = max(peak, live)
        for a in {id(a) for a in deps(x)}:
            users[a] -= 1
            if users[a] == 0:
                live -= 1
    return peak


# Opt-in (codegen.generate_source(..., guard_selects=True) / SIXDOF_GUARD_SELECTS=1): see _Emitter.block, "GUARDED SELECT".
_GUARD_SELECTS = [False]
# Fast-math builds fold single-use products into the sums that consume them (see _Emitter.block, "FUSED MULTIPLY-ADD");
# SIXDOF_FUSE_FMA=0 keeps them apart (A/B).  Exact builds never fuse: a reference evaluates every node to a rounded value.
_FUSE_FMA = [False]
# programs traced under dsl.relaxed_arithmetic (TracedProgram.fp_contract): see _Emitter.block `m_rcp_relaxed`
_RELAXED_EMIT = [False]
_GUARD_MIN_COST = 40
_NODE_COST = {"lane_read": 8, "lane_read_dyn": 10, "threefry": 90, "erfinv": 120, "sin": 12, "cos": 12, "tan": 20, "exp": 10, "log": 10, "pow": 25, "atan2": 27, "asin": 20,
              "acos": 20, "hypot": 12, "div": 4, "sqrt": 4, "interp": 30, "cbrt": 20, "sinh": 20, "cosh": 20, "erfc": 40, "log1p": 15,
              "expm1": 15, "mod": 8}


# Never inside a guarded arm.  Loops and window loads own scopes of their own.  lane_read / lane_read_dyn are wave-wide exchanges
# (__shfl -> ds_bpermute): EVERY lane of the world must execute them, or the lanes that do read the registers of lanes whose EXEC
# bit is off — zero or stale values, i.e. silently wrong joins and edge folds.  They stay outside: a would-be member
# that is an exchange is simply not a member (it and everything it needs is emitted before the branch, unconditionally).
_NEVER_GUARDED = ("while", "while_out", "wload")
_WAVE_WIDE = ("lane_read", "lane_read_dyn")


def _plan_guards(need: Dict[int, "dsl.Expr"], roots: Sequence["dsl.Expr"], available: set) -> Dict[int, tuple]:
    """Which selects of a block get a guarded arm: id(select) -> (arm index, member ids, [the selects of its group in creation order]).

    Selects with the SAME condition and the same guarded side form one group (`imu = where(due, fresh, held)` per axis: the three
    draws share their key derivation) and get one branch.  A group's MEMBERS are the nodes every use of which lies inside the
    group's arms (found in reverse creation order: a node joins once all its users have); the group is guarded when their summed
    cost is worth a branch.  Never members: nodes an earlier block already emitted (`available`), block outputs, loops, window
    loads.  A select whose condition or other arm needs a value computed inside the group stays a plain select."""
    users: Dict[int, list] = {}
    for n in need.values():
        if n.op in ("while_out", "while"):      # loop bodies reach nodes this walk does not see: no guards in such a block
            return {}
        for a in n.args:
            users.setdefault(id(a), []).append(n)
    root_ids = {id(r) for r in roots}
    by_seq = sorted(need.values(), key=lambda n_: -n_.seq)
    groups: Dict[tuple, list] = {}
    for s_ in need.values():
        if s_.op != "select" or s_.args[0].op == "const":      # a constant condition: the compiler folds it
            continue
        for arm in (1, 2):
            a = s_.args[arm]
            if a.op in ("const", "leaf") or id(a) not in need or id(a) in available or id(a) in root_ids or a is s_.args[0] \
                    or a is s_.args[3 - arm] or any(u is not s_ for u in users.get(id(a), [])):
                continue
            groups.setdefault((id(s_.args[0]), arm), []).append(s_)

    def reaches(start, blocked):
        seen, stack = set(), [start]
        while stack:
            x = stack.pop()
            if id(x) in blocked:
                return True
            if id(x) in seen or x.op in ("const", "leaf"):
                continue
            seen.add(id(x))
            stack.extend(x.args)
        return False

    def plan(sels, arm):
        """(cost, arm, members, sels) of one candidate region, or None when it cannot be one."""
        sel_ids = {id(s_) for s_ in sels}
        members = {id(s_.args[arm]) for s_ in sels}
        cost = sum(_NODE_COST.get(s_.args[arm].op, 1) for s_ in sels)
        top = max(s_.args[arm].seq for s_ in sels)
        if any(s_.args[arm].op in _NEVER_GUARDED + _WAVE_WIDE for s_ in sels):
            return None
        for x in by_seq:
            if x.seq >= top or id(x) in members or id(x) in available or id(x) in root_ids or id(x) in sel_ids:
                continue
            us = users.get(id(x), [])
            if us and all(id(u) in members for u in us):
                if x.op in _NEVER_GUARDED:
                    return None
                if x.op in _WAVE_WIDE:
                    continue        # computed outside the branch by every lane (so is everything only it needs: its users are not all members)
                members.add(id(x))
                cost += _NODE_COST.get(x.op, 1)
        if cost < _GUARD_MIN_COST:
            return None
        # nothing the region needs from outside (conditions, cheap sides) may be computed inside it, and no arm may need the
        # value of another select of the region
        blocked = members | sel_ids
        for s_ in sels:
            if reaches(s_.args[0], blocked) or reaches(s_.args[3 - arm], blocked) or reaches(s_.args[arm], sel_ids - {id(s_)}):
                return None
        return (cost, arm, members, sels)

    plans = []
    for (cond_id, arm), sels in groups.items():
        sels = sorted(sels, key=lambda n_: n_.seq)
        p_ = plan(sels, arm)
        if p_ is not None:
            plans.append(p_)
        elif len(sels) > 1:
            plans.extend(q for q in (plan([s_], arm) for s_ in sels) if q is not None)
    out: Dict[int, tuple] = {}
    taken: set = set()
    for cost, arm, members, sels in sorted(plans, key=lambda p_: -p_[0]):      # costliest first; a node belongs to one region
        if members & taken or any(id(s_) in taken for s_ in sels):
            continue
        for s_ in sels:
            out[id(s_)] = (arm, members, sels)
        taken |= members | {id(s_) for s_ in sels}
    return out


def emit_block(assign, leaves: Dict[str, str], indent: str = "        ") -> List[str]:
    return _Emitter(leaves).block(assign, indent)


def emit_apply(tp: dsl.TracedPipe) -> List[str]:
    names = ["F.tau_w.x", "F.tau_w.y", "F.tau_w.z", "F.f.x", "F.f.y", "F.f.z", "F.tau_b.x", "F.tau_b.y", "F.tau_b.z"]
    return emit_block(list(zip(names, tp.outputs)), _APPLY_LEAVES)


# Fallback layout of a program whose state does not fit a wave's 512 registers (the f64 Falcon 9 closed loop: ~250 doubles
# of component state + the temporaries of 20 systems): the register image `Regs` is declared `volatile`, i.e. it lives in
# the lane's private (scratch) memory and every access is a real load / store — deliberate, compiler-independent placement
# instead of register-allocator spills (a spilling build miscomputed on gfx950 before, see _compile).  The launch-level
# column load / store and the cold-column scheme stay as they are.  Costs scratch traffic (L1 / L2 resident) on every
# access: a parity build, not a fast one.  build() switches to it by itself when no register-resident build is spill-free.
# (Tried first: keeping every column in HBM and wrapping each system in loads / stores — the optimiser forwards the stored
# values to the next system's loads and the live set stays where it was: 150+ spills with every flag set.)
_MEMORY_COLUMNS = [False]
# Order in which a block's nodes are emitted: "program" = the order the user's code created them (see _Emitter.block),
# "demand" = depth-first from the outputs, each value right before its first use.  Program order is tried first; a
# machine-generated DAG with no meaningful creation order (the fuzzer's random programs) can do better on demand.
_EMIT_ORDER = ["program"]
# Last resort: the tick body compiled as a real function (step_kernel.hpp SIXDOF_TICK_OUT_OF_LINE) — for programs whose
# TEMPORARIES, not state, overflow the register file (a fuzz program of 200 inlined f64 libm calls: 43 values live in source
# order, 650 registers after LLVM's allocation across its 600 basic blocks, whatever the flags).
_TICK_OUT_OF_LINE = [False]


def _col_slots(names) -> set:
    return {int(n[1:].split("_")[0]) for n in names if n[0] == "c" and "_" in n and n[1:].split("_")[0].isdigit()}


def _cold_slots(tp, pipe_tp, pre, post, reg_cols) -> Dict[int, int]:
    """Program columns that only CADENCED systems (`every > 1`) touch: {slot: width}.  They need not occupy registers
    across the ticks in between — a guidance computer's navigator state, say, is read and written on every 10th tick only —
    so the cadence block loads them from their HBM column on entry and stores what it wrote on exit (same lane, program
    order: later blocks see the stores), and the launch-level load / store skips them.  On the Falcon 9 program that takes
    31 values (62 f64 registers) out of the tick loop."""
    if os.environ.get("SIXDOF_NO_COLD_COLUMNS", "") == "1":
        return {}
    hot = _col_slots(dsl._leaves_of(list(pipe_tp.outputs))) if pipe_tp is not None else set()
    touched = set()
    for s_ in pre + post:
        slots = _col_slots(dsl._leaves_of([e for _, e in s_.assign])) | _col_slots(s_.written)
        touched |= slots
        if s_.every <= 1:
            hot |= slots
    return {k: w for k, w in reg_cols if k in touched and k not in hot}


def _transient_slots(pipe_tp, pre, post, reg_cols, cold) -> Dict[int, int]:
    """Program columns that every tick OVERWRITES before anything reads them (a wrench recomputed per tick, q-bar, the
    geodetic altitude ...): {slot: width}.  Their value never crosses a tick boundary, so they need neither the launch-level
    load nor a register across the loop's back edge; they are stored from inside the tick body on the LAST tick of the
    launch.  (Kept in registers and stored after the loop, each would be a loop-carried value — initial or last written — and
    occupy its register for the whole iteration.)  40 values on the Falcon 9 program."""
    if os.environ.get("SIXDOF_NO_TRANSIENT_COLUMNS", "") == "1":
        return {}
    widths = dict(reg_cols)
    seen, out = set(), {}
    stages = [(s_.every, _col_slots(dsl._leaves_of([e for _, e in s_.assign])), s_.written) for s_ in pre]
    if pipe_tp is not None:
        stages.append((1, _col_slots(dsl._leaves_of(list(pipe_tp.outputs))), ()))
    stages += [(s_.every, _col_slots(dsl._leaves_of([e for _, e in s_.assign])), s_.written) for s_ in post]
    for every, reads, written in stages:
        seen |= reads & set(widths)
        by_slot: Dict[int, set] = {}
        for t in written:
            if t[0] == "c" and "_" in t and t[1:].split("_")[0].isdigit():
                by_slot.setdefault(int(t[1:].split("_")[0]), set()).add(t)
        for k, elems in by_slot.items():
            if k in widths and k not in seen and k not in cold and every <= 1 and len(elems) == widths[k]:
                out[k] = widths[k]
            seen.add(k)
    return out


def _store_only_slots(pipe_tp, pre, post, transient) -> Dict[int, int]:
    """Transient columns NOTHING in the program reads (telemetry a script derives for the database: a geodetic altitude, an
    inlet pressure): {slot: width}.  Their value is only ever looked at through the column — after the launch, or by the
    history ring — so they are evaluated on the ticks that store them (the last tick of a launch, every tick while the column
    is being recorded) instead of on all of them.  SIXDOF_NO_STORE_ONLY_COLUMNS=1 switches the analysis off."""
    if os.environ.get("SIXDOF_NO_STORE_ONLY_COLUMNS", "") == "1":
        return {}
    read = _col_slots(dsl._leaves_of(list(pipe_tp.outputs))) if pipe_tp is not None else set()
    for s_ in pre + post:
        read |= _col_slots(dsl._leaves_of([e for _, e in s_.assign]))
    return {k: w for k, w in transient.items() if k not in read}


def _emit_systems(systems, cold: Optional[Dict[int, int]] = None, store_only: Optional[Dict[int, int]] = None) -> str:
    out = []
    em = _Emitter(_SYSTEM_LEAVES)
    cold = cold or {}
    store_only = store_only or {}
    cadence = lambda s: f"tick % {s.every}ull == {s.phase}ull" + (f" || tick == {s.also_at}ull" if s.also_at is not None else "")
    reads_of = lambda s: sorted((_col_slots(dsl._leaves_of([e for _, e in s.assign])) | _col_slots([t for t, _ in s.assign])) & set(cold))
    # Cold columns are loaded AHEAD of the blocks that use them: one batch per cadence at the top of the function, so the
    # round trip (a lone wave per SIMD has nothing to hide it behind) overlaps the every-tick systems in between, and blocks
    # that share a cadence share one trip.  Every block still stores what it wrote on exit; a later block of the same tick
    # reads the registers, not memory, so it sees those writes.
    ahead: Dict[str, list] = {}
    for s in systems:
        if s.every > 1 and reads_of(s):
            ahead.setdefault(cadence(s), [[], set()])
            ahead[cadence(s)][0].append(s.name)
            ahead[cadence(s)][1].update(reads_of(s))
    for cond, (names, slots) in ahead.items():
        ld = "".join(f"            if (c_act) {{ {_col_ptr(k, cold[k], 'c_row')} "
                     + " ".join(f"r.c{k}[{j}] = g[{_col_idx(j)}];" for j in range(cold[k])) + " }\n" for k in sorted(slots))
        out.append(f"        if ({cond}) {{  // cold columns of {', '.join(names)}\n{ld}        }}")
    for s in systems:
        assign = [(_leaf_ref(t, _SYSTEM_LEAVES), e) for t, e in s.assign]
        written = [t for t, _ in s.assign]
        if s.every > 1:     # wave-uniform cadence branch: its temporaries stay inside
            body = "\n".join(em.block(assign, "            ", written, scoped=True))
            w_slots = sorted(_col_slots(written) & set(cold))
            st = "".join(f"\n            if (c_act) {{ {_col_ptr(k, cold[k], 'c_row', False)} "
                         + " ".join(f"g[{_col_idx(j)}] = r.c{k}[{j}];" for j in range(cold[k])) + " }" for k in w_slots)
            out.append(f"        if ({cadence(s)}) {{  // {s.name}\n{body}{st}\n        }}")
        else:
            late = [(t, e) for t, e in s.assign if _col_slots([t]) & set(store_only)]
            lazy = ""
            if late:
                # store-only columns (_store_only_slots): in front of the rest of the block (they read what the block has not
                # written yet), inside the condition under which somebody will look at the column
                # (P.hist_ring: the history ring is on for this launch — one kernarg word the kernel holds anyway; testing the
                # columns' own P.model_hist pointers costs a scalar load and its wait per block and tick)
                lazy = (f"        if (tick == P.tick0 + P.n_ticks || P.hist_ring != 0u) {{"
                        f"  // store-only columns of {s.name}\n"
                        + "\n".join(em.block([(_leaf_ref(t, _SYSTEM_LEAVES), e) for t, e in late], "            ", [t for t, _ in late], scoped=True))
                        + "\n        }\n")
                assign = [(_leaf_ref(t, _SYSTEM_LEAVES), e) for t, e in s.assign if not (_col_slots([t]) & set(store_only))]
                written = [t for t, _ in s.assign if not (_col_slots([t]) & set(store_only))]
            body = lazy + "\n".join(em.block(assign, "        ", written))
            w_slots = sorted(_col_slots(written) & set(cold))
            r_slots = sorted((_col_slots(dsl._leaves_of([e for _, e in s.assign])) | set(w_slots)) & set(cold))
            ld = "".join(f"        if (c_act) {{ {_col_ptr(k, cold[k], 'c_row')} "
                         + " ".join(f"r.c{k}[{j}] = g[{_col_idx(j)}];" for j in range(cold[k])) + " }\n" for k in r_slots)
            st = "".join(f"\n        if (c_act) {{ {_col_ptr(k, cold[k], 'c_row', False)} "
                         + " ".join(f"g[{_col_idx(j)}] = r.c{k}[{j}];" for j in range(cold[k])) + " }" for k in w_slots)
            out.append(f"        // {s.name}\n{ld}{body}{st}")
    return "\n".join(out)


_RELAXED_PRELUDE = '''// relaxed arithmetic only (dsl.relaxed_arithmetic): 1 / x as v_rcp_f64 + two Newton steps (1 ulp; x = 0 / inf keep the seed's
// inf / 0), 1 / sqrt(x) as the seed + one cubic correction
__device__ __forceinline__ double m_rcp_relaxed(double x) {
    const double r0 = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r0, 1.0);
    double r = fma(e, r0, r0);
    r = fma(fma(-x, r, 1.0), r, r);
    return e == e ? r : r0;
}
__device__ __forceinline__ float m_rcp_relaxed(float x) { return 1.0f / x; }
__device__ __forceinline__ double m_rsqrt_relaxed(double x) { return rsqrt_pos(x); }
__device__ __forceinline__ float m_rsqrt_relaxed(float x) { return 1.0f / fast_sqrt(x); }
'''

_PRELUDE = '''// SIXDOF_FAST_MATH (f32 programs, opt-in): hardware transcendentals (v_sin / v_cos / v_exp / v_log / v_rcp /
// v_sqrt, ~1e-6 relative) instead of the correctly rounded library calls — sinf+cosf alone are ~240 instructions, and a
// tick of the Falcon 9 program makes 32 of them.  f64 programs and the default f32 mode keep the library functions.
template <class T> __device__ __forceinline__ T m_div(T a, T b) { return a / b; }
__device__ __forceinline__ float m_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double m_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
#ifdef SIXDOF_FAST_MATH
__device__ __forceinline__ float m_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }   // v_rcp_f32, 1 ulp
__device__ __forceinline__ float m_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }               // v_sqrt_f32, 1 ulp
__device__ __forceinline__ double m_sqrt(double x) { return fast_sqrt(x); }
#else
template <class T> __device__ __forceinline__ T m_sqrt(T x) { return fast_sqrt(x); }
#endif
__device__ __forceinline__ double m_abs(double x) { return fabs(x); }
__device__ __forceinline__ float m_abs(float x) { return fabsf(x); }
#define SIXDOF_M1(name, fd, ff) \\
    __device__ __forceinline__ double name(double x) { return fd(x); } \\
    __device__ __forceinline__ float name(float x) { return ff(x); }
#ifdef SIXDOF_FAST_MATH
__device__ __forceinline__ float m_fast_tan(float x) { return __sinf(x) * __builtin_amdgcn_rcpf(__cosf(x)); }
SIXDOF_M1(m_sin, sin, __sinf) SIXDOF_M1(m_cos, cos, __cosf) SIXDOF_M1(m_tan, tan, m_fast_tan) SIXDOF_M1(m_exp, exp, __expf)
SIXDOF_M1(m_log, log, __logf)
#else
SIXDOF_M1(m_sin, sin, sinf) SIXDOF_M1(m_cos, cos, cosf) SIXDOF_M1(m_tan, tan, tanf) SIXDOF_M1(m_exp, exp, expf)
SIXDOF_M1(m_log, log, logf)
#endif
SIXDOF_M1(m_acos, acos, acosf) SIXDOF_M1(m_asin, asin, asinf)
SIXDOF_M1(m_log1p, log1p, log1pf) SIXDOF_M1(m_expm1, expm1, expm1f) SIXDOF_M1(m_cbrt, cbrt, cbrtf) SIXDOF_M1(m_floor, floor, floorf)
SIXDOF_M1(m_ceil, ceil, ceilf) SIXDOF_M1(m_trunc, trunc, truncf) SIXDOF_M1(m_rint, rint, rintf) SIXDOF_M1(m_sinh, sinh, sinhf)
SIXDOF_M1(m_cosh, cosh, coshf) SIXDOF_M1(m_erfc, erfc, erfcf)
SIXDOF_M1(m_erfinv, erfinv, erfinvf)
template <class T> __device__ __forceinline__ bool m_isfinite(T x) { return isfinite(x); }
// threefry2x32 (the generator behind jax.random): key and counter words are uint32 values held exactly in doubles
__device__ __forceinline__ double m_threefry(double k0d, double k1d, double c0d, double c1d, int which) {
    const uint32_t k0 = static_cast<uint32_t>(k0d), k1 = static_cast<uint32_t>(k1d);
    uint32_t x0 = static_cast<uint32_t>(c0d), x1 = static_cast<uint32_t>(c1d);
    const uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
    const int rot[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
    x0 += ks[0];
    x1 += ks[1];
#pragma unroll
    for (int r = 0; r < 5; r++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            x0 += x1;
            x1 = (x1 << rot[r & 1][j]) | (x1 >> (32 - rot[r & 1][j]));
            x1 ^= x0;
        }
        x0 += ks[(r + 1) % 3];
        x1 += ks[(r + 2) % 3] + static_cast<uint32_t>(r + 1);
    }
    return static_cast<double>(which ? x1 : x0);
}
__device__ __forceinline__ float m_threefry(float, float, float, float, int) { return 0.0f; }   // rejected at generation
template <class T> __device__ __forceinline__ T m_mod(T x, T y) { return x - m_floor(x / y) * y; }   // jnp.remainder: sign of y
#define SIXDOF_M2(name, fd, ff) \\
    __device__ __forceinline__ double name(double x, double y) { return fd(x, y); } \\
    __device__ __forceinline__ float name(float x, float y) { return ff(x, y); }
SIXDOF_M2(m_max, fmax, fmaxf) SIXDOF_M2(m_min, fmin, fminf)
#ifdef SIXDOF_FAST_MATH
// atan2 without the library call (45 instructions, and ECEF->geodetic makes nine of them): octant reduction to
// [0, 1], one more reduction about tan(pi/8), degree-7 odd minimax polynomial (the classic single-precision atan
// kernel), quadrant fix-ups.  ~1e-7 absolute.
__device__ __forceinline__ float m_fast_atan2(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float t = mx > 0.0f ? mn * __builtin_amdgcn_rcpf(mx) : 0.0f;
    const bool hi = t > 0.4142135623730950f;
    const float u = (hi ? t - 1.0f : t) * __builtin_amdgcn_rcpf(hi ? t + 1.0f : 1.0f);   // branch-free second reduction
    const float z = u * u;
    const float p = ((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f;
    float r = fmaf(p * z, u, u) + (hi ? 0.78539816339744831f : 0.0f);
    r = ay > ax ? 1.57079632679489662f - r : r;
    r = x < 0.0f ? 3.14159265358979323f - r : r;
    return copysignf(r, y);
}
SIXDOF_M2(m_atan2, atan2, m_fast_atan2)
#else
SIXDOF_M2(m_atan2, atan2, atan2f)
#endif
#ifdef SIXDOF_FAST_MATH
__device__ __forceinline__ float m_fast_pow(float x, float y) { return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); }   // x > 0 (x = 0 -> 0 for y > 0)
__device__ __forceinline__ float m_fast_hypot(float x, float y) { return __builtin_amdgcn_sqrtf(x * x + y * y); }
SIXDOF_M2(m_pow, pow, m_fast_pow) SIXDOF_M2(m_hypot, hypot, m_fast_hypot)
#else
SIXDOF_M2(m_pow, pow, powf) SIXDOF_M2(m_hypot, hypot, hypotf)
#endif
// table[idx, col] for a constant table in device memory (dsl.gather): a negative index counts from the end, then the index
// is clamped into [0, n) — jax's gather; a NaN index converts to 0.
template <class T>
__device__ __forceinline__ T m_gather(const double* __restrict__ tab, T idx, int n, int stride, int col) {
    int i = static_cast<int>(idx);
    i = i < 0 ? i + n : i;
    i = i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
    return T(tab[(size_t)i * stride + col]);
}
// jnp.interp over a constant table: i = clip(searchsorted(xp, x, 'right'), 1, N-1); fp[i-1] + (x-xp[i-1])/dx * df,
// clamped to the end values outside the table.
template <class T, int N>
__device__ __forceinline__ T m_interp(T x, const double (&xp)[N], const double (&fp)[N]) {
    if constexpr (N == 1) return T(fp[0]);
    T x0, f0, x1, f1;
    if constexpr (N <= 32) {
        // short tables: the bracketing breakpoints by a chain of selects over compile-time constants (tables sharing their
        // breakpoints and their x share the compares) — no per-lane table load, which a lone wave has nothing to hide behind.
        // Same interval as the count below for an ascending xp: the last k in [1, N-2] with xp[k] <= x, else 0.
        x0 = T(xp[0]); f0 = T(fp[0]); x1 = T(xp[1]); f1 = T(fp[1]);
#pragma unroll
        for (int k = 1; k < N - 1; k++) {
            const bool at = T(xp[k]) <= x;
            x0 = at ? T(xp[k]) : x0; f0 = at ? T(fp[k]) : f0;
            x1 = at ? T(xp[k + 1]) : x1; f1 = at ? T(fp[k + 1]) : f1;
        }
    } else {   // long tables: bisect (searchsorted side='right')
        int lo = 0, hi = N;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (x < T(xp[mid])) hi = mid; else lo = mid + 1;
        }
        const int i = lo < 1 ? 1 : (lo > N - 1 ? N - 1 : lo);
        x0 = T(xp[i - 1]); f0 = T(fp[i - 1]); x1 = T(xp[i]); f1 = T(fp[i]);
    }
    const T dx = x1 - x0, df = f1 - f0;
    T f = dx == T(0) ? f0 : f0 + m_div(x - x0, dx) * df;
    f = x < T(xp[0]) ? T(fp[0]) : f;
    return x > T(xp[N - 1]) ? T(fp[N - 1]) : f;
}
// Same result for an evenly spaced table: the interval comes from one multiply (then is corrected against the stored
// breakpoints, so it is exactly searchsorted's), instead of log2(N) dependent table loads.
template <class T, int N>
__device__ __forceinline__ T m_interp_uniform(T x, const double (&xp)[N], const double (&fp)[N], T inv_step) {
    T k = (x - T(xp[0])) * inv_step;
    k = k > T(0) ? (k < T(N) ? k : T(N)) : T(0);                      // also absorbs NaN and values far off the table
    int c = static_cast<int>(k) + 1;                                  // ~ number of breakpoints <= x
    c = c < 1 ? 1 : (c > N - 1 ? N - 1 : c);
    // the correction moves c by one at most, so the four breakpoints around it are all it can ask for: ONE round trip to
    // the table (8 independent loads) instead of three dependent ones.  (Indices clamped; a clamped entry is never selected.)
    T xs[4], fs[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int i = c - 2 + j;
        const int ic = i < 0 ? 0 : (i > N - 1 ? N - 1 : i);
        xs[j] = T(xp[ic]); fs[j] = T(fp[ic]);
    }
    const bool down = c > 1 && x < xs[1];                             // c -> c - 1
    const bool up = !down && c < N - 1 && xs[2] <= x;                 // c -> c + 1 (xp[c] <= x cannot hold after a step down)
    const T x0 = down ? xs[0] : (up ? xs[2] : xs[1]), f0 = down ? fs[0] : (up ? fs[2] : fs[1]);
    const T x1 = down ? xs[1] : (up ? xs[3] : xs[2]), f1 = down ? fs[1] : (up ? fs[3] : fs[2]);
    const T dx = x1 - x0, df = f1 - f0;
    T f = dx == T(0) ? f0 : f0 + m_div(x - x0, dx) * df;
    f = x < T(xp[0]) ? T(fp[0]) : f;
    return x > T(xp[N - 1]) ? T(fp[N - 1]) : f;
}
'''


def _emit_tables() -> str:
    out = []
    for key, stem in _GATHERS.items():      # row-major [rows, cols], doubles whatever the program's dtype (read through m_gather)
        t = dsl._GATHER_TABLES[key]
        out.append(f"__device__ const double {stem}[{t.size}] = {{{', '.join(repr(float(v)) for v in t.reshape(-1))}}};")
    for (stride, table), stem in _LANE_TABLES.items():
        out.append(f"__device__ const unsigned char {stem}[{len(table)}] = {{{', '.join(str(j) for j in table)}}};")
    for (xs, fs), stem in _TABLES.items():
        out.append(f"__device__ const double {stem}_x[{len(xs)}] = {{{', '.join(repr(v) for v in xs)}}};")
        out.append(f"__device__ const double {stem}_f[{len(fs)}] = {{{', '.join(repr(v) for v in fs)}}};")
    return "\n".join(out)


def _uses_op(tp, op: str) -> bool:
    seen = set()

    def walk(e):
        if id(e) in seen:
            return False
        seen.add(id(e))
        if e.op == op:
            return True
        sub = list(e.args)
        if e.op == "while":
            sub += [e.value[1], *e.value[2]]
        return any(walk(a) for a in sub)
    roots = []
    if isinstance(tp, dsl.TracedProgram):
        for s_ in tp.pre + tp.post:
            roots += [e for _, e in s_.assign]
        roots += tp.pipe.outputs
    else:
        roots += tp.outputs
    return any(walk(r) for r in roots)


def _slots_of(systems, extra_exprs=()) -> set:
    """Program column slots the given traced systems (and expressions) read or write."""
    names = set()
    for s_ in systems:
        names |= dsl._leaves_of([e for _, e in s_.assign])
        names |= set(s_.written)
    names |= dsl._leaves_of(list(extra_exprs))
    return {int(n[1:].split("_")[0]) for n in names if n[0] == "c" and "_" in n and n[1:].split("_")[0].isdigit()}


def _emit_pipe_struct(name: str, tp, pipe_tp, pre, post, used: Optional[set], pre_reads_accel: bool, n_aux: int, body_dead: bool = False) -> str:
    """One PIPE struct of csrc/step_kernel.hpp: effector stage from `pipe_tp` (None: no effectors), `pre` / `post` hooks from
    traced systems.  `used`: the program column slots this struct keeps in registers (None: all of tp.columns)."""
    body = "\n".join(emit_apply(pipe_tp)) if pipe_tp is not None else ""
    apply_loads = ""
    model = ""
    is_prog = tp is not None and isinstance(tp, dsl.TracedProgram)
    win_setup = ""
    if is_prog:
        cols = tp.columns
        reg_cols = [(k, w) for k, (_, w) in enumerate(cols) if k not in _WINDOWS and (used is None or k in used)]     # windows stay in HBM
        written = sorted(_slots_of(pre + post) & {int(t[1:].split("_")[0]) for s_ in pre + post for t in s_.written if t[0] == "c"}) \
            if used is not None else list(tp.written_slots)
        cold = _cold_slots(tp, pipe_tp, pre, post, reg_cols)
        transient = _transient_slots(pipe_tp, pre, post, reg_cols, cold) if used is None else {}
        store_only = _store_only_slots(pipe_tp, pre, post, transient)
        vol = "volatile " if _MEMORY_COLUMNS[0] else ""
        regs = "\n".join(f"        {vol}T c{k}[{w}];" + ("   // cold: lives in its HBM column between cadence blocks" if k in cold else "") for k, w in reg_cols)
        # wave-uniform columns (TracedProgram.uniform_slots): every lane reads the first row of the wavefront's block — one line per
        # wave instead of one value per row; stores stay per row, so every row keeps holding the value
        uni = set(getattr(tp, "uniform_slots", ()) or ()) if used is None else set()       # (either device layout: the row index alone changes)
        loads = "\n".join(
            f"            {{ {_col_ptr(k, w, '(row & ~uint32_t(kWave - 1))' if k in uni else 'row')} "
            + " ".join(f"r.c{k}[{j}] = col_ld<POL>(g + {_col_idx(j)});" for j in range(w)) + " }" for k, w in reg_cols if k not in cold and k not in transient)
        # element by element, not a loop: a loop the optimiser does not unroll (-O1, the low-register-pressure fallback build)
        # indexes the array dynamically, which pins the whole register file image in scratch memory
        zero = " ".join(" ".join(f"r.c{k}[{j}] = T(0);" for j in range(w)) for k, w in reg_cols)
        stores = "\n".join(
            f"        {{ {_col_ptr(k, cols[k][1], 'row', False)} "
            + " ".join(f"col_st<POL>(g + {_col_idx(j)}, r.c{k}[{j}]);" for j in range(cols[k][1])) + " }" for k in written if k not in cold and k not in transient)
        transient_stores = ""
        if transient:
            transient_stores = ("\n        if (tick == P.tick0 + P.n_ticks && c_act) {   // last tick of the launch: the per-tick (transient) columns\n"
                                + "".join(f"            {{ {_col_ptr(k, w, 'c_row', False)} " + " ".join(f"g[{_col_idx(j)}] = r.c{k}[{j}];" for j in range(w)) + " }\n"
                                          for k, w in sorted(transient.items()))
                                + "        }")
        records = "\n".join(
            (f"        if (P.model_hist[{k}]) {{ T* g = static_cast<T*>(P.model_hist[{k}]) + (slot * P.n + row) * {w}; "
             f"const T* g0 = static_cast<const T*>(P.model_cols[{k}]) + (size_t)row" + ("" if _COLUMN_SOA[0] else f" * {w}") + "; "
             + " ".join(f"g[{j}] = g0[{_col_idx(j)}];" for j in range(w)) + " }") if k in cold else
            (f"        if (P.model_hist[{k}]) {{ T* g = static_cast<T*>(P.model_hist[{k}]) + (slot * P.n + row) * {w}; "
             + " ".join(f"g[{j}] = r.c{k}[{j}];" for j in range(w)) + " }") for k, w in reg_cols)
        cold_setup = ("        const uint32_t c_row = blockIdx.x * kWave + threadIdx.x;\n        const bool c_act = c_row < P.n;\n" if (cold or transient) else "")
        if pipe_tp is not None:
            a_slots = sorted(_col_slots(dsl._leaves_of(list(pipe_tp.outputs))) & set(cold))
            if a_slots:      # memory-resident columns the effector stage reads (`r` is the kernel's own register image)
                apply_loads = (cold_setup + "        auto& rw = const_cast<R&>(r);\n" + "".join(
                    f"        if (c_act) {{ {_col_ptr(k, cold[k], 'c_row')} "
                    + " ".join(f"rw.c{k}[{j}] = g[{_col_idx(j)}];" for j in range(cold[k])) + " }\n" for k in a_slots))
        if _WINDOWS:
            # one lane = one entity, a workgroup is one wave (step_kernel.hpp).  Element e of this lane's window sits at
            # W[e * w_n]: w_n = n for the element-major layout of large executors, 1 (a compile-time constant, so the addresses
            # fold) for the entity-major one (see WINDOW_SOA_MIN_ROWS).  Lanes past the last row read the last row's elements and store nothing.
            win_setup = ("        const uint32_t w_row = blockIdx.x * kWave + threadIdx.x;\n"
                         "        const bool w_act = w_row < P.n;\n"
                         + ("        const size_t w_n = P.n;                    // element-major: stride between two elements of one entity\n" if _WINDOW_SOA[0]
                            else "        constexpr size_t w_n = 1;                  // entity-major: an entity's window is contiguous\n")
                         + "".join(f"        T* const W{k} = static_cast<T*>(P.model_cols[{k}]) + (size_t)(w_act ? w_row : P.n - 1) * (size_t){1 if _WINDOW_SOA[0] else rows * width};\n"
                                   for k, (rows, width, _) in _WINDOWS.items()))
        writes_inertia = any(s_.writes_inertia for s_ in pre + post)
        model = f'''
    static constexpr bool kHasModel = true;
    static constexpr bool kWritesInertia = {"true" if writes_inertia else "false"};
    static constexpr bool kPreReadsAccel = {"true" if pre_reads_accel else "false"};{chr(10) + "    static constexpr bool kBodyDead = true;      // no system touches a Body column: the kernel leaves the Body slabs alone" if body_dead else ""}
    template <class T>
    struct Regs {{
{regs}
    }};
    template <class T, int POL>
    __device__ static __forceinline__ void load(const StepParams& P, uint32_t row, bool active, Regs<T>& r) {{
        {zero}
        if (active) {{
{loads}
        }}
    }}
    template <class T, int POL>
    __device__ static __forceinline__ void store(const StepParams& P, uint32_t row, const Regs<T>& r) {{
{stores}
    }}
    template <class T>
    __device__ static __forceinline__ void record(const StepParams& P, size_t slot, uint32_t row, const Regs<T>& r) {{
{records}
    }}
    template <class T>
    __device__ static __forceinline__ void pre(const StepParams& P, uint64_t tick, Regs<T>& r, Quat<T>& q, Vec3<T>& p,
                                               Spatial<T>& v, Vec3<T>& I, T& mass, const Spatial<T>& accel) {{
        (void)P; (void)tick; (void)accel;
{win_setup}{cold_setup}{_emit_systems(pre, cold, store_only)}
    }}
    template <class T>
    __device__ static __forceinline__ void post(const StepParams& P, uint64_t tick, Regs<T>& r, Quat<T>& q, Vec3<T>& p,
                                                Spatial<T>& v, Vec3<T>& I, T& mass, const Spatial<T>& accel) {{
        (void)P; (void)tick; (void)accel;
{win_setup}{cold_setup}{_emit_systems(post, cold, store_only)}{transient_stores}
    }}'''
    wt = pipe_tp.world_torque if pipe_tp is not None else False
    bt = pipe_tp.body_torque if pipe_tp is not None else False
    rv = pipe_tp.reads_velocity if pipe_tp is not None else False
    return f'''struct {name} : NoModel {{
    static constexpr int kOps = {n_aux};
    static constexpr bool kStatic = true;
    static constexpr bool kWorldTorque = {"true" if wt else "false"};
    static constexpr bool kBodyTorque = {"true" if bt else "false"};
    template <int K>
    static constexpr bool uses_aux() {{ return K < kOps; }}
    __device__ static __forceinline__ bool vel_independent(const StepParams&) {{ return {"false" if rv else "true"}; }}
{model}
    template <class T, class R>
    __device__ static __forceinline__ void apply(const StepParams& P, const Vec3<T> (&aux)[kMaxOps], const R& r,
                                                 const Body<T>& b, Wrench<T>& F) {{
        (void)P; (void)r; (void)aux; (void)b; (void)F;
{apply_loads}{body}
    }}
}};
'''


def _graph_fold_kinds(outputs, w: int):
    """Per component of a traced stand-alone fold: "sum" (acc_k + g, g + acc_k or acc_k - g with g free of the accumulator), "keep"
    (acc_k itself) or "zero" (the constant 0) — or None when some component is none of these (such a fold cannot be regrouped)."""
    acc = {f"acc_{k}" for k in range(w)}
    free = lambda e: not (dsl._leaves_of([e]) & acc)
    kinds = []
    for k, e in enumerate(outputs):
        own = lambda x: x.op == "leaf" and x.name == f"acc_{k}"
        if e.op == "const" and e.value == 0.0:
            kinds.append("zero")
        elif own(e):
            kinds.append("keep")
        elif (e.op == "add" and ((own(e.args[0]) and free(e.args[1])) or (own(e.args[1]) and free(e.args[0])))) or \
                (e.op == "sub" and own(e.args[0]) and free(e.args[1])):
            kinds.append("sum")
        else:
            return None
    return kinds


def _emit_fold_stage(fs: "dsl.TracedFoldStage") -> str:
    """A stand-alone fold inside a program (dsl.TracedFoldStage): one lane per SOURCE folds its out-edges in spawn order into
    the scratch column, a second kernel commits scratch -> out on source rows (every fold reads the values from before it
    ran).  Components come from the program's columns (P.model_cols) or the Body columns; the CSR is baked in."""
    j = fs.index
    f = fs.traced.fold
    w = fs.out[2]
    body_ptr = {"world_pos": "P.pos", "world_vel": "P.vel", "inertia": "P.inertia"}
    ptr = lambda name, slot: f"static_cast<const T*>({body_ptr[name]})" if slot is None else f"static_cast<const T*>(P.model_cols[{slot}])"
    leaves = {f"acc_{k}": f"acc[{k}]" for k in range(w)}
    complete = int(getattr(fs, "complete", 0) or 0)
    # the target row of edge e: from the baked CSR, or — a complete graph — slot s of source i is row s + (s >= i)
    dst_of = (lambda e_: f"({e_} + (({e_}) >= i ? 1u : 0u))") if complete else (lambda e_: f"fold{j}_dst[{e_}]")
    e_lo, e_hi = ("0u", f"{complete - 1}u") if complete else (f"fold{j}_start[i]", f"fold{j}_start[i + 1]")
    loads_a, loads_b = [], []
    for i, (n, slot, wn) in enumerate(fs.left):
        for k in range(wn):
            leaves[f"a{i}_{k}"] = f"a{i}[{k}]"
        loads_a.append(f"    const T* a{i} = {ptr(n, slot)} + (size_t)row * {wn};")
    for i, (n, slot, wn) in enumerate(fs.right):
        for k in range(wn):
            leaves[f"b{i}_{k}"] = f"b{i}[{k}]"
        loads_b.append(f"        const T* b{i} = {ptr(n, slot)} + (size_t)(base + {dst_of('e')}) * {wn};")
    body = "\n".join(emit_block([(f"acc[{k}]", e) for k, e in enumerate(fs.traced.outputs)], leaves, indent="        "))
    init = ", ".join(f"T({v!r})" for v in f.init)
    nl = "\n"
    # One lane per source folds its out-edges in order: a chain of dependent gathers, ~350 ns a trip with nothing to hide it behind
    # (a 256-body complete graph: 255 trips, 4 folds per tick = 353 us: profiles/r06_fold_world_time_plain_loop.json).  So the targets'
    # rows are fetched FOLD_BATCH edges at a time — all their loads in flight together — and folded in order afterwards: the same
    # operations in the same order.  Asked for by the fold (dsl.GraphFold.gather_batch: the scans stablehlo.world_program lifts out of a
    # whole-world tick set it; folds written by hand keep the plain loop and their generated text), narrow right-hand sides only
    # (the row buffers are registers).
    batched = ""
    B = int(getattr(f, "gather_batch", 1) or 1)
    if B > 1 and sum(wn for _, _, wn in fs.right) <= 16:
        decl = "".join(f"        T rb{i}[{B}][{wn}];\n" for i, (_, _, wn) in enumerate(fs.right))
        fetch = "".join(f"            {{ const T* g = {ptr(n, slot)} + (size_t)(base + {dst_of('e + u')}) * {wn};\n"
                        f"#pragma unroll\n              for (int k = 0; k < {wn}; k++) rb{i}[u][k] = g[k]; }}\n" for i, (n, slot, wn) in enumerate(fs.right))
        use = "".join(f"            const T* b{i} = rb{i}[u];\n" for i in range(len(fs.right)))
        inner = "\n".join("    " + ln for ln in body.split("\n"))
        batched = (f"    for (; e + {B} <= {e_hi}; e += {B}) {{\n{decl}#pragma unroll\n        for (int u = 0; u < {B}; u++) {{\n{fetch}        }}\n"
                   f"#pragma unroll\n        for (int u = 0; u < {B}; u++) {{\n{use}{inner}\n        }}\n    }}\n")
    n_src = len(fs.src_rows)
    count, stride = fs.replicas if fs.replicas else (1, 0)
    arr = lambda xs: ", ".join(str(int(x)) for x in xs) if xs else "0"
    tables = (f"// (the complete graph over the {complete} rows of a world: no tables — slot s of source i is row s + (s >= i))" if complete else
              f"__device__ const uint32_t fold{j}_src[{max(n_src, 1)}] = {{{arr(fs.src_rows)}}};\n"
              f"__device__ const uint32_t fold{j}_start[{n_src + 1}] = {{{arr(fs.row_start)}}};\n"
              f"__device__ const uint32_t fold{j}_dst[{max(len(fs.dst), 1)}] = {{{arr(fs.dst)}}};")
    src_of = (lambda i_: i_) if complete else (lambda i_: f"fold{j}_src[{i_}]")
    n_edges = complete * (complete - 1) if complete else len(fs.dst)
    # a fold whose output column is none of the columns it reads (dsl.GraphFold.direct_out: the scans of stablehlo.world_program) needs
    # no scratch-then-commit: no lane can read what another has already replaced
    direct = bool(getattr(f, "direct_out", False)) and fs.out[0] not in [n for n, _, _ in fs.left + fs.right]
    out_slot = fs.out[1] if direct else fs.scratch_slot
    kinds = _graph_fold_kinds(fs.traced.outputs, w) if getattr(f, "wave_fold", False) else None
    if kinds is not None:
        # A WAVE per source (dsl.GraphFold.wave_fold: the long scans stablehlo.world_program lifts out of a whole-world tick ask for
        # it).  One lane per source walks its N - 1 out-edges alone — 2,047 dependent trips at 2,048 bodies, 1.4 ms per tick — although the
        # fold is a plain sum: every component is acc_k +- g_k(source, target), acc_k itself, or the constant 0.  So lane l folds edges
        # l, l + 64, ... from zero in slot order, the 64 partial sums are added by a fixed shuffle tree and the initial value joins at
        # the end: the same sum in another association (~1e-16 x sqrt(degree) relative, identical from run to run) — the trade the
        # hub kernels of csrc/pair_kernel.hpp make.  world_program(..., wave_folds=False) keeps the sequential, bit-for-bit fold.
        zero = ", ".join("T(0)" for _ in range(w))
        fin = "\n".join({"sum": f"        if (lane == 0) sc[{k}] = T({f.init[k]!r}) + v{k};", "keep": f"        if (lane == 0) sc[{k}] = T({f.init[k]!r});",
                         "zero": f"        if (lane == 0) sc[{k}] = ({e_hi} > {e_lo}) ? T(0) : T({f.init[k]!r});"}[kd] for k, kd in enumerate(kinds))
        red = "\n".join(f"        T v{k} = acc[{k}];\n#pragma unroll\n        for (int off = 32; off >= 1; off >>= 1) v{k} += __shfl_down(v{k}, off, 64);"
                        for k, kd in enumerate(kinds) if kd == "sum")
        return f'''// ---- fold stage {j}: {fs.name} ({n_edges} edges, {n_src} sources{f", x {count} replicas of {stride} rows" if fs.replicas else ""}), one WAVE per source ----
{tables}
template <class T>
__global__ __launch_bounds__(64) void fold{j}_kernel(const StepParams P) {{
    const uint32_t gi = blockIdx.x, lane = threadIdx.x;
    if (gi >= {n_src * count}u) return;
    const uint32_t i = gi % {max(n_src, 1)}u, base = (gi / {max(n_src, 1)}u) * {stride}u;   // source within the template, replica's first row
    const uint32_t row = base + {src_of("i")};
    if (row >= P.n) return;
{nl.join(loads_a)}
    T acc[{w}] = {{{zero}}};
    for (uint32_t e = {e_lo} + lane; e < {e_hi}; e += 64u) {{
{nl.join(loads_b)}
{body}
    }}
    T* sc = static_cast<T*>(P.model_cols[{out_slot}]) + (size_t)row * {w};
    {{
{red}
{fin}
    }}
}}
template <class T>
__global__ __launch_bounds__(64) void fold{j}_commit(const StepParams P) {{
    const uint32_t gi = blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= {n_src * count}u) return;
    const uint32_t row = (gi / {max(n_src, 1)}u) * {stride}u + {f"(gi % {max(n_src, 1)}u)" if complete else f"fold{j}_src[gi % {max(n_src, 1)}u]"};
    if (row >= P.n) return;
    const T* sc = static_cast<const T*>(P.model_cols[{fs.scratch_slot}]) + (size_t)row * {w};
    T* o = static_cast<T*>(P.model_cols[{fs.out[1]}]) + (size_t)row * {w};
    for (int k = 0; k < {w}; k++) o[k] = sc[k];
}}
'''
    return f'''// ---- fold stage {j}: {fs.name} ({n_edges} edges, {n_src} sources{f", x {count} replicas of {stride} rows" if fs.replicas else ""}) ----
{tables}
template <class T>
__global__ __launch_bounds__(64) void fold{j}_kernel(const StepParams P) {{
    const uint32_t gi = blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= {n_src * count}u) return;
    const uint32_t i = gi % {max(n_src, 1)}u, base = (gi / {max(n_src, 1)}u) * {stride}u;   // source within the template, replica's first row
    const uint32_t row = base + {src_of("i")};
    if (row >= P.n) return;
{nl.join(loads_a)}
    T acc[{w}] = {{{init}}};
{("    uint32_t e = " + e_lo + ";" + nl + batched + "    for (; e < " + e_hi + "; e++) {") if batched else ("    for (uint32_t e = " + e_lo + "; e < " + e_hi + "; e++) {")}
{nl.join(loads_b)}
{body}
    }}
    T* sc = static_cast<T*>(P.model_cols[{out_slot}]) + (size_t)row * {w};
    for (int k = 0; k < {w}; k++) sc[k] = acc[k];
}}
template <class T>
__global__ __launch_bounds__(64) void fold{j}_commit(const StepParams P) {{
    const uint32_t gi = blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= {n_src * count}u) return;
    const uint32_t row = (gi / {max(n_src, 1)}u) * {stride}u + {f"(gi % {max(n_src, 1)}u)" if complete else f"fold{j}_src[gi % {max(n_src, 1)}u]"};
    if (row >= P.n) return;
    const T* sc = static_cast<const T*>(P.model_cols[{fs.scratch_slot}]) + (size_t)row * {w};
    T* o = static_cast<T*>(P.model_cols[{fs.out[1]}]) + (size_t)row * {w};
    for (int k = 0; k < {w}; k++) o[k] = sc[k];
}}
'''


# Integer components (el.PrimitiveType.I64: flags, counters, examples/stablehlo/sim.py:243-251) live in the executor's float
# columns, exact up to 2^53 (2^24 in a float32 program); bitwise operators act on the integer the value holds.
_BITWISE = '''
template <class T> __device__ __forceinline__ T m_bxor(T a, T b) { return T(static_cast<long long>(a) ^ static_cast<long long>(b)); }
template <class T> __device__ __forceinline__ T m_bor(T a, T b) { return T(static_cast<long long>(a) | static_cast<long long>(b)); }
template <class T> __device__ __forceinline__ T m_band(T a, T b) { return T(static_cast<long long>(a) & static_cast<long long>(b)); }
template <class T> __device__ __forceinline__ T m_shl(T a, T b) { return T(static_cast<long long>(a) << static_cast<int>(b)); }
template <class T> __device__ __forceinline__ T m_shr(T a, T b) {      // logical: zero fill
    return T(static_cast<long long>(static_cast<unsigned long long>(static_cast<long long>(a)) >> static_cast<int>(b))); }
// bit casts between integer words held as integral doubles and floating-point values (stablehlo.bitcast_convert): a ui64 is two
// uint32 words (elodin_amd/stablehlo.py U64), so every bit pattern is exact; double programs only
__device__ __forceinline__ double m_bits2f(double hi, double lo) {
    return __hiloint2double(static_cast<int>(static_cast<uint32_t>(hi)), static_cast<int>(static_cast<uint32_t>(lo))); }
__device__ __forceinline__ double m_fbits(double x, int high) {
    return static_cast<double>(static_cast<uint32_t>(high ? __double2hiint(x) : __double2loint(x))); }
__device__ __forceinline__ double m_bits2f32(double w) { return static_cast<double>(__uint_as_float(static_cast<uint32_t>(w))); }
__device__ __forceinline__ double m_f32bits(double x) { return static_cast<double>(__float_as_uint(static_cast<float>(x))); }
'''

# fast_math programs draw their normal samples through the single-precision inverse error function of M. Giles, "Approximating
# the erfinv function" (GPU Computing Gems, 2010): 3e-7 relative against the double routine over (-1, 1), ~15 f32 instructions
# with the hardware log / sqrt where the library's double erfinv is several hundred f64 ones (the Falcon 9 program draws ten
# samples on every guidance tick).  The uniform sample and 1 - u^2 are still formed in double from the 52 random bits, so the
# tails are not quantised by a float argument; they end at 1 - u^2 = 2^-24, i.e. |z| <= 5.4 sigma (probability 6e-8 per draw).
_FAST_ERFINV = '''
__device__ __forceinline__ double m_erfinv_fast(double u) {
    const float t = fmaxf(static_cast<float>((1.0 - u) * (1.0 + u)), 5.9604645e-08f);
    float w = -__logf(t), p;
    if (w < 5.0f) {
        w -= 2.5f;
        p = 2.81022636e-08f;
        p = fmaf(p, w, 3.43273939e-07f);  p = fmaf(p, w, -3.5233877e-06f);  p = fmaf(p, w, -4.39150654e-06f);
        p = fmaf(p, w, 0.00021858087f);   p = fmaf(p, w, -0.00125372503f);  p = fmaf(p, w, -0.00417768164f);
        p = fmaf(p, w, 0.246640727f);     p = fmaf(p, w, 1.50140941f);
    } else {
        w = __fsqrt_rn(w) - 3.0f;
        p = -0.000200214257f;
        p = fmaf(p, w, 0.000100950558f);  p = fmaf(p, w, 0.00134934322f);   p = fmaf(p, w, -0.00367342844f);
        p = fmaf(p, w, 0.00573950773f);   p = fmaf(p, w, -0.0076224613f);   p = fmaf(p, w, 0.00943887047f);
        p = fmaf(p, w, 1.00167406f);      p = fmaf(p, w, 2.83297682f);
    }
    return static_cast<double>(p * static_cast<float>(u));
}
'''


def generate_source(tp, dtype: str, integrator: int, fast_math: bool = False, window_soa: bool = False,
                    column_soa: bool = False, guard_selects: Optional[bool] = None) -> str:
    """tp: dsl.TracedPipe (effectors only) or dsl.TracedProgram (pre | six_dof(effectors) | post).
    fast_math: f32 only — hardware transcendentals / reciprocal division in the generated user code (see _PRELUDE).
    window_soa: window columns are element-major on the device (executors of WINDOW_SOA_MIN_ROWS entities or more).
    guard_selects: expensive `where` arms nobody else needs are computed behind a wave-level branch (_Emitter.block);
    None = the SIXDOF_GUARD_SELECTS environment switch (off unless "1")."""
    _GUARD_SELECTS[0] = (os.environ.get("SIXDOF_GUARD_SELECTS", "") == "1") if guard_selects is None else bool(guard_selects)
    _FUSE_FMA[0] = bool(fast_math) and os.environ.get("SIXDOF_FUSE_FMA", "1") != "0"
    _RELAXED_EMIT[0] = (isinstance(tp, dsl.TracedProgram) and bool(getattr(tp, "fp_contract", False))
                        and os.environ.get("SIXDOF_RELAXED_IEEE_RCP", "") != "1")      # A/B: the shared reciprocals as IEEE divides
    _WINDOW_SOA[0] = bool(window_soa)
    _COLUMN_SOA[0] = bool(column_soa)
    if column_soa and isinstance(tp, dsl.TracedProgram) and tp.fold_stages:
        raise ValueError("element-major program columns are not available for programs with stand-alone folds")
    if fast_math and dtype != "float32":
        raise ValueError("fast_math applies to float32 programs only")
    _TABLES.clear()
    _GATHERS.clear()
    _LANE_TABLES.clear()
    T = {"float64": "double", "float32": "float"}[dtype]
    integ = {0: "kRk4", 1: "kSemiImplicit", 2: "kNone"}[integrator]
    is_prog = isinstance(tp, dsl.TracedProgram)
    if integrator == 2 and not is_prog:
        raise ValueError("integrator NONE needs a program of systems (there is no six_dof stage for effectors to feed)")
    pipe_tp = tp.pipe if is_prog else tp
    n_aux = 0 if is_prog else len(tp.columns)
    n_model = len(tp.columns) if is_prog else 0
    # rows per world when the program exchanges data between the entities of a world inside the wavefront (whole-world StableHLO ticks
    # in lane mode, ops lane_read / lane_read_dyn): sixdof_set_custom_pipe refuses an executor whose row count would split a world
    rows_multiple = lane_stride(tp) if is_prog else 1
    rows_export = (f"// a world of this program is {rows_multiple} consecutive rows (its entities exchange data inside the wavefront)\n"
                   f'extern "C" unsigned sixdof_custom_rows_multiple() {{ return {rows_multiple}u; }}\n\n') if rows_multiple > 1 else ""
    _WINDOWS.clear()
    col_widths = "{0u}"
    if is_prog:
        names_ = [c for c, _ in tp.columns]
        for wname, (wslot, wrows, wwidth) in tp.windows.items():
            _WINDOWS[wslot] = (wrows, wwidth, names_.index(wname + "#head"))
        col_widths = "{" + ", ".join(f"{w}u" + ((" | 0x80000000u" + (" | 0x40000000u" if window_soa else "")) if k in _WINDOWS else
                                                 (" | 0x20000000u" if column_soa else ""))
                                      for k, (_, w) in enumerate(tp.columns)) + "}"
    names = ", ".join(e.__name__ for e in pipe_tp.effectors)
    fast = "#define SIXDOF_FAST_MATH 1\n" if fast_math else ""
    only = _ONLY_POLICY[0]
    if only is None:
        launch_k = lambda pipe, ig, params: (
            f"    if (({params}.streaming & 255u) == kPolNt) hipLaunchKernelGGL((sixdof_step_kernel<{T}, {ig}, {pipe}, kPolNt>), grid, dim3(kWave), 0, s, {params});\n"
            f"    else if (({params}.streaming & 255u) == kPolNtStores || ({params}.streaming & 255u) == kPolSc1Stores) hipLaunchKernelGGL((sixdof_step_kernel<{T}, {ig}, {pipe}, kPolNtStores>), grid, dim3(kWave), 0, s, {params});\n"
            f"    else hipLaunchKernelGGL((sixdof_step_kernel<{T}, {ig}, {pipe}, kPolPlain>), grid, dim3(kWave), 0, s, {params});\n")
    else:
        # ONE cache-policy instantiation (build(policy=...)): the executor's size class is known when its program is built, a
        # policy only changes cache hints (never values), and each instantiation of the step kernel costs a third of the device
        # compile — so an object built for an executor carries the one it will be launched with
        pname = {0: "kPolPlain", 1: "kPolNtStores", 9: "kPolNt"}[only]
        launch_k = lambda pipe, ig, params: (
            f"    hipLaunchKernelGGL((sixdof_step_kernel<{T}, {ig}, {pipe}, {pname}>), grid, dim3(kWave), 0, s, {params});   // built for this executor's cache policy only\n")
    staged = is_prog and bool(tp.fold_stages)
    body_dead = False
    if is_prog and not staged and integrator == 2 and getattr(tp, "body_free", False):
        # the program says no system touches a Body column (a whole-world StableHLO tick, stablehlo.world_system): hold it to that —
        # column slots, window pushes, `tick` and the loop / probe leaves aside, nothing may be read, and only columns written
        systems_ = tp.pre + tp.post
        touched = {n for n in dsl._leaves_of([e for s_ in systems_ for _, e in s_.assign]) if n in _LEAF_CPP or n.startswith("aux") or n in ("aax", "aay", "aaz", "alx", "aly", "alz")}
        wrote = {t for s_ in systems_ for t in s_.written if not (t[0] == "c" or t.startswith("wst"))}
        if touched or wrote or any(s_.writes_inertia or s_.reads_accel for s_ in systems_):
            raise ValueError(f"a program declared free of Body state reads {sorted(touched)} / writes {sorted(wrote)}")
        body_dead = True
    if staged and integrator == 2:
        # the same for a chain of launches (a whole-world tick with fold stages, stablehlo.world_program): every system declares itself free
        # of Body state, every fold reads and writes program columns only — then no link of the chain touches the Body slabs
        systems_ = [s_ for s_ in tp.pre + tp.post if not isinstance(s_, dsl.TracedFoldStage)]
        if systems_ and all(getattr(s_, "body_free", False) for s_ in systems_) and \
                all(slot is not None for fs in tp.fold_stages for _, slot, _ in fs.left + fs.right + [fs.out]):
            touched = {n for n in dsl._leaves_of([e for s_ in systems_ for _, e in s_.assign]) if n in _LEAF_CPP or n.startswith("aux") or n in ("aax", "aay", "aaz", "alx", "aly", "alz")}
            wrote = {t for s_ in systems_ for t in s_.written if not (t[0] == "c" or t.startswith("wst"))}
            if touched or wrote or any(s_.writes_inertia or s_.reads_accel for s_ in systems_):
                raise ValueError(f"a program declared free of Body state reads {sorted(touched)} / writes {sorted(wrote)}")
            body_dead = True
    # bit 17 of the layout word: a one-kernel program whose code never looks at the absolute tick (no `tick` leaf, every system on
    # every tick, no windows) — its launches are identical whatever StepParams::tick0 says, so batches of them may replay from a
    # captured hipGraph like the hand-written kernel's (csrc/sixdof_capi.cpp graph_eligible)
    tick_free = False
    if is_prog and not tp.windows and (not staged or body_dead):      # (a whole-world tick run as a chain of launches — fold stages between
        #                                                                  its systems — replays just the same: every link is captured)
        systems_ = [s_ for s_ in tp.pre + tp.post if not isinstance(s_, dsl.TracedFoldStage)]
        exprs_ = [e for s_ in systems_ for _, e in s_.assign] + (list(pipe_tp.outputs) if pipe_tp is not None else []) + \
                 [e for fs in tp.fold_stages for e in fs.traced.outputs]
        tick_free = ("tick" not in dsl._leaves_of(exprs_) and all(s_.every == 1 and s_.also_at is None for s_ in systems_))
    tick_free_bit = " | (1u << 17)" if tick_free else ""
    if not staged:
        structs = _emit_pipe_struct("PipeCustom", tp if is_prog else None, pipe_tp, tp.pre if is_prog else [], tp.post if is_prog else [],
                                    None, tp.pre_reads_accel if is_prog else False, n_aux, body_dead)
        launch = launch_k("PipeCustom", integ, "(*p)")
        stage_comment = ""
    else:
        # the tick as a chain of launches: systems | fold | systems | ... | (systems | six_dof | systems) | fold | ...
        chain, cur, seen_six = [], [], False
        for s_ in tp.pre:
            if isinstance(s_, dsl.TracedFoldStage):
                if cur:
                    chain.append(("seg", cur, [], False))
                chain.append(("fold", s_))
                cur = []
            else:
                cur.append(s_)
        post_head, k = [], 0
        while k < len(tp.post) and not isinstance(tp.post[k], dsl.TracedFoldStage):
            post_head.append(tp.post[k])
            k += 1
        chain.append(("seg", cur, post_head, True))
        cur = []
        for s_ in tp.post[k:]:
            if isinstance(s_, dsl.TracedFoldStage):
                if cur:
                    chain.append(("seg", cur, [], False))
                chain.append(("fold", s_))
                cur = []
            else:
                cur.append(s_)
        if cur:
            chain.append(("seg", cur, [], False))
        parts, calls = [], []
        last_seg = max(i for i, c in enumerate(chain) if c[0] == "seg")
        for i, c in enumerate(chain):
            if c[0] == "fold":
                fs = c[1]
                parts.append(_emit_fold_stage(fs))
                nb = (len(fs.src_rows) * (fs.replicas[0] if fs.replicas else 1) + 63) // 64
                waves = (getattr(fs.traced.fold, "wave_fold", False) and _graph_fold_kinds(fs.traced.outputs, fs.out[2]) is not None)
                nk = len(fs.src_rows) * (fs.replicas[0] if fs.replicas else 1) if waves else nb      # one wave per source, or one lane
                direct = bool(getattr(fs.traced.fold, "direct_out", False)) and fs.out[0] not in [n_ for n_, _, _ in fs.left + fs.right]
                if nb:
                    calls.append(f"        hipLaunchKernelGGL(fold{fs.index}_kernel<{T}>, dim3({nk}), dim3(64), 0, s, q);" +
                                 ("" if direct else f"\n        hipLaunchKernelGGL(fold{fs.index}_commit<{T}>, dim3({nb}), dim3(64), 0, s, q);"))
                continue
            _, pre, post, six = c
            used = _slots_of(pre + post, pipe_tp.outputs if six else ())
            if i == last_seg:       # the link that records the tick into the history ring holds every column
                used = set(range(len(tp.columns)))
            reads_accel = any(s_.reads_accel for s_ in pre)
            parts.append(_emit_pipe_struct(f"PipeSeg{i}", tp, pipe_tp if six else None, pre, post, used, reads_accel, 0, body_dead))
            ig = integ if six else "kNone"
            tweak = "" if i == last_seg else " qs.hist_ring = 0; qs.hist_pos = qs.hist_vel = qs.hist_accel = qs.hist_force = nullptr;"          # only the last link records the tick
            tweak += "" if six else " qs.accel_in_check = 0;"
            calls.append(f"        {{ StepParams qs = q;{tweak}\n" + launch_k(f"PipeSeg{i}", ig, "qs").replace("    if", "          if", 1).replace("\n    else", "\n          else") + "        }")
        structs = "\n".join(parts)
        stage_comment = "// tick = " + " | ".join(("fold:" + c[1].name) if c[0] == "fold" else ("[" + " | ".join([s_.name for s_ in c[1]] + (["six_dof"] if c[3] and integrator != 2 else []) + [s_.name for s_ in c[2]]) + "]") for c in chain) + "\n"
        launch = ("    for (uint32_t t = 0; t < p->n_ticks; t++) {   // a fold needs every row of the link in front of it: one chain per tick\n"
                  "        StepParams q = *p;\n        q.n_ticks = 1;\n        q.tick0 = p->tick0 + t;\n        q.hist_slot0 = p->hist_slot0 + t;\n"
                  "        if (t) q.accel_in_check = 0;\n"
                  + "\n".join(calls) + "\n    }\n")
    tables = _emit_tables()
    if any(f"m_{k}(" in structs for k in ("bxor", "bor", "band", "shl", "shr", "bits2f", "fbits", "bits2f32", "f32bits")):      # only programs that use them carry this text
        structs = _BITWISE + structs
    fast_erfinv = ""
    if fast_math and "m_erfinv(" in structs:      # only programs that draw normal samples carry (and are keyed on) this text
        structs = structs.replace("m_erfinv(", "m_erfinv_fast(")
        fast_erfinv = _FAST_ERFINV
    # a program traced under dsl.relaxed_arithmetic: a * b + c may contract across statements (kernels.hpp keeps everything else at
    # `contract(on)`: within one source expression only — and the generated text has one operation per statement)
    relaxed = is_prog and getattr(tp, "fp_contract", False)
    contract_on = "#pragma clang fp contract(fast)   // relaxed arithmetic (dsl.relaxed_arithmetic): not the reference's bits\n" if relaxed else ""
    contract_off = "\n#pragma clang fp contract(on)" if relaxed else ""
    if relaxed:      # only relaxed programs carry (and are keyed on) this text
        structs = _RELAXED_PRELUDE + structs
    return f'''// generated by elodin_amd/codegen.py — do not edit.  Effectors: {names}
{stage_comment}{fast}{"#define SIXDOF_TICK_OUT_OF_LINE" + chr(10) if _TICK_OUT_OF_LINE[0] else ""}#include "step_kernel.hpp"

namespace sixdof {{

{_PRELUDE}{fast_erfinv}
{tables}

{contract_on}{structs}{contract_off}
}}  // namespace sixdof

extern "C" unsigned sixdof_custom_abi() {{ return static_cast<unsigned>(sizeof(sixdof::StepParams)); }}
extern "C" unsigned sixdof_custom_layout() {{ return {n_aux}u | ({n_model}u << 8) | ({1 if (is_prog and tp.writes_inertia) else 0}u << 16){tick_free_bit}; }}
// row width of every program column, in slot order (bit 31: window column, kept in HBM; bit 30: built for the element-major
// layout) — checked against the bound columns
extern "C" void sixdof_custom_column_widths(unsigned* out) {{
    static const unsigned w[] = {col_widths};
    for (unsigned k = 0; k < {n_model}u; k++) out[k] = w[k];
}}

{rows_export}extern "C" int sixdof_custom_launch(const sixdof::StepParams* p, int integrator, int dtype, void* stream) {{
    using namespace sixdof;
    if (integrator != {integrator} || dtype != {0 if dtype == "float64" else 1}) return static_cast<int>(hipErrorInvalidValue);
    if (p->n == 0) return static_cast<int>(hipSuccess);
    const dim3 grid((p->n + kWave - 1) / kWave);
    hipStream_t s = static_cast<hipStream_t>(stream);
{launch}    return static_cast<int>(hipGetLastError());
}}
'''


_PAIR_LEAVES = {**{f"acc{k}": f"acc[{k}]" for k in range(6)},
                "ax": "pa[0]", "ay": "pa[1]", "az": "pa[2]", "ma": "ma",
                "bx": "pb[0]", "by": "pb[1]", "bz": "pb[2]", "mb": "mb"}


def _fold_is_additive(tf: "dsl.TracedFold") -> bool:
    """True when every output k of the traced fold is `acc_k`, `acc_k + g`, `g + acc_k` or `acc_k - g` with g free of the
    accumulator — or the constant 0 (el.Force(linear=...) zeroes the torque on every edge)."""
    acc = {f"acc{k}" for k in range(6)}

    def free(e):
        return not (dsl._leaves_of([e]) & acc)
    for k, e in enumerate(tf.outputs):
        own = lambda x: x.op == "leaf" and x.name == f"acc{k}"
        if (e.op == "const" and e.value == 0.0) or own(e):     # a constant other than 0 would be counted once per partial
            continue
        if e.op == "add" and ((own(e.args[0]) and free(e.args[1])) or (own(e.args[1]) and free(e.args[0]))):
            continue
        if e.op == "sub" and own(e.args[0]) and free(e.args[1]):
            continue
        return False
    return True


def generate_pair_source(tf: "dsl.TracedFold", integrator: Optional[int] = None, small: Optional[bool] = None) -> str:
    """A user-written edge_fold function as the PAIR functor of csrc/pair_kernel.hpp (f64).  An executor names the integrator it
    steps with and whether its graph runs as the one-launch small kernel (n <= kPairSmallMax, csrc/sixdof_capi.cpp): the object
    then carries those kernels alone (5 or 2 instead of 11: a third of the device code to compile); launched any other way it
    returns hipErrorInvalidValue.  None keeps both."""
    only_i = -1 if integrator is None else int(integrator)
    if only_i not in (-1, 0, 1):
        raise ValueError(f"edge_fold effectors step under RK4 or the semi-implicit integrator, not integrator {integrator}")
    only_s = -1 if small is None else int(bool(small))
    _GUARD_SELECTS[0], _FUSE_FMA[0], _RELAXED_EMIT[0] = False, False, False      # switches of generate_source: exact arithmetic here
    _TABLES.clear()
    _GATHERS.clear()
    _LANE_TABLES.clear()
    body = "\n".join(emit_block([(f"acc[{k}]", e) for k, e in enumerate(tf.outputs)], _PAIR_LEAVES))
    tables = _emit_tables()
    additive = _fold_is_additive(tf)
    # an object built for ONE of the two launch shapes says so, and the library follows the object rather than re-deriving the choice
    # from the row count and the environment at step time (only specialised objects carry the export: other texts are unchanged)
    only_export = (f'extern "C" int sixdof_custom_pair_only_small() {{ return {only_s}; }}      // 1: the one-launch small-graph kernel only, 0: the multi-kernel tick only (pack, then fold + integrate)\n'
                   if only_s >= 0 else "")
    return f'''// generated by elodin_amd/codegen.py — do not edit.  edge_fold function: {tf.fold.__name__}
#include "pair_kernel.hpp"

namespace sixdof {{

{_PRELUDE}
{tables}

struct PairCustom {{
    // every component is acc_k +/- g_k(a, b) or the constant 0: partial folds over disjoint edge subsets may be summed, so hub
    // sources are folded by whole waves (pair_kernel.hpp 2c); anything else keeps the sequential fold per source
    static constexpr bool kAdditive = {"true" if additive else "false"};
    static constexpr int kEdgeBatch = 1;      // a generated fold keeps the plain per-edge loop (pair_kernel.hpp edge_accumulate_range: registers)
    __device__ static __forceinline__ void fold(double (&acc)[6], const double* pa, double ma, const double* pb,
                                                double mb, double, double) {{
        using T = double;
        (void)pa; (void)ma; (void)pb; (void)mb;
{body}
    }}
}};

}}  // namespace sixdof

extern "C" unsigned sixdof_custom_pair_abi() {{ return static_cast<unsigned>(sizeof(sixdof::PairParams)); }}
{only_export}
extern "C" int sixdof_custom_pair_launch(const sixdof::PairParams* p, int integrator, uint32_t n_ticks, int small,
                                         void* stream, uint64_t* launches) {{
    using namespace sixdof;
    constexpr int kOnlyIntegrator = {only_i}, kOnlySmall = {only_s};      // -1: both
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (kOnlySmall >= 0 && (small != 0) != (kOnlySmall != 0)) return static_cast<int>(hipErrorInvalidValue);
    if constexpr (kOnlySmall != 0) {{
        if (small) return static_cast<int>(launch_pair_small_t<PairCustom, kOnlyIntegrator>(*p, integrator, n_ticks, s, launches));
    }}
    if constexpr (kOnlySmall != 1) {{      // one pack launch (unless p->packed), then fold + integrate per tick
        const hipError_t e = launch_pair_ticks_t<PairCustom, false, kOnlyIntegrator>(*p, integrator, n_ticks, p->packed != 0, s, launches);
        if (e != hipSuccess) return static_cast<int>(e);
    }}
    return static_cast<int>(hipSuccess);
}}
'''


def build_pair(tf: "dsl.TracedFold", integrator: Optional[int] = None, small: Optional[bool] = None) -> Path:
    return _compile(generate_pair_source(tf, integrator, small), "pair")


def generate_graph_fold_source(tf: "dsl.TracedGraphFold") -> str:
    """A stand-alone GraphQuery.edge_fold over arbitrary components: one lane per source entity folds its out-edges
    (CSR by source, spawn order) into a scratch row; a second kernel moves the rows into the output component, so every
    fold sees the component values from before the system ran."""
    _GUARD_SELECTS[0], _FUSE_FMA[0], _RELAXED_EMIT[0] = False, False, False      # switches of generate_source: exact arithmetic here
    _TABLES.clear()
    _GATHERS.clear()
    _LANE_TABLES.clear()
    f = tf.fold
    leaves = {f"acc_{k}": f"acc[{k}]" for k in range(tf.widths[f.out])}
    loads_a, loads_b = [], []
    for i, n in enumerate(f.left):
        for k in range(tf.widths[n]):
            leaves[f"a{i}_{k}"] = f"a{i}[{k}]"
        loads_a.append(f"    const double* a{i} = P.left[{i}] + (size_t)row * {tf.widths[n]};")
    for i, n in enumerate(f.right):
        for k in range(tf.widths[n]):
            leaves[f"b{i}_{k}"] = f"b{i}[{k}]"
        loads_b.append(f"        const double* b{i} = P.right[{i}] + (size_t)P.dst[e] * {tf.widths[n]};")
    w = tf.widths[f.out]
    body = "\n".join(emit_block([(f"acc[{k}]", e) for k, e in enumerate(tf.outputs)], leaves, indent="        "))
    init = ", ".join(repr(v) for v in f.init)
    nl = "\n"
    return f'''// generated by elodin_amd/codegen.py — do not edit.  stand-alone edge_fold system: {f.__name__}
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cmath>
namespace sixdof {{
__device__ __forceinline__ double fast_sqrt(double x) {{ return sqrt(x); }}
__device__ __forceinline__ float fast_sqrt(float x) {{ return sqrtf(x); }}
{_PRELUDE}
{_emit_tables()}
struct GraphFoldParams {{
    const double* left[8];
    const double* right[8];
    double* scratch;             // [n_src, {w}]
    double* out;                 // the output component column, dense over the row set
    const uint32_t* row_start;   // [n_src + 1]
    const uint32_t* dst;         // [n_edges] target rows
    const uint32_t* src_rows;    // [n_src] source rows
    uint32_t n_src;
}};

__global__ __launch_bounds__(256) void graph_fold_kernel(const GraphFoldParams P) {{
    using T = double;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n_src) return;
    const uint32_t row = P.src_rows[i];
{nl.join(loads_a)}
    double acc[{w}] = {{{init}}};
    for (uint32_t e = P.row_start[i]; e < P.row_start[i + 1]; e++) {{
{nl.join(loads_b)}
{body}
    }}
    for (int k = 0; k < {w}; k++) P.scratch[(size_t)i * {w} + k] = acc[k];
}}

__global__ __launch_bounds__(256) void graph_fold_commit(const GraphFoldParams P) {{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n_src) return;
    for (int k = 0; k < {w}; k++) P.out[(size_t)P.src_rows[i] * {w} + k] = P.scratch[(size_t)i * {w} + k];
}}
}}  // namespace sixdof

extern "C" unsigned graph_fold_abi() {{ return static_cast<unsigned>(sizeof(sixdof::GraphFoldParams)); }}
extern "C" int graph_fold_launch(const sixdof::GraphFoldParams* p, unsigned n_ticks, void* stream) {{
    using namespace sixdof;
    if (p->n_src == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((p->n_src + 255) / 256);
    for (unsigned t = 0; t < n_ticks; t++) {{
        hipLaunchKernelGGL(graph_fold_kernel, grid, dim3(256), 0, s, *p);
        hipLaunchKernelGGL(graph_fold_commit, grid, dim3(256), 0, s, *p);
    }}
    return static_cast<int>(hipGetLastError());
}}
'''


def build_graph_fold(tf: "dsl.TracedGraphFold") -> Path:
    return _compile(generate_graph_fold_source(tf), "gfold")


def _headers_digest() -> str:
    h = hashlib.sha1()
    for name in ("step_kernel.hpp", "pair_kernel.hpp", "effectors.hpp", "spatial.hpp", "kernels.hpp"):
        h.update((CSRC / name).read_bytes())
    h.update((PKG.parent / "include" / "sixdof_hip.h").read_bytes())
    return h.hexdigest()


_ONLY_POLICY: List[Optional[int]] = [None]      # build(policy=...): the one cache policy the object is generated for (None: all three)


def policy_for(n_rows: int, row_elems: int, elem_bytes: int) -> Optional[int]:
    """The cache policy csrc/sixdof_capi.cpp fill_step_params picks for an executor of this size (plain loads + nt stores up to
    768 MiB of state, nt both ways beyond), or None when an environment override may pick another at run time (A/B tooling)."""
    if os.environ.get("SIXDOF_STREAMING") is not None or os.environ.get("SIXDOF_ALL_POLICIES", "") == "1":
        return None
    return 1 if int(n_rows) * int(row_elems) * int(elem_bytes) <= (768 << 20) else 9


def build(tp: dsl.TracedPipe, dtype: str = "float64", integrator: int = 0, fast_math: bool = False, window_soa: bool = False,
          column_soa: bool = False, guard_selects: Optional[bool] = None, policy: Optional[int] = None) -> Path:
    """Generate + compile (cached by content hash).  Returns the .so path.  A program that no flag set builds without VGPR
    spills is generated again with its columns memory-resident (_MEMORY_COLUMNS) before giving up.
    policy: instantiate the step kernel for this cache policy only (policy_for); None = all three, selectable at launch."""
    _ONLY_POLICY[0] = policy if policy in (0, 1, 9) else None
    try:
        return _build(tp, dtype, integrator, fast_math, window_soa, column_soa, guard_selects)
    finally:
        _ONLY_POLICY[0] = None


def _build(tp, dtype, integrator, fast_math, window_soa, column_soa, guard_selects) -> Path:
    if getattr(tp, "prebuilt_so", None) is not None:        # dsl.FrozenProgram(prebuilt_so=...): the object exists (stablehlo CLI)
        if not Path(tp.prebuilt_so).exists():
            raise FileNotFoundError(tp.prebuilt_so)
        return Path(tp.prebuilt_so)
    if getattr(tp, "frozen_source", None) is not None:      # dsl.FrozenProgram: the text exists, only the compiler is run
        return _compile(tp.frozen_source, "pipe")
    if dtype == "float32" and getattr(tp, "float32_refused", None):
        # a whole-world StableHLO system whose integer tensors (PRNG words, bit casts, shifts) are carried as integral floats:
        # exact in double programs only (stablehlo.float32_hazards) — refused rather than silently rounded at 2^24
        raise NotImplementedError("this program cannot be built with dtype float32: " + "; ".join(tp.float32_refused[:4]))
    variants = VARIANTS if isinstance(tp, dsl.TracedProgram) else VARIANTS[:2]
    first_src = None
    if isinstance(tp, dsl.TracedProgram) and not tp.fold_stages:
        # a tick that keeps more values live than a wave has registers (a whole-world module: every stage vector of its RK4 waits
        # for the final sum) would fail the first variants one hipcc run after another: estimate each emission order's peak of live
        # values from the DAG and, only when creation order is over budget, try the orders from the leanest up
        _ESTIMATE[0] = {}
        try:
            first_src = generate_variant(tp, variants[0], dtype, integrator, fast_math, window_soa, column_soa, guard_selects)
            est = dict(_ESTIMATE[0])
        finally:
            _ESTIMATE[0] = None
        regs_per_value = 2 if dtype == "float64" else 1
        held = sum(int(w_) for _, w_ in tp.columns)
        if est and (est.get("program", 0) + held) * regs_per_value > PRESSURE_BUDGET_REGS:
            lean = sorted([v for v in ("program", "demand", "pressure") if v in est], key=lambda v: (est[v], VARIANTS.index(v)))
            variants = tuple(lean) + tuple(v for v in variants if v not in lean)
            first_src = None
        last_estimate.clear()
        last_estimate.update(est)
    for k, variant in enumerate(variants):
        try:
            src_k = first_src if (k == 0 and first_src is not None) else generate_variant(tp, variant, dtype, integrator, fast_math, window_soa, column_soa, guard_selects)
            _EXPECT_SCRATCH[0] = variant == "memory"
            try:
                so = _compile(src_k, "pipe")
            finally:
                _EXPECT_SCRATCH[0] = False
            last_variant[0] = variant
            return so
        except SpillError:
            if os.environ.get(ALLOW_SPILLS_ENV, "") == "1" or k == len(variants) - 1:
                raise


# What build() tries, in order, until a build has no VGPR spills: the tick body emitted in the user's program order; in demand
# order (each value right before its first use); in register-pressure order (_pressure_order: the node that ends the most live
# ranges next); with the columns in a memory image instead of registers (_MEMORY_COLUMNS);
# with the tick body out of line (SIXDOF_TICK_OUT_OF_LINE).
VARIANTS = ("program", "demand", "pressure", "memory", "out_of_line")
PRESSURE_BUDGET_REGS = 440       # (estimated live values + column values) x registers per value above which build() reorders its variants
last_estimate: Dict[str, int] = {}      # peak live values per emission order of the last build (diagnostics, bench.py)
last_variant = ["program"]      # the variant the last build() settled on (fixture generators record it)


def generate_variant(tp, variant: str, dtype: str = "float64", integrator: int = 0, fast_math: bool = False, window_soa: bool = False,
                     column_soa: bool = False, guard_selects: Optional[bool] = None) -> str:
    if variant not in VARIANTS:
        raise ValueError(f"variant must be one of {VARIANTS}")
    _EMIT_ORDER[0] = variant if variant in ("demand", "pressure") else "program"
    _MEMORY_COLUMNS[0] = variant == "memory"
    _TICK_OUT_OF_LINE[0] = variant == "out_of_line"
    try:
        return generate_source(tp, dtype, integrator, fast_math, window_soa, column_soa, guard_selects)
    finally:
        _EMIT_ORDER[0], _MEMORY_COLUMNS[0], _TICK_OUT_OF_LINE[0] = "program", False, False


# Generated programs are one long straight-line tick body inside the kernel's tick loop.  Left alone, LLVM's MachineLICM
# hoists every constant materialisation (hundreds of 64-bit literals: polynomial coefficients of the inlined libm
# routines, the program's own constants) out of that loop; the kernel then wants far more than the 512 registers a
# single-wave workgroup can have and spills VGPRs to scratch.  Besides the cost, a spilling build of a fuzz-generated
# program computed wrong values on gfx950 (tests/test_gpu_fuzz.py, seed 2: low halves of spilled coefficients came back
# as garbage whenever a cadenced system's block was skipped; the same source is exact at -O1 and with the flags below), so
# register pressure is treated as a correctness matter: MachineLICM is off, and if VGPR spills remain a second build
# that also sinks instructions back into the loop is tried; the build with fewer spills wins and what was chosen is
# recorded next to the object (<name>.json) and in `last_resources`.
_BASE_FLAGS = ["-mllvm", "-disable-machine-licm"]
_RETRY_FLAGS = ["-mllvm", "-sink-insts-to-avoid-spills"]
# Builds are tried in this order and the FIRST without VGPR spills is kept.  The last resort trades speed for a smaller
# live set (-O1: no unrolling / less hoisting); the fuzz program that miscomputed when spilling is exact at -O1.
_ATTEMPTS = (("-O3", _BASE_FLAGS), ("-O3", _BASE_FLAGS + _RETRY_FLAGS), ("-O1", []))
_CACHE_TAG = "rp3"           # bump when the flag policy changes: cached objects are keyed on it
SPECULATIVE_MIN_SOURCE = 120_000      # characters of generated source from which the flag sets of _ATTEMPTS are compiled concurrently
build_stats: Dict[str, float] = {"hipcc_invocations": 0, "cache_hits": 0}      # since import (bench.py reports them per program)
ALLOW_SPILLS_ENV = "SIXDOF_ALLOW_SPILLS"   # "1": accept a build that still spills VGPRs (known-unsafe on gfx950, see above)
last_resources: Dict[str, int] = {}


class SpillError(RuntimeError):
    """No build of a generated program fits a wave's registers."""


_hipcc_id = None


def _hipcc_version() -> str:
    """Part of the cache key: an object built by another compiler is not the object this policy vetted."""
    global _hipcc_id
    if _hipcc_id is None:
        try:
            _hipcc_id = subprocess.run([HIPCC, "--version"], capture_output=True, text=True).stdout.strip()
        except OSError:
            _hipcc_id = "unknown"
    return _hipcc_id


def _resources(stderr: str) -> Dict[str, int]:
    """Worst case over the kernels of one translation unit, from -Rpass-analysis=kernel-resource-usage."""
    out = {"vgprs": 0, "agprs": 0, "scratch_bytes_per_lane": 0, "sgpr_spills": 0, "vgpr_spills": 0}
    keys = {"VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch_bytes_per_lane",
            "SGPRs Spill": "sgpr_spills", "VGPRs Spill": "vgpr_spills"}
    for line in stderr.splitlines():
        if "remark:" not in line:
            continue
        body = line.split("remark:", 1)[1].split("[-Rpass", 1)[0].strip()
        name, _, val = body.rpartition(":")
        if name.strip() in keys and val.strip().isdigit():
            k = keys[name.strip()]
            out[k] = max(out[k], int(val))
    return out


# ---- precompiled preamble -------------------------------------------------------------------------------------------------
# A small generated program is ~10 KB of straight-line code behind `#include "step_kernel.hpp"`; hipcc spends 1.2 of its 1.6 s
# parsing hip_runtime.h and the kernel headers, once for the device pass and once for the host pass (-ftime-report).  The leading
# comment / #define / #include lines of a generated source are the same for every program of a kind, so they are compiled ONCE
# per (preamble, flag set, header state, compiler) into a device and a host PCH (in a per-user temporary directory), and a build replays hipcc's own plan
# (`hipcc -###`: cc1 device, lld, bundler, cc1 host, ld) with `-include-pch` added to the two cc1 lines.  The device code is
# byte-identical to the plain build's (tests/test_codegen_pch.py compares the disassembly).  Anything unexpected — a plan that
# does not look like the above, a PCH clang refuses, a failing step — falls back to the plain hipcc command, which is also what
# reports real compile errors.  SIXDOF_PCH=0 turns it off.
_PCH_BROKEN: set = set()
PCH_KEEP = 8
PCH_PRUNE_MIN_AGE_S = 600.0
PCH_DIR: List[Optional[Path]] = [None]       # where the PCH pairs live (None: _pch_store()'s default)


def lane_stride(tp) -> int:
    """The largest `rows_per_world` among the program's lane_read nodes (1 when it has none): an executor of this program must
    hold a whole number of such worlds, or the last partial one would read lanes that do not exist."""
    stride, seen, todo = 1, set(), []
    for ts in list(getattr(tp, "pre", None) or []) + list(getattr(tp, "post", None) or []):
        for item in getattr(ts, "assign", None) or []:
            todo.extend(x for x in (item if isinstance(item, (tuple, list)) else [item]) if isinstance(x, dsl.Expr))
    while todo:
        x = todo.pop()
        if id(x) in seen:
            continue
        seen.add(id(x))
        if x.op in ("lane_read", "lane_read_dyn"):
            stride = max(stride, int(x.value[0]))
        todo.extend(a for a in x.args if isinstance(a, dsl.Expr))
        if x.op == "while":
            todo.extend([x.value[1], *x.value[2]])
    return stride


def _preamble(src: str) -> str:
    lines = []
    seen_include = False
    for line in src.splitlines(keepends=True):
        t = line.strip()
        if not t or t.startswith("//"):
            continue                      # (the first comment names the program's systems: not part of what is precompiled)
        elif t.startswith("#include") or (t.startswith("#define") and not t.endswith("\\")):
            lines.append(line)
            seen_include = seen_include or t.startswith("#include")
        else:
            break
    return "".join(lines) if seen_include else ""


def _plan(cmd: List[str]) -> Optional[List[List[str]]]:
    import shlex
    r = subprocess.run([*cmd, "-###"], capture_output=True, text=True)
    if r.returncode != 0:
        return None
    steps = [shlex.split(line) for line in r.stderr.splitlines() if line.startswith(' "')]
    cc1 = [c for c in steps if "-cc1" in c]
    if len(cc1) != 2 or any("-emit-obj" not in c or "-triple" not in c or "-x" not in c for c in cc1):
        return None
    return steps


def _cc1_side(c: List[str]) -> str:
    return "dev" if c[c.index("-triple") + 1].startswith("amdgcn") else "host"


def _header_state() -> str:
    out = []
    for f in sorted(CSRC.glob("*.hpp")) + [PKG.parent / "include" / "sixdof_hip.h"]:
        st = f.stat()
        out.append(f"{f.name}:{st.st_size}:{st.st_mtime_ns}")       # clang refuses a PCH whose inputs changed size or mtime
    return ";".join(out)


def _pch_store() -> Optional[Path]:
    """Where the PCH pairs live: PCH_DIR[0], else $SIXDOF_PCH_DIR, else a per-user directory under the system's temporary directory —
    NOT the package's _jit/: a pair is 24 MB and valid only for this machine's header files (clang checks their size and mtime), so
    it must not travel with the tree the way the built objects do.  The directory is private (0700) and must belong to this user."""
    import tempfile
    store = PCH_DIR[0] or (Path(os.environ["SIXDOF_PCH_DIR"]) if os.environ.get("SIXDOF_PCH_DIR") else
                           Path(tempfile.gettempdir()) / f"elodin_amd_pch_{os.getuid()}")
    try:
        import stat
        store.mkdir(mode=0o700, parents=True, exist_ok=True)
        st = os.lstat(store)
        if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o022):
            return None           # a link, somebody else's, or writable by others: compile without it
    except OSError:
        return None
    return store


def _pch_pair(flags: List[str], preamble: str) -> Optional[Dict[str, str]]:
    """The {dev, host} PCH files of `preamble` under `flags`, built on first use (both passes at once, ~1 s)."""
    import tempfile
    key = hashlib.sha1("\0".join([preamble, *flags, _header_state(), _hipcc_version(), _CACHE_TAG]).encode()).hexdigest()[:16]
    if key in _PCH_BROKEN:
        return None
    store = _pch_store()
    if store is None:
        return None
    pair = {side: str(store / f"pch_{key}.{side}.pch") for side in ("dev", "host")}
    if all(os.path.exists(f) for f in pair.values()):
        try:
            os.utime(pair["dev"])          # last use (the pruning below keeps the most recently used)
        except OSError:
            pass
        return pair
    import time
    t0 = time.perf_counter()
    head = store / f"pch_{key}.hip"
    made = []
    try:
        if not head.exists():
            fd, name = tempfile.mkstemp(prefix=head.name + ".", suffix=".tmp", dir=store)
            os.close(fd)
            made.append(name)
            Path(name).write_text(preamble)
            os.replace(name, head)
        steps = _plan([HIPCC, *flags, str(head), "-o", str(store / f"pch_{key}.so")])
        if steps is None:
            raise RuntimeError("unexpected hipcc plan")
        procs = []
        for c in steps:
            if "-cc1" not in c:
                continue
            side = _cc1_side(c)
            c = list(c)
            c[c.index("-emit-obj")] = "-emit-pch"
            fd, name = tempfile.mkstemp(prefix=f"pch_{key}.{side}.", suffix=".tmp", dir=store)
            os.close(fd)
            made.append(name)
            c[c.index("-o") + 1] = name
            if "-fcuda-include-gpubinary" in c:
                i = c.index("-fcuda-include-gpubinary")
                del c[i:i + 2]
            procs.append((side, name, subprocess.Popen(c, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)))
        bad = None
        for side, name, pr in procs:
            _, err = pr.communicate()
            if pr.returncode != 0:
                bad = err[-400:]
        if bad is not None:
            raise RuntimeError(bad)
        for side, name, _ in procs:
            os.replace(name, pair[side])
        keep = sorted(store.glob("pch_*.dev.pch"), key=lambda f: f.stat().st_mtime, reverse=True)[:PCH_KEEP]
        for f in store.glob("pch_*"):            # header edits and flag sets leave 24 MB pairs behind: the newest few stay
            if not any(f.name.startswith(k.name[:-len("dev.pch")]) for k in keep) and not f.name.endswith(".tmp"):
                try:
                    if time.time() - f.stat().st_mtime < PCH_PRUNE_MIN_AGE_S:
                        continue                 # another thread or process may be half-way through building this one
                    f.unlink()
                except OSError:
                    pass
        build_stats["pch_builds"] = build_stats.get("pch_builds", 0) + 1
        build_stats["pch_build_ms"] = build_stats.get("pch_build_ms", 0.0) + (time.perf_counter() - t0) * 1e3
        return pair
    except (OSError, RuntimeError, ValueError):
        _PCH_BROKEN.add(key)
        return None
    finally:
        for name in made:
            try:
                os.unlink(name)
            except OSError:
                pass


class _Hipcc:
    """One hipcc invocation, replayed step by step with the precompiled preamble when there is one; `stop()` from another
    thread ends it."""

    def __init__(self, cmd: List[str], preamble: str):
        self.cmd, self.preamble = cmd, preamble
        import threading
        self.current = None
        self._group = False
        self.stopped = False
        self._lock = threading.Lock()          # start-of-process and stop() exclude each other: no process is started once stopped
        self.returncode, self.stderr = None, ""

    def _run(self, c, group=False):
        with self._lock:
            if self.stopped:                   # (checked under the lock: a stop() that came first wins, one that comes later finds `current`)
                return 1, ""
            self._group = group
            self.current = subprocess.Popen(c, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, start_new_session=group)
        _, err = self.current.communicate()
        return self.current.returncode, err

    def stop(self):
        import signal
        with self._lock:
            self.stopped = True
            pr, group = self.current, self._group
        if pr is not None and pr.poll() is None:
            try:
                if group:
                    os.killpg(pr.pid, signal.SIGKILL)      # exactly the group started in _run: hipcc's children stop with it
                else:
                    pr.kill()
            except OSError:
                pass

    def run(self):
        import tempfile
        cmd = self.cmd
        steps = pair = None
        if self.preamble and os.environ.get("SIXDOF_PCH", "1") != "0":
            src, out = cmd[-3], cmd[-1]
            pair = _pch_pair([a for a in cmd[1:-3]], self.preamble)
            steps = _plan(cmd) if pair else None
        if steps:
            tmpdir = tempfile.gettempdir()
            scratch = set()
            errs = []
            ok = True
            try:
                for c in steps:
                    for i, a in enumerate(c):
                        f = c[i + 1] if a == "-o" and i + 1 < len(c) else a.split("=", 1)[1] if a.startswith("-output=") else None
                        if f and f != out and os.path.dirname(f) == tmpdir:
                            scratch.add(f)
                    if "-cc1" in c:
                        i = c.index("-x")
                        c = c[:i] + ["-include-pch", pair[_cc1_side(c)]] + c[i:]
                    rc, err = self._run(c)
                    errs.append(err)
                    if rc != 0:
                        ok = False
                        break
            finally:
                for f in scratch:
                    try:
                        os.unlink(f)
                    except OSError:
                        pass
            if ok:
                build_stats["pch_uses"] = build_stats.get("pch_uses", 0) + 1
                self.returncode, self.stderr = 0, "".join(errs)
                return self
            if not self.stopped:
                build_stats["pch_fallbacks"] = build_stats.get("pch_fallbacks", 0) + 1
        self.returncode, self.stderr = self._run(cmd, group=True)
        return self


def _spill_message(name: str, used: Dict[str, int]) -> str:
    return (f"generated kernel {name} spills {used.get('vgpr_spills', 0)} VGPRs to scratch "
            f"({used.get('scratch_bytes_per_lane', 0)} B/lane): the program holds more state than a wave's 512 registers; "
            "split it or narrow its columns")


def _compile(src: str, stem: str) -> Path:
    import json
    import tempfile
    import warnings
    extra = os.environ.get("SIXDOF_JIT_FLAGS", "").split()       # debugging aid, e.g. "-O1" or "-ffp-contract=off"
    allow_spills = os.environ.get(ALLOW_SPILLS_ENV, "") == "1"
    # fast-math builds schedule for instruction-level parallelism: a campaign kernel runs ONE wave per SIMD, so occupancy —
    # what the default strategy trades latency for — buys nothing, while "iterative-ilp" leaves 40 % fewer hazard s_nops and
    # 15 % fewer AGPR moves in the Falcon 9 tick (flight 0.765 -> 0.737 s; "max-ilp": 0.745).  Exact builds keep the
    # default scheduler they were validated under.
    ilp = ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"] if "#define SIXDOF_FAST_MATH" in src and os.environ.get("SIXDOF_ILP_SCHED", "1") != "0" else []
    # ... and WITHOUT LLVM's SLP vectoriser.  On gfx950 it pairs isomorphic f32 operations into v_pk_{add,mul,fma}_f32 (441 packed
    # instructions in the Falcon 9 tick, 700 fewer static VALU), and a packed instruction does take ONE issue slot of a lone wave
    # (tools/ubench/pk_f32.hip, profiles/r05_ubench_pk_f32.txt) — but the register pairs it needs are assembled and taken apart by
    # moves, and the flight is FASTER the less is packed: 3.763 us per tick (default) > 3.690 / 3.673 / 3.653 (slp-threshold 4 / 12 /
    # 32) > 3.622 with the vectoriser off (profiles/r05_falcon9_pk_ab.txt; three interleaved passes, +-0.002).  SIXDOF_SLP=1 keeps it.
    if ilp and os.environ.get("SIXDOF_SLP", "0") != "1":
        ilp = ilp + ["-fno-slp-vectorize"]
    digest = hashlib.sha1((src + _headers_digest() + " ".join(extra + ilp) + _CACHE_TAG + _hipcc_version()).encode()).hexdigest()[:16]
    JIT_DIR.mkdir(exist_ok=True)
    so = JIT_DIR / f"{stem}_{digest}.so"
    meta = JIT_DIR / f"{stem}_{digest}.json"
    global last_resources
    if so.exists():
        build_stats["cache_hits"] += 1
        try:
            os.utime(so)          # last use: lets a cache be pruned by age
        except OSError:
            pass
        last_resources = json.loads(meta.read_text()) if meta.exists() else {}
        if last_resources.get("vgpr_spills", 0) > 0:    # only an opted-in build can be here: say so on every use
            if not allow_spills and not (os.environ.get(ALLOW_SPILLS_ENV, "") == "checked" and last_resources.get("spills_checked")):
                raise SpillError(_spill_message(so.name, last_resources) + f" (cached object built under {ALLOW_SPILLS_ENV}=1; "
                                 "set it again to use it)")
            warnings.warn(_spill_message(so.name, last_resources), RuntimeWarning, stacklevel=3)
        return so
    if meta.exists() and json.loads(meta.read_text()).get("refused") and not allow_spills:
        # no flag set built this source without spills before: do not spend minutes finding that out again
        raise SpillError(_spill_message(so.name, json.loads(meta.read_text())) + " (cached verdict)")
    hip = JIT_DIR / f"{stem}_{digest}.hip"
    temps = []

    def temp(suffix):   # unique per process AND host (ranks on several nodes may share the directory), removed on every exit path
        fd, name = tempfile.mkstemp(prefix=f"{stem}_{digest}.", suffix=suffix, dir=JIT_DIR)
        os.close(fd)
        temps.append(name)
        return name

    def command(opt, flags, out):
        return [HIPCC, "--offload-arch=gfx950", opt, "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
                "-Rpass-analysis=kernel-resource-usage", *flags, *(ilp if opt != "-O1" else []), *extra, f"-I{CSRC}", str(hip), "-o", out]

    def verdict(opt, flags, returncode, stderr):
        if returncode != 0:
            raise RuntimeError(f"hipcc failed for generated code {hip}:\n{stderr[-4000:]}")
        used = dict(_resources(stderr), flags=" ".join([opt, *flags, *(ilp if opt != "-O1" else [])]))
        if used["vgpr_spills"] and not used.get("scratch_bytes_per_lane", 1):
            used["vgpr_spills"] = 0      # spill slots that never reached memory (parked in AGPRs / eliminated): no scratch, no spill
        return used
    try:
        if not hip.exists():
            t = temp(".hip.tmp")
            Path(t).write_text(src)
            os.replace(t, hip)
        best, best_obj = None, None
        preamble = _preamble(src)
        # A LARGE program (the Falcon 9 tick: ~6 s per hipcc run, and its first flag set spills) starts every flag set AT ONCE and
        # keeps the first spill-free one in priority order, stopping the rest: a cold build costs the slowest attempt, not their sum
        # (19 -> ~7 s).  Small programs pass on the first set nearly always: they try one at a time as before.
        speculative = len(src) > SPECULATIVE_MIN_SOURCE and os.environ.get("SIXDOF_SERIAL_BUILD", "") != "1"
        procs = []
        try:
            if speculative:
                import threading
                for opt, flags in _ATTEMPTS:
                    obj = temp(".so.tmp")
                    job = _Hipcc(command(opt, flags, obj), preamble)
                    th = threading.Thread(target=job.run, daemon=True)
                    th.start()
                    procs.append((opt, flags, obj, (job, th)))
                    build_stats["hipcc_invocations"] += 1
            for k, (opt, flags) in enumerate(_ATTEMPTS):
                if speculative:
                    _, _, obj, (job, th) = procs[k]
                    th.join()
                    rc_k, err_k = job.returncode, job.stderr
                else:
                    obj = temp(".so.tmp")
                    job = _Hipcc(command(opt, flags, obj), preamble).run()
                    build_stats["hipcc_invocations"] += 1
                    rc_k, err_k = job.returncode, job.stderr
                if rc_k != 0 and best is not None and best["vgpr_spills"] == 0:
                    continue          # this flag set does not even compile (an -O1 backend assertion): an acceptable object already exists
                used = verdict(opt, flags, rc_k, err_k)
                # fewest VGPR spills first, then least scratch: a build without VGPR spills may still keep something in scratch
                # memory (SGPR spill carriers), and a later flag set that needs none is the better object
                # (the memory-image variant keeps its columns in scratch on purpose: there only the spill count decides)
                by_design = _EXPECT_SCRATCH[0] or "        volatile T c" in src          # (a frozen text of that variant says so itself)
                # ... compared between attempts at the SAME optimisation level only: legitimate scratch (a program's private arrays)
                # must not let the -O1 object — a slower kernel — displace a spill-free -O3 one
                level = lambda u: 1 if u["flags"].startswith("-O1") else 0
                cost = lambda u: (u["vgpr_spills"], level(u), 0 if by_design else u.get("scratch_bytes_per_lane", 0))
                if best is None or cost(used) < cost(best):
                    best, best_obj = used, obj
                # done: no spills and either no scratch at all or no later flag set at this level that could need less
                if used["vgpr_spills"] == 0 and (cost(used)[2] == 0 or not any(o == opt for o, _ in _ATTEMPTS[k + 1:])):
                    break
        finally:
            for _, _, _, (job, th) in procs:
                job.stop()
            for _, _, _, (job, th) in procs:
                th.join()
        if best["vgpr_spills"] > 0 and os.environ.get(ALLOW_SPILLS_ENV, "") == "checked":
            # opt-in middle ground: accept the spilling object only when every spill slot (scratch and SGPR-in-lane) is provably
            # written on every path before it is read (elodin_amd/isa_check.py)
            from . import isa_check
            clean, why, _ = isa_check.check_object(Path(best_obj), int(best["vgpr_spills"]))      # fails closed: no proof = dirty
            if clean:
                best = dict(best, spills_checked=why + " (isa_check)")
                allow_spills = True
            else:
                best = dict(best, spills_check_failed=why)
        if best["vgpr_spills"] > 0:
            if not allow_spills:
                mt = temp(".json.tmp")
                Path(mt).write_text(json.dumps(dict(best, refused=True)))
                os.replace(mt, meta)
                raise SpillError(_spill_message(so.name, best) + f"; no build ({', '.join(o for o, _ in _ATTEMPTS)}) is spill-free. "
                                 f"A spilling build miscomputed on gfx950 before, so it is refused; {ALLOW_SPILLS_ENV}=1 accepts it.")
            warnings.warn(_spill_message(so.name, best), RuntimeWarning, stacklevel=3)
        mt = temp(".json.tmp")
        Path(mt).write_text(json.dumps(best))
        os.replace(mt, meta)
        os.replace(best_obj, so)
        last_resources = best
        return so
    finally:
        import glob
        for t in temps:
            # (+ what a stopped attempt's linker left half-written next to its output: ld.lld writes `<out>.tmpXXXXXX`, then renames)
            for f in [t, *glob.glob(glob.escape(t) + ".tmp*")]:
                try:
                    os.unlink(f)
                except OSError:
                    pass
