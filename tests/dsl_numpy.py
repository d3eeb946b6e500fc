"""Test-side evaluator of a traced effector pipe (elodin_amd.dsl.TracedPipe) with numpy, vectorised over
entities: the independent check of what elodin_amd/codegen.py generates.  TEST INFRASTRUCTURE."""
import sys

import numpy as np

from scipy.special import erfc as _erfc

sys.setrecursionlimit(max(sys.getrecursionlimit(), 20000))     # unrolled factorisations (dsl_mat) are deep, narrow DAGs
from scipy.special import erfinv as _erfinv


def _threefry(k0, k1, c0, c1):
    """threefry2x32 on uint32 arrays (the generator behind jax.random)."""
    k0, k1, x0, x1 = (np.asarray(v, dtype=np.float64).astype(np.uint32) for v in (k0, k1, c0, c1))
    ks = [k0, k1, k0 ^ k1 ^ np.uint32(0x1BD11BDA)]
    rot = ((13, 15, 26, 6), (17, 29, 16, 24))
    with np.errstate(over="ignore"):
        x0 = x0 + ks[0]
        x1 = x1 + ks[1]
        for r in range(5):
            for d in rot[r % 2]:
                x0 = x0 + x1
                x1 = (x1 << np.uint32(d)) | (x1 >> np.uint32(32 - d))
                x1 = x1 ^ x0
            x0 = x0 + ks[(r + 1) % 3]
            x1 = x1 + ks[(r + 2) % 3] + np.uint32(r + 1)
    return x0.astype(np.float64), x1.astype(np.float64)

_F1 = {"sqrt": np.sqrt, "abs": np.abs, "sin": np.sin, "cos": np.cos, "tan": np.tan, "exp": np.exp, "log": np.log,
       "acos": np.arccos, "asin": np.arcsin, "neg": np.negative, "not": np.logical_not, "log1p": np.log1p, "expm1": np.expm1,
       "cbrt": np.cbrt, "floor": np.floor, "ceil": np.ceil, "trunc": np.trunc, "rint": np.rint, "sinh": np.sinh, "cosh": np.cosh,
       "erfc": _erfc, "isfinite": np.isfinite, "erfinv": _erfinv,
       "bits2f32": lambda w: np.asarray(w, dtype=np.float64).astype(np.uint32).view(np.float32).astype(np.float64),
       "f32bits": lambda x: np.asarray(x, dtype=np.float64).astype(np.float32).view(np.uint32).astype(np.float64)}
_F2 = {"add": np.add, "sub": np.subtract, "mul": np.multiply, "div": np.divide, "max": np.fmax, "min": np.fmin,
       "atan2": np.arctan2, "hypot": np.hypot, "pow": np.power, "mod": np.mod, "lt": np.less, "le": np.less_equal, "eq": np.equal, "and": np.logical_and,
       "or": np.logical_or,
       "bxor": lambda a, b: (a.astype(np.int64) ^ b.astype(np.int64)).astype(np.float64),
       "bor": lambda a, b: (a.astype(np.int64) | b.astype(np.int64)).astype(np.float64),
       "band": lambda a, b: (a.astype(np.int64) & b.astype(np.int64)).astype(np.float64),
       "shl": lambda a, b: (a.astype(np.int64) << b.astype(np.int64)).astype(np.float64),
       "shr": lambda a, b: (a.astype(np.int64).view(np.uint64) >> b.astype(np.uint64)).astype(np.float64),
       "bits2f": lambda hi, lo: ((np.asarray(hi, dtype=np.float64).astype(np.uint64) << np.uint64(32))
                                 | np.asarray(lo, dtype=np.float64).astype(np.uint64)).view(np.float64)}


def evaluate(tp, xs, vs, inertia, columns):
    """-> F [n,6] (torque, force) for stage state xs [n,7], vs [n,6], inertia [n,7], columns {name: [n,w]}."""
    n = xs.shape[0]
    leaves = {"qi": xs[:, 0], "qj": xs[:, 1], "qk": xs[:, 2], "qw": xs[:, 3], "px": xs[:, 4], "py": xs[:, 5],
              "pz": xs[:, 6], "wx": vs[:, 0], "wy": vs[:, 1], "wz": vs[:, 2], "vx": vs[:, 3], "vy": vs[:, 4],
              "vz": vs[:, 5], "Ix": inertia[:, 0], "Iy": inertia[:, 1], "Iz": inertia[:, 2], "mass": inertia[:, 6]}
    for slot, (name, w) in enumerate(tp.columns):
        for k in range(w):
            leaves[f"aux{slot}_{k}"] = np.asarray(columns[name], dtype=np.float64).reshape(n, w)[:, k]
    return _world_wrench(np.stack(_eval(tp.outputs, leaves, n), axis=1), xs)


def _world_wrench(out9, xs):
    """[tau_world(3), f(3), tau_body(3)] -> the world-frame wrench [tau, f] the `force` column holds."""
    from tests import np_sixdof
    F = out9[:, :6].copy()
    if np.any(out9[:, 6:] != 0.0):
        F[:, :3] = F[:, :3] + np_sixdof.rot(xs[:, :4], out9[:, 6:])
    return F


# ---- edge folds -------------------------------------------------------------------------------------------------

def fold_force(tf, xs, inertia, src_rows, dst_rows, prior=None):
    """Sequential left fold of a traced edge_fold function (elodin_amd.dsl.TracedFold) per source, edges in the
    given (spawn) order, from a zero accumulator -> F [n,6]; rows that are not a source keep `prior` (or zero).
    Vectorised by rounds: round k applies every source's k-th out-edge."""
    n = xs.shape[0]
    F = np.zeros((n, 6)) if prior is None else np.array(prior, dtype=np.float64)
    src_rows, dst_rows = np.asarray(src_rows), np.asarray(dst_rows)
    order = np.argsort(src_rows, kind="stable")
    s, d = src_rows[order], dst_rows[order]
    rank = np.arange(len(s)) - np.searchsorted(s, s, side="left")     # position of an edge inside its source's list
    acc = np.zeros((n, 6))
    for k in range(int(rank.max()) + 1 if len(s) else 0):
        sel = rank == k
        a, b = s[sel], d[sel]
        lv = {f"acc{j}": acc[a, j] for j in range(6)}
        lv.update({"ax": xs[a, 4], "ay": xs[a, 5], "az": xs[a, 6], "ma": inertia[a, 6],
                   "bx": xs[b, 4], "by": xs[b, 5], "bz": xs[b, 6], "mb": inertia[b, 6]})
        with np.errstate(all="ignore"):
            acc[a] = np.stack(_eval(tf.outputs, lv, len(a)), axis=1)
    srcs = np.unique(s)
    F[srcs] = acc[srcs]                                               # the fold output replaces Force on source rows
    return F


# ---- whole programs (pre systems | six_dof(effectors) | post systems) ------------------------------------------------

def _leaf_arrays(pos, vel, inertia, comps, table, tick, accel=None):
    n = pos.shape[0]
    lv = {"qi": pos[:, 0], "qj": pos[:, 1], "qk": pos[:, 2], "qw": pos[:, 3], "px": pos[:, 4], "py": pos[:, 5],
          "pz": pos[:, 6], "wx": vel[:, 0], "wy": vel[:, 1], "wz": vel[:, 2], "vx": vel[:, 3], "vy": vel[:, 4],
          "vz": vel[:, 5], "Ix": inertia[:, 0], "Iy": inertia[:, 1], "Iz": inertia[:, 2], "mass": inertia[:, 6],
          "tick": np.full(n, float(tick))}
    if accel is not None:
        lv.update({"aax": accel[:, 0], "aay": accel[:, 1], "aaz": accel[:, 2], "alx": accel[:, 3], "aly": accel[:, 4], "alz": accel[:, 5]})
    wins = {v[0]: comps[name] for name, v in getattr(table, "windows", {}).items()}
    for slot, (name, w) in enumerate(table.cols):
        if slot in wins:
            continue                      # a window stays in its [n, rows*width] array (ring order, see window_rows)
        for k in range(w):
            lv[f"{table.prefix}{slot}_{k}"] = comps[name][:, k]
    lv["@win"] = wins
    return lv


def window_rows(comps, name, rows, width):
    """The window component `name` in the reference's order (oldest row first): [n, rows, width] from the ring + head."""
    ring = comps[name].reshape(-1, rows, width)
    head = comps[name + "#head"][:, 0].astype(int)
    idx = (head[:, None] + np.arange(rows)[None, :]) % rows
    return ring[np.arange(ring.shape[0])[:, None], idx]


def _eval(exprs, leaves, n):
    memo = {}

    def ev(e):
        if id(e) in memo:
            return memo[id(e)]
        if e.op == "const":
            r = np.full(n, e.value)
        elif e.op == "leaf":
            r = leaves[e.name]
        elif e.op in _F1:
            r = _F1[e.op](ev(e.args[0]))
        elif e.op in _F2:
            r = _F2[e.op](ev(e.args[0]), ev(e.args[1]))
        elif e.op == "select":
            r = np.where(ev(e.args[0]), ev(e.args[1]), ev(e.args[2]))
        elif e.op == "interp":
            r = np.interp(ev(e.args[0]), np.array(e.value[0]), np.array(e.value[1]))
        elif e.op == "gather":      # constant table in device memory: negative rows count from the end, then clamp (jax's gather)
            from elodin_amd import dsl as _dsl
            key, col, rows, _w = e.value
            i = np.nan_to_num(np.broadcast_to(ev(e.args[0]), (n,)), nan=0.0).astype(np.int64)
            i = np.clip(np.where(i < 0, i + rows, i), 0, rows - 1)
            r = _dsl._GATHER_TABLES[key][i, col]
        elif e.op == "threefry":
            r = _threefry(*[np.broadcast_to(ev(a), (n,)) for a in e.args])[e.value]
        elif e.op == "lane_read":      # the value entity table[i] of the row's world holds (worlds = `stride` consecutive rows)
            stride, table = e.value
            v = np.array(np.broadcast_to(ev(e.args[0]), (n,)))
            rows = np.arange(n)
            src = (rows // stride) * stride + np.asarray(table)[rows % stride]
            r = v[np.minimum(src, n - 1)]
        elif e.op == "lane_read_dyn":  # ... with the table picked by a traced index (row by row)
            stride, tables = e.value
            v = np.array(np.broadcast_to(ev(e.args[0]), (n,)))
            k = np.clip(np.broadcast_to(ev(e.args[1]), (n,)).astype(int), 0, len(tables) - 1)
            rows = np.arange(n)
            src = (rows // stride) * stride + np.asarray(tables)[k, rows % stride]
            r = v[np.minimum(src, n - 1)]
        elif e.op == "fbits":
            words = np.ascontiguousarray(np.broadcast_to(ev(e.args[0]), (n,)), dtype=np.float64).view(np.uint64)
            r = ((words >> np.uint64(32)) if e.value else (words & np.uint64(0xFFFFFFFF))).astype(np.float64)
        elif e.op == "wload":
            slot, rows, width, j, _ = e.value
            head = np.broadcast_to(ev(e.args[0]), (n,)).astype(int)
            idx = np.broadcast_to(ev(e.args[1]), (n,)).astype(int)
            r = leaves["@win"][slot][np.arange(n), ((head + idx) % rows) * width + j]
        elif e.op == "while_out":
            r = ev(e.args[0])[e.value]
        elif e.op == "while":
            # dsl.lax.while_loop, lane by lane: a lane keeps iterating while ITS condition holds
            names, cond, body, max_iter = e.value[:4]
            vals = [np.array(np.broadcast_to(ev(x), (n,)), dtype=np.float64) for x in e.args]
            active = np.ones(n, dtype=bool)
            for _ in range(max_iter):
                inner = dict(leaves)
                inner.update({nm: v for nm, v in zip(names, vals)})
                c = np.broadcast_to(_eval([cond], inner, n)[0].astype(bool), (n,))
                active = active & c
                if not active.any():
                    break
                new = _eval(list(body), inner, n)
                vals = [np.where(active, nv, v) for nv, v in zip(new, vals)]
            r = vals
        else:
            raise ValueError(e.op)
        memo[id(e)] = r
        return r
    with np.errstate(all="ignore"):
        return [np.asarray(ev(e), dtype=np.float64) for e in exprs]


def _run_systems(systems, pos, vel, inertia, comps, table, tick, accel=None):
    body = {"q": ("pos", {"i": 0, "j": 1, "k": 2, "w": 3}), "p": ("pos", {"x": 4, "y": 5, "z": 6}),
            "w": ("vel", {"x": 0, "y": 1, "z": 2}), "v": ("vel", {"x": 3, "y": 4, "z": 5}),
            "I": ("inertia", {"x": 0, "y": 1, "z": 2})}
    arrays = {"pos": pos, "vel": vel, "inertia": inertia}
    for s in systems:
        if hasattr(s, "row_start"):          # a stand-alone fold inside the program (dsl.TracedFoldStage)
            _run_fold_stage(s, pos, vel, inertia, comps)
            continue
        if s.every > 1 and tick % s.every != s.phase and tick != s.also_at:
            continue
        lv = _leaf_arrays(pos, vel, inertia, comps, table, tick, accel)
        vals = [np.array(v, copy=True) for v in _eval([e for _, e in s.assign], lv, pos.shape[0])]   # copies: an output that IS a
        # leaf (a delay line shifting its rows) would otherwise be a view of the column about to be overwritten
        for (target, _), val in zip(s.assign, vals):     # all outputs computed before any is written
            if target == "mass":
                inertia[:, 6] = val
            elif target.startswith("wst"):           # window push: the new row lands on the oldest (physical row = head)
                slot, j = (int(x) for x in target[3:].split("_"))
                name = table.cols[slot][0]
                _, _, width = table.windows[name][:3]
                head = comps[name + "#head"][:, 0].astype(int)      # still the old head: it is assigned after the stores
                comps[name][np.arange(pos.shape[0]), head * width + j] = val
            elif target[0] == "c" and "_" in target:
                slot, k = target[1:].split("_")
                comps[table.cols[int(slot)][0]][:, int(k)] = val
            else:
                arr, idx = body[target[0]]
                arrays[arr][:, idx[target[1]]] = val


def _run_fold_stage(fs, pos, vel, inertia, comps):
    """acc = init; for each out-edge of a source, in spawn order: acc = fn(acc, *left(source), *right(target)); every source's
    result replaces `out` on its row — all folds reading the values from before the stage ran.  Edge by edge (test sizes)."""
    body = {"world_pos": pos, "world_vel": vel, "inertia": inertia}
    col = lambda name: body[name] if name in body else comps[name]
    snap = {n: col(n).copy() for n, _, _ in fs.left + fs.right}
    f = fs.traced.fold
    results, rows_out = [], []
    count, stride = fs.replicas if getattr(fs, "replicas", None) else (1, 0)     # replicas: the same edge template per copy
    for rep in range(count):
        base = rep * stride
        for i, row0 in enumerate(fs.src_rows):
            row = base + row0
            rows_out.append(row)
            acc = np.array(f.init, dtype=np.float64)
            for e in range(fs.row_start[i], fs.row_start[i + 1]):
                lv = {f"acc_{k}": np.array([acc[k]]) for k in range(len(acc))}
                for j, (n, _, w) in enumerate(fs.left):
                    lv.update({f"a{j}_{k}": np.array([snap[n][row, k]]) for k in range(w)})
                for j, (n, _, w) in enumerate(fs.right):
                    lv.update({f"b{j}_{k}": np.array([snap[n][base + fs.dst[e], k]]) for k in range(w)})
                acc = np.array([v[0] for v in _eval(fs.traced.outputs, lv, 1)])
            results.append(acc)
    for row, acc in zip(rows_out, results):
        comps[fs.out[0]][row] = acc
        comps[fs.scratch_name][row] = acc


def program_tick(tp, pos, vel, accel, inertia, comps, tick, dt_g, integrator, dt=None):
    """One tick of a dsl.TracedProgram with numpy (in place on copies); `tick` = count after this tick.  `dt`: the
    six_dof(time_step=) override (rk4.rs:93-100,129 / semi_implicit.rs) when it differs from the world's step."""
    from tests import np_sixdof
    _run_systems(tp.pre, pos, vel, inertia, comps, tp.table, tick, accel if getattr(tp, "pre_reads_accel", False) else None)

    def eff(xs, vs):
        lv = _leaf_arrays(xs, vs, inertia, comps, tp.table, tick)
        return _world_wrench(np.stack(_eval(tp.pipe.outputs, lv, xs.shape[0]), axis=1), xs)
    pos2, vel2, acc2, F = np_sixdof.tick(pos, vel, accel, inertia, eff, dt_g, dt=dt, integrator=integrator)
    pos[:], vel[:], accel[:] = pos2, vel2, acc2
    _run_systems(tp.post, pos, vel, inertia, comps, tp.table, tick, accel)
    return F


def program_tick_systems_only(tp, pos, vel, accel, inertia, comps, tick):
    """A program without six_dof (integrator NONE): the systems and folds of `pre` then `post`, nothing integrated."""
    _run_systems(tp.pre, pos, vel, inertia, comps, tp.table, tick, accel)
    _run_systems(tp.post, pos, vel, inertia, comps, tp.table, tick, accel)


# ---- tracing plain helper functions (models/falcon9.py physics helpers) ----------------------------------------------------

def trace_eval(fn, *args):
    """Call `fn(dsl.np, *leaves)` with every numeric argument replaced by traced leaves (floats -> scalar nodes,
    sequences -> Vec), then evaluate the resulting DAG with numpy.  Mirrors the structure of fn's return value
    (scalar, vector or tuple of those) with floats / 1-D arrays: what the generated kernel code computes, on the host."""
    from elodin_amd import dsl
    leaves, traced = {}, []
    for k, a in enumerate(args):
        if np.ndim(a) == 0:
            leaves[f"arg{k}"] = np.array([float(a)])
            traced.append(dsl.leaf(f"arg{k}"))
        else:
            for j, v in enumerate(a):
                leaves[f"arg{k}_{j}"] = np.array([float(v)])
            traced.append(dsl.Vec([dsl.leaf(f"arg{k}_{j}") for j in range(len(a))]))
    out = fn(dsl.np, *traced)

    def value(o):
        if isinstance(o, (tuple, list)):
            return tuple(value(x) for x in o)
        if isinstance(o, dsl.Vec):
            return np.array([v[0] for v in _eval(list(o.e), leaves, 1)])
        return float(_eval([dsl._lift(o)], leaves, 1)[0][0])
    return value(out)
