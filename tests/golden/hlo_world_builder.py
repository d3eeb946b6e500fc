"""A small emitter of StableHLO TEXT in the spelling jax's lowering prints, and with it the WHOLE-WORLD tick modules the reference
would hand a backend (libs/nox-py/src/cranelift_compile.rs:47-68): `@main` over the world's component columns as entity-batched
`[N, w]` tensors, `jax.vmap`-ed arithmetic, constant-index row gathers for an edge_fold's targets, the fold as a `while` over the
edge slot with `dynamic_slice` by the counter and a `call` of the fold body, `jnp.linalg.norm` as its own function.

Why assembled here: the reference's dumped modules (libs/cranelift-mlir/testdata/*.stablehlo.mlir) are git-LFS POINTERS in the
checkout and there is no jax in this image to lower the examples again.  What IS in the checkout are the fragments of exactly those
dumps as inline test modules (libs/cranelift-mlir/tests/test_gather_3body.rs, test_dynamic_ops_3body.rs, test_while_dyn_slice.rs,
three_body_e2e.rs:16-50 for the shape: 7 inputs, 7 outputs, four functions main + inner + closed_call + norm, a while whose body
holds dynamic_slice + call) — the statements below are spelled like those fragments, the arithmetic is the reference's
(libs/nox-py/src/integrator/rk4.rs:87-135, six_dof.rs:137-150, libs/nox/src/{quaternion,spatial}.rs, examples/three-body/main.py:
59-76, libs/nox-py/src/graph.rs:187-343) in its operation order.  The modules are TEST INPUT: what pins them is the reference's
own golden trajectory (scripts/ci/baseline/three-body-csv -> tests/golden/three_body.csv), not this file.  TEST INFRASTRUCTURE."""
from typing import List, Sequence, Tuple


class V:
    """An SSA value: name, shape, element type."""
    def __init__(self, name, shape, dtype="f64"):
        self.name, self.shape, self.dtype = name, tuple(int(s) for s in shape), dtype

    @property
    def ty(self): return f"tensor<{'x'.join([str(s) for s in self.shape] + [self.dtype])}>"


class Fn:
    def __init__(self, name: str, args: Sequence[Tuple[Sequence[int], str]], public: bool = False):
        self.name, self.public = name, public
        self.args = [V(f"%arg{k}", shp, dt) for k, (shp, dt) in enumerate(args)]
        self.lines: List[str] = []
        self.n = 0
        self.nc = 0
        self.results: List[V] = []

    # -- plumbing --
    def _new(self, shape, dtype="f64") -> V:
        v = V(f"%{self.n}", shape, dtype)
        self.n += 1
        return v

    def emit(self, text: str): self.lines.append("    " + text)

    def const(self, value, shape=(), dtype="f64") -> V:
        name = "%cst" if dtype[0] == "f" else "%c"
        name += f"_{self.nc}" if self.nc else ""
        self.nc += 1
        v = V(name, shape, dtype)
        if isinstance(value, (list, tuple)):
            lit = "[" + ", ".join(self._lit(x, dtype) for x in value) + "]"
        else:
            lit = self._lit(value, dtype)
        self.emit(f"{v.name} = stablehlo.constant dense<{lit}> : {v.ty}")
        return v

    @staticmethod
    def _lit(x, dtype):
        if dtype[0] != "f":
            return str(int(x))
        s = f"{float(x):.17g}"                    # round-trips a double; jax prints the shortest form, any exact form reads the same
        if not any(c in s for c in ".en"):
            s += ".0"
        return s

    def ret(self, *vals: V):
        self.results = list(vals)
        self.emit(f"return {', '.join(v.name for v in vals)} : {', '.join(v.ty for v in vals)}")

    def text(self) -> str:
        args = ", ".join(f"{a.name}: {a.ty}" for a in self.args)
        res = ", ".join(v.ty for v in self.results)
        res = f"({res})" if len(self.results) != 1 else res
        head = f"  func.func {'public' if self.public else 'private'} @{self.name}({args}) -> {res} {{"
        return "\n".join([head] + self.lines + ["  }"])

    # -- element-wise --
    def _bin(self, op, a: V, b: V) -> V:
        assert a.shape == b.shape and a.dtype == b.dtype, (op, a.shape, b.shape)
        r = self._new(a.shape, a.dtype)
        self.emit(f"{r.name} = stablehlo.{op} {a.name}, {b.name} : {r.ty}")
        return r

    def add(self, a, b): return self._bin("add", a, b)
    def sub(self, a, b): return self._bin("subtract", a, b)
    def mul(self, a, b): return self._bin("multiply", a, b)
    def div(self, a, b): return self._bin("divide", a, b)

    def _un(self, op, a: V) -> V:
        r = self._new(a.shape, a.dtype)
        self.emit(f"{r.name} = stablehlo.{op} {a.name} : {r.ty}")
        return r

    def neg(self, a): return self._un("negate", a)
    def sqrt(self, a): return self._un("sqrt", a)

    def convert(self, a: V, dtype: str) -> V:
        r = self._new(a.shape, dtype)
        self.emit(f"{r.name} = stablehlo.convert {a.name} : ({a.ty}) -> {r.ty}")
        return r

    # -- shape --
    def bcast(self, a: V, shape, dims) -> V:
        r = self._new(shape, a.dtype)
        self.emit(f"{r.name} = stablehlo.broadcast_in_dim {a.name}, dims = [{', '.join(str(d) for d in dims)}] : ({a.ty}) -> {r.ty}")
        return r

    def splat(self, value, shape, dtype="f64") -> V:          # how jax spells a Python scalar against an array
        return self.bcast(self.const(value, (), dtype), shape, [])

    def reshape(self, a: V, shape) -> V:
        r = self._new(shape, a.dtype)
        self.emit(f"{r.name} = stablehlo.reshape {a.name} : ({a.ty}) -> {r.ty}")
        return r

    def transpose(self, a: V, perm) -> V:
        r = self._new([a.shape[p] for p in perm], a.dtype)
        self.emit(f"{r.name} = stablehlo.transpose {a.name}, dims = [{', '.join(str(p) for p in perm)}] : ({a.ty}) -> {r.ty}")
        return r

    def slice(self, a: V, ranges) -> V:
        r = self._new([hi - lo for lo, hi in ranges], a.dtype)
        self.emit(f"{r.name} = stablehlo.slice {a.name} [{', '.join(f'{lo}:{hi}' for lo, hi in ranges)}] : ({a.ty}) -> {r.ty}")
        return r

    def concat(self, parts: Sequence[V], dim: int) -> V:
        shape = list(parts[0].shape)
        shape[dim] = sum(p.shape[dim] for p in parts)
        r = self._new(shape, parts[0].dtype)
        self.emit(f"{r.name} = stablehlo.concatenate {', '.join(p.name for p in parts)}, dim = {dim} : ({', '.join(p.ty for p in parts)}) -> {r.ty}")
        return r

    def reduce_sum(self, a: V, dims) -> V:
        zero = self.const(0.0, (), a.dtype)
        r = self._new([s for d, s in enumerate(a.shape) if d not in dims], a.dtype)
        self.emit(f"{r.name} = stablehlo.reduce({a.name} init: {zero.name}) applies stablehlo.add across dimensions = [{', '.join(str(d) for d in dims)}] : "
                  f"({a.ty}, {zero.ty}) -> {r.ty}")
        return r

    def gather_rows(self, table: V, rows: Sequence[int]) -> V:
        """query.rs:599-621 filter_index: constant u32 row numbers, broadcast to [n, 1], one whole row per index."""
        n = len(rows)
        c = self.const(list(rows) if n > 1 else rows[0], (n,), "ui32")
        idx = self.bcast(c, (n, 1), [0])
        r = self._new((n,) + table.shape[1:], table.dtype)
        rank = len(table.shape)
        self.emit(f'{r.name} = "stablehlo.gather"({table.name}, {idx.name}) <{{dimension_numbers = #stablehlo.gather<offset_dims = [{", ".join(str(d) for d in range(1, rank))}], '
                  f'collapsed_slice_dims = [0], start_index_map = [0], index_vector_dim = 1>, indices_are_sorted = false, '
                  f'slice_sizes = array<i64: 1, {", ".join(str(s) for s in table.shape[1:])}>}}> : ({table.ty}, {idx.ty}) -> {r.ty}')
        return r

    def dynamic_slice(self, a: V, starts: Sequence[V], sizes) -> V:
        r = self._new(sizes, a.dtype)
        self.emit(f"{r.name} = stablehlo.dynamic_slice {a.name}, {', '.join(s.name for s in starts)}, sizes = [{', '.join(str(s) for s in sizes)}] : "
                  f"({a.ty}, {', '.join(s.ty for s in starts)}) -> {r.ty}")
        return r


    def _cmp(self, direction, a: V, b: V, kind="FLOAT") -> V:
        r = self._new(a.shape, "i1")
        self.emit(f"{r.name} = stablehlo.compare  {direction}, {a.name}, {b.name},  {kind} : ({a.ty}, {b.ty}) -> {r.ty}")
        return r

    def select(self, c: V, a: V, b: V) -> V:
        r = self._new(a.shape, a.dtype)
        self.emit(f"{r.name} = stablehlo.select {c.name}, {a.name}, {b.name} : {c.ty}, {r.ty}")
        return r

    def maximum(self, a, b): return self._bin("maximum", a, b)
    def bit_and(self, a, b): return self._bin("and", a, b)
    def bit_or(self, a, b): return self._bin("or", a, b)
    def shl(self, a, b): return self._bin("shift_left", a, b)
    def shr(self, a, b): return self._bin("shift_right_logical", a, b)

    def bitcast(self, a: V, dtype: str) -> V:
        r = self._new(a.shape, dtype)
        self.emit(f"{r.name} = stablehlo.bitcast_convert {a.name} : ({a.ty}) -> {r.ty}")
        return r

    def iota(self, shape, dim, dtype) -> V:
        r = self._new(shape, dtype)
        self.emit(f"{r.name} = stablehlo.iota dim = {dim} : {r.ty}")
        return r

    def erf_inv(self, a: V) -> V:
        r = self._new(a.shape, a.dtype)
        self.emit(f"{r.name} = chlo.erf_inv {a.name} : {a.ty} -> {r.ty}")
        return r

    def raw_function(self, text: str):
        """A function given as text (the reconstructed jax.random functions of make_stablehlo_world_golden.py)."""
        self._raw = text

    def call(self, fn: "Fn", args: Sequence[V]) -> List[V]:
        outs = [V(None, v.shape, v.dtype) for v in fn.results]
        base = f"%{self.n}"
        self.n += 1
        for k, o in enumerate(outs):
            o.name = base if len(outs) == 1 else f"{base}#{k}"
        lhs = base if len(outs) == 1 else f"{base}:{len(outs)}"
        rt = outs[0].ty if len(outs) == 1 else "(" + ", ".join(o.ty for o in outs) + ")"
        self.emit(f"{lhs} = call @{fn.name}({', '.join(a.name for a in args)}) : ({', '.join(a.ty for a in args)}) -> {rt}")
        return outs

    def while_counted(self, trips: int, carried: Sequence[V], body) -> List[V]:
        """lax.scan / fori_loop: `while (i < trips)`; body(fn, i, carried values) -> new carried values (the counter is ours)."""
        zero = self.const(0, (), "i64")
        inits = [zero] + list(carried)
        base = f"%{self.n}"
        self.n += 1
        it = [V("%iterArg" if k == 0 else f"%iterArg_{self.nc + k}", v.shape, v.dtype) for k, v in enumerate(inits)]
        self.nc += len(inits)
        self.emit(f"{base}:{len(inits)} = stablehlo.while({', '.join(f'{a.name} = {v.name}' for a, v in zip(it, inits))}) : {', '.join(v.ty for v in inits)}")
        self.emit(" cond {")
        lim = self.const(trips, (), "i64")
        self.lines[-1] = "  " + self.lines[-1]
        cmp_ = self._new((), "i1")
        self.emit(f"  {cmp_.name} = stablehlo.compare  LT, {it[0].name}, {lim.name},  SIGNED : ({it[0].ty}, {lim.ty}) -> {cmp_.ty}")
        self.emit(f"  stablehlo.return {cmp_.name} : {cmp_.ty}")
        self.emit("} do {")
        mark = len(self.lines)
        new = body(self, it[0], it[1:])
        one = self.const(1, (), "i64")
        nxt = self.add(it[0], one)
        outs = [nxt] + list(new)
        self.emit(f"stablehlo.return {', '.join(o.name for o in outs)} : {', '.join(o.ty for o in outs)}")
        for k in range(mark, len(self.lines)):
            self.lines[k] = "  " + self.lines[k]
        self.emit("}")
        return [V(f"{base}#{k}", v.shape, v.dtype) for k, v in enumerate(inits)][1:]


def module(fns: Sequence[Fn]) -> str:
    return "module @module {\n" + "\n".join(f.text() for f in fns) + "\n}\n"


# ---- the reference's arithmetic, entity-batched ------------------------------------------------------------------------------------

def col(f: Fn, x: V, k: int) -> V:
    """x[:, k] of an [n, w] tensor as [n]."""
    return f.reshape(f.slice(x, [(0, x.shape[0]), (k, k + 1)]), (x.shape[0],))


def stack_cols(f: Fn, cols: Sequence[V]) -> V:
    n = cols[0].shape[0]
    return f.concat([f.reshape(c, (n, 1)) for c in cols], 1)


def quat_mul(f: Fn, l, r):
    """Hamilton product, scalar-last, on tuples of four [n] vectors (libs/nox/src/quaternion.rs:268-281)."""
    li, lj, lk, lw = l
    ri, rj, rk, rw = r
    m, a, s = f.mul, f.add, f.sub
    i = s(a(a(m(lw, ri), m(li, rw)), m(lj, rk)), m(lk, rj))
    j = a(a(s(m(lw, rj), m(li, rk)), m(lj, rw)), m(lk, ri))
    k = a(s(a(m(lw, rk), m(li, rj)), m(lj, ri)), m(lk, rw))
    w = s(s(s(m(lw, rw), m(li, ri)), m(lj, rj)), m(lk, rk))
    return i, j, k, w


def quat_dot(f: Fn, q):
    m, a = f.mul, f.add
    return a(a(a(m(q[0], q[0]), m(q[1], q[1])), m(q[2], q[2])), m(q[3], q[3]))


def quat_inverse(f: Fn, q):                      # conjugate / norm_squared, quaternion.rs:141-155
    d = quat_dot(f, q)
    return f.div(f.neg(q[0]), d), f.div(f.neg(q[1]), d), f.div(f.neg(q[2]), d), f.div(q[3], d)


def quat_normalize(f: Fn, q):                    # quaternion.rs:147-149
    n = f.sqrt(quat_dot(f, q))
    return tuple(f.div(c, n) for c in q)


def quat_rotate(f: Fn, q, v):                    # q (x) [v, 0] (x) inverse(q), quaternion.rs:283-305
    zero = f.splat(0.0, v[0].shape)
    inv = quat_inverse(f, q)
    t = quat_mul(f, q, (v[0], v[1], v[2], zero))
    r = quat_mul(f, t, inv)
    return r[0], r[1], r[2]


def transform_add_motion(f: Fn, x: V, m6):
    """SpatialTransform + SpatialMotion (libs/nox/src/spatial.rs:530-549): q' = normalize(q + (w/2, 0) (x) q), p' = p + v.
    x: [n, 7]; m6: six [n] vectors (angular first).  -> [n, 7]"""
    n = x.shape[0]
    q = tuple(col(f, x, k) for k in range(4))
    two = f.splat(2.0, (n,))
    ho = (f.div(m6[0], two), f.div(m6[1], two), f.div(m6[2], two), f.splat(0.0, (n,)))
    t = quat_mul(f, ho, q)
    qn = quat_normalize(f, tuple(f.add(a, b) for a, b in zip(q, t)))
    p = [f.add(col(f, x, 4 + k), m6[3 + k]) for k in range(3)]
    return stack_cols(f, list(qn) + p)


def calc_accel(f: Fn, force: V, inertia: V, x: V) -> V:
    """six_dof.rs:137-146: world accel = q * ((q^-1 * F) / I), diagonal inertia, no gyroscopic term."""
    q = tuple(col(f, x, k) for k in range(4))
    qi = quat_inverse(f, q)
    bt = quat_rotate(f, qi, tuple(col(f, force, k) for k in range(3)))
    bf = quat_rotate(f, qi, tuple(col(f, force, 3 + k) for k in range(3)))
    mass = col(f, inertia, 6)
    ba_lin = tuple(f.div(c, mass) for c in bf)
    ba_ang = tuple(f.div(c, col(f, inertia, k)) for k, c in enumerate(bt))
    return stack_cols(f, list(quat_rotate(f, q, ba_ang)) + list(quat_rotate(f, q, ba_lin)))


def scaled(f: Fn, h: V, x: V) -> List[V]:
    """h * x for a rank-0 h and an [n, 6] x, as six [n] columns."""
    n = x.shape[0]
    hb = f.bcast(h, (n,), [])
    return [f.mul(hb, col(f, x, k)) for k in range(6)]


G = 6.6743e-11


def edge_fold_world(n: int, targets: dict, fold: str, params: Sequence[float]):
    """A world of n bodies whose forces come from ONE edge_fold over (WorldPos, Inertia) as ONE entity-batched tick:
    increment_sim_tick | six_dof(fold), RK4.  targets: source row -> target rows in spawn order (every source the same count e, so
    graph.rs:187-235 makes one bucket); fold: "newton" (examples/three-body/main.py:64-71, params = (G,)) or "softened"
    (examples/n-body/sim.py:349-361, params = (K, eps)).  -> (module text, slots)
    slots: the seven (component, shape, entity_axis_elided) of @main's arguments = its results, in the module's order."""
    e = len(targets[0])
    assert sorted(targets) == list(range(n)) and all(len(t) == e for t in targets.values())
    # ---- @norm: jnp.linalg.norm over the last axis, vmapped over the sources -------------------------------------------------------
    norm = Fn("norm", [((n, 3), "f64")])
    sq = norm.mul(norm.args[0], norm.args[0])
    norm.ret(norm.sqrt(norm.reduce_sum(sq, [1])))
    # ---- @closed_call: the fold body for all sources at once: (force, a_pos, a_inertia, b_pos, b_inertia) --------------------------
    cc = Fn("closed_call", [((n, 6), "f64"), ((n, 7), "f64"), ((n, 7), "f64"), ((n, 7), "f64"), ((n, 7), "f64")])
    force, a_pos, a_in, b_pos, b_in = cc.args
    ma, mb = col(cc, a_in, 6), col(cc, b_in, 6)
    if fold == "newton":
        r = cc.sub(cc.slice(a_pos, [(0, n), (4, 7)]), cc.slice(b_pos, [(0, n), (4, 7)]))
        (nr,) = cc.call(norm, [r])
        gmm = cc.mul(cc.mul(cc.splat(params[0], (n,)), mb), ma)                          # G * M * m
        num = cc.mul(cc.bcast(gmm, (n, 3), [0]), r)                                     # ... * r
        den = cc.mul(cc.mul(nr, nr), nr)                                                # norm * norm * norm
        fvec = cc.div(num, cc.bcast(den, (n, 3), [0]))
        lin = cc.sub(cc.slice(force, [(0, n), (3, 6)]), fvec)                           # el.Force(linear = force.force() - f): torque = 0
        cc.ret(cc.concat([cc.splat(0.0, (n, 3)), lin], 1))
    else:
        r = cc.sub(cc.slice(b_pos, [(0, n), (4, 7)]), cc.slice(a_pos, [(0, n), (4, 7)]))
        d2 = cc.add(cc.reduce_sum(cc.mul(r, r), [1]), cc.splat(params[1], (n,)))       # jnp.dot(r, r) + SOFTENING
        inv = cc.div(cc.splat(1.0, (n,)), cc.sqrt(d2))                                  # jnp.reciprocal(jnp.sqrt(.))
        inv3 = cc.mul(cc.mul(inv, inv), inv)
        sc = cc.mul(cc.mul(cc.mul(cc.splat(params[0], (n,)), ma), mb), inv3)            # K * m_a * m_b * inv_dist3
        tq = cc.add(cc.slice(force, [(0, n), (0, 3)]), cc.splat(0.0, (n, 3)))           # acc + SpatialForce(linear = ...): torque + 0
        lin = cc.add(cc.slice(force, [(0, n), (3, 6)]), cc.mul(cc.bcast(sc, (n, 3), [0]), r))
        cc.ret(cc.concat([tq, lin], 1))
        norm = None
    # ---- @inner: the whole tick ----------------------------------------------------------------------------------------------------
    inner = Fn("inner", [((), "i64"), ((), "f64"), ((n, 7), "f64"), ((n, 6), "f64"), ((n, 6), "f64"), ((n, 6), "f64"), ((n, 7), "f64")])
    tick, dt, pos0, vel0, accel_in, _force_in, inertia = inner.args
    f = inner
    tick1 = f.add(tick, f.const(1, (), "i64"))                                     # increment_sim_tick, globals.rs:40-44

    def pipe(xs: V):
        """clear_forces | fold | calc_accel on the stage transforms xs (six_dof.rs:176)."""
        zero_force = f.splat(0.0, (n, 6))                                          # clear_forces + the fold's init value el.Force()
        # graph.rs:187-235: per source its row and its targets' rows by constant-index gathers, concatenated over the sources
        frm_p = f.concat([f.gather_rows(xs, [s]) for s in range(n)], 0)
        frm_i = f.concat([f.gather_rows(inertia, [s]) for s in range(n)], 0)
        to_p = f.concat([f.reshape(f.gather_rows(xs, targets[s]), (1, e, 7)) for s in range(n)], 0)         # [n, e, 7]
        to_i = f.concat([f.reshape(f.gather_rows(inertia, targets[s]), (1, e, 7)) for s in range(n)], 0)
        # vmap over the sources of a scan over the edge slot: the scanned axis comes first (test_transpose_3body_pattern)
        to_p, to_i = f.transpose(to_p, [1, 0, 2]), f.transpose(to_i, [1, 0, 2])

        def body(fb, i, carried):
            acc, cfp, cfi, ctp, cti = carried
            z = fb.const(0, (), "i64")
            bp = fb.reshape(fb.dynamic_slice(ctp, [i, z, z], (1, n, 7)), (n, 7))   # test_dynamic_slice_3body_pattern
            z2 = fb.const(0, (), "i64")
            bi = fb.reshape(fb.dynamic_slice(cti, [i, z2, z2], (1, n, 7)), (n, 7))
            (new,) = fb.call(cc, [acc, cfp, cfi, bp, bi])
            return [new, cfp, cfi, ctp, cti]
        force_out = f.while_counted(e, [zero_force, frm_p, frm_i, to_p, to_i], body)[0]
        return force_out, calc_accel(f, force_out, inertia, xs)

    V_, A_ = [], []
    prev_a = accel_in
    F_last = None
    for c_ in (0.0, 0.5, 0.5, 1.0):                                                # rk4.rs:110-121
        h = f.mul(dt, f.const(c_))
        xs = transform_add_motion(f, pos0, scaled(f, h, vel0))                     # x0 (+) h v0: WorldVel is reset to v0 after every stage
        vs = f.add(vel0, stack_cols(f, scaled(f, h, prev_a)))
        F_last, a = pipe(xs)
        V_.append(vs)
        A_.append(a)
        prev_a = a
    g = f.mul(dt, f.const(1.0 / 6.0))                                              # rk4.rs:129

    def combine(K):
        two = f.splat(2.0, K[0].shape)
        return f.add(f.add(f.add(K[0], f.mul(two, K[1])), f.mul(two, K[2])), K[3])
    pos1 = transform_add_motion(f, pos0, scaled(f, g, combine(V_)))
    vel1 = f.add(vel0, stack_cols(f, scaled(f, g, combine(A_))))
    inner.ret(tick1, dt, pos1, vel1, A_[3], F_last, inertia)
    # ---- @main --------------------------------------------------------------------------------------------------------------------
    main = Fn("main", [(a.shape, a.dtype) for a in inner.args], public=True)
    main.ret(*main.call(inner, main.args))
    slots = [("tick", [], True), ("simulation_time_step", [], True), ("world_pos", [n, 7], False), ("world_vel", [n, 6], False),
             ("world_accel", [n, 6], False), ("force", [n, 6], False), ("inertia", [n, 7], False)]
    return module([main, inner, cc] + ([norm] if norm is not None else [])), slots


def three_body_world():
    """examples/three-body as ONE entity-batched tick: three bodies, two out-edges per source in spawn order (main.py:80-87)."""
    return edge_fold_world(3, {0: [1, 2], 1: [0, 2], 2: [0, 1]}, "newton", (G,))


def nbody_world(n: int, k: float, eps: float):
    """examples/n-body as ONE entity-batched tick: n bodies, the complete gravity graph in spawn order (sim.py:330-338: for every
    source its targets in ascending order), the softened fold of sim.py:349-361."""
    return edge_fold_world(n, {s: [t for t in range(n) if t != s] for s in range(n)}, "softened", (k, eps))


def independent_bodies_world(n: int):
    """BASELINE configs[1] as a whole-world tick: n bodies, constant gravity g*m plus a body-frame torque column (`torque` [n, 3],
    rotated into the world frame by the stage attitude), RK4 — no edges, so every statement is entity-parallel.
    -> (module text, slots); @main: tick, dt, world_pos, world_vel, world_accel, force, inertia, torque -> the same eight."""
    inner = Fn("inner", [((), "i64"), ((), "f64"), ((n, 7), "f64"), ((n, 6), "f64"), ((n, 6), "f64"), ((n, 6), "f64"), ((n, 7), "f64"), ((n, 3), "f64")])
    tick, dt, pos0, vel0, accel_in, _force_in, inertia, torque = inner.args
    f = inner
    tick1 = f.add(tick, f.const(1, (), "i64"))
    gvec = (0.0, 0.0, -9.81)

    def pipe(xs: V):
        q = tuple(col(f, xs, k) for k in range(4))
        zero = f.splat(0.0, (n,))
        mass = col(f, inertia, 6)
        tw = quat_rotate(f, q, tuple(col(f, torque, k) for k in range(3)))          # force + SpatialForce(torque = q @ t)
        cols = [f.add(zero, tw[k]) for k in range(3)] + [f.add(zero, f.mul(f.splat(gvec[k], (n,)), mass)) for k in range(3)]
        force = stack_cols(f, cols)
        return force, calc_accel(f, force, inertia, xs)
    V_, A_, prev_a, F_last = [], [], accel_in, None
    for c_ in (0.0, 0.5, 0.5, 1.0):
        h = f.mul(dt, f.const(c_))
        xs = transform_add_motion(f, pos0, scaled(f, h, vel0))
        vs = f.add(vel0, stack_cols(f, scaled(f, h, prev_a)))
        F_last, a = pipe(xs)
        V_.append(vs)
        A_.append(a)
        prev_a = a
    g = f.mul(dt, f.const(1.0 / 6.0))

    def combine(K):
        two = f.splat(2.0, K[0].shape)
        return f.add(f.add(f.add(K[0], f.mul(two, K[1])), f.mul(two, K[2])), K[3])
    pos1 = transform_add_motion(f, pos0, scaled(f, g, combine(V_)))
    vel1 = f.add(vel0, stack_cols(f, scaled(f, g, combine(A_))))
    inner.ret(tick1, dt, pos1, vel1, A_[3], F_last, inertia, torque)
    main = Fn("main", [(a.shape, a.dtype) for a in inner.args], public=True)
    main.ret(*main.call(inner, main.args))
    slots = [("tick", [], True), ("simulation_time_step", [], True), ("world_pos", [n, 7], False), ("world_vel", [n, 6], False),
             ("world_accel", [n, 6], False), ("force", [n, 6], False), ("inertia", [n, 7], False), ("torque", [n, 3], False)]
    return module([main, inner]), slots


class RawFn:
    """A private function whose text exists already; `results` is what call() needs to type its outputs."""
    def __init__(self, name: str, text: str, results: Sequence[V]):
        self.name, self._text, self.results = name, text, list(results)

    def text(self) -> str: return self._text


def ball_world():
    """examples/ball as ONE tick of its singleton world (one entity: every column has its entity axis elided, system.rs:12-23):
    increment_sim_tick | sample_wind | bounce | six_dof(gravity | apply_drag), RK4 — examples/ball/sim.py:56-125.  sample_wind is
    `random.normal(random.key(seed), (3,))`: jax.random's threefry2x32 + mantissa construction + erf_inv, the functions reconstructed
    in tests/golden/make_stablehlo_world_golden.py and pinned there on the reference's known answers.
    -> (module text, slots): @main(tick, seed, wind, world_pos, world_vel, force, inertia, world_accel, simulation_time_step) -> the same
    nine (the argument list of libs/cranelift-mlir/tests/test_uniform_pipeline.rs:86-99)."""
    import importlib.util
    from pathlib import Path
    spec = importlib.util.spec_from_file_location("_mk_world_golden_parts", Path(__file__).with_name("hlo_random_parts.py"))
    parts = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(parts)
    u3 = V(None, (3,), "ui32")
    threefry = RawFn("threefry2x32", parts.THREEFRY, [u3, u3])
    closed = RawFn("closed_call", parts.closed_call_fn(False), [])
    inner = Fn("inner", [((), "i64"), ((), "i64"), ((3,), "f64"), ((7,), "f64"), ((6,), "f64"), ((6,), "f64"), ((7,), "f64"), ((6,), "f64"), ((), "f64")])
    tick, seed, _wind_in, pos_in, vel_in, _force_in, inertia_in, accel_in, dt = inner.args
    f = inner
    tick1 = f.add(tick, f.const(1, (), "i64"))
    # ---- sample_wind: random.normal(random.key(seed), (3,)) ----------------------------------------------------------------------
    k0 = f.convert(f.shr(seed, f.const(32, (), "i64")), "ui32")                                  # threefry_seed: the two halves of the int64 seed
    k1 = f.convert(f.bit_and(seed, f.const(4294967295, (), "i64")), "ui32")
    ctr = f.iota((3,), 0, "ui64")                                                                 # partitionable counters (0, i)
    c_hi = f.convert(f.shr(ctr, f.splat(32, (3,), "ui64")), "ui32")
    c_lo = f.convert(f.bit_and(ctr, f.splat(4294967295, (3,), "ui64")), "ui32")
    b_hi, b_lo = f.call(threefry, [k0, k1, c_hi, c_lo])
    bits = f.bit_or(f.shl(f.convert(b_hi, "ui64"), f.splat(32, (3,), "ui64")), f.convert(b_lo, "ui64"))
    mant = f.bit_or(f.shr(bits, f.splat(12, (3,), "ui64")), f.splat(4607182418800017408, (3,), "ui64"))      # | bits of 1.0
    u01 = f.sub(f.bitcast(mant, "f64"), f.splat(1.0, (3,)))
    lo = f.splat(-0.99999999999999989, (3,))                                                     # nextafter(-1, inf)
    u = f.maximum(lo, f.add(f.mul(u01, f.splat(2.0, (3,))), lo))
    wind = f.mul(f.splat(1.4142135623730951, (3,)), f.erf_inv(u))
    # ---- batch1 -> batched (query.rs:627-631) for the vmapped arithmetic ----------------------------------------------------------
    pos0, vel_b, inertia, accel_b = (f.reshape(x, (1,) + x.shape) for x in (pos_in, vel_in, inertia_in, accel_in))
    windb = f.reshape(wind, (1, 3))
    # ---- bounce (sim.py:65-73): lax.cond on max(p.z, v.z) < 0 -> select under vmap ----------------------------------------------
    pz, vz = col(f, pos0, 6), col(f, vel_b, 5)
    hit = f._cmp("LT", f.maximum(pz, vz), f.splat(0.0, (1,)))
    refl = f.mul(f.mul(f.slice(vel_b, [(0, 1), (3, 6)]), f.bcast(f.const([1.0, 1.0, -1.0], (3,)), (1, 3), [1])), f.splat(0.85, (1, 3)))
    bounced = f.concat([f.splat(0.0, (1, 3)), refl], 1)                                          # el.SpatialMotion(linear=...): angular = 0
    vel0 = f.select(f.bcast(hit, (1, 6), [0]), bounced, vel_b)
    n = 1

    def pipe(xs: V, vs: V):
        mass = col(f, inertia, 6)
        zero = f.splat(0.0, (n,))
        grav = [f.add(zero, zero) for _ in range(3)] + [f.add(zero, f.mul(f.splat(g_, (n,)), mass)) for g_ in (0.0, 0.0, -9.81)]     # gravity: f + SpatialForce(linear = g * m)
        fl = [f.sub(col(f, windb, k), col(f, vs, 3 + k)) for k in range(3)]                     # apply_drag (sim.py:96-116)
        V_ = f.sqrt(f.add(f.add(f.mul(fl[0], fl[0]), f.mul(fl[1], fl[1])), f.mul(fl[2], fl[2])))
        drag = f.mul(f.splat(0.5, (n,)), f.mul(f.mul(f.splat(0.5 * 1.225, (n,)), f.mul(V_, V_)), f.splat(2 * 3.1415 * 0.2 ** 2, (n,))))
        cols = [f.splat(0.0, (n,)) for _ in range(3)] + [f.add(grav[3 + k], f.mul(drag, f.div(fl[k], V_))) for k in range(3)]       # torque zeroed: SpatialForce(linear=...)
        force = stack_cols(f, cols)
        return force, calc_accel(f, force, inertia, xs)
    V4, A4, prev_a, F_last = [], [], accel_b, None
    for c_ in (0.0, 0.5, 0.5, 1.0):
        h = f.mul(dt, f.const(c_))
        xs = transform_add_motion(f, pos0, scaled(f, h, vel0))
        vs = f.add(vel0, stack_cols(f, scaled(f, h, prev_a)))
        F_last, a_ = pipe(xs, vs)
        V4.append(vs)
        A4.append(a_)
        prev_a = a_
    g = f.mul(dt, f.const(1.0 / 6.0))

    def combine(K):
        two = f.splat(2.0, K[0].shape)
        return f.add(f.add(f.add(K[0], f.mul(two, K[1])), f.mul(two, K[2])), K[3])
    pos1 = transform_add_motion(f, pos0, scaled(f, g, combine(V4)))
    vel1 = f.add(vel0, stack_cols(f, scaled(f, g, combine(A4))))
    sq = lambda x: f.reshape(x, x.shape[1:])
    inner.ret(tick1, seed, wind, sq(pos1), sq(vel1), sq(F_last), inertia_in, sq(A4[3]), dt)
    main = Fn("main", [(a.shape, a.dtype) for a in inner.args], public=True)
    main.ret(*main.call(inner, main.args))
    slots = [("tick", [], True), ("seed", [], True), ("wind", [3], True), ("world_pos", [7], True), ("world_vel", [6], True), ("force", [6], True),
             ("inertia", [7], True), ("world_accel", [6], True), ("simulation_time_step", [], True)]
    return module([main, inner, threefry, closed]), slots


if __name__ == "__main__":
    import sys
    text, slots = three_body_world() if len(sys.argv) < 2 else independent_bodies_world(int(sys.argv[1]))
    sys.stdout.write(text)
