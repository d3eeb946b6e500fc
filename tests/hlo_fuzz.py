"""Seeded random ENTITY-PARALLEL StableHLO modules (the statements `jax.vmap` over a query produces, in random order and nesting) for
the differential test of elodin_amd.stablehlo's two whole-world evaluators: one lane per entity (_LaneEval: follows the entity axis
through every statement) against one lane per world (the plain evaluator, which knows nothing of entity axes).  Built with the
emitter of tests/golden/hlo_world_builder.py.  TEST INFRASTRUCTURE."""
import numpy as np

from tests.golden.hlo_world_builder import Fn, V, module


class _F(Fn):
    """The emitter plus the statements the fuzz uses that the world modules do not."""

    def un(self, op, a): return self._un(op, a)

    def maximum(self, a, b): return self._bin("maximum", a, b)

    def compare_select(self, a: V, b: V, x: V, y: V) -> V:
        c = self._new(a.shape, "i1")
        self.emit(f"{c.name} = stablehlo.compare  LT, {a.name}, {b.name},  FLOAT : ({a.ty}, {b.ty}) -> {c.ty}")
        r = self._new(x.shape, x.dtype)
        self.emit(f"{r.name} = stablehlo.select {c.name}, {x.name}, {y.name} : {c.ty}, {r.ty}")
        return r

    def iota(self, shape, dim, dtype="f64") -> V:
        r = self._new(shape, "i64")
        self.emit(f"{r.name} = stablehlo.iota dim = {dim} : {r.ty}")
        return self.convert(r, dtype) if dtype != "i64" else r

    def dot_general(self, a: V, b: V, batch, contract) -> V:
        (ba, bb), (ca, cb) = batch, contract
        fa = [d for d in range(len(a.shape)) if d not in ba + ca]
        fb = [d for d in range(len(b.shape)) if d not in bb + cb]
        r = self._new([a.shape[d] for d in ba] + [a.shape[d] for d in fa] + [b.shape[d] for d in fb], a.dtype)
        l = lambda xs: "[" + ", ".join(str(x) for x in xs) + "]"
        self.emit(f"{r.name} = stablehlo.dot_general {a.name}, {b.name}, batching_dims = {l(ba)} x {l(bb)}, contracting_dims = {l(ca)} x {l(cb)} : "
                  f"({a.ty}, {b.ty}) -> {r.ty}")
        return r

    def dynamic_update_slice(self, a: V, upd: V, starts) -> V:
        r = self._new(a.shape, a.dtype)
        self.emit(f"{r.name} = stablehlo.dynamic_update_slice {a.name}, {upd.name}, {', '.join(s.name for s in starts)} : "
                  f"({a.ty}, {upd.ty}, {', '.join(s.ty for s in starts)}) -> {r.ty}")
        return r

    def gather_table(self, table: V, idx: V) -> V:
        """table [R, w] (shared), idx [N] i64 -> [N, w]: a per-entity row of a table every entity shares."""
        n, w = idx.shape[0], table.shape[1]
        i2 = self.reshape(idx, (n, 1))
        r = self._new((n, w), table.dtype)
        self.emit(f'{r.name} = "stablehlo.gather"({table.name}, {i2.name}) <{{dimension_numbers = #stablehlo.gather<offset_dims = [1], collapsed_slice_dims = [0], '
                  f'start_index_map = [0], index_vector_dim = 1>, indices_are_sorted = false, slice_sizes = array<i64: 1, {w}>}}> : ({table.ty}, {i2.ty}) -> {r.ty}')
        return r


def make(seed: int, n: int, steps: int = 28, exchange: bool = False):
    """-> (module text, argument slots, result slots) of a random entity-parallel tick over n entities."""
    rng = np.random.default_rng(seed)
    f = _F("main", [((n, 4), "f64"), ((n, 3), "f64"), ((n, 2, 3), "f64"), ((), "f64"), ((n,), "i64")], public=True)
    a, b, c, k, idx = f.args
    pool = [a, b, c]                      # per-entity values: entity axis FIRST unless noted in `eax`
    eax = {id(a): 0, id(b): 0, id(c): 0}

    def pick(pred=lambda v: True):
        cand = [v for v in pool if pred(v)]
        return cand[int(rng.integers(len(cand)))] if cand else None

    joins = []

    def put(v, e=0):
        pool.append(v)
        eax[id(v)] = e
        return v
    for _ in range(steps):
        kind = int(rng.integers(15 if exchange else 13))
        x = pick(lambda v: eax[id(v)] == 0)
        if kind == 0:                                           # unary
            put(f.un(["sine", "tanh", "abs", "cosine"][int(rng.integers(4))], x))
        elif kind == 1:                                         # binary with a same-shaped partner (or a broadcast scalar / row)
            y = pick(lambda v: v.shape == x.shape and eax[id(v)] == 0 and v is not x)
            if y is None:
                y = f.bcast(k, x.shape, []) if rng.integers(2) else f.splat(float(rng.uniform(-1, 1)), x.shape)
            put([f.add, f.sub, f.mul, f.maximum][int(rng.integers(4))](x, y))
        elif kind == 2 and len(x.shape) == 2 and x.shape[1] > 1:       # slice along the trailing axis
            lo = int(rng.integers(x.shape[1] - 1))
            hi = int(rng.integers(lo + 1, x.shape[1] + 1))
            put(f.slice(x, [(0, n), (lo, hi)]))
        elif kind == 3 and len(x.shape) == 2:                   # concatenate along the trailing axis
            y = pick(lambda v: len(v.shape) == 2 and eax[id(v)] == 0)
            if x.shape[1] + y.shape[1] <= 12:
                put(f.concat([x, y], 1))
        elif kind == 4 and len(x.shape) == 2:                   # transpose away and back through an element-wise op
            t = f.transpose(x, [1, 0])
            t = f.un("tanh", t)
            put(f.transpose(t, [1, 0]))
        elif kind == 5:                                         # reshape keeping the entity axis whole
            if len(x.shape) == 3:
                put(f.reshape(x, (n, x.shape[1] * x.shape[2])))
            elif x.shape[1] % 2 == 0 and x.shape[1] >= 4:
                put(f.reshape(x, (n, 2, x.shape[1] // 2)))
            else:
                put(f.reshape(f.reshape(x, (n, x.shape[1], 1)), (n, x.shape[1])))
        elif kind == 6:                                         # reduce over a non-entity axis, then broadcast back along it
            if len(x.shape) >= 2:
                r = f.reduce_sum(x, [len(x.shape) - 1])
                if len(r.shape) == 1:
                    put(f.bcast(r, (n, 2), [0]))
                else:
                    put(r)
        elif kind == 7:                                         # batched matrix-vector: [n, 2, 3] . [n, 3] -> [n, 2]
            m = pick(lambda v: len(v.shape) == 3 and eax[id(v)] == 0)
            v3 = pick(lambda v: len(v.shape) == 2 and v.shape[1] == (m.shape[2] if m is not None else -1) and eax[id(v)] == 0)
            if m is not None and v3 is not None:
                put(f.dot_general(m, v3, ([0], [0]), ([2], [1])))
        elif kind == 8:                                         # select between two values by a comparison of two others
            y = pick(lambda v: v.shape == x.shape and eax[id(v)] == 0)
            put(f.compare_select(x, f.bcast(k, x.shape, []), x, y))
        elif kind == 9 and len(x.shape) == 2:                   # a counted while: accumulate a non-entity row-wise scaled copy
            def body(fb, i, carried):
                acc, src = carried
                scale = fb.bcast(fb.convert(i, "f64"), acc.shape, [])
                return [fb.add(acc, fb.mul(src, scale)), src]
            put(f.while_counted(int(rng.integers(2, 5)), [f.splat(0.0, x.shape), x], body)[0])
        elif kind == 10:                                        # a per-entity row of a shared table
            table = f.const([float(v) for v in rng.uniform(-2, 2, 5 * 3)], (15,), "f64")
            put(f.gather_table(f.reshape(table, (5, 3)), idx))
        elif kind == 11 and len(x.shape) == 2:                  # an index ramp along the trailing axis, broadcast over the entities
            ramp = f.iota((x.shape[1],), 0)
            put(f.mul(x, f.bcast(ramp, x.shape, [1])))
        elif kind == 12 and len(x.shape) == 2:                  # lax.scan's stacked outputs: rows written into a [T, n, w] buffer by the counter
            T = int(rng.integers(2, 4))
            w = x.shape[1]

            def body(fb, i, carried):
                buf, src = carried
                row = fb.mul(src, fb.bcast(fb.add(fb.convert(i, "f64"), fb.const(1.0)), src.shape, []))
                z = fb.const(0, (), "i64")
                return [fb.dynamic_update_slice(buf, fb.reshape(row, (1, n, w)), [i, z, z]), src]
            buf = f.while_counted(T, [f.splat(0.0, (T, n, w)), x], body)[0]
            pick_t = int(rng.integers(T))
            put(f.reshape(f.slice(buf, [(pick_t, pick_t + 1), (0, n), (0, w)]), (n, w)))
        elif kind == 13 and len(x.shape) == 2:                  # a join: every entity reads ANOTHER entity's row (constant table), k reads per entity
            e_ = int(rng.integers(1, 3))
            tgt = [[int(t) for t in rng.integers(0, n, e_)] for _ in range(n)]
            st = f.concat([f.reshape(f.gather_rows(x, tgt[s_]), (1, e_, x.shape[1])) for s_ in range(n)], 0)        # [n, e, w]
            joins.append(put(f.add(x, f.reduce_sum(f.transpose(st, [0, 2, 1]), [2]))))                            # own row + the sum of the rows read
        elif kind == 14 and len(x.shape) == 2:                  # an edge_fold's shape: a scan over the edge slot of the stacked target rows (graph.rs:187-235)
            e_, w = int(rng.integers(2, 4)), x.shape[1]
            tgt = [[int(t) for t in rng.integers(0, n, e_)] for _ in range(n)]
            st = f.concat([f.reshape(f.gather_rows(x, tgt[s_]), (1, e_, w)) for s_ in range(n)], 0)             # [n, e, w]
            st_t = f.transpose(st, [1, 0, 2])                                                                  # the scanned axis first

            def body(fb, i, carried):
                acc, src = carried
                z = fb.const(0, (), "i64")
                row = fb.reshape(fb.dynamic_slice(src, [i, z, z], (1, n, w)), (n, w))
                return [fb.add(fb.mul(acc, fb.splat(0.5, acc.shape)), row), src]
            joins.append(put(f.while_counted(e_, [x, st_t], body)[0]))
    if exchange and not joins:                                  # at least one join whose result is returned (below): the exchange is LIVE
        tgt = [[(s_ + 1) % n, int(rng.integers(0, n))] for s_ in range(n)]
        st = f.concat([f.reshape(f.gather_rows(a, tgt[s_]), (1, 2, 4)) for s_ in range(n)], 0)
        joins.append(put(f.add(a, f.reduce_sum(f.transpose(st, [0, 2, 1]), [2]))))
    outs = []
    for want in (lambda v: len(v.shape) == 2 and v.shape[1] <= 12, lambda v: True):
        v = None
        for cand in reversed(pool):
            if eax[id(cand)] == 0 and want(cand) and all(cand is not o for o in outs) and int(np.prod(cand.shape[1:])) <= 12:
                v = cand
                break
        outs.append(v if v is not None else a)
    if joins and all(joins[-1] is not o for o in outs):
        outs.append(joins[-1])
    f.ret(*outs)
    slots = [("a", [n, 4], False), ("b", [n, 3], False), ("c", [n, 2, 3], False), ("k", [], True), ("idx", [n], False)]
    out_slots = [(f"out{j}", list(o.shape), False) for j, o in enumerate(outs)]
    return module([f]), slots, out_slots


def inputs(seed: int, n: int):
    rng = np.random.default_rng(1000 + seed)
    return {"a": rng.uniform(-1, 1, (n, 4)), "b": rng.uniform(-1, 1, (n, 3)), "c": rng.uniform(-1, 1, (n, 2, 3)), "k": np.array(rng.uniform(-0.5, 0.5)),
            "idx": rng.integers(-1, 7, n).astype(float)}             # out-of-range rows included: gather clamps
