#!/usr/bin/env python3
"""Can ONE rollout's tick be split over several lanes or waves?  A model, run on the traced Falcon 9 campaign program.

Round 3's verdict asked for the tick of one rollout to be partitioned over L = 2 / 4 lanes (32,768 rollouts = 512 waves leave half
of the chip's 1,024 SIMDs without a wave).  Lanes of one wave execute ONE instruction stream, so different sub-graphs on
different lanes serialise (tools/ubench/lane_split.hip measures exactly that); what can shorten a tick is a split over WAVES
of one workgroup that exchange through LDS at barriers.  This script prices that on the real program, CPU only:

  1. per system: new DAG nodes (cost-weighted), depth, reads -> writes                      (the pipe's dependency chain)
  2. system-granular schedule onto P waves with a barrier + LDS publication per level       (best of 30 hill-climbs)
  3. node-granular: the whole tick as one SSA DAG (cadenced / looping systems atomic), work T1 and critical path Tinf,
     and an epoch list-scheduler onto P waves that charges every cross-wave value one LDS write + one LDS read and every
     epoch one barrier

Output is what profiles/r04_rollout_split_model.txt holds.   python tools/rollout_split_model.py"""
import collections
import heapq
import random
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.setrecursionlimit(200000)

from elodin_amd import dsl  # noqa: E402
from elodin_amd.dsl import Expr  # noqa: E402
from elodin_amd.models import falcon9 as f9  # noqa: E402

BARRIER, COMM = 30.0, 1.0          # issue slots per barrier / per LDS write or read of one value (a VALU op = 1)
_COST = {"sin": 4, "cos": 4, "exp": 4, "log": 4, "sqrt": 4, "div": 2, "atan2": 27, "pow": 8, "tan": 10, "acos": 20, "asin": 20, "interp": 30,
         "threefry": 90, "erfinv": 90}


def cost(e):
    return 0 if e.op in ("const", "leaf", "proj") else _COST.get(e.op, 1)


def walk(roots, into_loops=True):
    seen, stack = {}, list(roots)
    while stack:
        x = stack.pop()
        if id(x) in seen:
            continue
        seen[id(x)] = x
        stack.extend(x.args)
        if into_loops and x.op == "while":
            stack.append(x.value[1])
            stack.extend(x.value[2])
    return seen


def tree_cost(roots):
    """Cost of a system run as one block; bodies of branch_cond / while loops (a GPS fix on one tick in forty) at 1/40."""
    seen = walk(roots, into_loops=False)
    c = sum(cost(x) for x in seen.values())
    for w in (x for x in seen.values() if x.op == "while"):
        c += tree_cost([w.value[1], *w.value[2]]) / 40.0
    return c


def weighted(roots, scale=1.0):
    """{node id: cost} of what the roots need; nodes inside loop / branch_cond bodies at 1/40 (rare branches)."""
    seen = walk(roots, into_loops=False)
    out = {k: cost(v) * scale for k, v in seen.items()}
    for w in (x for x in seen.values() if x.op == "while"):
        for k, v in weighted([w.value[1], *w.value[2]], scale / 40.0).items():
            out.setdefault(k, v)
    return out


def program():
    cols = f9.initial_columns(f9.default_param_row()[None, :])
    return f9.build_program(origin=f9.pad_ecef(), algebraic_geodesy=True).trace({k: v.shape[1] for k, v in cols.items()})


# ---- 1 + 2: systems as tasks -----------------------------------------------------------------------------------------------

BODY = ["qi", "qj", "qk", "qw", "px", "py", "pz", "wx", "wy", "wz", "vx", "vy", "vz"]


def system_tasks(tp):
    tasks, seen_all = [], {}
    for s in tp.pre + ["six_dof"] + tp.post:
        if s == "six_dof":
            roots, name, every = list(tp.pipe.outputs), "six_dof(effectors + calc_accel + integrate)", 1
            writes = set(BODY) | {"aax", "aay", "aaz", "alx", "aly", "alz"}
            reads = dsl._leaves_of(roots) | set(BODY) | {"Ix", "Iy", "Iz", "mass"}
            extra = 220.0
        else:
            roots, name, every = [e for _, e in s.assign], s.name, s.every
            writes, reads, extra = set(s.written), dsl._leaves_of(roots), 0.0
        nodes = weighted(roots)
        new = {k: v for k, v in nodes.items() if k not in seen_all}
        seen_all.update(nodes)
        tasks.append(dict(name=name, every=every, cost=(sum(new.values()) + extra), reads=reads, writes=writes,
                          nodes=nodes, extra=extra, depth=None))
    return tasks


def schedule_systems(tasks, P, assign):
    n = len(tasks)
    level, last = [0] * n, [0] * P
    for j in range(n):
        lv = last[assign[j]]
        for i in range(j):
            if (tasks[i]["writes"] & tasks[j]["reads"]) or (tasks[i]["writes"] & tasks[j]["writes"]):
                lv = max(lv, level[i] + (1 if assign[i] != assign[j] else 0))
            if tasks[i]["reads"] & tasks[j]["writes"]:
                lv = max(lv, level[i])
        level[j] = last[assign[j]] = lv
    total, rows = 0.0, []
    for lv in range(max(level) + 1):
        loads = []
        for w in range(P):
            seen, c = set(), 0.0
            for j in range(n):
                if level[j] == lv and assign[j] == w:
                    new = {k: v for k, v in tasks[j]["nodes"].items() if k not in seen}
                    seen.update(new)
                    c += (sum(new.values()) + tasks[j]["extra"]) / tasks[j]["every"]
            loads.append(c)
        pub = sum(len(tasks[j]["writes"]) for j in range(n) if level[j] == lv)
        total += max(loads) + (BARRIER + 0.5 * COMM * pub if P > 1 else 0.0)
        rows.append((lv, loads, [f"{tasks[j]['name'].split('(')[0]}@{assign[j]}" for j in range(n) if level[j] == lv]))
    return total, rows


def best_system_schedule(tasks, P):
    random.seed(1)
    n, best = len(tasks), None
    for _ in range(30):
        a = [random.randrange(P) for _ in range(n)]
        cur, improved = schedule_systems(tasks, P, a)[0], True
        while improved:
            improved = False
            for j in range(n):
                for w in range(P):
                    if w != a[j]:
                        b = a[:j] + [w] + a[j + 1:]
                        v = schedule_systems(tasks, P, b)[0]
                        if v < cur - 1e-9:
                            a, cur, improved = b, v, True
        if best is None or cur < best[0]:
            best = (cur, a)
    return best


# ---- 3: the whole tick as one DAG -------------------------------------------------------------------------------------------

def tick_dag(tp):
    env, memo, macros = {}, {}, []

    def subst(e):
        r = memo.get(id(e))
        if r is None:
            if e.op == "leaf":
                r = env.get(e.name, e)
            elif not e.args:
                r = e
            else:
                a = tuple(subst(x) for x in e.args)
                r = e if all(x is y for x, y in zip(a, e.args)) else Expr(e.op, a, e.value, e.name)
            memo[id(e)] = r
        return r

    def macro(name, ins, targets, c, every):
        m = Expr("macro", tuple(dict.fromkeys(ins)), (name, len(macros)))
        macros.append(dict(node=m, name=name, cost=c, every=every))
        memo.clear()
        for k, t in enumerate(targets):
            env[t] = Expr("proj", (m,), (name, k))
    for s in tp.pre + ["six_dof"] + tp.post:
        if s == "six_dof":
            ins = [subst(dsl.leaf(n)) for n in BODY + ["Ix", "Iy", "Iz", "mass"]] + [subst(e) for e in tp.pipe.outputs]
            macro("six_dof", ins, BODY + ["aax", "aay", "aaz", "alx", "aly", "alz"], 220.0, 1)
            continue
        roots = [e for _, e in s.assign]
        atomic = s.every > 1 or any(x.op in ("while", "while_out", "wload") for x in walk(roots, False).values())
        if atomic:
            macro(s.name, [subst(dsl.leaf(n)) for n in sorted(dsl._leaves_of(roots))], [t for t, _ in s.assign], tree_cost(roots), s.every)
        else:
            new = [(t, subst(e)) for t, e in s.assign]
            memo.clear()
            for t, e in new:
                env[t] = e
    return list(env.values()), {id(m["node"]): m for m in macros}


def node_model(tp):
    outs, mac = tick_dag(tp)
    nodes = {k: v for k, v in walk(outs, False).items() if v.op not in ("const", "leaf")}

    def c(x):
        return mac[id(x)]["cost"] / mac[id(x)]["every"] if x.op == "macro" else cost(x)
    preds = {i: [a for a in x.args if a.op not in ("const", "leaf")] for i, x in nodes.items()}
    succs = collections.defaultdict(list)
    for i, x in nodes.items():
        for a in preds[i]:
            succs[id(a)].append(x)
    topo, seen = [], set()

    def dfs(x):
        if id(x) in seen:
            return
        seen.add(id(x))
        for a in preds[id(x)]:
            dfs(a)
        topo.append(x)
    for x in nodes.values():
        dfs(x)
    asap, bl = {}, {}
    for x in topo:
        asap[id(x)] = max([asap[id(a)] for a in preds[id(x)]] or [0]) + c(x)
    for x in reversed(topo):
        bl[id(x)] = c(x) + max([bl[id(s)] for s in succs[id(x)]] or [0])
    work, tinf = sum(c(x) for x in nodes.values()), max(asap.values())
    outset = {id(o) for o in outs}

    def schedule(P, Q):
        where, done, unsched, epochs = {}, set(), set(nodes), []
        e = 0
        while unsched:
            load, this, taken = [0.0] * P, {}, set()
            heap = [(-bl[i], nodes[i].seq, i) for i in unsched if all(id(a) in done for a in preds[i])]
            heapq.heapify(heap)
            local = [[] for _ in range(P)]
            while True:
                w = min(range(P), key=lambda k: load[k])
                if load[w] >= Q:
                    break
                pick = None
                while local[w] and pick is None:
                    i = heapq.heappop(local[w])[2]
                    pick = i if i not in taken else None
                if pick is None:
                    tmp = []
                    while heap and len(tmp) < 24:
                        it = heapq.heappop(heap)
                        if it[2] not in taken:
                            tmp.append(it)
                    best = max(tmp, key=lambda it: (sum(1 for a in preds[it[2]] if where.get(id(a), (None,))[0] == w), -it[0]), default=None)
                    for it in tmp:
                        if it is not best:
                            heapq.heappush(heap, it)
                    pick = best[2] if best else None
                if pick is None:
                    others = [k for k in range(P) if k != w and load[k] < Q and local[k]]
                    if not others:
                        break
                    w = min(others, key=lambda k: load[k])
                    while local[w] and pick is None:
                        i = heapq.heappop(local[w])[2]
                        pick = i if i not in taken else None
                    if pick is None:
                        continue
                taken.add(pick)
                this[pick] = w
                where[pick] = (w, e)
                load[w] += c(nodes[pick])
                for s in succs[pick]:
                    si = id(s)
                    if si not in taken and si in unsched and all((id(a) in done) or this.get(id(a)) == w for a in preds[si]):
                        heapq.heappush(local[w], (-bl[si], s.seq, si))
            pub = [0] * P
            for i, w in this.items():
                if i in outset or any(id(s) not in this for s in succs[i]):
                    pub[w] += 1
            unsched -= set(this)
            done |= set(this)
            epochs.append((load, pub))
            e += 1
        rd = [[0] * P for _ in epochs]
        for i, (w, ep) in where.items():
            for k in {where[id(s)][0] for s in succs[i] if where[id(s)][0] != w}:
                rd[ep][k] += 1
        return sum(max(ld + COMM * (p + q) for ld, p, q in zip(load, pub, r)) + BARRIER for (load, pub), r in zip(epochs, rd)), len(epochs)
    return work, tinf, len(nodes), schedule


def main():
    tp = program()
    tasks = system_tasks(tp)
    print("Falcon 9 ascent campaign program (models/falcon9.py, pad-relative f32 build): one tick, cost in VALU issue slots")
    print(f"{'system':46s} {'every':>5s} {'new cost':>9s}  reads -> writes (columns)")
    for t in tasks:
        rd = sorted({n.split('_')[0] for n in t["reads"]})
        wr = sorted({n.split('_')[0] for n in t["writes"]})
        print(f"{t['name']:46s} {t['every']:5d} {t['cost']:9.0f}  {','.join(rd)[:60]} -> {','.join(wr)[:40]}")
    serial = schedule_systems(tasks, 1, [0] * len(tasks))[0]
    print(f"\n[systems as tasks]  one wave: {serial:.0f} slots per tick (cadence-weighted)")
    for P in (2, 3, 4):
        total, a = best_system_schedule(tasks, P)
        print(f"  P = {P} waves: {total:.0f} slots -> {serial / total:.2f}x   (barrier {BARRIER:.0f} + {0.5 * COMM} per published value per level)")
        for lv, loads, names in schedule_systems(tasks, P, a)[1]:
            print(f"     level {lv}: loads {[round(x) for x in loads]}  {' '.join(names)}")
    work, tinf, n_nodes, schedule = node_model(tp)
    print(f"\n[whole tick as one DAG]  {n_nodes} nodes (cadenced / looping systems and the integrator atomic): work T1 = {work:.0f}, "
          f"critical path Tinf = {tinf:.0f}, T1 / Tinf = {work / tinf:.1f}")
    for P in (2, 3, 4):
        best = min(((schedule(P, Q), Q) for Q in (60, 100, 150, 200, 300, 400, 600)), key=lambda r: r[0][0])
        (t, epochs), Q = best
        print(f"  P = {P} waves: {t:.0f} slots in {epochs} epochs (quantum {Q}) -> {work / t:.2f}x   (barrier {BARRIER:.0f}, LDS write / read {COMM} each)")


if __name__ == "__main__":
    main()
