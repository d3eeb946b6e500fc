"""Randomly generated programs through the tracer + code generator + hipcc + the fused step kernel, against the numpy
interpreter evaluating the SAME traced DAG (tests/dsl_numpy.py) AND against an independent plain-numpy twin of each program
drawn from the same seed (tests/fuzz_gen.py: pins the tracer itself, tests/test_fuzz_twin.py shows it catches seeded tracer bugs).  The generator is seeded, so the set of programs is
fixed; it mixes every scalar op family the front-end offers, vector helpers, selects, cadenced systems and writes that
invalidate shared sub-expressions — the combinations a hand-written test would not think of."""
import numpy as np
import pytest

import elodin_amd as el
from elodin_amd import _lib as L
from elodin_amd import dsl, workloads
from tests import dsl_numpy

pytestmark = pytest.mark.gpu
np_ = dsl.np


from tests.fuzz_gen import Gen, columns, make_program, twin_run  # noqa: E402,F401


def run_both(prog, cols, ticks, dtype):
    # reuse_trace=True everywhere in this file: a fuzz program's functions DRAW their expressions from a seeded generator while they
    # are traced, so a second trace of the same object is another program — the first one is the program under test
    n = len(cols["x"])
    w = workloads.independent_bodies(n)
    hip = el.HipExec(w["world_pos"], w["world_vel"], w["inertia"], dtype=dtype, integrator=L.INTEGRATOR_NONE, effectors=prog, reuse_trace=True,
                     columns={k: v.copy() for k, v in cols.items()})
    hip.run(ticks)
    tp = prog.trace({k: v.shape[1] for k, v in cols.items()})
    pos, vel, inertia = (np.array(w[k], dtype=np.float64) for k in ("world_pos", "world_vel", "inertia"))
    want = {k: v.copy() for k, v in cols.items()}
    for t in range(1, ticks + 1):
        dsl_numpy._run_systems(tp.pre, pos, vel, inertia, want, tp.table, t)
        dsl_numpy._run_systems(tp.post, pos, vel, inertia, want, tp.table, t)
    return {k: np.asarray(hip._aux[k], dtype=np.float64) for k in cols}, want


def check_against_the_twin(got, seed, cols, ticks, depth=4, tol=1e-10, frac=0.9995):
    """The generated kernel against the INDEPENDENT numpy evaluation of the same random program (tests/fuzz_gen.twin_run: no
    tracer involved) — what pins elodin_amd/dsl.py itself; the walker comparison next to it pins the code generator."""
    twin = twin_run(seed, cols, ticks, depth)
    for k in ("a", "b", "c", "x"):
        err = np.abs(got[k] - twin[k]) / np.maximum(np.abs(twin[k]), 1.0)
        assert (err < tol).mean() > frac, ("twin", seed, k, float(err.max()), float((err >= tol).mean()))


@pytest.mark.parametrize("seed", range(6))
def test_random_programs_f64(seed):
    prog, cols = make_program(seed), columns(seed, 2048)
    got, want = run_both(prog, cols, 2, np.float64)
    for k in ("a", "b", "c", "x"):
        assert np.isfinite(want[k]).all(), k
        err = np.abs(got[k] - want[k]) / np.maximum(np.abs(want[k]), 1.0)
        # a select whose two sides differ may flip where its comparison sits on a rounding boundary (FMA contraction,
        # 1-ulp libm differences): allow a handful of such lanes, none of them may be an outright blow-up of the others
        assert (err < 1e-10).mean() > 0.9995, (seed, k, float(err.max()), float((err >= 1e-10).mean()))
        assert np.median(err) < 1e-14, (seed, k)
    assert np.abs(want["c"]).max() > 0.0 and not np.array_equal(want["x"], cols["x"])
    check_against_the_twin(got, seed, cols, 2)


@pytest.mark.parametrize("seed", [0, 3, 5])
def test_random_programs_with_guarded_selects_f64(seed, monkeypatch):
    """The same random programs generated with guarded selects (codegen: expensive `where` arms behind a wave-level branch, with
    _GUARD_MIN_COST lowered so that these small DAGs qualify): a select guarded or not is the same select."""
    from elodin_amd import codegen
    monkeypatch.setenv("SIXDOF_GUARD_SELECTS", "1")
    monkeypatch.setattr(codegen, "_GUARD_MIN_COST", 6)
    prog, cols = make_program(seed), columns(seed, 2048)
    src = codegen.generate_source(prog.trace({k: v.shape[1] for k, v in cols.items()}), "float64", 2)
    assert "if (__any(" in src, "no select of this program qualified: the test would prove nothing"
    got, want = run_both(make_program(seed), cols, 2, np.float64)
    for k in ("a", "b", "c", "x"):
        err = np.abs(got[k] - want[k]) / np.maximum(np.abs(want[k]), 1.0)
        assert (err < 1e-10).mean() > 0.9995 and np.median(err) < 1e-14, (seed, k, float(err.max()))


def test_random_program_f32():
    prog, cols = make_program(101, depth=3), columns(101, 2048)
    got, want = run_both(prog, cols, 1, np.float32)
    for k in ("a", "b", "x"):
        err = np.abs(got[k] - want[k]) / np.maximum(np.abs(want[k]), 1.0)
        assert (err < 2e-3).mean() > 0.995 and np.median(err) < 5e-6, (k, float(err.max()), float(np.median(err)))
    check_against_the_twin(got, 101, cols, 1, depth=3, tol=2e-3, frac=0.995)


@pytest.mark.parametrize("seed", range(3))
def test_random_effector_pipes_through_the_integrator(seed):
    """Random wrenches (world torque, body-frame torque, force) out of pose / velocity / inertia / a component column,
    integrated semi-implicitly for a few ticks: state vs tests/dsl_numpy.program_tick on the same trace."""
    g = Gen(500 + seed)

    @dsl.effector(k=3)
    def first(force, pos, vel, inertia, k):
        leaves = list(pos.linear().e) + list(vel.linear().e) + list(vel.angular().e) + list(k.e) + [inertia.mass()]
        return force + dsl.SpatialForce(linear=dsl.Vec([g.scalar(leaves, 3) for _ in range(3)]),
                                        torque=dsl.Vec([g.scalar(leaves, 2) * 0.1 for _ in range(3)]))

    @dsl.effector(k=3)
    def second(force, pos, k):
        leaves = list(k.e) + list(force.force().e)
        return force + dsl.SpatialForce(torque=pos.angular() @ dsl.Vec([g.scalar(leaves, 2) * 0.05 for _ in range(3)]))
    n, dt = 1024, 1.0 / 120.0
    w = workloads.independent_bodies(n, seed=seed)
    kcol = np.random.default_rng(seed).uniform(-1.0, 1.0, (n, 3))
    prog = dsl.Program([], first | second, [])
    hip = el.HipExec(w["world_pos"], w["world_vel"], w["inertia"], simulation_time_step=dt, integrator=L.SEMI_IMPLICIT,
                     effectors=prog, reuse_trace=True, columns={"k": kcol})
    hip.run(5)
    tp = prog.trace({"k": 3})
    pos, vel, inertia = (np.array(w[k], dtype=np.float64) for k in ("world_pos", "world_vel", "inertia"))
    acc, comps = np.zeros((n, 6)), {"k": kcol.copy()}
    for t in range(1, 6):
        dsl_numpy.program_tick(tp, pos, vel, acc, inertia, comps, t, dt, L.SEMI_IMPLICIT)
    for name, got, want in (("world_pos", hip.world_pos, pos), ("world_vel", hip.world_vel, vel), ("world_accel", hip.world_accel, acc)):
        err = np.abs(got - want) / np.maximum(np.abs(want), 1.0)
        assert (err < 1e-9).mean() > 0.999 and np.median(err) < 1e-14, (seed, name, float(err.max()))


@pytest.mark.parametrize("seed", range(3))
def test_random_edge_folds(seed):
    """Random pair functions (force and world torque out of both endpoints' positions and masses, plus a smooth
    feedback of the accumulator) as the PAIR functor of the generated pair kernels, over a sparse random graph in both launch shapes, one
    RK4 tick: Force against tests/dsl_numpy.fold_force walking the same DAG edge by edge in spawn order."""
    from tests import np_sixdof
    g = Gen(900 + seed)

    @dsl.edge_fold(edge_component="fuzz")
    def fold(acc, a_pos, a_inertia, b_pos, b_inertia):
        r = b_pos.linear() - a_pos.linear()
        # the accumulator feeds back smoothly only: a staircase or remainder of a value that is itself folded over ~8
        # edges amplifies rounding differences into different branches, which says nothing about the generated code
        # (and the generator's inputs are squashed to O(1): separations and masses of the workload are not)
        # with x / (1 + |x|), which unlike tanh does not round to exactly 1 for the workload's heavier bodies
        squash = lambda v: v / (1.0 + np_.abs(v))
        leaves = [squash(e * 0.3) for e in r.e] + [squash(a_inertia.mass() * 0.1), squash(b_inertia.mass() * 0.1)]
        return acc + dsl.SpatialForce(linear=dsl.Vec([g.scalar(leaves, 3) * 0.2 + 0.05 * np_.tanh(acc.force()[k]) for k in range(3)]),
                                      torque=dsl.Vec([g.scalar(leaves[:3] + leaves[4:], 2) * 0.05 for _ in range(3)]))
    tf = fold.trace()
    for n in (150, 2000):
        w = workloads.independent_bodies(n, seed=seed)
        rng = np.random.default_rng(50 + seed)
        m = 4 * n
        src = rng.integers(0, n, size=m)
        dst = (src + 1 + rng.integers(0, n - 1, size=m)) % n
        ids = np.arange(1, n + 1, dtype=np.uint64)
        hip = el.HipExec(w["world_pos"], w["world_vel"], w["inertia"], integrator=L.RK4, simulation_time_step=1 / 240.0,
                         effectors=[fold], edges=(ids[src], ids[dst]))
        hip.run(1)
        pos, vel, acc, F = np_sixdof.tick(w["world_pos"].copy(), w["world_vel"].copy(), np.zeros((n, 6)), w["inertia"],
                                          lambda xs, vs: dsl_numpy.fold_force(tf, xs, w["inertia"], src, dst, np.zeros((n, 6))),
                                          1 / 240.0, integrator=L.RK4)
        for name, got, want in (("force", hip.force, F), ("world_vel", hip.world_vel, vel), ("world_pos", hip.world_pos, pos)):
            err = np.abs(got - want) / np.maximum(np.abs(want), 1.0)
            # a staircase sitting on a rounding boundary may flip for an isolated row
            assert (err >= 1e-9).sum() <= max(3, err.size // 1000) and np.median(err) < 1e-13, (seed, n, name, float(err.max()))


def test_random_program_f32_fast_math():
    """The opt-in hardware-transcendental mode on a random program: same structure, the tolerance of v_sin / v_exp / v_log
    / v_rcp (1e-6-ish absolute on O(1) values, amplified by the expression depth)."""
    prog, cols = make_program(202, depth=3), columns(202, 2048)
    n = len(cols["x"])
    w = workloads.independent_bodies(n)
    hip = el.HipExec(w["world_pos"], w["world_vel"], w["inertia"], dtype=np.float32, integrator=L.INTEGRATOR_NONE, effectors=prog, reuse_trace=True,
                     columns={k: v.copy() for k, v in cols.items()}, fast_math=True)
    hip.run(1)
    _, want = run_both(prog, cols, 1, np.float32)
    for k in ("a", "b", "x"):
        got = np.asarray(hip._aux[k], dtype=np.float64)
        err = np.abs(got - want[k]) / np.maximum(np.abs(want[k]), 1.0)
        assert (err < 5e-3).mean() > 0.99 and np.median(err) < 2e-5, (k, float(err.max()), float(np.median(err)))


@pytest.mark.parametrize("seed", range(64))
def test_random_configurations_of_the_handwritten_path_vs_the_oracle(seed):
    """The library kernels under seeded random CONFIGURATIONS: body count (around the wave / tile boundaries), integrator,
    ticks per launch not dividing the horizon, time-step override, a random pipe of built-in effectors with their aux
    columns, optionally closed by a pair op over a random sparse graph / the complete graph, graph replay, the async
    step flag and the telemetry ring — full state against the C oracle at the north-star tolerance."""
    from oracle import oracle as orc
    from tests import parity
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.choice([1, 2, 7, 63, 64, 65, 130, 257, 1000, 4097, 20000]))
    integrator = int(rng.choice([L.RK4, L.SEMI_IMPLICIT]))
    k = int(rng.choice([1, 2, 3, 5, 16, 32]))
    ticks = int(rng.integers(1, 41))
    dt = float(rng.choice([1 / 120.0, 1 / 60.0, 0.01]))
    time_step = None if rng.random() < 0.7 else dt * float(rng.choice([0.5, 2.0]))
    w = workloads.independent_bodies(n, seed=100 + seed)
    w["world_pos"][:, 4:] *= 0.01                                     # bring bodies within reach of each other for pair ops
    ops = []
    for kind in rng.permutation([L.EFF_UNIFORM_GRAVITY, L.EFF_CONST_WRENCH, L.EFF_BODY_TORQUE, L.EFF_BODY_FORCE, L.EFF_BALL_DRAG])[:int(rng.integers(0, 5))]:
        if kind == L.EFF_UNIFORM_GRAVITY:
            ops.append(el.Effector(kind, tuple(rng.normal(size=3) * 5.0)))
        elif kind == L.EFF_CONST_WRENCH:
            ops.append(el.Effector(kind, tuple(rng.normal(size=6))))
        elif kind == L.EFF_BALL_DRAG:
            ops.append(el.Effector(kind, (0.5, 1.2, 0.3), aux_name="wind", aux=rng.normal(size=(n, 3)) * 3.0))
        else:
            name = "tq" if kind == L.EFF_BODY_TORQUE else "thrust"
            ops.append(el.Effector(kind, (), aux_name=name, aux=rng.normal(size=(n, 3))))
    edges = oracle_edges = None
    pair = rng.choice(["none", "newton", "softened", "allpairs"]) if n >= 2 else "none"
    if pair == "allpairs" and n > 1100:
        pair = "softened"
    if pair in ("newton", "softened"):
        m = int(min(6 * n, 20000))
        src = rng.integers(0, n, size=m)
        dst = (src + 1 + rng.integers(0, n - 1, size=m)) % n
        ids = w["entity_ids"]
        edges = (ids[src], ids[dst])
        oracle_edges = orc.resolve_edges(ids, *edges)
        ops.append(el.Effector(L.EFF_EDGE_GRAVITY_NEWTON, (1e-3,)) if pair == "newton"
                   else el.Effector(L.EFF_EDGE_GRAVITY_SOFTENED, (1e-3, 1e-2)))
    elif pair == "allpairs":
        ops.append(el.Effector(L.EFF_ALLPAIRS_GRAVITY_SOFTENED, (1e-3, 1e-2)))
        frm = np.repeat(w["entity_ids"], n - 1)
        to = np.concatenate([np.delete(w["entity_ids"], i) for i in range(n)])
        oracle_edges = orc.resolve_edges(w["entity_ids"], frm, to)
    use_graph, async_step, ring = bool(rng.random() < 0.4), bool(rng.random() < 0.4), bool(rng.random() < 0.4)
    label = dict(n=n, integrator=integrator, k=k, ticks=ticks, dt=dt, time_step=time_step, ops=[o.kind for o in ops], pair=str(pair),
                 graph=use_graph, async_step=async_step, ring=ring)
    hip = el.HipExec(w["world_pos"], w["world_vel"], w["inertia"], entity_ids=w["entity_ids"], simulation_time_step=dt, time_step=time_step,
                     integrator=integrator, effectors=ops, edges=edges, ticks_per_launch=k, use_graph=use_graph)
    if ring and pair == "none":
        assert hip._lib.sixdof_set_history(hip._h, ticks) == L.OK
    if async_step:
        hip.set_flags(L.FLAG_ASYNC_STEP)
    first = ticks // 2
    hip.run(first)
    hip.run(ticks - first)
    oracle_ops = [(orc.EFF_EDGE_GRAVITY_SOFTENED, o.p, None) if o.kind == L.EFF_ALLPAIRS_GRAVITY_SOFTENED else (o.kind, tuple(o.p), o.aux) for o in ops]
    ref = orc.OracleWorld(w["world_pos"], w["world_vel"], w["inertia"], simulation_time_step=dt, time_step=time_step, integrator=integrator,
                          ops=oracle_ops, edges=oracle_edges)
    mid = None
    if ring and pair == "none" and ticks >= 2:
        ref.step(ticks - 1, threads=4)
        mid = ref.world_pos.copy()
        ref.step(1, threads=4)
    else:
        ref.step(ticks, threads=4)
    errs = parity.state_errors(hip, ref)
    assert max(errs.values()) < parity.F64_RTOL and hip.tick == ref.tick == ticks, (label, errs)
    if mid is not None:
        got = hip.history("world_pos", ticks - 1, ticks)
        assert parity.pos_rel_err(got[0], mid) < parity.F64_RTOL and parity.pos_rel_err(got[1], ref.world_pos) < parity.F64_RTOL, label


@pytest.mark.parametrize("seed", range(24))
def test_random_entity_set_joins_vs_the_oracle(seed):
    """Every Body column and every effector column on its OWN random entity set, some with rows out of id order (components
    inserted late): six_dof runs on the ascending-id intersection (query.rs:136-208).  Integer surface: the gather tables
    equal the positions of the joined ids inside each column.  Float surface: joined rows against the oracle stepped on
    the gathered world, every other row of every column untouched bit for bit."""
    from oracle import oracle as orc
    from tests import parity
    rng = np.random.default_rng(8100 + seed)
    universe = np.arange(1, int(rng.choice([5, 40, 64, 65, 300, 3000])) + 1, dtype=np.uint64)
    widths = {"world_pos": 7, "world_vel": 6, "inertia": 7, "world_accel": 6, "force": 6, "tq": 3, "wind": 3}
    ids, data = {}, {}
    for name, wd in widths.items():
        keep = universe[rng.random(len(universe)) < rng.uniform(0.8, 1.0)]
        if len(keep) == 0:
            keep = universe[:1]
        if rng.random() < 0.4:                                   # a few rows appended out of id order
            tail = rng.choice(len(keep), size=min(len(keep), 3), replace=False)
            keep = np.concatenate([np.delete(keep, tail), keep[tail]])
        ids[name] = keep
        a = rng.normal(size=(len(keep), wd))
        if name == "world_pos":
            a[:, :4] /= np.linalg.norm(a[:, :4], axis=1, keepdims=True)
        if name == "inertia":
            a = np.concatenate([rng.uniform(0.5, 3.0, (len(keep), 3)), np.zeros((len(keep), 3)), rng.uniform(1.0, 9.0, (len(keep), 1))], axis=1)
        data[name] = a
    joined = universe
    for name in ("world_pos", "world_vel", "inertia", "world_accel", "force"):
        joined = np.intersect1d(joined, ids[name])
    if len(joined) == 0:
        pytest.skip("empty join for this seed")
    for name in ("tq", "wind"):       # effector columns must cover the Body join (a superset in any order is fine)
        extra = np.setdiff1d(ids[name], joined)
        keep = np.concatenate([joined, extra])
        keep = keep[rng.permutation(len(keep))] if rng.random() < 0.5 else np.sort(keep)
        ids[name], data[name] = keep, rng.normal(size=(len(keep), 3))
    integrator = int(rng.choice([L.RK4, L.SEMI_IMPLICIT]))
    k, ticks = int(rng.choice([1, 4, 7])), int(rng.integers(1, 20))
    ops = [el.Effector(L.EFF_UNIFORM_GRAVITY, (0.0, 0.3, -9.81)), el.Effector(L.EFF_BODY_TORQUE, (), aux_name="tq", aux=data["tq"]),
           el.Effector(L.EFF_BALL_DRAG, (0.5, 1.2, 0.3), aux_name="wind", aux=data["wind"])]
    hip = el.HipExec(data["world_pos"], data["world_vel"], data["inertia"], world_accel=data["world_accel"], force=data["force"],
                     entity_ids=ids["world_pos"], integrator=integrator, effectors=ops, ticks_per_launch=k, column_entity_ids=ids)
    assert hip.n == len(joined)
    if len(joined) > 1:               # ... and one that does not is refused, loudly
        short = dict(ids, tq=ids["tq"][ids["tq"] != joined[0]])
        with pytest.raises(Exception, match="does not cover"):
            el.HipExec(data["world_pos"], data["world_vel"], data["inertia"], world_accel=data["world_accel"], force=data["force"],
                       entity_ids=ids["world_pos"], effectors=[el.Effector(L.EFF_BODY_TORQUE, (), aux_name="tq", aux=data["tq"][ids["tq"] != joined[0]])],
                       column_entity_ids=short).run(1)          # effector columns are joined when the pipe is first assembled
    rows = {}

    def check_rows(names):
        for name in names:
            rows[name] = hip.join_rows(name)
            where = {int(e): r for r, e in enumerate(ids[name])}
            assert rows[name].dtype == np.uint32 and rows[name].tolist() == [where[int(j)] for j in joined], name
    check_rows(("world_pos", "world_vel", "inertia", "world_accel", "force"))
    hip.run(ticks)
    check_rows(("tq", "wind"))                  # effector columns join when the pipe is first assembled
    g = {name: data[name][rows[name]] for name in widths}
    ref = orc.OracleWorld(g["world_pos"], g["world_vel"], g["inertia"], world_accel=g["world_accel"], force=g["force"], integrator=integrator,
                          ops=[(orc.EFF_UNIFORM_GRAVITY, (0.0, 0.3, -9.81), None), (orc.EFF_BODY_TORQUE, (), g["tq"]),
                               (orc.EFF_BALL_DRAG, (0.5, 1.2, 0.3), g["wind"])]).step(ticks)
    assert parity.pos_rel_err(hip.world_pos[rows["world_pos"]], ref.world_pos) < parity.F64_RTOL
    for name in ("world_vel", "world_accel", "force"):
        got, want = getattr(hip, name)[rows[name]], getattr(ref, name)
        assert max(parity.field_rel_err(got[:, :3], want[:, :3]), parity.field_rel_err(got[:, 3:], want[:, 3:])) < parity.F64_RTOL, name
    for name in ("world_pos", "world_vel", "world_accel", "force", "inertia"):
        others = np.setdiff1d(np.arange(len(ids[name])), rows[name])
        assert np.array_equal(getattr(hip, name)[others], data[name][others]), name      # not in the join: untouched
    assert np.array_equal(hip.inertia, data["inertia"])
