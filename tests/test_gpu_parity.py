"""GPU parity: the HIP path through the C ABI vs the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest

import elodin_amd as ea
from elodin_amd import _lib as L
from elodin_amd import workloads
from oracle import oracle as orc
from tests import golden_util as gu
from tests import parity

pytestmark = pytest.mark.gpu


def _pair(n, ticks, integrator=L.RK4, ticks_per_launch=1, seed=workloads.SEED, time_step=None, use_graph=False):
    w = workloads.independent_bodies(n, seed=seed)
    eff = workloads.gravity_torque_effectors(w["body_torque"])
    hip = ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], entity_ids=w["entity_ids"],
                     simulation_time_step=workloads.DT_120HZ, time_step=time_step, integrator=integrator,
                     effectors=eff, ticks_per_launch=ticks_per_launch, use_graph=use_graph)
    ref = orc.OracleWorld(w["world_pos"], w["world_vel"], w["inertia"], simulation_time_step=workloads.DT_120HZ,
                          time_step=time_step, integrator=integrator, ops=parity.to_oracle_ops(eff))
    return hip, ref, w


@pytest.mark.parametrize("n", [1, 3, 63, 64, 255, 256, 257, 1000, 4099])
def test_rk4_ragged_sizes(n):
    hip, ref, _ = _pair(n, 10)
    hip.run(10)
    ref.step(10, threads=4)
    errs = parity.state_errors(hip, ref)
    assert max(errs.values()) < parity.F64_RTOL, errs
    assert hip.tick == ref.tick == 10


def test_empty_world():
    hip = ea.HipExec(np.zeros((0, 7)), np.zeros((0, 6)), np.zeros((0, 7)))
    hip.run(5)
    assert hip.tick == 5 and hip.world_pos.shape == (0, 7)


@pytest.mark.parametrize("integrator", [L.RK4, L.SEMI_IMPLICIT])
def test_config2_65536_bodies_1000_ticks(integrator):
    """BASELINE config 2 at full size; oracle on 8 threads takes a few seconds."""
    hip, ref, w = _pair(65536, 1000, integrator=integrator, ticks_per_launch=1, use_graph=True)
    checkpoints = [1, 10, 50, 100, 200, 300, 400, 600, 800, 900, 1000]      # 10 checkpoints and the final tick
    done = 0
    worst = parity.Worst()
    for cp in checkpoints:
        hip.run(cp - done)
        ref.step(cp - done, threads=8)
        done = cp
        worst.update(hip, ref, cp)
    worst.check(f"config2 65,536 x 1,000 ticks, integrator {integrator}")
    # entity indices are carried bit-exactly (row i <-> entity_ids[i])
    assert np.array_equal(hip.entity_ids, w["entity_ids"])


@pytest.mark.parametrize("k", [1, 7, 16, 256])
def test_fused_ticks_match_single_tick_launches(k):
    """ticks_per_launch (the reference's ticks_per_telemetry batch) must not change the result."""
    a, _, _ = _pair(5000, 300, ticks_per_launch=1)
    b, _, _ = _pair(5000, 300, ticks_per_launch=k)
    a.run(300)
    b.run(300)
    for f in parity.FIELDS:
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    assert a.tick == b.tick == 300


@pytest.mark.parametrize("n,dtype,integrator", [(131_072, np.float64, L.RK4), (262_144, np.float64, L.RK4), (786_432, np.float64, L.RK4),
                                                (3_200_000, np.float64, L.RK4), (262_147, np.float64, L.SEMI_IMPLICIT),
                                                (524_288, np.float32, L.RK4), (6_400_001, np.float32, L.RK4)])
def test_one_tick_launches_match_fused_launches_in_every_cache_policy_range(n, dtype, integrator):
    """The launch's cache policy is picked by state size (sixdof_capi.cpp): a policy that lets the NEXT launch read stale rows
    shows as one-tick launches (eager and replayed) drifting from a fused run of the same ticks — which is how the
    write-through `sc1` store policy, briefly shipped for 48-192 MiB of state, was caught (262,144 bodies: 10 % of the rows
    wrong).  Sizes: one per range of the table + the old sc1 range, f64 and f32 (half the bytes per body), a ragged last wave."""
    w = workloads.independent_bodies(n)
    eff = workloads.gravity_torque_effectors(w["body_torque"])
    ticks = 64 if n < 1_000_000 else 24

    def run(**kw):
        ex = ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], simulation_time_step=workloads.DT_120HZ, effectors=eff,
                        dtype=dtype, integrator=integrator, **kw)
        if kw.get("use_graph"):
            ex.prepare(ticks)
        ex.run(ticks)
        out = [getattr(ex, f).copy() for f in parity.FIELDS]
        ex.close()
        return out
    fused = run(ticks_per_launch=ticks)
    for kw in (dict(ticks_per_launch=1), dict(ticks_per_launch=1, use_graph=True), dict(ticks_per_launch=8, use_graph=True)):
        for f, a, b in zip(parity.FIELDS, run(**kw), fused):
            assert np.array_equal(a, b), (kw, f, int((a != b).any(axis=1).sum()))


def test_graph_replay_is_identical():
    a, _, _ = _pair(3000, 100, use_graph=False)
    b, _, _ = _pair(3000, 100, use_graph=True)
    a.run(100)
    b.run(100)
    for f in parity.FIELDS:
        assert np.array_equal(getattr(a, f), getattr(b, f)), f


def test_determinism_across_runs():
    a, _, _ = _pair(10000, 50)
    b, _, _ = _pair(10000, 50)
    a.run(50)
    b.run(50)
    for f in parity.FIELDS:
        assert np.array_equal(getattr(a, f), getattr(b, f)), f


def test_time_step_override_quirk():
    """RK4: stage offsets use the global dt, the final combination the override (rk4.rs:93-100,129)."""
    hip, ref, _ = _pair(777, 20, time_step=1.0 / 60.0)
    hip.run(20)
    ref.step(20)
    assert max(parity.state_errors(hip, ref).values()) < parity.F64_RTOL


def test_ball_golden_on_gpu():
    """G2 (scripts/ci/baseline/ball-csv) straight through the HIP path: gravity | drag, RK4."""
    g = gu.load("ball")
    wind = g["ball.wind"][1]
    eff = [ea.Effector(L.EFF_UNIFORM_GRAVITY, (0.0, 0.0, -9.81)),
           ea.Effector(L.EFF_BALL_DRAG, (0.5, 1.225, 2 * 3.1415 * 0.2**2), aux_name="wind", aux=wind[None, :])]
    hip = ea.HipExec(g["ball.world_pos"][0], g["ball.world_vel"][0], g["ball.inertia"][0],
                     simulation_time_step=float(g["globals.simulation_time_step"][0, 0]), effectors=eff)
    worst = 0.0
    for r in range(1, 101):
        hip.run(1)
        for comp in parity.FIELDS:
            got, ref = getattr(hip, comp), g[f"ball.{comp}"][r][None, :]
            if comp == "world_pos":
                worst = max(worst, parity.pos_rel_err(got, ref))
            else:
                worst = max(worst, parity.field_rel_err(got[:, 3:], ref[:, 3:]))
                assert np.all(got[:, :3] == 0.0)
    print("ball golden on GPU worst rel err", worst)
    assert worst < parity.F64_RTOL


def test_kat_on_gpu():
    """K8 (libs/nox-py/python/tests/test_all.py:67-83,228-292,342-364) through the HIP path."""
    one = dict(world_pos=[[0, 0, 0, 1, 0, 0, 0]], inertia=[[1, 1, 1, 0, 0, 0, 1]],
               simulation_time_step=workloads.DT_120HZ)
    h = ea.HipExec(world_vel=[[0, 0, 0, 1, 0, 0]], time_step=1.0 / 60.0, **one)
    h.run(1)
    assert np.allclose(h.world_pos[0], [0, 0, 0, 1, 0.01666667, 0, 0])
    for omega, q in [([0, 0, 1], [0.0, 0.0, 0.479425538604203, 0.8775825618903728]),
                     ([0, 1, 0], [0.0, 0.479425538604203, 0.0, 0.8775825618903728]),
                     ([1, 1, 0], [0.45936268493243, 0.45936268493243, 0.0, 0.76024459707606])]:
        h = ea.HipExec(world_vel=[omega + [0, 0, 0]], time_step=1.0 / 120.0, **one)
        h.run(120)
        assert np.isclose(h.world_pos[0], q + [0, 0, 0], rtol=1e-5).all()
    h = ea.HipExec(world_vel=[[0] * 6], time_step=1.0 / 120.0,
                   effectors=[ea.Effector(L.EFF_CONST_WRENCH, (0, 0, 0, 1, 0, 0))], **one)
    h.run(120)
    assert np.isclose(h.world_pos[0], [0, 0, 0, 1, 0.5, 0, 0], rtol=1e-5).all()


def test_f32_extension_tracks_f64_oracle():
    """BASELINE config 5 dtype. Parity unpinned (reference six_dof is f64 only); f32-appropriate tolerance."""
    w = workloads.independent_bodies(4096)
    eff = workloads.gravity_torque_effectors(w["body_torque"])
    hip = ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], dtype=np.float32,
                     simulation_time_step=workloads.DT_120HZ, effectors=eff)
    ref = orc.OracleWorld(w["world_pos"], w["world_vel"], w["inertia"], simulation_time_step=workloads.DT_120HZ,
                          ops=parity.to_oracle_ops(eff))
    hip.run(20)
    ref.step(20, threads=4)
    errs = parity.state_errors(hip, ref)
    assert max(errs.values()) < 2e-4, errs


def test_f32_config2_full_size_1000_ticks_drift_bound():
    """The f32 instantiation at BASELINE configs[1] size over a long horizon: 65,536 bodies x 1,000 RK4 ticks against the
    f64 oracle.  The reference has no f32 six_dof (six_dof.rs:12-14), so this bound is this build's own statement of f32
    drift (measured on MI355X: attitude 8.9e-5, position 6.4e-5 of the vector's scale): attitudes within 3e-4, positions
    within 2e-4, velocity / acceleration / force within 2e-3."""
    import os
    w = workloads.independent_bodies(65536)
    eff = workloads.gravity_torque_effectors(w["body_torque"])
    hip = ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], dtype=np.float32, simulation_time_step=workloads.DT_120HZ,
                     effectors=eff, ticks_per_launch=50)
    ref = orc.OracleWorld(w["world_pos"], w["world_vel"], w["inertia"], simulation_time_step=workloads.DT_120HZ,
                          ops=parity.to_oracle_ops(eff))
    hip.run(1000)
    ref.step(1000, threads=len(os.sched_getaffinity(0)))
    q_err = parity.field_rel_err(hip.world_pos[:, :4], ref.world_pos[:, :4])
    p_err = parity.field_rel_err(hip.world_pos[:, 4:], ref.world_pos[:, 4:])
    errs = parity.state_errors(hip, ref)
    print(f"f32 config 2, 65,536 x 1,000 ticks: attitude {q_err:.2e}, position {p_err:.2e}, rest {errs}")
    assert q_err < 3e-4 and p_err < 2e-4, (q_err, p_err)
    assert errs["world_vel"] < 2e-3 and errs["world_accel"] < 2e-3 and errs["force"] < 2e-3, errs


def test_tickfn_shim_roundtrip():
    """sixdof_tick(inputs**, outputs**) — the reference's TickFn ABI (cranelift_exec.rs:11)."""
    import ctypes as C
    hip, ref, w = _pair(500, 1)
    lib = L.lib()
    n_in, n_out = C.c_size_t(), C.c_size_t()
    ins, outs = (L.Slot * 16)(), (L.Slot * 16)()
    assert lib.sixdof_tick_slots(hip._h, ins, 16, C.byref(n_in), outs, 16, C.byref(n_out)) == 0
    by_id = {L.component_id(k): v for k, v in
             dict(world_pos=hip.world_pos, world_vel=hip.world_vel, world_accel=hip.world_accel, force=hip.force,
                  inertia=hip.inertia, body_torque=hip._aux["body_torque"]).items()}
    tick = np.array([41], dtype=np.uint64)
    dt = np.array([workloads.DT_120HZ])
    by_id[L.component_id("tick")] = tick
    by_id[L.component_id("simulation_time_step")] = dt
    in_ptrs = (C.c_void_p * n_in.value)(*[by_id[ins[i].component_id].ctypes.data for i in range(n_in.value)])
    out_bufs = [np.zeros(outs[i].bytes, dtype=np.uint8) for i in range(n_out.value)]
    out_ptrs = (C.c_void_p * n_out.value)(*[b.ctypes.data for b in out_bufs])
    # outputs come back in ascending ComponentId (system.rs:139-153)
    ids = [outs[i].component_id for i in range(n_out.value)]
    assert ids == sorted(ids)
    lib.sixdof_tick_bind(hip._h)
    lib.sixdof_tick(in_ptrs, out_ptrs)
    ref.step(1)
    got = {ids[i]: out_bufs[i] for i in range(n_out.value)}
    assert got[L.component_id("tick")].view(np.uint64)[0] == 42
    pos = got[L.component_id("world_pos")].view(np.float64).reshape(-1, 7)
    assert parity.pos_rel_err(pos, ref.world_pos) < parity.F64_RTOL
    assert np.array_equal(got[L.component_id("inertia")].view(np.float64).reshape(-1, 7), hip.inertia)


# ---- pairwise (edge_fold) path ---------------------------------------------------------------------------

G_NEWTON = 6.6743e-11
K_SQ = 2.9591220828e-4 / (86400.0 * 86400.0)   # examples/n-body/sim.py:14-17
EPS_AU2 = 1.0e-10


def test_three_body_golden_on_gpu():
    """G1 (scripts/ci/baseline/three-body-csv) through the HIP edge_fold path, 100 ticks."""
    g = gu.load("three_body")
    names = "abc"
    pos = np.stack([g[f"{e}.world_pos"][0] for e in names])
    vel = np.stack([g[f"{e}.world_vel"][0] for e in names])
    inertia = np.stack([g[f"{e}.inertia"][0] for e in names])
    edge_names = ["a_>_b", "b_>_a", "a_>_c", "b_>_c", "c_>_a", "c_>_b"]
    frm = np.array([g[f"{e}.gravity_edge"][0, 0] for e in edge_names], dtype=np.uint64)
    to = np.array([g[f"{e}.gravity_edge"][0, 1] for e in edge_names], dtype=np.uint64)
    hip = ea.HipExec(pos, vel, inertia, entity_ids=[1, 2, 3],
                     simulation_time_step=float(g["globals.simulation_time_step"][0, 0]),
                     effectors=[ea.Effector(L.EFF_EDGE_GRAVITY_NEWTON, (G_NEWTON,))], edges=(frm, to))
    # integer parity: resolved row indices are bit-identical to the oracle's
    src, dst = hip.edge_rows()
    osrc, odst = orc.resolve_edges(np.array([1, 2, 3], dtype=np.uint64), frm, to)
    assert src.dtype == np.uint32 and np.array_equal(src, osrc) and np.array_equal(dst, odst)
    worst = {}
    for r in range(1, 101):
        hip.run(1)
        assert hip.tick == int(g["globals.tick"][r, 0])
        for i, e in enumerate(names):
            worst["world_pos"] = max(worst.get("world_pos", 0), parity.pos_rel_err(hip.world_pos[i:i + 1], g[f"{e}.world_pos"][r][None]))
            for comp in ("world_vel", "world_accel", "force"):
                got, ref = getattr(hip, comp)[i:i + 1, 3:], g[f"{e}.{comp}"][r][None, 3:]
                worst[comp] = max(worst.get(comp, 0), parity.field_rel_err(got, ref))
    print("three-body golden on GPU worst rel err", worst)
    assert max(worst.values()) < parity.F64_RTOL, worst


def _plummer(n, seed=7):
    rng = np.random.default_rng(seed)
    # Plummer sphere, a = 1 AU; masses U(1e-9, 1e-3) solar masses
    u = rng.uniform(0.05, 0.95, n)
    r = 1.0 / np.sqrt(u ** (-2.0 / 3.0) - 1.0)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    p = d * r[:, None]
    v = rng.normal(scale=1e-7, size=(n, 3))
    m = rng.uniform(1e-9, 1e-3, n)
    pos = np.concatenate([np.tile([0, 0, 0, 1.0], (n, 1)), p], axis=1)
    vel = np.concatenate([np.zeros((n, 3)), v], axis=1)
    inertia = np.concatenate([np.tile(m[:, None], (1, 3)), np.zeros((n, 3)), m[:, None]], axis=1)
    return pos, vel, inertia


@pytest.mark.parametrize("n,ticks", [(2, 5), (35, 10), (257, 5), (1000, 5), (2048, 3)])
@pytest.mark.parametrize("integrator", [L.RK4, L.SEMI_IMPLICIT])
def test_allpairs_nbody_vs_oracle(n, ticks, integrator):
    pos, vel, inertia = _plummer(n)
    op = (K_SQ, EPS_AU2)
    hip = ea.HipExec(pos, vel, inertia, simulation_time_step=3600.0, integrator=integrator,
                     effectors=[ea.Effector(L.EFF_ALLPAIRS_GRAVITY_SOFTENED, op)])
    ref = orc.OracleWorld(pos, vel, inertia, simulation_time_step=3600.0, integrator=integrator,
                          ops=[(orc.EFF_ALLPAIRS_GRAVITY_SOFTENED, op, None)])
    hip.run(ticks)
    ref.step(ticks)
    errs = parity.state_errors(hip, ref)
    assert max(errs.values()) < parity.F64_RTOL, errs


def test_edge_list_equals_allpairs():
    """The complete graph given as explicit edges (n-body spawn order) and the tiled all-pairs kernel agree."""
    n = 300
    pos, vel, inertia = _plummer(n, seed=3)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    frm = np.repeat(ids, n - 1)
    to = np.concatenate([np.delete(ids, i) for i in range(n)])
    op = (K_SQ, EPS_AU2)
    a = ea.HipExec(pos, vel, inertia, simulation_time_step=3600.0,
                   effectors=[ea.Effector(L.EFF_EDGE_GRAVITY_SOFTENED, op)], edges=(frm, to))
    b = ea.HipExec(pos, vel, inertia, simulation_time_step=3600.0,
                   effectors=[ea.Effector(L.EFF_ALLPAIRS_GRAVITY_SOFTENED, op)])
    ref = orc.OracleWorld(pos, vel, inertia, simulation_time_step=3600.0,
                          ops=[(orc.EFF_EDGE_GRAVITY_SOFTENED, op, None)],
                          edges=orc.resolve_edges(ids, frm, to))
    a.run(4); b.run(4); ref.step(4)
    assert max(parity.state_errors(a, ref).values()) < parity.F64_RTOL
    assert max(parity.state_errors(b, ref).values()) < parity.F64_RTOL


@pytest.mark.parametrize("n,integrator", [(65536, L.RK4), (65536 + 37, L.SEMI_IMPLICIT)])
def test_sparse_lattice_at_bench_size_in_one_launch_per_tick_vs_the_oracle(n, integrator):
    """The sparse workload bench.py's side leg times (a ring lattice, 16 out-edges per source in spawn order; here every fifth body
    is a target only: 838,848 directed edges at 65,536 bodies) at its full size, through the fused fold-and-integrate launch (pair_kernel.hpp 3b: pack rows
    double-buffered, ONE launch per tick), against the oracle's sequential edge fold; a row count that is not a whole number
    of waves under the other integrator; every fifth body carries no out-edges and keeps its per-entity gravity instead."""
    deg = 16
    rng = np.random.default_rng(11)
    pos = np.concatenate([np.tile([0, 0, 0, 1.0], (n, 1)), rng.normal(size=(n, 3)) * 1e3], axis=1)
    vel = np.concatenate([np.zeros((n, 3)), rng.normal(size=(n, 3))], axis=1)
    m = rng.uniform(1.0, 10.0, n)
    inertia = np.concatenate([np.tile(m[:, None], (1, 3)), np.zeros((n, 3)), m[:, None]], axis=1)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    offs = np.array([k for k in range(-deg // 2, deg // 2 + 1) if k != 0][:deg])
    src_rows = np.arange(n)[np.arange(n) % 5 != 0]                       # rows 0, 5, 10, ... are targets only
    frm = np.repeat(ids[src_rows], deg)
    to = ((np.repeat(src_rows, deg) + np.tile(offs, len(src_rows))) % n + 1).astype(np.uint64)
    G = 6.6743e-11
    eff = [ea.Effector(L.EFF_UNIFORM_GRAVITY, (0.0, 0.0, -9.81)), ea.Effector(L.EFF_EDGE_GRAVITY_NEWTON, (G,))]
    hip = ea.HipExec(pos, vel, inertia, entity_ids=ids, simulation_time_step=0.01, integrator=integrator, effectors=eff, edges=(frm, to))
    ref = orc.OracleWorld(pos, vel, inertia, simulation_time_step=0.01, integrator=integrator,
                          ops=[(orc.EFF_UNIFORM_GRAVITY, (0.0, 0.0, -9.81), None), (orc.EFF_EDGE_GRAVITY_NEWTON, (G,), None)],
                          edges=orc.resolve_edges(ids, frm, to))
    worst = parity.Worst()
    for cp in (1, 6, 12):
        t = hip.run(cp - hip.tick)
        ref.step(cp - ref.tick, threads=8)
        worst.update(hip, ref, cp)
    assert t.launches == 1 + 6                                       # the last batch: one pack, then one launch per tick
    worst.check(f"sparse lattice {n} bodies x {len(frm)} edges, integrator {integrator}")
    assert np.abs(hip.force[::5, 5] + 9.81 * m[::5]).max() < 1e-9 * 98.1    # target-only rows: the per-entity gravity survived
    e_src, e_dst = hip.edge_rows()
    assert np.array_equal(e_src, np.repeat(src_rows, deg).astype(np.uint32)) and np.array_equal(e_dst, (to - 1).astype(np.uint32))      # edge rows bit-exact, spawn order


@pytest.mark.parametrize("kind,integrator", [("newton", L.RK4), ("softened", L.RK4), ("softened", L.SEMI_IMPLICIT)])
def test_hub_sources_are_folded_by_whole_waves_and_match_the_sequential_fold(kind, integrator):
    """A graph with hubs (the reference buckets sources by out-degree, graph.rs:290-328): two sources with thousands of
    out-edges (a dozen or more 256-edge chunks each), a band of sources around the 32-edge hub threshold, leaves with none.  Hubs
    are folded by whole waves + a fixed-order reduction (pair_kernel.hpp 2c); the oracle folds every source sequentially."""
    n = 5000
    rng = np.random.default_rng(5)
    pos, vel, inertia = _plummer(n, seed=9)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    frm, to = [], []
    for src, deg in [(7, n - 1), (4321, 3000)] + [(int(s), int(d)) for s, d in zip(rng.choice(np.arange(100, 4000), 300, replace=False), rng.integers(1, 70, 300))]:
        targets = rng.permutation(np.delete(np.arange(n), src))[:deg]
        frm += [src + 1] * deg
        to += list(targets + 1)
    order = rng.permutation(len(frm))                      # spawn order interleaves the sources; CSR keeps it per source
    frm, to = np.array(frm, dtype=np.uint64)[order], np.array(to, dtype=np.uint64)[order]
    if kind == "newton":
        op, hk, ok = (K_SQ,), L.EFF_EDGE_GRAVITY_NEWTON, orc.EFF_EDGE_GRAVITY_NEWTON
    else:
        op, hk, ok = (K_SQ, EPS_AU2), L.EFF_EDGE_GRAVITY_SOFTENED, orc.EFF_EDGE_GRAVITY_SOFTENED
    hip = ea.HipExec(pos, vel, inertia, simulation_time_step=3600.0, integrator=integrator, effectors=[ea.Effector(hk, op)], edges=(frm, to))
    ref = orc.OracleWorld(pos, vel, inertia, simulation_time_step=3600.0, integrator=integrator, ops=[(ok, op, None)],
                          edges=orc.resolve_edges(ids, frm, to))
    hip.run(5); ref.step(5)
    errs = parity.state_errors(hip, ref)
    assert max(errs.values()) < parity.F64_RTOL, errs
    assert hip.last_timings().launches == 1 + 5 * 3        # one pack for the batch, then per tick: hub chunks, hub reduce, the fused fold-and-integrate launch


def test_edge_fold_replaces_force_only_on_source_rows():
    """Rows without out-edges keep the per-entity effectors' Force (graph.rs:239-361)."""
    n = 6
    pos, vel, inertia = _plummer(n, seed=11)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    frm = np.array([1, 1, 3, 4, 4, 4], dtype=np.uint64)   # sources: rows 0, 2, 3 ; rows 1, 4, 5 are not
    to = np.array([2, 3, 1, 6, 5, 1], dtype=np.uint64)
    op = (K_SQ, EPS_AU2)
    eff = [ea.Effector(L.EFF_UNIFORM_GRAVITY, (0.0, 0.0, -1e-12)), ea.Effector(L.EFF_EDGE_GRAVITY_SOFTENED, op)]
    hip = ea.HipExec(pos, vel, inertia, simulation_time_step=3600.0, effectors=eff, edges=(frm, to))
    ref = orc.OracleWorld(pos, vel, inertia, simulation_time_step=3600.0,
                          ops=[(orc.EFF_UNIFORM_GRAVITY, (0.0, 0.0, -1e-12), None),
                               (orc.EFF_EDGE_GRAVITY_SOFTENED, op, None)],
                          edges=orc.resolve_edges(ids, frm, to))
    hip.run(7); ref.step(7)
    assert max(parity.state_errors(hip, ref).values()) < parity.F64_RTOL
    assert np.allclose(hip.force[1, 3:], [0, 0, -1e-12 * inertia[1, 6]], rtol=1e-12)


def test_pair_op_errors():
    pos, vel, inertia = _plummer(4)
    with pytest.raises(KeyError):  # Error::ComponentNotFound: edge effector without edges
        h = ea.HipExec(pos, vel, inertia, effectors=[ea.Effector(L.EFF_EDGE_GRAVITY_NEWTON, (1.0,))])
        h.run(1)
    with pytest.raises(KeyError):  # edge endpoint that is not a Body
        ea.HipExec(pos, vel, inertia, effectors=[ea.Effector(L.EFF_EDGE_GRAVITY_NEWTON, (1.0,))],
                   edges=(np.array([1], dtype=np.uint64), np.array([99], dtype=np.uint64)))
    with pytest.raises(ValueError):  # pair op must be last in the pipe
        ea.HipExec(pos, vel, inertia, effectors=[ea.Effector(L.EFF_ALLPAIRS_GRAVITY_SOFTENED, (1.0, 0.0)),
                                                 ea.Effector(L.EFF_UNIFORM_GRAVITY, (0, 0, -1))])


# ---- telemetry ring -----------------------------------------------------------------------------------------

@pytest.mark.parametrize("n,k", [(1000, 16), (4099, 7), (64, 1)])
def test_history_ring_records_every_tick_of_fused_launches(n, k):
    """With a ring, ticks_per_launch > 1 keeps every intermediate tick: history == stepping one tick at a time."""
    fused, _, _ = _pair(n, 50, ticks_per_launch=k)
    single, _, _ = _pair(n, 50, ticks_per_launch=1)
    fused.enable_history(64)
    fused.run(50)
    hist = {f: fused.history(f, 1, 50) for f in parity.FIELDS}
    for tick in range(1, 51):
        single.run(1)
        for f in parity.FIELDS:
            assert np.array_equal(hist[f][tick - 1], getattr(single, f)), (f, tick)
    # the live columns hold the last tick as usual
    for f in parity.FIELDS:
        assert np.array_equal(getattr(fused, f), hist[f][-1])


def test_history_ring_window_and_errors():
    hip, _, _ = _pair(300, 1, ticks_per_launch=8)
    with pytest.raises(ValueError):          # no ring yet
        hip.history("world_pos", 1, 1)
    hip.run(5)
    hip.enable_history(10)                   # recording starts at tick 6
    hip.run(25)                              # ticks 6..30 recorded, ring keeps 21..30
    assert hip.tick == 30
    hip.history("world_vel", 21, 30)
    for bad in (20, 31, 5):
        with pytest.raises(ValueError):
            hip.history("world_vel", bad, bad)
    with pytest.raises(KeyError):
        hip.history("inertia", 30, 30)
    hip.enable_history(0)
    with pytest.raises(ValueError):
        hip.history("world_pos", 30, 30)


def test_config2_full_baseline_horizon_10000_ticks():
    """BASELINE configs[1] at its full size AND horizon: 65,536 bodies x 10,000 RK4 ticks, fused
    100 ticks per launch, against the oracle on all host cores (~25 s).  Measured worst error 1.6e-12."""
    import os
    hip, ref, w = _pair(65536, 10000, ticks_per_launch=100)
    th = len(os.sched_getaffinity(0))
    worst = parity.Worst()
    for cp in (100, 1000, 2000, 3000, 4000, 5000, 6000, 7000, 8000, 9000, 10000):      # 10 checkpoints and the final tick
        hip.run(cp - hip.tick)
        ref.step(cp - ref.tick, threads=th)
        worst.update(hip, ref, cp)
    # every column per field vector AND the integrated state element by element within 1e-9 (parity.Worst says what is gated)
    worst.check("config2 65,536 x 10,000 ticks")
    assert hip.tick == ref.tick == 10000


def test_nbody_config3_full_size_vs_oracle_over_many_ticks():
    """BASELINE configs[2] at its stated size: 16,384 bodies, all-pairs softened gravity, RK4, dt = 3600 s,
    against the sequential-fold oracle (its all-pairs fold spread over the host cores) at ticks 1, 10, 50 and 100 — the FULL stated
    horizon, about a second of oracle time per tick on the GPU box's 16-CPU quota (~100 s, nearly all of it the oracle;
    SIXDOF_SHORT_TESTS=1 stops at tick 25)."""
    import os
    # 10 checkpoints and the final tick of the stated horizon (the oracle's second per tick is the cost, not the compares)
    checkpoints = (1, 10, 25) if os.environ.get("SIXDOF_SHORT_TESTS") == "1" else (1, 10, 20, 30, 40, 50, 60, 70, 80, 90, 100)
    n = 16384
    pos, vel, inertia = _plummer(n, seed=16384)
    op = (K_SQ, EPS_AU2)
    hip = ea.HipExec(pos, vel, inertia, simulation_time_step=3600.0, effectors=[ea.Effector(L.EFF_ALLPAIRS_GRAVITY_SOFTENED, op)])
    ref = orc.OracleWorld(pos, vel, inertia, simulation_time_step=3600.0, ops=[(orc.EFF_ALLPAIRS_GRAVITY_SOFTENED, op, None)])
    th = len(os.sched_getaffinity(0))
    worst = parity.Worst()
    for cp in checkpoints:
        hip.run(cp - hip.tick)
        ref.step(cp - ref.tick, threads=th)
        worst.update(hip, ref, cp)
    worst.check(f"n-body 16,384 x {checkpoints[-1]} ticks")


def test_nbody_config3_full_size_vs_oracle_and_momentum():
    """BASELINE configs[2] at full size: 16,384 bodies, all-pairs softened gravity, RK4, dt = 3600 s.
    (a) 2 ticks against the sequential-fold oracle (sources spread over the host cores);
    (b) size-independent property over 40 ticks: pair forces are antisymmetric, so total linear momentum is conserved."""
    import os
    n = 16384
    pos, vel, inertia = _plummer(n, seed=16384)
    op = (K_SQ, EPS_AU2)
    hip = ea.HipExec(pos, vel, inertia, simulation_time_step=3600.0, effectors=[ea.Effector(L.EFF_ALLPAIRS_GRAVITY_SOFTENED, op)])
    ref = orc.OracleWorld(pos, vel, inertia, simulation_time_step=3600.0, ops=[(orc.EFF_ALLPAIRS_GRAVITY_SOFTENED, op, None)])
    hip.run(2)
    ref.step(2, threads=len(os.sched_getaffinity(0)))
    errs = parity.state_errors(hip, ref)
    print("n-body 16384 x 2 ticks", errs)
    assert max(errs.values()) < parity.F64_RTOL, errs
    m = inertia[:, 6:7]
    p0 = (m * vel[:, 3:]).sum(axis=0)
    hip.run(38)
    p1 = (m * hip.world_vel[:, 3:]).sum(axis=0)
    scale = np.abs(m * hip.world_vel[:, 3:]).sum(axis=0)
    assert np.all(np.abs(p1 - p0) / scale < 1e-12), (p0, p1)
    assert np.all(hip.force[:, :3] == 0.0)           # el.Force(linear=...) carries no torque


def test_streaming_path_at_2m_bodies():
    """Worlds past the Infinity Cache take the non-temporal instantiation of the step kernel (state > 400 MB):
    2,097,152 bodies x 3 ticks against the oracle, plus bit-equality with the default cache policy."""
    import os
    n = 1 << 21
    hip, ref, w = _pair(n, 3)
    hip.run(3)
    ref.step(3, threads=len(os.sched_getaffinity(0)))
    errs = parity.state_errors(hip, ref)
    assert max(errs.values()) < parity.F64_RTOL, errs
    os.environ["SIXDOF_STREAMING"] = "0"
    try:
        plain, _, _ = _pair(n, 3)
        plain.run(3)
    finally:
        del os.environ["SIXDOF_STREAMING"]
    for f in parity.FIELDS:
        assert np.array_equal(getattr(hip, f), getattr(plain, f)), f


@pytest.mark.parametrize("kind", ["three_body", "allpairs"])
def test_small_graph_single_launch_path_is_bit_identical(kind):
    """n <= 256: pack + fold + integrate (and ticks_per_launch ticks) in one single-workgroup launch; must equal the
    three-kernel path bit for bit."""
    import os
    if kind == "three_body":
        g = gu.load("three_body")
        pos = np.stack([g[f"{e}.world_pos"][0] for e in "abc"])
        vel = np.stack([g[f"{e}.world_vel"][0] for e in "abc"])
        inertia = np.stack([g[f"{e}.inertia"][0] for e in "abc"])
        kw = dict(entity_ids=[1, 2, 3], simulation_time_step=0.008333333,
                  effectors=[ea.Effector(L.EFF_EDGE_GRAVITY_NEWTON, (G_NEWTON,))],
                  edges=(np.array([1, 2, 1, 2, 3, 3], dtype=np.uint64), np.array([2, 1, 3, 3, 1, 2], dtype=np.uint64)))
    else:
        pos, vel, inertia = _plummer(200, seed=5)
        kw = dict(simulation_time_step=3600.0, effectors=[ea.Effector(L.EFF_ALLPAIRS_GRAVITY_SOFTENED, (K_SQ, EPS_AU2))])
    runs = {}
    for label, env, k in (("three_kernels", "0", 1), ("small_k1", None, 1), ("small_k16", None, 16)):
        if env is not None:
            os.environ["SIXDOF_PAIR_SMALL"] = env
        try:
            ex = ea.HipExec(pos, vel, inertia, ticks_per_launch=k, **kw)
            t = ex.run(100)
        finally:
            os.environ.pop("SIXDOF_PAIR_SMALL", None)
        runs[label] = (ex, t.launches)
    # the multi-kernel path: ONE pack launch per batch, then per tick fold + integrate (all-pairs) or ONE fused launch (an edge list
    # without hub sources: pair_kernel.hpp 3b); either way the integrate half writes the next tick's pack rows
    assert runs["three_kernels"][1] == 1 + (1 if kind == "three_body" else 2) * 100 and runs["small_k1"][1] == 100 and runs["small_k16"][1] == 7
    for f in parity.FIELDS:
        a = getattr(runs["three_kernels"][0], f)
        assert np.array_equal(a, getattr(runs["small_k1"][0], f)), f
        assert np.array_equal(a, getattr(runs["small_k16"][0], f)), f


@pytest.mark.parametrize("small", [True, False])
def test_solar_system_833_days_matches_the_oracle_and_the_ephemeris(small, monkeypatch):
    """SURVEY 8(d) config 3's physical sanity case: sun + nine planets from the example's truth CSV, 20,000 one-hour RK4
    ticks of softened all-pairs gravity.  GPU (one-launch small-graph kernel and the three-kernel path) vs the CPU oracle
    at 1e-9, and vs the JPL-derived truth through the example's own accuracy coefficient."""
    from tests import solar_util as su
    if not small:
        monkeypatch.setenv("SIXDOF_PAIR_SMALL", "0")
    d, pos, vel, inertia = su.load()
    ops = [ea.Effector(L.EFF_ALLPAIRS_GRAVITY_SOFTENED, (su.K_SQUARED, su.SOFTENING_AU2))]
    hip = ea.HipExec(pos, vel, inertia, simulation_time_step=su.DT, effectors=ops, ticks_per_launch=240)
    ref = orc.OracleWorld(pos, vel, inertia, simulation_time_step=su.DT, ops=parity.to_oracle_ops(ops))
    ticks = 20_000 if small else 2_400
    days = [x for x in d["days"] if x * su.TICKS_PER_DAY <= ticks]
    sim = np.zeros((len(d["bodies"]), len(days), 3))
    done = 0
    for k, day in enumerate(days):
        step = day * su.TICKS_PER_DAY - done
        if step:
            hip.run(step)
            ref.step(step)
        done += step
        sim[:, k] = hip.world_pos[1:, 4:]
    errs = parity.state_errors(hip, ref)
    rms, coeff, _ = su.accuracy(sim, np.array(d["truth_au"])[:, :len(days)])
    print(f"solar system, {done} ticks ({'one-launch' if small else 'three-kernel'} path): vs oracle {errs}, vs ephemeris RMS {rms:.2e} AU, coefficient {coeff:.6f}")
    assert max(errs.values()) < parity.F64_RTOL, errs
    assert coeff > 0.999 and rms < 5e-3


@pytest.mark.parametrize("path", ["entity", "pair"])
def test_non_finite_world_accel_input_poisons_the_tick_like_the_reference(path):
    """rk4.rs:96-100: stage 0 forms v_s = v0 + 0 * a_in with a_in the world_accel COLUMN; a NaN / Inf there turns that
    stage velocity — and with it sum(v_s), i.e. the new world_pos — into NaN while v' = v0 + (dt/6) sum(A_s) stays finite
    when the forces do not read the velocity.  The HIP path reads the column on the first tick after an upload (later
    ticks' a_in is its own output, already inside v0) and must end up where the oracle does, NaNs included."""
    n = 300
    w = workloads.independent_bodies(n)
    accel = np.zeros((n, 6))
    accel[7, 4] = np.nan
    accel[100, 1] = np.inf
    accel[299, 5] = -np.inf
    if path == "entity":
        eff = workloads.gravity_torque_effectors(w["body_torque"])
        kw = dict(effectors=eff)
        okw = dict(ops=parity.to_oracle_ops(eff))
    else:
        kw = dict(effectors=[ea.Effector(L.EFF_ALLPAIRS_GRAVITY_SOFTENED, (1e-3, 1e-2))])
        okw = dict(ops=[(orc.EFF_ALLPAIRS_GRAVITY_SOFTENED, (1e-3, 1e-2), None)])
    hip = ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], world_accel=accel, simulation_time_step=workloads.DT_120HZ, **kw)
    ref = orc.OracleWorld(w["world_pos"], w["world_vel"], w["inertia"], world_accel=accel, simulation_time_step=workloads.DT_120HZ, **okw)
    for _ in range(3):
        hip.run(1)
        ref.step(1)
        for f in parity.FIELDS:
            g, r = getattr(hip, f), getattr(ref, f)
            if path == "pair" and f == "force":
                # the run-time op interpreter forms q * tau_body for the `force` column even when no op produced a body
                # torque, so a NaN attitude shows as NaN torque there where the reference writes 0: a superset, not a miss
                assert (np.isnan(g) | ~np.isnan(r)).all(), f
            else:
                assert np.array_equal(np.isnan(g), np.isnan(r)), f
            ok = ~np.isnan(g).any(axis=1)
            assert parity.field_rel_err(g[ok][:, -3:], r[ok][:, -3:]) < parity.F64_RTOL, f
    bad = [7, 100, 299]
    assert np.isnan(hip.world_pos[bad]).any(axis=1).all()
    if path == "entity":       # independent rows: the damage stays where it was (a pair fold spreads it to every body by tick 2)
        # rows 7 / 299 took the hit in a linear component only: attitude and velocity stay finite.  Row 100's attitude is
        # NaN, and calc_accel carries the linear half through the attitude (six_dof.rs:137-146), so its velocity goes too.
        assert np.isfinite(hip.world_pos[[0, 8, 150]]).all() and np.isfinite(hip.world_vel[[7, 299]]).all()
        assert np.isnan(hip.world_vel[100]).all()


@pytest.mark.parametrize("integrator", [L.RK4, L.SEMI_IMPLICIT])
def test_infinite_and_zero_mass_rows_follow_the_reference_division(integrator):
    """six_dof.rs:137-146 divides: a static anchor (mass and inertia +inf) gets acceleration f / inf = 0 and keeps
    coasting; zero mass gives +-inf components that the reference then ROTATES (q * ((q^-1 f) / m)), i.e. NaN.  The kernel
    multiplies by a reciprocal computed once per launch (spatial.hpp `recip`), whose Newton refinement must not turn the
    anchor into NaN and whose 1/0 must end where the reference does.  A constant world-frame force and a
    body-frame torque, so no effector multiplies by the mass."""
    n = 200
    w = workloads.independent_bodies(n)
    inertia = w["inertia"].copy()
    inertia[3, 6] = np.inf                 # infinite mass only
    inertia[50, [0, 1, 2, 6]] = np.inf     # a static anchor
    inertia[120, 1] = np.inf               # one infinite principal moment
    inertia[199, 6] = 0.0                  # zero mass under a force
    eff = [ea.Effector(L.EFF_CONST_WRENCH, (0.0, 0.0, 0.0, 1.0, -2.0, 3.0)), ea.Effector(L.EFF_BODY_TORQUE, (), aux_name="body_torque", aux=w["body_torque"])]
    hip = ea.HipExec(w["world_pos"], w["world_vel"], inertia, simulation_time_step=workloads.DT_120HZ, effectors=eff, integrator=integrator)
    ref = orc.OracleWorld(w["world_pos"], w["world_vel"], inertia, simulation_time_step=workloads.DT_120HZ, ops=parity.to_oracle_ops(eff),
                          integrator=integrator)
    hip.run(5)
    ref.step(5)
    for f in parity.FIELDS:
        g, r = getattr(hip, f), getattr(ref, f)
        assert np.array_equal(np.isfinite(g), np.isfinite(r)), (f, np.argwhere(np.isfinite(g) != np.isfinite(r))[:5])
        assert np.array_equal(np.isnan(g), np.isnan(r)), f
    ok = np.isfinite(ref.world_pos).all(axis=1) & np.isfinite(ref.world_vel).all(axis=1) & np.isfinite(ref.world_accel).all(axis=1)
    assert ok[[3, 50, 120]].all() and not ok[199] and np.isnan(hip.world_accel[199, 3:]).all()
    assert np.all(hip.world_accel[50] == 0.0) and np.all(hip.world_accel[3, 3:] == 0.0)
    for f in parity.FIELDS:
        g, r = getattr(hip, f)[ok], getattr(ref, f)[ok]
        err = parity.pos_rel_err(g, r) if f == "world_pos" else max(parity.field_rel_err(g[:, :3], r[:, :3]), parity.field_rel_err(g[:, 3:], r[:, 3:]))
        assert err < parity.F64_RTOL, (f, err)


def test_unsafe_cache_policy_is_not_selectable_in_the_product_library():
    """SIXDOF_STREAMING=2 (write-through `sc1` stores: stale rows in the next launch, profiles/r02_sc1_store_policy_is_unsafe.txt)
    is compiled into the A/B library only; the product ignores the code, so 256 one-tick launches at the size where the
    policy failed equal a fused run bit for bit."""
    import os
    n, ticks = 262_144, 256
    w = workloads.independent_bodies(n)
    eff = workloads.gravity_torque_effectors(w["body_torque"])
    fused = ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], simulation_time_step=workloads.DT_120HZ, effectors=eff, ticks_per_launch=ticks)
    fused.run(ticks)
    os.environ["SIXDOF_STREAMING"] = "2"
    try:
        one = ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], simulation_time_step=workloads.DT_120HZ, effectors=eff, ticks_per_launch=1, use_graph=True)
        one.run(ticks)
    finally:
        del os.environ["SIXDOF_STREAMING"]
    for f in parity.FIELDS:
        assert np.array_equal(getattr(one, f), getattr(fused, f)), f


@pytest.mark.parametrize("n,k", [(1000, 1), (65536, 1), (4099, 16)])
def test_results_do_not_depend_on_which_lane_or_workgroup_an_entity_lands_in(n, k):
    """SURVEY §5 (race detection): the same world with its rows PERMUTED — every entity in another lane, wave and workgroup, ragged
    tails elsewhere — gives the same rows, permuted, bit for bit (RK4, gravity + body torque; 1 and 16 ticks per launch).  An order
    dependence (a lane reading a neighbour's LDS row, a tail lane touching a real row) would show here."""
    w = workloads.independent_bodies(n)
    eff = workloads.gravity_torque_effectors(w["body_torque"])
    a = ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], simulation_time_step=workloads.DT_120HZ, effectors=eff, ticks_per_launch=k)
    a.run(32)
    perm = np.random.default_rng(n + k).permutation(n)
    effp = workloads.gravity_torque_effectors(w["body_torque"][perm])
    b = ea.HipExec(w["world_pos"][perm], w["world_vel"][perm], w["inertia"][perm], simulation_time_step=workloads.DT_120HZ, effectors=effp, ticks_per_launch=k)
    b.run(32)
    for f in parity.FIELDS:
        assert np.array_equal(getattr(a, f)[perm], getattr(b, f)), f
    a.close()
    b.close()
