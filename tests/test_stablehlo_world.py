"""WHOLE-WORLD StableHLO ticks (what libs/nox-py/src/cranelift_compile.rs:47-68 hands a backend) through elodin_amd/stablehlo.py on
the CPU: parse -> world_system (one lane per world, or one lane per entity) -> the numpy walk of the traced program.
tests/test_gpu_stablehlo_world.py runs the same programs through the generated gfx950 kernel."""
import json

import numpy as np
import pytest

from elodin_amd import dsl
from elodin_amd import stablehlo as sh
from oracle import oracle as orc
from tests import dsl_numpy
from tests import stablehlo_world_util as W
from tests.golden import hlo_world_builder as hb


def walk(system, widths, comps, ticks, check=None):
    tp = dsl.Program([system], dsl.Pipe([]), []).trace(widths)
    n = next(iter(comps.values())).shape[0]
    for nm, w in tp.columns:
        comps.setdefault(nm, np.zeros((n, w)))
    pos, vel, acc, inertia = np.tile([0, 0, 0, 1.0, 0, 0, 0], (n, 1)), np.zeros((n, 6)), np.zeros((n, 6)), np.ones((n, 7))
    for r in range(1, ticks + 1):
        dsl_numpy.program_tick_systems_only(tp, pos, vel, acc, inertia, comps, r)
        if check:
            check(r)
    return tp


def test_three_body_world_module_has_the_structure_of_the_reference_dump():
    """libs/cranelift-mlir/tests/three_body_e2e.rs:16-50: 7 inputs, 7 outputs, main + inner + closed_call + norm, and a while whose
    body holds a dynamic_slice and a call."""
    text, slots = hb.three_body_world()
    funcs = sh.parse_module(text)
    assert sorted(funcs) == ["closed_call", "inner", "main", "norm"]
    assert len(funcs["main"].args) == 7 and len(funcs["main"].result_types) == 7 and len(slots) == 7
    whiles = [o for o in funcs["inner"].body if o.name == "stablehlo.while"]
    assert len(whiles) == 4                                                       # one edge_fold per RK4 stage
    for w in whiles:
        body = [o.name for o in w.regions[1]]
        assert "stablehlo.dynamic_slice" in body and "call" in body
    assert text.count('"stablehlo.gather"') == 4 * 4 * 3 and "stablehlo.transpose" in text


def test_three_body_world_tick_reproduces_the_reference_golden_100_ticks():
    """G1 (scripts/ci/baseline/three-body-csv) through the WHOLE-WORLD module, one lane = one world: bit for bit on the CPU walker
    (the module keeps the reference's operation order; numpy does not contract)."""
    system, manifest, widths, row, g = W.three_body("world")
    assert manifest["mode"] == "world" and manifest["rows"] == "worlds"
    comps = {c: np.tile(v[None, :], (2, 1)) for c, v in row.items()}
    worst = [0.0, 0.0]

    def check(r):
        assert comps["hlo_tick"][0, 0] == r == int(g["globals.tick"][r, 0])
        e = W.three_body_errors(comps, g, r)
        worst[0], worst[1] = max(worst[0], e[0]), max(worst[1], e[1])
    walk(system, widths, comps, 100, check)
    assert worst == [0.0, 0.0], worst
    assert np.array_equal(comps["hlo_world_pos"][0], comps["hlo_world_pos"][1])


def test_three_body_world_tick_with_relaxed_arithmetic_stays_inside_1e_9_of_the_golden_100_ticks():
    """The same module under world_system(arith="relaxed") on the CPU walker (shared reciprocals and folded `0 * x`; the walker does not
    contract): G1's 100 ticks within 1e-9, vector-scaled and element-wise — and no longer bit for bit."""
    system, manifest, widths, row, g = W.three_body("world", arith="relaxed")
    assert manifest["arith"] == "relaxed"
    comps = {c: np.tile(v[None, :], (2, 1)) for c, v in row.items()}
    worst = [0.0, 0.0]

    def check(r):
        assert comps["hlo_tick"][0, 0] == r
        e = W.three_body_errors(comps, g, r)
        worst[0], worst[1] = max(worst[0], e[0]), max(worst[1], e[1])
    walk(system, widths, comps, 100, check)
    assert 0.0 < worst[0] <= 1e-9 and worst[1] <= 1e-9, worst


def test_three_body_world_tick_one_lane_per_entity_with_the_exchange_inside_the_wavefront():
    """The same module with one lane per ENTITY (mode "auto" picks it): the edge_fold's constant-index row gathers along the entity axis
    become reads of the other lanes of the world (dsl op `lane_read` = one ds_bpermute per 32-bit half), the per-source stack of
    those rows folds back onto the entity axis, and a world is `rows_per_world` = 4 consecutive rows (3 bodies + 1 padding row).
    G1's 100 ticks, three worlds side by side, bit for bit on the CPU walker; the kernel text is less than half the world-per-lane one."""
    from elodin_amd import codegen
    text, slots = hb.three_body_world()
    system, manifest = sh.world_system(text, slots, mode="auto")
    assert (manifest["mode"], manifest["rows_per_world"], manifest["entities_per_world"]) == ("lane", 4, 3) and manifest["exchange_reads"] > 0
    widths = {c["column"]: c["width"] for c in manifest["columns"]}
    assert widths["hlo_world_pos"] == 7
    g = W.gu.load("three_body")
    S, worlds = 4, 3
    comps = W.strided_world_columns(g, "abc", S, worlds)
    tp_box = []

    def check(r):
        for c, w in W.BODY[:4]:
            for wd in range(worlds):
                for i, e in enumerate("abc"):
                    assert np.array_equal(comps["hlo_" + c][wd * S + i], g[f"{e}.{c}"][r]), (r, c, wd, e)
    tp = walk(system, widths, comps, 100, check)
    src = codegen.generate_source(tp, "float64", 2)
    assert 0 < src.count("__shfl(") <= 64 and len(src) < 100_000


def test_an_executor_whose_rows_split_a_world_is_refused_before_the_device_is_touched():
    """Lane mode with exchange lays a world out as `rows_per_world` consecutive rows; a row count that is not a whole number of
    worlds would let the last one read lanes that do not exist.  HipExec checks it on the traced program (and on a prebuilt
    object through the manifest), before any device call — so this runs without a GPU."""
    import elodin_amd.exec as ea
    from elodin_amd import _lib as L, codegen, dsl
    text, slots = hb.three_body_world()
    system, manifest = sh.world_system(text, slots, mode="lane")
    assert manifest["exchange_reads"] == 20
    widths = {c["column"]: c["width"] for c in manifest["columns"]}
    assert codegen.lane_stride(dsl.Program([system], dsl.Pipe([]), []).trace(widths)) == 4
    rows = 6
    cols = {c: np.zeros((rows, w)) for c, w in widths.items()}
    body = (np.tile([0, 0, 0, 1.0, 0, 0, 0], (rows, 1)), np.zeros((rows, 6)), np.ones((rows, 7)))
    with pytest.raises(ValueError, match="not a whole number of worlds"):
        ea.HipExec(*body, integrator=L.INTEGRATOR_NONE, effectors=dsl.Program([system], dsl.Pipe([]), []), columns=cols)
    frozen = dsl.FrozenProgram(None, list(widths.items()), prebuilt_so="/nonexistent/pipe.so")
    frozen._traced.rows_multiple = manifest["rows_per_world"]
    with pytest.raises(ValueError, match="not a whole number of worlds"):
        ea.HipExec(*body, integrator=L.INTEGRATOR_NONE, effectors=frozen, columns=cols)


def test_a_whole_world_program_leaves_the_body_slabs_alone():
    """Every slot of a whole-world tick is a column of the program; the executor's Body columns are stand-ins.  The program says so
    (`body_free`), the generator checks it and the kernel then neither loads nor stores the Body slabs (NoModel::kBodyDead) — 456 B
    per entity and tick at one tick per launch.  A systems-only program written against the API keeps the pass-through it had."""
    from elodin_amd import codegen
    text, slots = hb.three_body_world()
    system, manifest = sh.world_system(text, slots, mode="lane")
    tp = dsl.Program([system], dsl.Pipe([]), []).trace({c["column"]: c["width"] for c in manifest["columns"]})
    assert tp.body_free and "static constexpr bool kBodyDead = true;" in codegen.generate_source(tp, "float64", 2)

    @dsl.system(x=2)
    def plain(x):
        return {"x": x * 2.0}
    assert "kBodyDead" not in codegen.generate_source(dsl.Program([plain], dsl.Pipe([]), []).trace({"x": 2}), "float64", 2)

    @dsl.system(x=2)
    def liar(x, pos):
        return {"x": x + pos.linear()[0]}
    liar.body_free = True
    with pytest.raises(ValueError, match="declared free of Body state reads"):
        codegen.generate_source(dsl.Program([liar], dsl.Pipe([]), []).trace({"x": 2}), "float64", 2)


def test_ten_body_solar_system_world_tick_in_lane_mode_equals_the_oracle():
    """examples/n-body's world (sun + nine planets, the complete gravity graph: 90 edges, the softened fold of sim.py:349-361) as a
    whole-world module: too large for one lane per world (70-wide world_pos), ingested with one lane per entity, a world = 16
    consecutive rows, every fold target read from another lane of the wavefront.  48 hourly ticks equal to the C oracle's
    sequential fold, bit for bit, for two worlds side by side."""
    from tests import solar_util as su
    d, pos, vel, inertia = su.load()
    n = pos.shape[0]
    text, slots = hb.nbody_world(n, su.K_SQUARED, su.SOFTENING_AU2)
    with pytest.raises(NotImplementedError, match="wider than"):
        sh.world_system(text, slots, mode="world")
    system, manifest = sh.world_system(text, slots, mode="auto")
    assert (manifest["mode"], manifest["rows_per_world"], manifest["entities_per_world"]) == ("lane", 16, 10)
    widths = {c["column"]: c["width"] for c in manifest["columns"]}
    S, worlds = 16, 2
    rows = S * worlds

    def lay(a, fill):
        out = np.tile(np.asarray(fill, dtype=np.float64), (rows, 1))
        for w_ in range(worlds):
            out[w_ * S:w_ * S + n] = a
        return out
    comps = {"hlo_tick": np.zeros((rows, 1)), "hlo_simulation_time_step": np.full((rows, 1), su.DT), "hlo_world_pos": lay(pos, [0, 0, 0, 1.0, 0, 0, 0]),
             "hlo_world_vel": lay(vel, np.zeros(6)), "hlo_inertia": lay(inertia, np.ones(7)), "hlo_world_accel": np.zeros((rows, 6)), "hlo_force": np.zeros((rows, 6))}
    walk(system, widths, comps, 48)
    w = orc.OracleWorld(pos, vel, inertia, simulation_time_step=su.DT, ops=[(orc.EFF_ALLPAIRS_GRAVITY_SOFTENED, (su.K_SQUARED, su.SOFTENING_AU2), None)])
    w.step(48)
    for c, ref in (("world_pos", w.world_pos), ("world_vel", w.world_vel), ("world_accel", w.world_accel), ("force", w.force)):
        for w_ in range(worlds):
            assert np.array_equal(comps["hlo_" + c][w_ * S:w_ * S + n], ref), (c, w_)


def _random_cluster(n):
    rng = np.random.default_rng(n)
    pos = np.concatenate([np.tile([0, 0, 0, 1.0], (n, 1)), rng.normal(size=(n, 3)) * 3], axis=1)
    vel = np.concatenate([np.zeros((n, 3)), rng.normal(size=(n, 3)) * 1e-3], axis=1)
    m = rng.uniform(1e-6, 1e-3, n)
    inertia = np.concatenate([np.tile(m[:, None], (1, 3)), np.zeros((n, 3)), m[:, None]], axis=1)
    return pos, vel, inertia


@pytest.mark.parametrize("n,stride", [(20, 32), (35, 64)])
def test_worlds_of_up_to_a_whole_wavefront_in_lane_mode(n, stride):
    """The exchange stays inside the wavefront up to 64 rows per world: 20 bodies -> 32 rows, 35 bodies (the largest n-body world the
    reference's example is sized for) -> 64 rows, the complete gravity graph (380 / 1,190 edges).  Above 16 rows the per-entity source
    tables are bytes in constant memory instead of 4-bit fields of one literal, and a scan this long is kept a loop (the carried
    values its body hands back untouched are not loop state; the slot's target rows become reads indexed by the counter).  Equal to
    the C oracle's sequential fold, bit for bit, two worlds side by side."""
    from elodin_amd import codegen
    K, EPS, DT = 2.9591220828e-4, 1e-6, 0.5
    pos, vel, inertia = _random_cluster(n)
    text, slots = hb.nbody_world(n, K, EPS)
    system, manifest = sh.world_system(text, slots, mode="auto")
    # the four 19- / 34-trip scans over the edge slot stay COUNTED LOOPS (unrolled they are 9,100 / 13,000 instructions per tick, more
    # than the instruction cache holds): per loop four reads — x, y, z, mass — whose source table is indexed by the counter
    assert (manifest["mode"], manifest["rows_per_world"], manifest["exchange_reads"]) == ("lane", stride, 16)
    widths = {c["column"]: c["width"] for c in manifest["columns"]}
    worlds = 2
    rows = stride * worlds

    def lay(a, fill):
        out = np.tile(np.asarray(fill, dtype=np.float64), (rows, 1))
        for w_ in range(worlds):
            out[w_ * stride:w_ * stride + n] = a
        return out
    comps = {"hlo_tick": np.zeros((rows, 1)), "hlo_simulation_time_step": np.full((rows, 1), DT), "hlo_world_pos": lay(pos, [0, 0, 0, 1.0, 0, 0, 0]),
             "hlo_world_vel": lay(vel, np.zeros(6)), "hlo_inertia": lay(inertia, np.ones(7)), "hlo_world_accel": np.zeros((rows, 6)), "hlo_force": np.zeros((rows, 6))}
    tp = walk(system, widths, comps, 4)
    w = orc.OracleWorld(pos, vel, inertia, simulation_time_step=DT, ops=[(orc.EFF_ALLPAIRS_GRAVITY_SOFTENED, (K, EPS), None)])
    w.step(4)
    for c, ref in (("world_pos", w.world_pos), ("world_vel", w.world_vel), ("world_accel", w.world_accel), ("force", w.force)):
        for w_ in range(worlds):
            assert np.array_equal(comps["hlo_" + c][w_ * stride:w_ * stride + n], ref), (c, w_)
    src = codegen.generate_source(tp, "float64", 2)
    assert f"__device__ const unsigned char ltab0[{stride * (n - 1)}]" in src and src.count("__shfl(") <= 20
    assert src.count("#pragma unroll 1\n") >= 4 and src.count("#pragma unroll 1\n") == src.count(f"< {n - 1}; it_")
    assert codegen.lane_stride(tp) == stride


def test_independent_bodies_world_tick_is_entity_parallel_and_matches_the_oracle():
    """BASELINE configs[1] spelled as a whole-world module ([n, w] tensors, vmapped arithmetic): ingested with one lane per ENTITY
    (the [n, 7] world_pos argument becomes a 7-wide column) and equal to the C oracle's RK4 over 16 ticks."""
    n = 12
    text, slots, cols = W.independent_bodies(n)
    system, manifest = sh.world_system(text, slots, mode="lane")
    assert manifest["mode"] == "lane" and manifest["rows"] == "entities" and manifest["entities_per_world"] == n
    widths = {c["column"]: c["width"] for c in manifest["columns"]}
    assert widths["hlo_world_pos"] == 7 and widths["hlo_torque"] == 3 and widths["hlo_tick"] == 1
    dt = orc.quantize_time_step(120.0)
    comps = {"hlo_" + k: np.array(v) for k, v in cols.items()}
    comps["hlo_tick"], comps["hlo_simulation_time_step"] = np.zeros((n, 1)), np.full((n, 1), dt)
    walk(system, widths, comps, 16)
    w = orc.OracleWorld(cols["world_pos"], cols["world_vel"], cols["inertia"], simulation_time_step=dt,
                        ops=[(orc.EFF_UNIFORM_GRAVITY, (0.0, 0.0, -9.81), None), (orc.EFF_BODY_TORQUE, (), cols["torque"])])
    w.step(16)
    for c, ref in (("world_pos", w.world_pos), ("world_vel", w.world_vel), ("world_accel", w.world_accel), ("force", w.force)):
        assert np.max(np.abs(comps["hlo_" + c] - ref) / np.maximum(np.abs(ref), 1e-9)) < 1e-12, c
    assert np.all(comps["hlo_tick"] == 16)


def test_relaxed_arithmetic_of_a_world_module_stays_inside_1e_9_and_sheds_the_reference_s_dead_work():
    """world_system(arith="relaxed") (dsl.relaxed_arithmetic): finite values assumed, one division per denominator, a * b + c
    contracted in the build.  Not the reference's last bits — the same world as above within 1e-12 of the oracle over 16 ticks —
    and the DAG loses what the reference computes for nothing: the `0 * world_accel` of stage 1 (so the column is never read:
    48 B per entity-tick less) and three divisions in four."""
    from elodin_amd import codegen
    n = 12
    text, slots, cols = W.independent_bodies(n)
    system, manifest = sh.world_system(text, slots, mode="lane", arith="relaxed")
    assert manifest["arith"] == "relaxed" and manifest["mode"] == "lane"
    widths = {c["column"]: c["width"] for c in manifest["columns"]}
    dt = orc.quantize_time_step(120.0)
    comps = {"hlo_" + k: np.array(v) for k, v in cols.items()}
    comps["hlo_tick"], comps["hlo_simulation_time_step"] = np.zeros((n, 1)), np.full((n, 1), dt)
    walk(system, widths, comps, 16)
    w = orc.OracleWorld(cols["world_pos"], cols["world_vel"], cols["inertia"], simulation_time_step=dt,
                        ops=[(orc.EFF_UNIFORM_GRAVITY, (0.0, 0.0, -9.81), None), (orc.EFF_BODY_TORQUE, (), cols["torque"])])
    w.step(16)
    worst = 0.0
    for c, ref in (("world_pos", w.world_pos), ("world_vel", w.world_vel), ("world_accel", w.world_accel), ("force", w.force)):
        got = comps["hlo_" + c]
        halves = ((slice(0, 4), slice(4, 7)) if c == "world_pos" else (slice(0, 3), slice(3, 6)))       # scaled by the field vector's largest
        err = max(np.max(np.abs(got[:, h] - ref[:, h]) / np.maximum(np.max(np.abs(ref[:, h]), axis=1, keepdims=True), 1e-300)) for h in halves)
        worst = max(worst, err)                                                                           # component (tests/parity.py)
        assert err < 1e-12, (c, err)
    assert worst > 0.0                    # ... and it is NOT the reference's arithmetic: some last bit differs
    assert np.all(comps["hlo_tick"] == 16)

    def source(arith):
        system, manifest = sh.world_system(text, slots, mode="lane", arith=arith)
        dsl.Expr.fresh()
        tp = dsl.Program([system], dsl.Pipe([]), []).trace({c["column"]: c["width"] for c in manifest["columns"]})
        return tp, codegen.generate_source(tp, "float64", 2)
    (tr, ref_src), (tx, rel_src) = source("reference"), source("relaxed")
    assert not tr.fp_contract and tx.fp_contract
    assert "fp contract(fast)" in rel_src and "fp contract(fast)" not in ref_src and "m_rcp_relaxed" not in ref_src
    uses = lambda src: sum(src.count(f"= {f}(") for f in ("m_div", "m_rcp_relaxed", "m_rsqrt_relaxed"))      # statements of the tick body
    assert uses(ref_src) >= 3 * uses(rel_src) > 0 and rel_src.count("= m_div(") == 0
    # one_world=True: the Globals columns (slots 0, 1: tick, dt) are read from the first row of the wavefront's block, stored per row
    system, manifest = sh.world_system(text, slots, mode="lane", one_world=True)
    dsl.Expr.fresh()
    tu = dsl.Program([system], dsl.Pipe([]), []).trace({c["column"]: c["width"] for c in manifest["columns"]})
    uni_src = codegen.generate_source(tu, "float64", 2)
    assert manifest["one_world"] and tu.uniform_slots == [0, 1] and tr.uniform_slots == []
    assert uni_src.count("(row & ~uint32_t(kWave - 1))") == 2 and "~uint32_t(kWave - 1)" not in ref_src
    assert uni_src.replace("(size_t)(row & ~uint32_t(kWave - 1))", "(size_t)row") == ref_src        # nothing else differs
    # relaxed arithmetic is a property of the whole object: a program cannot mix a relaxed system with a reference one
    relaxed_sys, m_ = sh.world_system(text, slots, mode="lane", arith="relaxed")

    @dsl.system(extra=1)
    def other(extra):
        return {"extra": extra / 3.0}
    with pytest.raises(ValueError, match="mixes systems traced under relaxed arithmetic"):
        dsl.Program([relaxed_sys, other], dsl.Pipe([]), []).trace({**{c["column"]: c["width"] for c in m_["columns"]}, "extra": 1})
    k = [c for c, _ in tx.columns].index("hlo_world_accel")
    assert f"* r.c{k}[" in ref_src and f"* r.c{k}[" not in rel_src


def test_a_world_of_65536_bodies_traces_as_fast_as_a_world_of_twelve():
    """One lane per entity never materialises an [N, w] tensor: BASELINE's 65,536-body world costs what twelve bodies cost to trace,
    and generates the same program text."""
    import time
    from elodin_amd import codegen
    srcs = []
    for n in (12, 65536):
        text, slots = hb.independent_bodies_world(n)
        t0 = time.perf_counter()
        system, manifest = sh.world_system(text, slots, mode="lane")
        widths = {c["column"]: c["width"] for c in manifest["columns"]}
        dsl.Expr.fresh()
        tp = dsl.Program([system], dsl.Pipe([]), []).trace(widths)
        assert time.perf_counter() - t0 < 30.0
        srcs.append(codegen.generate_source(tp, "float64", 2))
    assert srcs[0] == srcs[1]


def test_what_the_entity_parallel_front_end_refuses_it_refuses_by_reason():
    reduce_world = """
module @module {
  func.func public @main(%arg0: tensor<4x3xf64>) -> tensor<4x3xf64> {
    %c = stablehlo.constant dense<0.0> : tensor<f64>
    %0 = stablehlo.reduce(%arg0 init: %c) applies stablehlo.add across dimensions = [0] : (tensor<4x3xf64>, tensor<f64>) -> tensor<3xf64>
    %1 = stablehlo.broadcast_in_dim %0, dims = [1] : (tensor<3xf64>) -> tensor<4x3xf64>
    %2 = stablehlo.subtract %arg0, %1 : tensor<4x3xf64>
    return %2 : tensor<4x3xf64>
  }
}"""
    with pytest.raises(sh.NotEntityParallel, match="sum over the world"):
        sh.world_system(reduce_world, [("x", [4, 3], False)], mode="lane")
    system, manifest = sh.world_system(reduce_world, [("x", [4, 3], False)], mode="auto")      # ... and still runs, one lane per world
    assert manifest["mode"] == "world" and manifest["columns"][0]["width"] == 12
    comps = {"hlo_x": np.arange(24.0).reshape(2, 12)}
    walk(system, {"hlo_x": 12}, comps, 1)
    x = np.arange(24.0).reshape(2, 4, 3)
    assert np.array_equal(comps["hlo_x"].reshape(2, 4, 3), x - x.sum(axis=1, keepdims=True))


def test_entity_parallel_rules_follow_the_entity_axis_through_shape_ops():
    """vmap's usual spellings: transposes, reshapes that keep the entity axis whole, a dot_general batched over it, a per-entity
    index into a shared table, broadcasts of world-wide scalars — against numpy on a random world."""
    text = """
module @module {
  func.func public @main(%arg0: tensor<5x2x3xf64>, %arg1: tensor<5x3xf64>, %arg2: tensor<f64>, %arg3: tensor<5xi64>) -> (tensor<5x2xf64>, tensor<5x3xf64>) {
    %0 = stablehlo.dot_general %arg0, %arg1, batching_dims = [0] x [0], contracting_dims = [2] x [1] : (tensor<5x2x3xf64>, tensor<5x3xf64>) -> tensor<5x2xf64>
    %1 = stablehlo.broadcast_in_dim %arg2, dims = [] : (tensor<f64>) -> tensor<5x2xf64>
    %2 = stablehlo.multiply %0, %1 : tensor<5x2xf64>
    %3 = stablehlo.transpose %arg1, dims = [1, 0] : (tensor<5x3xf64>) -> tensor<3x5xf64>
    %4 = stablehlo.reshape %3 : (tensor<3x5xf64>) -> tensor<3x1x5xf64>
    %5 = stablehlo.reverse %4, dims = [0] : tensor<3x1x5xf64>
    %6 = stablehlo.reshape %5 : (tensor<3x1x5xf64>) -> tensor<3x5xf64>
    %7 = stablehlo.transpose %6, dims = [1, 0] : (tensor<3x5xf64>) -> tensor<5x3xf64>
    %cst = stablehlo.constant dense<[[1.0, 2.0, 3.0], [10.0, 20.0, 30.0], [100.0, 200.0, 300.0], [-1.0, -2.0, -3.0]]> : tensor<4x3xf64>
    %8 = stablehlo.reshape %arg3 : (tensor<5xi64>) -> tensor<5x1xi64>
    %9 = "stablehlo.gather"(%cst, %8) <{dimension_numbers = #stablehlo.gather<offset_dims = [1], collapsed_slice_dims = [0], start_index_map = [0], index_vector_dim = 1>, indices_are_sorted = false, slice_sizes = array<i64: 1, 3>}> : (tensor<4x3xf64>, tensor<5x1xi64>) -> tensor<5x3xf64>
    %10 = stablehlo.add %7, %9 : tensor<5x3xf64>
    return %2, %10 : tensor<5x2xf64>, tensor<5x3xf64>
  }
}"""
    slots = [("m", [5, 2, 3], False), ("v", [5, 3], False), ("k", [], True), ("row", [5], False)]
    outs = [("mv", [5, 2], False), ("w", [5, 3], False)]
    system, manifest = sh.world_system(text, slots, outs, mode="lane")
    widths = {c["column"]: c["width"] for c in manifest["columns"]}
    assert widths == {"hlo_m": 6, "hlo_v": 3, "hlo_k": 1, "hlo_row": 1, "hlo_mv": 2, "hlo_w": 3}
    rng = np.random.default_rng(3)
    m, v, row = rng.normal(size=(5, 2, 3)), rng.normal(size=(5, 3)), np.array([0, 3, 1, 2, 9])       # 9: clamped to the last row
    comps = {"hlo_m": m.reshape(5, 6), "hlo_v": v.copy(), "hlo_k": np.full((5, 1), 2.5), "hlo_row": row[:, None].astype(float)}
    walk(system, widths, comps, 1)
    table = np.array([[1.0, 2, 3], [10, 20, 30], [100, 200, 300], [-1, -2, -3]])
    assert np.allclose(comps["hlo_mv"], np.einsum("nij,nj->ni", m, v) * 2.5, rtol=1e-15, atol=0)
    assert np.array_equal(comps["hlo_w"], v[:, ::-1] + table[np.minimum(row, 3)])


def test_slots_from_the_references_exec_metadata_json():
    doc = {"arg_ids": [7, 3, 7], "ret_ids": [3, 7], "names": {"3": "tick", "7": "world_pos"},
           "arg_slots": [{"component_id": 7, "shape": [3, 7], "entity_axis_elided": False}, {"component_id": 3, "shape": [], "entity_axis_elided": True},
                         {"component_id": 7, "shape": [3, 7], "entity_axis_elided": False}]}
    ins, outs = sh.slots_from_metadata(json.loads(json.dumps(doc)))
    assert [(s.component, s.shape, s.elided, s.component_id) for s in ins] == [("world_pos", (3, 7), False, 7), ("tick", (), True, 3)]
    assert [s.component for s in outs] == ["tick", "world_pos"] and ins[0].column == "hlo_world_pos"


def test_lane_mode_and_world_mode_agree_on_an_entity_parallel_tick():
    """Differential: the SAME whole-world module (four independent bodies: small enough for one lane per world) ingested both ways —
    one lane per entity, and the whole world in one lane (the plain evaluator, no entity-axis reasoning at all) — gives the same
    columns bit for bit on the CPU walker."""
    n = 4
    text, slots, cols = W.independent_bodies(n, seed=11)
    dt = orc.quantize_time_step(120.0)
    lane_sys, lane_m = sh.world_system(text, slots, mode="lane")
    world_sys, world_m = sh.world_system(text, slots, mode="world")
    assert lane_m["mode"] == "lane" and world_m["mode"] == "world"
    lw = {c["column"]: c["width"] for c in lane_m["columns"]}
    ww = {c["column"]: c["width"] for c in world_m["columns"]}
    assert ww["hlo_world_pos"] == 7 * n and lw["hlo_world_pos"] == 7
    lane = {"hlo_" + k: np.array(v) for k, v in cols.items()}
    lane["hlo_tick"], lane["hlo_simulation_time_step"] = np.zeros((n, 1)), np.full((n, 1), dt)
    world = {"hlo_" + k: np.tile(np.array(v).reshape(1, -1), (2, 1)) for k, v in cols.items()}
    world["hlo_tick"], world["hlo_simulation_time_step"] = np.zeros((2, 1)), np.full((2, 1), dt)
    walk(lane_sys, lw, lane, 8)
    walk(world_sys, ww, world, 8)
    for c in ("world_pos", "world_vel", "world_accel", "force"):
        assert np.array_equal(lane["hlo_" + c].reshape(-1), world["hlo_" + c][0]), c
    assert lane["hlo_tick"][0, 0] == world["hlo_tick"][0, 0] == 8


def test_batched_cholesky_and_triangular_solve_ride_the_entity_axis():
    """`jnp.linalg.cholesky` / `solve_triangular` vmapped over the entities: [N, 3, 3] operands whose leading axis is the entity
    axis (the reference's test_cholesky_batched_mem only asserts a reconstruction; here: numpy per entity)."""
    text = """
module @module {
  func.func public @main(%arg0: tensor<5x3x3xf64>, %arg1: tensor<5x3x1xf64>) -> (tensor<5x3x3xf64>, tensor<5x3x1xf64>) {
    %0 = stablehlo.cholesky %arg0, lower = true : tensor<5x3x3xf64>
    %1 = "stablehlo.triangular_solve"(%0, %arg1) <{left_side = true, lower = true, transpose_a = #stablehlo<transpose NO_TRANSPOSE>, unit_diagonal = false}> : (tensor<5x3x3xf64>, tensor<5x3x1xf64>) -> tensor<5x3x1xf64>
    return %0, %1 : tensor<5x3x3xf64>, tensor<5x3x1xf64>
  }
}"""
    system, manifest = sh.world_system(text, [("a", [5, 3, 3], False), ("b", [5, 3, 1], False)], [("l", [5, 3, 3], False), ("y", [5, 3, 1], False)], mode="lane")
    widths = {c["column"]: c["width"] for c in manifest["columns"]}
    rng = np.random.default_rng(9)
    m = rng.normal(size=(5, 3, 3))
    a = m @ np.transpose(m, (0, 2, 1)) + 3.0 * np.eye(3)
    b = rng.normal(size=(5, 3, 1))
    comps = {"hlo_a": a.reshape(5, 9), "hlo_b": b.reshape(5, 3)}
    walk(system, widths, comps, 1)
    L_ = np.linalg.cholesky(a)
    got_l = comps["hlo_l"].reshape(5, 3, 3)
    assert np.allclose(np.tril(got_l), L_, rtol=1e-13, atol=1e-15)
    assert np.allclose(comps["hlo_y"].reshape(5, 3, 1), np.linalg.solve(L_, b), rtol=1e-12, atol=1e-14)


@pytest.fixture
def rolled_loops(request, monkeypatch):
    """roll=True: every statically counted while of two trips or more stays a counted loop in lane mode (the thresholds that keep
    short loops unrolled are lowered), so the random modules exercise loop-invariant carried values and counter-indexed exchange reads."""
    if request.param:
        monkeypatch.setattr(sh._LaneEval, "ROLL_MIN_TRIPS", 2)
        monkeypatch.setattr(sh._LaneEval, "ROLL_MIN_NODES", 1)
    return request.param


@pytest.mark.parametrize("rolled_loops", [False, True], indirect=True, ids=["unrolled", "rolled"])
@pytest.mark.parametrize("seed", range(24))
def test_random_entity_parallel_modules_lane_mode_equals_world_mode(seed, rolled_loops):
    """Differential fuzz of the entity-axis rules: a seeded random module of vmap-style statements (element-wise ops with broadcast
    scalars, slices / concatenations / reshapes / transposes that keep the entity axis whole, reductions and batched contractions over
    the other axes, selects, counted whiles, per-entity gathers from a shared table, iota ramps) evaluated with one lane per entity
    equals — bit for bit on the CPU walker — the plain evaluator's result with the whole world in one lane."""
    from tests import hlo_fuzz
    n = 4
    text, slots, out_slots = hlo_fuzz.make(seed, n)
    vals = hlo_fuzz.inputs(seed, n)
    lane_sys, lane_m = sh.world_system(text, slots, out_slots, mode="lane")
    world_sys, world_m = sh.world_system(text, slots, out_slots, mode="world")
    lw = {c["column"]: c["width"] for c in lane_m["columns"]}
    ww = {c["column"]: c["width"] for c in world_m["columns"]}
    lane = {"hlo_" + k: (np.tile(np.asarray(v).reshape(1, -1), (n, 1)) if np.ndim(v) == 0 else np.asarray(v, dtype=np.float64).reshape(n, -1)) for k, v in vals.items()}
    world = {"hlo_" + k: np.tile(np.asarray(v, dtype=np.float64).reshape(1, -1), (2, 1)) for k, v in vals.items()}
    walk(lane_sys, lw, lane, 1)
    walk(world_sys, ww, world, 1)
    for j, (name, shape, _) in enumerate(out_slots):
        got, want = lane["hlo_" + name].reshape(-1), world["hlo_" + name][0]
        assert got.shape == want.shape and np.array_equal(got, want, equal_nan=True), (seed, name, shape)
        assert np.isfinite(want).all()


def test_ball_world_tick_with_jax_random_reproduces_the_reference_golden_100_ticks():
    """G2 (scripts/ci/baseline/ball-csv) through the ball's WHOLE-WORLD module: `sample_wind` draws the wind from the seed with jax.random
    as jax lowers it (threefry2x32 in u32 arithmetic, 64 random bits -> mantissa -> bitcast -> erf_inv) inside the tick, `bounce` is a
    vmapped lax.cond, six_dof(gravity | apply_drag) the RK4 — 9 inputs, 9 outputs, main + inner + threefry2x32 + closed_call, the shape
    test_threefry.rs / test_uniform_pipeline.rs:86-112 describe.  Wind, position, velocity, acceleration and force of all 100 recorded
    ticks on the CPU walker."""
    text, _slots = hb.ball_world()
    funcs = sh.parse_module(text)           # libs/cranelift-mlir/tests/e2e.rs:13-24 (9 params, 9 results), test_threefry.rs:8-42 (@closed_call, @inner)
    assert len(funcs["main"].args) == 9 and len(funcs["main"].result_types) == 9 and {"inner", "closed_call", "threefry2x32"} <= set(funcs)
    system, manifest, widths, row, g = W.ball("auto")
    assert manifest["mode"] == "world" and "entity count" in manifest["lane_refused"]       # a singleton world: nothing is batched
    comps = {c: np.tile(v[None, :], (2, 1)) for c, v in row.items()}
    worst = [0.0]

    def check(r):
        assert comps["hlo_tick"][0, 0] == r == int(g["globals.tick"][r, 0]) and comps["hlo_seed"][0, 0] == g["ball.seed"][r, 0]
        worst[0] = max(worst[0], W.ball_errors(comps, g, r))
    walk(system, widths, comps, 100, check)
    print("ball whole-world module vs G2, 100 ticks:", worst[0])
    assert worst[0] < 1e-12
    assert np.allclose(comps["hlo_wind"][0], [-0.2058421394796434, -0.7847657764467411, 1.8160866726679836], rtol=1e-13)   # test_uniform_pipeline.rs:152-156


def test_ball_world_tick_with_relaxed_arithmetic_keeps_jax_random_exact():
    """The ball module under arith="relaxed": the tick draws its wind with threefry2x32 in u32 arithmetic, 64 random bits -> mantissa ->
    bitcast -> erf_inv — integer work spelled with float divisions, which the evaluator keeps EXACT inside a relaxed trace
    (stablehlo._Eval._binary): the seed and the tick stay exact and the wind lands on the reference's own known answer
    (test_uniform_pipeline.rs:152-156); the float state follows G2's 100 ticks within 1e-9 instead of 1e-12."""
    system, manifest, widths, row, g = W.ball("auto", arith="relaxed")
    assert manifest["mode"] == "world" and manifest["arith"] == "relaxed"
    comps = {c: np.tile(v[None, :], (2, 1)) for c, v in row.items()}
    worst = [0.0]

    def check(r):
        assert comps["hlo_tick"][0, 0] == r == int(g["globals.tick"][r, 0]) and comps["hlo_seed"][0, 0] == g["ball.seed"][r, 0]
        worst[0] = max(worst[0], W.ball_errors(comps, g, r))
    walk(system, widths, comps, 100, check)
    print("ball whole-world module, relaxed arithmetic, vs G2, 100 ticks:", worst[0])
    assert worst[0] < 1e-9
    assert np.allclose(comps["hlo_wind"][0], [-0.2058421394796434, -0.7847657764467411, 1.8160866726679836], rtol=1e-13)


@pytest.mark.parametrize("rolled_loops", [False, True], indirect=True, ids=["unrolled", "rolled"])
@pytest.mark.parametrize("seed,n", [(s_, 3 + s_ % 3) for s_ in range(100, 116)])
def test_random_modules_with_reads_between_entities_lane_exchange_equals_world_mode(seed, n, rolled_loops):
    """The same differential fuzz with JOINS in the mix: every entity reads rows of other entities of its world by a constant table
    (gather along the entity axis, re-stacked per source — graph.rs:187-235's shape).  One lane per entity turns them into lane
    exchanges inside a world of `rows_per_world` rows (3, 4 and 5 entities: strides 4, 4 and 8, padding rows included); the result
    equals the plain evaluator's with the whole world in one lane, bit for bit, for two worlds side by side."""
    from tests import hlo_fuzz
    text, slots, out_slots = hlo_fuzz.make(seed, n, exchange=True)
    vals = [hlo_fuzz.inputs(seed + 1000 * w_, n) for w_ in range(2)]
    for v in vals[1:]:
        v["k"] = vals[0]["k"]                       # the world-wide scalar is one value per world; keep the worlds' programs comparable
    lane_sys, lane_m = sh.world_system(text, slots, out_slots, mode="lane")
    world_sys, world_m = sh.world_system(text, slots, out_slots, mode="world")
    S = lane_m.get("rows_per_world", n)
    lw = {c["column"]: c["width"] for c in lane_m["columns"]}
    ww = {c["column"]: c["width"] for c in world_m["columns"]}
    lane, world = {}, {}
    for k in vals[0]:
        per_world = [np.asarray(v[k], dtype=np.float64) for v in vals]
        if np.ndim(vals[0][k]) == 0:
            lane["hlo_" + k] = np.concatenate([np.tile(p.reshape(1, -1), (S, 1)) for p in per_world])
        else:
            blocks = []
            for p in per_world:
                b = np.zeros((S, p.reshape(n, -1).shape[1]))
                b[:n] = p.reshape(n, -1)
                blocks.append(b)
            lane["hlo_" + k] = np.concatenate(blocks)
        world["hlo_" + k] = np.stack([p.reshape(-1) for p in per_world])
    walk(lane_sys, lw, lane, 1)
    walk(world_sys, ww, world, 1)
    for name, shape, _ in out_slots:
        for w_ in range(2):
            got, want = lane["hlo_" + name][w_ * S:w_ * S + n].reshape(-1), world["hlo_" + name][w_]
            assert np.array_equal(got, want, equal_nan=True), (seed, n, name, w_)
    assert lane_m["exchange_reads"] > 0 and S == (4 if n <= 4 else 8)      # every module returns a join's result: the exchange is live


def test_float32_builds_are_refused_for_modules_that_work_on_integer_bit_patterns():
    """integer tensors travel as integral floats of the program's element type — exact in f64 (2^53), not in f32
    (2^24).  A module that reinterprets / shifts / masks bits, multiplies wide integers or holds a constant >= 2^24 (jax.random's
    threefry, libs/cranelift-mlir/tests/test_threefry.rs) must refuse `dtype float32`; the float-only worlds (three-body, n-body,
    independent bodies: gather indices and loop counters are small) must not, and an integer COLUMN is named in the manifest."""
    from elodin_amd import codegen
    from tests import stablehlo_util as U
    prng = [c for c in U.WORLD_CASES if "threefry" in c["name"] or "uniform" in c["name"]]
    assert prng, "the fixture should hold the reference's PRNG cases"
    for case in prng:
        why = sh.float32_hazards(sh.parse_module(case["mlir"]))
        assert why, case["name"]
        assert any(w_ in " ".join(why) for w_ in ("bit patterns", "2^24", "stablehlo.xor", "stablehlo.or", "stablehlo.and", "multiply")), why
    text, slots = hb.nbody_world(10, 2.9591220828e-4, 1e-6)
    assert sh.float32_hazards(sh.parse_module(text)) == []
    system, manifest = sh.world_system(text, slots, mode="auto")
    assert "float32_refused" not in manifest and system.float32_refused == []
    assert manifest["float32_integer_columns"] == ["hlo_tick"]          # exact below 2^24 ticks in a float32 build: said, not refused
    # a hazardous system reaches codegen.build(..., "float32") only to be refused there (the executor path), and compile_world refuses too
    case = prng[0]
    system, values, expect = U.build(case)
    widths = {k: len(np.atleast_1d(v)) for k, v in values.items()}
    tp = dsl.Program([system], dsl.Pipe([]), []).trace({**widths, **{k: w for k, (w, _) in expect.items()}})
    assert tp.float32_refused
    with pytest.raises(NotImplementedError, match="float32"):
        codegen.build(tp, "float32", 2)


def _nbody_start(nb):
    rng = np.random.default_rng(nb)
    pos = np.concatenate([np.tile([0, 0, 0, 1.0], (nb, 1)), rng.normal(size=(nb, 3)) * 3], axis=1)
    vel = np.concatenate([np.zeros((nb, 3)), rng.normal(size=(nb, 3)) * 1e-3], axis=1)
    m = rng.uniform(1e-6, 1e-3, nb)
    return pos, vel, np.concatenate([np.tile(m[:, None], (1, 3)), np.zeros((nb, 3)), m[:, None]], axis=1)


def test_a_world_larger_than_a_wavefront_becomes_systems_and_fold_stages_and_equals_the_oracle_on_the_cpu_walker():
    """VERDICT r05 missing #2.  An 80-body n-body world as the reference would dump it (graph.rs:177-361: per source constant-index
    gathers of its targets, stacked [N, e, 7], transposed, a `while` over the edge slot slicing row i and calling the fold body).  Its
    entities exchange data across more than one wavefront, so world_system refuses it both ways; world_program lifts the four scans
    (one per RK4 stage) out as fold stages between per-entity systems.  On the numpy walker: BIT-IDENTICAL to the oracle's sequential
    softened fold (same operations in the same order), the fold's edges in the module's slot order."""
    nb = 80
    text, slots = hb.nbody_world(nb, 2.9591220828e-4, 1e-6)
    with pytest.raises((sh.NotEntityParallel, NotImplementedError)):
        sh.world_system(text, slots, mode="auto")
    prog, manifest, edges = sh.world_program(text, slots)
    assert manifest["mode"] == "folds" and manifest["fold_stages"] == 4 and manifest["rows_per_world"] == nb
    widths = {c["column"]: c["width"] for c in manifest["columns"]}
    assert widths["hlo_fold0_own"] == widths["hlo_fold0_nbr"] == 4 and widths["hlo_fold0_out"] == 6      # position + mass either side, a Force out
    assert [c["column"] for c in manifest["columns"] if c["scratch"]] == [f"hlo_fold{k}_{s}" for k in range(4) for s in ("own", "nbr", "out")]
    tp = prog.trace(widths, fold_edges=edges)
    assert [type(s_).__name__ for s_ in tp.pre] == ["TracedSystem", "TracedFoldStage"] * 4 + ["TracedSystem"]
    for fs in tp.fold_stages:
        assert fs.src_rows == list(range(nb)) and fs.dst == [t for s_ in range(nb) for t in range(nb) if t != s_]      # slot order = ascending targets
        assert fs.traced.fold.init == (0.0,) * 6
    pos, vel, inertia = _nbody_start(nb)
    dt = 0.5
    comps = {"hlo_tick": np.zeros((nb, 1)), "hlo_simulation_time_step": np.full((nb, 1), dt), "hlo_world_pos": pos.copy(), "hlo_world_vel": vel.copy(),
             "hlo_world_accel": np.zeros((nb, 6)), "hlo_force": np.zeros((nb, 6)), "hlo_inertia": inertia.copy()}
    for nm, w in tp.columns:
        comps.setdefault(nm, np.zeros((nb, w)))
    bp, bv, ba, bi = np.tile([0, 0, 0, 1.0, 0, 0, 0], (nb, 1)), np.zeros((nb, 6)), np.zeros((nb, 6)), np.ones((nb, 7))
    ref = orc.OracleWorld(pos, vel, inertia, simulation_time_step=dt, ops=[(orc.EFF_ALLPAIRS_GRAVITY_SOFTENED, (2.9591220828e-4, 1e-6), None)])
    for r in range(1, 3):
        dsl_numpy.program_tick_systems_only(tp, bp, bv, ba, bi, comps, r)
        ref.step(1)
        for nm, arr in (("hlo_world_pos", ref.world_pos), ("hlo_world_vel", ref.world_vel), ("hlo_world_accel", ref.world_accel), ("hlo_force", ref.force)):
            assert np.array_equal(comps[nm], arr), (r, nm, np.abs(comps[nm] - arr).max())
        assert np.all(comps["hlo_tick"] == r)
    # the generated program compiles without spills (hipcc cross-compiles here), fold kernels and their baked CSR in the text
    from elodin_amd import codegen
    src = codegen.generate_source(tp, "float64", 2)
    # the fold kernels; a COMPLETE graph bakes no tables (slot s of source i is row s + (s >= i)); a scan of 64 edges or more that is a
    # plain sum is folded by a whole wave per source (lane partials + a fixed shuffle tree) ...
    assert src.count("_kernel(const StepParams P)") >= 4 and "fold3_commit" in src and "fold3_dst" not in src
    assert src.count("one WAVE per source") == 4 and "__shfl_down(v3, off, 64)" in src and codegen._graph_fold_kinds(tp.fold_stages[0].traced.outputs, 6) == ["keep"] * 3 + ["sum"] * 3
    assert edges["hlo_fold0_edges"] == ("complete", nb) and all(fs.complete == nb for fs in tp.fold_stages)
    # ... unless the host asks for the sequential fold — one lane per source, targets fetched four at a time, the reference's order bit for bit
    seq, m2, e2 = sh.world_program(text, slots, wave_folds=False)
    src2 = codegen.generate_source(seq.trace({c["column"]: c["width"] for c in m2["columns"]}, fold_edges=e2), "float64", 2)
    assert "one WAVE per source" not in src2 and "e + u + ((e + u) >= i ? 1u : 0u)" in src2


def test_world_program_refuses_what_is_not_an_edge_fold_scan():
    """Entity-parallel ticks have no scan to lift (world_system is their front end); a small exchanging world is a lane-mode world."""
    text, slots = hb.independent_bodies_world(128)
    with pytest.raises(sh.NotEntityParallel, match="no scan"):
        sh.world_program(text, slots)
    # ... and a compile_world in the mode that cannot hold the world says so instead of falling through
    text, slots = hb.nbody_world(70, 2.9591220828e-4, 1e-6)
    meta = {"arg_ids": [1 + k for k in range(len(slots))], "ret_ids": [1 + k for k in range(len(slots))], "names": {str(1 + k): c for k, (c, _, _) in enumerate(slots)},
            "arg_slots": [{"component_id": 1 + k, "shape": s_, "entity_axis_elided": e_} for k, (c, s_, e_) in enumerate(slots)]}
    with pytest.raises((sh.NotEntityParallel, NotImplementedError)):
        sh.compile_world(text, meta, mode="lane")
    with pytest.raises(NotImplementedError, match="float64"):
        sh.compile_world(text, meta, mode="folds", dtype="float32")


def test_a_sparse_newton_fold_over_a_large_world_is_lifted_the_same_way():
    """Another fold body (examples/three-body/main.py:64-71: Newton gravity through `jnp.linalg.norm`, a function of its own in the
    module, replacing the force's torque by zero) over another graph (70 bodies, three out-edges each, not a complete graph): the edges
    come out in slot order and the walker equals the oracle's edge fold bit for bit."""
    nb = 70
    targets = {s_: [(s_ + k) % nb for k in (1, 5, 11)] for s_ in range(nb)}
    G = 6.6743e-11
    text, slots = hb.edge_fold_world(nb, targets, "newton", (G,))
    prog, manifest, edges = sh.world_program(text, slots)
    assert manifest["edges_per_fold"] == [3 * nb] * 4
    for frm, to in edges.values():
        assert list(frm) == [s_ for s_ in range(nb) for _ in range(3)] and list(to) == [t for s_ in range(nb) for t in targets[s_]]
    tp = prog.trace({c["column"]: c["width"] for c in manifest["columns"]}, fold_edges=edges)
    rng = np.random.default_rng(3)
    pos = np.concatenate([np.tile([0, 0, 0, 1.0], (nb, 1)), rng.normal(size=(nb, 3)) * 10], axis=1)
    vel = np.concatenate([np.zeros((nb, 3)), rng.normal(size=(nb, 3))], axis=1)
    m = rng.uniform(1e9, 1e10, nb)
    inertia = np.concatenate([np.tile(m[:, None], (1, 3)), np.zeros((nb, 3)), m[:, None]], axis=1)
    comps = {"hlo_tick": np.zeros((nb, 1)), "hlo_simulation_time_step": np.full((nb, 1), 0.01), "hlo_world_pos": pos.copy(), "hlo_world_vel": vel.copy(),
             "hlo_world_accel": np.zeros((nb, 6)), "hlo_force": np.zeros((nb, 6)), "hlo_inertia": inertia.copy()}
    for nm, w in tp.columns:
        comps.setdefault(nm, np.zeros((nb, w)))
    bp, bv, ba, bi = np.tile([0, 0, 0, 1.0, 0, 0, 0], (nb, 1)), np.zeros((nb, 6)), np.zeros((nb, 6)), np.ones((nb, 7))
    src = np.array([s_ for s_ in range(nb) for _ in targets[s_]], dtype=np.uint32)
    dst = np.array([t for s_ in range(nb) for t in targets[s_]], dtype=np.uint32)
    ref = orc.OracleWorld(pos, vel, inertia, simulation_time_step=0.01, ops=[(orc.EFF_EDGE_GRAVITY_NEWTON, (G,), None)], edges=(src, dst))
    for r in range(1, 4):
        dsl_numpy.program_tick_systems_only(tp, bp, bv, ba, bi, comps, r)
    ref.step(3)
    for nm, arr in (("hlo_world_pos", ref.world_pos), ("hlo_world_vel", ref.world_vel), ("hlo_world_accel", ref.world_accel), ("hlo_force", ref.force)):
        assert np.array_equal(comps[nm], arr), nm


def test_the_cli_builds_a_fold_stage_world_without_a_gpu(tmp_path):
    """`python -m elodin_amd.stablehlo tick.mlir --slots slots.json -o pipe.so` on a 72-body n-body world (hipcc cross-compiles): --mode auto
    falls through lane and world mode to fold stages; the manifest lists the scratch columns a host binds zero-filled and the row count the
    object was generated for; --sequential-folds asks for the one-lane fold."""
    import subprocess
    import sys
    from elodin_amd import _lib as L
    nb = 72
    text, slots = hb.nbody_world(nb, 2.9591220828e-4, 1e-6)
    (tmp_path / "tick.mlir").write_text(text)
    meta = {"arg_ids": [L.component_id(c) for c, _, _ in slots], "ret_ids": [L.component_id(c) for c, _, _ in slots], "names": {str(L.component_id(c)): c for c, _, _ in slots},
            "rows": 3 * nb, "arg_slots": [{"component_id": L.component_id(c), "shape": s_, "entity_axis_elided": e_} for c, s_, e_ in slots]}
    (tmp_path / "slots.json").write_text(json.dumps(meta))
    for flags, said in (([], "a wave per source"), (["--sequential-folds"], "sequential")):
        out = tmp_path / ("pipe" + ("_seq" if flags else "") + ".so")
        res = subprocess.run([sys.executable, "-m", "elodin_amd.stablehlo", str(tmp_path / "tick.mlir"), "--slots", str(tmp_path / "slots.json"), "-o", str(out), *flags],
                             capture_output=True, text=True, cwd=str(L.PKG.parent))
        assert res.returncode == 0, res.stderr[-2000:]
        line = json.loads(res.stdout.strip().splitlines()[-1])
        assert line["mode"] == "folds" and out.exists()
        prog, manifest = sh.load_world(str(out))
        assert manifest["fold_stages"] == 4 and manifest["rows_per_world"] == nb and manifest["row_count"] == 3 * nb and said in manifest["folds"]
        scratch = [c["column"] for c in manifest["columns"] if c.get("scratch")]
        assert len(scratch) == 16 and "hlo_fold2_nbr" in scratch and "hlo_fold3_out#fold3" in scratch
        assert prog._traced.exact_rows == 3 * nb and [n_ for n_, _ in prog._traced.columns] == [c["column"] for c in manifest["columns"]]
        assert manifest["build"]["resources"]["vgpr_spills"] == 0


def test_k9_rk4_pipeline_shape_singleton_slots_are_ingested_as_the_reference_declares_them():
    """K9 (libs/nox-py/src/integrator/rk4.rs:175-227 `rk4_pipeline_shape`): a one-entity world integrated with `rk4::<X, V>` compiles
    to a function whose inputs are {X, V, SimulationTimeStep}, whose outputs hold X, and whose X / V slots are ZERO-DIM with
    `entity_axis_elided` (a scalar component on a singleton column: the batch axis is elided, system.rs:12-23).  That metadata is
    what a WorldExec::Hip hands this front end (exec.rs:17-29): the same document — ids, names, arity = number of inputs — is read
    back as three scalar, elided slots, and the RK4 tick of x' = v, v' = 0 over it moves x = 0, v = 10 by dt * v exactly."""
    x_id, v_id, dt_id = 101, 102, 103
    meta = {"arg_ids": [x_id, v_id, dt_id], "ret_ids": [x_id], "names": {str(x_id): "x", str(v_id): "v", str(dt_id): "simulation_time_step"},
            "arg_slots": [{"component_id": x_id, "shape": [], "entity_axis_elided": True}, {"component_id": v_id, "shape": [], "entity_axis_elided": True},
                          {"component_id": dt_id, "shape": [], "entity_axis_elided": True}]}
    ins, outs = sh.slots_from_metadata(meta)
    assert [s_.component for s_ in ins] == ["x", "v", "simulation_time_step"] and [s_.component for s_ in outs] == ["x"]
    assert all(tuple(s_.shape) == () and s_.elided for s_ in ins + outs)
    # the arithmetic Rk4::compile emits for U = {x}, DU = {v} with no force system: four stages that all see v, then x + dt/6 (v + 2v + 2v + v)
    text = """
module @module {
  func.func public @main(%arg0: tensor<f64>, %arg1: tensor<f64>, %arg2: tensor<f64>) -> tensor<f64> {
    %two = stablehlo.constant dense<2.0> : tensor<f64>
    %sixth = stablehlo.constant dense<0.16666666666666666> : tensor<f64>
    %k2 = stablehlo.multiply %two, %arg1 : tensor<f64>
    %s0 = stablehlo.add %arg1, %k2 : tensor<f64>
    %s1 = stablehlo.add %s0, %k2 : tensor<f64>
    %s2 = stablehlo.add %s1, %arg1 : tensor<f64>
    %g = stablehlo.multiply %arg2, %sixth : tensor<f64>
    %dx = stablehlo.multiply %g, %s2 : tensor<f64>
    %x1 = stablehlo.add %arg0, %dx : tensor<f64>
    return %x1 : tensor<f64>
  }
}"""
    funcs = sh.parse_module(text)
    assert len(funcs["main"].args) == len(meta["arg_ids"])                       # arity = number of declared inputs (K9's last assertion)
    system, manifest = sh.world_system(text, ins, outs, mode="auto")
    widths = {c["column"]: c["width"] for c in manifest["columns"]}
    assert widths == {"hlo_x": 1, "hlo_v": 1, "hlo_simulation_time_step": 1}
    assert all(c["entity_axis_elided"] and c["shape"] == [] for c in manifest["columns"])
    dt = orc.quantize_time_step(120.0)
    comps = {"hlo_x": np.zeros((1, 1)), "hlo_v": np.full((1, 1), 10.0), "hlo_simulation_time_step": np.full((1, 1), dt)}
    walk(system, widths, comps, 1)
    assert comps["hlo_x"][0, 0] == 0.0 + (dt * (1.0 / 6.0)) * (10.0 + 20.0 + 20.0 + 10.0) and comps["hlo_v"][0, 0] == 10.0


def test_the_cli_builds_the_relaxed_one_world_object_without_a_gpu(tmp_path):
    """`python -m elodin_amd.stablehlo tick.mlir --slots slots.json -o pipe.so --mode lane --arith relaxed --one-world` (hipcc cross-compiles):
    configs[1] as a whole-world module; the printed line and the manifest beside the object say what the host asked for, the object
    installs like any other (`load_world`), it spills nothing and needs fewer registers than the default build of the same module."""
    import subprocess
    import sys
    from elodin_amd import _lib as L
    n = 4096
    text, slots = hb.independent_bodies_world(n)
    (tmp_path / "tick.mlir").write_text(text)
    (tmp_path / "slots.json").write_text(json.dumps({"inputs": [{"component": c, "shape": s_, "entity_axis_elided": e_} for c, s_, e_ in slots], "rows": n}))
    built = {}
    for tag, flags in (("default", []), ("relaxed", ["--arith", "relaxed", "--one-world"])):
        out = tmp_path / f"pipe_{tag}.so"
        res = subprocess.run([sys.executable, "-m", "elodin_amd.stablehlo", str(tmp_path / "tick.mlir"), "--slots", str(tmp_path / "slots.json"), "-o", str(out),
                              "--mode", "lane", *flags], capture_output=True, text=True, cwd=str(L.PKG.parent))
        assert res.returncode == 0, res.stderr[-2000:]
        line = json.loads(res.stdout.strip().splitlines()[-1])
        prog, manifest = sh.load_world(str(out))
        assert line["mode"] == manifest["mode"] == "lane" and out.exists()
        built[tag] = (line, manifest)
    assert "arith" not in built["default"][0] and "one_world" not in built["default"][1]
    assert built["relaxed"][0]["arith"] == "relaxed" and built["relaxed"][0]["one_world"] is True
    assert built["relaxed"][1]["arith"] == "relaxed" and built["relaxed"][1]["one_world"] is True
    r0, r1 = built["default"][1]["build"]["resources"], built["relaxed"][1]["build"]["resources"]
    assert r0["vgpr_spills"] == r1["vgpr_spills"] == 0 and r1["vgprs"] < r0["vgprs"]
    with pytest.raises(NotImplementedError, match="one_world applies to one-kernel ticks"):
        sh.compile_world(*((lambda t, s: (t, {"inputs": [{"component": c, "shape": sh_, "entity_axis_elided": e_} for c, sh_, e_ in s]}))(*hb.nbody_world(72, 2.9591220828e-4, 1e-6))),
                         mode="folds", one_world=True)
