"""WHOLE-WORLD StableHLO ticks through the generated gfx950 kernel: the reference's world-fragment known answers,
the assembled three-body world module against G1's 100 ticks (one lane = one world, a Monte-Carlo of worlds), BASELINE configs[1]
as an entity-batched module with one lane per entity against the hand-written step kernel, and the build-time CLI's object
installed as it is.  CPU twins: tests/test_stablehlo_ingest.py, tests/test_stablehlo_world.py."""
import json
import subprocess
import sys

import numpy as np
import pytest

import elodin_amd as ea
from elodin_amd import _lib as L
from elodin_amd import dsl, workloads
from elodin_amd import stablehlo as sh
from oracle import oracle as orc
from tests import stablehlo_util as U
from tests import stablehlo_world_util as W

pytestmark = pytest.mark.gpu


def _exec(program, columns, n, **kw):
    w = workloads.independent_bodies(n)
    return ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], integrator=L.INTEGRATOR_NONE, effectors=program, columns=columns, **kw)


def test_reference_world_fragment_known_answers_through_the_generated_kernel():
    """libs/cranelift-mlir/tests/test_{gather_3body,dynamic_ops_3body,while_dyn_slice,closed_call,threefry,threefry_e2e,
    uniform_pipeline}.rs: all 24 cases as systems of ONE generated kernel; integer results exact (u32 wrap-around, ui64 words)."""
    systems, columns, expects = [], {}, []
    n = 70
    for k, case in enumerate(U.WORLD_CASES):
        system, values, expect = U.build(case, prefix=f"k{k}_")
        systems.append(system)
        for nm, v in values.items():
            columns[nm] = np.tile(v.reshape(1, -1), (n, 1))
        for nm, (w, _) in expect.items():
            columns[nm] = np.zeros((n, w))
        expects.append((case, expect))
    assert len(columns) <= dsl.MAX_PROGRAM_COLUMNS
    hip = _exec(dsl.Program(systems, dsl.Pipe([]), []), columns, n)
    hip.run(1)
    for case, expect in expects:
        for nm, (width, exp) in expect.items():
            got = np.asarray(hip._aux[nm], dtype=np.float64)
            out_k = nm.split("_out")[1]
            integer = case["expected"].get(out_k, {"type": "f64"})["type"] != "f64"
            U.check(case["name"], got[0], width, exp, 0.0 if integer else max(case["tol"], 1e-15))
            assert np.array_equal(got, np.repeat(got[:1], n, axis=0), equal_nan=True), (case["name"], nm)
    print(f"{len(expects)} world-fragment known answers of the reference through one generated kernel")
    hip.close()


def test_three_body_whole_world_module_reproduces_g1_on_the_gpu():
    """The assembled whole-world tick (7 in / 7 out, main + inner + closed_call + norm; gather + transpose + while + dynamic_slice +
    call) with one lane per WORLD: lane 0 flies the reference's golden initial state for its 100 recorded ticks (<= 1e-9, vector-
    scaled AND element-wise), the other lanes fly perturbed worlds and are compared with the C oracle world by world."""
    system, manifest, widths, row, g = W.three_body("world")
    assert manifest["mode"] == "world"
    n = 96
    rng = np.random.default_rng(17)
    cols = {c: np.tile(v[None, :], (n, 1)) for c, v in row.items()}
    cols["hlo_world_pos"][1:, [4, 5, 11, 12, 18, 19]] += rng.uniform(-0.05, 0.05, (n - 1, 6))        # other worlds: other starts
    cols["hlo_world_vel"][1:, [3, 4, 9, 10, 15, 16]] += rng.uniform(-0.05, 0.05, (n - 1, 6))
    start = {k: v.copy() for k, v in cols.items()}
    hip = _exec(dsl.Program([system], dsl.Pipe([]), []), cols, n)
    worst = [0.0, 0.0]
    for r in range(1, 101):
        hip.run(1)
        e = W.three_body_errors(hip._aux, g, r)
        worst = [max(worst[0], e[0]), max(worst[1], e[1])]
        assert hip._aux["hlo_tick"][0, 0] == r
    print(f"three-body whole-world module, 100 ticks vs G1: {worst[0]:.2e} (vector-scaled), {worst[1]:.2e} (element-wise)")
    assert worst[0] <= 1e-9 and worst[1] <= 1e-9, worst
    dt = float(g["globals.simulation_time_step"][0, 0])
    G = 6.6743e-11
    src, dst = np.array([0, 1, 0, 1, 2, 2], dtype=np.uint32), np.array([1, 0, 2, 2, 0, 1], dtype=np.uint32)
    bad = 0.0
    for lane in (1, 7, 40, 95):
        w = orc.OracleWorld(start["hlo_world_pos"][lane].reshape(3, 7), start["hlo_world_vel"][lane].reshape(3, 6),
                            start["hlo_inertia"][lane].reshape(3, 7), simulation_time_step=dt,
                            ops=[(orc.EFF_EDGE_GRAVITY_NEWTON, (G,), None)], edges=(src, dst))
        w.step(100)
        for c, ref in (("world_pos", w.world_pos), ("world_vel", w.world_vel), ("world_accel", w.world_accel), ("force", w.force)):
            got = hip._aux["hlo_" + c][lane].reshape(ref.shape)
            bad = max(bad, float(np.max(np.abs(got - ref) / np.maximum(np.max(np.abs(ref), axis=1, keepdims=True), 1e-300))))
    print(f"Monte-Carlo of three-body worlds vs the oracle, 100 ticks: {bad:.2e}")
    assert bad <= 1e-9
    hip.close()


@pytest.mark.parametrize("n", [1000, 65536])
def test_independent_bodies_whole_world_module_one_lane_per_entity(n):
    """BASELINE configs[1] as the reference would dump it ([n, 7] / [n, 6] tensors, vmapped arithmetic): the entity axis becomes
    the executor's rows; 16 ticks against the hand-written step kernel on the same world (and the oracle on a sample)."""
    text, slots, cols = W.independent_bodies(n)
    system, manifest = sh.world_system(text, slots, mode="lane")
    assert manifest["mode"] == "lane"
    dt = orc.quantize_time_step(120.0)
    columns = {"hlo_" + k: np.array(v) for k, v in cols.items()}
    columns["hlo_tick"], columns["hlo_simulation_time_step"] = np.zeros((n, 1)), np.full((n, 1), dt)
    hip = _exec(dsl.Program([system], dsl.Pipe([]), []), columns, n)
    hip.run(16)
    ref = ea.HipExec(cols["world_pos"], cols["world_vel"], cols["inertia"], simulation_time_step=dt,
                     effectors=[ea.Effector(L.EFF_UNIFORM_GRAVITY, (0.0, 0.0, -9.81)), ea.Effector(L.EFF_BODY_TORQUE, (), "torque", cols["torque"])])
    ref.run(16)
    for c, r in (("world_pos", ref.world_pos), ("world_vel", ref.world_vel), ("world_accel", ref.world_accel), ("force", ref.force)):
        got = hip._aux["hlo_" + c]
        err = float(np.max(np.abs(got - r) / np.maximum(np.max(np.abs(r), axis=1, keepdims=True), 1e-300)))
        assert err <= 1e-9, (c, err)
    assert np.all(hip._aux["hlo_tick"] == 16)
    k = min(n, 64)
    w = orc.OracleWorld(cols["world_pos"][:k], cols["world_vel"][:k], cols["inertia"][:k], simulation_time_step=dt,
                        ops=[(orc.EFF_UNIFORM_GRAVITY, (0.0, 0.0, -9.81), None), (orc.EFF_BODY_TORQUE, (), cols["torque"][:k])])
    w.step(16)
    assert np.max(np.abs(hip._aux["hlo_world_pos"][:k] - w.world_pos) / np.maximum(np.abs(w.world_pos), 1e-9)) <= 1e-9
    hip.close()
    ref.close()


def test_the_build_time_cli_object_is_installed_as_it_is(tmp_path):
    """`python -m elodin_amd.stablehlo module.mlir --slots slots.json -o pipe.so` (what a Rust WorldExec::Hip runs
    once per world) -> the object + manifest; a fresh executor binds the manifest's columns and installs the object without
    tracing or compiling anything."""
    from tests.golden import hlo_world_builder as hb
    text, slots = hb.three_body_world()
    (tmp_path / "tick.mlir").write_text(text)
    names = {str(L.component_id(c)): c for c, _, _ in slots}
    meta = {"arg_ids": [L.component_id(c) for c, _, _ in slots], "ret_ids": [L.component_id(c) for c, _, _ in slots], "names": names,
            "arg_slots": [{"component_id": L.component_id(c), "shape": s, "entity_axis_elided": e} for c, s, e in slots]}      # ExecMetadata, exec.rs:17-29
    (tmp_path / "slots.json").write_text(json.dumps(meta))
    out = tmp_path / "pipe.so"
    res = subprocess.run([sys.executable, "-m", "elodin_amd.stablehlo", str(tmp_path / "tick.mlir"), "--slots", str(tmp_path / "slots.json"), "-o", str(out), "--mode", "world"],
                         capture_output=True, text=True, cwd=str(L.PKG.parent))
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line["mode"] == "world" and out.exists()
    program, manifest = sh.load_world(out)
    assert [c["component_id"] for c in manifest["columns"]] == [L.component_id(c["component"]) for c in manifest["columns"]]
    _, _, _, row, g = W.three_body("world")
    n = 64
    hip = _exec(program, {c: np.tile(v[None, :], (n, 1)) for c, v in row.items()}, n)
    hip.run(100)
    worst = W.three_body_errors(hip._aux, g, 100)
    print("CLI-built object, tick 100 vs G1:", worst, "build:", manifest["build"])
    assert max(worst) <= 1e-9
    hip.close()


@pytest.mark.parametrize("arith", ["reference", "relaxed"])
def test_ball_whole_world_module_with_jax_random_reproduces_g2_on_the_gpu(arith):
    """examples/ball's singleton world as the reference would dump it, jax.random's threefry in u32 arithmetic included (sample_wind runs
    every tick): 100 ticks of G2 — wind, position, velocity, acceleration, force — through the generated kernel, <= 1e-9.  With
    arith="relaxed" too: the integer pipeline of the draw stays exact inside a relaxed trace (the wind is the same draw), the
    float state keeps the bound."""
    system, manifest, widths, row, g = W.ball("auto", arith=arith)
    assert manifest["mode"] == "world" and manifest.get("arith", "reference") == arith
    n = 70
    hip = _exec(dsl.Program([system], dsl.Pipe([]), []), {c: np.tile(v[None, :], (n, 1)) for c, v in row.items()}, n)
    worst = 0.0
    for r in range(1, 101):
        hip.run(1)
        worst = max(worst, W.ball_errors(hip._aux, g, r))
        assert hip._aux["hlo_tick"][0, 0] == r
    print(f"ball whole-world module (jax.random in the tick, arith = {arith}), 100 ticks vs G2: {worst:.2e}")
    assert worst <= 1e-9
    assert np.array_equal(hip._aux["hlo_world_pos"], np.repeat(hip._aux["hlo_world_pos"][:1], n, axis=0))
    hip.close()


def test_the_references_first_tick_checkpoint_harness_with_this_backend_in_cranelifts_place(tmp_path):
    """libs/cranelift-mlir/tests/checkpoint_test.rs's layout: a debug directory holding stablehlo.mlir, input_<i>.bin and
    xla_output_<i>.bin (here written from the golden CSV: G1's row 0 as the inputs, row 1 as the outputs XLA produced).
    `python -m elodin_amd.stablehlo --checkpoint DIR` compiles the module, runs the first tick on the GPU, writes hip_output_<i>.bin
    and compares."""
    from tests.golden import hlo_world_builder as hb
    text, slots = hb.three_body_world()
    (tmp_path / "stablehlo.mlir").write_text(text)
    _, _, _, row, g = W.three_body("world")
    order = [c for c, _, _ in slots]
    for k, c in enumerate(order):
        np.asarray(row["hlo_" + c], dtype=np.int64 if c == "tick" else np.float64).tofile(tmp_path / f"input_{k}.bin")
    want = {"tick": np.array([1], dtype=np.int64), "simulation_time_step": np.asarray(row["hlo_simulation_time_step"]),
            "inertia": np.asarray(row["hlo_inertia"])}
    for c, _ in W.BODY[:4]:
        want[c] = np.concatenate([g[f"{e}.{c}"][1] for e in "abc"])
    for k, c in enumerate(order):
        np.asarray(want[c]).tofile(tmp_path / f"xla_output_{k}.bin")
    ids = [{"index": k, "component_id": L.component_id(c), "byte_size": int(np.asarray(want[c]).nbytes)} for k, c in enumerate(order)]
    (tmp_path / "checkpoint.json").write_text(json.dumps({"inputs": ids, "outputs": ids, "num_output_slots": len(ids)}))      # cranelift_exec.rs:211-243
    res = subprocess.run([sys.executable, "-m", "elodin_amd.stablehlo", "--checkpoint", str(tmp_path)], capture_output=True, text=True, cwd=str(L.PKG.parent))
    assert res.returncode == 0, (res.stdout[-500:], res.stderr[-1500:])
    rep = json.loads(res.stdout.strip().splitlines()[-1])
    assert rep["ok"] and rep["mode"] == "lane" and len(rep["outputs"]) == 7          # auto: one lane per entity, the fold's reads inside the wavefront
    assert rep["outputs"][0] == {"index": 0, "against": "xla_output", "equal": True}
    assert max(o.get("max_rel_err", 0.0) for o in rep["outputs"]) <= 1e-12
    assert np.array_equal(np.fromfile(tmp_path / "hip_output_0.bin", dtype=np.int64), [1])
    assert (tmp_path / "hip_checkpoint.json").exists()


def test_three_body_world_one_lane_per_entity_exchange_inside_the_wavefront_on_the_gpu(tmp_path):
    """mode "auto" on the three-body module: one lane per ENTITY, a world = 4 consecutive rows, the edge_fold's target rows read from
    the other lanes of the world with ds_bpermute (`lane_read`).  4,096 worlds: world 0 flies G1's initial state (bit for bit over its
    100 ticks), the others perturbed states checked against the C oracle; the CLI's default object (auto) is what runs."""
    from tests.golden import hlo_world_builder as hb
    text, slots = hb.three_body_world()
    (tmp_path / "tick.mlir").write_text(text)
    (tmp_path / "slots.json").write_text(json.dumps({"inputs": [{"component": c, "shape": s_, "entity_axis_elided": e} for c, s_, e in slots], "rows": 16384}))
    res = subprocess.run([sys.executable, "-m", "elodin_amd.stablehlo", str(tmp_path / "tick.mlir"), "--slots", str(tmp_path / "slots.json"), "-o", str(tmp_path / "pipe.so")],
                         capture_output=True, text=True, cwd=str(L.PKG.parent))
    assert res.returncode == 0, res.stderr[-2000:]
    program, manifest = sh.load_world(tmp_path / "pipe.so")
    assert (manifest["mode"], manifest["rows_per_world"], manifest["entities_per_world"]) == ("lane", 4, 3)
    g = W.gu.load("three_body")
    S, worlds = 4, 4096
    cols = W.strided_world_columns(g, "abc", S, worlds)
    rng = np.random.default_rng(23)
    body_rows = np.array([w_ * S + i for w_ in range(1, worlds) for i in range(3)])
    cols["hlo_world_pos"][body_rows, 4:6] += rng.uniform(-0.05, 0.05, (len(body_rows), 2))
    cols["hlo_world_vel"][body_rows, 3:5] += rng.uniform(-0.05, 0.05, (len(body_rows), 2))
    start = {k: v.copy() for k, v in cols.items()}
    hip = _exec(program, cols, S * worlds)
    for r in range(1, 101):
        hip.run(1)
        for c, w in W.BODY[:4]:
            for i, e in enumerate("abc"):
                assert np.array_equal(hip._aux["hlo_" + c][i], g[f"{e}.{c}"][r]), (r, c, e)
    dt = float(g["globals.simulation_time_step"][0, 0])
    src, dst = np.array([0, 1, 0, 1, 2, 2], dtype=np.uint32), np.array([1, 0, 2, 2, 0, 1], dtype=np.uint32)
    worst = 0.0
    for wd in (1, 17, 2048, 4095):
        sl = slice(wd * S, wd * S + 3)
        w = orc.OracleWorld(start["hlo_world_pos"][sl], start["hlo_world_vel"][sl], start["hlo_inertia"][sl], simulation_time_step=dt,
                            ops=[(orc.EFF_EDGE_GRAVITY_NEWTON, (6.6743e-11,), None)], edges=(src, dst))
        w.step(100)
        for c, ref in (("world_pos", w.world_pos), ("world_vel", w.world_vel), ("world_accel", w.world_accel), ("force", w.force)):
            worst = max(worst, float(np.max(np.abs(hip._aux["hlo_" + c][sl] - ref) / np.maximum(np.max(np.abs(ref), axis=1, keepdims=True), 1e-300))))
    print(f"three-body worlds, one lane per entity (lane_read): G1 bit for bit; perturbed worlds vs the oracle {worst:.2e}")
    assert worst <= 1e-9
    hip.close()
    # a host without Python reads "rows_per_world" from the manifest; should it get the row count wrong, the C ABI itself refuses:
    # the object exports the rows a world occupies and sixdof_set_custom_pipe checks the handle against it
    program._traced.rows_multiple = 0                                    # (switch off the Python-side check of the same thing)
    with pytest.raises(ValueError, match="lays a world out as 4 consecutive rows; 6 rows are not a whole number of worlds"):
        _exec(program, {k: v[:6].copy() for k, v in start.items()}, 6)


def test_ten_body_solar_system_world_in_lane_mode_on_the_gpu():
    """examples/n-body's 10-body world (90 edges, softened fold): one lane per entity, a world = 16 rows, every fold target an exchange
    read; 1,024 worlds x 240 hourly ticks against the C oracle's sequential fold."""
    from tests import solar_util as su
    from tests.golden import hlo_world_builder as hb
    d, pos, vel, inertia = su.load()
    n = pos.shape[0]
    text, slots = hb.nbody_world(n, su.K_SQUARED, su.SOFTENING_AU2)
    system, manifest = sh.world_system(text, slots, mode="auto")
    assert (manifest["mode"], manifest["rows_per_world"]) == ("lane", 16)
    S, worlds = 16, 1024
    rows = S * worlds

    def lay(a, fill):
        out = np.tile(np.asarray(fill, dtype=np.float64), (rows, 1))
        for w_ in range(worlds):
            out[w_ * S:w_ * S + n] = a
        return out
    cols = {"hlo_tick": np.zeros((rows, 1)), "hlo_simulation_time_step": np.full((rows, 1), su.DT), "hlo_world_pos": lay(pos, [0, 0, 0, 1.0, 0, 0, 0]),
            "hlo_world_vel": lay(vel, np.zeros(6)), "hlo_inertia": lay(inertia, np.ones(7)), "hlo_world_accel": np.zeros((rows, 6)), "hlo_force": np.zeros((rows, 6))}
    hip = _exec(dsl.Program([system], dsl.Pipe([]), []), cols, rows, ticks_per_launch=24)
    hip.run(240)
    w = orc.OracleWorld(pos, vel, inertia, simulation_time_step=su.DT, ops=[(orc.EFF_ALLPAIRS_GRAVITY_SOFTENED, (su.K_SQUARED, su.SOFTENING_AU2), None)])
    w.step(240)
    worst = 0.0
    for c, ref in (("world_pos", w.world_pos), ("world_vel", w.world_vel), ("world_accel", w.world_accel), ("force", w.force)):
        for w_ in (0, 511, 1023):
            got = hip._aux["hlo_" + c][w_ * S:w_ * S + n]
            worst = max(worst, float(np.max(np.abs(got - ref) / np.maximum(np.max(np.abs(ref), axis=1, keepdims=True), 1e-300))))
        assert np.array_equal(hip._aux["hlo_" + c][:n], hip._aux["hlo_" + c][1023 * S:1023 * S + n])          # identical worlds, identical rows
    print(f"10-body solar system, 1,024 worlds x 240 ticks, lane mode vs the oracle: {worst:.2e}")
    assert worst <= 1e-9
    hip.close()


@pytest.mark.parametrize("seed,n,rolled", [(s_, 3 + s_ % 3, False) for s_ in (101, 102, 104, 106, 107, 108, 110, 113)] +
                         [(s_, 3 + s_ % 3, True) for s_ in (100, 103, 104, 105, 107, 114, 115)])
def test_random_modules_with_reads_between_entities_through_the_generated_kernel(seed, n, rolled, monkeypatch):
    """tests/hlo_fuzz.py's modules with JOINS (constant-table gathers along the entity axis, re-stacked per source) in lane mode on the
    GPU: `lane_read` with per-entity source tables as one ds_bpermute per 32-bit half, worlds of 4 and 8 rows at every position of a
    wavefront, the last wavefront only partly filled.  Against the numpy walker of the same traced program: 1e-12 of each result's
    scale (the device's libm differs from numpy's in the last place), NaNs in the same places."""
    from tests import hlo_fuzz
    from tests.test_stablehlo_world import walk
    if rolled:      # every counted while of two trips or more stays a loop: carried values the body hands back untouched leave the loop
        # state, an edge_fold-shaped scan reads its slot's targets through a table indexed by the counter (lane_read_dyn)
        monkeypatch.setattr(sh._LaneEval, "ROLL_MIN_TRIPS", 2)
        monkeypatch.setattr(sh._LaneEval, "ROLL_MIN_NODES", 1)
    text, slots, out_slots = hlo_fuzz.make(seed, n, exchange=True)
    system, manifest = sh.world_system(text, slots, out_slots, mode="lane")
    S = manifest.get("rows_per_world", n)
    assert manifest["exchange_reads"] > 0 and S in (4, 8)
    if rolled:
        from elodin_amd import codegen
        src = codegen.generate_source(dsl.Program([system], dsl.Pipe([]), []).trace({c["column"]: c["width"] for c in manifest["columns"]}), "float64", 2)
        assert "#pragma unroll 1" in src and f") * {S} + static_cast<int>(threadIdx.x" in src
    worlds = 24 if S == 4 else 12                        # 96 rows: one full wavefront + half of a second
    vals = [hlo_fuzz.inputs(seed + 1000 * w_, n) for w_ in range(worlds)]
    widths = {c["column"]: c["width"] for c in manifest["columns"]}
    cols = {}
    for k in vals[0]:
        per_world = [np.asarray(v[k], dtype=np.float64) for v in vals]
        if np.ndim(vals[0][k]) == 0:
            cols["hlo_" + k] = np.concatenate([np.tile(p.reshape(1, -1), (S, 1)) for p in per_world])
        else:
            blocks = []
            for p in per_world:
                b = np.zeros((S, p.reshape(n, -1).shape[1]))
                b[:n] = p.reshape(n, -1)
                blocks.append(b)
            cols["hlo_" + k] = np.concatenate(blocks)
    rows = S * worlds
    for c, w in widths.items():
        cols.setdefault(c, np.zeros((rows, w)))
    host = {k: v.copy() for k, v in cols.items()}
    walk(system, widths, host, 1)
    hip = _exec(dsl.Program([system], dsl.Pipe([]), []), cols, rows)
    hip.run(1)
    worst = 0.0
    for name, shape, _ in out_slots:
        c = "hlo_" + name
        got, want = np.asarray(hip._aux[c], dtype=np.float64), host[c]
        for w_ in range(worlds):
            g, e = got[w_ * S:w_ * S + n], want[w_ * S:w_ * S + n]
            assert np.array_equal(np.isnan(g), np.isnan(e)), (seed, name, w_)
            fin = np.isfinite(e)
            assert np.array_equal(g[~fin & ~np.isnan(e)], e[~fin & ~np.isnan(e)]), (seed, name, w_)
            if fin.any():
                worst = max(worst, float(np.max(np.abs(g[fin] - e[fin])) / max(1.0, float(np.max(np.abs(e[fin]))))))
    print(f"seed {seed}{' (loops kept)' if rolled else ''}: n {n}, rows_per_world {S}, {manifest['exchange_reads']} exchange reads, worst {worst:.2e}")
    assert worst <= 1e-12
    hip.close()


@pytest.mark.parametrize("n,stride", [(20, 32), (35, 64)])
def test_worlds_of_32_and_64_rows_in_lane_mode_on_the_gpu(n, stride):
    """Lane exchange up to a whole wavefront per world: 20 bodies in 32 rows and 35 bodies in 64 rows (the complete gravity graph,
    380 / 1,190 edges), source tables as bytes in constant memory.  129 worlds (the last wavefront partly filled for 32 rows) x 48
    ticks against the C oracle's sequential fold."""
    from tests.golden import hlo_world_builder as hb
    from tests.test_stablehlo_world import _random_cluster
    K, EPS, DT = 2.9591220828e-4, 1e-6, 0.5
    pos, vel, inertia = _random_cluster(n)
    text, slots = hb.nbody_world(n, K, EPS)
    system, manifest = sh.world_system(text, slots, mode="auto")
    assert (manifest["mode"], manifest["rows_per_world"]) == ("lane", stride)
    worlds = 129
    rows = stride * worlds

    def lay(a, fill):
        out = np.tile(np.asarray(fill, dtype=np.float64), (rows, 1))
        for w_ in range(worlds):
            out[w_ * stride:w_ * stride + n] = a
        return out
    cols = {"hlo_tick": np.zeros((rows, 1)), "hlo_simulation_time_step": np.full((rows, 1), DT), "hlo_world_pos": lay(pos, [0, 0, 0, 1.0, 0, 0, 0]),
            "hlo_world_vel": lay(vel, np.zeros(6)), "hlo_inertia": lay(inertia, np.ones(7)), "hlo_world_accel": np.zeros((rows, 6)), "hlo_force": np.zeros((rows, 6))}
    hip = _exec(dsl.Program([system], dsl.Pipe([]), []), cols, rows, ticks_per_launch=12)
    tm = hip.run(48)
    w = orc.OracleWorld(pos, vel, inertia, simulation_time_step=DT, ops=[(orc.EFF_ALLPAIRS_GRAVITY_SOFTENED, (K, EPS), None)])
    w.step(48)
    worst = 0.0
    for c, ref in (("world_pos", w.world_pos), ("world_vel", w.world_vel), ("world_accel", w.world_accel), ("force", w.force)):
        for w_ in (0, 64, 128):
            got = hip._aux["hlo_" + c][w_ * stride:w_ * stride + n]
            worst = max(worst, float(np.max(np.abs(got - ref) / np.maximum(np.max(np.abs(ref), axis=1, keepdims=True), 1e-300))))
        assert np.array_equal(hip._aux["hlo_" + c][:n], hip._aux["hlo_" + c][128 * stride:128 * stride + n])
    print(f"{n}-body world, rows_per_world {stride}, 129 worlds x 48 ticks vs the oracle: {worst:.2e}; {tm.kernel_device_ms / 48 * 1e3:.1f} us per tick")
    assert worst <= 1e-9
    hip.close()


@pytest.mark.parametrize("fast", [False, True])
def test_whole_world_ticks_build_in_float32_too(fast):
    """`--dtype float32 [--fast-math]` of the CLI: the 10-body solar system in lane mode (exchange reads of floats: one ds_bpermute
    each) and the three-body world with one lane per world, 48 / 100 ticks against the f64 oracle / G1 at single-precision accuracy."""
    from tests import solar_util as su
    from tests.golden import hlo_world_builder as hb
    d, pos, vel, inertia = su.load()
    n = pos.shape[0]
    text, slots = hb.nbody_world(n, su.K_SQUARED, su.SOFTENING_AU2)
    system, manifest = sh.world_system(text, slots, mode="auto")
    S, worlds = manifest["rows_per_world"], 256
    rows = S * worlds

    def lay(a, fill):
        out = np.tile(np.asarray(fill, dtype=np.float64), (rows, 1))
        for w_ in range(worlds):
            out[w_ * S:w_ * S + n] = a
        return out
    cols = {"hlo_tick": np.zeros((rows, 1)), "hlo_simulation_time_step": np.full((rows, 1), su.DT), "hlo_world_pos": lay(pos, [0, 0, 0, 1.0, 0, 0, 0]),
            "hlo_world_vel": lay(vel, np.zeros(6)), "hlo_inertia": lay(inertia, np.ones(7)), "hlo_world_accel": np.zeros((rows, 6)), "hlo_force": np.zeros((rows, 6))}
    hip = _exec(dsl.Program([system], dsl.Pipe([]), []), cols, rows, ticks_per_launch=12, dtype=np.float32, fast_math=fast)
    hip.run(48)
    w = orc.OracleWorld(pos, vel, inertia, simulation_time_step=su.DT, ops=[(orc.EFF_ALLPAIRS_GRAVITY_SOFTENED, (su.K_SQUARED, su.SOFTENING_AU2), None)])
    w.step(48)
    worst = 0.0
    for c, ref in (("world_pos", w.world_pos), ("world_vel", w.world_vel)):
        for w_ in (0, 255):
            got = np.asarray(hip._aux["hlo_" + c][w_ * S:w_ * S + n], dtype=np.float64)
            worst = max(worst, float(np.max(np.abs(got - ref) / np.maximum(np.max(np.abs(ref), axis=1, keepdims=True), 1e-300))))
    hip.close()
    row = W.three_body("world")[3]
    g = W.three_body("world")[4]
    wsys, wman = sh.world_system(*hb.three_body_world(), mode="world")
    hip = _exec(dsl.Program([wsys], dsl.Pipe([]), []), {c: np.tile(v[None, :], (64, 1)) for c, v in row.items()}, 64, dtype=np.float32, fast_math=fast)
    hip.run(100)
    e3 = max(W.three_body_errors({k: np.asarray(v, dtype=np.float64) for k, v in hip._aux.items()}, g, 100))
    hip.close()
    print(f"float32{' fast-math' if fast else ''}: 10-body lane mode, 48 ticks vs the f64 oracle {worst:.2e}; three-body world, 100 ticks vs G1 {e3:.2e}")
    assert worst <= 2e-5 and e3 <= 2e-4


def test_a_program_that_never_looks_at_the_tick_replays_from_captured_graphs():
    """Batches of identical one-tick launches of the hand-written kernel replay from captured hipGraphs (SIXDOF_FLAG_USE_GRAPH).  A
    generated program qualifies when its code never looks at the absolute tick (layout bit 17: no `tick` leaf, no cadence, no
    window) — a whole-world StableHLO tick carries its own tick COLUMN.  configs[1] as a whole-world module: 96 launches, all
    replayed, the same bits as eager launches; a program that reads the kernel's tick stays eager."""
    from tests.golden import hlo_world_builder as hb
    n = 4096
    text, slots = hb.independent_bodies_world(n)
    system, manifest = sh.world_system(text, slots, mode="lane")
    w = workloads.independent_bodies(n)

    def run(use_graph):
        cols = {"hlo_tick": np.zeros((n, 1)), "hlo_simulation_time_step": np.full((n, 1), workloads.DT_120HZ), "hlo_world_pos": w["world_pos"].copy(),
                "hlo_world_vel": w["world_vel"].copy(), "hlo_world_accel": np.zeros((n, 6)), "hlo_force": np.zeros((n, 6)), "hlo_inertia": w["inertia"].copy(),
                "hlo_torque": w["body_torque"].copy()}
        hip = _exec(dsl.Program([system], dsl.Pipe([]), []), cols, n, use_graph=use_graph)
        t = hip.invoke_batch(96)
        hip.download()
        out = {k: np.array(v) for k, v in hip._aux.items()}
        hip.close()
        return t, out
    tg, og = run(True)
    te, oe = run(False)
    assert tg.graph_launches == 96 and te.graph_launches == 0 and tg.launches == te.launches == 96
    for k in oe:
        assert np.array_equal(og[k], oe[k]), k
    assert np.all(og["hlo_tick"] == 96)

    @dsl.system(x=1)
    def counts(x, tick):
        return {"x": x + tick}
    hip = _exec(dsl.Program([counts], dsl.Pipe([]), []), {"x": np.zeros((n, 1))}, n, use_graph=True)
    t = hip.invoke_batch(96)
    hip.download()
    assert t.graph_launches == 0 and np.all(hip._aux["x"] == 96 * 97 / 2)
    hip.close()


def test_relaxed_arithmetic_module_on_the_gpu():
    """BASELINE configs[1] as a whole-world module under world_system(arith="relaxed") (dsl.relaxed_arithmetic: finite values assumed,
    one division per denominator, v_rcp / v_rsq seeds + Newton, a * b + c contracted — the forms the hand-written kernel uses) against
    the C oracle over 64 ticks on 4,096 rows: inside BASELINE's 1e-9 (vector-scaled like tests/parity.py; it measures ~1e-14), tick
    column exact, and NOT the reference's last bits — which the default build of the same module is held to right beside it
    (<= 1e-12: every operation the oracle's, in its order)."""
    from tests.golden import hlo_world_builder as hb
    keep = None
    n, ticks = 4096 + 37, 64                  # a ragged last wavefront
    text, slots = hb.independent_bodies_world(n)
    w = workloads.independent_bodies(n)
    o = orc.OracleWorld(w["world_pos"], w["world_vel"], w["inertia"], simulation_time_step=workloads.DT_120HZ,
                        ops=[(orc.EFF_UNIFORM_GRAVITY, (0.0, 0.0, -9.81), None), (orc.EFF_BODY_TORQUE, (), w["body_torque"])])
    o.step(ticks)
    figures = {}
    # one_world: the executor is ONE world, so the Globals columns (tick, dt: replicated per row) are read once per wavefront —
    # same values, so the reference build stays bit for bit what it was; every row still gets the new tick stored
    for arith, one_world, bound in (("reference", False, 1e-12), ("reference", True, 1e-12), ("relaxed", True, 1e-9)):
        system, manifest = sh.world_system(text, slots, mode="lane", arith=arith, one_world=one_world)
        assert manifest.get("arith", "reference") == arith and manifest.get("one_world", False) == one_world
        cols = {"hlo_tick": np.zeros((n, 1)), "hlo_simulation_time_step": np.full((n, 1), workloads.DT_120HZ), "hlo_world_pos": w["world_pos"].copy(),
                "hlo_world_vel": w["world_vel"].copy(), "hlo_world_accel": np.zeros((n, 6)), "hlo_force": np.zeros((n, 6)), "hlo_inertia": w["inertia"].copy(),
                "hlo_torque": w["body_torque"].copy()}
        hip = _exec(dsl.Program([system], dsl.Pipe([]), []), cols, n, use_graph=True)
        t = hip.invoke_batch(ticks)
        assert t.graph_launches == t.launches == ticks                 # replayed like the headline
        hip.download()
        got = {k: np.array(v) for k, v in hip._aux.items()}
        hip.close()
        assert np.all(got["hlo_tick"] == ticks)
        worst = {}
        for c, ref in (("world_pos", o.world_pos), ("world_vel", o.world_vel), ("world_accel", o.world_accel), ("force", o.force)):
            g = got["hlo_" + c]
            halves = ((slice(0, 4), slice(4, 7)) if c == "world_pos" else (slice(0, 3), slice(3, 6)))
            worst[c] = float(max(np.max(np.abs(g[:, h] - ref[:, h]) / np.maximum(np.max(np.abs(ref[:, h]), axis=1, keepdims=True), 1e-300)) for h in halves))
            assert worst[c] < bound, (arith, c, worst[c])
        figures[arith + ("+one_world" if one_world else "")] = worst
        if arith == "reference":
            keep = got if not one_world else keep
            assert all(np.array_equal(got[k], keep[k]) for k in got)
    assert max(figures["relaxed+one_world"].values()) > 0.0
    print("whole-world configs[1] module vs the oracle after 64 ticks (vector-scaled): " + json.dumps(figures))


@pytest.mark.parametrize("mode", ["world", "lane"])
def test_three_body_world_module_with_relaxed_arithmetic_stays_inside_1e_9_of_g1(mode):
    """The assembled three-body tick (gathers, `while` + `dynamic_slice` fold, `call`s; in lane mode the fold's targets are ds_bpermute
    exchanges) built with arith="relaxed": 100 ticks against the reference's golden CSVs, vector-scaled AND element-wise <= 1e-9 (it
    measures ~1e-13 — the default build of the same module is bit for bit), tick exact."""
    system, manifest, widths, row, g = W.three_body(mode, arith="relaxed")
    assert manifest["mode"] == mode and manifest["arith"] == "relaxed"
    if mode == "world":
        n = 64
        cols = {c: np.tile(v[None, :], (n, 1)) for c, v in row.items()}
        view = lambda aux: aux
    else:
        S, worlds = manifest["rows_per_world"], 16
        n = S * worlds
        cols = W.strided_world_columns(g, "abc", S, worlds)
        view = lambda aux: {"hlo_" + c: np.concatenate([aux["hlo_" + c][i] for i in range(3)])[None, :] for c, _ in W.BODY[:4]}
    hip = _exec(dsl.Program([system], dsl.Pipe([]), []), cols, n)
    worst = [0.0, 0.0]
    for r in range(1, 101):
        hip.run(1)
        e = W.three_body_errors(view(hip._aux), g, r)
        worst = [max(worst[0], e[0]), max(worst[1], e[1])]
        assert hip._aux["hlo_tick"][0, 0] == r
    print(f"three-body whole-world module ({mode}), relaxed arithmetic, 100 ticks vs G1: {worst[0]:.2e} (vector-scaled), {worst[1]:.2e} (element-wise)")
    assert 0.0 < worst[0] <= 1e-9 and worst[1] <= 1e-9, worst
    hip.close()
