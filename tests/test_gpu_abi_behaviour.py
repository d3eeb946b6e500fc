"""Error behaviour and lifetime of the C ABI on a real device (mirrors nox-py `Error` -> Python exceptions)."""
import ctypes as C
import threading

import numpy as np
import pytest

import elodin_amd as ea
from elodin_amd import _lib as L
from elodin_amd import workloads
from tests import parity

pytestmark = pytest.mark.gpu


def _desc(n, **kw):
    d = L.Desc()
    d.struct_size = C.sizeof(L.Desc)
    d.n_entities = n
    d.simulation_time_step = workloads.DT_120HZ
    d.ticks_per_launch = 1
    for k, v in kw.items():
        setattr(d, k, v)
    return d


def test_create_rejects_bad_descriptors():
    lib = L.lib()
    h = C.c_void_p()
    d = _desc(4)
    d.struct_size = 8
    assert lib.sixdof_create(C.byref(d), C.byref(h)) == L.ERR_INVALID_ARGUMENT
    assert b"struct_size" in lib.sixdof_last_error(None)
    assert lib.sixdof_create(C.byref(_desc(4, integrator=7)), C.byref(h)) == L.ERR_INVALID_ARGUMENT
    assert lib.sixdof_create(C.byref(_desc(4, dtype=9)), C.byref(h)) == L.ERR_INVALID_ARGUMENT
    assert lib.sixdof_create(C.byref(_desc(4, device_ordinal=99)), C.byref(h)) == L.ERR_INVALID_ARGUMENT
    assert lib.sixdof_create(None, C.byref(h)) == L.ERR_INVALID_ARGUMENT
    lib.sixdof_destroy(None)   # no-op


def test_step_before_bind_and_missing_or_misshapen_columns():
    lib = L.lib()
    h = C.c_void_p()
    assert lib.sixdof_create(C.byref(_desc(3)), C.byref(h)) == L.OK
    assert lib.sixdof_step(h, 1, None) == L.ERR_COMPONENT_NOT_FOUND          # Error::ComponentNotFound
    assert lib.sixdof_upload(h) == L.ERR_COMPONENT_NOT_FOUND
    ids = np.arange(1, 4, dtype=np.uint64)
    pos = np.tile([0, 0, 0, 1.0, 0, 0, 0], (3, 1))

    def col(name, arr):
        c = L.Column()
        c.component_id = L.component_id(name)
        c.prim_type, c.ndim, c.n_rows = L.PRIM_F64, 1, arr.shape[0]
        c.dims[0] = arr.shape[1]
        c.entity_ids = ids.ctypes.data_as(C.POINTER(C.c_uint64))
        c.host_ptr = arr.ctypes.data
        return c
    one = (L.Column * 1)(col("world_pos", pos))
    assert lib.sixdof_bind_columns(h, one, 1) == L.ERR_COMPONENT_NOT_FOUND   # the other Body columns are missing
    assert b"world_vel" in lib.sixdof_last_error(h)
    bad = np.zeros((3, 5))
    five = (L.Column * 5)(col("world_pos", pos), col("world_vel", bad), col("world_accel", np.zeros((3, 6))),
                          col("force", np.zeros((3, 6))), col("inertia", np.ones((3, 7))))
    assert lib.sixdof_bind_columns(h, five, 5) == L.ERR_VALUE_SIZE_MISMATCH    # Error::ValueSizeMismatch
    lib.sixdof_destroy(h)


def test_python_surface_maps_errors_like_the_reference():
    w = workloads.independent_bodies(10)
    with pytest.raises(ValueError):      # unknown effector kind
        ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], effectors=[ea.Effector(77)])
    with pytest.raises(ValueError):      # five per-entity ops
        ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], effectors=[ea.Effector(L.EFF_UNIFORM_GRAVITY, (0, 0, -1))] * 5)
    with pytest.raises(ValueError):      # f32 pair effectors are not provided
        ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], dtype=np.float32,
                   effectors=[ea.Effector(L.EFF_ALLPAIRS_GRAVITY_SOFTENED, (1.0, 1e-3))]).run(1)
    with pytest.raises(ValueError):      # aux column of the wrong width
        ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"],
                   effectors=[ea.Effector(L.EFF_BODY_TORQUE, (), aux_name="t", aux=np.zeros((10, 3)))],
                   column_entity_ids={"t": np.arange(1, 8, dtype=np.uint64)})


def test_many_handles_and_two_threads_on_one_gpu():
    """One handle = one caller thread; different handles may run from different threads (cranelift_exec.rs:31-51)."""
    w = workloads.independent_bodies(3000)
    eff = workloads.gravity_torque_effectors(w["body_torque"])
    for _ in range(40):                  # create / destroy does not leak or wedge the device
        ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], effectors=eff).close()
    serial = ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], simulation_time_step=workloads.DT_120HZ, effectors=eff)
    serial.run(200)
    results = [None, None]

    def worker(k):
        ex = ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], simulation_time_step=workloads.DT_120HZ,
                        effectors=eff, ticks_per_launch=1 + 3 * k)
        for _ in range(10):
            ex.run(20)
        results[k] = ex.world_pos.copy()
        ex.close()
    ts = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert np.array_equal(results[0], serial.world_pos) and np.array_equal(results[1], serial.world_pos)


def test_c_world_bound_to_a_handle_end_to_end():
    """Pure C-ABI host: sixdof_world_* builds the columns, sixdof_bind_world hands them to the backend
    (what a non-Python host would do); K8 known answer test_six_dof_force through it."""
    lib = L.lib()
    w = C.c_void_p(lib.sixdof_world_create())

    def insert(eid, name, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        dims = (C.c_uint64 * 2)(arr.shape[0], 0)
        assert lib.sixdof_world_insert(w, eid, name.encode(), L.PRIM_F64, dims, 1, arr.ctypes.data, arr.nbytes) == L.OK
    static = lib.sixdof_world_spawn(w)                      # a scene object: world_pos only
    insert(static, "world_pos", [0, 0, 0, 1.0, 9, 9, 9])
    e1 = lib.sixdof_world_spawn(w)
    for name, val in (("world_pos", [0, 0, 0, 1.0, 0, 0, 0]), ("world_vel", [0] * 6), ("world_accel", [0] * 6),
                      ("force", [0] * 6), ("inertia", [1, 1, 1, 0, 0, 0, 1.0])):
        insert(e1, name, val)
    assert lib.sixdof_world_set_rates(w, 120.0, 0.0) == L.OK
    h = C.c_void_p()
    d = _desc(0, has_time_step=1, time_step=1.0 / 120.0)    # n_entities = 0: sized by the join
    assert lib.sixdof_create(C.byref(d), C.byref(h)) == L.OK
    assert lib.sixdof_bind_world(h, w) == L.OK
    op = (L.EffectorOp * 1)()
    op[0].kind = L.EFF_CONST_WRENCH
    op[0].p[3] = 1.0
    assert lib.sixdof_set_effectors(h, op, 1) == L.OK
    assert lib.sixdof_upload(h) == L.OK
    assert lib.sixdof_step(h, 120, None) == L.OK
    assert lib.sixdof_download(h, L.COL_ALL) == L.OK
    lib.sixdof_world_advance_tick(w, 120)
    c = L.Column()
    lib.sixdof_world_column(w, L.component_id("world_pos"), C.byref(c))
    pos = np.frombuffer(C.string_at(c.host_ptr, 2 * 56), dtype="<f8").reshape(2, 7)
    assert pos[0].tolist() == [0, 0, 0, 1, 9, 9, 9]                                        # static object untouched
    assert np.isclose(pos[1], [0, 0, 0, 1, 0.5, 0, 0], rtol=1e-5).all()                    # test_all.py:342-364
    t = C.c_uint64()
    lib.sixdof_get_tick(h, C.byref(t))
    assert t.value == 120 == lib.sixdof_world_tick(w)
    lib.sixdof_destroy(h)
    lib.sixdof_world_destroy(w)


def test_failure_sentinel_checkpoint_and_timings():
    w = workloads.independent_bodies(1000)
    eff = workloads.gravity_torque_effectors(w["body_torque"])
    inertia = w["inertia"].copy()
    inertia[[7, 500], 6] = 0.0                      # zero mass -> f/m = inf -> the rollout "crashes"
    ex = ea.HipExec(w["world_pos"], w["world_vel"], inertia, simulation_time_step=workloads.DT_120HZ, effectors=eff)
    assert not ex.nonfinite_rows().any()
    ex.run(3)
    bad = ex.nonfinite_rows()
    assert bad.sum() == 2 and bad[7] and bad[500]                          # only the broken rollouts are flagged
    assert np.isfinite(ex.world_pos[~bad]).all()
    # checkpoint / resume: 50 + 50 ticks == 100 ticks
    a = ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], simulation_time_step=workloads.DT_120HZ, effectors=eff)
    a.run(50)
    state = a.checkpoint()
    b = ea.HipExec(w["world_pos"] * 0 + [0, 0, 0, 1, 0, 0, 0], w["world_vel"] * 0, w["inertia"],
                   simulation_time_step=workloads.DT_120HZ, effectors=eff)
    b.restore(state)
    a.run(50)
    b.run(50)
    assert b.tick == 100 and all(np.array_equal(getattr(a, f), getattr(b, f)) for f in parity.FIELDS)
    t = a.last_timings()                                                   # profile.rs phases
    assert t.h2d_upload_ms > 0 and t.d2h_download_ms > 0 and t.kernel_invoke_ms > 0 and t.ticks == 50


def test_streaming_commit_equals_synchronous_run():
    """sixdof_download_async / SIXDOF_FLAG_ASYNC_STEP: batch i's columns reach the (page-locked) host buffers on a second
    stream while batch i+1 computes; what the consumer sees per batch is exactly what a synchronous run+download shows."""
    from elodin_amd import workloads
    w = workloads.independent_bodies(20_000)
    eff = workloads.gravity_torque_effectors(w["body_torque"])
    mk = lambda: ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], simulation_time_step=workloads.DT_120HZ,
                            effectors=eff, ticks_per_launch=4)
    a, b = mk(), mk()
    seen = []
    wall = a.run_streaming(7, 12, consume=lambda i: seen.append((i, a.world_pos.copy(), a.world_vel.copy(), a.world_accel.copy(), a.force.copy())))
    assert [s[0] for s in seen] == list(range(7)) and a.tick == 84 and wall > 0
    for i, pos, vel, acc, force in seen:
        b.run(12)
        assert np.array_equal(pos, b.world_pos) and np.array_equal(vel, b.world_vel), i
        assert np.array_equal(acc, b.world_accel) and np.array_equal(force, b.force), i
    # the handle is back in synchronous mode and keeps going from the same state
    a.run(5)
    b.run(5)
    assert np.array_equal(a.world_pos, b.world_pos)


def test_every_tick_streamed_to_the_host_equals_single_tick_runs():
    """sixdof_history_stream: the ring is two batches deep, batch i's ticks travel to page-locked host buffers while batch
    i+1 computes; every streamed tick equals what a one-tick-at-a-time run shows, across the ring wrap."""
    from elodin_amd import workloads
    w = workloads.independent_bodies(3000)
    eff = workloads.gravity_torque_effectors(w["body_torque"])
    mk = lambda k: ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], simulation_time_step=workloads.DT_120HZ,
                              effectors=eff, ticks_per_launch=k)
    a, b = mk(5), mk(1)
    got = {}

    def consume(i, first_tick, cols):
        for k in range(cols["world_pos"].shape[0]):
            got[first_tick + k] = {c: v[k].copy() for c, v in cols.items()}
    a.stream_history(5, 10, consume)             # 5 batches of 10 ticks, 2 launches of 5 ticks each
    assert sorted(got) == list(range(1, 51)) and a.tick == 50
    for t in range(1, 51):
        b.run(1)
        for c in ("world_pos", "world_vel", "world_accel", "force"):
            assert np.array_equal(got[t][c], getattr(b, c)), (t, c)
    a.download()
    assert np.array_equal(a.world_pos, b.world_pos)


def test_handle_lifecycles_return_their_device_and_pinned_memory():
    """Every buffer a handle owns goes back at destroy: columns, telemetry rings (body + program columns), the async
    snapshot, pair packs / partials / CSR tables, page-locked registrations.  Device-wide free memory (hipMemGetInfo)
    after many create -> use -> destroy cycles of every flavour must be where it was after the first cycle."""
    import ctypes as C
    from elodin_amd import dsl
    hip = C.CDLL("libamdhip64.so")

    def free_bytes():
        f, t = C.c_size_t(), C.c_size_t()
        assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value

    n = 200_000
    w = workloads.independent_bodies(n)
    eff = workloads.gravity_torque_effectors(w["body_torque"])

    @dsl.system(temp=4)
    def heat(temp, vel):
        return {"temp": temp * 0.999 + dsl.np.linalg.norm(vel.linear()) * 0.001}
    m = 2000
    ids = np.arange(1, m + 1, dtype=np.uint64)

    def cycle():
        ex = ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], effectors=eff, ticks_per_launch=8)
        ex._lib.sixdof_set_history(ex._h, 16)
        ex.run(16)
        ex.history("world_pos", 9, 16)
        ex.run_streaming(3, 8)
        ex.stream_history(2, 8)
        ex.close()
        ex = ea.HipExec(w["world_pos"], w["world_vel"], w["inertia"], dtype=np.float32, integrator=L.SEMI_IMPLICIT,
                        effectors=dsl.Program([heat], dsl.Pipe([]), []), columns={"temp": np.ones((n, 4))})
        ex._lib.sixdof_set_history(ex._h, 4)
        ex.run(4)
        ex.close()
        ex = ea.HipExec(w["world_pos"][:m], w["world_vel"][:m], w["inertia"][:m],
                        effectors=[ea.Effector(L.EFF_EDGE_GRAVITY_SOFTENED, (1e-3, 1e-2))], edges=(ids, np.roll(ids, 1)))
        ex.run(2)
        ex.close()
        ex = ea.HipExec(w["world_pos"][:m], w["world_vel"][:m], w["inertia"][:m],
                        effectors=[ea.Effector(L.EFF_ALLPAIRS_GRAVITY_SOFTENED, (1e-3, 1e-2))])
        ex.run(2)
        ex.close()

    cycle()                              # first use pays one-off costs (code objects, scratch, runtime pools)
    cycle()
    base = free_bytes()
    for _ in range(12):
        cycle()
    lost = base - free_bytes()
    per_cycle = (n * (7 + 6 * 4 + 7) * 8) * 2          # what one cycle allocates at least: columns + a ring, twice
    print(f"device memory after 12 more cycles: {lost / 2**20:+.1f} MiB (one cycle allocates > {per_cycle / 2**20:.0f} MiB)")
    assert lost < 32 * 2**20


@pytest.mark.parametrize("n", [3, 700])
def test_history_ring_on_pair_path_worlds_holds_every_tick(n):
    """the pair (edge_fold / all-pairs) kernels do not record in-line; with a ring enabled such worlds run one
    tick per launch and the live columns are snapshot into the ring on the device — history() must equal what single-tick
    runs leave in the columns (n = 3: the one-workgroup small-graph kernel, n = 700: pack / all-pairs / integrate)."""
    from elodin_amd.exec import Effector
    rng = np.random.default_rng(5)
    pos = np.concatenate([np.tile([0, 0, 0, 1.0], (n, 1)), rng.normal(size=(n, 3)) * 10.0], axis=1)
    vel = np.concatenate([np.zeros((n, 3)), rng.normal(size=(n, 3))], axis=1)
    inertia = np.concatenate([np.ones((n, 3)), np.zeros((n, 3)), rng.uniform(1.0, 5.0, (n, 1))], axis=1)
    eff = [Effector(L.EFF_ALLPAIRS_GRAVITY_SOFTENED, (0.5, 1e-2))]
    a = ea.HipExec(pos, vel, inertia, simulation_time_step=0.01, effectors=eff, ticks_per_launch=8)
    a.enable_history(16)
    a.run(12)
    b = ea.HipExec(pos, vel, inertia, simulation_time_step=0.01, effectors=eff)
    for tick in range(1, 13):
        b.run(1)
        for name in ("world_pos", "world_vel", "world_accel", "force"):
            assert np.array_equal(a.history(name, tick, tick)[0], getattr(b, name)), (name, tick)
    assert np.array_equal(a.world_pos, b.world_pos)          # recording did not change the flight


def test_history_ring_on_the_apollo_model_holds_every_tick():
    from elodin_amd.models import apollo
    ref = apollo.load_reference()
    d = apollo.default_params(ref)
    P = np.tile([d[k] for k in apollo.PARAM_NAMES], (5, 1))
    P[:, apollo.PARAM_NAMES.index("init_altitude_m")] += np.arange(5) * 10.0
    a = apollo.ApolloExec(P, ref=ref, ticks_per_launch=60)
    a.enable_history(32)
    a.run(30)
    b = apollo.ApolloExec(P, ref=ref, ticks_per_launch=1)
    for tick in range(1, 31):
        b.run(1)
        for name in ("world_pos", "world_vel", "world_accel", "force"):
            assert np.array_equal(a.history(name, tick, tick)[0], getattr(b, name)), (name, tick)
    with pytest.raises(KeyError):                            # the model's own columns are not recorded: a clear refusal
        a.history("apollo_state", 1, 1)
