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
