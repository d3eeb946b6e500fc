"""The C++ column store behind the C ABI (csrc/world.cpp = libs/nox-py/src/world.rs for this path), no GPU."""
import ctypes as C

import numpy as np

from elodin_amd import _lib as L


def _insert(lib, w, eid, name, arr, prim=L.PRIM_F64):
    arr = np.ascontiguousarray(arr)
    dims = (C.c_uint64 * 2)(arr.shape[0] if arr.ndim else 1, 0)
    return lib.sixdof_world_insert(w, eid, name.encode(), prim, dims, 1 if arr.ndim else 0, arr.ctypes.data, arr.nbytes)


def test_globals_ids_and_column_bytes():
    lib = L.lib()
    w = C.c_void_p(lib.sixdof_world_create())
    assert lib.sixdof_world_entity_len(w) == 1                     # entity 0 = Globals (world.rs:174-183)
    assert lib.sixdof_world_time_step(w) == 0.008333333            # DEFAULT_TIME_STEP = 1e9/120 ns (world.rs:20)
    a, b = lib.sixdof_world_spawn(w), lib.sixdof_world_spawn(w)
    assert (a, b) == (1, 2)
    assert _insert(lib, w, a, "world_pos", np.array([0, 0, 0, 1.0, 1, 2, 3])) == L.OK
    assert _insert(lib, w, b, "world_pos", np.array([0, 0, 0, 1.0, 4, 5, 6])) == L.OK
    assert _insert(lib, w, b, "world_pos", np.zeros(6)) == L.ERR_VALUE_SIZE_MISMATCH     # Error::ValueSizeMismatch
    assert b"value size mismatch" in lib.sixdof_world_last_error(w)
    assert _insert(lib, w, 7, "world_pos", np.zeros(7)) == L.ERR_INVALID_ARGUMENT        # entity never spawned
    c = L.Column()
    assert lib.sixdof_world_column(w, L.component_id("world_pos"), C.byref(c)) == L.OK
    assert c.n_rows == 2 and c.dims[0] == 7 and [c.entity_ids[0], c.entity_ids[1]] == [1, 2]
    raw = C.string_at(c.host_ptr, 2 * 56)                          # row-major little-endian rows, spawn order
    assert np.frombuffer(raw, dtype="<f8").reshape(2, 7)[1].tolist() == [0, 0, 0, 1, 4, 5, 6]
    assert lib.sixdof_world_column(w, L.component_id("nope"), C.byref(c)) == L.ERR_COMPONENT_NOT_FOUND
    # components iterate in ascending ComponentId (BTreeMap order = output slot order)
    ids = (C.c_uint64 * 8)()
    n = lib.sixdof_world_components(w, ids, 8)
    assert list(ids[:n]) == sorted(ids[:n]) and L.component_id("tick") in ids[:n]
    lib.sixdof_world_destroy(w)


def test_rates_quantise_dt_and_tick_column():
    lib = L.lib()
    w = C.c_void_p(lib.sixdof_world_create())
    assert lib.sixdof_world_set_rates(w, 1000.0, 0.0) == L.OK and lib.sixdof_world_time_step(w) == 0.001
    assert lib.sixdof_world_set_rates(w, 120.0, 40.0) == L.OK and lib.sixdof_world_ticks_per_telemetry(w) == 3
    assert lib.sixdof_world_set_rates(w, 120.0, 50.0) == L.ERR_INVALID_ARGUMENT
    assert b"must evenly divide" in lib.sixdof_world_last_error(w)
    assert lib.sixdof_world_set_rates(w, -1.0, 0.0) == L.ERR_INVALID_ARGUMENT
    c = L.Column()
    lib.sixdof_world_column(w, L.component_id("simulation_time_step"), C.byref(c))
    assert np.frombuffer(C.string_at(c.host_ptr, 8), dtype="<f8")[0] == 0.008333333   # set_globals wrote the column
    lib.sixdof_world_advance_tick(w, 5)
    lib.sixdof_world_column(w, L.component_id("tick"), C.byref(c))
    assert c.prim_type == L.PRIM_U64 and np.frombuffer(C.string_at(c.host_ptr, 8), dtype="<u8")[0] == 5
    assert lib.sixdof_world_tick(w) == 5
    lib.sixdof_world_destroy(w)
