"""Campaign collectives of the C ABI (sixdof_comm_*, sixdof_campaign_broadcast / _gather; csrc/campaign_comm.cpp)."""
import ctypes as C
import multiprocessing as mp
import os

import numpy as np
import pytest

from elodin_amd import _lib as L
from elodin_amd import shard


def test_shard_range_matches_the_python_partition_and_covers_every_row():
    lib = L.lib()
    for n in (0, 1, 7, 8, 30, 8192, 8193, 32768):
        for world in (1, 2, 3, 4, 8):
            prev = 0
            for rank in range(world):
                lo, hi = C.c_uint64(), C.c_uint64()
                lib.sixdof_shard_range(n, world, rank, C.byref(lo), C.byref(hi))
                assert (lo.value, hi.value) == shard.shard_range(n, world, rank)
                assert lo.value == prev
                prev = hi.value
            assert prev == n


def test_comm_init_rejects_nonsense_without_touching_a_device():
    lib = L.lib()
    h = C.c_void_p()
    idbuf = (C.c_uint8 * 128)()
    assert lib.sixdof_comm_init(C.byref(h), idbuf, 0, 0, 0) == L.ERR_INVALID_ARGUMENT
    assert lib.sixdof_comm_init(C.byref(h), idbuf, 2, 2, 0) == L.ERR_INVALID_ARGUMENT
    assert lib.sixdof_comm_init(C.byref(h), None, 2, 0, 0) == L.ERR_INVALID_ARGUMENT
    assert lib.sixdof_campaign_broadcast(None, None, 0, 0) == L.ERR_INVALID_ARGUMENT
    if lib.sixdof_device_count() == 0:
        assert lib.sixdof_comm_init(C.byref(h), None, 1, 0, 0) == L.ERR_NO_DEVICE
        assert b"no such HIP device" in lib.sixdof_comm_last_error(None)


@pytest.mark.gpu
def test_single_rank_comm_is_a_copy():
    c = shard.CapiComm(None, 1, 0, 0)
    table = np.arange(30 * 17, dtype=np.float64).reshape(30, 17)
    assert np.array_equal(c.broadcast_table(table, table.shape), table)
    rows = np.random.default_rng(0).normal(size=(30, 12))
    assert np.array_equal(c.gather_rows(rows, 30), rows)
    with pytest.raises(L.BackendError, match="not this rank's block"):
        c.gather_rows(rows[:5], 30)
    c.close()


@pytest.mark.gpu
def test_one_rank_rccl_communicator_really_goes_through_rccl(monkeypatch):
    """SIXDOF_COMM_FORCE_RCCL=1: a ONE-rank RCCL communicator (dlopen, ncclGetUniqueId, ncclCommInitRank) and the
    collectives through ncclBroadcast / ncclAllGather on the GPU — the same code a multi-rank campaign runs, on the one
    GPU a gpurun box has."""
    monkeypatch.setenv("SIXDOF_COMM_FORCE_RCCL", "1")
    assert len(shard.CapiComm.unique_id()) == 128
    c = shard.CapiComm(None, 1, 0, 0)
    table = np.random.default_rng(1).normal(size=(8192, 17))
    got = c.broadcast_table(table, table.shape)
    assert np.array_equal(got, table)
    rows = np.random.default_rng(2).normal(size=(8192, 12))
    assert np.array_equal(c.gather_rows(rows, 8192), rows)
    assert np.array_equal(c.gather_rows(rows[:0].reshape(0, 12), 0), rows[:0])
    c.close()


def _rank_main(rank, world, id_path, out_path):
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    if rank == 0:
        cid = shard.CapiComm.unique_id()
        with open(id_path + ".tmp", "wb") as f:
            f.write(cid)
        os.replace(id_path + ".tmp", id_path)
    else:
        import time
        while not os.path.exists(id_path):
            time.sleep(0.01)
        cid = open(id_path, "rb").read()
    c = shard.CapiComm(cid, world, rank, rank)
    n, width = 101, 12
    table = np.arange(n * 17, dtype=np.float64).reshape(n, 17) if rank == 0 else None
    got = c.broadcast_table(table, (n, 17))
    lo, hi = shard.shard_range(n, world, rank)
    local = got[lo:hi, :width] * 2.0 + rank                    # something only this rank can produce
    allrows = c.gather_rows(local, n)
    np.save(out_path % rank, np.concatenate([got.ravel(), allrows.ravel()]))
    c.close()


@pytest.mark.gpu
def test_two_ranks_over_rccl_broadcast_and_gather_in_run_id_order(tmp_path):
    """One process per GPU, the id shipped through a file — the way a non-Python host would drive it."""
    if L.lib().sixdof_device_count() < 2:
        pytest.skip("needs two GPUs (the driver's multi-GPU tier); the single-rank and gloo tests cover the logic")
    world = 2
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_rank_main, args=(r, world, str(tmp_path / "id"), str(tmp_path / "out%d.npy"))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    n = 101
    table = np.arange(n * 17, dtype=np.float64).reshape(n, 17)
    want_rows = np.concatenate([table[slice(*shard.shard_range(n, world, r)), :12] * 2.0 + r for r in range(world)])
    for r in range(world):
        got = np.load(str(tmp_path / ("out%d.npy" % r)))
        assert np.array_equal(got[: n * 17].reshape(n, 17), table)
        assert np.array_equal(got[n * 17:].reshape(n, 12), want_rows)


def _shared_device_rank(rank, world, id_path, out_path):
    """Like _rank_main, but every rank opens its communicator on device 0; the outcome (rows or the refusal text) is written down."""
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    try:
        if rank == 0:
            cid = shard.CapiComm.unique_id()
            with open(id_path + ".tmp", "wb") as f:
                f.write(cid)
            os.replace(id_path + ".tmp", id_path)
        else:
            import time
            while not os.path.exists(id_path):
                time.sleep(0.01)
            cid = open(id_path, "rb").read()
        c = shard.CapiComm(cid, world, rank, 0)
        n, width = 101, 12                                       # 101 rows over 2 ranks: blocks of 51 and 50, one padded row on the wire
        table = np.arange(n * 17, dtype=np.float64).reshape(n, 17) if rank == 0 else None
        got = c.broadcast_table(table, (n, 17))
        lo, hi = shard.shard_range(n, world, rank)
        allrows = c.gather_rows(got[lo:hi, :width] * 2.0 + rank, n)
        np.save(out_path % rank, np.concatenate([got.ravel(), allrows.ravel()]))
        c.close()
    except L.BackendError as e:
        open((out_path % rank) + ".refused", "w").write(str(e))


@pytest.mark.gpu
def test_two_ranks_on_one_device_over_rccl_or_the_recorded_refusal(tmp_path):
    """A gpurun box has ONE GPU.  RCCL with two ranks on it either works — then the uneven 101-row broadcast + gather is checked
    exactly like the two-GPU test — or refuses the duplicate device at ncclCommInitRank (NCCL's "Duplicate GPU detected"); the
    refusal must come back through the C ABI as a status with RCCL's text, not as a hang or a crash, and is written to
    gpurun_out/rccl_two_ranks_one_gpu.txt.  (Uneven-block padding itself is covered on the CPU: tests/test_gather_packing.py.)"""
    if L.lib().sixdof_device_count() < 1:
        pytest.skip("needs a GPU")
    world = 2
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_shared_device_rank, args=(r, world, str(tmp_path / "id"), str(tmp_path / "out%d.npy"))) for r in range(world)]
    for p in procs:
        p.start()
    hung = False
    for p in procs:
        p.join(90)
        if p.is_alive():
            hung = True
            p.terminate()
            p.join(10)
    refusals = [(tmp_path / ("out%d.npy.refused" % r)).read_text() for r in range(world) if (tmp_path / ("out%d.npy.refused" % r)).exists()]
    from pathlib import Path
    log = Path(__file__).resolve().parent.parent / "gpurun_out"
    log.mkdir(exist_ok=True)
    if refusals:
        (log / "rccl_two_ranks_one_gpu.txt").write_text("refused (as expected of two ranks on one device):\n" + "\n".join(refusals) + "\n")
        assert all("ncclCommInitRank" in t or "comm_init" in t for t in refusals), refusals
        return
    assert not hung, "two ranks on one device neither finished nor were refused within 90 s"
    n = 101
    table = np.arange(n * 17, dtype=np.float64).reshape(n, 17)
    want_rows = np.concatenate([table[slice(*shard.shard_range(n, world, r)), :12] * 2.0 + r for r in range(world)])
    for r in range(world):
        got = np.load(str(tmp_path / ("out%d.npy" % r)))
        assert np.array_equal(got[: n * 17].reshape(n, 17), table)
        assert np.array_equal(got[n * 17:].reshape(n, 12), want_rows)
    (log / "rccl_two_ranks_one_gpu.txt").write_text("two RCCL ranks on one device ran: uneven 101-row broadcast + gather bit-identical\n")
