"""The gather's packing — equal zero-padded blocks around ONE all-gather — is C-ABI code (sixdof_gather_pack / _unpack in
csrc/campaign_comm.cpp) shared by sixdof_campaign_gather (RCCL) and elodin_amd.shard.gather_rows (torch.distributed).  RCCL with
N > 1 needs N GPUs; the packing does not: here it is unit-tested for uneven blocks, and driven through real multi-process
all-gathers over gloo for world sizes 2, 3 and 8.  Row order = run-id order (sample.py:149: row = idx)."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from elodin_amd import _lib as L
from elodin_amd import shard


def _table(rows, width, seed=0):
    return np.random.default_rng(seed).normal(size=(rows, width))


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("rows", [0, 1, 5, 30, 8193])
def test_pack_then_unpack_is_the_identity_and_pads_with_zeros(world, rows):
    width = 12
    table = _table(rows, width, seed=rows + world)
    pad_rows = int(L.lib().sixdof_gather_block_rows(rows, world))
    assert pad_rows == -(-rows // world)
    blocks = []
    for r in range(world):
        lo, hi = shard.shard_range(rows, world, r)
        b = shard.pack_block(table[lo:hi], rows, world, r)
        assert b.shape == (pad_rows, width)
        assert np.array_equal(b[: hi - lo], table[lo:hi]) and not b[hi - lo:].any()       # its rows first, zeros after
        blocks.append(b)
    got = shard.unpack_blocks(np.stack(blocks) if blocks else np.zeros((0, pad_rows, width)), width, rows, world)
    assert got.shape == (rows, width) and np.array_equal(got, table)                        # bit-identical, run-id order


def test_pack_refuses_a_block_that_is_not_the_ranks():
    table = _table(30, 4)
    with pytest.raises(ValueError, match="not rank 1's block"):
        shard.pack_block(table[:5], 30, 3, 1)                # rank 1 of 3 owns 10 rows
    dp = __import__("ctypes").POINTER(__import__("ctypes").c_double)
    out = np.empty((10, 4))
    assert L.lib().sixdof_gather_pack(table.ctypes.data_as(dp), 10, 4, 30, 3, 3, out.ctypes.data_as(dp)) == L.ERR_INVALID_ARGUMENT      # rank out of range
    assert L.lib().sixdof_gather_pack(table.ctypes.data_as(dp), 10, 4, 30, 0, 0, out.ctypes.data_as(dp)) == L.ERR_INVALID_ARGUMENT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank(rank, world, port, cases, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    for rows, width, unit, dtype in cases:
        table = _table(rows, width, seed=rows).astype(dtype)
        lo, hi = shard.shard_range(rows, world, rank, unit)
        got = shard.gather_rows(table[lo:hi], rows, unit=unit)
        assert got.dtype == table.dtype and got.shape == table.shape
        assert np.array_equal(got, table), (rank, rows, width, unit)
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")


@pytest.mark.parametrize("world", [2, 3, 8])
def test_uneven_blocks_over_gloo_through_the_c_packing(world, tmp_path):
    """Every rank gathers 30-row and 8,193-row tables (neither divides by 2, 3 or 8 evenly... 30 by 8 leaves ranks with 4 and 3
    rows, 8,193 leaves one rank a row more) through shard.gather_rows: C packing, gloo all-gather, C unpacking; also a table of
    4-row units (whole-world shards) and an f32 table."""
    cases = [(30, 12, 1, np.float64), (8193, 14, 1, np.float64), (5, 3, 1, np.float64), (120, 6, 4, np.float64), (30, 5, 1, np.float32)]
    if world == 8:
        cases.append((3, 2, 1, np.float64))          # fewer rows than ranks: five ranks contribute nothing but padding
    mp.spawn(_rank, args=(world, _free_port(), cases, str(tmp_path)), nprocs=world, join=True)
    assert sorted(p.name for p in tmp_path.iterdir()) == sorted(f"ok{r}" for r in range(world))
