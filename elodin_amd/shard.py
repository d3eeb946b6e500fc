"""Multi-GPU plumbing for the shardable part of the path: independent bodies / Monte-Carlo rollouts.

Rows are independent (the reference runs rollouts as separate OS processes,
libs/monte-carlo/src/lib.rs:2083), so rank r owns a contiguous row block and steps it with no
per-step exchange.  The only collectives are campaign-level: a broadcast of the shared parameter
table from rank 0 and a gather of per-rollout result rows (RCCL over xGMI when the process group
is "nccl"; the CPU tests run the same code over gloo).  run_id <-> row mapping follows
libs/nox-py/python/elodin/monte_carlo/sample.py:149 (row = idx, run_id = "run_%07d", seed = idx+1).
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def shard_range(total_rows: int, world_size: int, rank: int, unit: int = 1) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of rank `rank`; blocks differ by at most one unit.  `unit`: rows that must stay on one rank — 1 for
    independent bodies / rollouts; the `rows_per_world` of a whole-world StableHLO tick in lane mode (elodin_amd/stablehlo.py), whose
    entities exchange data inside the wavefront: a Monte-Carlo of such worlds shards by WORLD.  (A C host calls sixdof_shard_range on
    total_rows / unit and multiplies.)"""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    if unit < 1 or total_rows % unit:
        raise ValueError(f"{total_rows} rows are not a whole number of {unit}-row units")
    base, extra = divmod(total_rows // unit, world_size)
    lo = rank * base + min(rank, extra)
    return lo * unit, (lo + base + (1 if rank < extra else 0)) * unit


def run_id(idx: int) -> str:
    return f"run_{idx:07d}"


def _dist():
    import torch.distributed as dist
    return dist


def max_over_ranks(seconds: float, device="cpu") -> float:
    """Timing reduction of the bench contract: MAX over ranks."""
    import torch
    dist = _dist()
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def broadcast_table(table: np.ndarray | None, shape, dtype=np.float64, src: int = 0, device="cpu") -> np.ndarray:
    """Rank `src` holds the campaign parameter table (n_runs x n_params); everyone gets a copy."""
    import torch
    dist = _dist()
    if not (dist.is_available() and dist.is_initialized()):
        return np.ascontiguousarray(table, dtype=dtype)
    t = torch.empty(tuple(shape), dtype=getattr(torch, np.dtype(dtype).name), device=device)
    if dist.get_rank() == src:
        t.copy_(torch.from_numpy(np.ascontiguousarray(table, dtype=dtype)))
    dist.broadcast(t, src=src)
    return t.cpu().numpy()


def gather_rows(local_rows: np.ndarray, total_rows: int, device="cpu", unit: int = 1) -> np.ndarray:
    """All-gather per-rollout result rows back into run-id order (shards are contiguous blocks; `unit` as in shard_range)."""
    import torch
    dist = _dist()
    local_rows = np.ascontiguousarray(local_rows)
    if not (dist.is_available() and dist.is_initialized()):
        return local_rows
    world, rank = dist.get_world_size(), dist.get_rank()
    if unit < 1 or total_rows % unit or local_rows.shape[0] % unit:
        raise ValueError(f"{total_rows} rows are not a whole number of {unit}-row units")
    # the packing is the C ABI's (sixdof_gather_pack / _unpack: equal zero-padded blocks, what sixdof_campaign_gather runs around
    # its ncclAllGather); only the transport is torch's.  A `unit` of rows travels as one wider row; f64 on the wire (exact for f32).
    if local_rows.dtype.kind in "iu" and local_rows.dtype.itemsize >= 8:
        raise TypeError("gather_rows moves rows as float64: 64-bit integer rows would not survive beyond 2^53 (gather them as 32-bit halves)")
    dtype, width = local_rows.dtype, local_rows.shape[1] * unit
    local = np.ascontiguousarray(local_rows, dtype=np.float64).reshape(local_rows.shape[0] // unit, width)
    pad = pack_block(local, total_rows // unit, world, rank)
    buf = torch.from_numpy(pad).to(device)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    rows = unpack_blocks(torch.stack(parts).cpu().numpy(), width, total_rows // unit, world)
    return rows.reshape(total_rows, width // unit).astype(dtype, copy=False)


def pack_block(local_rows: np.ndarray, total_rows: int, world: int, rank: int) -> np.ndarray:
    """Rank `rank`'s zero-padded block of the gather ([ceil(total_rows / world), width] f64) through sixdof_gather_pack."""
    import ctypes as C
    from . import _lib as L
    lib, dp = L.lib(), C.POINTER(C.c_double)
    local = np.ascontiguousarray(local_rows, dtype=np.float64)
    width = local.shape[1]
    block = np.empty((int(lib.sixdof_gather_block_rows(int(total_rows), int(world))), width), dtype=np.float64)
    rc = lib.sixdof_gather_pack(local.ctypes.data_as(dp), local.shape[0], width, int(total_rows), int(world), int(rank), block.ctypes.data_as(dp))
    if rc == L.ERR_VALUE_SIZE_MISMATCH:
        raise ValueError(f"gather: {local.shape[0]} rows are not rank {rank}'s block of {total_rows} rows over {world} ranks (shard_range)")
    if rc != L.OK:
        raise ValueError(f"sixdof_gather_pack: status {rc}")
    return block


def unpack_blocks(blocks: np.ndarray, width: int, total_rows: int, world: int) -> np.ndarray:
    """The `world` padded blocks in rank order -> [total_rows, width] in run-id order through sixdof_gather_unpack."""
    import ctypes as C
    from . import _lib as L
    dp = C.POINTER(C.c_double)
    blocks = np.ascontiguousarray(blocks, dtype=np.float64)
    out = np.empty((int(total_rows), int(width)), dtype=np.float64)
    rc = L.lib().sixdof_gather_unpack(blocks.ctypes.data_as(dp), int(width), int(total_rows), int(world), out.ctypes.data_as(dp))
    if rc != L.OK:
        raise ValueError(f"sixdof_gather_unpack: status {rc}")
    return out


class CapiComm:
    """The same two campaign collectives through the C ABI (sixdof_comm_* / sixdof_campaign_* over RCCL, no torch): what a
    non-Python host of the path calls.  `unique_id()` on rank 0, ship the 128 bytes to the other ranks, then every rank
    constructs `CapiComm(id, world, rank, device)`."""

    def __init__(self, comm_id: bytes | None, world: int, rank: int, device: int = 0):
        import ctypes as C
        from . import _lib as L
        self._lib, self._C = L.lib(), C
        self.world, self.rank = int(world), int(rank)
        self._h = C.c_void_p()
        if comm_id is not None and len(comm_id) != 128:
            raise ValueError("comm_id is the 128 bytes unique_id() returned on rank 0")
        idbuf = (C.c_uint8 * 128)(*comm_id) if comm_id is not None else None   # None: one rank, the library makes its own id
        rc = self._lib.sixdof_comm_init(C.byref(self._h), idbuf, self.world, self.rank, int(device))
        if rc != L.OK:
            msg = self._lib.sixdof_comm_last_error(None)
            raise L.BackendError(f"sixdof_comm_init: {msg.decode() if msg else ''} (status {rc})")

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C
        from . import _lib as L
        buf = (C.c_uint8 * 128)()
        rc = L.lib().sixdof_comm_unique_id(buf)
        if rc != L.OK:
            msg = L.lib().sixdof_comm_last_error(None)
            raise L.BackendError(f"sixdof_comm_unique_id: {msg.decode() if msg else ''} (status {rc})")
        return bytes(buf)

    def _check(self, rc, what):
        from . import _lib as L
        if rc != L.OK:
            msg = self._lib.sixdof_comm_last_error(self._h)
            raise L.BackendError(f"{what}: {msg.decode() if msg else ''} (status {rc})")

    def broadcast_table(self, table: np.ndarray | None, shape, dtype=np.float64, src: int = 0) -> np.ndarray:
        buf = (np.ascontiguousarray(table, dtype=dtype).reshape(shape).copy() if self.rank == src
               else np.empty(tuple(shape), dtype=dtype))
        self._check(self._lib.sixdof_campaign_broadcast(self._h, buf.ctypes.data, buf.nbytes, int(src)), "sixdof_campaign_broadcast")
        return buf

    def gather_rows(self, local_rows: np.ndarray, total_rows: int) -> np.ndarray:
        C = self._C
        local = np.ascontiguousarray(local_rows, dtype=np.float64)
        width = local.shape[1]
        out = np.empty((int(total_rows), width), dtype=np.float64)
        dp = C.POINTER(C.c_double)
        self._check(self._lib.sixdof_campaign_gather(self._h, local.ctypes.data_as(dp), local.shape[0], width, out.ctypes.data_as(dp),
                                                     int(total_rows)), "sixdof_campaign_gather")
        return out

    def close(self):
        if self._h:
            self._lib.sixdof_comm_destroy(self._h)
            self._h = None

    __del__ = close
