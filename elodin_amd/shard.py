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


def shard_range(total_rows: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of rank `rank`; blocks differ by at most one row."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, extra = divmod(total_rows, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


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


def gather_rows(local_rows: np.ndarray, total_rows: int, device="cpu") -> np.ndarray:
    """All-gather per-rollout result rows back into run-id order (shards are contiguous blocks)."""
    import torch
    dist = _dist()
    local_rows = np.ascontiguousarray(local_rows)
    if not (dist.is_available() and dist.is_initialized()):
        return local_rows
    world = dist.get_world_size()
    width = local_rows.shape[1]
    sizes = [shard_range(total_rows, world, r) for r in range(world)]
    pad = max(hi - lo for lo, hi in sizes)
    buf = torch.zeros((pad, width), dtype=getattr(torch, local_rows.dtype.name), device=device)
    buf[: local_rows.shape[0]] = torch.from_numpy(local_rows)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return np.concatenate([o.cpu().numpy()[: hi - lo] for o, (lo, hi) in zip(out, sizes)], axis=0)
