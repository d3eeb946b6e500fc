"""N>1 path on CPU: world_size-2 gloo process group exercising the same shard / broadcast / gather /
max-over-ranks code bench.py and a Monte-Carlo campaign use on RCCL.  The stepping itself is done by the CPU
oracle here (this is a test of the partitioning, not of the kernels)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from elodin_amd import shard, workloads
from oracle import oracle as orc

TOTAL, TICKS = 1000, 5


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _ops(w):
    return [(orc.EFF_UNIFORM_GRAVITY, (0.0, 0.0, -9.81), None), (orc.EFF_BODY_TORQUE, (), w["body_torque"])]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard.shard_range(TOTAL, world, rank)
        # campaign parameter table lives on rank 0 only
        table = np.arange(TOTAL * 3, dtype=np.float64).reshape(TOTAL, 3) if rank == 0 else None
        table = shard.broadcast_table(table, (TOTAL, 3))
        w = workloads.independent_bodies(hi - lo, first_row=lo)
        o = orc.OracleWorld(w["world_pos"], w["world_vel"], w["inertia"], simulation_time_step=workloads.DT_120HZ,
                            ops=_ops(w)).step(TICKS)
        result = np.concatenate([o.world_pos, table[lo:hi], w["entity_ids"][:, None].astype(np.float64)], axis=1)
        gathered = shard.gather_rows(result, TOTAL)
        t = shard.max_over_ranks(1.0 + rank)
        if rank == 0:
            q.put((gathered, t))
    finally:
        dist.destroy_process_group()


def test_shard_range_partitions_exactly():
    for total in (0, 1, 7, 65536, 8192 * 3 + 5):
        for world in (1, 2, 3, 8):
            r = [shard.shard_range(total, world, k) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(hi - lo for lo, hi in r) - min(hi - lo for lo, hi in r) <= 1
    assert shard.run_id(41) == "run_0000041"
    with pytest.raises(ValueError):
        shard.shard_range(10, 2, 2)


def test_two_rank_gloo_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    gathered, t = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert t == 2.0  # MAX over ranks
    w = workloads.independent_bodies(TOTAL)
    o = orc.OracleWorld(w["world_pos"], w["world_vel"], w["inertia"], simulation_time_step=workloads.DT_120HZ,
                        ops=_ops(w)).step(TICKS)
    assert np.array_equal(gathered[:, :7], o.world_pos)            # sharding does not change a single bit
    assert np.array_equal(gathered[:, 7:10], np.arange(TOTAL * 3, dtype=np.float64).reshape(TOTAL, 3))
    assert np.array_equal(gathered[:, 10].astype(np.uint64), w["entity_ids"])  # entity indices bit-exact


# ---- Apollo campaign over 2 gloo ranks (executor = CPU oracle; the GPU executor is covered by -m gpu tests) ----

class _OracleExec:
    def __init__(self, block, first_row):
        from elodin_amd.models import apollo
        from oracle.apollo import ApolloOracle
        ref = apollo.load_reference()
        self._o = ApolloOracle(apollo.initial_columns(block, ref), ref, max_ticks=apollo.max_ticks(ref))

    def run(self, n):
        self._o.step(n)

    @property
    def result(self):
        return self._o.result


def _campaign_worker(rank, world, port, q):
    from pathlib import Path
    from elodin_amd import monte_carlo as mc
    from elodin_amd.models import apollo
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        table = None
        if rank == 0:
            table = mc.materialize(mc.load_spec(Path(__file__).parent / "golden" / "plans" / "apollo.toml")).table()
        res = apollo.run_campaign(table, 30, 59041, make_exec=_OracleExec)
        if rank == 0:
            q.put(res)
    finally:
        dist.destroy_process_group()


def test_apollo_campaign_two_ranks_equals_one():
    from pathlib import Path
    from elodin_amd import monte_carlo as mc
    from elodin_amd.models import apollo
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_campaign_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res2 = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    table = mc.materialize(mc.load_spec(Path(__file__).parent / "golden" / "plans" / "apollo.toml")).table()
    res1 = apollo.run_campaign(table, 30, 59041, make_exec=_OracleExec)
    assert res2.shape == (30, 12) and np.array_equal(res1, res2)     # run-id order, bit-identical
    assert np.all(res2[:, 8] == 1.0)


# ---- Falcon 9 ascent campaign (config 5): same sharding scheme, program stepped on the host by the numpy evaluator ----------

class _NumpyAscentExec:
    """Stand-in for models.falcon9.AscentExec on a CPU-only box: the SAME traced program, stepped by
    tests/dsl_numpy.program_tick.  `result` = a few state columns (the metrics latch needs a whole flight)."""

    def __init__(self, block, first_row):
        from elodin_amd.models import falcon9 as f9
        self.f9 = f9
        cols = f9.initial_columns(block)
        self.tp = f9.build_program().trace({k: v.shape[1] for k, v in cols.items()})
        self.pos, self.vel, self.inertia = cols.pop("world_pos"), cols.pop("world_vel"), cols.pop("inertia")
        self.acc = np.zeros_like(self.vel)
        self.comps = {name: cols[name] for name, _ in self.tp.columns}
        self.tick = 0

    def run(self, n):
        from elodin_amd import _lib as L
        from tests import dsl_numpy
        for _ in range(n):
            self.tick += 1
            dsl_numpy.program_tick(self.tp, self.pos, self.vel, self.acc, self.inertia, self.comps, self.tick,
                                   self.f9.SIM_TIME_STEP, L.SEMI_IMPLICIT)

    @property
    def result(self):
        c = self.comps
        return np.concatenate([c["thrust_total"], c["propellant_lox"], c["engine_spool"][:, :1], c["valve_state"][:, 4:5],
                               c["fsw_state"][:, :1], self.inertia[:, 6:7], c["tank_pressure_lox"], c["params"][:, :1]], axis=1)


def _falcon9_worker(rank, world, port, q):
    from elodin_amd.models import falcon9 as f9
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = f9.run_campaign(f9.sample_params(5) if rank == 0 else None, 5, 450, make_exec=_NumpyAscentExec)
        if rank == 0:
            q.put(res)
    finally:
        dist.destroy_process_group()


def test_falcon9_campaign_two_ranks_equals_one():
    from elodin_amd.models import falcon9 as f9
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_falcon9_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res2 = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    params = f9.sample_params(5)
    res1 = f9.run_campaign(params, 5, 450, make_exec=_NumpyAscentExec)
    assert res2.shape == (5, 8) and np.array_equal(res1, res2)         # run-id order, bit-identical (3 + 2 rows)
    assert np.array_equal(res2[:, 7], params[:, 0])                    # every rank flew ITS rows of rank 0's table
    assert np.all(res2[:, 0] > 1.0e5) and np.all(res2[:, 4] == 1.0)     # 0.25 s after ignition: engines spooling up, VerticalRise
