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


# ---- a Monte-Carlo of WHOLE-WORLD ticks (three-body worlds, one lane per entity: a world = 4 consecutive rows that exchange data
#      inside the wavefront) shards by world: shard_range(..., unit=rows_per_world).  Stepped by the numpy walker of the traced program.
WORLDS = 9


def _world_columns(lo_world, hi_world, S):
    from tests import stablehlo_world_util as W
    g = W.gu.load("three_body")
    cols = W.strided_world_columns(g, "abc", S, WORLDS)
    rng = np.random.default_rng(5)
    body_rows = np.array([w_ * S + i for w_ in range(WORLDS) for i in range(3)])
    cols["hlo_world_pos"][body_rows, 4:6] += rng.uniform(-0.05, 0.05, (len(body_rows), 2))       # every world its own state
    return {k: v[lo_world * S:hi_world * S].copy() for k, v in cols.items()}


def _world_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from elodin_amd import stablehlo as sh
        from tests.golden import hlo_world_builder as hb
        from tests.test_stablehlo_world import walk
        text, slots = hb.three_body_world()
        system, manifest = sh.world_system(text, slots, mode="auto")
        S = manifest["rows_per_world"]
        lo, hi = shard.shard_range(WORLDS * S, world, rank, unit=S)
        assert lo % S == 0 and hi % S == 0
        cols = _world_columns(lo // S, hi // S, S)
        walk(system, {c["column"]: c["width"] for c in manifest["columns"]}, cols, 10)
        gathered = shard.gather_rows(np.concatenate([cols["hlo_world_pos"], cols["hlo_world_vel"]], axis=1), WORLDS * S, unit=S)
        if rank == 0:
            q.put((gathered, (lo, hi)))
    finally:
        dist.destroy_process_group()


def test_shards_of_whole_world_ticks_are_whole_worlds():
    assert [shard.shard_range(36, 2, r, unit=4) for r in range(2)] == [(0, 20), (20, 36)]            # 9 worlds: 5 + 4
    assert [shard.shard_range(64 * 3, 8, r, unit=64) for r in range(8)] == [(0, 64), (64, 128), (128, 192)] + [(192, 192)] * 5
    with pytest.raises(ValueError, match="whole number"):
        shard.shard_range(38, 2, 0, unit=4)


def test_two_rank_gloo_monte_carlo_of_three_body_worlds_in_lane_mode_matches_single_process():
    from elodin_amd import stablehlo as sh
    from tests.golden import hlo_world_builder as hb
    from tests.test_stablehlo_world import walk
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_world_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    gathered, rank0 = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    text, slots = hb.three_body_world()
    system, manifest = sh.world_system(text, slots, mode="auto")
    S = manifest["rows_per_world"]
    assert rank0 == (0, 5 * S)
    cols = _world_columns(0, WORLDS, S)
    walk(system, {c["column"]: c["width"] for c in manifest["columns"]}, cols, 10)
    want = np.concatenate([cols["hlo_world_pos"], cols["hlo_world_vel"]], axis=1)
    body = np.array([w_ * S + i for w_ in range(WORLDS) for i in range(3)])
    assert np.isfinite(want[body]).all() and np.array_equal(gathered[body], want[body])      # not a bit differs (padding rows hold nothing)
    assert np.array_equal(gathered, want, equal_nan=True)
