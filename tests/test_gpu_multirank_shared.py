"""What of the N > 1 path can run on a ONE-GPU box, with the REAL executors: two ranks (a gloo group — RCCL refuses two ranks
on one device) sharing cuda:0 fly the Apollo and Falcon 9 campaigns through models.*.run_campaign with the HIP kernels — the
CPU-only twin of this (tests/test_shard_gloo.py) flies the oracle — with rollout counts that do NOT divide by the world size,
and the driver's own multi-rank command line prints the whole line (config 2 + both campaigns, strong and weak scaling)."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]
APOLLO_RUNS, APOLLO_TICKS = 67, 4000            # 34 + 33 rollouts
F9_RUNS, F9_TICKS = 133, 3000                  # 67 + 66


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _tables():
    from elodin_amd import monte_carlo as mc
    from elodin_amd.models import falcon9
    spec = mc.load_spec(ROOT / "tests" / "golden" / "plans" / "apollo.toml")
    spec["monte_carlo"]["n_samples"] = APOLLO_RUNS
    return mc.materialize(spec).table(), falcon9.sample_params(F9_RUNS)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from elodin_amd.models import apollo, falcon9
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a_table, f_table = _tables() if rank == 0 else (None, None)
        a = apollo.run_campaign(a_table, APOLLO_RUNS, APOLLO_TICKS, device=0)                     # HIP executor, cuda:0 on both ranks
        f = falcon9.run_campaign(f_table, F9_RUNS, F9_TICKS, device=0, dtype=np.float64, fast_math=False,
                                 make_exec=lambda block, first_row: _StateExec(block))
        if rank == 0:
            q.put((a, f))
    finally:
        dist.destroy_process_group()


class _StateExec:
    """models.falcon9.AscentExec with a few state columns as its result (the metrics latch needs a whole flight)."""

    def __init__(self, block):
        from elodin_amd.models import falcon9
        self.ex = falcon9.AscentExec(block, dtype=np.float64, ticks_per_launch=500, device=0)

    def run(self, n): self.ex.run(n)
    def close(self): self.ex.close()

    @property
    def result(self):
        return np.concatenate([self.ex.hip.world_pos, self.ex.hip.world_vel, np.asarray(self.ex.column("thrust_total"), dtype=np.float64)], axis=1)


def test_two_gloo_ranks_fly_both_campaigns_with_the_hip_executors_on_one_gpu():
    import torch.multiprocessing as mp
    from elodin_amd.models import apollo, falcon9
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    import queue
    import time
    deadline = time.time() + 600
    while True:                      # a rank that dies (no GPU, a failed assertion) must fail the test now, not after the timeout
        try:
            a2, f2 = q.get(timeout=2)
            break
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            assert not dead and time.time() < deadline, f"rank processes exited with {dead}"
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    a_table, f_table = _tables()
    a1 = apollo.run_campaign(a_table, APOLLO_RUNS, APOLLO_TICKS, device=0)
    f1 = falcon9.run_campaign(f_table, F9_RUNS, F9_TICKS, device=0, dtype=np.float64, fast_math=False,
                              make_exec=lambda block, first_row: _StateExec(block))
    assert a2.shape == a1.shape == (APOLLO_RUNS, 12) and np.array_equal(a1, a2, equal_nan=True)      # run-id order, bit-identical
    assert f2.shape == f1.shape == (F9_RUNS, 14) and np.array_equal(f1, f2)
    assert np.abs(f1[:, 13]).max() > 1e6                                                                # engines lit: the flights really flew


def test_the_drivers_multi_rank_command_prints_config_2_and_both_campaigns_strong_and_weak(tmp_path):
    env = dict(os.environ, SIXDOF_BENCH_SHARED_GPU="1", PYTHONPATH=str(ROOT), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=tmp_path, env=env)
    assert r.returncode == 0, r.stderr[-1500:]
    from tests.test_bench_line import _check
    line = _check(r.stdout, 2, 20, 5)                        # compact (< 4 KB), strict JSON, the contract's keys: what the driver parses
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert line["scaling"] == "weak" and line["config"]["entities_per_gpu"] == 65536 and line["value"] > 0 and line["data"] == "synthetic"
    assert "share GPU 0" in line["config"]["parallelism"]
    # the line attests what the process group IS: backend, the group's own world size, the ranks' devices
    att = line["rccl"]
    assert att["backend"] == "gloo" and att["world_size"] == 2 and len(att["devices"]) == 2 and att["distinct_devices"] is False      # one GPU, said so
    assert abs(line["n1_reference_value"] * 2 - line["value"]) < 1.0
    assert line["parity"]["max_rel_err"] < 1e-9 and line["parity"]["max_rel_err_elementwise"] < 1e-9 and line["parity"]["entity_rows_bit_exact"] is True
    camp = line["campaigns"]                                 # BASELINE configs[3] / [4] over the same two ranks: rollout-steps/s, strong and weak
    for which, total in (("apollo", 8192), ("falcon9", 32768)):
        assert camp["totals"][which] == total
        for s_ in ("strong", "weak"):
            assert camp[which][s_] > 0 and camp[which][s_ + "_seconds"] > 0
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "bench_2rank_shared_gpu.json").write_text(lines[0] + "\n")


def test_the_drivers_multi_rank_command_over_real_rccl_when_the_box_has_two_gpus(tmp_path):
    """The same command line with ONE GPU PER RANK over RCCL (backend "nccl") — exactly what the driver's scaling run launches.  Needs two
    GPUs: on the one-GPU boxes this suite usually runs on it SKIPS (RCCL refuses two ranks on one device: tests/test_campaign_comm.py
    records that refusal); on a multi-GPU box it is the first real N > 1 evidence: distinct devices attested by the group itself, the
    weak-scaled value, both campaigns' broadcast / gather through RCCL."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU on this box: two RCCL ranks need two devices")
    env = dict(os.environ, PYTHONPATH=str(ROOT), HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("SIXDOF_BENCH_SHARED_GPU", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=tmp_path, env=env)
    assert r.returncode == 0, r.stderr[-1500:]
    from tests.test_bench_line import _check
    line = _check(r.stdout, 2, 20, 5)
    att = line["rccl"]
    assert att["backend"] == "nccl" and att["world_size"] == 2 and att["distinct_devices"] is True and len(set(att["devices"])) == 2
    assert "share GPU 0" not in line["config"]["parallelism"] and line["value"] > 0
    assert line["parity"]["max_rel_err"] < 1e-9
    assert "error" not in line["campaigns"], line["campaigns"]
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "bench_2rank_rccl.json").write_text(json.dumps(line) + "\n")
