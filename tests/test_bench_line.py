"""The line bench.py hands the driver: ONE compact, strictly parseable JSON object as the LAST stdout line.

Round 5's line had grown to 20.7 KB and came back from the driver as `parsed: null` — the round's one driver-timed number
was lost.  These CPU tests run bench.py's own argument handling, process group, max-over-ranks and line composition with
fake timings (`--dry-run`; the line says so in `data`) for `--gpus 1` and for a 2-rank gloo launch exactly as the driver
launches N > 1, and gate the schema and the size.  What the reference prints for the same purpose is a handful of scalars
(libs/nox-py/src/profile.rs:14-59).
"""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import bench  # noqa: E402

CONTRACT = {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int, "warmup": int, "ms_per_step": float,
            "higher_is_better": bool, "scaling": str, "dtype": str, "data": str, "config": dict, "roofline": dict, "parity": dict}
ROOFLINE = {"bound": str, "achieved": float, "peak": float, "unit": str, "frac": float, "kernel": str, "avg_launch_us": float}
LIMIT = 4096


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env():
    env = dict(os.environ)
    env.pop("SIXDOF_BENCH_SHARED_GPU", None)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _check(stdout, n_gpus, steps, warmup):
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    assert lines, "bench.py printed nothing"
    last = lines[-1]
    assert len(last) < LIMIT, f"the line is {len(last)} bytes"
    # the libraries may chat on stdout ("[Gloo] Rank 0 is connected ..."); bench.py itself prints exactly one line, the last
    ours = [ln for ln in lines if ln.lstrip().startswith("{")]
    assert ours == [last], f"bench.py printed {len(ours)} JSON lines, and the last stdout line must be the one"

    def no_constants(x):        # strict JSON: NaN / Infinity are Python extensions a strict parser refuses
        raise ValueError(f"non-standard JSON constant {x}")
    doc = json.loads(last, parse_constant=no_constants)
    for k, t in CONTRACT.items():
        assert k in doc, f"missing {k}"
        assert isinstance(doc[k], (int, float) if t is float else t), (k, doc[k])
    assert "vs_baseline" in doc and doc["vs_baseline"] is None          # BASELINE.md holds no published number for this metric
    assert doc["n_gpus"] == n_gpus and doc["steps"] == steps and doc["warmup"] == warmup
    assert doc["metric"].startswith("entity-steps/s") and doc["unit"] == "entity-steps/s" and doc["dtype"] == "f64"
    assert doc["scaling"] == "weak" and doc["higher_is_better"] is True
    assert "workload" in doc["config"] and "model" not in doc["config"]
    for k, t in ROOFLINE.items():
        assert isinstance(doc["roofline"].get(k), (int, float) if t is float else t), ("roofline", k)
    assert "traffic" in doc["roofline"] and doc["roofline"]["bound"] == "hbm" and doc["roofline"]["peak"] == 8000.0
    assert abs(doc["roofline"]["frac"] - doc["roofline"]["achieved"] / doc["roofline"]["peak"]) < 1e-3
    # value = units all ranks processed / the (max over ranks) time; ms_per_step is that time per step
    assert doc["value"] == pytest.approx(doc["config"]["entities_per_gpu"] * n_gpus / (doc["ms_per_step"] * 1e-3), rel=1e-3)
    assert set(doc["parity"]) >= {"max_rel_err", "max_rel_err_elementwise", "tolerance", "entity_rows_bit_exact"}
    return doc


def test_single_gpu_line_is_compact_and_strict():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--dry-run"],
                       capture_output=True, text=True, timeout=600, env=_env(), cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    doc = _check(r.stdout, 1, 20, 5)
    assert "DRY RUN" in doc["data"]
    cb = doc["cpu_baseline"]                      # N = 1 carries the CPU oracle's figure (here on a tiny sample)
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["unit"] == "entity-steps/s" and cb["value"] > 0 and cb["sample"]
    assert "rccl" not in doc and "campaigns" not in doc


def test_two_rank_gloo_line_is_compact_and_strict():
    """Launched the way the driver launches N > 1 (one process per rank, torch.distributed.run, 127.0.0.1)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--dry-run"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=_env(), cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    doc = _check(r.stdout, 2, 20, 5)
    # rank 1's fake time is 1 % longer than rank 0's: the line must carry the MAX over ranks
    assert doc["ms_per_step"] == pytest.approx(6.0e-3 * 1.01, rel=1e-6)
    assert doc["rccl"]["world_size"] == 2 and doc["rccl"]["backend"] == "gloo"
    assert doc["n1_reference_value"] == pytest.approx(doc["value"] / 2, rel=1e-6)
    assert "cpu_baseline" not in doc               # rank 0 at N = 1 only
    camp = doc["campaigns"]                        # BASELINE configs[3] / [4] over the same ranks: numbers only
    assert set(camp["apollo"]) == set(camp["falcon9"]) == {"strong", "weak", "strong_seconds", "weak_seconds"}


def test_fit_line_drops_detail_before_it_outgrows_the_driver():
    import argparse
    args = argparse.Namespace(steps=20, warmup=5)
    tm = {"launches": 20, "graph_launches": 20, "kernel_device_ms": 0.12}
    fat = {f: 1.2345678e-15 for f in ("world_pos", "world_vel", "world_accel", "force")}
    out = bench.compose_line(args, 8, 65536, 1, 1.4e-4, 3.0e-4, tm,
                             roofline=bench.roofline_from(6e-3, 65536, 20, "x" * 3000),
                             parity={"max_rel_err": 2e-15, "by_column": fat, "max_rel_err_elementwise": 1e-12, "by_column_elementwise": fat,
                                     "tolerance": 1e-9, "entity_rows_bit_exact": True},
                             rccl={"backend": "nccl", "world_size": 8, "devices": ["0000:%02x:00" % i for i in range(8)]},
                             campaigns={"apollo": {"strong": 1.0}})
    text = bench.fit_line(out)
    assert len(text) < LIMIT
    doc = json.loads(text)
    assert "timing" not in doc["roofline"]                     # the 3 KB string went, the contract keys stayed
    for k in bench.REQUIRED_KEYS:
        assert k in doc


def test_real_sized_lines_fit_with_margin():
    """The largest line bench.py can compose (8 ranks, campaigns, every optional key) stays well under the limit untrimmed."""
    import argparse
    args = argparse.Namespace(steps=20, warmup=5)
    tm = {"launches": 20, "graph_launches": 20, "kernel_device_ms": 0.12}
    fat = {f: 1.234e-15 for f in ("world_pos", "world_vel", "world_accel", "force")}
    camp = {"unit": "rollout-steps/s", "totals": bench.CAMPAIGN_TOTALS}
    for which in bench.CAMPAIGN_TOTALS:
        camp[which] = {"strong": 9283645618.3, "weak": 69283645618.3, "strong_seconds": 0.0521, "weak_seconds": 0.0621}
    roof = bench.roofline_from(6e-3, 65536, 20, "HIP events around the timed region on the launch stream / launches")
    roof["traffic"], roof["traffic_source"] = 25401234.0, "profiles/pmc_traffic.json (rocprofv3 --pmc passes; valid for step-kernel sources 0123456789abcdef)"
    roof["long_batch"] = {"launches": 4096, "avg_launch_us": 5.59, "frac": 0.5627}
    out = bench.compose_line(args, 8, 65536, 1, 1.4e-4, 3.0e-4, tm, roofline=roof,
                             parity={"max_rel_err": 2.062e-15, "by_column": fat, "max_rel_err_elementwise": 1.098e-12, "by_column_elementwise": fat,
                                     "tolerance": 1e-9, "entity_rows_bit_exact": True, "rows": 4096, "ticks": 16,
                                     "vs": "oracle/sixdof_oracle.c (bit-exact on the reference's golden CSVs)"},
                             cpu={"value": 3934129.8, "unit": "entity-steps/s", "cores": 1, "kind": "port", "sample": "s" * 180,
                                  "multi_thread": {"value": 64469184.5, "cores": 16, "ticks": 3535, "threads_rule": "r" * 70}},
                             rccl={"backend": "nccl", "world_size": 8, "rccl_version": "2.26.6", "distinct_devices": True,
                                   "devices": ["0000:%02x:00" % i for i in range(8)]},
                             campaigns=camp)
    raw = json.dumps(out, separators=(", ", ": "))
    assert len(raw) < 3800, len(raw)
    assert json.loads(bench.fit_line(out)) == json.loads(raw)            # nothing had to be dropped


@pytest.mark.parametrize("fault", ["hang:1", "raise:1", "raise:0"])
def test_a_rank_failing_alone_in_the_campaign_leg_does_not_cost_the_headline(fault):
    """The campaign numbers are the one part of the N > 1 line that runs collectives inside library code.  A rank that hangs or
    raises there alone must not take the headline — what the driver's scaling curve is made of — with it: the leg runs last,
    under a watchdog, and the line then carries `campaigns: {"error": ...}`."""
    env = _env()
    env["SIXDOF_BENCH_DRYRUN_FAULT"] = fault
    env["SIXDOF_BENCH_CAMPAIGN_TIMEOUT"] = "8"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--dry-run"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=str(ROOT))
    doc = _check(r.stdout, 2, 20, 5)
    assert doc["value"] > 0 and doc["rccl"]["world_size"] == 2
    assert "error" in doc["campaigns"] or "closing_barrier_error" in doc["campaigns"], doc["campaigns"]
    assert "campaign" in r.stderr
