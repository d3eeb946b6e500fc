"""Campaign artefact writers (elodin_amd/campaign.py) against the behaviour libs/monte-carlo/src/lib.rs pins in its own
unit tests (read_post_run_outcome_extracts_hook_fields :4425-4456, passed_respects_post_run_hook_verdict :4366-4388,
invalid_runs_count_toward_fail_on_errors :4390-4423) and the file layouts of write_results_csv / CampaignSummary."""
import csv
import json
from pathlib import Path

import numpy as np
import pytest

from elodin_amd import campaign as cp
from elodin_amd import monte_carlo as mc
from elodin_amd.models import apollo


def test_post_run_outcome_fields():
    o = cp.read_post_run_outcome({"pass": False, "traj_rmse_m": 12.5})
    assert o.passed is False and o.valid is None and o.scalars["traj_rmse_m"] == "12.5"
    o = cp.read_post_run_outcome({"pass": True, "valid": False})
    assert o.passed is True and o.valid is False
    o = cp.read_post_run_outcome({"traj_rmse_m": 1.0})
    assert o.passed is None and o.scalars["traj_rmse_m"] == "1.0"
    o = cp.read_post_run_outcome(None)
    assert o.passed is None and o.scalars == {}
    o = cp.read_post_run_outcome({"status": "invalid", "n": 3, "tiny": 1e-7, "name": "x", "arr": [1], "obj": {}, "nul": None, "flag": True})
    assert o.valid is False and o.scalars == {"status": "invalid", "n": "3", "tiny": "1e-7", "name": "x", "flag": "true"}


def test_passed_respects_hook_verdict_and_invalid_runs_are_counted_apart(tmp_path):
    m = cp.RunMetric("run_0000000", exit_ok=True)
    assert m.passed()
    m.scored_pass = False
    assert not m.passed()
    m.scored_pass = True
    assert m.passed()
    m.exit_ok = False
    assert not m.passed()
    ok = cp.RunMetric("run_0000000", exit_ok=True, scored_pass=True, scored_valid=True)
    invalid = cp.RunMetric("run_0000001", exit_ok=True, scored_pass=True, scored_valid=False)
    import datetime as dt
    now = dt.datetime.now(dt.timezone.utc)
    s = cp.summarize_campaign(tmp_path, [ok, invalid], now, now, 0, 1)
    assert s["failed"] == 0 and s["invalid"] == 1 and s["passed"] == 1 and s["parallel_efficiency"] == 0.0


def test_hook_metric_summary_uses_the_reference_percentile():
    ms = [cp.RunMetric(f"r{i}", hook_scalars={"x": cp._number_text(float(i)), "pass": "true", "label": "abc"}) for i in range(1, 41)]
    ms.append(cp.RunMetric("bad", scored_valid=False, hook_scalars={"x": "1000.0"}))       # invalid runs do not contribute
    s = cp.summarize_hook_metrics(ms)
    assert set(s) == {"x"} and s["x"] == {"count": 40, "min": 1.0, "mean": 20.5, "p95": 38.0, "max": 40.0}   # ceil(40*.95)-1 = 37


def _score(ctx):
    """Shape of examples/apollo-lander/hooks/score.py:post_run, own code: reads result.json, returns scalars + verdict."""
    r = json.loads((Path(ctx.run_dir) / "result.json").read_text())
    name = next(iter(ctx.params))                  # a run's parameters: a mapping AND attributes, like run_hook.py's namespace
    assert getattr(ctx.params, name) == ctx.params[name] and dict(ctx.params.items())[name] == ctx.params[name]
    return {"landed": bool(r["landed"]), "soft_landing": bool(r["soft_landing"]), "valid": bool(r), "pass": bool(r["soft_landing"]),
            "touchdown_speed_mps": r["touchdown_speed"], "fuel_remaining_kg": r["fuel_remaining"],
            "downrange_miss_m": float("inf") if not r["landed"] else r["downrange_miss"], "seed_seen": ctx.seed}


def test_campaign_directory_layout(tmp_path):
    spec = mc.load_spec(Path(__file__).parent / "golden" / "plans" / "apollo.toml")
    plan = mc.materialize(spec)
    n = len(plan)
    rng = np.random.default_rng(1)
    res = np.zeros((n, len(apollo.RESULT_NAMES)))
    res[:, apollo.RESULT_NAMES.index("touchdown_speed")] = rng.uniform(0.2, 3.0, n)
    res[:, apollo.RESULT_NAMES.index("fuel_remaining")] = rng.uniform(50, 400, n)
    res[:, apollo.RESULT_NAMES.index("landed")] = 1.0
    res[-1, apollo.RESULT_NAMES.index("landed")] = 0.0
    res[:, apollo.RESULT_NAMES.index("soft_landing")] = (res[:, 0] < 2.0).astype(float)
    res[:, apollo.RESULT_NAMES.index("tick")] = 59000
    failed = np.zeros(n, dtype=bool)
    failed[3] = True
    out = cp.write_campaign(tmp_path / "camp", plan, res, apollo.RESULT_NAMES, wall_ms=1234.0, workers=2, post_run=_score,
                            result_record=apollo.result_record, failed_rows=failed, rows_per_worker=(n + 1) // 2)
    root = tmp_path / "camp"
    assert (root / "plan.csv").read_bytes().decode() == plan.to_csv()
    rows = list(csv.reader(open(root / "results.csv")))
    assert rows[0][:15] == cp.RESULTS_HEADER
    assert rows[0][15:] == sorted(["landed", "soft_landing", "touchdown_speed_mps", "fuel_remaining_kg", "downrange_miss_m", "seed_seen"])
    assert len(rows) == n + 1 and [r[0] for r in rows[1:]] == plan.run_ids
    by_id = {r[0]: dict(zip(rows[0], r)) for r in rows[1:]}
    r0 = by_id["run_0000000"]
    assert r0["status"] == "ok" and r0["result_json"] == "runs/run_0000000/result.json" and r0["seed_seen"] == "1"
    assert r0["passed"] == r0["scored_pass"] == ("true" if res[0, 0] < 2.0 else "false") and r0["worker_id"] == "0"
    assert by_id[plan.run_ids[-1]]["worker_id"] == "1"
    assert by_id[plan.run_ids[-1]]["downrange_miss_m"] == ""                  # inf -> null in JSON -> no cell
    bad = by_id["run_0000003"]
    assert bad["status"] == "failed" and bad["passed"] == "false" and bad["failure_reason"] == "non-finite state" and bad["scored_pass"] == ""
    rec = json.loads((root / "runs" / "run_0000000" / "result.json").read_text())
    assert rec["landed"] is True and isinstance(rec["tick"], int) and "reserved" not in rec
    s = json.loads((root / "summary.json").read_text())
    need = {"started_at", "finished_at", "total_runs", "passed", "failed", "invalid", "degraded", "skipped", "workers", "wall_ms",
            "total_run_wall_ms", "average_run_wall_ms", "max_run_wall_ms", "parallel_efficiency", "disk_bytes", "resource_summary",
            "sim_phase_summary", "phase_attribution", "concurrency_summary", "hook_metrics", "pacing"}
    assert set(s) == need and s["total_runs"] == n and s["workers"] == 2 and s["started_at"].endswith("Z")
    n_soft = int(np.sum((res[:, 0] < 2.0) & ~failed))
    assert s["passed"] == n_soft and s["failed"] == n - n_soft and s["invalid"] == 0
    assert s["hook_metrics"]["touchdown_speed_mps"]["count"] == n - 1 and "pass" not in s["hook_metrics"]
    assert set(s["phase_attribution"]) == {"samples", "average_python_import_ms", "average_compile_ms", "average_loop_ms",
                                           "average_teardown_ms", "average_process_shutdown_ms", "p95_python_import_ms",
                                           "p95_compile_ms", "p95_loop_ms", "p95_teardown_ms", "p95_process_shutdown_ms"}


def test_sim_side_context_api(tmp_path, monkeypatch):
    """el.monte_carlo.{Param, params_spec, params, result, port, spec_json} (libs/nox-py/src/monte_carlo.rs): declared
    defaults overlaid with the run's context document, which `Plan.context(i)` writes exactly as the runner's read_plan
    splits a plan row (param.* / meta.* columns, JSON-parsed cells)."""
    import json
    import elodin_amd.frontend as el
    from elodin_amd import monte_carlo as mc
    spec = el.monte_carlo.params_spec(mass_kg=el.monte_carlo.Param(float, 15103.0, min=14000.0, max=16000.0),
                                      engine=el.monte_carlo.Param(str, "descent"), n_jets=el.monte_carlo.Param(int, 16))
    assert json.loads(spec.to_json())["params"]["mass_kg"] == {"type_name": "float", "default": 15103.0, "min": 14000.0, "max": 16000.0}
    assert json.loads(el.monte_carlo.spec_json()) == json.loads(spec.to_json())
    with pytest.raises(TypeError, match="must be el.monte_carlo.Param"):
        el.monte_carlo.params_spec(x=3.0)
    with pytest.raises(ValueError, match="finite"):
        el.monte_carlo.Param(float, float("inf"))
    # no context: the declared defaults
    monkeypatch.delenv(mc.CONTEXT_ENV, raising=False)
    p = el.monte_carlo.params(spec)
    assert (p["mass_kg"], p.get("engine"), p.get("missing", 7), p.run_id, p.seed, p.run_dir) == (15103.0, "descent", 7, None, None, None)
    with pytest.raises(KeyError):
        p["missing"]
    with pytest.raises(RuntimeError, match="requires ELODIN_MONTE_CARLO_CONTEXT"):
        el.monte_carlo.result(ok=True)
    # a plan row -> context document -> params()
    plan = mc.materialize({"sim_sweep": {"engine": ["descent", "ascent"]}, "meta_sweep": {"wind": [0, 5]},
                           "monte_carlo": {"n_samples": 3, "seed": 4, "variables": {"mass_kg": {"dist": "uniform", "min": 14000.0, "max": 16000.0}}}})
    ctx = plan.context(7, run_dir=tmp_path, slots={"ports": {"db": 2240}, "ctrl_port": 9001, "bad_port": 70000})
    assert ctx["run_id"] == "run_0000007" and ctx["seed"] == 8 and set(ctx["params"]) == {"engine", "mass_kg"} and set(ctx["meta"]) == {"wind"}
    path = tmp_path / "context.json"
    path.write_text(json.dumps(ctx))
    monkeypatch.setenv(mc.CONTEXT_ENV, str(path))
    p = el.monte_carlo.params()                      # the spec declared last
    assert p.run_id == "run_0000007" and p.seed == 8 and p.run_dir == str(tmp_path) and p["n_jets"] == 16
    assert p["mass_kg"] == plan.rows[7]["param.mass_kg"] and p["engine"] == plan.rows[7]["param.engine"] and p.meta == {"wind": plan.rows[7]["meta.wind"]}
    assert p.as_overrides_dict()["mass_kg"] == p["mass_kg"] and p.ports() == {"db": 2240, "ctrl": 9001}
    assert el.monte_carlo.port("db") == 2240 and el.monte_carlo.port("nope", 5) == 5
    monkeypatch.setenv("ELODIN_MC_PORT_MY_SVC", "4000")
    assert el.monte_carlo.port("my-svc") == 4000
    with pytest.raises(KeyError):
        el.monte_carlo.port("nope")
    el.monte_carlo.result(landed=True, touchdown_speed=1.25, notes=["a", 2])
    assert json.loads((tmp_path / "result.json").read_text()) == {"landed": True, "touchdown_speed": 1.25, "notes": ["a", 2]}
    monkeypatch.setenv(mc.CONTEXT_ENV, str(tmp_path / "gone.json"))
    with pytest.raises(RuntimeError, match="failed to read"):
        el.monte_carlo.params()
