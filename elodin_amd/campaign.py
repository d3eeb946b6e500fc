"""Campaign artefacts of a GPU Monte-Carlo run in the layout `elodin monte-carlo run` leaves behind, so the reference's
post-processing (`elodin monte-carlo report`, example `hooks/score.py` / `report.py`) can consume a GPU campaign:

    <out>/plan.csv                       the plan that was flown (monte_carlo.Plan.to_csv, byte-identical to sample.py's)
    <out>/runs/<run_id>/result.json      the sim's own result record of that rollout (what e.g. apollo-lander/main.py
                                         writes at touchdown; here: the rollout's row of in-kernel result scoring)
    <out>/runs/<run_id>/post_run_context.json  what the runner hands the post_run hook (lib.rs:2264-2277)
    <out>/runs/<run_id>/post_run_result.json   the scoring hook's outcome, when a `post_run` hook is given
    <out>/campaign_hook_context.json     what the runner hands the post_campaign hook (lib.rs:1335-1347)
    <out>/results.csv                    one row per run: fixed columns + sorted hook scalar columns (lib.rs:3109-3187)
    <out>/summary.json                   CampaignSummary (lib.rs:348-376, built like summarize_campaign lib.rs:1715-1787)

Restated from libs/monte-carlo/src/lib.rs: RunMetric.passed/valid (:338-345), read_post_run_outcome + scalar_value
(:2447-2479), summarize_hook_metrics (:1815-1852), write_results_csv (:3109-3187), RESERVED_HOOK_KEYS (:31).
The process-orchestration fields of the reference (resource sampling, per-phase process attribution, pacing) have no
GPU counterpart: all rollouts of a rank run inside one process in lock-step, so those summaries are written with zero
samples and `wall_ms` of a run is the campaign wall time amortised over the rollouts.
"""
from __future__ import annotations

import csv
import datetime as _dt
import json
import math
import os
from dataclasses import dataclass, field
from pathlib import Path
from types import SimpleNamespace
from typing import Any, Callable, Dict, List, Mapping, Optional, Sequence

import numpy as np

RESERVED_HOOK_KEYS = ("pass", "valid")
RESULTS_HEADER = ["run_id", "status", "passed", "valid", "degraded", "scored_pass", "scored_valid", "failure_reason",
                  "behind_deadline_frac", "real_time_factor", "drift_resets", "worker_id", "wall_ms", "db_path", "result_json"]


def _number_text(v) -> str:
    """serde_json's Number::to_string: integers plain, floats shortest round-trip (ryu: `1e-7`, not `1e-07`)."""
    if isinstance(v, (bool, np.bool_)):
        return "true" if v else "false"
    if isinstance(v, (int, np.integer)):
        return str(int(v))
    t = repr(float(v))
    if "e" in t:
        mant, exp = t.split("e")
        t = f"{mant}e{int(exp)}"
    return t


def scalar_value(value) -> Optional[str]:
    """lib.rs:2472-2479: bool / number / string become cells, null / arrays / objects are dropped."""
    if isinstance(value, (bool, np.bool_, int, np.integer)):
        return _number_text(value)
    if isinstance(value, (float, np.floating)):
        return _number_text(value) if math.isfinite(float(value)) else None    # json has no inf / nan: serde reads null
    if isinstance(value, str):
        return value
    return None


@dataclass
class HookOutcome:
    passed: Optional[bool] = None
    valid: Optional[bool] = None
    scalars: Dict[str, str] = field(default_factory=dict)


def read_post_run_outcome(obj: Any) -> HookOutcome:
    """lib.rs:2447-2470 on the already-parsed JSON value of post_run_result.json."""
    if not isinstance(obj, Mapping):
        return HookOutcome()
    p = obj.get("pass")
    passed = bool(p) if isinstance(p, (bool, np.bool_)) else None
    v = obj.get("valid")
    if isinstance(v, (bool, np.bool_)):
        valid = bool(v)
    else:
        status = obj.get("status")
        valid = (status != "invalid") if isinstance(status, str) else None
    scalars = {}
    for key in sorted(obj):
        cell = scalar_value(obj[key])
        if cell is not None:
            scalars[key] = cell
    return HookOutcome(passed, valid, scalars)


@dataclass
class RunMetric:
    run_id: str
    status: str = "ok"
    exit_ok: bool = True
    scored_pass: Optional[bool] = None
    scored_valid: Optional[bool] = None
    failure_reason: Optional[str] = None
    degraded: bool = False
    worker_id: Optional[int] = None
    wall_ms: int = 0
    hook_scalars: Dict[str, str] = field(default_factory=dict)
    db_path: str = ""
    run_dir: str = ""

    def valid(self) -> bool:                                       # lib.rs:342-344
        return self.status != "skipped" and (True if self.scored_valid is None else self.scored_valid)

    def passed(self) -> bool:                                      # lib.rs:338-340
        return self.valid() and not self.degraded and self.exit_ok and (True if self.scored_pass is None else self.scored_pass)


def summarize_hook_metrics(metrics: Sequence[RunMetric]) -> Dict[str, Dict[str, float]]:
    """lib.rs:1815-1852: per hook scalar over VALID runs: count, min, mean, p95 (index ceil(0.95 n) - 1), max."""
    grouped: Dict[str, List[float]] = {}
    for m in metrics:
        if not m.valid():
            continue
        for key, text in m.hook_scalars.items():
            if key in RESERVED_HOOK_KEYS:
                continue
            try:
                v = float(text)
            except ValueError:
                continue
            if math.isfinite(v):
                grouped.setdefault(key, []).append(v)
    out = {}
    for key in sorted(grouped):
        vals = sorted(grouped[key])
        n = len(vals)
        idx = min(max(int(math.ceil(n * 0.95)) - 1, 0), n - 1)
        out[key] = {"count": n, "min": vals[0], "mean": sum(vals) / n, "p95": vals[idx], "max": vals[-1]}
    return out


def write_results_csv(path, out_dir, metrics: Sequence[RunMetric]) -> List[str]:
    """lib.rs:3109-3187.  Returns the header."""
    hook_keys = sorted({k for m in metrics for k in m.hook_scalars if k not in RESERVED_HOOK_KEYS})
    header = RESULTS_HEADER + hook_keys
    opt = lambda b: "" if b is None else ("true" if b else "false")
    with open(path, "w", newline="") as f:
        wr = csv.writer(f, lineterminator="\n")
        wr.writerow(header)
        for m in metrics:
            result_json = os.path.join(m.run_dir, "result.json")
            rel = os.path.relpath(result_json, out_dir) if str(result_json).startswith(str(out_dir)) else result_json
            wr.writerow([m.run_id, m.status, opt(m.passed()), opt(m.valid()), opt(m.degraded), opt(m.scored_pass),
                         opt(m.scored_valid), m.failure_reason or "", "", "", "",
                         "" if m.worker_id is None else str(m.worker_id), str(int(m.wall_ms)), m.db_path, rel]
                        + [m.hook_scalars.get(k, "") for k in hook_keys])
    return header


def summarize_campaign(out_dir, metrics: Sequence[RunMetric], started_at: _dt.datetime, finished_at: _dt.datetime,
                       wall_ms: int, workers: int) -> Dict[str, Any]:
    """lib.rs:1715-1787."""
    passed = sum(m.passed() for m in metrics)
    skipped = sum(m.status == "skipped" for m in metrics)
    invalid = sum(m.status != "skipped" and not m.valid() for m in metrics)
    degraded = sum(m.status != "skipped" and m.valid() and m.degraded for m in metrics)
    failed = sum(m.status != "skipped" and m.valid() and not m.degraded and not m.passed() for m in metrics)
    total = sum(int(m.wall_ms) for m in metrics)
    disk = sum(p.stat().st_size for p in Path(out_dir).rglob("*") if p.is_file())
    rfc = lambda t: t.astimezone(_dt.timezone.utc).isoformat().replace("+00:00", "Z")
    return {
        "started_at": rfc(started_at), "finished_at": rfc(finished_at), "total_runs": len(metrics), "passed": passed,
        "failed": failed, "invalid": invalid, "degraded": degraded, "skipped": skipped, "workers": workers,
        "wall_ms": int(wall_ms), "total_run_wall_ms": total,
        "average_run_wall_ms": total / len(metrics) if metrics else 0.0,
        "max_run_wall_ms": max((int(m.wall_ms) for m in metrics), default=0),
        "parallel_efficiency": 0.0 if wall_ms == 0 or workers == 0 else total / (wall_ms * workers),
        "disk_bytes": disk,
        "resource_summary": {"samples": 0, "average_cpu_percent": 0.0, "peak_cpu_percent": 0.0, "peak_cpu_core_percent": 0.0,
                             "peak_load_average_1m": 0.0, "peak_context_switches_per_sec": 0.0, "peak_memory_used_kib": 0,
                             "peak_campaign_disk_bytes": disk},
        "sim_phase_summary": None,
        "phase_attribution": {"samples": 0, **{f"{s}_{p}_ms": 0.0 for s in ("average", "p95")
                                               for p in ("python_import", "compile", "loop", "teardown", "process_shutdown")}},
        "concurrency_summary": {"mean_active_runs": float(len(metrics)), "peak_active_runs": len(metrics), "buckets": []},
        "hook_metrics": summarize_hook_metrics(metrics),
        "pacing": None,
    }


def load_hook(path, name: str) -> Callable:
    """A lifecycle hook from a user file, loaded the way the reference's hook runner does (`python -m
    elodin.monte_carlo.run_hook HOOK.py post_run|post_campaign CTX.json`, libs/nox-py/python/elodin/monte_carlo/run_hook.py:12-58):
    the file's own directory goes on sys.path so sibling helpers (`mc_metrics.py`) import, a missing function is an error."""
    import importlib.util
    import sys
    path = Path(path)
    spec = importlib.util.spec_from_file_location(f"elodin_user_monte_carlo_hook_{path.stem}", path)
    if spec is None or spec.loader is None:
        raise RuntimeError(f"could not load hook: {path}")
    module = importlib.util.module_from_spec(spec)
    hook_dir = str(path.resolve().parent)
    if hook_dir not in sys.path:
        sys.path.insert(0, hook_dir)
    spec.loader.exec_module(module)
    hook = getattr(module, name, None)
    if hook is None or not callable(hook):
        raise RuntimeError(f"hook {path} does not define a callable `{name}`")
    return hook


def _namespace(value):
    """run_hook.py:26-31: the hook sees its JSON context as nested attribute namespaces."""
    if isinstance(value, dict):
        return SimpleNamespace(**{k: _namespace(v) for k, v in value.items()})
    if isinstance(value, list):
        return [_namespace(v) for v in value]
    return value


def run_post_campaign(out_dir, hook) -> Any:
    """The post_campaign step of `elodin monte-carlo run` (lib.rs:1335-1347): write campaign_hook_context.json, call
    `hook(ctx)` (a callable or a path to a hook file defining `post_campaign`), write post_campaign_result.json."""
    out = Path(out_dir)
    if not callable(hook):
        hook = load_hook(hook, "post_campaign")
    payload = {"out_dir": str(out), "results": str(out / "results.csv"), "perf": str(out / "perf.csv"), "memory": None,
               "resources": str(out / "resources.csv"), "summary": str(out / "summary.json")}
    context = out / "campaign_hook_context.json"
    context.write_text(json.dumps(payload, indent=2) + "\n")
    result = hook(_namespace(payload))
    if result is not None:
        context.with_name("post_campaign_result.json").write_text(json.dumps(_jsonable(result), indent=2, sort_keys=True))
    return result


def write_campaign(out_dir, plan, results: np.ndarray, result_names: Sequence[str], *, wall_ms: float, workers: int = 1,
                   post_run: Optional[Callable] = None, result_record: Optional[Callable] = None,
                   failed_rows: Optional[np.ndarray] = None, rows_per_worker: Optional[int] = None,
                   started_at: Optional[_dt.datetime] = None) -> Dict[str, Any]:
    """Write plan.csv, runs/<run_id>/result.json (+ post_run_result.json), results.csv and summary.json.

    plan            monte_carlo.Plan (run ids, seeds, parameters) — row i of `results` belongs to plan.run_ids[i]
    results         [n_runs, len(result_names)] in-kernel result rows (e.g. models.apollo.RESULT_NAMES)
    result_record   (row dict) -> dict written as result.json (default: the row with 0/1 flags left numeric)
    post_run        scoring hook with the reference's signature `post_run(ctx) -> dict` — a callable, or the path of a hook
                    file defining `post_run` (e.g. the reference's examples/apollo-lander/hooks/score.py, unmodified).  It
                    gets the context the runner writes to post_run_context.json (lib.rs:2264-2277: run_id, params, meta,
                    metrics, db_path, run_dir); its outcome decides pass / valid like lib.rs:2269-2290
    failed_rows     bool [n_runs]: rollouts whose state went non-finite (sixdof_count_nonfinite) -> status "failed"
    """
    out = Path(out_dir)
    (out / "runs").mkdir(parents=True, exist_ok=True)
    finished = _dt.datetime.now(_dt.timezone.utc)
    started = started_at or finished - _dt.timedelta(milliseconds=float(wall_ms))
    (out / "plan.csv").write_bytes(plan.to_csv().encode())   # keeps csv.DictWriter's \r\n row ends, like sample.py's file
    results = np.asarray(results)
    n = len(plan)
    if results.shape[0] != n:
        raise ValueError(f"{results.shape[0]} result rows for a plan of {n} runs")
    per_run_ms = int(round(float(wall_ms) / max(n, 1)))
    if post_run is not None and not callable(post_run):
        post_run = load_hook(post_run, "post_run")
    metrics: List[RunMetric] = []
    for i, run_id in enumerate(plan.run_ids):
        run_dir = out / "runs" / run_id
        run_dir.mkdir(exist_ok=True)
        row = {name: float(results[i, k]) for k, name in enumerate(result_names)}
        record = result_record(row) if result_record else row
        (run_dir / "result.json").write_text(json.dumps(_jsonable(record), indent=2, sort_keys=True, allow_nan=False) + "\n")
        bad = bool(failed_rows[i]) if failed_rows is not None else False
        m = RunMetric(run_id, status="failed" if bad else "ok", exit_ok=not bad,
                      failure_reason="non-finite state" if bad else None, wall_ms=per_run_ms,
                      worker_id=(i // rows_per_worker) if rows_per_worker else 0,
                      db_path=str(Path("runs") / run_id / "db"), run_dir=str(run_dir))
        if post_run is not None and not bad:
            row_i = plan.rows[i]
            payload = {"run_id": run_id,
                       "params": {k[len("param."):]: _plan_cell(v) for k, v in row_i.items() if k.startswith("param.")},
                       "meta": {k[len("meta."):]: _plan_cell(v) for k, v in row_i.items() if k.startswith("meta.")},
                       "metrics": {"run_id": run_id, "status": m.status, "exit_ok": m.exit_ok, "wall_ms": m.wall_ms,
                                   "worker_id": m.worker_id, "db_path": m.db_path, "run_dir": m.run_dir},
                       "db_path": m.db_path, "run_dir": str(run_dir)}
            (run_dir / "post_run_context.json").write_text(json.dumps(payload, indent=2) + "\n")
            ctx = _namespace(payload)
            ctx.params = _ParamsView(payload["params"])   # `ctx.params.x` like run_hook.py's namespace AND `ctx.params["x"]`
            ctx.out_dir, ctx.seed = str(out), int(plan.seeds[i])
            outcome_json = post_run(ctx)
            (run_dir / "post_run_result.json").write_text(json.dumps(_jsonable(outcome_json), indent=2, sort_keys=True) + "\n")
            outcome = read_post_run_outcome(_jsonable(outcome_json))
            m.scored_pass, m.scored_valid, m.hook_scalars = outcome.passed, outcome.valid, outcome.scalars
            if not m.valid():
                m.status = "invalid"                                   # lib.rs:2284-2286
        metrics.append(m)
    header = write_results_csv(out / "results.csv", out, metrics)
    summary = summarize_campaign(out, metrics, started, finished, int(round(wall_ms)), workers)
    (out / "summary.json").write_text(json.dumps(summary, indent=2) + "\n")
    return {"summary": summary, "results_header": header, "metrics": metrics}


class _ParamsView(dict):
    """The run's parameters for a hook: attribute access like the namespace the reference's run_hook.py builds
    (`_namespace` recursion, elodin/monte_carlo/run_hook.py) and a mapping as well (`ctx.params["x"]`, `.items()`)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None


def _plan_cell(v):
    """A plan.csv cell as the JSON value the runner's context carries (numbers as numbers, anything else as text)."""
    if isinstance(v, (int, float, np.integer, np.floating)) and not isinstance(v, bool):
        return _jsonable(v)
    try:
        f = float(v)
        return int(v) if str(v).lstrip("+-").isdigit() else f
    except (TypeError, ValueError):
        return v


def _jsonable(obj):
    """What the hook's dict looks like after a JSON round trip (inf / nan -> null, numpy scalars -> Python)."""
    if isinstance(obj, Mapping):
        return {str(k): _jsonable(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_jsonable(v) for v in obj]
    if isinstance(obj, (bool, np.bool_)):
        return bool(obj)
    if isinstance(obj, (int, np.integer)):
        return int(obj)
    if isinstance(obj, (float, np.floating)):
        return float(obj) if math.isfinite(float(obj)) else None
    return obj
