"""The reference's OWN campaign hooks, unmodified, on a campaign directory this repo wrote.

examples/apollo-lander/hooks/{score,report,ci_score,ci_gate}.py + mc_metrics.py are stdlib-only Python.  A campaign — the
example's 30-run LHS plan flown by the pinned CPU oracle (same rows / results layout as the GPU campaign,
models/apollo.run_campaign) — is written with elodin_amd.campaign; then
  (1) campaign.write_campaign / run_post_campaign call the hook FILES in-process with the runner's context contract
      (lib.rs:2264-2277, 1335-1347; run_hook.py:12-58), and
  (2) the reference's own hook runner, libs/nox-py/python/elodin/monte_carlo/run_hook.py, is run as a subprocess on the
      context files this repo wrote,
and both must accept the directory and agree.  Needs /root/reference (build container); skipped elsewhere."""
import csv
import json
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from elodin_amd import campaign as cp
from elodin_amd import monte_carlo as mc
from elodin_amd.models import apollo
from oracle.apollo import ApolloOracle

REF = Path("/root/reference")
HOOKS = REF / "examples" / "apollo-lander" / "hooks"
RUN_HOOK = REF / "libs" / "nox-py" / "python" / "elodin" / "monte_carlo" / "run_hook.py"
pytestmark = pytest.mark.skipif(not HOOKS.exists(), reason="needs the reference checkout (build container only)")


@pytest.fixture(scope="module")
def flown():
    ref = apollo.load_reference()
    plan = mc.materialize(mc.load_spec(Path(__file__).parent / "golden" / "plans" / "apollo.toml"))
    o = ApolloOracle(apollo.initial_columns(plan.table(), ref), ref, max_ticks=apollo.max_ticks(ref))
    o.step(apollo.max_ticks(ref), threads=8)
    return plan, o.result.copy()


def test_reference_score_and_report_hooks_consume_the_campaign_directory(flown, tmp_path):
    plan, results = flown
    out = tmp_path / "campaign"
    info = cp.write_campaign(out, plan, results, apollo.RESULT_NAMES, wall_ms=900.0, workers=1, post_run=HOOKS / "score.py",
                             result_record=apollo.result_record)
    n = len(plan)
    # score.py ran on every run and produced the reference's scalar set
    post = json.loads((out / "runs" / plan.run_ids[0] / "post_run_result.json").read_text())
    assert set(post) == {"landed", "soft_landing", "valid", "pass", "touchdown_speed_mps", "horizontal_speed_mps", "fuel_remaining_kg",
                         "rcs_fuel_remaining_kg", "traj_rmse_m", "pitch_rmse_deg", "downrange_miss_m"}
    rows = list(csv.DictReader(open(out / "results.csv")))
    assert len(rows) == n and all(r["status"] == "ok" and r["valid"] == "true" for r in rows)
    soft = results[:, apollo.RESULT_NAMES.index("soft_landing")] > 0.5
    assert [r["passed"] == "true" for r in rows] == soft.tolist()               # the hook's verdict = the sim's soft_landing flag
    assert all(abs(float(r["touchdown_speed_mps"]) - results[i, 0]) < 1e-12 for i, r in enumerate(rows))
    s = info["summary"]
    assert s["total_runs"] == n and s["passed"] == int(soft.sum()) and s["failed"] == n - int(soft.sum()) and s["invalid"] == 0
    assert s["hook_metrics"]["traj_rmse_m"]["count"] == n
    # report.py, in-process through the runner's post_campaign contract
    rep = cp.run_post_campaign(out, HOOKS / "report.py")
    assert rep["completed"] == int(soft.sum()) and rep["soft_landings"] == int(soft.sum()) and rep["success_rate"] == 1.0
    text = (out / "post_campaign" / "apollo_lander_report.txt").read_text()
    assert f"runs completed: {int(soft.sum())}/{n}" in text and "best-fit params:" in text and "init_altitude_m:" in text
    best = rep["best_run"]
    rmse = {r["run_id"]: float(r["traj_rmse_m"]) for r in rows if r["passed"] == "true"}
    assert best == min(rmse, key=rmse.get) and abs(rep["best_traj_rmse_m"] - rmse[best]) < 1e-12
    assert json.loads((out / "post_campaign_result.json").read_text())["best_run"] == best

    # (2) the reference's own hook runner on the context files this repo wrote: same outcome files
    run0 = out / "runs" / plan.run_ids[0]
    before = json.loads((run0 / "post_run_result.json").read_text())
    (run0 / "post_run_result.json").unlink()
    subprocess.run([sys.executable, str(RUN_HOOK), str(HOOKS / "score.py"), "post_run", str(run0 / "post_run_context.json")], check=True)
    assert json.loads((run0 / "post_run_result.json").read_text()) == before
    (out / "post_campaign_result.json").unlink()
    r = subprocess.run([sys.executable, str(RUN_HOOK), str(HOOKS / "report.py"), "post_campaign", str(out / "campaign_hook_context.json")],
                       check=True, capture_output=True, text=True)
    assert "Apollo 11 lander Monte Carlo report" in r.stdout
    assert json.loads((out / "post_campaign_result.json").read_text())["best_run"] == best


def test_reference_ci_gate_passes_a_clean_campaign_and_fails_a_dirty_one(flown, tmp_path):
    plan, results = flown
    out = tmp_path / "ci"
    cp.write_campaign(out, plan, results, apollo.RESULT_NAMES, wall_ms=900.0, post_run=HOOKS / "ci_score.py", result_record=apollo.result_record)
    gate = cp.run_post_campaign(out, HOOKS / "ci_gate.py")         # ci_score passes any run with a result artefact
    assert gate == {"pass": True, "passed": len(plan), "failed": 0, "invalid": 0, "total_runs": len(plan)}
    # score.py's verdicts instead: runs that missed the soft-landing criteria are failures and the gate must raise, naming them
    out2 = tmp_path / "ci_scored"
    res2 = results.copy()
    res2[5, apollo.RESULT_NAMES.index("soft_landing")] = 0.0
    cp.write_campaign(out2, plan, res2, apollo.RESULT_NAMES, wall_ms=900.0, post_run=HOOKS / "score.py", result_record=apollo.result_record)
    with pytest.raises(RuntimeError, match=r"CI gate: \d+ failed and 0 invalid of 30 run\(s\).*run_0000005"):
        cp.run_post_campaign(out2, HOOKS / "ci_gate.py")
    bad = subprocess.run([sys.executable, str(RUN_HOOK), str(HOOKS / "ci_gate.py"), "post_campaign", str(out2 / "campaign_hook_context.json")],
                         capture_output=True, text=True)
    assert bad.returncode != 0 and "run_0000005" in bad.stderr
