#!/usr/bin/env python3
"""`elodin monte-carlo run examples/apollo-lander/...` as one GPU job: the plan is sampled exactly like the reference's
sampler would, every rollout is a row of the entity axis, ranks (if launched under torch.distributed.run) each fly a
contiguous block of run ids, results are gathered into run-id order.
  python examples/apollo_campaign.py [n_runs] [out_dir]
With an out_dir the campaign directory `elodin monte-carlo run` would leave (plan.csv, runs/<id>/result.json,
results.csv, summary.json) is written, scored by a post_run hook shaped like the example's hooks/score.py."""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from elodin_amd import monte_carlo as mc  # noqa: E402
from elodin_amd.models import apollo  # noqa: E402


def score(ctx):
    """hooks/score.py's contract: read the run's result.json, return scalars plus the pass / valid verdict."""
    r = json.loads((Path(ctx.run_dir) / "result.json").read_text())
    return {"landed": r["landed"], "soft_landing": r["soft_landing"], "valid": bool(r), "pass": r["soft_landing"],
            "touchdown_speed_mps": r["touchdown_speed"], "horizontal_speed_mps": r["horizontal_speed"],
            "fuel_remaining_kg": r["fuel_remaining"], "rcs_fuel_remaining_kg": r["rcs_fuel_remaining"],
            "traj_rmse_m": r["traj_rmse"], "pitch_rmse_deg": r["pitch_rmse"], "downrange_miss_m": r["downrange_miss"]}


def main(n_runs=8192, out_dir=None):
    spec = mc.load_spec(ROOT / "tests" / "golden" / "plans" / "apollo.toml")   # the example's spec.toml
    spec["monte_carlo"]["n_samples"] = n_runs
    plan = mc.materialize(spec)
    ref = apollo.load_reference()
    t0 = time.perf_counter()
    res = apollo.run_campaign(plan.table(), len(plan), apollo.max_ticks(ref))
    wall_ms = (time.perf_counter() - t0) * 1e3
    r = dict(zip(apollo.RESULT_NAMES, res.T))
    print(f"{len(plan)} runs: landed {r['landed'].mean():.3f}, soft {r['soft_landing'].mean():.3f}, "
          f"touchdown {np.median(r['touchdown_speed']):.2f} m/s median, fuel left {np.median(r['fuel_remaining']):.0f} kg median")
    worst = int(np.argmax(r["horizontal_speed"]))
    print("worst horizontal speed:", plan.run_ids[worst], f"{r['horizontal_speed'][worst]:.2f} m/s")
    if out_dir is not None:
        from elodin_amd import campaign
        art = campaign.write_campaign(out_dir, plan, res, apollo.RESULT_NAMES, wall_ms=wall_ms, post_run=score,
                                      result_record=apollo.result_record, failed_rows=~np.isfinite(res).all(axis=1))
        s = art["summary"]
        print(f"wrote {out_dir}: {s['total_runs']} runs, {s['passed']} passed, {s['failed']} failed; touchdown p95 "
              f"{s['hook_metrics']['touchdown_speed_mps']['p95']:.2f} m/s")
    return res


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 8192, sys.argv[2] if len(sys.argv) > 2 else None)
