#!/usr/bin/env python3
"""`elodin monte-carlo run examples/apollo-lander/...` as one GPU job: the plan is sampled exactly like the reference's
sampler would, every rollout is a row of the entity axis, ranks (if launched under torch.distributed.run) each fly a
contiguous block of run ids, results are gathered into run-id order.  python examples/apollo_campaign.py [n_runs]"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from elodin_amd import monte_carlo as mc  # noqa: E402
from elodin_amd.models import apollo  # noqa: E402


def main(n_runs=8192):
    spec = mc.load_spec(ROOT / "tests" / "golden" / "plans" / "apollo.toml")   # the example's spec.toml
    spec["monte_carlo"]["n_samples"] = n_runs
    plan = mc.materialize(spec)
    ref = apollo.load_reference()
    res = apollo.run_campaign(plan.table(), len(plan), apollo.max_ticks(ref))
    r = dict(zip(apollo.RESULT_NAMES, res.T))
    print(f"{len(plan)} runs: landed {r['landed'].mean():.3f}, soft {r['soft_landing'].mean():.3f}, "
          f"touchdown {np.median(r['touchdown_speed']):.2f} m/s median, fuel left {np.median(r['fuel_remaining']):.0f} kg median")
    worst = int(np.argmax(r["horizontal_speed"]))
    print("worst horizontal speed:", plan.run_ids[worst], f"{r['horizontal_speed'][worst]:.2f} m/s")
    return res


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 8192)
