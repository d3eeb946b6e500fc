"""a23: the product-side plan sampler reproduces, byte for byte, plans written by the reference's own
sampler (golden files made by tests/golden/make_plan_golden.py, which imports the reference's sample.py)."""
from pathlib import Path

import numpy as np
import pytest

from elodin_amd import monte_carlo as mc

PLANS = Path(__file__).resolve().parent / "golden" / "plans"


@pytest.mark.parametrize("name", ["apollo", "apollo_512", "mixed", "lhs_normal", "no_mc"])
def test_plan_csv_is_byte_identical_to_reference_sampler(name):
    plan = mc.materialize(mc.load_spec(PLANS / f"{name}.toml"))
    ref = (PLANS / f"{name}.plan.csv").read_bytes().decode()
    assert plan.to_csv() == ref


def test_plan_table_and_ids():
    plan = mc.materialize(mc.load_spec(PLANS / "apollo.toml"))
    assert len(plan) == 30 and plan.run_ids[0] == "run_0000000" and plan.run_ids[29] == "run_0000029"
    assert plan.seeds.dtype == np.uint64 and plan.seeds.tolist() == list(range(1, 31))
    t = plan.table()
    assert t.shape == (30, 17) and plan.param_names == sorted(plan.param_names)
    # LHS: every column has exactly one sample per stratum
    lo, hi = 11650.0, 11950.0
    col = t[:, plan.param_names.index("init_altitude_m")]
    strata = np.floor((col - lo) / (hi - lo) * 30).astype(int)
    assert sorted(strata.tolist()) == list(range(30))
    # explicit column selection with defaults for parameters the spec does not vary
    t2 = plan.table(["dry_mass_kg", "not_in_plan"], defaults={"not_in_plan": 3.0})
    assert np.all(t2[:, 1] == 3.0) and np.array_equal(t2[:, 0], t[:, plan.param_names.index("dry_mass_kg")])
    with pytest.raises(KeyError):
        plan.table(["missing"])


def test_spec_validation_messages():
    def spec(var):
        return {"monte_carlo": {"n_samples": 2, "seed": 1, "variables": {"x": var}}}
    with pytest.raises(ValueError, match='unknown dist "gauss"'):
        mc.materialize(spec({"dist": "gauss"}))
    with pytest.raises(ValueError, match="needs min/max"):
        mc.materialize(spec({"dist": "uniform", "min": 0.0}))
    with pytest.raises(ValueError, match="positive min/max"):
        mc.materialize(spec({"dist": "loguniform", "min": 0.0, "max": 1.0}))
    with pytest.raises(ValueError, match="needs mean/std"):
        mc.materialize(spec({"dist": "normal", "mean": 0.0}))
    with pytest.raises(ValueError, match="n_samples must be >= 1"):
        mc.materialize({"monte_carlo": {"n_samples": 0}})
    with pytest.raises(ValueError, match='unknown method "sobol"'):
        mc.materialize({"monte_carlo": {"n_samples": 1, "method": "sobol"}})
