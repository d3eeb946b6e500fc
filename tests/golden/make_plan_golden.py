#!/usr/bin/env python3
"""Generate golden Monte-Carlo plans by running the REFERENCE's own sampler
(libs/nox-py/python/elodin/monte_carlo/sample.py is stdlib-only Python, importable here) on a
few specs.  Outputs tests/golden/plans/<name>.toml + <name>.plan.csv.  Build container only."""
import sys as _sys
_sys.dont_write_bytecode = True      # the reference checkout is read-only: no __pycache__ next to what is imported from it
import importlib.util
import shutil
import sys
from pathlib import Path

REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
OUT = Path(__file__).resolve().parent / "plans"
OUT.mkdir(exist_ok=True)

spec = importlib.util.spec_from_file_location("ref_sample", REF / "libs/nox-py/python/elodin/monte_carlo/sample.py")
ref_sample = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_sample)

SPECS = {
    # the reference's own Apollo spec (LHS, seed 19690720, 30 samples, 17 uniforms)
    "apollo": (REF / "examples/apollo-lander/spec.toml").read_text(),
    # same with n_samples raised to a BASELINE-like shard size
    "apollo_512": (REF / "examples/apollo-lander/spec.toml").read_text().replace("n_samples = 30", "n_samples = 512"),
    "mixed": '''
[sim_sweep]
integrator = ["rk4", "semi"]
stage = [1, 2, 3]

[meta_sweep]
label = ["a", "b"]

[monte_carlo]
n_samples = 7
seed = 42
method = "random"

[monte_carlo.variables]
zeta = { dist = "normal", mean = 1.5, std = 0.25 }
alpha = { dist = "loguniform", lo = 0.001, hi = 10.0 }
pick = { dist = "choice", values = [3, 5, 8, 13] }
hold = { dist = "fixed", value = 2.5 }
beta = { dist = "uniform", low = -1.0, high = 1.0 }
''',
    "lhs_normal": '''
[monte_carlo]
n_samples = 64
seed = 7
[monte_carlo.variables]
a = { dist = "normal", mean = 0.0, std = 1.0 }
b = { dist = "uniform", min = 0.0, max = 1.0 }
''',
    "no_mc": '''
[sim_sweep]
x = [1.0, 2.0]
''',
}

for name, text in SPECS.items():
    (OUT / f"{name}.toml").write_text(text)
    ref_sample.materialize(OUT / f"{name}.toml", OUT / f"{name}.plan.csv")
    print(name, sum(1 for _ in open(OUT / f"{name}.plan.csv")) - 1, "runs")
