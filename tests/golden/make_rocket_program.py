#!/usr/bin/env python3
"""The reference's rocket example compiled by THIS repo's front end, frozen for the GPU box.

examples/rocket/main.py is imported UNMODIFIED under elodin_amd.compat (build container only: it lives in /root/reference;
its polars table preparation runs on elodin_amd/compat_polars.py), its recorded `world.run(system, simulation_rate=120, ...)`
is resolved like World.build resolves it, and what the GPU test needs is written to tests/golden/rocket_program.json: the HIP
source the code generator emits for the program (this repo's compiler output, not reference code), its column table, the
window component and the spawned initial columns.  tests/test_gpu_rocket.py compiles that source on the GPU box, runs 100
ticks and compares with the reference's CI baseline (tests/golden/rocket.json <- scripts/ci/baseline/rocket-csv).
python tests/golden/make_rocket_program.py"""
import sys as _sys
_sys.dont_write_bytecode = True      # the reference checkout is read-only: no __pycache__ next to what is imported from it
import importlib.util
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.setrecursionlimit(50000)
REF = Path("/root/reference/examples/rocket")

import numpy as np  # noqa: E402

import elodin_amd.compat as compat  # noqa: E402
from elodin_amd import codegen  # noqa: E402

compat.install(run="record")
sys.path.insert(0, str(REF))
spec = importlib.util.spec_from_file_location("ref_rocket_main", REF / "main.py")
main = importlib.util.module_from_spec(spec)
sys.modules["ref_rocket_main"] = main
spec.loader.exec_module(main)
world = next(v for v in vars(main).values() if hasattr(v, "compat_run"))
run = world.compat_run
plan = world.build(run["system"], simulation_rate=run["simulation_rate"], telemetry_rate=run["telemetry_rate"], _dry=True)
tp = plan["effectors"].trace()
codegen.build(tp, "float64", plan["integrator"])            # settles on the first variant that fits a wave's registers
n = len(plan["body"]["world_pos"])
doc = {
    "variant": codegen.last_variant[0],
    "source": codegen.generate_variant(tp, codegen.last_variant[0], "float64", plan["integrator"]),
    "columns": [[n_, w] for n_, w in tp.columns], "mats": {k: list(v) for k, v in tp.table.mats.items()},
    "windows": {k: list(v) for k, v in tp.windows.items()},
    "integrator": plan["integrator"], "simulation_time_step": plan["dt"], "time_step": plan["time_step"],
    "simulation_rate": run["simulation_rate"],
    "body": {k: np.asarray(v, dtype=np.float64).tolist() for k, v in plan["body"].items()},
    "initial": {n_: np.asarray(plan["columns"][n_], dtype=np.float64).reshape(n, -1).tolist() for n_, _ in tp.columns if not n_.endswith("#head")},
    "systems": {"pre": [[s.name, s.every, s.phase] for s in tp.pre], "post": [[s.name, s.every, s.phase] for s in tp.post]},
}
out = ROOT / "tests" / "golden" / "rocket_program.json"
out.write_text(json.dumps(doc))
print(out, out.stat().st_size, "bytes; variant", doc["variant"] + ";", len(doc["columns"]), "columns,", doc["source"].count("\n"), "source lines, windows", doc["windows"])
