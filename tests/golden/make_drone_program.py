#!/usr/bin/env python3
"""The reference's drone example compiled by THIS repo's front end, frozen for the GPU box.

examples/drone/main.py is imported UNMODIFIED under elodin_amd.compat (build container only: it lives in /root/reference),
its recorded `world.run(system(), simulation_rate=300, telemetry_rate=100, ...)` is resolved like World.build resolves it
(control systems on the first of three integrator sub-steps, `six_dof(1/900, SemiImplicit) | imu | telemetry` on every one),
and what the GPU test needs is written to tests/golden/drone_program.json: the HIP source the code generator emits for the
program (this repo's compiler output, not reference code), its column table and the spawned initial columns.
tests/test_gpu_drone.py compiles that source on the GPU box, runs 100 ticks and compares with the reference's CI baseline
(tests/golden/drone.json <- scripts/ci/baseline/drone-csv).   python tests/golden/make_drone_program.py"""
import sys as _sys
_sys.dont_write_bytecode = True      # the reference checkout is read-only: no __pycache__ next to what is imported from it
import importlib.util
import json
import sys
import types
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.setrecursionlimit(20000)
REF = Path("/root/reference/examples/drone")

import numpy as np  # noqa: E402

import elodin_amd.compat as compat  # noqa: E402
from elodin_amd import codegen  # noqa: E402

compat.install(run="record")
sys.modules.setdefault("polars", types.ModuleType("polars"))       # main.py imports it for its --telemetry branch only
sys.path.insert(0, str(REF))
spec = importlib.util.spec_from_file_location("ref_drone_main", REF / "main.py")
main = importlib.util.module_from_spec(spec)
sys.modules["ref_drone_main"] = main
spec.loader.exec_module(main)
run = main.world.compat_run
plan = main.world.build(run["system"], simulation_rate=run["simulation_rate"], telemetry_rate=run["telemetry_rate"], _dry=True)
tp = plan["effectors"].trace()
codegen.build(tp, "float64", plan["integrator"])            # settles on the first variant that fits a wave's registers
doc = {
    "variant": codegen.last_variant[0],
    "source": codegen.generate_variant(tp, codegen.last_variant[0], "float64", plan["integrator"]),
    "columns": [[n, w] for n, w in tp.columns], "mats": {k: list(v) for k, v in tp.table.mats.items()},
    "substeps": plan["substeps"], "integrator": plan["integrator"], "simulation_time_step": plan["dt"], "time_step": plan["time_step"],
    "simulation_rate": run["simulation_rate"], "telemetry_rate": run["telemetry_rate"],
    "body": {k: np.asarray(v, dtype=np.float64).tolist() for k, v in plan["body"].items()},
    "initial": {n: np.asarray(plan["columns"][n], dtype=np.float64).reshape(len(plan["body"]["world_pos"]), -1).tolist() for n, _ in tp.columns},
    "systems": {"pre": [[s.name, s.every, s.phase] for s in tp.pre], "post": [[s.name, s.every, s.phase] for s in tp.post]},
}
out = ROOT / "tests" / "golden" / "drone_program.json"
out.write_text(json.dumps(doc))
print(out, out.stat().st_size, "bytes; variant", doc["variant"] + ";", len(doc["columns"]), "columns,", doc["source"].count("\n"), "source lines, substeps", doc["substeps"])
