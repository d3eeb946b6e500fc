#!/usr/bin/env python3
"""The reference's Falcon 9 plant compiled by THIS repo's front end, frozen for the GPU box (the counterpart of
make_drone_program.py / make_cube_sat_program.py): examples/falcon9/sim.py imported UNMODIFIED under elodin_amd.compat, built as
tests/falcon9_unmodified_util.py describes for the `maxq` window of tests/golden/falcon9_plant.json (transonic, engines
running, TVC + fins + RCS active, wind), and written to tests/golden/falcon9_plant_program.json: the generated HIP source (this
repo's compiler output; the build settles on the memory-image variant for the f64 program, like this repo's own model of the
vehicle does), its column table and the window's spawn state.   python tests/golden/make_falcon9_plant_program.py"""
import sys as _sys
_sys.dont_write_bytecode = True      # the reference checkout is read-only: no __pycache__ next to what is imported from it
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.setrecursionlimit(50000)

import elodin_amd.compat as compat  # noqa: E402
from elodin_amd import codegen  # noqa: E402

compat.install(run="record")
from tests import falcon9_unmodified_util as fu  # noqa: E402

import base64  # noqa: E402
import os  # noqa: E402
import zlib  # noqa: E402

pack = lambda text: base64.b64encode(zlib.compress(text.encode(), 9)).decode()      # sources are ~230 KB each: stored deflated
CASE = "maxq"
plan, tp, a = fu.build(CASE)
codegen.build(tp, "float64", plan["integrator"])            # settles on the first variant that fits a wave's registers
variant = codegen.last_variant[0]
os.environ["SIXDOF_GUARD_SELECTS"] = "1"                    # the same program with guarded selects (codegen._Emitter.block)
codegen.build(tp, "float64", plan["integrator"])
variant_guarded = codegen.last_variant[0]
source_f64_guarded = codegen.generate_variant(tp, variant_guarded, "float64", plan["integrator"])
os.environ.pop("SIXDOF_GUARD_SELECTS")
doc = {
    "case": CASE, "variant": variant, "packed": ["source", "source_f64_guarded", "source_f32_fast", "source_f32_fast_guarded"],
    "source": pack(codegen.generate_variant(tp, variant, "float64", plan["integrator"])),
    "variant_guarded": variant_guarded, "source_f64_guarded": pack(source_f64_guarded),
    "source_f32_fast_guarded": pack(codegen.generate_source(tp, "float32", plan["integrator"], fast_math=True, guard_selects=True)),
    "columns": [[n, w] for n, w in tp.columns], "mats": {k: list(v) for k, v in tp.table.mats.items()},
    # the campaign-style build of the same program (float32, hardware transcendentals): register-resident, for the throughput
    # measurement of tools/falcon9_unmodified_throughput.py
    "source_f32_fast": pack(codegen.generate_variant(tp, "program", "float32", plan["integrator"], fast_math=True)),
    "integrator": plan["integrator"], "simulation_time_step": plan["dt"],
    "initial": {k: v.tolist() for k, v in a.items()},
}
out = ROOT / "tests" / "golden" / "falcon9_plant_program.json"
out.write_text(json.dumps(doc))
print(out, out.stat().st_size, "bytes; variants", doc["variant"], "/ guarded:", doc["variant_guarded"] + ";", len(doc["columns"]), "columns;",
      source_f64_guarded.count("if (__any("), "guarded selects")
