"""Shared by the CPU and GPU StableHLO-ingestion tests: the reference's op-test known answers (tests/golden/stablehlo_ops.json <-
libs/cranelift-mlir/tests/ops.rs, make_stablehlo_ops_golden.py) as per-entity systems.  TEST INFRASTRUCTURE."""
import json
from pathlib import Path

import numpy as np

from elodin_amd import dsl
from elodin_amd import stablehlo as sh

DOC = json.loads((Path(__file__).parent / "golden" / "stablehlo_ops.json").read_text())
CASES = DOC["cases"]
# the pieces of a dumped WORLD tick (libs/cranelift-mlir/tests/test_gather_3body.rs ... test_uniform_pipeline.rs): same layout, the
# expected values as full lists (make_stablehlo_world_golden.py)
WORLD_DOC = json.loads((Path(__file__).parent / "golden" / "stablehlo_world_fragments.json").read_text())
WORLD_CASES = [dict(c, expected={k: {"type": e["type"], "values": {str(j): v for j, v in enumerate(e["values"])}} for k, e in c["expected"].items()})
               for c in WORLD_DOC["cases"]]
# ops elodin_amd.stablehlo refuses by name (its docstring lists them) and the one case beyond f64's integers
UNSUPPORTED = {}          # (round 4, late: dgetrf / dgesv / dgesdd, scatter, real_dynamic_slice, reduce_window, select_and_scatter were the last)
BEYOND_F64_INTEGERS = set()          # round 5: ui64 elements are two uint32 words (stablehlo.U64) — 1 + (2^64 - 1) wraps to 0 exactly


def build(case, prefix=""):
    """(system, {input column: [w] values}, {output column: (width, {index: expected})}) of one case."""
    main = sh.parse_module(case["mlir"])["main"]
    ins = [f"{prefix}in{k}" for k in range(len(main.args))]
    outs = [f"{prefix}out{k}" for k in range(len(main.result_types))]
    system = sh.system(case["mlir"], ins, outs, name=case["name"])
    values = {}
    for nm, inp, (_, ty) in zip(ins, case["inputs"], main.args):
        v = np.array(inp["values"], dtype=np.float64)
        assert v.size == ty.size, (case["name"], nm, v.size, ty)
        values[nm] = v
    expect = {nm: (ty.size, {}) for nm, ty in zip(outs, main.result_types)}
    for k, e in case["expected"].items():
        expect[f"{prefix}out{k}"][1].update({int(j): float(v) for j, v in e["values"].items()})
    return system, values, expect


def check(case_name, got_row, width, expected, tol):
    assert len(got_row) == width
    for j, want in expected.items():
        g = float(got_row[j])
        if want != want:
            assert g != g, (case_name, j, g)
        elif abs(want) == float("inf"):
            assert g == want, (case_name, j, g, want)
        else:
            assert abs(g - want) <= tol * max(1.0, abs(want)), (case_name, j, g, want)
