"""StableHLO text ingestion (elodin_amd/stablehlo.py; SURVEY §8 f1 in its literal form) on the CPU: every known answer extracted
from the reference's own op tests (libs/cranelift-mlir/tests/ops.rs) — parse -> scalar DAG -> the numpy walk of the traced
program (tests/dsl_numpy.py: the DAG codegen.py turns into kernel code).  tests/test_gpu_stablehlo.py runs the same cases
through the generated gfx950 kernel."""
import numpy as np
import pytest

from elodin_amd import dsl
from elodin_amd import stablehlo as sh
from tests import dsl_numpy
from tests import stablehlo_util as U


def walk(system, values, expect):
    widths = {n: len(v) for n, v in values.items()}
    widths.update({n: w for n, (w, _) in expect.items()})
    tp = dsl.Program([system], dsl.Pipe([]), []).trace(widths)
    n = 2
    comps = {nm: np.tile(v.reshape(1, -1), (n, 1)) for nm, v in values.items()}
    for nm, w in tp.columns:
        comps.setdefault(nm, np.zeros((n, w)))
    pos, vel, acc, inertia = np.tile([0, 0, 0, 1.0, 0, 0, 0], (n, 1)), np.zeros((n, 6)), np.zeros((n, 6)), np.ones((n, 7))
    dsl_numpy.program_tick_systems_only(tp, pos, vel, acc, inertia, comps, 1)
    return comps


@pytest.mark.parametrize("case", U.CASES, ids=[c["name"] for c in U.CASES])
def test_reference_op_test_known_answers_on_the_cpu_walker(case):
    if case["name"] in U.UNSUPPORTED:
        with pytest.raises(NotImplementedError, match=U.UNSUPPORTED[case["name"]]):      # refused by name when the module is traced
            walk(*U.build(case))
        return
    if case["name"] in U.BEYOND_F64_INTEGERS:
        pytest.skip("an unsigned 64-bit constant beyond the integers a double holds")
    system, values, expect = U.build(case)
    comps = walk(system, values, expect)
    for nm, (w, exp) in expect.items():
        U.check(case["name"], comps[nm][0], w, exp, 1e-9)
        assert np.array_equal(comps[nm][0], comps[nm][1], equal_nan=True)


def test_the_fixture_covers_the_front_ends_op_set():
    import re
    ops = set()
    for c in U.CASES:
        if c["name"] not in U.UNSUPPORTED:
            ops |= set(re.findall(r"(?:stablehlo|chlo)\.([a-z_0-9]+)", c["mlir"]))
    assert len(U.CASES) >= 100 and len(U.CASES) - len(U.UNSUPPORTED) - len(U.BEYOND_F64_INTEGERS) >= 90
    for op in ("add", "dot_general", "reduce", "while", "case", "gather", "dynamic_slice", "dynamic_update_slice", "broadcast_in_dim", "transpose",
               "concatenate", "slice", "compare", "select", "convert", "iota", "sort", "cholesky", "triangular_solve", "custom_call", "erf_inv",
               "shift_right_logical", "remainder", "clamp", "reverse", "map"):
        assert op in ops, op


def test_parser_reads_wrapped_statements_multi_result_functions_and_regions():
    text = """
module @m {
  func.func private @two(%a: tensor<2xf64>,
      %b: tensor<f64>) -> (tensor<2xf64>, tensor<f64>) {
    %0 = stablehlo.broadcast_in_dim %b, dims = [] : (tensor<f64>) -> tensor<2xf64>
    %1 = stablehlo.multiply %a, %0 : tensor<2xf64>
    %c = stablehlo.constant dense<0.0> : tensor<f64>
    %2 = stablehlo.reduce(%1 init: %c) across dimensions = [0] : (tensor<2xf64>, tensor<f64>) -> tensor<f64>
     reducer(%x: tensor<f64>, %y: tensor<f64>)  {
      %s = stablehlo.add %x, %y : tensor<f64>
      stablehlo.return %s : tensor<f64>
    }
    return %1, %2 : tensor<2xf64>, tensor<f64>
  }
  func.func public @main(%arg0: tensor<2xf64>, %arg1: tensor<f64>) -> tensor<f64> {
    %r:2 = call @two(%arg0, %arg1) : (tensor<2xf64>, tensor<f64>) -> (tensor<2xf64>, tensor<f64>)
    %0 = stablehlo.dot_general %r#0, %arg0,
      contracting_dims = [0] x [0] :
      (tensor<2xf64>, tensor<2xf64>) -> tensor<f64>
    %1 = stablehlo.add %0, %r#1 : tensor<f64>
    return %1 : tensor<f64>
  }
}
"""
    funcs = sh.parse_module(text)
    assert [str(t) for t in funcs["two"].result_types] == ["tensor<2xf64>", "tensor<f64>"] and len(funcs["two"].args) == 2
    out = dsl_numpy.trace_eval(lambda xp, a, b: sh.trace(text, [a, b])[0].a[()], np.array([2.0, 3.0]), 4.0)
    assert out == (2 * 4) * 2 + (3 * 4) * 3 + (8 + 12)
    with pytest.raises(NotImplementedError, match="scatter"):
        sh.trace(text.replace("stablehlo.add %0, %r#1", "stablehlo.scatter %0, %r#1"), [dsl.Vec([dsl.leaf("a"), dsl.leaf("b")]), dsl.leaf("c")])
