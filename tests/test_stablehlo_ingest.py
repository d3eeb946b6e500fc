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


@pytest.mark.parametrize("case", U.WORLD_CASES, ids=[c["name"] for c in U.WORLD_CASES])
def test_reference_world_fragment_known_answers_on_the_cpu_walker(case):
    """The entity-batched pieces of a dumped world tick and jax.random's integer pipeline (u32 wrap-around, ui64 as two words,
    bitcast to f64), libs/cranelift-mlir/tests/test_{gather_3body,dynamic_ops_3body,while_dyn_slice,closed_call,threefry,
    threefry_e2e,uniform_pipeline}.rs.  Integer results are compared EXACTLY."""
    system, values, expect = U.build(case)
    comps = walk(system, values, expect)
    for nm, (w, exp) in expect.items():
        integer = case["expected"].get(nm[len("out"):], {"type": "f64"})["type"] != "f64"
        U.check(case["name"], comps[nm][0], w, exp, 0.0 if integer else max(case["tol"], 1e-15))
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
               "shift_right_logical", "remainder", "clamp", "reverse", "map", "scatter", "real_dynamic_slice", "reduce_window", "select_and_scatter"):
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
    with pytest.raises(NotImplementedError, match="convolution"):          # an op the front end does not read says so by name
        sh.trace(text.replace("stablehlo.add %0, %r#1", "stablehlo.convolution %0, %r#1"), [dsl.Vec([dsl.leaf("a"), dsl.leaf("b")]), dsl.leaf("c")])


def test_scatter_with_a_traced_row_index_windowed_updates_and_an_add_region():
    """What `x.at[i].add(row)` lowers to, with the row index an INPUT (the reference's own scatter tests use constants): a 3-wide
    window added into row i of a 4 x 3 operand; an index that does not fit is skipped (scatter does not clamp); negative too."""
    text = """
module @m {
  func.func public @main(%arg0: tensor<4x3xf64>, %arg1: tensor<1xi32>, %arg2: tensor<3xf64>) -> tensor<4x3xf64> {
    %0 = "stablehlo.scatter"(%arg0, %arg1, %arg2) <{
      indices_are_sorted = true,
      scatter_dimension_numbers = #stablehlo.scatter<update_window_dims = [0], inserted_window_dims = [0], scatter_dims_to_operand_dims = [0]>,
      unique_indices = true
    }> ({
    ^bb0(%a: tensor<f64>, %b: tensor<f64>):
      %s = stablehlo.add %a, %b : tensor<f64>
      stablehlo.return %s : tensor<f64>
    }) : (tensor<4x3xf64>, tensor<1xi32>, tensor<3xf64>) -> tensor<4x3xf64>
    return %0 : tensor<4x3xf64>
  }
}
"""
    x = np.arange(12.0).reshape(4, 3)
    row = np.array([100.0, 200.0, 300.0])
    for i in (0, 2, 3, 4, -1):
        got = dsl_numpy.trace_eval(lambda xp, a, k, r: dsl.Vec(list(sh.trace(text, [a, k, r])[0].a.reshape(-1))), x.reshape(-1), np.array([float(i)]), row)
        want = x.copy()
        if 0 <= i < 4:
            want[i] += row
        assert np.array_equal(np.asarray(got).reshape(4, 3), want), i


def test_reduce_window_with_padding_strides_and_a_max_region_and_select_and_scatter_in_two_dimensions():
    """A 2 x 2 / stride 2 max pool over a padded 3 x 3 input (padding reads the init value), and the select_and_scatter that
    routes one value per window back to the window's maximum — against numpy."""
    pool = """
module @m {
  func.func public @main(%arg0: tensor<3x3xf64>, %arg1: tensor<f64>) -> tensor<2x2xf64> {
    %0 = "stablehlo.reduce_window"(%arg0, %arg1) ({
    ^bb0(%a: tensor<f64>, %b: tensor<f64>):
      %1 = stablehlo.maximum %a, %b : tensor<f64>
      stablehlo.return %1 : tensor<f64>
    }) {window_dimensions = array<i64: 2, 2>, window_strides = array<i64: 2, 2>, padding = dense<[[0, 1], [0, 1]]> : tensor<2x2xi64>} : (tensor<3x3xf64>, tensor<f64>) -> tensor<2x2xf64>
    return %0 : tensor<2x2xf64>
  }
}
"""
    x = np.array([[1.0, 7.0, 2.0], [3.0, 4.0, 9.0], [8.0, 5.0, 6.0]])
    got = dsl_numpy.trace_eval(lambda xp, a, i: dsl.Vec(list(sh.trace(pool, [a, i])[0].a.reshape(-1))), x.reshape(-1), -1e9)
    padded = np.full((4, 4), -1e9)
    padded[:3, :3] = x
    want = np.array([[padded[2 * i:2 * i + 2, 2 * j:2 * j + 2].max() for j in range(2)] for i in range(2)])
    assert np.array_equal(np.asarray(got).reshape(2, 2), want)
    route = """
module @m {
  func.func public @main(%arg0: tensor<4x4xf64>, %arg1: tensor<2x2xf64>, %arg2: tensor<f64>) -> tensor<4x4xf64> {
    %0 = "stablehlo.select_and_scatter"(%arg0, %arg1, %arg2) ({
    ^bb0(%a: tensor<f64>, %b: tensor<f64>):
      %1 = stablehlo.compare GE, %a, %b, FLOAT : (tensor<f64>, tensor<f64>) -> tensor<i1>
      stablehlo.return %1 : tensor<i1>
    }, {
    ^bb0(%x: tensor<f64>, %y: tensor<f64>):
      %1 = stablehlo.add %x, %y : tensor<f64>
      stablehlo.return %1 : tensor<f64>
    }) {window_dimensions = dense<[2, 2]> : tensor<2xi64>, window_strides = dense<[2, 2]> : tensor<2xi64>, padding = dense<0> : tensor<2x2xi64>} : (tensor<4x4xf64>, tensor<2x2xf64>, tensor<f64>) -> tensor<4x4xf64>
    return %0 : tensor<4x4xf64>
  }
}
"""
    rng = np.random.default_rng(5)
    op, src = rng.permutation(16).astype(float).reshape(4, 4), np.array([[10.0, 20.0], [30.0, 40.0]])
    got = dsl_numpy.trace_eval(lambda xp, a, b, i: dsl.Vec(list(sh.trace(route, [a, b, i])[0].a.reshape(-1))), op.reshape(-1), src.reshape(-1), 0.5)
    want = np.full((4, 4), 0.5)
    for i in range(2):
        for j in range(2):
            blk = op[2 * i:2 * i + 2, 2 * j:2 * j + 2]
            r, c = np.unravel_index(np.argmax(blk), (2, 2))
            want[2 * i + r, 2 * j + c] += src[i, j]
    assert np.array_equal(np.asarray(got).reshape(4, 4), want)


def test_dgetrf_keeps_lapacks_pivots_and_row_order():
    """lapack_dgetrf_ffi on a 4 x 4 matrix that needs a swap in every column: LU and ipiv equal scipy.linalg.lu_factor (LAPACK's
    own dgetrf) — the factor in LAPACK's row order, not merely a valid factorisation."""
    from scipy.linalg import lu_factor
    text = """
module @m {
  func.func public @main(%arg0: tensor<4x4xf64>) -> (tensor<4x4xf64>, tensor<4xi32>, tensor<i32>) {
    %0:3 = stablehlo.custom_call @lapack_dgetrf_ffi(%arg0) {backend_config = "", mhlo.backend_config = {}} -> (tensor<4x4xf64>, tensor<4xi32>, tensor<i32>)
    return %0#0, %0#1, %0#2 : tensor<4x4xf64>, tensor<4xi32>, tensor<i32>
  }
}
"""
    a = np.array([[1.0, 2.0, 3.0, 4.0], [2.0, -1.0, 0.5, 7.0], [9.0, 1.0, 2.0, -3.0], [4.0, 8.0, -6.0, 1.0]])
    out = dsl_numpy.trace_eval(lambda xp, m: dsl.Vec([v for r in sh.trace(text, [m]) for v in r.a.reshape(-1)]), a.reshape(-1))
    out = np.asarray(out)
    lu, piv = lu_factor(a)
    assert np.allclose(out[:16].reshape(4, 4), lu, rtol=0, atol=1e-13) and np.array_equal(out[16:20], piv + 1.0) and out[20] == 0.0


def test_qr_and_eigh_custom_calls_against_lapack_itself():
    """What jnp.linalg.qr (dgeqrf + dorgqr) and jnp.linalg.eigh (dsyevd) lower to on CPU.  The reference's op tests hold no
    answers for these three: dgeqrf's packed output and tau are compared with scipy's binding of LAPACK's own routine
    (same reflectors, not merely a valid factorisation), Q R = A, and the eigen-decomposition by its defining properties."""
    from scipy.linalg import lapack
    qr_text = """
module @m {
  func.func public @main(%arg0: tensor<4x3xf64>) -> (tensor<4x3xf64>, tensor<3xf64>, tensor<4x3xf64>) {
    %0:2 = stablehlo.custom_call @lapack_dgeqrf_ffi(%arg0) {mhlo.backend_config = {}} -> (tensor<4x3xf64>, tensor<3xf64>)
    %1 = stablehlo.custom_call @lapack_dorgqr_ffi(%0#0, %0#1) {mhlo.backend_config = {}} -> (tensor<4x3xf64>)
    return %0#0, %0#1, %1 : tensor<4x3xf64>, tensor<3xf64>, tensor<4x3xf64>
  }
}
"""
    a = np.array([[2.0, -1.0, 0.5], [1.0, 3.0, -2.0], [0.0, 4.0, 1.0], [-3.0, 0.5, 2.5]])
    out = np.asarray(dsl_numpy.trace_eval(lambda xp, m: dsl.Vec([v for r in sh.trace(qr_text, [m]) for v in r.a.reshape(-1)]), a.reshape(-1)))
    packed, tau, q = out[:12].reshape(4, 3), out[12:15], out[15:].reshape(4, 3)
    want_packed, want_tau, _, info = lapack.dgeqrf(a)
    assert info == 0 and np.allclose(packed, want_packed, rtol=0, atol=1e-13) and np.allclose(tau, want_tau, rtol=0, atol=1e-13)
    assert np.allclose(q @ np.triu(packed)[:3], a, atol=1e-13) and np.allclose(q.T @ q, np.eye(3), atol=1e-13)
    eigh_text = """
module @m {
  func.func public @main(%arg0: tensor<3x3xf64>) -> (tensor<3x3xf64>, tensor<3xf64>, tensor<i32>) {
    %0:3 = stablehlo.custom_call @lapack_dsyevd_ffi(%arg0) {mhlo.backend_config = {mode = 86 : ui8, uplo = 76 : ui8}} -> (tensor<3x3xf64>, tensor<3xf64>, tensor<i32>)
    return %0#0, %0#1, %0#2 : tensor<3x3xf64>, tensor<3xf64>, tensor<i32>
  }
}
"""
    s_ = np.array([[4.0, 99.0, 99.0], [1.0, 3.0, 99.0], [-2.0, 0.5, 5.0]])          # uplo = L: the upper triangle is not read
    out = np.asarray(dsl_numpy.trace_eval(lambda xp, m: dsl.Vec([v for r in sh.trace(eigh_text, [m]) for v in r.a.reshape(-1)]), s_.reshape(-1)))
    vecs, vals = out[:9].reshape(3, 3), out[9:12]
    full = np.tril(s_) + np.tril(s_, -1).T
    assert np.allclose(vals, np.linalg.eigvalsh(full), atol=1e-12) and np.all(np.diff(vals) > 0) and out[12] == 0.0
    assert np.allclose(full @ vecs, vecs * vals, atol=1e-12) and np.allclose(vecs.T @ vecs, np.eye(3), atol=1e-12)


def test_pad_edge_interior_and_negative_padding_against_numpy():
    """stablehlo.pad (the reference's parser reads it, libs/cranelift-mlir/src/parser.rs; its op tests hold no case): jnp.pad's
    edge padding, lax.pad's interior padding and a negative (cropping) edge, against numpy."""
    text = """
module @m {
  func.func public @main(%arg0: tensor<2x3xf64>, %arg1: tensor<f64>) -> tensor<4x6xf64> {
    %0 = stablehlo.pad %arg0, %arg1, low = [1, -1], high = [1, 2], interior = [0, 1] : (tensor<2x3xf64>, tensor<f64>) -> tensor<4x6xf64>
    return %0 : tensor<4x6xf64>
  }
}
"""
    x = np.arange(1.0, 7.0).reshape(2, 3)
    got = dsl_numpy.trace_eval(lambda xp, a, v: dsl.Vec(list(sh.trace(text, [a, v])[0].a.reshape(-1))), x.reshape(-1), -7.0)
    want = np.full((2, 5), -7.0)
    want[:, ::2] = x                                                   # interior padding: x0 . x1 . x2
    want = np.pad(want[:, 1:], ((1, 1), (0, 2)), constant_values=-7.0)  # low -1 crops the first column, high 2 appends two
    assert np.array_equal(np.asarray(got).reshape(4, 6), want)


def _conv_reference(x, k, stride, pad, ldil, rdil):
    """stablehlo.convolution for [b, 0, 1, f] x [0, 1, i, o] -> [b, 0, 1, f] with plain loops (the spec's definition)."""
    B, H, W, C = x.shape
    xd = np.zeros((B, (H - 1) * ldil[0] + 1, (W - 1) * ldil[1] + 1, C))
    xd[:, ::ldil[0], ::ldil[1]] = x
    xp = np.pad(xd, ((0, 0), pad[0], pad[1], (0, 0)))
    KH, KW, _, O = k.shape
    eh, ew = (KH - 1) * rdil[0] + 1, (KW - 1) * rdil[1] + 1
    oh, ow = (xp.shape[1] - eh) // stride[0] + 1, (xp.shape[2] - ew) // stride[1] + 1
    out = np.zeros((B, oh, ow, O))
    for b in range(B):
        for i in range(oh):
            for j in range(ow):
                for o in range(O):
                    acc = 0.0
                    for a in range(KH):
                        for c in range(KW):
                            for f in range(C):
                                acc += xp[b, i * stride[0] + a * rdil[0], j * stride[1] + c * rdil[1], f] * k[a, c, f, o]
                    out[b, i, j, o] = acc
    return out


def test_convolution_the_references_own_ignored_case_and_a_strided_padded_dilated_one():
    """stablehlo.convolution (libs/cranelift-mlir/ARCHITECTURE.md:910 lists it as supported; the reference's one known answer for it,
    ops.rs:4211-4230, is #[ignore]d there: "runtime indexing needs further debugging").  That case — [1, 2, 3, 4] with the kernel
    [1, 1] -> [3, 5, 7] — and a 2-D one with strides, asymmetric padding, lhs and rhs dilation and a channel-last layout against
    plain loops over the spec's definition."""
    text = """
module @module {
  func.func public @main(%arg0: tensor<1x4x1xf64>, %arg1: tensor<2x1x1xf64>) -> tensor<1x3x1xf64> {
    %0 = "stablehlo.convolution"(%arg0, %arg1) {window_strides = array<i64: 1>, padding = dense<[[0, 0]]> : tensor<1x2xi64>, lhs_dilation = array<i64: 1>, rhs_dilation = array<i64: 1>, dimension_numbers = #stablehlo.conv<[b, 0, f]x[0, i, o]->[b, 0, f]>, batch_group_count = 1 : i64, feature_group_count = 1 : i64} : (tensor<1x4x1xf64>, tensor<2x1x1xf64>) -> tensor<1x3x1xf64>
    return %0 : tensor<1x3x1xf64>
  }
}
"""
    got = dsl_numpy.trace_eval(lambda xp, a, k: dsl.Vec(list(sh.trace(text, [a, k])[0].a.reshape(-1))), np.array([1.0, 2.0, 3.0, 4.0]), np.array([1.0, 1.0]))
    assert np.array_equal(np.asarray(got), [3.0, 5.0, 7.0])
    rng = np.random.default_rng(12)
    x, k = rng.normal(size=(2, 4, 5, 3)), rng.normal(size=(2, 3, 3, 2))
    stride, pad, ldil, rdil = (2, 1), ((1, 0), (2, 1)), (1, 2), (2, 1)
    want = _conv_reference(x, k, stride, pad, ldil, rdil)
    text2 = f"""
module @module {{
  func.func public @main(%arg0: tensor<2x4x5x3xf64>, %arg1: tensor<2x3x3x2xf64>) -> tensor<{'x'.join(str(d) for d in want.shape)}xf64> {{
    %0 = stablehlo.convolution(%arg0, %arg1) dim_numbers = [b, 0, 1, f]x[0, 1, i, o]->[b, 0, 1, f], window = {{stride = [2, 1], pad = [[1, 0], [2, 1]], lhs_dilate = [1, 2], rhs_dilate = [2, 1]}} {{batch_group_count = 1 : i64, feature_group_count = 1 : i64}} : (tensor<2x4x5x3xf64>, tensor<2x3x3x2xf64>) -> tensor<{'x'.join(str(d) for d in want.shape)}xf64>
    return %0 : tensor<{'x'.join(str(d) for d in want.shape)}xf64>
  }}
}}
"""
    got = dsl_numpy.trace_eval(lambda xp, a, kk: dsl.Vec(list(sh.trace(text2, [a, kk])[0].a.reshape(-1))), x.reshape(-1), k.reshape(-1))
    assert np.allclose(np.asarray(got).reshape(want.shape), want, rtol=1e-13, atol=1e-13)
    with pytest.raises(NotImplementedError, match="batch_group_count"):
        sh.trace(text.replace("batch_group_count = 1", "batch_group_count = 2"), [dsl.Vec([dsl.leaf(f"a{i}") for i in range(4)]), dsl.Vec([dsl.leaf("k0"), dsl.leaf("k1")])])
    # a depthwise convolution (feature_group_count = the number of channels): every channel with its own 3-tap kernel
    xd, kd = rng.normal(size=(1, 6, 2)), rng.normal(size=(3, 1, 2))
    text3 = """
module @module {
  func.func public @main(%arg0: tensor<1x6x2xf64>, %arg1: tensor<3x1x2xf64>) -> tensor<1x4x2xf64> {
    %0 = stablehlo.convolution(%arg0, %arg1) dim_numbers = [b, 0, f]x[0, i, o]->[b, 0, f], window = {} {batch_group_count = 1 : i64, feature_group_count = 2 : i64} : (tensor<1x6x2xf64>, tensor<3x1x2xf64>) -> tensor<1x4x2xf64>
    return %0 : tensor<1x4x2xf64>
  }
}
"""
    got = np.asarray(dsl_numpy.trace_eval(lambda xp, a, kk: dsl.Vec(list(sh.trace(text3, [a, kk])[0].a.reshape(-1))), xd.reshape(-1), kd.reshape(-1))).reshape(1, 4, 2)
    want = np.stack([np.correlate(xd[0, :, c], kd[:, 0, c], mode="valid") for c in range(2)], axis=1)[None]
    assert np.allclose(got, want, rtol=1e-13, atol=1e-13)


def test_rng_is_the_references_deterministic_fill():
    """stablehlo.rng in the reference is not a generator but a deterministic fill (libs/cranelift-mlir/src/tensor_rt.rs:2103-2140:
    UNIFORM = n values linearly spaced from a to b inclusive, NORMAL = the span's midpoints (i + 0.5) / n; ARCHITECTURE.md:916).  A
    module that carries the op gets the same values here; ops.rs:4414-4431 (`test_rng_uniform_mem`) asserts the range only."""
    def module(dist, n):
        return f"""
module @module {{
  func.func public @main(%arg0: tensor<f64>, %arg1: tensor<f64>) -> tensor<{n}xf64> {{
    %0 = "stablehlo.rng"(%arg0, %arg1) {{rng_distribution = #stablehlo<rng_distribution {dist}>}} : (tensor<f64>, tensor<f64>) -> tensor<{n}xf64>
    return %0 : tensor<{n}xf64>
  }}
}}
"""
    run = lambda text, a, b: np.asarray(dsl_numpy.trace_eval(lambda xp, x, y: dsl.Vec(list(sh.trace(text, [x, y])[0].a.reshape(-1))), a, b))
    got = run(module("UNIFORM", 4), 0.0, 1.0)
    assert np.array_equal(got, [0.0, 1.0 / 3.0, 2.0 / 3.0, 1.0]) and np.all((got >= 0.0) & (got <= 1.0))        # the reference's own assertion
    assert np.array_equal(run(module("UNIFORM", 5), -2.0, 6.0), [-2.0 + (i / 4.0) * 8.0 for i in range(5)])
    assert np.array_equal(run(module("UNIFORM", 1), 3.0, 5.0), [4.0])
    assert np.array_equal(run(module("NORMAL", 4), 0.0, 2.0), [0.0 + ((i + 0.5) / 4.0) * 2.0 for i in range(4)])


def test_relaxed_arithmetic_keeps_integer_semantics_exact_and_the_known_answers_inside_tolerance():
    """dsl.relaxed_arithmetic (world_system(arith="relaxed")) turns a / b into a * (1 / b) — 49 * (1 / 49) = 0.9999999999999999, whose
    truncation is 0.  Integer tensors are integral floats whose division, remainder, wrap-around and shifts are spelled with float
    divisions: the evaluator suspends the relaxation inside every integer op (stablehlo._Eval._binary).  (a) i64 quotients and
    remainders of pairs that the reciprocal form gets wrong stay exact; (b) every known answer of the reference's op tests and world
    fragments holds under relaxed arithmetic too — integer results exactly, floats to 1e-9."""
    pairs = [(a * b, b) for a in (1, 2, 3, 7, 100) for b in (49, 98, 103, 107, 161, 187, 196, 197, 206, 214)]
    assert any(np.trunc(x * (1.0 / y)) != x // y for x, y in pairs)                       # the hazard is real
    n = len(pairs)
    text = f"""
module @m {{
  func.func public @main(%arg0: tensor<{n}xi64>, %arg1: tensor<{n}xi64>, %arg2: tensor<{n}xf64>, %arg3: tensor<{n}xf64>) -> (tensor<{n}xi64>, tensor<{n}xi64>, tensor<{n}xf64>) {{
    %0 = stablehlo.divide %arg0, %arg1 : tensor<{n}xi64>
    %1 = stablehlo.remainder %arg0, %arg1 : tensor<{n}xi64>
    %2 = stablehlo.divide %arg2, %arg3 : tensor<{n}xf64>
    return %0, %1, %2 : tensor<{n}xi64>, tensor<{n}xi64>, tensor<{n}xf64>
  }}
}}"""
    system = sh.system(text, ["x", "y", "fx", "fy"], ["q", "r", "fq"])
    xs, ys = np.array([p[0] for p in pairs], dtype=np.float64), np.array([p[1] for p in pairs], dtype=np.float64)
    values = {"x": xs, "y": ys, "fx": xs, "fy": ys}
    expect = {"q": (n, {}), "r": (n, {}), "fq": (n, {})}
    with dsl.relaxed_arithmetic():
        comps = walk(system, values, expect)
    assert np.array_equal(comps["q"][0], xs // ys) and np.array_equal(comps["r"][0], np.zeros(n))
    assert np.max(np.abs(comps["fq"][0] - xs / ys) / (xs / ys)) < 4e-16 and not np.array_equal(comps["fq"][0], xs / ys)      # floats ARE relaxed
    for group, cases in (("ops", U.CASES), ("world", U.WORLD_CASES)):
        for case in cases:
            if case["name"] in U.UNSUPPORTED or case["name"] in U.BEYOND_F64_INTEGERS:
                continue
            system, values, expect = U.build(case)
            with dsl.relaxed_arithmetic():
                comps = walk(system, values, expect)
            for nm, (w, exp) in expect.items():
                integer = group == "world" and case["expected"].get(nm[len("out"):], {"type": "f64"})["type"] != "f64"
                U.check(case["name"], comps[nm][0], w, exp, 0.0 if integer else 1e-9)
