#!/usr/bin/env python3
"""Known answers for WHOLE-WORLD (entity-batched) StableHLO ingestion from the reference's own tests of those fragments.

Build container only (/root/reference):   python tests/golden/make_stablehlo_world_golden.py

libs/cranelift-mlir/tests/{test_gather_3body, test_dynamic_ops_3body, test_while_dyn_slice, test_closed_call, test_threefry,
test_threefry_e2e, test_uniform_pipeline, test_sret_large}.rs hold the pieces a dumped world tick is made of (constant-index row gathers, the
`edge_fold` while with dynamic_slice / dynamic_update_slice by the loop counter, the transposes and broadcasts around it, jax.random's
threefry rounds and its bits -> uniform float construction), each as an inline MLIR module with inputs and asserted outputs.  Their
expectations are computed by Rust expressions, so each case is transcribed here BY HAND: the module text is read from the
reference's file (test DATA, like the golden CSVs), inputs and expected outputs are restated next to the Rust lines they come from.

Three tests of those files do not carry their module inline: they `include_str!` libs/cranelift-mlir/testdata/ball.stablehlo.mlir
and tests/closed_call_test.mlir, which are git-LFS POINTERS in the checkout (no content).  Their known answers are still the
reference's (closed_call: counter 0 -> 1, x = 0, y = 1 on zero inputs, test_closed_call.rs:83-96; threefry2x32(key (0, 0), counters
(0, 0..2)) = 0x6b200159.. / 0x99ba4efe.., test_threefry_e2e.rs:108-109; the ball's wind for seed 0, test_uniform_pipeline.rs:152-156);
the modules they run are RECONSTRUCTED below in the spelling jax's lowering gives `jax._src.prng._threefry2x32_lowering` (rolled
loop: a 5-trip while around @closed_call = four rounds + key injection) and `jax.random.normal` — every statement shape of them
appears inline somewhere in the same test files (test_one_threefry_round is one round verbatim).  Cases of that kind say
"reconstructed": true.  -> tests/golden/stablehlo_world_fragments.json"""
import json
import re
import struct
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))

REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
OUT = Path(__file__).resolve().parent / "stablehlo_world_fragments.json"
T = REF / "libs" / "cranelift-mlir" / "tests"


def inline_module(file: str, test: str) -> str:
    src = (T / file).read_text()
    body = src[src.index(f"fn {test}()"):]
    return re.search(r'let mlir = r#"(.*?)"#;', body, re.S).group(1)


cases = []


def case(name, source, mlir, inputs, expected, tol=0.0, **extra):
    """inputs: [(type, [values])]; expected: {output index: (type, [values])}."""
    cases.append(dict(name=name, source=source, mlir=mlir, inputs=[{"type": t, "values": list(v)} for t, v in inputs],
                      expected={str(k): {"type": t, "values": list(v)} for k, (t, v) in expected.items()}, tol=tol, **extra))


# ---- test_gather_3body.rs ------------------------------------------------------------------------------------------------------
f = "test_gather_3body.rs"
case("test_gather_1x1_index_from_constant", f + ":5-24", inline_module(f, "test_gather_1x1_index_from_constant"),
     [("f64", range(21))], {0: ("f64", range(7, 14))})                                   # :17-23 row 1 of the 3x7 table 0..21
case("test_gather_2x1_index_from_constant", f + ":27-48", inline_module(f, "test_gather_2x1_index_from_constant"),
     [("f64", range(21))], {0: ("f64", list(range(14, 21)) + list(range(7)))})          # :41-47 row 2, then row 0
case("test_three_body_inner_fragment", f + ":51-71", inline_module(f, "test_three_body_inner_fragment"),
     [("f64", range(1, 22))], {0: ("f64", range(15, 22))})                               # :64-70

# ---- test_dynamic_ops_3body.rs --------------------------------------------------------------------------------------------------
f = "test_dynamic_ops_3body.rs"
m = inline_module(f, "test_dynamic_slice_3body_pattern")
case("test_dynamic_slice_3body_pattern__index0", f + ":5-31", m, [("f64", range(1, 43)), ("i64", [0])], {0: ("f64", range(1, 22))})
case("test_dynamic_slice_3body_pattern__index1", f + ":33-38", m, [("f64", range(1, 43)), ("i64", [1])], {0: ("f64", range(22, 43))})
m = inline_module(f, "test_dynamic_update_slice_3body_pattern")
case("test_dynamic_update_slice_3body_pattern__index0", f + ":41-58", m, [("i64", [100, 200]), ("i64", [999]), ("i64", [0])], {0: ("i64", [999, 200])})
case("test_dynamic_update_slice_3body_pattern__index1", f + ":60-63", m, [("i64", [100, 200]), ("i64", [999]), ("i64", [1])], {0: ("i64", [100, 999])})
case("test_broadcast_in_dim_6_to_3x6", f + ":66-91", inline_module(f, "test_broadcast_in_dim_6_to_3x6"),
     [("f64", [1, 2, 3, 4, 5, 6])], {0: ("f64", [1, 2, 3, 4, 5, 6] * 3)}, tol=1e-10)
case("test_broadcast_in_dim_3x1_to_3x3", f + ":94-112", inline_module(f, "test_broadcast_in_dim_3x1_to_3x3"),
     [("f64", [10, 20, 30])], {0: ("f64", [10, 10, 10, 20, 20, 20, 30, 30, 30])})
case("test_transpose_3body_pattern", f + ":115-148", inline_module(f, "test_transpose_3body_pattern"),
     [("f64", range(42))], {0: ("f64", [i * 14 + j * 7 + k for j in range(2) for i in range(3) for k in range(7)])}, tol=1e-10)   # out[j][i][k] = in[i][j][k]

# ---- test_while_dyn_slice.rs ------------------------------------------------------------------------------------------------------
f = "test_while_dyn_slice.rs"
case("test_while_with_dynamic_slice_accumulate", f + ":5-37", inline_module(f, "test_while_with_dynamic_slice_accumulate"),
     [("f64", [1, 2, 3, 4, 5, 6])], {0: ("f64", [5, 7, 9])})

# ---- test_closed_call.rs / test_threefry.rs / test_uniform_pipeline.rs: the inline ones -----------------------------------------
f = "test_closed_call.rs"
x, y = [100, 200, 300], [0xDEAD, 0xBEEF, 0xCAFE]
rotl = lambda v, d: ((v << d) | (v >> (32 - d))) & 0xFFFFFFFF
xn = [(a + b) & 0xFFFFFFFF for a, b in zip(x, y)]                                       # :32-44 wrapping_add, rotate_left(13), xor
case("test_one_threefry_round", f + ":5-48", inline_module(f, "test_one_threefry_round"),
     [("u32", x), ("u32", y)], {0: ("u32", xn), 1: ("u32", [a ^ rotl(b, 13) for a, b in zip(xn, y)])})
f = "test_threefry.rs"
case("test_simple_shift_right_logical_ui32", f + ":44-63", inline_module(f, "test_simple_shift_right_logical_ui32"),
     [("u32", [0xDEADBEEF]), ("u32", [4])], {0: ("u32", [0xDEADBEEF >> 4])})
case("test_ui32_add_overflow", f + ":66-79", inline_module(f, "test_ui32_add_overflow"), [("u32", [0xFFFFFFFF]), ("u32", [1])], {0: ("u32", [0])})
case("test_i64_to_ui32_convert", f + ":82-95", inline_module(f, "test_i64_to_ui32_convert"), [("i64", [4294967295])], {0: ("u32", [0xFFFFFFFF])})
case("test_ui64_shift_right_logical", f + ":98-116", inline_module(f, "test_ui64_shift_right_logical"),
     [("u64", [0x0000000100000002]), ("u64", [32])], {0: ("u64", [1])})
case("test_bitcast_convert_ui64_to_f64", f + ":119-133", inline_module(f, "test_bitcast_convert_ui64_to_f64"),
     [("u64", [0x3FF0000000000000])], {0: ("f64", [1.0])}, tol=1e-15)
f = "test_uniform_pipeline.rs"
case("test_ui32_to_ui64_shift_left_32", f + ":7-30", inline_module(f, "test_ui32_to_ui64_shift_left_32"),
     [("u32", [0xDEADBEEF])], {0: ("u64", [0xDEADBEEF << 32])})
hi, lo = 0xABCD1234, 0x56789ABC                                                          # :60-75 the reference computes it in Rust
bits = ((((hi << 32) | lo) >> 12) | 0x3FF0000000000000)
case("test_uniform_float_construction", f + ":33-76", inline_module(f, "test_uniform_float_construction"),
     [("u32", [hi]), ("u32", [lo])], {0: ("f64", [struct.unpack("<d", struct.pack("<Q", bits))[0] - 1.0])}, tol=1e-15)

# ---- test_sret_large.rs: calls returning large / several tensors (how @inner and @closed_call return) -----------------------------------
f = "test_sret_large.rs"
a18 = list(range(1, 19))
case("test_sret_call_3x6_return", f + ":5-28", inline_module(f, "test_sret_call_3x6_return"),
     [("f64", a18), ("f64", [10 * v for v in a18])], {0: ("f64", [11 * v for v in a18])})                     # :20-27
case("test_sret_call_multi_return", f + ":31-53", inline_module(f, "test_sret_call_multi_return"),
     [("f64", a18)], {0: ("f64", [2 * v for v in a18]), 1: ("i64", [42])})                                   # :46-52

# ---- the three tests whose modules are LFS pointers: reconstructed in jax's spelling, pinned on the reference's asserted outputs ----
pointer = (T / "closed_call_test.mlir").read_text()
assert pointer.startswith("version https://git-lfs.github.com/spec/v1"), "closed_call_test.mlir has content now: read it instead"
assert (T.parent / "testdata" / "ball.stablehlo.mlir").read_text().startswith("version https://git-lfs"), "ball.stablehlo.mlir has content now"


from hlo_random_parts import THREEFRY, closed_call_fn  # noqa: E402  (the reconstructed functions: tests/golden/hlo_random_parts.py)

f = "test_closed_call.rs"
case("test_closed_call_standalone", f + ":51-96", "\nmodule @module {\n" + closed_call_fn(True) + "\n}\n",
     [("i64", [0]), ("u32", [0, 0, 0]), ("u32", [0, 0, 0]), ("u32", [0]), ("u32", [0]), ("u32", [0]), ("u32", [13, 15, 26, 6]), ("u32", [17, 29, 16, 24])],
     {0: ("i64", [1]), 1: ("u32", [0, 0, 0]), 2: ("u32", [1, 1, 1])}, reconstructed=True)              # :83-96
f = "test_threefry_e2e.rs"
main = """  func.func public @main(%arg0: tensor<ui32>, %arg1: tensor<ui32>, %arg2: tensor<3xui32>, %arg3: tensor<3xui32>) -> (tensor<3xui32>, tensor<3xui32>) {
    %0:2 = call @threefry2x32(%arg0, %arg1, %arg2, %arg3) : (tensor<ui32>, tensor<ui32>, tensor<3xui32>, tensor<3xui32>) -> (tensor<3xui32>, tensor<3xui32>)
    return %0#0, %0#1 : tensor<3xui32>, tensor<3xui32>
  }"""                                                                                    # the wrapper IS inline in the test (:36-41)
case("test_threefry2x32_with_known_inputs", f + ":9-123", "module @module {\n" + main + "\n" + THREEFRY + "\n" + closed_call_fn(False) + "\n}\n",
     [("u32", [0]), ("u32", [0]), ("u32", [0, 0, 0]), ("u32", [0, 1, 2])],
     {0: ("u32", [0x6b200159, 0x375f238f, 0xf71f4ea9]), 1: ("u32", [0x99ba4efe, 0xcddb151d, 0xa20e4081])}, reconstructed=True)   # :108-109
f = "test_uniform_pipeline.rs"
BALL_WIND = """  func.func public @main(%arg0: tensor<i64>) -> tensor<3xf64> {
    %c = stablehlo.constant dense<32> : tensor<i64>
    %0 = stablehlo.shift_right_logical %arg0, %c : tensor<i64>
    %1 = stablehlo.convert %0 : (tensor<i64>) -> tensor<ui32>
    %c_0 = stablehlo.constant dense<4294967295> : tensor<i64>
    %2 = stablehlo.and %arg0, %c_0 : tensor<i64>
    %3 = stablehlo.convert %2 : (tensor<i64>) -> tensor<ui32>
    %4 = stablehlo.iota dim = 0 : tensor<3xui64>
    %c_1 = stablehlo.constant dense<32> : tensor<ui64>
    %5 = stablehlo.broadcast_in_dim %c_1, dims = [] : (tensor<ui64>) -> tensor<3xui64>
    %6 = stablehlo.shift_right_logical %4, %5 : tensor<3xui64>
    %7 = stablehlo.convert %6 : (tensor<3xui64>) -> tensor<3xui32>
    %c_2 = stablehlo.constant dense<4294967295> : tensor<ui64>
    %8 = stablehlo.broadcast_in_dim %c_2, dims = [] : (tensor<ui64>) -> tensor<3xui64>
    %9 = stablehlo.and %4, %8 : tensor<3xui64>
    %10 = stablehlo.convert %9 : (tensor<3xui64>) -> tensor<3xui32>
    %11:2 = call @threefry2x32(%1, %3, %7, %10) : (tensor<ui32>, tensor<ui32>, tensor<3xui32>, tensor<3xui32>) -> (tensor<3xui32>, tensor<3xui32>)
    %12 = stablehlo.convert %11#0 : (tensor<3xui32>) -> tensor<3xui64>
    %13 = stablehlo.convert %11#1 : (tensor<3xui32>) -> tensor<3xui64>
    %c_3 = stablehlo.constant dense<32> : tensor<ui64>
    %14 = stablehlo.broadcast_in_dim %c_3, dims = [] : (tensor<ui64>) -> tensor<3xui64>
    %15 = stablehlo.shift_left %12, %14 : tensor<3xui64>
    %16 = stablehlo.or %15, %13 : tensor<3xui64>
    %c_4 = stablehlo.constant dense<12> : tensor<ui64>
    %17 = stablehlo.broadcast_in_dim %c_4, dims = [] : (tensor<ui64>) -> tensor<3xui64>
    %18 = stablehlo.shift_right_logical %16, %17 : tensor<3xui64>
    %c_5 = stablehlo.constant dense<4607182418800017408> : tensor<ui64>
    %19 = stablehlo.broadcast_in_dim %c_5, dims = [] : (tensor<ui64>) -> tensor<3xui64>
    %20 = stablehlo.or %18, %19 : tensor<3xui64>
    %21 = stablehlo.bitcast_convert %20 : (tensor<3xui64>) -> tensor<3xf64>
    %cst = stablehlo.constant dense<1.000000e+00> : tensor<f64>
    %22 = stablehlo.broadcast_in_dim %cst, dims = [] : (tensor<f64>) -> tensor<3xf64>
    %23 = stablehlo.subtract %21, %22 : tensor<3xf64>
    %cst_6 = stablehlo.constant dense<2.000000e+00> : tensor<f64>
    %24 = stablehlo.broadcast_in_dim %cst_6, dims = [] : (tensor<f64>) -> tensor<3xf64>
    %25 = stablehlo.multiply %23, %24 : tensor<3xf64>
    %cst_7 = stablehlo.constant dense<-0.99999999999999989> : tensor<f64>
    %26 = stablehlo.broadcast_in_dim %cst_7, dims = [] : (tensor<f64>) -> tensor<3xf64>
    %27 = stablehlo.add %25, %26 : tensor<3xf64>
    %28 = stablehlo.maximum %26, %27 : tensor<3xf64>
    %29 = chlo.erf_inv %28 : tensor<3xf64> -> tensor<3xf64>
    %cst_8 = stablehlo.constant dense<1.4142135623730951> : tensor<f64>
    %30 = stablehlo.broadcast_in_dim %cst_8, dims = [] : (tensor<f64>) -> tensor<3xf64>
    %31 = stablehlo.multiply %30, %29 : tensor<3xf64>
    return %31 : tensor<3xf64>
  }"""
# examples/ball/sim.py:91-92 `random.normal(random.key(s), shape=(3,))`: threefry_seed, 64 random bits per sample from the
# partitionable counters (0, i), jax.random._uniform's mantissa construction over (nextafter(-1, inf), 1), sqrt(2) erf_inv(u)
case("test_full_prng_pipeline_seed_zero__wind", f + ":79-176", "module @module {\n" + BALL_WIND + "\n" + THREEFRY + "\n" + closed_call_fn(False) + "\n}\n",
     [("i64", [0])], {0: ("f64", [-0.2058421394796434, -0.7847657764467411, 1.8160866726679836])}, tol=1e-12, reconstructed=True)   # :152-156

doc = {"source": "libs/cranelift-mlir/tests/{test_gather_3body,test_dynamic_ops_3body,test_while_dyn_slice,test_closed_call,test_threefry,"
                 "test_threefry_e2e,test_uniform_pipeline,test_sret_large}.rs (inline modules + asserted outputs; see make_stablehlo_world_golden.py)",
       "not_known_answers": {"test_threefry.rs::test_threefry_round": "parses the LFS-pointer ball module and checks that @closed_call exists: no output asserted",
                             "test_threefry.rs::test_inner_prng": "compiles the LFS-pointer ball module and checks that @inner / @main exist: no output asserted"},
       "cases": cases}
OUT.write_text(json.dumps(doc, indent=0))
print(len(cases), "cases ->", OUT, OUT.stat().st_size, "bytes")
