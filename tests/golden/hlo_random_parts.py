"""jax.random's functions as jax's lowering spells them — `threefry2x32` (rolled: a 5-trip while around @closed_call = four rounds + key
injection; jax._src.prng._threefry2x32_lowering) — RECONSTRUCTED, because the reference's copies live in git-LFS pointers
(libs/cranelift-mlir/testdata/ball.stablehlo.mlir, tests/closed_call_test.mlir).  Pinned on the reference's asserted outputs by
tests/golden/make_stablehlo_world_golden.py (test_closed_call.rs:83-96, test_threefry_e2e.rs:108-109, test_uniform_pipeline.rs:152-156);
every statement shape appears inline in the same test files.  Shared by that generator and tests/golden/hlo_world_builder.ball_world.
TEST INFRASTRUCTURE (data)."""


def closed_call_fn(public_main: bool) -> str:
    """jax._src.prng._threefry2x32_lowering's rolled_loop_step as fori_loop's scan body: (i, x0, x1, ks0, ks1, ks2, rot0, rot1) ->
    (i + 1, x0', x1', ks1, ks2, ks0, rot1, rot0); four apply_round with the entries of rot0, then the key injection."""
    sig_in = ("%arg0: tensor<i64>, %arg1: tensor<3xui32>, %arg2: tensor<3xui32>, %arg3: tensor<ui32>, %arg4: tensor<ui32>, "
              "%arg5: tensor<ui32>, %arg6: tensor<4xui32>, %arg7: tensor<4xui32>")
    tys = "tensor<i64>, tensor<3xui32>, tensor<3xui32>, tensor<ui32>, tensor<ui32>, tensor<ui32>, tensor<4xui32>, tensor<4xui32>"
    L = [f"  func.func {'public @main' if public_main else 'private @closed_call'}({sig_in}) -> ({tys}) {{"]
    n = [0]
    def v():
        n[0] += 1
        return f"%{n[0] - 1}"
    x0, x1 = "%arg1", "%arg2"
    for r in range(4):
        sl, rot, s, b, shl, c32name, sub, b2, shr, o, xr = v(), v(), v(), v(), v(), f"%c_{r}" if r else "%c", v(), v(), v(), v(), v()
        L += [f"    {sl} = stablehlo.slice %arg6 [{r}:{r + 1}] : (tensor<4xui32>) -> tensor<1xui32>",
              f"    {rot} = stablehlo.reshape {sl} : (tensor<1xui32>) -> tensor<ui32>",
              f"    {s} = stablehlo.add {x0}, {x1} : tensor<3xui32>",
              f"    {b} = stablehlo.broadcast_in_dim {rot}, dims = [] : (tensor<ui32>) -> tensor<3xui32>",
              f"    {shl} = stablehlo.shift_left {x1}, {b} : tensor<3xui32>",
              f"    {c32name} = stablehlo.constant dense<32> : tensor<ui32>",
              f"    {sub} = stablehlo.subtract {c32name}, {rot} : tensor<ui32>",
              f"    {b2} = stablehlo.broadcast_in_dim {sub}, dims = [] : (tensor<ui32>) -> tensor<3xui32>",
              f"    {shr} = stablehlo.shift_right_logical {x1}, {b2} : tensor<3xui32>",
              f"    {o} = stablehlo.or {shl}, {shr} : tensor<3xui32>",
              f"    {xr} = stablehlo.xor {s}, {o} : tensor<3xui32>"]
        x0, x1 = s, xr
    k0b, nx0, k1b, t1, one, ip1, cv, cb, nx1 = v(), v(), v(), v(), "%c_4", v(), v(), v(), v()
    L += [f"    {k0b} = stablehlo.broadcast_in_dim %arg3, dims = [] : (tensor<ui32>) -> tensor<3xui32>",
          f"    {nx0} = stablehlo.add {x0}, {k0b} : tensor<3xui32>",
          f"    {k1b} = stablehlo.broadcast_in_dim %arg4, dims = [] : (tensor<ui32>) -> tensor<3xui32>",
          f"    {t1} = stablehlo.add {x1}, {k1b} : tensor<3xui32>",
          f"    {one} = stablehlo.constant dense<1> : tensor<i64>",
          f"    {ip1} = stablehlo.add %arg0, {one} : tensor<i64>",
          f"    {cv} = stablehlo.convert {ip1} : (tensor<i64>) -> tensor<ui32>",
          f"    {cb} = stablehlo.broadcast_in_dim {cv}, dims = [] : (tensor<ui32>) -> tensor<3xui32>",
          f"    {nx1} = stablehlo.add {t1}, {cb} : tensor<3xui32>",
          f"    return {ip1}, {nx0}, {nx1}, %arg4, %arg5, %arg3, %arg7, %arg6 : {tys}",
          "  }"]
    return "\n".join(L)


THREEFRY = """  func.func private @threefry2x32(%arg0: tensor<ui32>, %arg1: tensor<ui32>, %arg2: tensor<3xui32>, %arg3: tensor<3xui32>) -> (tensor<3xui32>, tensor<3xui32>) {
    %c = stablehlo.constant dense<[13, 15, 26, 6]> : tensor<4xui32>
    %c_0 = stablehlo.constant dense<[17, 29, 16, 24]> : tensor<4xui32>
    %0 = stablehlo.xor %arg0, %arg1 : tensor<ui32>
    %c_1 = stablehlo.constant dense<466688986> : tensor<ui32>
    %1 = stablehlo.xor %0, %c_1 : tensor<ui32>
    %2 = stablehlo.broadcast_in_dim %arg0, dims = [] : (tensor<ui32>) -> tensor<3xui32>
    %3 = stablehlo.add %arg2, %2 : tensor<3xui32>
    %4 = stablehlo.broadcast_in_dim %arg1, dims = [] : (tensor<ui32>) -> tensor<3xui32>
    %5 = stablehlo.add %arg3, %4 : tensor<3xui32>
    %c_2 = stablehlo.constant dense<0> : tensor<i64>
    %c_3 = stablehlo.constant dense<0> : tensor<i64>
    %6:9 = stablehlo.while(%iterArg = %c_3, %iterArg_4 = %c_2, %iterArg_5 = %3, %iterArg_6 = %5, %iterArg_7 = %arg1, %iterArg_8 = %1, %iterArg_9 = %arg0, %iterArg_10 = %c, %iterArg_11 = %c_0) : tensor<i64>, tensor<i64>, tensor<3xui32>, tensor<3xui32>, tensor<ui32>, tensor<ui32>, tensor<ui32>, tensor<4xui32>, tensor<4xui32>
     cond {
      %c_12 = stablehlo.constant dense<5> : tensor<i64>
      %7 = stablehlo.compare  LT, %iterArg, %c_12,  SIGNED : (tensor<i64>, tensor<i64>) -> tensor<i1>
      stablehlo.return %7 : tensor<i1>
    } do {
      %7:8 = func.call @closed_call(%iterArg_4, %iterArg_5, %iterArg_6, %iterArg_7, %iterArg_8, %iterArg_9, %iterArg_10, %iterArg_11) : (tensor<i64>, tensor<3xui32>, tensor<3xui32>, tensor<ui32>, tensor<ui32>, tensor<ui32>, tensor<4xui32>, tensor<4xui32>) -> (tensor<i64>, tensor<3xui32>, tensor<3xui32>, tensor<ui32>, tensor<ui32>, tensor<ui32>, tensor<4xui32>, tensor<4xui32>)
      %c_12 = stablehlo.constant dense<1> : tensor<i64>
      %8 = stablehlo.add %iterArg, %c_12 : tensor<i64>
      stablehlo.return %8, %7#0, %7#1, %7#2, %7#3, %7#4, %7#5, %7#6, %7#7 : tensor<i64>, tensor<i64>, tensor<3xui32>, tensor<3xui32>, tensor<ui32>, tensor<ui32>, tensor<ui32>, tensor<4xui32>, tensor<4xui32>
    }
    return %6#2, %6#3 : tensor<3xui32>, tensor<3xui32>
  }"""

