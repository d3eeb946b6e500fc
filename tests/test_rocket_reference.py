"""The reference's rocket example end to end against its own CI baseline (scripts/ci/baseline/rocket-csv, 100 ticks of all 24
columns + the 480 x 3 sample window): thirteen systems in front of six_dof(RK4) with gravity | thrust | aero effectors, a
PID loop closed through a Butterworth filter over the window, aerodynamic TORQUE on a non-uniform inertia diagonal — the
closed loop is free-running from the spawn state, nothing is teacher-forced.  Here: the traced program on the CPU walker
(tests/dsl_numpy.py — the DAG codegen.py turns into kernel code, integrated by the numpy restatement of RK4), i.e. the pin
of the program text in tests/rocket_dsl.py and of the tracer's window / scan / accel-in-front-of-six_dof semantics.
tests/test_gpu_rocket.py runs the same program through the generated gfx950 kernel."""
import numpy as np

from elodin_amd import _lib as L
from tests import dsl_numpy, rocket_dsl as R, rocket_util as U


def test_rocket_program_reproduces_the_reference_baseline_on_the_cpu_walker():
    tp = R.program().trace()
    assert tp.windows == {"v_rel_accel_buffer": (tp.windows["v_rel_accel_buffer"][0], 480, 3)} and tp.pre_reads_accel
    assert len(tp.columns) == 21      # 19 register columns + the window + its head
    pos, vel, inertia, comps = R.spawn(1)
    comps["v_rel_accel_buffer#head"] = np.zeros((1, 1))
    acc, worst = np.zeros((1, 6)), {}
    assert U.GOLDEN["tick"] == list(range(101))
    for tick in range(1, 101):
        F = dsl_numpy.program_tick(tp, pos, vel, acc, inertia, comps, tick, U.GOLDEN["simulation_time_step"], L.RK4)
        got = {k: v[0] for k, v in comps.items()}
        got.update(world_pos=pos[0], world_vel=vel[0], world_accel=acc[0], force=F[0], inertia=inertia[0])
        U.check_row(tick, got, worst)
    print("rocket vs reference baseline, CPU walker:", {k: f"{v:.1e}" for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:6]})
    U.assert_all_columns(worst)
    window = dsl_numpy.window_rows(comps, "v_rel_accel_buffer", 480, 3)[0].ravel()
    want = np.array(U.GOLDEN["v_rel_accel_buffer_final"])
    assert np.abs(window - want).max() < 1e-9 * np.abs(want).max()
    assert U.GOLDEN["v_rel_accel_buffer_nonzero_rows_per_tick"][-1] >= 99      # the window really filled up


def test_window_component_api_guards():
    import pytest
    from elodin_amd import dsl

    @dsl.system(buf=dsl.window(4, 2), x=2)        # (a bare (4, 2) would be a register matrix: <= 36 values, dsl._is_matrix_shape)
    def whole(buf, x):
        return {"buf": buf}
    with pytest.raises(TypeError, match="push"):
        dsl.Program([whole], dsl.Pipe([]), []).trace()

    @dsl.system(buf=dsl.window(4, 2), x=3)
    def wrong_width(buf, x):
        return {"buf": buf.push(x)}
    with pytest.raises(ValueError, match="row has 2"):
        dsl.Program([wrong_width], dsl.Pipe([]), []).trace()

    @dsl.system(buf=65)
    def too_wide(buf):
        return {"buf": buf}
    with pytest.raises(ValueError, match="window"):
        dsl.Program([too_wide], dsl.Pipe([]), []).trace()


def test_generated_window_code_in_both_layouts():
    """The window column's device layout is fixed when the program is built (codegen.WINDOW_SOA_MIN_ROWS): element stride n for
    large executors, a compile-time 1 below; the object says which (bit 30 of its column widths) next to the window bit."""
    from elodin_amd import codegen
    tp = R.program().trace()
    small = codegen.generate_source(tp, "float64", 0, window_soa=False)
    large = codegen.generate_source(tp, "float64", 0, window_soa=True)
    slot = tp.windows["v_rel_accel_buffer"][0]
    assert "constexpr size_t w_n = 1;" in small and "const size_t w_n = P.n;" in large
    assert f"* (size_t)1440;" in small and f"W{slot} = static_cast<T*>(P.model_cols[{slot}]) + (size_t)(w_act ? w_row : P.n - 1) * (size_t)1;" in large
    assert "1440u | 0x80000000u}" in small.replace(", 1u", "") or "1440u | 0x80000000u," in small
    assert "1440u | 0x80000000u | 0x40000000u" in large
    for src in (small, large):
        assert "#pragma unroll 8" in src and "if (w_act) W" in src          # counted scan loop; stores only from lanes that own a row
        assert "kPreReadsAccel = true" in src                               # v_rel_accel reads world_accel in front of six_dof
