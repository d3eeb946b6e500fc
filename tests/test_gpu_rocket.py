"""The reference's rocket example through the generated gfx950 kernel against the reference's own CI baseline rows
(scripts/ci/baseline/rocket-csv; see tests/test_rocket_reference.py for what the example exercises): one tick per launch so
every golden row is compared, then fused launches and a wave-and-a-bit of identical rockets for the same end state."""
import numpy as np
import pytest

import elodin_amd as ea
from elodin_amd import _lib as L
from tests import rocket_dsl as R, rocket_util as U

pytestmark = pytest.mark.gpu


def _exec(n, **kw):
    pos, vel, inertia, comps = R.spawn(n)
    comps["v_rel_accel_buffer"] = comps["v_rel_accel_buffer"].reshape(n, R.LP_BUFFER_SIZE, 3)    # [n, rows, w] = a window
    return ea.HipExec(pos, vel, inertia, simulation_time_step=U.GOLDEN["simulation_time_step"], integrator=L.RK4, effectors=R.program(),
                      columns=comps, **kw)


def _row(hip, k=0):
    got = {name: hip.component(name)[k] for name in U.GOLDEN["rows"] if name in hip._aux}
    got.update(world_pos=hip.world_pos[k], world_vel=hip.world_vel[k], world_accel=hip.world_accel[k], force=hip.force[k], inertia=hip.inertia[k])
    return got


def test_rocket_every_golden_row_one_tick_per_launch():
    hip = _exec(1)
    worst = {}
    for tick in range(1, 101):
        hip.run(1)
        U.check_row(tick, _row(hip), worst)
    print("rocket vs reference baseline, HIP:", {k: f"{v:.1e}" for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:6]})
    U.assert_all_columns(worst)
    want = np.array(U.GOLDEN["v_rel_accel_buffer_final"])
    assert np.abs(hip.component("v_rel_accel_buffer")[0].ravel() - want).max() < 1e-9 * np.abs(want).max()


@pytest.mark.parametrize("n,k", [(1, 25), (97, 100), (97, 7)])
def test_rocket_fused_launches_and_many_rockets_end_on_the_golden_row(n, k):
    """The window and the previous tick's world_accel carry across the ticks of one launch (and across launches when 100 is not
    a multiple of k); 97 rockets = one full wave + a ragged one, every row must be the golden rocket."""
    hip = _exec(n, ticks_per_launch=k)
    hip.run(100)
    for row in (0, n - 1, n // 2):
        worst = {}
        U.check_row(100, _row(hip, row), worst)
        U.assert_all_columns(worst)
    want = np.array(U.GOLDEN["v_rel_accel_buffer_final"])
    assert np.abs(hip.component("v_rel_accel_buffer")[n - 1].ravel() - want).max() < 1e-9 * np.abs(want).max()


def test_rocket_window_in_both_device_layouts():
    """A window column is entity-major below codegen.WINDOW_SOA_MIN_ROWS entities and element-major from there on (the host lays it
    out, the kernel picks its strides from n): a batch just past the switch ends on the golden row like the small ones do."""
    from elodin_amd import codegen
    n = codegen.WINDOW_SOA_MIN_ROWS + 5
    hip = _exec(n, ticks_per_launch=10)
    hip.run(100)
    for row in (0, n - 1, 12345):
        worst = {}
        U.check_row(100, _row(hip, row), worst)
        U.assert_all_columns(worst)
    want = np.array(U.GOLDEN["v_rel_accel_buffer_final"])
    assert np.abs(hip.component("v_rel_accel_buffer")[n - 1].ravel() - want).max() < 1e-9 * np.abs(want).max()


def test_unmodified_rocket_script_through_its_frozen_kernel():
    """examples/rocket/main.py imported UNMODIFIED under elodin_amd.compat in the build container (polars subset, map_coordinates,
    the window spelled concatenate / lax.scan): the kernel text the code generator emitted for it there
    (tests/golden/make_rocket_program.py -> rocket_program.json; tests/test_compat_reference_scripts.py checks the script still
    generates it and walks it on the CPU), compiled here, every golden row at one tick per launch and the end state fused."""
    import json
    from pathlib import Path
    from elodin_amd import dsl
    doc = json.loads((Path(__file__).parent / "golden" / "rocket_program.json").read_text())
    for k, n in ((1, 1), (20, 70)):
        prog = dsl.FrozenProgram(doc["source"], doc["columns"], doc["mats"], windows=doc["windows"])
        rep = lambda a: np.repeat(np.asarray(a, dtype=np.float64).reshape(1, -1), n, axis=0)
        body = doc["body"]
        cols = {name: rep(v) for name, v in doc["initial"].items()}
        for name, (_, rows, width) in doc["windows"].items():
            cols[name] = cols[name].reshape(n, rows, width)
        hip = ea.HipExec(rep(body["world_pos"]), rep(body["world_vel"]), rep(body["inertia"]), world_accel=rep(body["world_accel"]),
                         simulation_time_step=doc["simulation_time_step"], time_step=doc["time_step"], integrator=doc["integrator"],
                         effectors=prog, columns=cols, ticks_per_launch=k)
        worst = {}
        if k == 1:
            for tick in range(1, 101):
                hip.run(1)
                U.check_row(tick, _row(hip), worst)
        else:
            hip.run(100)
            for row in (0, n - 1):
                U.check_row(100, _row(hip, row), worst)
        print(f"examples/rocket/main.py unmodified, frozen kernel, {k} tick(s) per launch:", {a: f"{b:.1e}" for a, b in sorted(worst.items(), key=lambda kv: -kv[1])[:4]})
        U.assert_all_columns(worst)
        want = np.array(U.GOLDEN["v_rel_accel_buffer_final"])
        assert np.abs(hip.component("v_rel_accel_buffer")[n - 1].ravel() - want).max() < 1e-9 * np.abs(want).max()
