"""The reference's closed-loop quadcopter (examples/drone: flight plan -> attitude PID -> rate PID -> mixer -> motor model ->
six_dof(SemiImplicit) at 900 Hz sub-steps -> IMU / magnetometer models, 300 Hz simulation, 100 Hz telemetry) on the GPU.

The script itself can only be imported where the reference checkout is, so what runs here is the kernel THIS repo's code
generator emitted for it there (tests/golden/make_drone_program.py: the unmodified main.py under elodin_amd.compat ->
generated HIP source + column table + spawned columns, tests/golden/drone_program.json), compiled on this box and stepped
100 ticks against the reference's own CI baseline rows (tests/golden/drone.json <- scripts/ci/baseline/drone-csv).
tests/test_compat_reference_scripts.py checks in the build container that the unmodified script still generates this program
and that the CPU walk of the trace lands on the same rows."""
import json
from pathlib import Path

import numpy as np
import pytest

from elodin_amd import dsl
from tests.drone_util import drone_errors, drone_verdict

pytestmark = pytest.mark.gpu
GOLDEN = Path(__file__).resolve().parent / "golden"


def build(doc, ticks_per_launch, n=1):
    import elodin_amd as ea
    prog = dsl.FrozenProgram(doc["source"], doc["columns"], doc["mats"], substeps=doc["substeps"])
    rep = lambda a: np.repeat(np.asarray(a, dtype=np.float64).reshape(1, -1), n, axis=0)
    body = doc["body"]
    return ea.HipExec(rep(body["world_pos"]), rep(body["world_vel"]), rep(body["inertia"]), world_accel=rep(body["world_accel"]),
                      simulation_time_step=doc["simulation_time_step"], time_step=doc["time_step"], integrator=doc["integrator"],
                      effectors=prog, columns={k: rep(v) for k, v in doc["initial"].items()}, ticks_per_launch=ticks_per_launch)


@pytest.mark.parametrize("ticks_per_launch", [3, 9])
def test_drone_generated_kernel_lands_on_the_reference_baseline(ticks_per_launch):
    doc = json.loads((GOLDEN / "drone_program.json").read_text())
    gold = json.loads((GOLDEN / "drone.json").read_text())
    k = doc["substeps"]
    n = 130                                    # every lane flies the same drone: three waves, the last one ragged
    hip = build(doc, ticks_per_launch, n)
    worst, at = {}, 0
    for tick in [t for t in gold["tick"] if t > 0]:
        hip.run((tick - at) * k)               # k integrator sub-steps per world tick
        at = tick
        cur = {name: np.asarray(hip.component(name), dtype=np.float64).reshape(n, -1) for name, _ in doc["columns"]}
        cur.update(world_pos=hip.world_pos, world_vel=hip.world_vel, world_accel=hip.world_accel)
        for name, v in cur.items():
            assert np.array_equal(v, np.repeat(v[:1], n, axis=0), equal_nan=True), (name, tick)      # lanes do not interact
        drone_errors(gold, gold["tick"].index(tick), {name: v[0] for name, v in cur.items()}, worst)
    print(f"drone example on the GPU ({ticks_per_launch} sub-steps per launch) vs reference baseline, worst per component:",
          {name: f"{v:.1e}" for name, v in sorted(worst.items(), key=lambda kv: -kv[1])})
    drone_verdict(worst)
