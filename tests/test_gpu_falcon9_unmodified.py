"""The reference's Falcon 9 plant — examples/falcon9/sim.py `build_powered`: nine engines with their ignition state machine,
valves, TVC, grid fins, RCS allocation, tanks, US-76 atmosphere, aero tables, WGS84 frames, pad clamp, leg contact, sensor
models; 23 systems around six_dof(SemiImplicit), 62 components — on the GPU, in the kernel THIS repo's code generator emitted for
the unmodified script (tests/golden/make_falcon9_plant_program.py -> tests/golden/falcon9_plant_program.json; the script can only
be imported where the reference checkout is).  One whole 10 s window at 1 kHz (the transonic `maxq` case: engines running, TVC +
fins + RCS active, wind) against every checkpoint of the trajectory the reference's own functions flew
(tests/golden/falcon9_plant.json), 1e-9 on 43 columns — the same fixtures and bound as for this repo's own model of the vehicle
(tests/test_gpu_falcon9_plant.py).  tests/test_compat_reference_scripts.py checks in the build container that the script still
generates this text and walks all three windows on the CPU."""
from pathlib import Path

import numpy as np
import pytest

from elodin_amd import dsl
from tests import falcon9_plant_util as pu

pytestmark = pytest.mark.gpu
GOLDEN = Path(__file__).resolve().parent / "golden"


@pytest.mark.parametrize("source", ["source", "source_f64_guarded"])
def test_unmodified_falcon9_plant_kernel_flies_the_reference_window(source, ticks_per_launch=500):
    """source_f64_guarded: the same program generated with guarded selects (codegen: the expensive arm of a `where` that nobody
    else needs — a sensor's noise draw behind its sample-time test — runs behind a wave-level branch): same values."""
    import elodin_amd as ea
    doc = pu.load_program_fixture()
    case = doc["case"]
    init = {k: np.asarray(v, dtype=np.float64) for k, v in doc["initial"].items()}
    names = [n for n, _ in doc["columns"]]
    prog = dsl.FrozenProgram(doc[source], doc["columns"], doc["mats"])
    hip = ea.HipExec(init["world_pos"], init["world_vel"], init["inertia"], world_accel=init["world_accel"],
                     simulation_time_step=doc["simulation_time_step"], integrator=doc["integrator"], effectors=prog,
                     columns={n: init[n] for n in names}, ticks_per_launch=ticks_per_launch)
    worst, at = {}, 0
    checkpoints = [c["tick"] for c in pu.PLANT[case]["checkpoints"] if c["tick"] % ticks_per_launch == 0]
    assert checkpoints[-1] == 10_000 and len(checkpoints) == 20
    for tick in checkpoints:
        hip.run(tick - at)
        at = tick
        body = {"world_pos": hip.world_pos, "world_vel": hip.world_vel, "world_accel": hip.world_accel, "force": hip.force, "inertia": hip.inertia}
        for k, e in pu.compare(case, tick, lambda name: body[name] if name in body else hip.component(name)).items():
            worst[k] = max(worst.get(k, 0.0), e)
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:5]
    print(f"unmodified falcon9 plant on the GPU ({source}), window {case}, 10,000 ticks: worst of {len(worst)} columns:", ", ".join(f"{k} {e:.1e}" for k, e in top))
    assert len(worst) == 43 and max(worst.values()) < 1e-9, top
