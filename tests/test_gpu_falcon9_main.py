"""examples/falcon9/main.py — the reference's FULL mission script, 65 component columns — through its generated gfx950 kernel,
closed loop from the pad through liftoff, against the ascent the reference's own code flew (tests/golden/falcon9_closed_loop.json).

The script, its post_step and the flight-software oracle live where the reference checkout is; what runs here is the kernel
THIS repo's code generator emitted for the unmodified script there, with the flight software's answers replayed from the
command stream recorded there (tests/golden/make_falcon9_main_program.py: main.py's own post_step + oracle/falcon9_fsw.c around
the CPU walk of the same program — the loop tests/test_compat_reference_scripts.py pins to 4e-15).  Plant, sensors and every
other system are computed on the GPU tick by tick; after each tick the recorded writes of that tick are applied and uploaded,
like copy_db_to_world does before the next batch."""
import json
from pathlib import Path

import numpy as np
import pytest

from elodin_amd import dsl
from tests import falcon9_closed_loop_util as cu

pytestmark = pytest.mark.gpu
GOLDEN = Path(__file__).resolve().parent / "golden"


@pytest.mark.skipif(not cu.FLIGHTS, reason="closed-loop fixture not generated")
def test_full_mission_script_kernel_flies_the_reference_ascent_through_liftoff():
    import elodin_amd as ea
    doc = json.loads((GOLDEN / "falcon9_main_program.json").read_text())
    flight = cu.FLIGHTS["0"]
    n = 3                                                            # identical boosters: lanes do not interact
    rep = lambda a: np.repeat(np.asarray(a, dtype=np.float64).reshape(1, -1), n, axis=0)
    prog = dsl.FrozenProgram(doc["source"], doc["columns"], doc["mats"])
    body = doc["body"]
    hip = ea.HipExec(rep(body["world_pos"]), rep(body["world_vel"]), rep(body["inertia"]), world_accel=rep(body["world_accel"]),
                     simulation_time_step=doc["simulation_time_step"], integrator=doc["integrator"], effectors=prog,
                     columns={k: rep(v) for k, v in doc["initial"].items()}, ticks_per_launch=1)
    cps = {c["tick"]: c for c in flight["checkpoints"] if c["tick"] <= doc["ticks"]}
    worst, seen = {}, 0
    for tick in range(1, doc["ticks"] + 1):
        hip.run(1)
        w = doc["writes"].get(str(tick))
        if w:                                                         # post_step's writes after this tick (the FSW's commands)
            for comp, v in w.items():
                (getattr(hip, comp) if comp in ("world_pos", "world_vel") else hip._aux[comp])[:] = np.asarray(v)
            hip.upload()
        if tick in cps:
            cp = dict(cps[tick], state={k: v for k, v in cps[tick]["state"].items() if k != "fsw"})
            cp["state"]["fsw"] = {}
            state = {"world_pos": hip.world_pos[:1], "world_vel": hip.world_vel[:1], "world_accel": hip.world_accel[:1], "force": hip.force[:1],
                     "inertia": hip.inertia[:1]}
            for k, e in cu.compare(flight, cp, lambda name: state[name] if name in state else hip._aux[name][:1]).items():
                worst[k] = max(worst.get(k, 0.0), e)
            for name in ("world_pos", "thrust_total", "gps_pos"):
                a = state[name] if name in state else hip._aux[name]
                full = getattr(hip, name) if name in state else hip._aux[name]
                assert np.array_equal(full, np.repeat(full[:1], n, axis=0)), name
            seen += 1
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:5]
    print(f"examples/falcon9/main.py unmodified, generated kernel, closed loop (FSW replayed): worst of {len(worst)} quantities at {seen} "
          f"checkpoints:", ", ".join(f"{k} {e:.1e}" for k, e in top))
    assert seen >= 7 and len(worst) >= 45 and max(worst.values()) < 1e-9, top
    assert float(hip._aux["lifted"][0, 0]) == doc["final"]["lifted"] == 1.0
