#!/usr/bin/env python3
"""examples/falcon9/main.py — the reference's FULL mission script — compiled by THIS repo's front end, frozen for the GPU box,
together with the command stream its own post_step produced when the program flew closed-loop here.

Build container only (/root/reference):   python tests/golden/make_falcon9_main_program.py

main.py is imported UNMODIFIED under elodin_amd.compat; its recorded `world.run(...)` is resolved like World.build resolves it
(65 component columns: plant + sensors + truth ghost + display scoring).  The program is then stepped on the CPU walker with
main.py's OWN post_step on the server loop's cadence, the UDP bridge replaced by oracle/falcon9_fsw.c (exactly the loop
tests/test_compat_reference_scripts.py::test_falcon9_full_mission_script_... checks against the reference-flown fixture), and
every write post_step makes is recorded per tick.  Output tests/golden/falcon9_main_program.json: the generated HIP source
(this repo's compiler output), its column table, the booster's spawned row of every column, and the recorded writes of the
first 1,000 ticks — so tests/test_gpu_falcon9_main.py can fly the SAME closed loop through the generated gfx950 kernel without
the reference checkout (the flight software's answers replayed, the plant and sensors computed on the GPU) and compare with
tests/golden/falcon9_closed_loop.json."""
import sys as _sys
_sys.dont_write_bytecode = True
import importlib.util
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.setrecursionlimit(100000)
REF = Path("/root/reference/examples/falcon9")
TICKS = 1000

import numpy as np  # noqa: E402

import elodin_amd.compat as compat  # noqa: E402
from elodin_amd import _lib as L  # noqa: E402
from elodin_amd import codegen  # noqa: E402
from oracle import falcon9_fsw as fsw_mod  # noqa: E402
from tests import dsl_numpy  # noqa: E402

compat.install(run="record", inert=("polars",))
sys.path.insert(0, str(REF))
spec = importlib.util.spec_from_file_location("ref_falcon9_main", REF / "main.py")
main = importlib.util.module_from_spec(spec)
sys.modules["ref_falcon9_main"] = main
spec.loader.exec_module(main)
world = main.world
run = world.compat_run
plan = world.build(run["system"], simulation_rate=run["simulation_rate"], telemetry_rate=run["telemetry_rate"], _dry=True)
tp = plan["effectors"].trace()
booster = next(e for e, nm in world._names.items() if nm == "booster")


def row_of(name, width):
    try:
        rows, ids = world.column(name)
    except KeyError:
        return np.zeros((1, width))
    hit = np.nonzero(ids == booster)[0]
    return rows[hit[:1]].astype(np.float64).reshape(1, -1) if len(hit) else np.zeros((1, width))


comps = {name: (np.array([[1.0 if booster in world.column(name[4:])[1] else 0.0]]) if name.startswith("has:") else row_of(name, w))
         for name, w in tp.columns}
initial = {k: v.copy() for k, v in comps.items()}
pos, vel, inertia = (row_of(k, w) for k, w in (("world_pos", 7), ("world_vel", 6), ("inertia", 7)))
body0 = {"world_pos": pos.copy(), "world_vel": vel.copy(), "inertia": inertia.copy(), "world_accel": np.zeros((1, 6))}
acc = np.zeros((1, 6))
body = {"world_pos": pos, "world_vel": vel, "world_accel": acc, "inertia": inertia}
fsw = fsw_mod.Fsw(fsw_mod.read_raw_profile(REF / "data" / "crs12" / "stage1_raw.json"))


class OracleBridge:
    def exchange(self, state):
        return fsw.step(np.asarray(state, dtype=np.float64))


main.bridge = OracleBridge()
writes = {}
now = [0]


class Ctx:
    def component_batch_operation(self, reads=None, writes_=None, **kw):
        w = kw.get("writes", writes_)
        if w:
            for name, v in w.items():
                k = name.split(".", 1)[1]
                v = np.asarray(v, dtype=np.float64).reshape(-1)
                (body[k] if k in body else comps[k])[0] = v
                writes.setdefault(str(now[0]), {})[k] = v.tolist()
            return None
        out = {}
        for name in reads:
            k = name.split(".", 1)[1]
            src = body[k] if k in body else (comps[k] if k in comps else row_of(k, 1))
            out[name] = np.array(src[0], dtype=np.float64).reshape(-1)
        return out


ctx = Ctx()
for tick in range(1, TICKS + 1):
    dsl_numpy.program_tick(tp, pos, vel, acc, inertia, comps, tick, plan["dt"], L.SEMI_IMPLICIT)
    now[0] = tick
    main.post_step(tick - 1, ctx)

codegen.build(tp, "float64", plan["integrator"])
doc = {
    "variant": codegen.last_variant[0],
    "source": codegen.generate_variant(tp, codegen.last_variant[0], "float64", plan["integrator"]),
    "columns": [[n_, w] for n_, w in tp.columns], "mats": {k: list(v) for k, v in tp.table.mats.items()},
    "integrator": plan["integrator"], "simulation_time_step": plan["dt"], "ticks": TICKS,
    "body": {k: v.tolist() for k, v in body0.items()},
    "initial": {k: v.tolist() for k, v in initial.items()},
    "writes": writes,                  # {tick after which post_step wrote: {component: values}} — the flight software's answers
    # what main.py's post_step packs besides the sensor reads: its own module-level constants, recorded as data for the live bridge
    # (tests/falcon9_bridge.py) that flies this loop on the GPU box
    "exchange": {"period_ticks": int(main.guidance_period_ticks), "sim_time_step": float(main.SIM_TIME_STEP),
                 "guidance_values": [float(v) for v in main.guidance_values], "fin_wn": float(main.fin_wn),
                 "divert_speed_cap": float(main.divert_speed_cap), "steer_tilt_cap": float(main.steer_tilt_cap), "upper_kg": float(main.upper_kg),
                 "state_floats": int(main.STATE_FLOATS), "reads": list(main.READS)},
    "final": {"lifted": float(comps["lifted"][0, 0]), "fsw_phase": float(fsw.peek()["phase"])},
}
out = ROOT / "tests" / "golden" / "falcon9_main_program.json"
out.write_text(json.dumps(doc))
print(out, out.stat().st_size, "bytes; variant", doc["variant"] + ";", len(doc["columns"]), "columns;", len(writes), "ticks with writes;",
      "lifted", doc["final"]["lifted"], "phase", doc["final"]["fsw_phase"])
