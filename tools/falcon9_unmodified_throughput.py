"""Throughput of the kernel the front end generates for the reference's UNMODIFIED Falcon 9 plant (examples/falcon9/sim.py
`build_powered` + the open-loop `maxq` command script; tests/golden/falcon9_plant_program.json — no reference checkout needed),
built the way the campaign builds its programs: float32, hardware transcendentals, state in registers for 1,000 ticks per launch.

    python tools/falcon9_unmodified_throughput.py [rollouts] [ticks] [guarded]     # default 32768 5000, plain selects
`guarded`: the text generated with guarded selects (codegen._Emitter.block: expensive `where` arms behind a wave-level branch).
Every rollout flies the same window (identical spawn state), so the result is also compared with the float64 trajectory the
reference's own functions flew (tests/golden/falcon9_plant.json) at the f32 tolerances of tests/falcon9_plant_util.py."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402

import elodin_amd as ea  # noqa: E402
from elodin_amd import dsl  # noqa: E402
from tests import falcon9_plant_util as pu  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
doc = pu.load_program_fixture()
init = {k: np.repeat(np.asarray(v, dtype=np.float64).reshape(1, -1), n, axis=0) for k, v in doc["initial"].items()}
names = [c for c, _ in doc["columns"]]
guarded = len(sys.argv) > 3 and sys.argv[3] == "guarded"
prog = dsl.FrozenProgram(doc["source_f32_fast_guarded" if guarded else "source_f32_fast"], doc["columns"], doc["mats"])
# float32 cannot hold ECEF metres to better than 0.5 m: this is a throughput run; positions are compared at that resolution
hip = ea.HipExec(init["world_pos"], init["world_vel"], init["inertia"], world_accel=init["world_accel"], dtype=np.float32,
                 simulation_time_step=doc["simulation_time_step"], integrator=doc["integrator"], effectors=prog,
                 columns={c: init[c] for c in names}, ticks_per_launch=1000)
hip.invoke_batch(1000)
t0 = time.perf_counter()
tm = hip.invoke_batch(ticks - 1000)
dt = time.perf_counter() - t0
hip.download()
per_tick = tm.kernel_device_ms / (ticks - 1000) * 1e3
print(f"unmodified Falcon 9 plant, float32 fast-math{', guarded selects' if guarded else ''}, {n} rollouts x {ticks - 1000} timed ticks (1,000 per launch): "
      f"{per_tick:.3f} us per tick = {n * (ticks - 1000) / dt:.3e} rollout-steps/s ({len(names)} component columns, "
      f"{sum(w for _, w in doc['columns'])} values per rollout)")
body = {"world_pos": hip.world_pos, "world_vel": hip.world_vel, "world_accel": hip.world_accel, "force": hip.force, "inertia": hip.inertia}
errs = pu.compare(doc["case"], ticks, lambda name: (body[name] if name in body else hip.component(name))[:1], floors=pu.FLOORS_F32)
top = sorted(errs.items(), key=lambda kv: -kv[1])[:6]
print(f"vs the reference-flown float64 window at tick {ticks} (f32 floors): worst " + ", ".join(f"{k} {e:.1e}" for k, e in top))
same = all(np.array_equal(hip.world_vel[0], hip.world_vel[k]) for k in (1, n // 2, n - 1))
print("all rollouts identical:", same)
