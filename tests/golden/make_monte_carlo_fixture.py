#!/usr/bin/env python3
"""Golden runs of the reference's Monte-Carlo SITL example from the REFERENCE's own code, executed here on numpy.

Run in the build container only (needs /root/reference):   python tests/golden/make_monte_carlo_fixture.py

examples/monte-carlo is plain Python against jax.numpy + the elodin wheel; under tests/golden/refshim.py (numpy in jax's
clothes) `sim.py` and `main.py` import and run UNMODIFIED from /root/reference:

  sim.py     build(params): the spawned `vehicle` components and the `point_mass` map (a drag coefficient gathered from a
             lookup table by a velocity-derived row index, optional probe rows, semi-implicit point-mass update)
  main.py    the module body and post_step(tick, ctx): with ELODIN_MONTE_CARLO_CONTROLLER=0 the saturated PD law
             `command = clip((target - position) * 1.2 - velocity * 0.35, -20, 20)` written back with ctx.write_component, and the
             run's result record at tick MAX - 1
  the loop   impeller2_server.rs:553-678 with ticks_per_telemetry = 1 (no telemetry_rate): one tick, commit, post_step(tick)

for rows of the example's own plan.csv (plus the declared defaults), at the sweep's grid size 4,096
(monte_carlo_scaling_sweep.py --grid-size) with 0 and with 64 probe rows.

Output: tests/golden/monte_carlo_example.json — per run the parameters, position / velocity / command / specific_force every
TICK_STRIDE ticks and at the end, and the result record.
"""
import sys as _sys
_sys.dont_write_bytecode = True      # the reference checkout is read-only
import csv
import importlib
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
OUT = Path(__file__).resolve().parent
EX = REF / "examples" / "monte-carlo"

import numpy as np  # noqa: E402

from tests.golden import refshim  # noqa: E402

GRID = 4096
TICK_STRIDE = 20
PLAN_ROWS = (0, 1, 2, 3, 17, 42, 63, 99)

os.environ["ELODIN_MONTE_CARLO_GRID_SIZE"] = str(GRID)
os.environ["ELODIN_MONTE_CARLO_CONTROLLER"] = "0"
jax, jnp, el = refshim.install(str(EX))


class Ctx:
    """el.StepContext over the one entity's components (elodin.pyi:25-88)."""

    def __init__(self, comps):
        self.comps = comps

    def read_component(self, pair_name, timestamp=None):
        entity, comp = pair_name.split(".")
        assert entity == "vehicle", pair_name
        return np.asarray(self.comps[comp], dtype=np.float64)

    def write_component(self, pair_name, data, timestamp=None):
        entity, comp = pair_name.split(".")
        assert entity == "vehicle" and comp in self.comps, pair_name
        self.comps[comp] = np.asarray(data, dtype=np.float64).copy()


def fly(params, probe_rows):
    os.environ["ELODIN_MONTE_CARLO_PROBE_ROWS"] = str(probe_rows)
    el.monte_carlo.CONTEXT = dict(params)
    del el.monte_carlo.RESULTS[:]
    for m in ("main", "sim"):
        sys.modules.pop(m, None)
    main = importlib.import_module("main")                      # runs build(params) and the module body of main.py
    comps = {k: np.asarray(v, dtype=np.float64).copy() for k, v in main.world.entities["vehicle"].items()}
    ctx = Ctx(comps)
    rows = []
    for tick in range(main.DEFAULT_MAX_TICKS):
        pos, vel, sf = main.system(comps["position"], comps["velocity"], comps["command"])        # the point_mass map
        comps["position"], comps["velocity"], comps["specific_force"] = (np.asarray(x, dtype=np.float64).reshape(1) for x in (pos, vel, sf))
        main.post_step(tick, ctx)                                                                  # main.py:88-106
        if (tick + 1) % TICK_STRIDE == 0 or tick == main.DEFAULT_MAX_TICKS - 1:
            rows.append([tick + 1] + [float(comps[c][0]) for c in ("position", "velocity", "command", "specific_force")])
    assert len(el.monte_carlo.RESULTS) == 1, el.monte_carlo.RESULTS
    return {"params": params, "probe_rows": probe_rows, "columns": ["ticks_done", "position", "velocity", "command", "specific_force"],
            "rows": rows, "result": {k: float(v) for k, v in el.monte_carlo.RESULTS[0].items()}}


def main():
    plan = list(csv.DictReader(open(EX / "plan.csv")))
    runs = []
    defaults = {"mass": 1.5, "target_x": 30.0, "thrust_gain": 1.0, "wind": 0.0}      # sim.py:18-23 PARAMS defaults
    runs.append(dict(fly(defaults, 0), run_id="defaults"))
    for r in PLAN_ROWS:
        params = {k[len("param."):]: float(v) for k, v in plan[r].items() if k.startswith("param.")}
        for probe in (0, 64):
            runs.append(dict(fly(params, probe), run_id=plan[r]["run_id"], plan_row=r))
    doc = {"source": "examples/monte-carlo/{sim,main}.py run unmodified under tests/golden/refshim.py", "grid_size": GRID,
           "max_ticks": 360, "simulation_rate_hz": 120.0, "runs": runs}
    path = OUT / "monte_carlo_example.json"
    path.write_text(json.dumps(doc))
    print(path, path.stat().st_size, "bytes;", len(runs), "runs; final positions", [round(r["result"]["final_position"], 3) for r in runs[:5]])


if __name__ == "__main__":
    main()
