#!/usr/bin/env python3
"""Closed-loop Falcon 9 ascents flown by the REFERENCE's own code — fixtures for the product's generated kernel.

Run in the build container only (needs /root/reference):
    python tests/golden/make_falcon9_closed_loop.py [row ...]        # default: rows 0 1 2 3, one process each

What flies, and whose code it is:

  plant + sensors   examples/falcon9/main.py is IMPORTED unmodified under tests/golden/refshim.py (numpy in jax's clothes,
                    spatial types delegated to the pinned C oracle, jax.random as JAX computes it): its module body
                    reads the run's parameters (`el.monte_carlo.params`), builds the mission world (`build_mission`) and
                    defines `post_step`.  Every tick runs the reference's own @el.map / @el.system functions in
                    build_mission's pipe order (sim.py:1433-1590): propulsion_systems | six_dof(SemiImplicit) | pad_clamp
                    | ground_contact | derive_geodetic_telemetry | imu_model | gps_model | radar_altimeter_model |
                    pressure_transducers.  Systems that feed neither the dynamics nor the flight software are not run
                    (descent_metrics_latch, effect_visualization, display_model, truth_playback, display_scoring).
  bridge            main.py's own `post_step(tick, ctx)` — packet packing, the guidance cadence (`tick % 10`), the writes
                    back into the command components, stage separation — called like the server loop calls it: once per
                    tick with the index of the tick just finished (impeller2_server.rs:553-678 with the example's
                    ticks_per_telemetry = 1: post_step(k) sees the world after k + 1 ticks).  Only the UDP socket is
                    replaced: `main.bridge` is an object whose exchange() hands the packet to ...
  flight software   oracle/falcon9_fsw.c, the C restatement of controller/src/{main,math,profile}.rs (no rustc here), flying
                    the recorded CRS-12 profile (ELODIN_F9_PROFILE = data/crs12/stage1_raw.json) like the reference's
                    recipe sets it up (main.py:193-199).

Nothing of elodin_amd/ is in this loop.  Rows: 0 = the calibrated defaults of main.py:53-100; 1.. = rows of the example's
own plan (spec.toml, LHS, seed 20170814) as the reference's sampler (`elodin.monte_carlo.sample`, imported from
/root/reference) draws them.  Each flight runs from the pad until the flight software leaves the ascent (Meco -> Flip, 3 s
after cutoff).  Output: tests/golden/falcon9_closed_loop.json — per row the parameter context, the tick of every phase
transition, liftoff / MECO observables, and the full component state + the flight software's navigator at checkpoints.
"""
import sys as _sys
_sys.dont_write_bytecode = True      # the reference checkout is read-only: no __pycache__ next to what is imported from it
import importlib
import json
import math
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
REF = Path(os.environ.get("ELODIN_REFERENCE", "/root/reference"))
OUT = Path(__file__).resolve().parent
PART = OUT / "_closed_loop_parts"
CHECKPOINT_EVERY = 10_000
EARLY_CHECKPOINTS = (1, 10, 41, 51, 201, 211, 1000, 3000)      # first exchange, GPS init, ignition command ...


def contexts(n):
    """Row 0 = main.py's calibrated defaults (an empty context); rows 1.. = the first rows of the example's own plan
    (spec.toml: LHS, seed 20170814, 24 samples) as the reference's sampler writes them (`materialize`, the same import
    tests/golden/make_plan_golden.py uses; `param.<name>` columns are the run's context, lib.rs read_plan)."""
    import csv
    import importlib.util
    import tempfile
    spec = importlib.util.spec_from_file_location("ref_sample", REF / "libs/nox-py/python/elodin/monte_carlo/sample.py")
    ref_sample = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_sample)
    out = [{}]
    if n > 1:
        with tempfile.TemporaryDirectory() as d:
            ref_sample.materialize(REF / "examples" / "falcon9" / "spec.toml", Path(d) / "plan.csv")
            rows = list(csv.DictReader(open(Path(d) / "plan.csv")))
        for r in rows[: n - 1]:
            out.append({k.split(".", 1)[1]: float(v) for k, v in r.items() if k.startswith("param.")})
    return out


def fly(row, context):
    from oracle import falcon9_fsw as fsw_mod
    from tests.golden import refshim
    import numpy as np
    sys.argv = [sys.argv[0], str(REF)]                                # make_falcon9_fixtures reads the reference path from argv[1]
    from tests.golden import make_falcon9_fixtures as base          # plant_tick, flat, L (installs the shim, imports sim)
    jax, jnp, el = sys.modules["jax"], sys.modules["jax.numpy"], sys.modules["elodin"]
    el.monte_carlo.CONTEXT = dict(context)                            # the run's parameter overrides, read by main.py:103
    os.environ.pop("ELODIN_MONTE_CARLO_CONTEXT", None)
    import main                                                       # the reference's harness, unmodified
    for k, v in context.items():
        assert main.get_param(k) == v, (k, main.get_param(k), v)
    sim = base.sim
    S = {name: main.system[name] for name in main.system.names() if name != "six_dof"}
    assert main.system["six_dof"].integrator is not None and [e.__name__ for e in main.system["six_dof"].effectors] == \
        ["gravity_and_frame_forces", "apply_body_wrenches"]

    s = dict(main.world.entities["booster"])
    s["world_accel"], s["force"] = np.zeros(6), np.zeros(6)
    s["attitude_setpoint"] = el.Quaternion(np.asarray(s["attitude_setpoint"].vector() if hasattr(s["attitude_setpoint"], "vector") else s["attitude_setpoint"]))

    profile = fsw_mod.read_raw_profile(REF / "examples" / "falcon9" / "data" / "crs12" / "stage1_raw.json")
    fsw = fsw_mod.Fsw(profile)
    exchanges = []

    class OracleBridge:                     # main.py:221-243 minus the socket
        def exchange(self, state):
            cmd = fsw.step(np.asarray(state, dtype=np.float64))
            exchanges.append(1)
            return cmd
    main.bridge = OracleBridge()

    class Ctx:                              # el.StepContext.component_batch_operation over the component dict
        def component_batch_operation(self, reads=None, writes=None):
            if writes:
                for name, v in writes.items():
                    k = name.split(".", 1)[1]
                    s[k] = el.Quaternion(v) if k == "attitude_setpoint" else jnp.asarray(np.asarray(v, dtype=np.float64))
                return None
            out = {}
            for name in reads:
                v = s[name.split(".", 1)[1]]
                if hasattr(v, "asarray"):
                    v = v.asarray()
                elif isinstance(v, el.Quaternion):
                    v = v.vector()
                out[name] = np.asarray(v, dtype=np.float64).reshape(-1)
            return out
    ctx = Ctx()

    def snapshot(tick):
        st = base.flat({k: v for k, v in s.items() if k not in ("display_alt", "display_speed", "thrust_viz", "plume_viz", "smoke_viz",
                                                                 "pad_smoke_viz", "landing_smoke_viz", "score_state", "descent_metrics")})
        pk = fsw.peek()
        st["fsw"] = {k: (base.L(v) if hasattr(v, "shape") else v) for k, v in pk.items()}
        return {"tick": tick, "state": st}

    init = snapshot(0)["state"]
    checkpoints, transitions, last_phase = [], {}, 0.0
    liftoff_tick = None
    t0 = time.time()
    tick = 0
    while True:
        tick += 1
        base.plant_tick(s, tick, S)
        s["sensor_tick"], s["imu_accel"], s["imu_gyro"] = S["imu_model"](s["sensor_tick"], s["world_pos"], s["world_vel"], s["inertia"],
                                                                         s["engine_wrench"], s["aero_wrench"], s["fin_wrench"], s["rcs_wrench"])
        s["gps_timer"], s["gps_pos"], s["gps_vel"], s["gps_count"] = S["gps_model"](
            s["sensor_tick"], s["gps_timer"], s["world_pos"], s["world_vel"], s["mach"], s["thrust_total"], s["gps_pos"], s["gps_vel"], s["gps_count"])
        s["radar_timer"], s["radar_range"], s["radar_count"] = S["radar_altimeter_model"](s["radar_timer"], s["world_pos"], s["radar_range"], s["radar_count"])
        s["pressure_meas"] = S["pressure_transducers"](s["sensor_tick"], s["tank_pressure_lox"], s["tank_pressure_rp1"],
                                                       s["inlet_pressure_lox"], s["inlet_pressure_rp1"])
        main.post_step(tick - 1, ctx)       # the server loop's call after the tick (ticks_per_telemetry = 1)
        if liftoff_tick is None and float(np.asarray(s["lifted"]).reshape(-1)[0]) > 0.5:
            liftoff_tick = tick
        phase_now = fsw.peek()["phase"]
        if phase_now != last_phase:
            transitions[str(int(phase_now))] = tick
            last_phase = phase_now
            checkpoints.append(snapshot(tick))
        elif tick % CHECKPOINT_EVERY == 0 or tick in EARLY_CHECKPOINTS:
            checkpoints.append(snapshot(tick))
        if tick % 20_000 == 0:
            print(f"  row {row}: tick {tick}, phase {int(phase_now)}, alt {float(s['altitude_geodetic'][0]) / 1e3:.2f} km, "
                  f"speed {float(s['ground_speed'][0]):.1f} m/s, {time.time() - t0:.0f} s", flush=True)
        if fsw.beyond_ascent or tick >= int(os.environ.get("F9_CLOSED_LOOP_MAX_TICKS", "1000000")):   # (trial runs only)
            break
        assert tick < 260_000, "no MECO within 260 s"
    if checkpoints[-1]["tick"] != tick:
        checkpoints.append(snapshot(tick))
    print(f"  row {row}: {tick} ticks, transitions {transitions}, liftoff tick {liftoff_tick}, "
          f"final alt {float(s['altitude_geodetic'][0]) / 1e3:.2f} km, speed {float(s['ground_speed'][0]):.1f} m/s, {time.time() - t0:.0f} s", flush=True)
    return {"context": dict(context), "guidance_values": [float(v) for v in main.guidance_values], "upper_kg": float(main.upper_kg),
            "lox_kg": float(main.lox_kg), "rp1_kg": float(main.rp1_kg), "ticks": tick, "exchanges": len(exchanges),
            "transitions": transitions, "liftoff_tick": liftoff_tick, "init": init, "checkpoints": checkpoints}


def main_():
    args = [a for a in sys.argv[1:]]
    if args and args[0] == "--one":
        row = int(args[1])
        ctxs = json.loads(Path(args[2]).read_text())
        PART.mkdir(exist_ok=True)
        res = fly(row, ctxs[row])
        (PART / f"row{row}.json").write_text(json.dumps(res, separators=(",", ":")))
        return
    rows = [int(a) for a in args] or [0, 1, 2, 3]
    ctxs = contexts(max(rows) + 1)
    PART.mkdir(exist_ok=True)
    (PART / "contexts.json").write_text(json.dumps(ctxs))
    procs = [subprocess.Popen([sys.executable, __file__, "--one", str(r), str(PART / "contexts.json")]) for r in rows]
    for p in procs:
        if p.wait() != 0:
            raise SystemExit("a flight failed")
    merged = {str(r): json.loads((PART / f"row{r}.json").read_text()) for r in rows}
    (OUT / "falcon9_closed_loop.json").write_text(json.dumps(merged, separators=(",", ":")))
    print("falcon9_closed_loop.json:", (OUT / "falcon9_closed_loop.json").stat().st_size, "bytes")


if __name__ == "__main__":
    main_()
