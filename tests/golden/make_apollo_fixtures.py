#!/usr/bin/env python3
"""Golden Apollo-lander descents flown by the REFERENCE's own Python code, executed here on numpy.

Run in the build container only (needs /root/reference):   python tests/golden/make_apollo_fixtures.py

What runs unmodified from /root/reference/examples/apollo-lander (under tests/golden/refshim.py):
  sim.py     build(params): the spawn of the `lander` entity and every @el.map system of the pipe
             `engine_response | attitude_control | mass_props | six_dof(lunar_gravity | apply_main_thrust |
             apply_rcs_torque, SemiImplicit) | ground_contact | derive_telemetry` (sim.py:225-526), reference.py behind it
  main.py    the module body (parameter plumbing, initial command state) and post_step(tick, ctx): state packing for the
             guidance computer, the 3 deg/exchange attitude slew, RMSE accumulation, the el.monte_carlo.result(...) record
What is restated here, with citations, because it is not Python:
  the six_dof step     pinned C oracle: orc_calc_accel + orc_transform_add_motion (semi_implicit.rs:17-62)
  the server loop      libs/nox-py/src/impeller2_server.rs:553-678,790-791: batches of ticks_per_telemetry = 120/40 = 3
                       ticks, post_step(end_tick = batch start + batch - 1) after each, last batch cut at max_ticks
  the guidance law     controller/src/main.rs:100-262 (Rust, no tests, no vectors; the UDP bridge is assumed never to time
                       out): ported line by line below as the stand-in for main.py's SitlBridge.  The law is therefore on
                       BOTH sides of every comparison made with this fixture — it is not pinned by it; everything else is.

Output: tests/golden/apollo_reference_runs.json — per rollout the plan row, the result record, the post_step tick it was
emitted on, and the lander's components at batch ends every 3,000 ticks.
"""
import sys as _sys
_sys.dont_write_bytecode = True      # the reference checkout is read-only: no __pycache__ next to what is imported from it
import csv
import importlib
import json
import math
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
OUT = Path(__file__).resolve().parent

import numpy as np  # noqa: E402

from oracle import oracle as orc  # noqa: E402
from tests.golden import refshim  # noqa: E402

jax, jnp, el = refshim.install(str(REF / "examples" / "apollo-lander"))
import sim  # noqa: E402  (examples/apollo-lander/sim.py)

CHECKPOINT_EVERY = 3000
TICKS_PER_TELEMETRY = round(sim.SIMULATION_RATE_HZ / sim.TELEMETRY_RATE_HZ)      # world_builder.rs:222-242
assert TICKS_PER_TELEMETRY == 3


# ---- controller/src/main.rs, line by line -------------------------------------------------------------------------------

MIN_THROTTLE = 4_670.0 / 45_040.0
FTP_THROTTLE, EROSION_BAND_MIN = 0.925, 0.65
MAX_DESCENT_RATE_MPS, MIN_DESCENT_RATE_MPS, MIN_VERTICAL_ACCEL_MPS2 = 120.0, 0.5, 0.05
MAX_TILT_BRAKING_DEG, MAX_TILT_APPROACH_DEG, TILT_BLEND_HI_MPS, TILT_BLEND_LO_MPS = 82.0, 30.0, 150.0, 40.0
HSPEED_GAIN, POSITION_AUTHORITY_MPS2, RATE_TRACK_AUTHORITY_MPS = 0.25, 0.5, 12.0
VERTICAL_FB_AUTHORITY_MPS2, HSPEED_FB_AUTHORITY_MPS2, TERMINAL_NULL_ALT_M, R_MOON_M = 0.8, 0.8, 40.0, 1_737_400.0
clamp = lambda x, lo, hi: min(max(x, lo), hi)


def quat_from_body_z(direction):                                             # main.rs:100-121
    n = math.sqrt(sum(c * c for c in direction))
    d = [0.0, 0.0, 1.0] if n < 1e-9 else [c / n for c in direction]
    cross = [-d[1], d[0], 0.0]
    dot = clamp(d[2], -1.0, 1.0)
    if dot < -0.999_999:
        return [1.0, 0.0, 0.0, 0.0]
    q = [cross[0], cross[1], cross[2], 1.0 + dot]
    n = math.sqrt(sum(c * c for c in q))
    return [c / n for c in q]


def cap_tilt_preserve_magnitude(ax, ay, az, max_tilt_deg):                    # main.rs:127-142
    az = max(az, MIN_VERTICAL_ACCEL_MPS2)
    ah = math.hypot(ax, ay)
    if ah < 1e-9:
        return ax, ay, az
    max_tilt = math.radians(max_tilt_deg)
    if math.atan2(ah, az) <= max_tilt:
        return ax, ay, az
    mag = math.sqrt(ah * ah + az * az)
    scale_h = mag * math.sin(max_tilt) / ah
    return ax * scale_h, ay * scale_h, mag * math.cos(max_tilt)


def clamp_horizontal(ax, ay, vertical_accel, max_tilt_deg):                   # main.rs:147-156
    limit = max(vertical_accel, MIN_VERTICAL_ACCEL_MPS2) * math.tan(math.radians(max_tilt_deg))
    mag = math.hypot(ax, ay)
    if mag <= limit or mag < 1e-9:
        return ax, ay
    return ax * limit / mag, ay * limit / mag


class FakeBridge:
    """main.py:39-88 SitlBridge with the UDP round trip replaced by the controller's `command` (main.rs:188-262); the
    struct.pack / unpack of 20 + 6 little-endian f64 is value-preserving and omitted."""

    def __init__(self, throttle, attitude):
        self.ftp_latched = True                                              # main.rs:257: window opens at FTP
        self.last = (throttle, list(attitude), 0.0)

    def throttle_logic(self, demand):                                        # main.rs:163-186
        if self.ftp_latched and demand < 0.60:
            self.ftp_latched = False
        elif not self.ftp_latched and demand > 0.80:
            self.ftp_latched = True
        if demand <= EROSION_BAND_MIN and not self.ftp_latched:
            return max(demand, MIN_THROTTLE)
        return FTP_THROTTLE if self.ftp_latched else EROSION_BAND_MIN

    def step(self, s):
        vx, vy, _ = (float(x) for x in s["world_vel"])
        px, py = float(s["world_pos"][0]), float(s["world_pos"][1])
        altitude, vertical_speed, mass = float(s["altitude"]), float(s["vertical_speed"]), float(s["mass"])
        h_speed = math.hypot(vx, vy)
        g_eff = max(s["gravity"] - h_speed * h_speed / R_MOON_M, 0.05 * s["gravity"])
        rate_track = clamp(s["track_gain"] * (s["ref_alt"] - altitude), -RATE_TRACK_AUTHORITY_MPS, RATE_TRACK_AUTHORITY_MPS)
        rate_cmd = clamp(s["ref_rate"] + rate_track, -MAX_DESCENT_RATE_MPS, -MIN_DESCENT_RATE_MPS)
        vertical_fb = clamp(s["vertical_gain"] * (rate_cmd - vertical_speed), -VERTICAL_FB_AUTHORITY_MPS2, VERTICAL_FB_AUTHORITY_MPS2)
        vertical_accel = max(g_eff + vertical_fb, MIN_VERTICAL_ACCEL_MPS2)
        position_gain = 0.01 * s["horizontal_gain"]
        trim_fade = clamp((altitude - 30.0) / 120.0, 0.0, 1.0)
        trim_x = clamp(position_gain * (s["ref_downrange"] - px), -POSITION_AUTHORITY_MPS2, POSITION_AUTHORITY_MPS2) * trim_fade
        trim_y = clamp(position_gain * (-py), -POSITION_AUTHORITY_MPS2, POSITION_AUTHORITY_MPS2) * trim_fade
        target_vx, target_decel = (0.0, 0.0) if altitude < TERMINAL_NULL_ALT_M else (s["ref_hspeed"], s["ref_hdecel"])
        hspeed_fb = clamp(HSPEED_GAIN * (target_vx - vx), -HSPEED_FB_AUTHORITY_MPS2, HSPEED_FB_AUTHORITY_MPS2)
        ax = -target_decel + hspeed_fb + trim_x
        ay = clamp(HSPEED_GAIN * (-vy), -HSPEED_FB_AUTHORITY_MPS2, HSPEED_FB_AUTHORITY_MPS2) + trim_y
        blend = clamp((h_speed - TILT_BLEND_LO_MPS) / (TILT_BLEND_HI_MPS - TILT_BLEND_LO_MPS), 0.0, 1.0)
        max_tilt_deg = MAX_TILT_APPROACH_DEG + (MAX_TILT_BRAKING_DEG - MAX_TILT_APPROACH_DEG) * blend
        if h_speed > TILT_BLEND_LO_MPS:
            ax, ay, vertical_accel = cap_tilt_preserve_magnitude(ax, ay, vertical_accel, max_tilt_deg)
        else:
            ax, ay = clamp_horizontal(ax, ay, vertical_accel, MAX_TILT_APPROACH_DEG)
        thrust_required = mass * math.sqrt(ax * ax + ay * ay + vertical_accel * vertical_accel)
        demand = clamp(thrust_required / max(s["max_thrust"] * s["thrust_scale"], 1.0), MIN_THROTTLE, FTP_THROTTLE)
        throttle = self.throttle_logic(demand)
        self.last = (throttle, quat_from_body_z([ax, ay, vertical_accel]), rate_cmd)
        return self.last


# ---- the reference's systems chained per tick, batched like the server loop ------------------------------------------------

class Ctx:
    """el.StepContext stand-in over the lander's component dict (step_context.rs)."""

    def __init__(self, state):
        self.s = state

    def component_batch_operation(self, reads=None, writes=None):
        out = {}
        for key in reads or ():
            name = key.split(".", 1)[1]
            v = self.s[name]
            out[key] = v.asarray() if hasattr(v, "asarray") else (v.vector() if isinstance(v, el.Quaternion) else np.asarray(v, dtype=np.float64))
        for key, v in (writes or {}).items():
            name = key.split(".", 1)[1]
            self.s[name] = el.Quaternion(v) if name == "attitude_setpoint" else jnp.asarray(v)
        return out


def snapshot(s, bridge_latched, main):
    flat = lambda v: (v.asarray() if hasattr(v, "asarray") else (v.vector() if isinstance(v, el.Quaternion) else np.asarray(v, dtype=np.float64))).reshape(-1).tolist()
    keep = ("world_pos", "world_vel", "inertia", "throttle", "throttle_cmd", "attitude_setpoint", "propellant", "rcs_propellant",
            "thrust", "rcs_torque", "landed", "touchdown_speed", "touchdown_horizontal_speed", "altitude", "vertical_speed",
            "horizontal_speed", "pitch")
    out = {k: flat(s[k]) for k in keep}
    out["guidance"] = [float(main.last_throttle)] + [float(x) for x in main.last_attitude] + [float(main.last_rate_setpoint), 1.0 if bridge_latched else 0.0]
    out["score"] = [float(main.altitude_error_sum), float(main.pitch_error_sum), float(main.error_samples)]
    return out


def fly(params, stop_after_result_ticks=600):
    import os
    os.environ["ELODIN_MONTE_CARLO_CONTEXT"] = "refshim"      # main.py:256: the result record is only written inside a campaign
    el.monte_carlo.CONTEXT = dict(params)
    el.monte_carlo.RESULTS.clear()
    sys.modules.pop("main", None)
    main = importlib.import_module("main")                 # runs build(params) and the module body of main.py
    main.SitlBridge = FakeBridge
    s = dict(main.world.entities["lander"])
    pipe = main.system
    assert pipe.names() == ["truth_playback", "engine_response", "attitude_control", "mass_props", "thrust_visualization",
                            "six_dof", "ground_contact", "derive_telemetry"], pipe.names()
    six = pipe["six_dof"]
    assert [e.__name__ for e in six.effectors] == ["lunar_gravity", "apply_main_thrust", "apply_rcs_torque"] and six.time_step is None
    gravity, main_thrust, rcs_torque = six.effectors
    dt = orc.quantize_time_step(sim.SIMULATION_RATE_HZ)    # the globals time step six_dof reads (world_builder.rs:221)
    ctx, max_ticks = Ctx(s), main.max_ticks
    checkpoints, tick_counter, sim_tick, emitted_at = [], 0, 0, None
    while tick_counter < max_ticks:
        batch = max(1, min(TICKS_PER_TELEMETRY, max_ticks - tick_counter))              # impeller2_server.rs:557-558
        for _ in range(batch):
            sim_tick += 1
            s["throttle"], s["thrust"] = pipe["engine_response"](s["throttle"], s["throttle_cmd"], s["propellant"], s["landed"])
            s["rcs_torque"] = pipe["attitude_control"](s["world_pos"], s["world_vel"], s["attitude_setpoint"], s["landed"])
            s["propellant"], s["rcs_propellant"], s["inertia"] = pipe["mass_props"](s["thrust"], s["rcs_torque"], s["propellant"], s["rcs_propellant"], s["landed"])
            F = el.SpatialForce()                                                        # clear_forces, six_dof.rs:148-150
            F = gravity(F, s["inertia"], s["world_vel"])
            F = main_thrust(s["thrust"], F, s["world_pos"])
            F = rcs_torque(s["rcs_torque"], F, s["world_pos"])
            x, v = s["world_pos"].asarray(), s["world_vel"].asarray()
            a = orc.calc_accel(F.asarray(), s["inertia"].asarray(), x)                   # six_dof.rs:137-146
            v = v + dt * a                                                               # semi_implicit.rs:17-31
            x = orc.transform_add_motion(x, dt * v)
            s["world_pos"], s["world_vel"] = el.SpatialTransform(x), el.SpatialMotion(angular=v[:3], linear=v[3:])
            (s["world_pos"], s["world_vel"], s["landed"], s["touchdown_speed"], s["touchdown_horizontal_speed"]) = pipe["ground_contact"](
                s["world_pos"], s["world_vel"], s["landed"], s["touchdown_speed"], s["touchdown_horizontal_speed"])
            s["altitude"], s["vertical_speed"], s["horizontal_speed"], s["pitch"] = pipe["derive_telemetry"](s["world_pos"], s["world_vel"])
        end_tick = tick_counter + batch - 1                                              # impeller2_server.rs:566
        main.post_step(end_tick, ctx)                                                    # :671
        tick_counter += batch                                                            # :790-791
        latched = main.bridge.ftp_latched if main.bridge is not None else True
        if sim_tick % CHECKPOINT_EVERY == 0:
            checkpoints.append({"ticks_done": sim_tick, "state": snapshot(s, latched, main)})
        if emitted_at is None and el.monte_carlo.RESULTS:
            emitted_at = end_tick
            checkpoints.append({"ticks_done": sim_tick, "state": snapshot(s, latched, main)})
        if emitted_at is not None and end_tick - emitted_at >= stop_after_result_ticks:
            break
    assert len(el.monte_carlo.RESULTS) == 1, "exactly one result record per run"
    result = {k: (bool(v) if isinstance(v, (bool, np.bool_)) else float(v)) for k, v in el.monte_carlo.RESULTS[0].items()}
    return {"params": {k: float(v) for k, v in params.items()}, "max_ticks": int(max_ticks), "result": result,
            "result_end_tick": int(emitted_at), "ticks_flown": int(sim_tick), "checkpoints": checkpoints}


def plan_rows():
    with open(OUT / "plans" / "apollo.plan.csv", newline="") as f:      # the reference sampler's own output (make_plan_golden.py)
        rows = list(csv.DictReader(f))
    pick = {}
    for i in (0, 11, 23):
        pick[rows[i]["run_id"]] = {k[len("param."):]: float(v) for k, v in rows[i].items() if k.startswith("param.")}
    return pick


def main_():
    import tomli as tomllib
    nominal = tomllib.loads((REF / "examples/apollo-lander/spec.ci.toml").read_text())["monte_carlo"]["variables"]
    runs = {"nominal": {k: float(v["value"]) for k, v in nominal.items()}}     # spec.ci.toml: every variable fixed
    runs.update(plan_rows())
    out = {}
    for name, params in runs.items():
        r = fly(params)
        out[name] = r
        print(f"{name}: result at end_tick {r['result_end_tick']}: " + ", ".join(f"{k}={v:.6g}" if not isinstance(v, bool) else f"{k}={v}" for k, v in r["result"].items()))
    (OUT / "apollo_reference_runs.json").write_text(json.dumps(out, separators=(",", ":")))
    print("apollo_reference_runs.json:", (OUT / "apollo_reference_runs.json").stat().st_size, "bytes")


if __name__ == "__main__":
    main_()
