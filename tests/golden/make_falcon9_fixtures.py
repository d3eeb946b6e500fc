#!/usr/bin/env python3
"""Golden vectors for the Falcon 9 model from the REFERENCE's own code, executed here on numpy.

Run in the build container only (needs /root/reference):   python tests/golden/make_falcon9_fixtures.py

The reference example (examples/falcon9) is plain Python against jax.numpy + the elodin wheel; neither is installed, but
every physics function is ordinary array code, so under tests/golden/refshim.py (numpy in jax's clothes, spatial types
delegated to the pinned C oracle) the reference's modules import and run UNMODIFIED from /root/reference.  Two fixtures:

  falcon9_helpers.json   known answers of every physics helper the model restates — atmosphere.py, frames.py,
                         propulsion.py, aero.py, rcs.py — on seeded inputs spanning their domains.
  falcon9_plant.json     closed-plant trajectories: the reference's own @el.map systems of sim.py chained per tick in
                         build_powered's pipe order (sim.py:1433-1530) around a semi-implicit six_dof step done by the
                         pinned oracle (orc_calc_accel + orc_transform_add_motion, semi_implicit.rs:17-62), driven by
                         the open-loop command scripts of tests/falcon9_script.py; three 10 s windows (pad ignition +
                         release, a Max-Q-like powered window with wind / fins / TVC, an unpowered coast with RCS and
                         a three-engine relight), every column checkpointed every 500 ticks.

What this does NOT cover: the flight software (a Rust process, controller/src/main.rs — not executable here; the model's
restatement of its ascent phases stays unpinned), sensors, leg contact beyond "inactive during ascent".
"""
import sys as _sys
_sys.dont_write_bytecode = True      # the reference checkout is read-only: no __pycache__ next to what is imported from it
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
from tests import falcon9_script as fs  # noqa: E402
from tests.golden import refshim  # noqa: E402

jax, jnp, el = refshim.install(str(REF / "examples" / "falcon9"))
import aero  # noqa: E402
import atmosphere  # noqa: E402
import frames  # noqa: E402
import propulsion  # noqa: E402
import rcs  # noqa: E402
import sim  # noqa: E402


def L(x):
    """-> nested lists of python floats (repr round-trips f64 exactly)."""
    if isinstance(x, (tuple, list)):
        return [L(v) for v in x]
    a = np.asarray(x, dtype=np.float64)
    return float(a) if a.ndim == 0 else a.tolist()


# ---- part A: helper known answers -------------------------------------------------------------------------------------

def helper_cases():
    rng = np.random.default_rng(20170814)
    cases = {}

    def add(name, fn, arglist):
        cases[name] = [{"args": L(list(a)), "out": L(fn(*a))} for a in arglist]

    alts = np.concatenate([np.linspace(0.0, 120_000.0, 25), [10_999.0, 11_001.0, 20_000.0, 47_000.0, 84_852.0, 90_000.0]])
    add("geopotential_altitude", atmosphere.geopotential_altitude, [(h,) for h in alts])
    add("pressure_temperature_at_geopotential", atmosphere.pressure_temperature_at_geopotential,
        [(h,) for h in np.concatenate([alts, [-50.0, 260_000.0]])])
    add("pressure", atmosphere.pressure, [(h,) for h in alts])
    add("density", atmosphere.density, [(h,) for h in alts])
    add("speed_of_sound", atmosphere.speed_of_sound, [(h,) for h in alts])

    geod = [(math.radians(la), math.radians(lo), al) for la, lo, al in
            [(28.60839, -80.60433, 3.0), (0.0, 0.0, 0.0), (89.9, 10.0, 1000.0), (-45.0, 170.0, 80_000.0), (33.0, -75.0, 120_000.0)]]
    add("geodetic_to_ecef", frames.geodetic_to_ecef, geod)
    ecefs = [np.asarray(frames.geodetic_to_ecef(*g)) for g in geod] + [rng.normal(size=3) * 6.5e6 for _ in range(6)]
    add("ecef_to_geodetic", frames.ecef_to_geodetic, [(r,) for r in ecefs])
    add("ned_basis", frames.ned_basis, [(g[0], g[1]) for g in geod])
    add("gravity_accel", frames.gravity_accel, [(r,) for r in ecefs])
    vels = [rng.normal(size=3) * 1500.0 for _ in ecefs]
    add("frame_accel", frames.frame_accel, list(zip(ecefs, vels)))
    add("apparent_gravity", frames.apparent_gravity, [(r,) for r in ecefs])

    add("engine_thrust_per_engine", propulsion.engine_thrust_per_engine,
        [(u, p) for u in (0.0, 0.3, 0.57, 1.0) for p in (0.0, 2.0e4, 101_325.0, 2.0e6)])
    add("cluster_mdot", propulsion.cluster_mdot, [(n, u) for n in (0.0, 1.0, 3.0, 9.0) for u in (0.57, 1.0)])
    add("split_mdot", propulsion.split_mdot, [(m,) for m in (0.0, 271.3, 2555.0)])
    act = []
    for _ in range(12):
        x, cmd = rng.uniform(-0.2, 0.2, size=2)
        act.append((x, cmd, 0.001, float(rng.choice([0.015, 0.05, 0.15, 1.5]))))
    add("actuator_step", propulsion.actuator_step, act)
    add("actuator_step_limited", lambda x, c, dt, tau, rl, lo, hi: propulsion.actuator_step(x, c, dt, tau, rate_limit=rl, lo=lo, hi=hi),
        [(x, c, dt, tau, 0.35, -0.087, 0.087) for (x, c, dt, tau) in act])
    add("actuator_step_vec", lambda x, c, dt, tau: propulsion.actuator_step(jnp.asarray(x), jnp.asarray(c), dt, jnp.asarray(tau), lo=0.0, hi=1.0),
        [(rng.uniform(0, 1, 9), rng.uniform(0, 1, 9), 0.001, rng.choice([0.15, 1.5, 0.35], 9)) for _ in range(4)])
    props = [(rng.uniform(0, 287_000), rng.uniform(0, 123_000), float(rng.choice([0.0, 116_000.0]))) for _ in range(10)] + [(0.0, 0.0, 0.0)]
    add("stack_mass_props", propulsion.stack_mass_props, props)
    add("tank_pressure_step", propulsion.tank_pressure_step,
        [(rng.uniform(2e5, 4e5), rng.uniform(0, 287_000), rng.uniform(0, 1800), propulsion.V_TANK_LOX_M3, propulsion.RHO_LOX,
          float(rng.uniform(0, 1)), float(rng.uniform(0, 1)), 0.001) for _ in range(10)])
    add("inlet_pressure", propulsion.inlet_pressure,
        [(rng.uniform(2e5, 4e5), rng.uniform(0, 287_000), propulsion.RHO_LOX, propulsion.LOX_TANK_BOTTOM_M, rng.uniform(15, 25),
          rng.uniform(-5, 40), rng.uniform(0, 1800)) for _ in range(10)])

    add("config_blend", aero.config_blend, [(v,) for v in (-400.0, -50.0, -1.0, 0.0, 3.0, 50.0, 900.0)])
    add("plume_dominance", aero.plume_dominance, [(t, q) for t in (0.0, 8e5, 7.6e6) for q in (0.0, 5.0, 2.2e4, 6e4)])
    wrench_in = []
    for _ in range(16):
        v = rng.normal(size=3) * rng.choice([1.0, 80.0, 600.0])
        wrench_in.append((v, float(rng.uniform(0, 6)), float(rng.uniform(0, 5e4)), float(rng.uniform(15, 25)),
                          rng.normal(size=3) * 0.05, float(rng.uniform(0.8, 1.2)), float(rng.uniform(0.8, 1.2))))
    add("body_aero_wrench", lambda v, m, q, cg, om, ca, cn: aero.body_aero_wrench(jnp.asarray(v), m, q, cg, omega_body=jnp.asarray(om), ca_scale=ca, cn_scale=cn),
        wrench_in)
    add("fin_mix", lambda c: aero.fin_mix(jnp.asarray(c)), [(rng.uniform(-0.3, 0.3, 3),) for _ in range(5)])
    add("fin_wrench", lambda d, m, q, cg: aero.fin_wrench(jnp.asarray(d), m, q, cg),
        [(rng.uniform(-0.35, 0.35, 4), float(rng.uniform(0, 6)), float(rng.uniform(0, 5e4)), float(rng.uniform(15, 25))) for _ in range(8)])

    add("rcs_wrench", lambda lv, cg: rcs.rcs_wrench(jnp.asarray(lv), cg), [(rng.uniform(0, 1, 8), float(rng.uniform(15, 25))) for _ in range(6)])
    tq = [rng.normal(size=3) * s for s in (1.0, 50.0, 3e3, 2e4, 1e5) for _ in range(3)] + [np.zeros(3), np.array([0.0, 1e4, 0.0])]
    add("allocate_torque", lambda t, cg: rcs.allocate_torque(jnp.asarray(t), cg), [(t, float(rng.uniform(15, 25))) for t in tq])
    return cases


# ---- part B: the reference's plant systems chained per tick --------------------------------------------------------------

DT = sim.SIM_TIME_STEP
A1 = lambda v: jnp.array([float(v)])


def spawn(case):
    """build_powered's spawn (sim.py:1461-1510) -> dict of component values; windows aloft get a flight-like state."""
    c = fs.CASES[case]
    if c["aloft"] is None:
        r0, v0, att = np.asarray(sim.pad_ecef()), np.zeros(3), sim.upright_attitude()
        lox, rp1 = sim.LOX_LOAD_KG, sim.RP1_LOAD_KG
    else:
        alt, speed, pitch_deg, lox, rp1 = c["aloft"]
        lat, lon = math.radians(28.9), math.radians(-80.2)
        r0 = np.asarray(frames.geodetic_to_ecef(lat, lon, alt))
        ned = np.asarray(frames.ned_basis(lat, lon))
        az, pitch = math.radians(48.0), math.radians(pitch_deg)
        track = ned[0] * math.cos(az) + ned[1] * math.sin(az)
        direction = -ned[2] * math.cos(pitch) + track * math.sin(pitch)
        direction /= np.linalg.norm(direction)
        v0 = direction * speed
        x = np.array([1.0, 0.0, 0.0])                     # body +X along the velocity, as upright_attitude() builds it
        axis = np.cross(x, direction)
        att = el.Quaternion.from_axis_angle(axis / np.linalg.norm(axis), math.acos(float(np.clip(x @ direction, -1, 1))))
    mass0, cg0, inertia0 = propulsion.stack_mass_props(lox, rp1, c["upper_kg"])
    on_pad = float(np.linalg.norm(r0 - np.asarray(sim.pad_ecef()))) < 100.0
    s = dict(
        world_pos=el.SpatialTransform(angular=att, linear=r0), world_vel=el.SpatialMotion(linear=v0),
        inertia=el.SpatialInertia(float(mass0), inertia0),
        engine_cmd=jnp.zeros(sim.N_ENGINES), valve_cmd=jnp.zeros(sim.N_VALVES), engine_spool=jnp.zeros(sim.N_ENGINES),
        engine_armed=jnp.zeros(sim.N_ENGINES), teateb_charges=jnp.asarray(sim.INITIAL_TEATEB_CHARGES),
        valve_state=jnp.zeros(sim.N_VALVES), thrust_total=A1(0), mdot_total=A1(0), propellant_lox=A1(lox),
        propellant_rp1=A1(rp1), tank_pressure_lox=A1(sim.TANK_P_NOM_PA), tank_pressure_rp1=A1(sim.TANK_P_NOM_PA),
        inlet_pressure_lox=A1(sim.TANK_P_NOM_PA), inlet_pressure_rp1=A1(sim.TANK_P_NOM_PA),
        cg_station=A1(propulsion.DRY_CG_STATION_M), axial_specific_force=A1(0), wind_ecef=jnp.zeros(3), wind_gust_ned=jnp.zeros(3),
        qbar=A1(0), mach=A1(0), tvc_cmd=jnp.zeros(2), tvc_state=jnp.zeros(2), fin_cmd=jnp.zeros(3), fin_state=jnp.zeros(4),
        rcs_torque_cmd=jnp.zeros(3), rcs_levels=jnp.zeros(rcs.N_RCS), nitrogen_kg=A1(sim.N2_INITIAL_KG),
        aero_wrench=jnp.zeros(6), fin_wrench=jnp.zeros(6), rcs_wrench=jnp.zeros(6), engine_wrench=jnp.zeros(6), leg_wrench=jnp.zeros(6),
        attitude_setpoint=sim.upright_attitude(), ctrl_enable=jnp.zeros(2), fsw_phase=A1(0), landed=A1(0), deck_metrics=jnp.zeros(5),
        upper_mass=A1(c["upper_kg"]), lifted=A1(0.0 if on_pad else 1.0), liftoff_time=A1(0), touchdown_metrics=jnp.zeros(6),
        altitude_geodetic=A1(0), ground_speed=A1(0), world_accel=np.zeros(6), force=np.zeros(6))
    if c["engines_running"]:   # as after a nominal ignition: all nine armed and at full spool, feed valves open, one charge spent
        s["engine_spool"], s["engine_armed"] = jnp.ones(sim.N_ENGINES), jnp.ones(sim.N_ENGINES)
        s["teateb_charges"] = jnp.asarray(sim.INITIAL_TEATEB_CHARGES) - 1.0
        s["valve_state"] = jnp.array([1.0, 0.0, 1.0, 0.0, 1.0, 1.0, 0.0, 0.0])
        s["engine_cmd"] = jnp.ones(sim.N_ENGINES)
    return s, att


def flat(s):
    """component values -> {name: list of floats} in this repo's column spelling."""
    out = {}
    for k, v in s.items():
        if k in ("wind_gust_ned", "landed", "deck_metrics", "touchdown_metrics", "leg_wrench"):
            continue
        if hasattr(v, "asarray"):
            v = v.asarray()
        elif isinstance(v, el.Quaternion):
            v = v.vector()
        out[k] = L(np.asarray(v, dtype=np.float64).reshape(-1))
    return out


def plant_tick(s, tick, S=None):
    """ONE tick of build_powered's plant (sim.py:1433-1530) on the component dict `s`, in the reference's pipe order:
    propulsion_systems | six_dof(gravity_and_frame_forces | apply_body_wrenches, SemiImplicit) | pad_clamp |
    ground_contact | derive_geodetic_telemetry.  `S` maps a system name to the callable to use for it (the closures
    make_engine_dynamics / make_wind_model / make_aero_dynamics build per rollout); everything else is sim.<name>."""
    S = S or {}
    g = lambda name: S.get(name) or getattr(sim, name)
    s["tvc_cmd"], s["rcs_torque_cmd"] = g("attitude_control")(s["world_pos"], s["world_vel"], s["attitude_setpoint"], s["ctrl_enable"],
                                                              s["inertia"], s["thrust_total"], s["cg_station"], s["fsw_phase"])
    s["valve_state"] = g("valve_dynamics")(s["valve_state"], s["valve_cmd"])
    s["tvc_state"] = g("tvc_actuators")(s["tvc_state"], s["tvc_cmd"])
    s["fin_state"] = g("fin_actuators")(s["fin_state"], s["fin_cmd"])
    (s["engine_spool"], s["engine_armed"], s["teateb_charges"], s["thrust_total"], s["mdot_total"]) = g("engine_dynamics")(
        s["world_pos"], s["engine_cmd"], s["engine_spool"], s["engine_armed"], s["teateb_charges"], s["valve_state"],
        s["propellant_lox"], s["propellant_rp1"])
    (s["propellant_lox"], s["propellant_rp1"], s["inertia"], s["cg_station"], s["axial_specific_force"]) = g("mass_props")(
        s["mdot_total"], s["propellant_lox"], s["propellant_rp1"], s["thrust_total"], s["upper_mass"])
    (s["tank_pressure_lox"], s["tank_pressure_rp1"], s["inlet_pressure_lox"], s["inlet_pressure_rp1"]) = g("tank_dynamics")(
        s["tank_pressure_lox"], s["tank_pressure_rp1"], s["propellant_lox"], s["propellant_rp1"], s["mdot_total"],
        s["valve_state"], s["axial_specific_force"], s["cg_station"])
    s["rcs_levels"], s["rcs_wrench"], s["nitrogen_kg"] = g("rcs_dynamics")(s["rcs_levels"], s["rcs_torque_cmd"], s["cg_station"], s["nitrogen_kg"])
    s["engine_wrench"] = g("engine_wrench")(s["thrust_total"], s["tvc_state"], s["cg_station"])
    s["leg_wrench"] = g("leg_contact_wrench")(s["world_pos"], s["world_vel"], s["cg_station"], s["lifted"], s["landed"])
    assert not np.any(np.asarray(s["leg_wrench"])), "legs must be inactive in these windows"
    s["wind_ecef"], s["wind_gust_ned"] = g("wind_model")(s["world_pos"], s["wind_ecef"], s["wind_gust_ned"], A1(tick))
    s["qbar"], s["mach"], s["aero_wrench"], s["fin_wrench"] = g("aero_dynamics")(s["world_pos"], s["world_vel"], s["wind_ecef"],
                                                                                 s["thrust_total"], s["fin_state"], s["cg_station"])
    # six_dof(sys = gravity_and_frame_forces | apply_body_wrenches, SemiImplicit): six_dof.rs:137-150,176-180
    F = el.SpatialForce()
    F = g("gravity_and_frame_forces")(F, s["inertia"], s["world_pos"], s["world_vel"])
    F = g("apply_body_wrenches")(s["engine_wrench"], s["aero_wrench"], s["fin_wrench"], s["rcs_wrench"], s["leg_wrench"], F, s["world_pos"])
    x, v = s["world_pos"].asarray(), s["world_vel"].asarray()
    a = orc.calc_accel(F.asarray(), s["inertia"].asarray(), x)
    v = v + DT * a                                                      # semi_implicit.rs:17-31
    x = orc.transform_add_motion(x, DT * v)
    s["world_pos"], s["world_vel"] = el.SpatialTransform(x), el.SpatialMotion(angular=v[:3], linear=v[3:])
    s["world_accel"], s["force"] = a, F.asarray()
    # pad_clamp | ground_contact | derive_geodetic_telemetry, sim.py:1511-1530
    s["world_pos"], s["world_vel"], s["lifted"], s["liftoff_time"] = g("pad_clamp")(
        refshim._Query(float(tick)), refshim._Query(s["world_pos"], s["world_vel"], s["lifted"], s["liftoff_time"], s["thrust_total"], s["inertia"]))
    gp, gv, landed, tm, dm = g("ground_contact")(s["world_pos"], s["world_vel"], s["landed"], s["touchdown_metrics"], s["deck_metrics"],
                                                 s["lifted"], s["tvc_state"], s["cg_station"])
    assert float(np.asarray(landed).reshape(-1)[0]) == 0.0 and np.array_equal(gp.asarray(), s["world_pos"].asarray()) \
        and np.array_equal(gv.asarray(), s["world_vel"].asarray()), "ground contact must be a no-op in these windows"
    s["altitude_geodetic"], s["ground_speed"] = g("derive_geodetic_telemetry")(s["world_pos"], s["world_vel"])


def run_case(case):
    c = fs.CASES[case]
    s, att0 = spawn(case)
    script = fs.make_script(case, att0.vector())
    S = dict(engine_dynamics=sim.make_engine_dynamics(c["thrust_scale"], c["isp_scale"]), wind_model=sim.make_wind_model(*c["wind_ned"], 0.0),
             aero_dynamics=sim.make_aero_dynamics(c["ca_scale"], c["cn_scale"]))
    init = flat(s)
    checkpoints = []
    for tick in range(1, c["ticks"] + 1):
        t = tick * 0.001                                                     # test_propulsion.py:113-122 `_script`
        cmd = script(jnp, t)
        for k, v in cmd.items():
            s[k] = el.Quaternion(v) if k == "attitude_setpoint" else jnp.asarray(v)
        plant_tick(s, tick, S)
        if tick % fs.CHECKPOINT_EVERY == 0 or tick in (1, 2, 10):
            checkpoints.append({"tick": tick, "state": flat(s)})
    print(f"  {case}: {c['ticks']} ticks, final alt {float(s['altitude_geodetic'][0]):.1f} m, speed {float(s['ground_speed'][0]):.2f} m/s, "
          f"thrust {float(s['thrust_total'][0]):.0f} N, qbar {float(s['qbar'][0]):.0f} Pa, N2 {float(s['nitrogen_kg'][0]):.2f} kg")
    return {"config": {k: L(v) if not isinstance(v, (bool, type(None))) else v for k, v in c.items()}, "init": init,
            "base_attitude": L(att0.vector()), "checkpoints": checkpoints}


def main():
    helpers = helper_cases()
    (OUT / "falcon9_helpers.json").write_text(json.dumps(helpers, indent=None, separators=(",", ":")))
    print(f"falcon9_helpers.json: {sum(len(v) for v in helpers.values())} cases of {len(helpers)} functions")
    plant = {case: run_case(case) for case in fs.CASES}
    (OUT / "falcon9_plant.json").write_text(json.dumps(plant, indent=None, separators=(",", ":")))
    print("falcon9_plant.json:", (OUT / "falcon9_plant.json").stat().st_size, "bytes")


if __name__ == "__main__":
    main()
