"""The Apollo-lander sim (examples/apollo-lander/sim.py:334-444,517-526) written against elodin_amd.dsl — the same
systems, in the same pipe order, as user code for the effector / system front-end (guidance commands held by the
caller: the reference delivers them from outside the tick, through `external_control` components)."""
import numpy as np

from elodin_amd import dsl

np_ = dsl.np
G0, R_MOON_M = 9.80665, 1_737_400.0
DPS_MAX_THRUST_N, DPS_MIN_THRUST_N = 45_040.0, 4_670.0
THROTTLE_MIN, THROTTLE_MAX = DPS_MIN_THRUST_N / DPS_MAX_THRUST_N, 1.0
RCS_ISP_S, RCS_MOMENT_ARM_M = 290.0, 2.0
RCS_AXIS_TORQUE_LIMIT_NM = 4.0 * 445.0 * RCS_MOMENT_ARM_M
FOOTPAD_HEIGHT_M, SIM_TIME_STEP = 2.40, 1.0 / 120.0
BASE_INERTIA = np_.array([78_000.0, 72_000.0, 45_000.0])
# per-rollout constants of build(params) (sim.py:233-256) live in one [n,8] component `cfg`:
# dry_mass, total_mass, thrust_scale, isp, attitude_gain/0.040, response_alpha, lunar_g, 0


@dsl.system
def engine_response(throttle, throttle_cmd, propellant, landed, cfg):
    cmd = np_.clip(throttle_cmd, THROTTLE_MIN, THROTTLE_MAX)
    actual = throttle + (cmd - throttle) * cfg[5]
    active = np_.logical_and(propellant > 0.0, landed < 0.5)
    actual = np_.where(active, actual, 0.0)
    return {"throttle": actual, "thrust": actual * DPS_MAX_THRUST_N * cfg[2]}


@dsl.system
def attitude_control(pos, vel, attitude_setpoint, landed, cfg):
    q_err = pos.angular().inverse() * dsl.Quaternion(attitude_setpoint)
    err = q_err.vector()
    sign = np_.where(err[3] >= 0.0, 1.0, -1.0)
    body_rate = pos.angular().inverse() @ vel.angular()
    rcs_k = np_.array([4_500.0, 5_500.0, 4_500.0]) * cfg[4]
    rcs_d = np_.array([19_000.0, 21_000.0, 19_000.0])
    torque = sign * err[:3] * rcs_k - body_rate * rcs_d
    torque = np_.clip(torque, -RCS_AXIS_TORQUE_LIMIT_NM, RCS_AXIS_TORQUE_LIMIT_NM)
    return {"rcs_torque": np_.where(landed > 0.5, np_.zeros(3), torque)}


@dsl.system
def mass_props(thrust, rcs_torque, propellant, rcs_propellant, landed, cfg):
    dps_burn = thrust / (cfg[3] * G0) * SIM_TIME_STEP
    rcs_force_equivalent = np_.sum(np_.abs(rcs_torque)) / RCS_MOMENT_ARM_M
    rcs_burn = rcs_force_equivalent / (RCS_ISP_S * G0) * SIM_TIME_STEP
    next_prop = np_.maximum(propellant - dps_burn, 0.0)
    next_rcs_prop = np_.maximum(rcs_propellant - rcs_burn, 0.0)
    mass = cfg[0] + next_prop + next_rcs_prop
    inertia_scale = mass / cfg[1]
    inertia = np_.where(landed > 0.5, np_.array([1.0e9, 1.0e9, 1.0e9]), BASE_INERTIA * inertia_scale)
    return {"propellant": next_prop, "rcs_propellant": next_rcs_prop, "inertia": dsl.SpatialInertia(inertia, mass)}


@dsl.effector
def lunar_gravity(force, inertia, vel, cfg):
    v_h_sq = np_.sum(vel.linear()[:2] ** 2)
    g_eff = np_.maximum(cfg[6] - v_h_sq / R_MOON_M, 0.0)
    return force + dsl.SpatialForce(linear=np_.array([0.0, 0.0, -1.0]) * g_eff * inertia.mass())


@dsl.effector
def apply_main_thrust(thrust, force, pos):
    return force + dsl.SpatialForce(linear=pos.angular() @ np_.array([0.0, 0.0, thrust[0]]))


@dsl.effector
def apply_rcs_torque(rcs_torque, force, pos):
    return force + dsl.SpatialForce(torque=pos.angular() @ rcs_torque)


@dsl.system
def ground_contact(pos, vel, landed, touchdown):
    altitude, vertical_speed = pos.linear()[2], vel.linear()[2]
    contact = altitude <= FOOTPAD_HEIGHT_M
    was_landed = landed > 0.5
    landed_now = np_.logical_or(was_landed, contact)
    first_contact = np_.logical_and(np_.logical_not(was_landed), contact)
    impact_speed = np_.where(first_contact, np_.abs(vertical_speed), touchdown[0])
    impact_horizontal = np_.where(first_contact, np_.linalg.norm(vel.linear()[:2]), touchdown[1])
    p = pos.linear()
    linear_pos = np_.where(landed_now, np_.array([p[0], p[1], FOOTPAD_HEIGHT_M]), p)
    return {"world_pos": dsl.SpatialTransform(pos.angular(), linear_pos),
            "world_vel": dsl.SpatialMotion(np_.where(landed_now, np_.zeros(3), vel.angular()),
                                           np_.where(landed_now, np_.zeros(3), vel.linear())),
            "landed": np_.where(landed_now, 1.0, 0.0), "touchdown": np_.array([impact_speed, impact_horizontal])}


@dsl.system
def derive_telemetry(pos):
    body_up = pos.angular() @ np_.array([0.0, 0.0, 1.0])
    return {"pitch": np_.rad2deg(np_.arccos(np_.clip(body_up[2], -1.0, 1.0)))}


NON_EFFECTORS = [engine_response, attitude_control, mass_props]
EFFECTORS = lunar_gravity | apply_main_thrust | apply_rcs_torque
POST = [ground_contact, derive_telemetry]


def components_from(cols):
    """Reference-named component columns from the packed columns of elodin_amd.models.apollo.initial_columns."""
    st, P = cols["apollo_state"], cols["apollo_params"]
    n = st.shape[0]
    total = P[:, 1] + P[:, 11] + P[:, 12]
    cfg = np.stack([P[:, 1], total, P[:, 14], P[:, 10], P[:, 0] / 0.040,
                    np.minimum(np.maximum(P[:, 13] * SIM_TIME_STEP, 0.0), 1.0), 1.622 * P[:, 2], np.zeros(n)], axis=1)
    return {"throttle": st[:, 0:1].copy(), "throttle_cmd": st[:, 1:2].copy(), "attitude_setpoint": st[:, 2:6].copy(),
            "propellant": st[:, 6:7].copy(), "rcs_propellant": st[:, 7:8].copy(), "thrust": st[:, 8:9].copy(),
            "rcs_torque": st[:, 9:12].copy(), "landed": st[:, 12:13].copy(), "touchdown": st[:, 13:15].copy(),
            "pitch": st[:, 15:16].copy(), "cfg": cfg}
