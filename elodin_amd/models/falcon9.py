"""Falcon 9 booster ascent Monte-Carlo (BASELINE config 5) as USER CODE on the generated-program path.

The reference's example (examples/falcon9) is a 1 kHz plant in rotating WGS84 ECEF — point-mass gravity + Coriolis /
centrifugal, nine Merlin engines with ignition gating and spool dynamics, TVC, propellant depletion with a cylinder-stack
mass model, US Standard Atmosphere 1976, Mach-tabulated body aerodynamics with plume dominance, a hold-down clamp —
integrated by `el.six_dof(integrator=SemiImplicit)` and flown by an external flight-software process (Rust,
examples/falcon9/controller) exchanging sensor/command packets every 10 ticks.

This module restates the ASCENT part of that stack (pad -> vertical rise -> pitch kick -> gravity turn -> MECO) against
`elodin_amd.dsl`, so that the whole closed loop is traced, generated and fused into the step kernel
(`pre | six_dof(effectors) | post`), one lane per Monte-Carlo rollout:

  plant systems   sim.py:350-733 (attitude_control, valve_dynamics, tvc_actuators, fin_actuators, engine_dynamics,
                  mass_props, tank_dynamics, rcs_dynamics, engine_wrench, wind_model (steady part), aero_dynamics incl.
                  grid fins, gravity_and_frame_forces, apply_body_wrenches), pad_clamp sim.py:984-1013,
                  derive_geodetic_telemetry sim.py:1128-1143
  physics helpers frames.py:29-114, atmosphere.py:25-90, propulsion.py:46-149, aero.py:17-140, rcs.py:26-107
  sensors         sensors.py:23-120, sim.py:1016-1110 (IMU, 25 Hz GPS with the plume blackout, 40 Hz radar altimeter, pressure
                  transducers; noise = jax.random threefry keyed by (20170814, salt, sample counter), drawn in the kernel)
  flight software controller/src/main.rs:213-327 (IMU + GPS + radar navigator in the rotating frame), :384-533 (phases
                  PadPress..Meco, flying the recorded CRS-12 profile: flight-path-angle feed-forward with an altitude trim,
                  speed closed by throttle, Max-Q bucket, 3.5 g limit), profile.rs (the profile's resampling), math.rs; called
                  on main.py's cadence: after ticks 1, 11, 21, ... with t = (ticks done - 1) dt (impeller2_server.rs:553-678)
  parameters      spec.toml (LHS, seed 20170814) / main.py:53-100 calibrated defaults

Parity: the whole closed loop is pinned on flights flown by the REFERENCE's own plant, sensors and post_step bridge (imported
unmodified) with an independent C restatement of the Rust flight software (oracle/falcon9_fsw.c): five ascents, pad to MECO + 3 s
— every phase transition on the same tick, every component and the navigator's private state within 1.5e-10 of its scale at 133
checkpoints through the generated f64 kernel (tests/test_gpu_falcon9_closed_loop.py, profiles/r03_falcon9_closed_loop.txt;
tests/test_falcon9_closed_loop_reference.py walks the traced program through the first 3 s on the CPU: 4e-15).  Helpers and the
open-loop plant are pinned separately (tests/test_falcon9_reference_fixtures.py, test_falcon9_plant_reference.py) and the
reference's own verification ladder is reproduced (tests/test_falcon9_host.py, tests/test_gpu_falcon9.py).

Deliberate scope limits (stated, not hidden): the recovery half of the mission (flip, boostback, entry, landing) is not
flown — the state machine stops at Meco -> Flip (3 s after cutoff), its guidance phases, the landing-leg contact model and
ground contact are not built, and fins / RCS, whose plant models ARE here, stay at rest because the ascent flight software
never commands them (RCS attitude hold runs in the Meco phase).  The navigator's wind estimate (main.rs:272, read by the
landing burn only) is not carried.  The wind model carries its steady part only (per-rollout `wind_ned`, zero in spec.toml),
not the gust process (gust_sigma_mps = 0 in the ascent spec.toml; spec.landing.toml, i.e. the recovery half, turns it on).
The IMU and the pressure transducers are sampled on the guidance-exchange ticks only (the only ticks their samples are
consumed on; their noise is keyed by the tick, so those samples equal the reference's).

Every physics helper takes the array namespace `xp` first: `numpy` gives the host-side f64 evaluation (initial
conditions, known-answer tests), `elodin_amd.dsl.np` traces the same code into the kernel.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import numpy as np

from .. import dsl

# ---- constants (constants.py; EST = public estimate / calibration prior) ------------------------------------------------
WGS84_A_M = 6_378_137.0
WGS84_F = 1.0 / 298.257223563
WGS84_B_M = WGS84_A_M * (1.0 - WGS84_F)
WGS84_E2 = WGS84_F * (2.0 - WGS84_F)
WGS84_EP2 = WGS84_E2 / (1.0 - WGS84_E2)
MU_EARTH_M3S2 = 3.986004418e14
OMEGA_EARTH_RADPS = 7.292115e-5
G0 = 9.80665

SIM_RATE_HZ = 1000.0
SIM_TIME_STEP = 1.0 / SIM_RATE_HZ
GUIDANCE_PERIOD_TICKS = 10                       # main.py:187 (GUIDANCE_RATE_HZ = 100)

PAD_LAT_DEG, PAD_LON_DEG, PAD_ALT_M = 28.60839, -80.60433, 3.0

STAGE1_LENGTH_M = 47.0
STAGE1_DIAMETER_M = 3.66
S_REF_M2 = math.pi * STAGE1_DIAMETER_M ** 2 / 4.0
STAGE1_DRY_MASS_KG = 25_600.0
STAGE1_PROP_KG = 398_000.0
OF_RATIO = 2.33
LOX_LOAD_KG = STAGE1_PROP_KG * OF_RATIO / (1.0 + OF_RATIO)
RP1_LOAD_KG = STAGE1_PROP_KG / (1.0 + OF_RATIO)
STAGE2_WET_KG = 111_500.0
PAYLOAD_KG = 7_100.0
UPPER_KG = STAGE2_WET_KG + PAYLOAD_KG             # main.py:165
LIFTOFF_MASS_KG = STAGE1_DRY_MASS_KG + STAGE1_PROP_KG + STAGE2_WET_KG + PAYLOAD_KG

N_ENGINES = 9
ENGINE_A_E_M2 = 0.681
ENGINE_T_SL_N = 760e3
P_SL_PA = 101_325.0
ENGINE_T_VAC_N = ENGINE_T_SL_N + P_SL_PA * ENGINE_A_E_M2
ENGINE_ISP_SL_S = 282.0
ENGINE_ISP_VAC_S = ENGINE_ISP_SL_S * ENGINE_T_VAC_N / ENGINE_T_SL_N
THROTTLE_MIN = 0.57
RELIGHT_CAPABLE_ENGINES = 3
ENGINE_SPINUP_TAU_S, ENGINE_SHUTDOWN_TAU_S, ENGINE_THROTTLE_TAU_S = 1.5, 0.35, 0.15
TVC_MAX_RAD = math.radians(5.0)
TVC_RATE_RADPS = math.radians(20.0)
TVC_TAU_S = 0.030
TANK_P_NOM_PA = 3.5e5
VALVE_TAU_S = 0.015

(VALVE_HE_INFILL_LOX, VALVE_HE_VENT_LOX, VALVE_HE_INFILL_RP1, VALVE_HE_VENT_RP1, VALVE_MAIN_LOX, VALVE_MAIN_RP1,
 VALVE_TEATEB, VALVE_N2_PURGE) = range(8)
N_VALVES = 8

# propulsion.py:26-45
DRY_CG_STATION_M = 18.8
RHO_LOX, RHO_RP1 = 1220.0, 830.0
TANK_AREA_M2 = S_REF_M2
RP1_TANK_BOTTOM_M, LOX_TANK_BOTTOM_M = 3.0, 17.5
TANK_ULLAGE_FRAC = 0.05
V_TANK_LOX_M3 = LOX_LOAD_KG / RHO_LOX * (1.0 + TANK_ULLAGE_FRAC)
V_TANK_RP1_M3 = RP1_LOAD_KG / RHO_RP1 * (1.0 + TANK_ULLAGE_FRAC)
STAGE_RADIUS_M = 1.83
P_REGULATOR_PA = TANK_P_NOM_PA + 0.2e5
K_INFILL_PER_S, K_VENT_PER_S, P_AMBIENT_MIN_PA = 0.5, 0.3, 1.0e4
STAGE2_CG_STATION_M, STAGE2_LENGTH_M = 58.0, 16.0

# aero.py:17-50
MACH_PTS = (0.0, 0.6, 0.9, 1.1, 1.5, 2.0, 3.0, 5.0, 10.0)
CA_ASCENT = (0.30, 0.32, 0.45, 0.55, 0.50, 0.42, 0.35, 0.30, 0.28)
CA_DESCENT = (1.90, 1.95, 2.10, 2.40, 2.30, 2.20, 2.10, 2.00, 1.90)
CN_CROSS = (1.20, 1.20, 1.25, 1.35, 1.30, 1.25, 1.20, 1.15, 1.10)
X_CP_ASCENT_M, X_CP_DESCENT_M = 28.0, 26.0
CMQ_ASCENT, CMQ_DESCENT = -2.5, -12.0
L_REF_DAMP_M = STAGE1_LENGTH_M
PLUME_CT0 = 1.0

# cold-gas RCS (constants.py:76-81, rcs.py:26-57) and grid fins (constants.py:83-87, aero.py:50-73)
RCS_THRUST_PER_THRUSTER_N, RCS_VALVE_TAU_S, RCS_STATION_M = 7_500.0, 0.007, 46.0
N_RCS = 8
_RCS_R = STAGE1_DIAMETER_M / 2.0
RCS_POS = tuple((RCS_STATION_M, y, 0.0) for y in (+_RCS_R, +_RCS_R, -_RCS_R, -_RCS_R, +_RCS_R, +_RCS_R, -_RCS_R, -_RCS_R))
RCS_FORCE_DIR = ((0.0, 0.0, 1.0), (0.0, 0.0, -1.0), (0.0, 0.0, 1.0), (0.0, 0.0, -1.0),
                 (0.0, 1.0, 0.0), (0.0, -1.0, 0.0), (0.0, -1.0, 0.0), (0.0, 1.0, 0.0))
RCS_AXIS_GROUPS = ((0, (0, 3), (1, 2)), (1, (1, 3), (0, 2)), (2, (4, 7), (5, 6)))
N2_INITIAL_KG, N2_ISP_S = 800.0, 70.0                     # sim.py:1381, :347
FIN_MAX_RAD, FIN_RATE_RADPS, FIN_TAU_S = math.radians(20.0), math.radians(20.0), 0.050
FIN_STATION_M, S_FIN_M2 = 44.0, 1.5
CN_DELTA_FIN = (1.2, 1.2, 0.9, 0.8, 1.1, 1.3, 1.25, 1.2, 1.1)
_FIN_AZ = tuple(math.radians(a) for a in (45.0, 135.0, 225.0, 315.0))
FIN_FORCE_DIR = tuple((0.0, -math.sin(a), math.cos(a)) for a in _FIN_AZ)
FIN_POS = tuple((FIN_STATION_M, 1.83 * math.cos(a), 1.83 * math.sin(a)) for a in _FIN_AZ)

# sim.py:640-646 attitude inner loop
ATT_WN_TVC, ATT_WN_TVC_LANDING, ATT_ZETA_TVC, ATT_WN_RCS, ATT_ZETA_RCS = 0.9, 1.7, 0.9, 0.35, 0.8

# atmosphere.py:19-47
R_STAR, M_AIR = 8.31432, 28.9644e-3
R_AIR = R_STAR / M_AIR
GMR = G0 * M_AIR / R_STAR
GAMMA = 1.4
R0_GEOPOT_M = 6_356_766.0
_H_B = (0.0, 11_000.0, 20_000.0, 32_000.0, 47_000.0, 51_000.0, 71_000.0, 84_852.0)
_T_B = (288.15, 216.65, 216.65, 228.65, 270.65, 270.65, 214.65, 186.946)
_L_B = (-6.5e-3, 0.0, 1.0e-3, 2.8e-3, 0.0, -2.8e-3, -2.0e-3, 0.0)


def _base_pressures():
    p = [P_SL_PA]
    for i in range(1, len(_H_B)):
        dh, t_b, lapse = _H_B[i] - _H_B[i - 1], _T_B[i - 1], _L_B[i - 1]
        p.append(p[-1] * math.exp(-GMR * dh / t_b) if lapse == 0.0 else p[-1] * (t_b / (t_b + lapse * dh)) ** (GMR / lapse))
    return tuple(p)


_P_B = _base_pressures()

# per-rollout parameter column (`params`, [n,16]); the first 14 are read inside the loop
PARAM_NAMES = ["thrust_scale", "isp_scale", "ca_scale", "cn_scale", "kick_deg", "kick_start_s", "kick_ramp_s",
               "ascent_throttle", "bucket_throttle", "bucket_q_on_pa", "meco_fpa_deg", "pitch_exp", "meco_speed_mps",
               "azimuth_deg", "lox_kg", "rp1_kg"]
P = {name: k for k, name in enumerate(PARAM_NAMES)}
# main.py:53-100: the calibrated best fit against the recorded CRS-12 flight
DEFAULT_PARAMS = dict(thrust_scale=1.0323, isp_scale=1.0215, ca_scale=0.9574, cn_scale=1.3038, kick_deg=6.17,
                      kick_start_s=7.81, kick_ramp_s=11.74, ascent_throttle=0.9969, bucket_throttle=0.7105,
                      bucket_q_on_pa=18_942.0, meco_fpa_deg=35.27, pitch_exp=0.5626, meco_speed_mps=1_645.1,
                      azimuth_deg=47.67, lox_kg=275_357.0, rp1_kg=120_449.0)
# spec.toml [monte_carlo.variables] (all uniform).  The whole table is sampled — the LHS draws depend on the variable
# set — and the columns that reach the ascent are kept.
SPEC_RANGES = dict(lox_kg=(272000.0, 285000.0), rp1_kg=(116000.0, 123000.0), thrust_scale=(0.98, 1.14),
                   isp_scale=(0.97, 1.03), ca_scale=(0.6, 1.5), cn_scale=(0.6, 1.5), display_lag_s=(0.0, 2.5),
                   kick_deg=(2.5, 6.5), kick_start_s=(6.0, 12.0), kick_ramp_s=(5.0, 12.0), ascent_throttle=(0.97, 1.0),
                   bucket_throttle=(0.62, 0.85), bucket_q_on_pa=(14000.0, 24000.0), meco_fpa_deg=(34.0, 46.0),
                   pitch_exp=(0.4, 0.75), meco_speed_mps=(1630.0, 1680.0), azimuth_deg=(40.0, 50.0),
                   boostback_overshoot_m=(-2500.0, -500.0), entry_ignite_speed_mps=(1220.0, 1380.0),
                   entry_ignite_alt_m=(46000.0, 56000.0), entry_dv_mps=(320.0, 430.0), entry_throttle=(0.57, 0.8),
                   landing_arm_alt_m=(5000.0, 8000.0), landing_accel_margin=(1.05, 1.45), fsw_cd_s_m2=(15.0, 45.0),
                   boostback_throttle=(0.57, 0.85), fin_wn=(0.9, 2.2), divert_speed_cap=(22.0, 45.0),
                   steer_tilt_cap=(0.14, 0.24))
SPEC_SEED = 20170814

PHASE_PAD_PRESS, PHASE_VERTICAL_RISE, PHASE_PITCH_KICK, PHASE_GRAVITY_TURN, PHASE_MECO, PHASE_FLIP = 0.0, 1.0, 2.0, 3.0, 4.0, 5.0

# sensors.py:23-40 (error model EST; noise keyed by jax.random.key(20170814), salt, sample counter)
SENSOR_SEED = 20170814
GPS_DT_S, RADAR_DT_S = 1.0 / 25.0, 1.0 / 40.0                     # constants.py GPS_RATE_HZ / ALTIMETER_RATE_HZ
IMU_ACCEL_SIGMA, IMU_GYRO_SIGMA, GPS_POS_SIGMA, GPS_VEL_SIGMA, PRESSURE_SIGMA_PA = 0.02, 1.0e-3, 1.5, 0.05, 1.0e3
RADAR_MAX_RANGE_M, RADAR_FOV_COS, RADAR_SIGMA_M = 500.0, math.cos(math.radians(35.0)), 0.15
BLACKOUT_MACH_MIN, BLACKOUT_THRUST_MIN_N = 2.5, 1.0e5
METRIC_NAMES = ["max_qbar_pa", "t_max_qbar_s", "max_accel_mps2", "meco_t_s", "meco_alt_m", "meco_speed_mps",
                "meco_fpa_deg", "meco_downrange_m"]


# ---- geodesy and the rotating frame (frames.py) -------------------------------------------------------------------------

def geodetic_to_ecef(xp, lat, lon, alt):                                   # frames.py:29-40
    s, c = xp.sin(lat), xp.cos(lat)
    n = WGS84_A_M / xp.sqrt(1.0 - WGS84_E2 * s ** 2)
    return xp.array([(n + alt) * c * xp.cos(lon), (n + alt) * c * xp.sin(lon), (n * (1.0 - WGS84_E2) + alt) * s])


def ecef_to_geodetic(xp, r):                                               # frames.py:43-66: Bowring, 4 fixed iterations
    x, y, z = r[0], r[1], r[2]
    lon = xp.arctan2(y, x)
    p = xp.hypot(x, y)
    beta = xp.arctan2(z, (1.0 - WGS84_F) * p)
    lat = beta
    for _ in range(4):
        lat = xp.arctan2(z + WGS84_EP2 * WGS84_B_M * xp.sin(beta) ** 3, p - WGS84_E2 * WGS84_A_M * xp.cos(beta) ** 3)
        beta = xp.arctan((1.0 - WGS84_F) * xp.tan(lat))
    s = xp.sin(lat)
    w = xp.sqrt(1.0 - WGS84_E2 * s ** 2)
    alt = p * xp.cos(lat) + z * s - WGS84_A_M * w
    return lat, lon, alt


GEODETIC_SINCOS_PASSES = 2


def geodetic_sincos(xp, r, passes: int = GEODETIC_SINCOS_PASSES):
    """ecef_to_geodetic without a single angle: the same Bowring recurrence (frames.py:43-66) carried on tan(beta) —
    tan(lat) = (z + e'^2 b sin^3 beta) / (p - e^2 a cos^3 beta), tan(beta) = (1 - f) tan(lat), sin / cos of an angle in
    (-pi/2, pi/2) from its tangent by 1 / sqrt(1 + t^2) — returning (sin lat, cos lat, sin lon, cos lon, alt), which is all
    the plant ever asks of a latitude or a longitude.  Algebraically the reference's function (identical in exact arithmetic
    for p > 0, i.e. off the polar axis); ~50 instructions where the ten arctan / tan / sin / cos round trips of the angle
    form are ~400.  The reference makes 4 fixed passes; this makes `passes` = 2: the recurrence converges quadratically from
    Bowring's starting value, and from -100 m to 400 km altitude at every latitude the second pass already leaves sin lat
    within 2.2e-16 and the altitude within the 3e-9 m cancellation noise of the 4-pass value in FLOAT64 (in float32 two passes
    are as close to four as three are) — tests/test_falcon9_host.py.  The f32 campaign builds use it
    (build_program(algebraic_geodesy=True)); the f64 parity builds keep the reference's angle arithmetic operation for
    operation, four passes included."""
    x, y, z = r[0], r[1], r[2]
    p = xp.hypot(x, y)
    k = 1.0 - WGS84_F
    t = z / (k * p)
    for _ in range(passes):
        cb = 1.0 / xp.sqrt(1.0 + t * t)
        sb = t * cb
        num = z + WGS84_EP2 * WGS84_B_M * sb ** 3
        den = p - WGS84_E2 * WGS84_A_M * cb ** 3
        t = k * num / den
    h = xp.hypot(num, den)
    sl, cl = num / h, den / h
    alt = p * cl + z * sl - WGS84_A_M * xp.sqrt(1.0 - WGS84_E2 * sl ** 2)
    return sl, cl, y / p, x / p, alt


def ned_rows(xp, sl, cl, so, co):                                          # frames.py:74-84 from the sines and cosines
    return (xp.array([-sl * co, -sl * so, cl]), xp.array([-so, co, 0.0]), xp.array([-cl * co, -cl * so, -sl]))


def ned_basis(xp, lat, lon):                                               # frames.py:74-84: rows north, east, down
    return ned_rows(xp, xp.sin(lat), xp.cos(lat), xp.sin(lon), xp.cos(lon))


def gravity_accel(xp, r):                                                  # frames.py:92-95
    rn = xp.linalg.norm(r)
    return -MU_EARTH_M3S2 * r / rn ** 3


def frame_accel(xp, r, v):                                                 # frames.py:98-111: Coriolis + centrifugal
    om = xp.array([0.0, 0.0, OMEGA_EARTH_RADPS])
    return -2.0 * xp.cross(om, v) + -xp.cross(om, xp.cross(om, r))


def apparent_gravity(xp, r):                                               # frames.py:113-114
    om = xp.array([0.0, 0.0, OMEGA_EARTH_RADPS])
    return gravity_accel(xp, r) + -xp.cross(om, xp.cross(om, r))


# ---- US Standard Atmosphere 1976 (atmosphere.py) -------------------------------------------------------------------------

def pressure_temperature_at_geopotential(xp, h_geopot):                    # atmosphere.py:55-69
    h = xp.clip(h_geopot, 0.0, 250_000.0)
    # searchsorted(H_B, h, 'right') - 1 as a select chain over the constant layer table
    t_b, lapse, p_b, h_b = _T_B[0], _L_B[0], _P_B[0], _H_B[0]
    iso, expo = 0.0, GMR / _L_B[0]
    for i in range(1, len(_H_B)):
        above = h >= _H_B[i]
        t_b = xp.where(above, _T_B[i], t_b)
        lapse = xp.where(above, _L_B[i], lapse)
        p_b = xp.where(above, _P_B[i], p_b)
        h_b = xp.where(above, _H_B[i], h_b)
        iso = xp.where(above, 1.0 if _L_B[i] == 0.0 else 0.0, iso)
        expo = xp.where(above, GMR / (_L_B[i] if _L_B[i] != 0.0 else 1.0), expo)   # lapse_safe
    dh = h - h_b
    temp = t_b + lapse * dh
    p_gradient = p_b * (t_b / temp) ** expo
    p_isothermal = p_b * xp.exp(-GMR * dh / t_b)
    return xp.where(iso > 0.5, p_isothermal, p_gradient), temp


def pressure_temperature(xp, h_geometric):                                 # atmosphere.py:51-52,72-73
    return pressure_temperature_at_geopotential(xp, R0_GEOPOT_M * h_geometric / (R0_GEOPOT_M + h_geometric))


def pressure(xp, h):
    return pressure_temperature(xp, h)[0]


def density(xp, h):                                                        # atmosphere.py:80-82
    p, t = pressure_temperature(xp, h)
    return p / (R_AIR * t)


def speed_of_sound(xp, h):                                                 # atmosphere.py:85-87
    return xp.sqrt(GAMMA * R_AIR * pressure_temperature(xp, h)[1])


# ---- propulsion / actuators / mass properties (propulsion.py) ------------------------------------------------------------

def engine_thrust_per_engine(xp, throttle, p_ambient):                     # propulsion.py:46-48
    return xp.maximum(throttle * ENGINE_T_VAC_N - p_ambient * ENGINE_A_E_M2, 0.0)


def cluster_mdot(xp, engines_lit, throttle):                               # propulsion.py:51-53
    return engines_lit * throttle * ENGINE_T_VAC_N / (ENGINE_ISP_VAC_S * G0)


def split_mdot(mdot_total):                                                # propulsion.py:56-59
    mdot_lox = mdot_total * OF_RATIO / (1.0 + OF_RATIO)
    return mdot_lox, mdot_total - mdot_lox


def actuator_step(xp, x, cmd, dt, tau, rate_limit=None, lo=None, hi=None):  # propulsion.py:62-73
    alpha = 1.0 - xp.exp(-dt / tau)
    dx = alpha * (cmd - x)
    if rate_limit is not None:
        dx = xp.clip(dx, -rate_limit * dt, rate_limit * dt)
    x_new = x + dx
    if lo is not None or hi is not None:
        x_new = xp.clip(x_new, lo, hi)
    return x_new


def _column(mass, rho, bottom):                                            # propulsion.py:76-84
    length = mass / (rho * TANK_AREA_M2)
    r2 = STAGE_RADIUS_M ** 2
    return bottom + 0.5 * length, mass * (length ** 2 / 12.0 + r2 / 4.0), 0.5 * mass * r2


def stack_mass_props(xp, m_lox, m_rp1, m_upper=0.0):                       # propulsion.py:92-128
    r2 = STAGE_RADIUS_M ** 2
    dry_i_trans = STAGE1_DRY_MASS_KG * STAGE1_LENGTH_M ** 2 / 12.0
    dry_i_axial = 0.5 * STAGE1_DRY_MASS_KG * r2
    cg_lox, it_lox, ia_lox = _column(m_lox, RHO_LOX, LOX_TANK_BOTTOM_M)
    cg_rp1, it_rp1, ia_rp1 = _column(m_rp1, RHO_RP1, RP1_TANK_BOTTOM_M)
    up_i_trans = m_upper * STAGE2_LENGTH_M ** 2 / 12.0
    up_i_axial = 0.5 * m_upper * r2
    mass = STAGE1_DRY_MASS_KG + m_lox + m_rp1 + m_upper
    cg = (STAGE1_DRY_MASS_KG * DRY_CG_STATION_M + m_lox * cg_lox + m_rp1 * cg_rp1 + m_upper * STAGE2_CG_STATION_M) / mass

    def par_axis(i_own, m, station):
        return i_own + m * (station - cg) ** 2
    i_trans = (par_axis(dry_i_trans, STAGE1_DRY_MASS_KG, DRY_CG_STATION_M) + par_axis(it_lox, m_lox, cg_lox)
               + par_axis(it_rp1, m_rp1, cg_rp1) + par_axis(up_i_trans, m_upper, STAGE2_CG_STATION_M))
    i_axial = dry_i_axial + ia_lox + ia_rp1 + up_i_axial
    return mass, cg, xp.array([i_axial, i_trans, i_trans])


def tank_pressure_step(xp, p, m_prop, mdot_out, v_tank, rho, infill, vent, dt):   # propulsion.py:131-142
    v_ullage = xp.maximum(v_tank - m_prop / rho, 1e-2 * v_tank)
    dv_ullage = mdot_out / rho * dt
    p_drain = p * v_ullage / (v_ullage + dv_ullage)
    dp_infill = K_INFILL_PER_S * (P_REGULATOR_PA - p_drain) * infill * dt
    dp_vent = K_VENT_PER_S * (p_drain - P_AMBIENT_MIN_PA) * vent * dt
    return xp.maximum(p_drain + xp.maximum(dp_infill, 0.0) - xp.maximum(dp_vent, 0.0), 0.0)


def inlet_pressure(xp, p_tank, m_prop, rho, bottom, cg, a_axial, mdot):    # propulsion.py:145-149
    head_height = bottom + m_prop / (rho * TANK_AREA_M2)
    return p_tank + rho * xp.maximum(a_axial, 0.0) * head_height - 2.0e-2 * mdot ** 2


# ---- aerodynamics (aero.py) ----------------------------------------------------------------------------------------------

def config_blend(xp, v_axial_body):                                        # aero.py:75-81
    return 0.5 * (1.0 + xp.tanh(v_axial_body / 50.0))


def plume_dominance(xp, thrust, qbar):                                     # aero.py:84-87
    ct = thrust / xp.maximum(qbar * S_REF_M2, 1.0)
    return ct / (ct + PLUME_CT0)


def body_aero_wrench(xp, v_air_body, mach, qbar, cg, omega_body=None, ca_scale=1.0, cn_scale=1.0):   # aero.py:90-125
    speed = xp.linalg.norm(v_air_body)
    v_hat = v_air_body / xp.maximum(speed, 1e-6)
    blend = config_blend(xp, v_air_body[0])
    ca = (blend * xp.interp(mach, MACH_PTS, CA_ASCENT) + (1.0 - blend) * xp.interp(mach, MACH_PTS, CA_DESCENT)) * ca_scale
    cn = xp.interp(mach, MACH_PTS, CN_CROSS) * cn_scale
    x_hat = xp.array([1.0, 0.0, 0.0])
    axial = v_hat[0]
    cross = v_hat - axial * x_hat
    force = -qbar * S_REF_M2 * (ca * axial * x_hat + cn * cross)
    x_cp = blend * X_CP_ASCENT_M + (1.0 - blend) * X_CP_DESCENT_M
    torque = xp.cross((x_cp - cg) * x_hat, force)
    if omega_body is not None:
        cmq = blend * CMQ_ASCENT + (1.0 - blend) * CMQ_DESCENT
        damp = qbar * S_REF_M2 * (L_REF_DAMP_M ** 2) / (2.0 * xp.maximum(speed, 1.0)) * cmq
        torque = torque + damp * xp.array([0.0, omega_body[1], omega_body[2]])
    return force, torque


def fin_mix(xp, pitch_yaw_roll):                                            # aero.py:62-73,138-140: X-configuration mixing
    p, y, r = pitch_yaw_roll[0], pitch_yaw_roll[1], pitch_yaw_roll[2]
    return xp.array([d[2] * p + d[1] * y + r for d in FIN_FORCE_DIR])


def fin_wrench(xp, deltas, mach, qbar, cg, eff_scale=1.0):                  # aero.py:128-135
    cnd = xp.interp(mach, MACH_PTS, CN_DELTA_FIN) * eff_scale
    force, torque = xp.array([0.0, 0.0, 0.0]), xp.array([0.0, 0.0, 0.0])
    for i in range(4):
        f = (qbar * S_FIN_M2 * cnd * deltas[i]) * xp.array(FIN_FORCE_DIR[i])
        lever = xp.array([FIN_POS[i][0] - cg, FIN_POS[i][1], FIN_POS[i][2]])
        force, torque = force + f, torque + xp.cross(lever, f)
    return force, torque


# ---- cold-gas RCS (rcs.py) -------------------------------------------------------------------------------------------------

def rcs_wrench(xp, levels, cg, thrust=RCS_THRUST_PER_THRUSTER_N):           # rcs.py:60-65
    force, torque = xp.array([0.0, 0.0, 0.0]), xp.array([0.0, 0.0, 0.0])
    for i in range(N_RCS):
        f = (levels[i] * thrust) * xp.array(RCS_FORCE_DIR[i])
        lever = xp.array([RCS_POS[i][0] - cg, RCS_POS[i][1], RCS_POS[i][2]])
        force, torque = force + f, torque + xp.cross(lever, f)
    return force, torque


def rcs_torque_authority(xp, cg, thrust=RCS_THRUST_PER_THRUSTER_N):         # rcs.py:68-76, torque rows of B (3 x 8)
    cols = []
    for i in range(N_RCS):
        f = thrust * xp.array(RCS_FORCE_DIR[i])
        cols.append(xp.cross(xp.array([RCS_POS[i][0] - cg, RCS_POS[i][1], RCS_POS[i][2]]), f))
    return cols        # cols[i][axis]


def allocate_torque(xp, torque_cmd, cg, thrust=RCS_THRUST_PER_THRUSTER_N):  # rcs.py:84-107
    b = rcs_torque_authority(xp, cg, thrust)
    levels = [0.0] * N_RCS
    for axis, group_a, group_b in RCS_AXIS_GROUPS:
        cmd = torque_cmd[axis]
        auth_a = b[group_a[0]][axis] + b[group_a[1]][axis]
        auth_b = b[group_b[0]][axis] + b[group_b[1]][axis]
        use_a = xp.equal(xp.sign(cmd), xp.sign(auth_a))
        auth = xp.where(use_a, xp.abs(auth_a), xp.abs(auth_b))
        lvl = xp.clip(xp.abs(cmd) / xp.maximum(auth, 1e-9), 0.0, 1.0)
        active = xp.abs(cmd) > 0.02 * auth
        for i in group_a:
            levels[i] = levels[i] + xp.where(xp.logical_and(active, use_a), lvl, 0.0)
        for i in group_b:
            levels[i] = levels[i] + xp.where(xp.logical_and(active, xp.logical_not(use_a)), lvl, 0.0)
    return xp.clip(xp.array(levels), 0.0, 1.0)


# ---- quaternions on plain 4-vectors [x, y, z, w] --------------------------------------------------------------------------

def quat_mul(xp, l, r):                                                    # quaternion.rs:268-281
    return xp.array([l[3] * r[0] + l[0] * r[3] + l[1] * r[2] - l[2] * r[1],
                     l[3] * r[1] - l[0] * r[2] + l[1] * r[3] + l[2] * r[0],
                     l[3] * r[2] + l[0] * r[1] - l[1] * r[0] + l[2] * r[3],
                     l[3] * r[3] - l[0] * r[0] - l[1] * r[1] - l[2] * r[2]])


def quat_inverse(xp, q):                                                   # quaternion.rs:141-155: conj / |q|^2
    n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]
    return xp.array([-q[0] / n2, -q[1] / n2, -q[2] / n2, q[3] / n2])


def quat_rotate(xp, q, v):                                                 # quaternion.rs:283-305 (unit q)
    u = xp.array([q[0], q[1], q[2]])
    t = 2.0 * xp.cross(u, v)
    return v + q[3] * t + xp.cross(u, t)


def quat_between_x(xp, to):                                                # math.rs:122-138 with from = +X
    """Shortest rotation taking body +X onto the unit vector `to`."""
    c = xp.clip(to[0], -1.0, 1.0)
    ay, az = -to[2], to[1]                                                 # cross([1,0,0], to) = [0, -to_z, to_y]
    n = xp.sqrt(ay * ay + az * az)
    n_safe = xp.where(n > 0.0, n, 1.0)
    half = 0.5 * xp.arccos(c)
    s, w = xp.sin(half), xp.cos(half)
    q = xp.array([0.0, ay / n_safe * s, az / n_safe * s, w])
    ident = xp.array([0.0, 0.0, 0.0, 1.0])
    flip = xp.array([0.0, 0.0, 1.0, 0.0])                                   # antipodal: 180 deg about normalize(x cross y) = +Z
    q = xp.where(c > 1.0 - 1e-12, ident, q)
    return xp.where(c < -1.0 + 1e-12, flip, q)


def fsw_density(xp, alt):                                                  # math.rs:191-199 (the FSW's own two-piece model)
    h = xp.maximum(alt, 0.0)
    return xp.where(h < 25_000.0, 1.225 * xp.exp(-h / 8_440.0), 0.0642 * xp.exp(-(h - 25_000.0) / 6_580.0))


# ---- the recorded ascent profile the flight software flies (controller/src/profile.rs) ------------------------------------

def resample_profile(time_s, velocity_mps, altitude_km):
    """AscentProfile::load after the JSON parse (profile.rs:44-85): the recorded webcast telemetry on a uniform 0.5 s grid,
    a 9-point moving average (window [i-4, i+5) clipped to the table) over speed and altitude, central differences of the
    smoothed altitude for the vertical speed.  Returns (time, speed, alt_m, vspeed)."""
    time_s, velocity_mps, altitude_km = (np.asarray(a, dtype=np.float64) for a in (time_s, velocity_mps, altitude_km))
    n = int(time_s[-1] / 0.5) + 1
    grid = np.arange(n, dtype=np.float64) * 0.5

    def interp(x, xs, ys):                                                  # profile.rs:23-38 (bisection on `xs[mid] <= x`)
        if x <= xs[0]:
            return ys[0]
        if x >= xs[-1]:
            return ys[-1]
        lo = int(np.searchsorted(xs, x, side="right")) - 1
        return ys[lo] + (x - xs[lo]) / (xs[lo + 1] - xs[lo]) * (ys[lo + 1] - ys[lo])

    def smooth(v):
        out = np.empty(n)
        for i in range(n):
            lo, hi = max(i - 4, 0), min(i + 5, n)
            acc = 0.0
            for k in range(lo, hi):                                         # left-to-right, like iter().sum()
                acc += v[k]
            out[i] = acc / (hi - lo)
        return out

    speed = smooth(np.array([interp(t, time_s, velocity_mps) for t in grid]))
    alt_m = smooth(np.array([interp(t, time_s, altitude_km) * 1000.0 for t in grid]))
    vspeed = np.zeros(n)
    for i in range(n):
        lo, hi = max(i - 1, 0), min(i + 1, n - 1)
        vspeed[i] = 0.0 if hi == lo else (alt_m[hi] - alt_m[lo]) / ((hi - lo) * 0.5)
    return grid, speed, alt_m, vspeed


_PROFILE_CACHE: dict = {}


def ascent_profile(mission: str = "crs12"):
    """(time, speed, alt_m, vspeed) tuples of the mission's resampled profile, from elodin_amd/data/falcon9_<mission>_profile.csv
    (written by tests/golden/make_falcon9_profile.py: `resample_profile` over the reference's data/<mission>/stage1_raw.json,
    the file ELODIN_F9_PROFILE points the flight software at, main.py:193-199)."""
    if mission not in _PROFILE_CACHE:
        from pathlib import Path
        path = Path(__file__).resolve().parents[1] / "data" / f"falcon9_{mission}_profile.csv"
        tab = np.loadtxt(path, delimiter=",", skiprows=1)
        _PROFILE_CACHE[mission] = tuple(tuple(float(v) for v in tab[:, k]) for k in range(4))
    return _PROFILE_CACHE[mission]


# ---- host-side frames (numpy) ------------------------------------------------------------------------------------------

def pad_ecef() -> np.ndarray:                                              # sim.py:1187-1188
    return np.asarray(geodetic_to_ecef(np, math.radians(PAD_LAT_DEG), math.radians(PAD_LON_DEG), PAD_ALT_M))


def pad_up() -> np.ndarray:                                                # frames.py:87-89 at the pad
    return -np.asarray(ned_basis(np, math.radians(PAD_LAT_DEG), math.radians(PAD_LON_DEG))[2])


def upright_attitude() -> np.ndarray:                                      # sim.py:1204-1211: body +X along pad up
    up, x = pad_up(), np.array([1.0, 0.0, 0.0])
    axis = np.cross(x, up)
    axis /= np.linalg.norm(axis)
    ang = math.acos(float(np.clip(x @ up, -1.0, 1.0)))
    return np.concatenate([axis * math.sin(ang / 2.0), [math.cos(ang / 2.0)]])


# ---- the plant, as dsl systems (one function per reference system) ---------------------------------------------------------

def build_program(origin: Optional[Sequence[float]] = None, fsw: bool = True, scripted=None,
                  algebraic_geodesy: bool = False) -> dsl.Program:
    """`propulsion_systems | six_dof(gravity_and_frame_forces | apply_body_wrenches) | pad_clamp | telemetry | fsw`
    (sim.py:1433-1530) for the ascent.

    `origin`: world_pos is stored RELATIVE to this ECEF point (None = plain ECEF like the reference).  f32 campaigns must
    use the pad as origin: an f32 ECEF coordinate has a 0.5 m ulp, larger than a millisecond of flight.
    `scripted(xp, t) -> (engine_cmd[9], valve_cmd[8])` — or a dict of any of the command columns engine_cmd, valve_cmd,
    attitude_setpoint, ctrl_enable, fin_cmd, fsw_phase — replaces the flight software by an open-loop script
    (test_propulsion.py:113-135 `_script`) for the reference's open-loop known-answer tests and for the plant
    trajectories tests/golden/make_falcon9_fixtures.py records from the reference's own systems.
    `algebraic_geodesy`: ECEF -> geodetic by geodetic_sincos (the same recurrence without the angle round trips) — the f32
    campaign builds; off, every conversion is the reference's ecef_to_geodetic operation for operation.
    Run the returned program with the semi-implicit integrator at 1 kHz (build_powered's default, sim.py:1476).
    """
    xp = dsl.np
    org = tuple(float(v) for v in (origin if origin is not None else (0.0, 0.0, 0.0)))
    dt = SIM_TIME_STEP
    pad_rel = tuple(float(a - b) for a, b in zip(pad_ecef(), org))

    def ecef(pos):                      # stored coordinates -> ECEF
        return pos.linear() + xp.array(org)

    def geodetic(r):                    # (sin lat, cos lat, sin lon, cos lon, altitude) of an ECEF point
        if algebraic_geodesy:
            return geodetic_sincos(xp, r)
        lat, lon, alt = ecef_to_geodetic(xp, r)
        return xp.sin(lat), xp.cos(lat), xp.sin(lon), xp.cos(lon), alt

    @dsl.system
    def attitude_control(pos, vel, inertia, attitude_setpoint, ctrl_enable, thrust_total, cg_station, fsw_phase):
        """sim.py:649-709: inertia-scaled quaternion-error PD -> TVC gimbal command (+ RCS torque request)."""
        q = pos.angular().vector()
        q_inv = quat_inverse(xp, q)
        err = quat_mul(xp, q_inv, attitude_setpoint)
        sign = xp.where(err[3] >= 0.0, 1.0, -1.0)
        err_vec = sign * err[:3]
        omega_body = quat_rotate(xp, q_inv, vel.angular())
        i_diag = inertia.inertia_diag()
        tvc_on = (ctrl_enable[0] > 0.5) & (thrust_total > 2.0e5)
        rcs_on = ctrl_enable[1] > 0.5
        landing_burn = (fsw_phase >= 10.0) & (fsw_phase < 11.0)
        wn = xp.where(tvc_on, xp.where(landing_burn, ATT_WN_TVC_LANDING, ATT_WN_TVC), ATT_WN_RCS)
        zeta = xp.where(tvc_on, ATT_ZETA_TVC, ATT_ZETA_RCS)
        torque_des = i_diag * (wn ** 2 * err_vec - 2.0 * zeta * wn * omega_body)
        lever = xp.maximum(cg_station * thrust_total, 1.0)
        tvc_cmd = xp.where(tvc_on, xp.array([-torque_des[1] / lever, -torque_des[2] / lever]), xp.zeros(2))
        in_deadband = (xp.linalg.norm(err_vec) < 0.009) & (xp.linalg.norm(omega_body) < 0.01)
        rcs_torque = xp.where(tvc_on, xp.array([torque_des[0], 0.0, 0.0]), torque_des)
        rcs_torque = xp.where(rcs_on & ~in_deadband, rcs_torque, xp.zeros(3))
        return {"tvc_cmd": tvc_cmd, "rcs_torque_cmd": rcs_torque}

    @dsl.system
    def valve_dynamics(valve_state, valve_cmd):                             # sim.py:364-369
        return {"valve_state": actuator_step(xp, valve_state, xp.clip(valve_cmd, 0.0, 1.0), dt, VALVE_TAU_S, lo=0.0, hi=1.0)}

    @dsl.system
    def tvc_actuators(tvc_state, tvc_cmd):                                  # sim.py:513-523
        return {"tvc_state": actuator_step(xp, tvc_state, xp.clip(tvc_cmd, -TVC_MAX_RAD, TVC_MAX_RAD), dt, TVC_TAU_S,
                                           rate_limit=TVC_RATE_RADPS, lo=-TVC_MAX_RAD, hi=TVC_MAX_RAD)}

    @dsl.system
    def fin_actuators(fin_state, fin_cmd):                                  # sim.py:526-538
        deltas_cmd = fin_mix(xp, xp.clip(fin_cmd, -FIN_MAX_RAD, FIN_MAX_RAD))
        return {"fin_state": actuator_step(xp, fin_state, xp.clip(deltas_cmd, -FIN_MAX_RAD, FIN_MAX_RAD), dt, FIN_TAU_S,
                                           rate_limit=FIN_RATE_RADPS, lo=-FIN_MAX_RAD, hi=FIN_MAX_RAD)}

    @dsl.system
    def rcs_dynamics(rcs_levels, rcs_torque_cmd, cg_station, nitrogen_kg):
        """sim.py:560-576: allocate the requested torque to the eight cold-gas thrusters, valve dynamics, meter N2."""
        def active(levels, torque_cmd, cg, n2):
            have_gas = n2 > 0.0
            cmd_levels = xp.where(have_gas, allocate_torque(xp, torque_cmd, cg), xp.zeros(N_RCS))
            levels_next = actuator_step(xp, levels, cmd_levels, dt, RCS_VALVE_TAU_S, lo=0.0, hi=1.0)
            force, torque = rcs_wrench(xp, levels_next, cg)
            thrust_sum = xp.sum(levels_next) * RCS_THRUST_PER_THRUSTER_N
            n2_next = xp.maximum(n2 - thrust_sum / (N2_ISP_S * G0) * dt, 0.0)
            return levels_next, xp.concatenate([force, torque]), n2_next

        def quiescent(levels, torque_cmd, cg, n2):
            # closed valves and no torque request: the allocation is zero (|cmd| > 2 % of the authority fails), the valves stay
            # where they are, the wrench is zero and no gas flows — what `active` computes for these inputs, in 3 instructions
            return levels, xp.zeros(6), xp.maximum(n2, 0.0)
        busy = None
        for v in list(rcs_levels) + list(rcs_torque_cmd):
            b = ~xp.equal(v, 0.0)
            busy = b if busy is None else (busy | b)
        # the cold-gas system is idle for the whole powered ascent (it flies the coast and the flip): a wave whose rollouts all
        # have it idle skips the allocation / valve / wrench arithmetic (~280 issue slots of a tick)
        levels_next, wrench, n2_next = dsl.lax.branch_cond(busy, active, quiescent, rcs_levels, rcs_torque_cmd, cg_station, nitrogen_kg)
        return {"rcs_levels": levels_next, "rcs_wrench": wrench, "nitrogen_kg": n2_next}

    @dsl.system
    def wind_model(pos, wind_ned):
        """sim.py:579-611 without the gust process (gust_sigma = 0 in every shipped spec; the gust draws from jax.random):
        steady NED wind with the near-surface shear factor, rotated into ECEF."""
        sl, cl, so, co, alt = geodetic(ecef(pos))
        north, east, down = ned_rows(xp, sl, cl, so, co)
        shear = xp.clip(1.0 + 0.15 * (500.0 - xp.minimum(alt, 500.0)) / 500.0, 1.0, 1.15)
        w = wind_ned * shear
        return {"wind_ecef": north * w[0] + east * w[1] + down * w[2]}

    @dsl.system
    def engine_dynamics(pos, engine_cmd, engine_spool, engine_armed, teateb_charges, valve_state, propellant_lox,
                        propellant_rp1, params):
        """sim.py:372-430: ignition gating (TEA-TEB charge + feed + igniter valves), three-regime spool, thrust with
        ambient back-pressure, mass flow."""
        feed_open = (valve_state[VALVE_MAIN_LOX] > 0.5) & (valve_state[VALVE_MAIN_RP1] > 0.5)
        teateb_open = valve_state[VALVE_TEATEB] > 0.5
        prop_ok = (propellant_lox > 0.0) & (propellant_rp1 > 0.0)
        alt = geodetic(ecef(pos))[4]
        p_amb = pressure(xp, xp.maximum(alt, 0.0))
        thrust_scale, isp_scale = params[P["thrust_scale"]], params[P["isp_scale"]]

        def step(cmd, spool, armed, charged):
            """The reference's per-engine update over m engines (vectors of length m; `charged` = 1 where the engine still holds a
            TEA-TEB charge, the only way the charge COUNT enters): new spool / armed state, 1 where an engine lights on this
            tick, per-engine thrust and mass flow."""
            m = len(cmd)
            ones, zeros = xp.ones(m), xp.zeros(m)
            cmd_c = xp.clip(cmd, 0.0, 1.0)
            cmd_on = cmd_c >= THROTTLE_MIN * 0.5
            lighting = cmd_on & (armed < 0.5) & (charged > 0.5) & feed_open & teateb_open & prop_ok
            lit_now = xp.where(lighting, ones, zeros)
            armed_next = xp.where(cmd_on & ((armed > 0.5) | lighting), ones, zeros)
            burn_ok = (armed_next > 0.5) & feed_open & prop_ok
            target = xp.where(burn_ok, xp.maximum(cmd_c, THROTTLE_MIN), zeros)
            running = spool > 0.5 * THROTTLE_MIN
            tau = xp.where(target > spool, xp.where(running, ones * ENGINE_THROTTLE_TAU_S, ones * ENGINE_SPINUP_TAU_S),
                           ones * ENGINE_SHUTDOWN_TAU_S)
            spool_next = dsl.Vec([actuator_step(xp, s_, t_, dt, ta, lo=0.0, hi=1.0) for s_, t_, ta in zip(spool, target, tau)])
            lit = spool_next > 1e-3
            thrust_per = xp.where(lit, engine_thrust_per_engine(xp, spool_next, p_amb) * thrust_scale, zeros)
            mdot = cluster_mdot(xp, xp.where(lit, ones, zeros), spool_next) * (thrust_scale / isp_scale)
            return spool_next, armed_next, lit_now, thrust_per, mdot

        def all_nine(cmd, spool, armed, charged):
            s_, a_, l_, thrust_per, mdot = step(cmd, spool, armed, charged)
            return s_, a_, l_, xp.sum(thrust_per), xp.sum(mdot)

        def one_for_all(cmd, spool, armed, charged):
            # nine engines in the same state under the same command take the same step: engine 0's, nine times over — the same
            # arithmetic on the same numbers (the sums add nine equal terms in the same order), so not an approximation
            s_, a_, l_, thrust_per, mdot = step(cmd[:1], spool[:1], armed[:1], charged[:1])
            rep = lambda v: dsl.Vec([v[0]] * N_ENGINES)
            return rep(s_), rep(a_), rep(l_), xp.sum(rep(thrust_per)), xp.sum(rep(mdot))
        # what an engine's TEA-TEB charges can still do: light it, if it is not burning.  (The counts differ by design — three
        # engines carry relight charges — and so does "has a charge left" once the cluster has lit; an armed engine ignores both.)
        charged = xp.where((teateb_charges >= 1.0) & (engine_armed < 0.5), xp.ones(N_ENGINES), xp.zeros(N_ENGINES))
        differ = None
        for v in (engine_cmd, engine_spool, engine_armed, charged):
            for k in range(1, N_ENGINES):
                d = ~xp.equal(v[k], v[0])
                differ = d if differ is None else (differ | d)
        # most of an ascent the cluster runs as one (an engine-out scenario, or a relight of three, splits it): then a wave
        # takes the cheap side only — 822 -> ~170 issue slots of a ~3,000-slot tick (tools/rollout_split_model.py's cost table)
        spool_next, armed_next, lit_now, thrust_total, mdot_total = dsl.lax.branch_cond(
            differ, all_nine, one_for_all, engine_cmd, engine_spool, engine_armed, charged)
        return {"engine_spool": spool_next, "engine_armed": armed_next, "teateb_charges": teateb_charges - lit_now,
                "thrust_total": thrust_total, "mdot_total": mdot_total}

    @dsl.system
    def mass_props(mdot_total, propellant_lox, propellant_rp1, thrust_total, upper_mass):
        """sim.py:433-454: deplete propellant, rebuild mass / CG / inertia from the cylinder stack."""
        mdot_lox, mdot_rp1 = split_mdot(mdot_total)
        lox_next = xp.maximum(propellant_lox - mdot_lox * dt, 0.0)
        rp1_next = xp.maximum(propellant_rp1 - mdot_rp1 * dt, 0.0)
        mass, cg, inertia_diag = stack_mass_props(xp, lox_next, rp1_next, xp.maximum(upper_mass, 0.0))
        return {"propellant_lox": lox_next, "propellant_rp1": rp1_next, "inertia": dsl.SpatialInertia(inertia_diag, mass),
                "cg_station": cg, "axial_specific_force": thrust_total / mass}

    @dsl.system
    def tank_dynamics(tank_pressure_lox, tank_pressure_rp1, propellant_lox, propellant_rp1, mdot_total, valve_state,
                      axial_specific_force, cg_station):
        """sim.py:457-507: ullage and engine-inlet pressures (telemetry; nothing downstream in the ascent reads them)."""
        mdot_lox, mdot_rp1 = split_mdot(mdot_total)
        p_lox = tank_pressure_step(xp, tank_pressure_lox, propellant_lox, mdot_lox, V_TANK_LOX_M3, RHO_LOX,
                                   valve_state[VALVE_HE_INFILL_LOX], valve_state[VALVE_HE_VENT_LOX], dt)
        p_rp1 = tank_pressure_step(xp, tank_pressure_rp1, propellant_rp1, mdot_rp1, V_TANK_RP1_M3, RHO_RP1,
                                   valve_state[VALVE_HE_INFILL_RP1], valve_state[VALVE_HE_VENT_RP1], dt)
        return {"tank_pressure_lox": p_lox, "tank_pressure_rp1": p_rp1,
                "inlet_pressure_lox": inlet_pressure(xp, p_lox, propellant_lox, RHO_LOX, LOX_TANK_BOTTOM_M, cg_station,
                                                     axial_specific_force, mdot_lox),
                "inlet_pressure_rp1": inlet_pressure(xp, p_rp1, propellant_rp1, RHO_RP1, RP1_TANK_BOTTOM_M, cg_station,
                                                     axial_specific_force, mdot_rp1)}

    @dsl.system
    def engine_wrench_sys(thrust_total, tvc_state, cg_station):             # sim.py:541-550
        d = xp.array([1.0, tvc_state[1], -tvc_state[0]])
        d = d / xp.linalg.norm(d)
        force = thrust_total * d
        torque = xp.cross(xp.array([-cg_station, 0.0, 0.0]), force)
        return {"engine_wrench": xp.concatenate([force, torque])}

    @dsl.system
    def aero_dynamics(pos, vel, wind_ecef, thrust_total, fin_state, cg_station, params):
        """sim.py:614-660: air data, body aero wrench with plume dominance, grid-fin wrench."""
        alt = geodetic(ecef(pos))[4]
        alt = xp.maximum(alt, 0.0)
        rho = density(xp, alt)
        a_sound = speed_of_sound(xp, alt)
        q_inv = quat_inverse(xp, pos.angular().vector())
        v_air_body = quat_rotate(xp, q_inv, vel.linear() - wind_ecef)
        omega_body = quat_rotate(xp, q_inv, vel.angular())
        speed = xp.linalg.norm(v_air_body)
        qbar = 0.5 * rho * speed ** 2
        mach = speed / a_sound
        f_aero, t_aero = body_aero_wrench(xp, v_air_body, mach, qbar, cg_station, omega_body=omega_body,
                                          ca_scale=params[P["ca_scale"]], cn_scale=params[P["cn_scale"]])
        kappa = plume_dominance(xp, thrust_total, qbar)
        f_fin, t_fin = fin_wrench(xp, fin_state, mach, qbar, cg_station)
        return {"qbar": qbar, "mach": mach, "aero_wrench": xp.concatenate([f_aero * (1.0 - kappa), t_aero * (1.0 - kappa)]),
                "fin_wrench": xp.concatenate([f_fin, t_fin])}

    @dsl.effector
    def gravity_and_frame_forces(force, inertia, pos, vel):                 # sim.py:350-358
        r = ecef(pos)
        accel = gravity_accel(xp, r) + frame_accel(xp, r, vel.linear())
        return force + dsl.SpatialForce(linear=accel * inertia.mass())

    @dsl.effector(engine_wrench=6, aero_wrench=6, fin_wrench=6, rcs_wrench=6)
    def apply_body_wrenches(engine_wrench, aero_wrench, fin_wrench, rcs_wrench, force, pos):   # sim.py:663-676 (no leg contact)
        total = engine_wrench + aero_wrench + fin_wrench + rcs_wrench
        q = pos.angular()
        return force + dsl.SpatialForce(linear=q @ total[:3], torque=q @ total[3:])

    @dsl.system
    def pad_clamp(pos, vel, lifted, liftoff_time, thrust_total, inertia, tick):
        """sim.py:984-1013: hold-down clamps until thrust exceeds weight, latch the release time."""
        t_s = tick * dt
        weight = inertia.mass() * 9.79
        was_lifted = lifted > 0.5
        release = was_lifted | (thrust_total > weight)
        first = ~was_lifted & release
        held = dsl.SpatialTransform(pos.angular(), xp.where(release, pos.linear(), xp.array(pad_rel)))
        held_vel = dsl.SpatialMotion(xp.where(release, vel.angular(), xp.zeros(3)), xp.where(release, vel.linear(), xp.zeros(3)))
        return {"pos": held, "vel": held_vel, "lifted": xp.where(release, 1.0, 0.0),
                "liftoff_time": xp.where(first, t_s, liftoff_time)}

    @dsl.system
    def derive_geodetic_telemetry(pos, vel):                                # sim.py:1128-1137
        alt = geodetic(ecef(pos))[4]
        return {"altitude_geodetic": alt, "ground_speed": xp.linalg.norm(vel.linear())}

    @dsl.system
    def ascent_metrics_latch(ascent_metrics, qbar, engine_wrench, aero_wrench, inertia, tick, fsw_phase, fsw_state, pos, vel):
        """Campaign observables the reference's hooks derive from DB telemetry afterwards (hooks/score.py): Max-Q, peak
        sensed acceleration, and the state at MECO, latched in the loop.  Altitude and ground speed are derive_geodetic_
        telemetry's expressions on the same state (nothing moves the vehicle in between), spelled again here so that the
        COLUMNS have no reader: codegen then evaluates them where they are stored, and these uses sit behind `at_meco`."""
        t_s = tick * dt
        m = ascent_metrics
        new_q = qbar > m[0]
        f_body = (engine_wrench[:3] + aero_wrench[:3]) / inertia.mass()     # fin / rcs forces are zero during the ascent
        a_sensed = xp.linalg.norm(f_body)
        r = ecef(pos)
        sl, cl, so, co, altitude_geodetic = geodetic(r)
        up = -ned_rows(xp, sl, cl, so, co)[2]
        v = vel.linear()
        ground_speed = xp.linalg.norm(v)
        fpa = xp.rad2deg(xp.arcsin(xp.clip(xp.dot(v, up) / xp.maximum(ground_speed, 1e-9), -1.0, 1.0)))
        downrange = xp.linalg.norm(r - xp.array(tuple(float(a) for a in pad_ecef())))
        at_meco = (fsw_state[3] > 0.5) & (m[3] <= 0.0)          # cutoff commanded, not latched yet
        return {"ascent_metrics": xp.array([xp.where(new_q, qbar, m[0]), xp.where(new_q, t_s, m[1]), xp.maximum(m[2], a_sensed),
                                            xp.where(at_meco, t_s, m[3]), xp.where(at_meco, altitude_geodetic, m[4]),
                                            xp.where(at_meco, ground_speed, m[5]), xp.where(at_meco, fpa, m[6]),
                                            xp.where(at_meco, downrange, m[7])])}

    # ---- sensors (sensors.py, sim.py:1016-1110): what the flight software flies on ---------------------------------------------
    sensor_key = dsl.random.key(SENSOR_SEED)

    def noise(count, salt, n, sigma):                                        # sensors.py:105-107
        key = dsl.random.fold_in(dsl.random.fold_in(sensor_key, salt), count)
        return (dsl.random.normal(key, shape=(n,)) if n else dsl.random.normal(key)) * sigma

    # The IMU and the pressure transducers sample every tick in the reference; their samples are consumed at the guidance
    # exchanges only (main.py:280-303) and depend on nothing but the tick they are taken on (noise keyed by the sample
    # counter = ticks done), so they are evaluated on the exchange ticks: same packets, a tenth of the draws.
    @dsl.system(every=GUIDANCE_PERIOD_TICKS, phase=1)
    def imu_model(tick, pos, vel, inertia, engine_wrench, aero_wrench, fin_wrench, rcs_wrench):
        """sim.py:1019-1038: specific force = summed non-gravitational body force / mass; the gyro measures the inertial
        rate (frame rate + Earth rate)."""
        count = tick                                                        # sensor_tick + 1 = ticks done
        f_body = (engine_wrench[:3] + aero_wrench[:3] + fin_wrench[:3] + rcs_wrench[:3]) / inertia.mass()
        q_inv = quat_inverse(xp, pos.angular().vector())
        gyro = quat_rotate(xp, q_inv, vel.angular() + xp.array([0.0, 0.0, OMEGA_EARTH_RADPS]))
        return {"sensor_tick": count, "imu_accel": f_body + noise(count, 1, 3, IMU_ACCEL_SIGMA),
                "imu_gyro": gyro + noise(count, 2, 3, IMU_GYRO_SIGMA)}

    @dsl.system
    def gps_model(gps_timer, pos, vel, mach, thrust_total, gps_pos, gps_vel, gps_count):
        """sim.py:1041-1068: 25 Hz position / velocity on a timer-accumulator + hold, blacked out under a supersonic
        plume.  gps_pos is in the executor's coordinates (ECEF minus `origin`)."""
        t = gps_timer + dt
        fired = t >= GPS_DT_S
        t = xp.where(fired, t - GPS_DT_S, t)
        blackout = (mach > BLACKOUT_MACH_MIN) & (thrust_total > BLACKOUT_THRUST_MIN_N)
        fresh = fired & ~blackout
        n = gps_count + xp.where(fresh, 1.0, 0.0)
        # the fix (six normal draws) is computed on the ticks that deliver one — a real branch, taken once in 40 ticks
        pos_new, vel_new = dsl.lax.branch_cond(
            fresh, lambda n_, p_, v_: (p_ + noise(n_, 3, 3, GPS_POS_SIGMA), v_ + noise(n_, 4, 3, GPS_VEL_SIGMA)),
            lambda n_, p_, v_: (gps_pos, gps_vel), n, pos.linear(), vel.linear())
        return {"gps_timer": t, "gps_pos": pos_new, "gps_vel": vel_new, "gps_count": n}

    @dsl.system
    def radar_altimeter_model(radar_timer, pos, radar_range, radar_count):
        """sim.py:1071-1096: boresight (-X body) range to the ellipsoid at 40 Hz inside the FOV / range gates, -1 otherwise."""
        t = radar_timer + dt
        fired = t >= RADAR_DT_S
        t = xp.where(fired, t - RADAR_DT_S, t)
        n = radar_count + xp.where(fired, 1.0, 0.0)

        def ping(n_, held, p_, q_):
            # the boresight geometry (and, inside the gates, the noise draw) on the ticks the altimeter fires: 1 in 25
            sl, cl, so, co, alt = geodetic(p_ + xp.array(org))
            up = xp.array([cl * co, cl * so, sl])
            bore_world = quat_rotate(xp, q_, xp.array([-1.0, 0.0, 0.0]))
            cos_tilt = xp.dot(bore_world, -up)
            slant = alt / xp.maximum(cos_tilt, 1e-3)
            valid = (cos_tilt > RADAR_FOV_COS) & (slant <= RADAR_MAX_RANGE_M) & (alt > 0.0)
            return xp.where(valid, slant + noise(n_, 5, 0, RADAR_SIGMA_M), -1.0)
        rng_new = dsl.lax.branch_cond(fired, ping, lambda n_, held, p_, q_: held, n, radar_range, pos.linear(), pos.angular().vector())
        return {"radar_timer": t, "radar_range": rng_new, "radar_count": n}

    @dsl.system(every=GUIDANCE_PERIOD_TICKS, phase=1)
    def pressure_transducers(sensor_tick, tank_pressure_lox, tank_pressure_rp1, propellant_lox, propellant_rp1, mdot_total,
                             axial_specific_force, cg_station):
        # sim.py:1099-1109.  The inlet pressures are tank_dynamics' expressions on the values it left in the columns (same
        # numbers), evaluated on the sampling ticks: the inlet_pressure_* columns themselves then have no reader
        mdot_lox, mdot_rp1 = split_mdot(mdot_total)
        inlet_lox = inlet_pressure(xp, tank_pressure_lox, propellant_lox, RHO_LOX, LOX_TANK_BOTTOM_M, cg_station,
                                   axial_specific_force, mdot_lox)
        inlet_rp1 = inlet_pressure(xp, tank_pressure_rp1, propellant_rp1, RHO_RP1, RP1_TANK_BOTTOM_M, cg_station,
                                   axial_specific_force, mdot_rp1)
        truth = xp.array([tank_pressure_lox, tank_pressure_rp1, inlet_lox, inlet_rp1])
        return {"pressure_meas": truth + noise(sensor_tick, 6, 4, PRESSURE_SIGMA_PA)}

    # ---- the flight software (controller/src/main.rs), one exchange per 10 ticks ---------------------------------------------
    prof_t, prof_speed, prof_alt, prof_vspeed = ascent_profile()

    def up_and_ned(r_ecef):
        sl, cl, so, co, alt = geodetic(r_ecef)
        north, east, down = ned_rows(xp, sl, cl, so, co)
        return -down, north, east, alt

    def normalize(v):                                                       # math.rs:41-48
        n = xp.linalg.norm(v)
        return xp.where(n < 1e-12, xp.zeros(3), v * (1.0 / xp.maximum(n, 1e-300)))

    @dsl.system(every=GUIDANCE_PERIOD_TICKS, phase=1)
    def fsw_ascent(tick, params, fsw_state, fsw_frame, nav_pos, nav_vel, nav_att, nav_aux, imu_accel, imu_gyro, gps_pos,
                   gps_vel, gps_count, radar_range):
        """controller/src/main.rs:384-533 with its navigator (main.rs:213-327), called where main.py's post_step exchanges
        packets: after ticks 1, 11, 21, ... with t = (ticks done - 1) * dt (impeller2_server.rs:553-678: post_step gets the
        index of the tick just finished; main.py:276-281).  Branch-free: every `if` of the Rust is a select.
        fsw_state = [phase, phase_t0, purge_until, meco latched, t_liftoff, pad frame set]; nav_aux = [initialized,
        last_gps_count, last_t, radar_alt_m]; fsw_frame = [up_pad, track_dir].  The `fsw_phase` column carries the phase the
        command was computed in (Command.phase is set before the transition, main.rs:386-390).  Positions are in the
        executor's coordinates (ECEF minus `origin`); `ecef()` is applied where the Rust needs the geocentric vector."""
        t = (tick - 1.0) * dt
        phase, phase_t0, purge_until, meco, t_liftoff, pad_set = (fsw_state[k] for k in range(6))
        inited, last_gps, last_t, radar_alt = nav_aux[0] > 0.5, nav_aux[1], nav_aux[2], nav_aux[3]
        org_v = xp.array(org)
        x_axis = xp.array([1.0, 0.0, 0.0])

        # -- Navigator::step (main.rs:241-295) on an initialised navigator
        dtn = xp.clip(t - last_t, 0.0, 0.1)
        att_conj = xp.array([-nav_att[0], -nav_att[1], -nav_att[2], nav_att[3]])
        omega_frame = imu_gyro - quat_rotate(xp, att_conj, xp.array([0.0, 0.0, OMEGA_EARTH_RADPS]))
        wn = xp.linalg.norm(omega_frame)
        angle = wn * dtn
        axis = normalize(omega_frame)
        sh, ch = xp.sin(angle * 0.5), xp.cos(angle * 0.5)
        att_q = quat_mul(xp, nav_att, xp.array([axis[0] * sh, axis[1] * sh, axis[2] * sh, ch]))
        att_n = xp.sqrt(att_q[0] * att_q[0] + att_q[1] * att_q[1] + att_q[2] * att_q[2] + att_q[3] * att_q[3])
        att_p = xp.where(angle < 1e-12, nav_att, att_q / att_n)
        f_e = quat_rotate(xp, att_p, imu_accel)
        r_nav = nav_pos + org_v
        acc = f_e + gravity_accel(xp, r_nav) + frame_accel(xp, r_nav, nav_vel)
        vel_p = nav_vel + acc * dtn
        pos_p = nav_pos + vel_p * dtn
        fresh = gps_count > last_gps                                        # complementary GPS blend
        pos_b = xp.where(fresh, pos_p + (gps_pos - pos_p) * 0.20, pos_p)
        vel_b = xp.where(fresh, vel_p + (gps_vel - vel_p) * 0.50, vel_p)
        radar_ok = (radar_range >= 0.0) & (radar_range < 500.0)             # radar altimeter below 500 m
        up_b, _, _, geo_alt_b = up_and_ned(pos_b + org_v)
        radar_alt_new = xp.where(radar_alt < 0.0, radar_range, 0.7 * radar_alt + 0.3 * radar_range)
        dh = radar_alt_new - geo_alt_b
        pos_r = xp.where(radar_ok & (xp.abs(dh) < 50.0), pos_b + up_b * (0.35 * dh), pos_b)
        radar_alt_p = xp.where(radar_ok, radar_alt_new, -1.0)
        # -- Navigator::init (main.rs:227-239) on the first packet that carries a GPS fix
        init_now = ~inited & (gps_count > 0.0)
        up_gps, _, _, _ = up_and_ned(gps_pos + org_v)
        pos_n = xp.where(inited, pos_r, xp.where(init_now, gps_pos, nav_pos))
        vel_n = xp.where(inited, vel_b, xp.where(init_now, xp.zeros(3), nav_vel))
        att_nx = xp.where(inited, att_p, xp.where(init_now, quat_between_x(xp, up_gps), nav_att))
        last_gps_n = xp.where(inited, xp.where(fresh, gps_count, last_gps), xp.where(init_now, gps_count, last_gps))
        last_t_n = xp.where(inited | init_now, t, last_t)
        radar_alt_n = xp.where(inited, radar_alt_p, xp.where(init_now, -1.0, radar_alt))
        ready = inited | init_now                                           # `if !self.nav.initialized { return cmd; }`

        # -- Fsw::step (main.rs:384-533)
        up_here, north, east, geo_alt = up_and_ned(pos_n + org_v)
        alt = xp.where(radar_alt_n >= 0.0, radar_alt_n, geo_alt)            # Navigator::altitude
        speed = xp.linalg.norm(vel_n)
        set_pad = ready & (phase < 0.5) & (pad_set < 0.5)                   # the pad frame, once (main.rs:405-413)
        az = xp.deg2rad(params[P["azimuth_deg"]])
        up_pad = xp.where(set_pad, up_here, fsw_frame[:3])
        track = xp.where(set_pad, normalize(north * xp.cos(az) + east * xp.sin(az)), fsw_frame[3:])
        t_liftoff_n = xp.where(ready & (t_liftoff < 0.0) & (xp.dot(vel_n, up_here) > 1.0), t, t_liftoff)
        u_ascent = params[P["ascent_throttle"]]
        in_pad, in_rise = phase < 0.5, (phase > 0.5) & (phase < 1.5)
        in_kick, in_turn = (phase > 1.5) & (phase < 2.5), (phase > 2.5) & (phase < 3.5)
        powered = phase < 3.5
        # PitchKick: ramp the nose from vertical toward the track azimuth
        f_kick = xp.clip((t - phase_t0) / params[P["kick_ramp_s"]], 0.0, 1.0)
        ang = f_kick * xp.deg2rad(params[P["kick_deg"]])
        dir_kick = normalize(up_pad * xp.cos(ang) + track * xp.sin(ang))
        # GravityTurn: the recorded profile's flight-path angle with an altitude trim, speed closed by throttle
        t_ref = t - t_liftoff_n
        v_ref = xp.interp(t_ref, prof_t, prof_speed)
        gamma_ref = xp.arcsin(xp.clip(xp.interp(t_ref, prof_t, prof_vspeed) / xp.maximum(v_ref, 30.0), -1.0, 1.0))
        alt_err = xp.interp(t_ref, prof_t, prof_alt) - alt
        gamma_cmd = xp.clip(gamma_ref + xp.clip(alt_err * 2.0e-4, -0.12, 0.12), 0.0, 1.55)
        u_prof = xp.clip(u_ascent + (v_ref - speed) * 2.0e-3, 0.62, 1.0)
        # ... and the parametric pitch program it falls back to until the profile clock runs (t_liftoff still unset)
        v0 = 90.0
        f_turn = xp.clip((speed - v0) / (params[P["meco_speed_mps"]] - v0), 0.0, 1.0)
        gamma_par = xp.deg2rad(90.0 - (90.0 - params[P["meco_fpa_deg"]]) * xp.power(f_turn, params[P["pitch_exp"]]))
        on_profile = t_liftoff_n >= 0.0
        gamma = xp.where(on_profile, gamma_cmd, gamma_par)
        dir_turn = normalize(up_here * xp.sin(gamma) + track * xp.cos(gamma))
        u = xp.where(on_profile, u_prof, u_ascent)
        qbar_est = 0.5 * fsw_density(xp, alt) * speed * speed               # throttle bucket through Max-Q
        u = xp.where((qbar_est > params[P["bucket_q_on_pa"]]) & (speed < 500.0), xp.minimum(u, params[P["bucket_throttle"]]), u)
        a_meas = xp.linalg.norm(imu_accel)                                  # ~3.5 g limit toward MECO
        u_turn = xp.where(a_meas > 34.0, xp.maximum(u * 34.0 / xp.maximum(a_meas, 1e-9), THROTTLE_MIN), u)
        dir_meco = normalize(vel_n)
        direction = xp.where(in_kick, dir_kick, xp.where(in_turn, dir_turn, xp.where(powered, up_pad, dir_meco)))
        attitude = xp.where(ready, quat_between_x(xp, direction), nav_att)  # an uninitialised navigator answers with its attitude
        meco_now = ready & in_turn & (speed >= params[P["meco_speed_mps"]])
        light = ready & in_pad & (t >= 0.2)
        throttle = xp.where(in_pad, xp.where(light, u_ascent, 0.0),
                            xp.where(in_rise | in_kick, u_ascent, xp.where(in_turn & ~meco_now, u_turn, 0.0)))
        throttle = xp.where(ready, throttle, 0.0)
        main_open = xp.where(ready & powered, 1.0, 0.0)
        # valves: helium pressurisation always; the purge bit is decided BEFORE this step's cutoff moves the deadline
        valve_cmd = xp.array([1.0, 0.0, 1.0, 0.0, main_open, main_open, main_open, xp.where(t < purge_until, 1.0, 0.0)])
        purge_until_n = xp.where(meco_now, t + 5.0, purge_until)            # cutoff_with_purge
        to_rise = light
        to_kick = ready & in_rise & (t >= params[P["kick_start_s"]])
        to_turn = ready & in_kick & (f_kick >= 1.0) & (speed > 80.0)
        to_flip = ready & (phase > 3.5) & (phase < 4.5) & (t - phase_t0 > 3.0)   # Meco -> Flip: the ascent is over (not flown further)
        phase_n = xp.where(to_rise, PHASE_VERTICAL_RISE, xp.where(to_kick, PHASE_PITCH_KICK, xp.where(
            to_turn, PHASE_GRAVITY_TURN, xp.where(meco_now, PHASE_MECO, xp.where(to_flip, PHASE_FLIP, phase)))))
        changed = to_rise | to_kick | to_turn | meco_now | to_flip
        tvc_on = xp.where(ready & powered, 1.0, 0.0)
        rcs_on = xp.where(ready & ~powered, 1.0, 0.0)
        return {"engine_cmd": xp.ones(N_ENGINES) * throttle, "valve_cmd": valve_cmd, "attitude_setpoint": attitude,
                "fin_cmd": xp.zeros(3), "ctrl_enable": xp.array([tvc_on, rcs_on]), "fsw_phase": phase,
                "nav_pos": pos_n, "nav_vel": vel_n, "nav_att": att_nx,
                "nav_aux": xp.array([xp.where(ready, 1.0, 0.0), last_gps_n, last_t_n, radar_alt_n]),
                "fsw_frame": xp.concatenate([up_pad, track]),
                "fsw_state": xp.array([phase_n, xp.where(changed, t, phase_t0), purge_until_n, xp.where(meco_now, 1.0, meco),
                                       t_liftoff_n, xp.where(set_pad, 1.0, pad_set)])}

    pre = [attitude_control, valve_dynamics, tvc_actuators, fin_actuators, engine_dynamics, mass_props, tank_dynamics,
           rcs_dynamics, engine_wrench_sys, wind_model, aero_dynamics]       # propulsion_systems, sim.py:1433-1458 (no legs)
    post = [pad_clamp, derive_geodetic_telemetry]
    if scripted is not None:
        @dsl.system
        def script(tick):
            out = scripted(xp, tick * dt)
            if isinstance(out, dict):     # any of the flight software's command columns (sim.py:97-137,285-300)
                return out
            eng, valves = out
            return {"engine_cmd": eng, "valve_cmd": valves}
        pre = [script] + pre
    elif fsw:
        post = post + [imu_model, gps_model, radar_altimeter_model, pressure_transducers, ascent_metrics_latch, fsw_ascent]
    return dsl.Program(pre, gravity_and_frame_forces | apply_body_wrenches, post)


def passive_effector(origin: Optional[Sequence[float]] = None) -> dsl.Effector:
    """build_passive (sim.py:1341-1378): gravitation + frame forces only — the plant of the reference's verification
    ladder (test_ladder.py)."""
    xp = dsl.np
    org = tuple(float(v) for v in (origin if origin is not None else (0.0, 0.0, 0.0)))

    @dsl.effector
    def gravity_and_frame_forces(force, inertia, pos, vel):
        r = pos.linear() + xp.array(org)
        accel = gravity_accel(xp, r) + frame_accel(xp, r, vel.linear())
        return force + dsl.SpatialForce(linear=accel * inertia.mass())
    return gravity_and_frame_forces


# ---- initial state and campaign ----------------------------------------------------------------------------------------------

COLUMN_WIDTHS = dict(engine_cmd=9, valve_cmd=8, engine_spool=9, engine_armed=9, teateb_charges=9, valve_state=8,
                     thrust_total=1, mdot_total=1, propellant_lox=1, propellant_rp1=1, tank_pressure_lox=1,
                     tank_pressure_rp1=1, inlet_pressure_lox=1, inlet_pressure_rp1=1, cg_station=1, axial_specific_force=1,
                     qbar=1, mach=1, tvc_cmd=2, tvc_state=2, rcs_torque_cmd=3, aero_wrench=6, engine_wrench=6,
                     attitude_setpoint=4, ctrl_enable=2, fsw_phase=1, upper_mass=1, lifted=1, liftoff_time=1,
                     altitude_geodetic=1, ground_speed=1, params=16, fsw_state=6, fsw_frame=6, ascent_metrics=8,
                     nav_pos=3, nav_vel=3, nav_att=4, nav_aux=4, sensor_tick=1, imu_accel=3, imu_gyro=3, gps_timer=1, gps_pos=3,
                     gps_vel=3, gps_count=1, radar_timer=1, radar_range=1, radar_count=1, pressure_meas=4,
                     fin_cmd=3, fin_state=4, fin_wrench=6, rcs_levels=8, rcs_wrench=6, nitrogen_kg=1, wind_ecef=3, wind_ned=3)


def initial_columns(params: np.ndarray, origin: Optional[Sequence[float]] = None, upper_kg: float = UPPER_KG,
                    init_pos_ecef=None, init_vel_ecef=None, init_attitude=None) -> Dict[str, np.ndarray]:
    """build_powered's spawn (sim.py:1384-1431,1461-1510) for `n = len(params)` rollouts: the booster on the pad, upright,
    tanks loaded per rollout."""
    params = np.asarray(params, dtype=np.float64).reshape(-1, len(PARAM_NAMES))
    n = params.shape[0]
    org = np.zeros(3) if origin is None else np.asarray(origin, dtype=np.float64)
    r0 = pad_ecef() if init_pos_ecef is None else np.asarray(init_pos_ecef, dtype=np.float64)
    att = upright_attitude() if init_attitude is None else np.asarray(init_attitude, dtype=np.float64)
    on_pad = float(np.linalg.norm(r0 - pad_ecef())) < 100.0
    lox, rp1 = params[:, P["lox_kg"]], params[:, P["rp1_kg"]]
    mass, cg, idiag = stack_mass_props(np, lox, rp1, upper_kg)
    cols = {k: np.zeros((n, w)) for k, w in COLUMN_WIDTHS.items()}
    cols["world_pos"] = np.tile(np.concatenate([att, r0 - org]), (n, 1))
    cols["world_vel"] = np.zeros((n, 6))
    if init_vel_ecef is not None:
        cols["world_vel"][:, 3:] = np.asarray(init_vel_ecef, dtype=np.float64)
    cols["inertia"] = np.zeros((n, 7))
    cols["inertia"][:, :3] = np.stack([np.broadcast_to(x, (n,)) for x in idiag], axis=1)
    cols["inertia"][:, 6] = mass
    cols["teateb_charges"][:] = [4.0] * RELIGHT_CAPABLE_ENGINES + [1.0] * (N_ENGINES - RELIGHT_CAPABLE_ENGINES)
    cols["propellant_lox"][:, 0], cols["propellant_rp1"][:, 0] = lox, rp1
    for k in ("tank_pressure_lox", "tank_pressure_rp1", "inlet_pressure_lox", "inlet_pressure_rp1"):
        cols[k][:] = TANK_P_NOM_PA
    cols["cg_station"][:] = DRY_CG_STATION_M
    cols["attitude_setpoint"][:] = upright_attitude()
    cols["upper_mass"][:] = upper_kg
    cols["nitrogen_kg"][:] = N2_INITIAL_KG
    cols["lifted"][:] = 0.0 if on_pad else 1.0
    cols["radar_range"][:] = -1.0                                           # sensors.py:119 / Navigator::new, Fsw::new (main.rs:214-225,355-381)
    cols["nav_att"][:] = [0.0, 0.0, 0.0, 1.0]
    cols["nav_aux"][:] = [0.0, 0.0, 0.0, -1.0]
    cols["fsw_state"][:] = [PHASE_PAD_PRESS, 0.0, -1.0, 0.0, -1.0, 0.0]
    cols["fsw_frame"][:] = [0.0, 0.0, 1.0, 1.0, 0.0, 0.0]
    cols["params"][:] = params
    return cols


def default_param_row() -> np.ndarray:
    return np.array([DEFAULT_PARAMS[k] for k in PARAM_NAMES])


def sample_params(n: int, seed: int = SPEC_SEED) -> np.ndarray:
    """spec.toml's LHS plan for the ascent variables through the same sampler as `elodin monte-carlo plan`
    (elodin_amd.monte_carlo, byte-identical to sample.py:84-151); the remaining columns keep their calibrated default."""
    from .. import monte_carlo as mc
    plan = mc.materialize({"monte_carlo": {"n_samples": n, "seed": seed, "method": "lhs", "variables": {
        k: {"dist": "uniform", "min": lo, "max": hi} for k, (lo, hi) in SPEC_RANGES.items()}}})
    return plan.table(PARAM_NAMES, DEFAULT_PARAMS)


ASCENT_TICKS = 180_000            # 180 s of flight at 1 kHz: every rollout of spec.toml's ranges reaches MECO by then (latest ~T+170 s)


_PROGRAMS: Dict[tuple, "dsl.Program"] = {}


class AscentExec:
    """One block of ascent rollouts on one GPU: rows = rollouts, the closed loop runs as a generated program
    (sixdof_set_custom_pipe) with `ticks_per_launch` ticks per launch and all state in registers in between."""

    def __init__(self, params: np.ndarray, *, dtype=np.float64, local_origin: Optional[bool] = None,
                 ticks_per_launch: int = 1000, device: int = 0, fsw: bool = True, scripted=None, columns=None,
                 fast_math: bool = False):
        from .. import _lib as L
        from ..exec import HipExec
        dtype = np.dtype(dtype)
        # f32 state cannot hold ECEF metres (0.5 m ulp): integrate pad-relative coordinates instead
        local = (dtype == np.float32) if local_origin is None else bool(local_origin)
        self.origin = pad_ecef() if local else np.zeros(3)
        # the f32 campaign build converts ECEF -> geodetic without angles (geodetic_sincos: the same recurrence, a fifth of
        # the instructions); an f64 executor flies the reference's arithmetic operation for operation
        key = (tuple(float(v) for v in self.origin) if local else None, bool(fsw), bool(fast_math))
        if scripted is None and key in _PROGRAMS:      # one program object per configuration: HipExec keeps its trace and object
            self.program = _PROGRAMS[key]
        else:
            self.program = build_program(origin=self.origin if local else None, fsw=fsw, scripted=scripted,
                                         algebraic_geodesy=bool(fast_math))
            if scripted is None:
                _PROGRAMS[key] = self.program
        cols = initial_columns(params, origin=self.origin) if columns is None else dict(columns)
        body = {k: cols.pop(k) for k in ("world_pos", "world_vel", "inertia")}
        # campaign builds (fast math) put the expensive arm of a `where` nobody else needs behind a wave-level branch
        # (codegen guarded selects: the GPS fix's noise draws on one tick in forty ...): same values, -2.7 % per tick
        # (profiles/r04_falcon9_guard_ab.txt); the precise builds keep plain selects unless SIXDOF_GUARD_SELECTS says otherwise
        self.hip = HipExec(body["world_pos"], body["world_vel"], body["inertia"], integrator=L.SEMI_IMPLICIT, dtype=dtype,
                           simulation_time_step=SIM_TIME_STEP, effectors=self.program, columns=cols,
                           ticks_per_launch=ticks_per_launch, device=device, fast_math=fast_math,
                           guard_selects=True if fast_math else None,
                           reuse_trace=scripted is None)     # _PROGRAMS' objects are built here from fixed code: nothing a trace reads changes

    def run(self, ticks: int):
        return self.hip.run(ticks)

    def column(self, name: str) -> np.ndarray:
        if name in ("world_pos", "world_vel", "world_accel", "force", "inertia"):
            return getattr(self.hip, name)
        return self.hip._aux[name]

    @property
    def ecef(self) -> np.ndarray:
        return self.hip.world_pos[:, 4:].astype(np.float64) + self.origin

    @property
    def result(self) -> np.ndarray:
        """[n, 8] METRIC_NAMES."""
        return np.asarray(self.hip._aux["ascent_metrics"], dtype=np.float64)

    def close(self):
        self.hip.close()


def prebuild() -> list:
    """Generate + compile the campaign programs ahead of time (f32 pad-relative, f64 ECEF): __graft_entry__.build() calls
    this so a GPU box finds them in elodin_amd/_jit instead of running hipcc at first use."""
    from .. import codegen
    out = []
    cols = initial_columns(default_param_row()[None, :])
    widths = {k: v.shape[1] for k, v in cols.items()}
    for dtype, origin, fast, soa in (("float32", pad_ecef(), True, False), ("float32", pad_ecef(), False, False), ("float64", None, False, False),
                                     ("float64", pad_ecef(), False, False),
                                     # campaign-size executors (>= codegen.COLUMN_SOA_MIN_ROWS rollouts): element-major columns
                                     ("float32", pad_ecef(), True, True), ("float64", pad_ecef(), False, True)):
        out.append(codegen.build(build_program(origin=origin, algebraic_geodesy=fast).trace(widths), dtype, 1, fast_math=fast, column_soa=soa,
                                 guard_selects=True if fast else None))
    return out


last_campaign_phases: Dict[str, float] = {}      # wall seconds of the last run_campaign on this rank, by phase


def run_campaign(plan_table: Optional[np.ndarray], n_runs: int, n_ticks: int = ASCENT_TICKS, *, dtype=np.float32,
                 ticks_per_launch: int = 1000, device: int = 0, comm_device="cpu", make_exec=None,
                 fast_math: Optional[bool] = None, comm=None) -> np.ndarray:
    """One ascent campaign across the ranks of the current torch.distributed group (or one process): rank 0's plan table
    ([n_runs, 16], sample_params) is broadcast, every rank flies its contiguous block of run ids with no per-step
    exchange, result rows are gathered back in run-id order (same scheme as models/apollo.run_campaign)."""
    from .. import shard
    import torch.distributed as dist
    if comm is not None:      # shard.CapiComm: the C ABI's RCCL collectives instead of torch.distributed
        world, rank = comm.world, comm.rank
        table = comm.broadcast_table(plan_table, (n_runs, len(PARAM_NAMES)))
    else:
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        rank = dist.get_rank() if world > 1 else 0
        table = shard.broadcast_table(plan_table, (n_runs, len(PARAM_NAMES)), device=comm_device)
    import time
    t = [time.perf_counter()]
    mark = lambda: t.append(time.perf_counter())
    lo, hi = shard.shard_range(n_runs, world, rank)
    if make_exec is None:
        fast = (np.dtype(dtype) == np.float32) if fast_math is None else bool(fast_math)   # the fast-math f32 build: 4.2x
        # faster than plain f32 (3.8 vs 15.8 us per tick), and its deviation from the f64 flight is plain f32's to two digits
        # in every campaign metric (tools/falcon9_fastmath.py, profiles/r04_falcon9_fastmath.txt)
        make_exec = lambda block, first_row: AscentExec(block, dtype=dtype, ticks_per_launch=ticks_per_launch, device=device,
                                                         fast_math=fast)
    ex = make_exec(table[lo:hi], lo)
    mark()
    ex.run(n_ticks)
    mark()
    local = np.ascontiguousarray(ex.result)
    if hasattr(ex, "close"):
        ex.close()
    out = comm.gather_rows(local, n_runs) if comm is not None else shard.gather_rows(local, n_runs, device=comm_device)
    mark()
    last_campaign_phases.update(build_and_upload_s=t[1] - t[0], flight_and_download_s=t[2] - t[1], gather_s=t[3] - t[2])
    return out
