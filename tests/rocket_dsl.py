"""The reference's rocket example (examples/rocket/main.py:297-537) written against elodin_amd.dsl — the same thirteen
systems in the same pipe order in front of six_dof(RK4) with the same three effectors, as user code for the system front-end.
What differs is spelling only: `jax.numpy` is `dsl.np`; the 480 x 3 sample buffer is a `dsl.Window` (push instead of
concatenate, window.scan instead of lax.scan over the array); `map_coordinates(order=1, mode="nearest")` over the static
3 x 5 x 4 coefficient grid is written as the multilinear interpolation it is; `tick` is the value the reference's
SimulationTick holds while the tick runs (the golden thrust column pins it: row k is interp(k * dt)).
TEST INFRASTRUCTURE (the golden-CSV tests run it on the CPU walker and on the GPU)."""
import math

import numpy as np

from elodin_amd import dsl

np_ = dsl.np
la = dsl.np.linalg
lax = dsl.lax

SIM_TIME_STEP = 1.0 / 120.0                                        # main.py:11-14
LP_SAMPLE_FREQ = round(1.0 / SIM_TIME_STEP)
LP_BUFFER_SIZE = LP_SAMPLE_FREQ * 4
LP_CUTOFF_FREQ = 1
THRUST_VECTOR_BODY = np_.array([-1.0, 0.0, 0.0])                   # main.py:16-21
A_REF = 24.89130 / 100 ** 2
L_REF = 5.43400 / 100
XMC = 0.40387
PITCH_PID = [1.1, 0.8, 3.8]

# main.py:150-156 (aero_df), columns Mach, Alphac, Delta, CmR, CA, CZR — 3 Mach x 5 Delta x 4 Alphac
_MACH = [0.1, 0.5, 0.9]
_DELTA = [-40.0, -20.0, 0.0, 20.0, 40.0]
_ALPHAC = [0.0, 5.0, 10.0, 15.0]
_CMR = [-5.997, -6.905, -8.235, -10.83, -5.315, -6.008, -5.918, -5.714, 0.0, 1.313, 2.335, 0.4163, 5.315, 3.642, 2.977, 1.061, 5.997, 5.372, 4.191, 1.882, -7.269, -8.373, -9.873, -12.93, -6.323, -7.255, -7.14, -6.846, 0.0, 1.486, 2.681, 0.445, 6.323, 4.263, 3.463, 1.222, 7.269, 6.396, 4.963, 2.27, -11.53, -12.49, -13.88, -15.71, -9.056, -8.891, -8.448, -8.155, 0.0, 1.921, 3.144, 1.169, 9.056, 8.419, 7.126, 4.228, 11.53, 10.14, 8.19, 4.94]   # noqa: E501
_CA = [1.121, 1.028, 0.9495, 0.9803, 0.6405, 0.5852, 0.4342, 0.217, 0.2942, 0.2873, 0.2591, 0.2032, 0.6405, 0.5988, 0.635, 0.6333, 1.121, 1.215, 1.246, 1.267, 1.242, 1.137, 1.051, 1.095, 0.6902, 0.6278, 0.4588, 0.2184, 0.2924, 0.2856, 0.2577, 0.2025, 0.6902, 0.6434, 0.6895, 0.6967, 1.242, 1.351, 1.392, 1.425, 1.851, 1.747, 1.621, 1.48, 0.9888, 0.8509, 0.658, 0.4269, 0.448, 0.4446, 0.4345, 0.418, 0.9888, 1.06, 1.111, 1.154, 1.851, 1.961, 2.03, 2.098]   # noqa: E501
_CZR = [-1.092, -0.3878, 0.3984, 1.141, -1.141, -0.4069, 0.7324, 2.176, 0.0, 1.061, 2.368, 3.494, 1.141, 1.561, 2.483, 3.64, 1.092, 1.789, 2.577, 3.68, -1.191, -0.4161, 0.4355, 1.252, -1.274, -0.4526, 0.8073, 2.408, 0.0, 1.178, 2.63, 3.88, 1.274, 1.736, 2.755, 4.043, 1.191, 1.973, 2.844, 4.07, -1.609, -0.8494, 0.1373, 1.323, -1.639, -0.5395, 0.9159, 2.704, 0.0, 1.304, 2.894, 4.443, 1.639, 2.532, 3.576, 4.981, 1.609, 2.483, 3.481, 4.811]   # noqa: E501
# aero_interp_table (main.py:247-262): [coef][mach][delta][alphac]; the rows of aero_df are already in that order
AERO = np.array([_CMR, _CA, _CZR]).reshape(3, len(_MACH), len(_DELTA), len(_ALPHAC))

THRUST_TIME = [0.01] + [round(0.67 + 0.66 * k, 2) for k in range(48)] + [32.15]      # main.py:158-161
THRUST_N = [322.148] + [88.426] * 48 + [0.0]


def to_coord(series, val):                                          # main.py:266-270
    s_min, s_max, s_count = min(series), max(series), len(set(series))
    return (val - s_min) * (s_count - 1) / max(s_max - s_min, 1e-06)


def map_coordinates_linear(grid: np.ndarray, coords):
    """jax.scipy.ndimage.map_coordinates(grid, coords, order=1, mode="nearest") for a static 3-D grid: per axis the two
    neighbours floor(c), floor(c)+1 with weights 1-(c-floor(c)), c-floor(c), indices clipped to the grid; the product of
    the per-axis weights over the eight corners.  The grid is a constant, so each corner value is picked with selects on
    the (traced) lower index."""
    lows, fracs = [], []
    for c in coords:
        lo = np_.floor(c)
        lows.append(lo)
        fracs.append(c - lo)

    def pick(axis_len, index):          # one-hot over the clipped integer index
        idx = np_.clip(index, 0.0, float(axis_len - 1))
        return [np_.where(np_.equal(idx, float(k)), 1.0, 0.0) for k in range(axis_len)]
    total = 0.0
    for corner in range(8):
        bits = [(corner >> a) & 1 for a in range(3)]
        weight = 1.0
        hots = []
        for a in range(3):
            weight = weight * (fracs[a] if bits[a] else 1.0 - fracs[a])
            hots.append(pick(grid.shape[a], lows[a] + float(bits[a])))
        value = 0.0
        for i in range(grid.shape[0]):
            for j in range(grid.shape[1]):
                row = 0.0
                for k in range(grid.shape[2]):
                    row = row + hots[2][k] * float(grid[i, j, k])
                value = value + hots[0][i] * hots[1][j] * row
        total = total + weight * value
    return total


def quat_from_vecs(v1, v2):                                         # main.py:237-244
    v1 = v1 / la.norm(v1)
    v2 = v2 / la.norm(v2)
    n = np_.cross(v1, v2)
    w = np_.dot(v2, v2) * np_.dot(v1, v1) + np_.dot(v1, v2)
    return dsl.Quaternion(np_.concatenate([n, w])).normalize()


def euler_to_quat(angles_deg):                                      # main.py:205-219, host side (spawn data)
    roll, pitch, yaw = np.deg2rad(angles_deg)
    cr, sr, cp, sp, cy, sy = np.cos(roll * 0.5), np.sin(roll * 0.5), np.cos(pitch * 0.5), np.sin(pitch * 0.5), np.cos(yaw * 0.5), np.sin(yaw * 0.5)
    return np.array([sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy])


# ---- systems (main.py:297-537), pipe order of main.py:556-570 ----------------------------------------------------------

@dsl.system(wind=3, v_body=3)
def compute_v_body(pos, vel, wind):
    return {"v_body": pos.angular().inverse() @ (vel.linear() - wind)}


@dsl.system(wind=3)
def mach(pos, vel, wind):
    h = [0.0, 11_000.0, 20_000.0, 32_000.0, 47_000.0, 51_000.0, 71_000.0, 84_852.0]
    T = [15.0, -56.5, -56.5, -44.5, -2.5, -2.5, -58.5, -86.2]
    d = [1.225, 0.3639, 0.0880, 0.0132, 0.0014, 0.0009, 0.0001, 0.]
    altitude = pos.linear()[2]
    temperature = np_.interp(altitude, h, T) + 273.15
    density = np_.interp(altitude, h, d)
    speed_of_sound = np_.sqrt(1.4 * 287.05 * temperature)
    local_flow_velocity = la.norm(vel.linear() - wind)
    dynamic_pressure = 0.5 * density * local_flow_velocity ** 2
    return {"mach": local_flow_velocity / speed_of_sound, "dynamic_pressure": np_.maximum(dynamic_pressure, 1e-6)}


@dsl.system(wind=3)
def angle_of_attack(pos, vel, wind):
    u = pos.angular().inverse() @ (vel.linear() - wind)
    c = np_.dot(u, THRUST_VECTOR_BODY) / np_.maximum(la.norm(u), 1e-6)
    return {"angle_of_attack": np_.rad2deg(np_.arccos(c)) * -np_.sign(u[2])}


@dsl.system(accel_setpoint=2, accel_setpoint_smooth=2)
def accel_setpoint_smooth(accel_setpoint, accel_setpoint_smooth):
    a, a_s = accel_setpoint, accel_setpoint_smooth
    return {"accel_setpoint_smooth": a_s + (a - a_s) * math.exp(-0.5 * SIM_TIME_STEP)}


@dsl.system(v_rel_accel=3)
def v_rel_accel(vel, accel):
    v = lax.cond(la.norm(vel.linear()) < 1e-6, lambda _: THRUST_VECTOR_BODY, lambda _: vel.linear(), operand=None)
    v_rot = quat_from_vecs(THRUST_VECTOR_BODY, v)
    return {"v_rel_accel": v_rot.inverse() @ accel.linear()}


@dsl.system(v_rel_accel=3, v_rel_accel_buffer=(LP_BUFFER_SIZE, 3))
def v_rel_accel_buffer(v_rel_accel, v_rel_accel_buffer):
    return {"v_rel_accel_buffer": v_rel_accel_buffer.push(v_rel_accel)}          # concatenate((buffer[1:], a_rel))


def second_order_butterworth_last(signal: dsl.Window, f_sampling, f_cutoff):
    """main.py:164-202, method="forward", last output only (what v_rel_accel_filtered takes: `[...][-1]`)."""
    ff = f_cutoff / f_sampling
    ita = 1.0 / math.tan(math.pi * ff)
    q = math.sqrt(2.0)
    b0 = 1.0 / (1.0 + q * ita + ita ** 2)
    b1 = 2 * b0
    b2 = b0
    a1 = 2.0 * (ita ** 2 - 1.0) * b0
    a2 = -(1.0 - q * ita + ita ** 2) * b0

    def f(carry, x_i):
        x_im1, x_im2, y_im1, y_im2 = carry
        y_i = b0 * x_i + b1 * x_im1 + b2 * x_im2 + a1 * y_im1 + a2 * y_im2
        return (x_i, x_im1, y_i, y_im1), y_i
    init = (signal[1], signal[0]) * 2
    carry = signal.scan(f, init, start=2)
    return carry[2]                     # y of the last step


@dsl.system(v_rel_accel_buffer=(LP_BUFFER_SIZE, 3), v_rel_accel_filtered=3)
def v_rel_accel_filtered(v_rel_accel_buffer):
    return {"v_rel_accel_filtered": second_order_butterworth_last(v_rel_accel_buffer, LP_SAMPLE_FREQ, LP_CUTOFF_FREQ)}


@dsl.system(accel_setpoint_smooth=2, v_rel_accel_filtered=3, pitch_pid_state=3)
def pitch_pid_state(accel_setpoint_smooth, v_rel_accel_filtered, pitch_pid_state):
    s = pitch_pid_state
    e = v_rel_accel_filtered[2] - accel_setpoint_smooth[0]
    i = np_.clip(s[1] + e * SIM_TIME_STEP * 2, -2.0, 2.0)
    d = e - s[0]
    return {"pitch_pid_state": np_.array([e, i, d])}


@dsl.system(pitch_pid=3, pitch_pid_state=3)
def pitch_pid_control(pitch_pid, pitch_pid_state):
    Kp, Ki, Kd = pitch_pid
    e, i, d = pitch_pid_state
    return {"fin_control": (Kp * e + Ki * i + Kd * d) * SIM_TIME_STEP}


@dsl.system
def fin_control(fin_deflect, fin_control, mach):
    fc = np_.clip(fin_control / (0.1 + mach), -0.2, 0.2)
    return {"fin_deflect": np_.clip(fin_deflect + fc, -40.0, 40.0)}


@dsl.system(aero_coefs=6)
def aero_coefs(mach, angle_of_attack, fin_deflect, fin_control_trim):
    fin_trim = fin_control_trim
    effective_fin_deflect = np_.clip(fin_deflect + fin_trim, -40.0, 40.0)
    aoa_sign = lax.cond(np_.abs(angle_of_attack) < 1e-6, lambda _: 1.0, lambda _: np_.sign(angle_of_attack), operand=None)
    effective_fin_deflect = effective_fin_deflect * aoa_sign
    coords = [to_coord(_MACH, mach), to_coord(_DELTA, effective_fin_deflect), to_coord(_ALPHAC, np_.abs(angle_of_attack))]
    coefs = [map_coordinates_linear(AERO[c], coords) for c in range(3)]
    cl = fin_trim * 0.1
    return {"aero_coefs": np_.array([cl, 0.0, coefs[0] * aoa_sign, coefs[1], coefs[2] * aoa_sign, 0.0])}


@dsl.system(aero_coefs=6, aero_force=6)
def aero_forces(aero_coefs, center_of_gravity, dynamic_pressure):
    Cl, CnR, CmR, CA, CZR, CYR = aero_coefs
    xcg, q = center_of_gravity, dynamic_pressure
    CmR = CmR - CZR * (xcg - XMC) / L_REF
    CnR = CnR - CYR * (xcg - XMC) / L_REF
    f_aero_linear = np_.array([CA, CYR, CZR]) * q * A_REF
    f_aero_torque = np_.array([Cl, -CmR, CnR]) * q * A_REF * L_REF
    return {"aero_force": np_.concatenate([f_aero_torque, f_aero_linear])}       # SpatialForce layout: torque, force


@dsl.system
def thrust(tick, rocket_motor):
    t = tick * SIM_TIME_STEP_NS                                       # tick[0] * dt[0] with the QUANTISED time step
    return {"thrust": np_.interp(t, THRUST_TIME, THRUST_N)}


SIM_TIME_STEP_NS = round(SIM_TIME_STEP * 1e9) / 1e9                   # what SimulationTimeStep holds (world.rs dt quantisation)


@dsl.effector
def gravity(force, inertia):
    return force + dsl.SpatialForce(linear=np_.array([0.0, 0.0, -9.81]) * inertia.mass())


@dsl.effector(thrust=1)
def apply_thrust(thrust, force, pos):
    return force + dsl.SpatialForce(linear=pos.angular() @ THRUST_VECTOR_BODY * thrust[0])   # effector columns arrive as vectors


@dsl.effector(aero_force=6)
def apply_aero_forces(pos, aero_force, force):
    q = pos.angular()
    return force + dsl.SpatialForce(torque=q @ aero_force[:3], linear=q @ aero_force[3:])


NON_EFFECTORS = [compute_v_body, mach, angle_of_attack, accel_setpoint_smooth, v_rel_accel, v_rel_accel_buffer,
                 v_rel_accel_filtered, pitch_pid_state, pitch_pid_control, fin_control, aero_coefs, aero_forces, thrust]
EFFECTORS = gravity | apply_thrust | apply_aero_forces


def program() -> dsl.Program:
    return dsl.Program(NON_EFFECTORS, EFFECTORS, [])


def spawn(n: int = 1):
    """main.py:540-553 + the Rocket archetype defaults (main.py:273-294), n identical rockets."""
    pos = np.tile(np.concatenate([euler_to_quat(np.array([0.0, 70.0, 0.0])), [0.0, 0.0, 1.0]]), (n, 1))
    vel = np.zeros((n, 6))
    inertia = np.tile(np.array([0.1, 1.0, 1.0, 0.0, 0.0, 0.0, 3.0]), (n, 1))
    z = lambda w: np.zeros((n, w))
    comps = {"angle_of_attack": z(1), "aero_coefs": z(6), "center_of_gravity": np.full((n, 1), 0.2), "mach": z(1),
             "dynamic_pressure": z(1), "aero_force": z(6), "wind": z(3), "rocket_motor": z(1), "fin_deflect": z(1),
             "fin_control": z(1), "fin_control_trim": z(1), "v_body": z(3), "v_rel_accel_buffer": z(LP_BUFFER_SIZE * 3),
             "v_rel_accel": z(3), "v_rel_accel_filtered": z(3), "pitch_pid": np.tile(np.array(PITCH_PID), (n, 1)),
             "pitch_pid_state": z(3), "accel_setpoint": z(2), "accel_setpoint_smooth": z(2), "thrust": z(1)}
    return pos, vel, inertia, comps
