// apollo_kernels.hip — Apollo-lander Monte-Carlo rollouts (BASELINE config 4) as rows of the entity axis.
//
// The reference runs ONE rollout per OS process (libs/monte-carlo/src/lib.rs:2083), each JIT-compiling
//   truth_playback | engine_response | attitude_control | mass_props | thrust_visualization
//     | six_dof(lunar_gravity | apply_main_thrust | apply_rcs_torque, SemiImplicit) | ground_contact
//     | derive_telemetry                                      examples/apollo-lander/sim.py:517-526
// and closing the loop through a sidecar guidance process over UDP, polled from post_step once per telemetry batch
//   examples/apollo-lander/controller/src/main.rs:188-262 (command), main.py:166-283 (post_step).
// Here a rollout is one lane: its whole state lives in VGPRs for `n_ticks` ticks, the guidance law
// runs in-line at the batch ends whose end_tick is a multiple of `guidance_period`, and a campaign of N rollouts is ceil(N/64) waves.
// Column layout: include/sixdof_apollo.h.  Visualisation-only systems are not modelled.
//
// The descent reference profile is identical for every rollout at a given tick, so the host
// interpolates it once per tick (same formula as reference.py:164-175) into `tick_refs[n_ticks][8]`
// and the kernel reads it with wave-uniform (scalar) loads instead of searching tables per lane.
//
// Bound: f64 VALU + transcendental (acos every tick; atan2/sin/cos/tan at guidance ticks); with
// n_ticks >= 100 the 944 B of per-rollout state traffic per launch is negligible.
#include "effectors.hpp"
#include "kernels.hpp"
#include "spatial.hpp"
#include "../../include/sixdof_apollo.h"

namespace sixdof {

namespace {

constexpr double G0 = 9.80665, LUNAR_GRAVITY = 1.622, R_MOON_M = 1737400.0;
constexpr double DPS_MAX_THRUST_N = 45040.0, DPS_MIN_THRUST_N = 4670.0;
constexpr double THROTTLE_MIN = DPS_MIN_THRUST_N / DPS_MAX_THRUST_N, THROTTLE_MAX = 1.0;
constexpr double RCS_ISP_S = 290.0, RCS_MOMENT_ARM_M = 2.0, RCS_LIMIT = 4.0 * 445.0 * 2.0;
constexpr double FOOTPAD_HEIGHT_M = 2.40, SIM_TIME_STEP = 1.0 / 120.0;
constexpr double kPi = 3.14159265358979323846;
// controller/src/main.rs:8-45
constexpr double C_MIN_THROTTLE = 4670.0 / 45040.0, C_FTP = 0.925, C_EROSION_MIN = 0.65;
constexpr double C_MAX_DESCENT = 120.0, C_MIN_DESCENT = 0.5, C_MIN_VACC = 0.05;
constexpr double C_TILT_BRAKING = 82.0, C_TILT_APPROACH = 30.0, C_BLEND_HI = 150.0, C_BLEND_LO = 40.0;
constexpr double C_HSPEED_GAIN = 0.25, C_POS_AUTH = 0.5, C_RATE_AUTH = 12.0, C_VFB_AUTH = 0.8, C_HFB_AUTH = 0.8;
constexpr double C_TERMINAL_ALT = 40.0;

__device__ __forceinline__ double clampd(double x, double lo, double hi) { return fmin(fmax(x, lo), hi); }

using V3 = Vec3<double>;
using Q = Quat<double>;

// Hamilton product (only needed for q^-1 (x) setpoint in attitude_control)
__device__ __forceinline__ Q qmul(Q l, Q r) {
    return {l.w * r.i + l.i * r.w + l.j * r.k - l.k * r.j, l.w * r.j - l.i * r.k + l.j * r.w + l.k * r.i,
            l.w * r.k + l.i * r.j - l.j * r.i + l.k * r.w, l.w * r.w - l.i * r.i - l.j * r.j - l.k * r.k};
}

__device__ __forceinline__ double pitch_deg(Q q) {
    return acos(clampd(1.0 - 2.0 * (q.i * q.i + q.j * q.j), -1.0, 1.0)) * (180.0 / kPi);
}

}  // namespace

// tick_refs row: ref_alt, ref_rate, |ref_pitch|, ref_hspeed, ref_downrange, ref_hdecel, reserved, reserved
__global__ __launch_bounds__(64) void apollo_rollout_kernel(const ApolloParams P) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n) return;
    double* gpos = P.pos + (size_t)i * 7;
    double* gvel = P.vel + (size_t)i * 6;
    double* gst = P.state + (size_t)i * APOLLO_N_STATE;
    const double* pr = P.params + (size_t)i * APOLLO_N_PARAMS;
    double* ggd = P.guidance + (size_t)i * APOLLO_N_GUIDANCE;
    double* gsc = P.score + (size_t)i * APOLLO_N_SCORE;

    Q q = {gpos[0], gpos[1], gpos[2], gpos[3]};
    V3 p = {gpos[4], gpos[5], gpos[6]};
    V3 om = {gvel[0], gvel[1], gvel[2]}, v = {gvel[3], gvel[4], gvel[5]};
    double throttle = gst[APOLLO_S_THROTTLE], throttle_cmd = gst[APOLLO_S_THROTTLE_CMD];
    Q setpoint = {gst[2], gst[3], gst[4], gst[5]};
    double prop = gst[APOLLO_S_PROPELLANT], rcs_prop = gst[APOLLO_S_RCS_PROPELLANT];
    double thrust = gst[APOLLO_S_THRUST];
    V3 torque = {gst[9], gst[10], gst[11]};
    double landed = gst[APOLLO_S_LANDED], td_speed = gst[APOLLO_S_TOUCHDOWN_SPEED], td_hspeed = gst[APOLLO_S_TOUCHDOWN_HSPEED];
    double pitch = gst[APOLLO_S_PITCH];
    double last_throttle = ggd[APOLLO_G_LAST_THROTTLE];
    Q last_att = {ggd[1], ggd[2], ggd[3], ggd[4]};
    double last_rate = ggd[APOLLO_G_LAST_RATE], latched = ggd[APOLLO_G_FTP_LATCHED], emitted = ggd[APOLLO_G_RESULT_EMITTED];
    double e_alt = gsc[0], e_pitch = gsc[1], e_n = gsc[2];

    // per-rollout constants (sim.py:233-256)
    const double dry_mass = pr[APOLLO_P_DRY_MASS];
    const double total_mass = dry_mass + pr[APOLLO_P_PROPELLANT] + pr[APOLLO_P_RCS_PROPELLANT];
    const double inv_total_mass = 1.0 / total_mass;
    const double thrust_scale = pr[APOLLO_P_THRUST_SCALE];
    const double inv_isp_g0 = 1.0 / (pr[APOLLO_P_ISP] * G0);
    const double ag = pr[APOLLO_P_ATTITUDE_GAIN] / 0.040;
    const V3 rcs_k = {4500.0 * ag, 5500.0 * ag, 4500.0 * ag};
    const V3 rcs_d = {19000.0, 21000.0, 19000.0};
    const V3 base_I = {78000.0, 72000.0, 45000.0};
    const V3 inv_base_I = {1.0 / 78000.0, 1.0 / 72000.0, 1.0 / 45000.0};
    const double alpha = fmin(fmax(pr[APOLLO_P_THROTTLE_RESPONSE_HZ] * SIM_TIME_STEP, 0.0), 1.0);
    const double gravity = LUNAR_GRAVITY * pr[APOLLO_P_GRAVITY_SCALE];
    const double track_gain = pr[APOLLO_P_TRACK_GAIN], vertical_gain = pr[APOLLO_P_VERTICAL_GAIN];
    const double position_gain = 0.01 * pr[APOLLO_P_HORIZONTAL_GAIN];
    const double inv_max_thrust = 1.0 / fmax(DPS_MAX_THRUST_N * thrust_scale, 1.0);
    const double dt = P.dt;

    V3 I_diag = {0, 0, 0}, inv_I = {0, 0, 0};
    double mass = 0.0, inv_m = 0.0;
    Spatial<double> A = {{0, 0, 0}, {0, 0, 0}}, Fw = {{0, 0, 0}, {0, 0, 0}};

    // tick % ticks_per_telemetry and (tick - 1) % guidance_period as wrapping counters: a run-time 64-bit modulo is ~100
    // scalar instructions, a third of a tick's issue slots
    if (P.n_ticks != 0) q = normalized(q);     // user-supplied initial attitudes need not be unit; q * v is scale-invariant
    uint32_t tel_phase = (uint32_t)((P.tick0 + 1) % P.ticks_per_telemetry);
    uint32_t gd_phase = (uint32_t)(P.tick0 % P.guidance_period);
    for (uint32_t k = 0; k < P.n_ticks; k++) {
        const uint64_t tick = P.tick0 + k + 1;
        const bool exchange = tel_phase == 0 || tick == P.max_ticks;
        const bool guidance_due = gd_phase == 0;
        tel_phase = tel_phase + 1 == P.ticks_per_telemetry ? 0 : tel_phase + 1;
        gd_phase = gd_phase + 1 == P.guidance_period ? 0 : gd_phase + 1;
        const double* ref = P.tick_refs + (size_t)k * 8;  // wave-uniform
        const bool is_landed = landed > 0.5;
        // engine_response (sim.py:334-343)
        {
            const double cmd = clampd(throttle_cmd, THROTTLE_MIN, THROTTLE_MAX);
            const double actual = throttle + (cmd - throttle) * alpha;
            throttle = (prop > 0.0 && !is_landed) ? actual : 0.0;
            thrust = throttle * DPS_MAX_THRUST_N * thrust_scale;
        }
        // attitude_control (sim.py:368-378).  q is unit here (normalised before the loop, and by every integration), so the
        // conjugate is the inverse.
        const Q qn = q;
        {
            const Q qc = {-qn.i, -qn.j, -qn.k, qn.w};
            const Q err = qmul(qc, setpoint);
            const double sign = err.w >= 0.0 ? 1.0 : -1.0;
            const V3 rate = rotate_inv(qn, om);
            V3 t = {sign * err.i * rcs_k.x - rate.x * rcs_d.x, sign * err.j * rcs_k.y - rate.y * rcs_d.y,
                    sign * err.k * rcs_k.z - rate.z * rcs_d.z};
            t = {clampd(t.x, -RCS_LIMIT, RCS_LIMIT), clampd(t.y, -RCS_LIMIT, RCS_LIMIT), clampd(t.z, -RCS_LIMIT, RCS_LIMIT)};
            torque = is_landed ? V3{0, 0, 0} : t;
        }
        // mass_props (sim.py:345-366)
        {
            const double dps_burn = thrust * inv_isp_g0 * SIM_TIME_STEP;
            const double rcs_burn = (fabs(torque.x) + fabs(torque.y) + fabs(torque.z)) * (1.0 / RCS_MOMENT_ARM_M) *
                                    (1.0 / (RCS_ISP_S * G0)) * SIM_TIME_STEP;
            prop = fmax(prop - dps_burn, 0.0);
            rcs_prop = fmax(rcs_prop - rcs_burn, 0.0);
            mass = dry_mass + prop + rcs_prop;
            const double sc = mass * inv_total_mass;
            I_diag = is_landed ? V3{1.0e9, 1.0e9, 1.0e9} : V3{base_I.x * sc, base_I.y * sc, base_I.z * sc};
            // one divide per tick: 1/m, and 1/I = (1/I_base) * (m_total / m)
            inv_m = recip(mass);      // v_rcp_f64 + two Newton steps (spatial.hpp): 5 instructions against an IEEE divide's 11
            const double inv_sc = total_mass * inv_m;
            inv_I = is_landed ? V3{1.0e-9, 1.0e-9, 1.0e-9}
                              : V3{inv_base_I.x * inv_sc, inv_base_I.y * inv_sc, inv_base_I.z * inv_sc};
        }
        // six_dof(lunar_gravity | apply_main_thrust | apply_rcs_torque), semi-implicit (sim.py:380-398,523)
        {
            const double g_eff = fmax(gravity - (v.x * v.x + v.y * v.y) * (1.0 / R_MOON_M), 0.0);
            V3 f = rotate(qn, V3{0.0, 0.0, thrust});
            f.z = fma(-g_eff, mass, f.z);
            // torque is given in the body frame: alpha = q * (tau_b / I)
            const V3 alpha_w = rotate(qn, hadamard(torque, inv_I));
            A.ang = alpha_w;
            A.lin = inv_m * f;
            Fw.ang = rotate(qn, torque);
            Fw.lin = f;
            om = axpy(dt, A.ang, om);
            v = axpy(dt, A.lin, v);
            q = integrate_world(q, dt * om);
            p = axpy(dt, v, p);
        }
        // ground_contact (sim.py:400-431)
        {
            const bool contact = p.z <= FOOTPAD_HEIGHT_M;
            const bool first = !is_landed && contact;
            const bool now = is_landed || contact;
            if (first) {   // rare: latch the impact speeds before the velocity is zeroed
                td_speed = fabs(v.z);
                td_hspeed = sqrt(v.x * v.x + v.y * v.y);
            }
            if (now) {
                p.z = FOOTPAD_HEIGHT_M;
                v = V3{0, 0, 0};
                om = V3{0, 0, 0};
            }
            landed = now ? 1.0 : 0.0;
        }
        const double altitude = p.z, vertical_speed = v.z;

        // ---- post_step (main.py:166-283), once per telemetry batch (impeller2_server.rs:553-678): the server loop runs
        // ticks_per_telemetry ticks, then calls post_step(end_tick) with end_tick = ticks completed - 1 — that is the
        // `tick` main.py sees (its t_s, its `tick % guidance_period_ticks`, its `tick >= max_ticks - 1`).  The last batch
        // of a run is cut short at max_ticks.  Wave-uniform condition.
        if (!exchange) continue;
        const uint64_t end_tick = tick - 1;
        const bool landed_now = landed > 0.5;
        // derive_telemetry (sim.py:433-444): pitch = acos(body_up.z); body_up.z = 1 - 2(qi^2 + qj^2).  A pure function of this
        // tick's attitude that only the exchange below (and the final store) reads: evaluated where it is read — the f64 acos
        // is a quarter of a tick's instructions, and two ticks in three have no exchange.
        pitch = pitch_deg(q);
        {
            const double da = altitude - ref[0], dp = pitch - ref[2];
            e_alt = fma(da, da, e_alt);
            e_pitch = fma(dp, dp, e_pitch);
            e_n += 1.0;
        }
        if (guidance_due && !landed_now) {  // end_tick % guidance_period == 0: wave-uniform; landed is per lane
            // controller/src/main.rs:188-262
            const double h_speed = sqrt(v.x * v.x + v.y * v.y);
            const double m_now = dry_mass + prop + rcs_prop;
            const double g_eff = fmax(gravity - h_speed * h_speed * (1.0 / R_MOON_M), 0.05 * gravity);
            const double rate_track = clampd(track_gain * (ref[0] - altitude), -C_RATE_AUTH, C_RATE_AUTH);
            const double rate_cmd = clampd(ref[1] + rate_track, -C_MAX_DESCENT, -C_MIN_DESCENT);
            const double vfb = clampd(vertical_gain * (rate_cmd - vertical_speed), -C_VFB_AUTH, C_VFB_AUTH);
            double az = fmax(g_eff + vfb, C_MIN_VACC);
            const double trim_fade = clampd((altitude - 30.0) * (1.0 / 120.0), 0.0, 1.0);
            const double trim_x = clampd(position_gain * (ref[4] - p.x), -C_POS_AUTH, C_POS_AUTH) * trim_fade;
            const double trim_y = clampd(position_gain * (-p.y), -C_POS_AUTH, C_POS_AUTH) * trim_fade;
            const bool terminal = altitude < C_TERMINAL_ALT;
            const double target_vx = terminal ? 0.0 : ref[3], target_decel = terminal ? 0.0 : ref[5];
            const double hfb = clampd(C_HSPEED_GAIN * (target_vx - v.x), -C_HFB_AUTH, C_HFB_AUTH);
            double ax = -target_decel + hfb + trim_x;
            double ay = clampd(C_HSPEED_GAIN * (-v.y), -C_HFB_AUTH, C_HFB_AUTH) + trim_y;
            const double blend = clampd((h_speed - C_BLEND_LO) * (1.0 / (C_BLEND_HI - C_BLEND_LO)), 0.0, 1.0);
            const double max_tilt = (C_TILT_APPROACH + (C_TILT_BRAKING - C_TILT_APPROACH) * blend) * (kPi / 180.0);
            const double ah = hypot(ax, ay);
            if (h_speed > C_BLEND_LO) {  // cap_tilt_preserve_magnitude (main.rs:128-142)
                if (!(ah < 1e-9) && !(atan2(ah, az) <= max_tilt)) {
                    const double mag = sqrt(ah * ah + az * az);
                    const double sh = mag * sin(max_tilt) / ah;
                    ax *= sh;
                    ay *= sh;
                    az = mag * cos(max_tilt);
                }
            } else {  // clamp_horizontal (main.rs:147-156)
                const double limit = fmax(az, C_MIN_VACC) * tan(C_TILT_APPROACH * (kPi / 180.0));
                if (!(ah <= limit || ah < 1e-9)) {
                    const double sh = limit / ah;
                    ax *= sh;
                    ay *= sh;
                }
            }
            const double thrust_required = m_now * sqrt(ax * ax + ay * ay + az * az);
            const double demand = clampd(thrust_required * inv_max_thrust, C_MIN_THROTTLE, C_FTP);
            bool lat = latched > 0.5;  // ThrottleLogic (main.rs:163-186)
            if (lat && demand < 0.60) lat = false;
            else if (!lat && demand > 0.80) lat = true;
            last_throttle = (demand <= C_EROSION_MIN && !lat) ? fmax(demand, C_MIN_THROTTLE) : (lat ? C_FTP : C_EROSION_MIN);
            latched = lat ? 1.0 : 0.0;
            last_rate = rate_cmd;
            // quat_from_body_z (main.rs:100-121)
            Q tq;
            {
                const double n = sqrt(ax * ax + ay * ay + az * az);
                const V3 d = n < 1e-9 ? V3{0, 0, 1} : V3{ax / n, ay / n, az / n};
                const double dot = clampd(d.z, -1.0, 1.0);
                if (dot < -0.999999) tq = Q{1, 0, 0, 0};
                else tq = normalized(Q{-d.y, d.x, 0.0, 1.0 + dot});
            }
            // _slew_quat(last_attitude, target, 3 deg) (main.py:147-163)
            {
                const double nc = sqrt(last_att.i * last_att.i + last_att.j * last_att.j + last_att.k * last_att.k + last_att.w * last_att.w);
                const Q cur = nc < 1e-12 ? Q{0, 0, 0, 1} : Q{last_att.i / nc, last_att.j / nc, last_att.k / nc, last_att.w / nc};
                double dot = cur.i * tq.i + cur.j * tq.j + cur.k * tq.k + cur.w * tq.w;
                if (dot < 0.0) { tq = Q{-tq.i, -tq.j, -tq.k, -tq.w}; dot = -dot; }
                dot = fmin(fmax(dot, -1.0), 1.0);
                const double angle = 2.0 * acos(dot), max_angle = 3.0 * (kPi / 180.0);
                if (angle <= max_angle || angle < 1e-9) last_att = tq;
                else {
                    const double fr = max_angle / angle;
                    const Q bl = {(1.0 - fr) * cur.i + fr * tq.i, (1.0 - fr) * cur.j + fr * tq.j,
                                  (1.0 - fr) * cur.k + fr * tq.k, (1.0 - fr) * cur.w + fr * tq.w};
                    const double nb = sqrt(bl.i * bl.i + bl.j * bl.j + bl.k * bl.k + bl.w * bl.w);
                    last_att = nb < 1e-12 ? Q{0, 0, 0, 1} : Q{bl.i / nb, bl.j / nb, bl.k / nb, bl.w / nb};
                }
            }
        }
        throttle_cmd = last_throttle;
        setpoint = last_att;
        if (!(emitted > 0.5) && (landed_now || end_tick >= P.max_ticks - 1)) {  // main.py:240-272
            double* res = P.result + (size_t)i * APOLLO_N_RESULT;
            const double td = landed_now ? td_speed : fabs(vertical_speed);
            const double tdh = landed_now ? td_hspeed : sqrt(v.x * v.x + v.y * v.y);
            const double nn = fmax(e_n, 1.0);
            const double upright = cos(fabs(pitch) * (kPi / 180.0));
            res[APOLLO_R_TOUCHDOWN_SPEED] = td;
            res[APOLLO_R_HORIZONTAL_SPEED] = tdh;
            res[APOLLO_R_FUEL_REMAINING] = prop;
            res[APOLLO_R_RCS_FUEL_REMAINING] = rcs_prop;
            res[APOLLO_R_TRAJ_RMSE] = sqrt(e_alt / nn);
            res[APOLLO_R_PITCH_RMSE] = sqrt(e_pitch / nn);
            res[APOLLO_R_DOWNRANGE_MISS] = hypot(p.x, p.y);
            res[APOLLO_R_UPRIGHT_DOT] = upright;
            res[APOLLO_R_LANDED] = landed_now ? 1.0 : 0.0;
            res[APOLLO_R_SOFT_LANDING] = (landed_now && td <= 3.0 && tdh <= 1.0 && upright >= 0.94 && prop > 0.0) ? 1.0 : 0.0;
            res[APOLLO_R_TICK] = (double)end_tick;   // the tick main.py's post_step was called with
            emitted = 1.0;
        }
    }
    if (P.n_ticks == 0) return;
    pitch = pitch_deg(q);          // the column holds the last tick's value
    gpos[0] = q.i; gpos[1] = q.j; gpos[2] = q.k; gpos[3] = q.w; gpos[4] = p.x; gpos[5] = p.y; gpos[6] = p.z;
    gvel[0] = om.x; gvel[1] = om.y; gvel[2] = om.z; gvel[3] = v.x; gvel[4] = v.y; gvel[5] = v.z;
    double* ga = P.accel + (size_t)i * 6;
    ga[0] = A.ang.x; ga[1] = A.ang.y; ga[2] = A.ang.z; ga[3] = A.lin.x; ga[4] = A.lin.y; ga[5] = A.lin.z;
    double* gf = P.force + (size_t)i * 6;
    gf[0] = Fw.ang.x; gf[1] = Fw.ang.y; gf[2] = Fw.ang.z; gf[3] = Fw.lin.x; gf[4] = Fw.lin.y; gf[5] = Fw.lin.z;
    double* gi = P.inertia + (size_t)i * 7;
    gi[0] = I_diag.x; gi[1] = I_diag.y; gi[2] = I_diag.z; gi[3] = 0.0; gi[4] = 0.0; gi[5] = 0.0; gi[6] = mass;
    gst[APOLLO_S_THROTTLE] = throttle; gst[APOLLO_S_THROTTLE_CMD] = throttle_cmd;
    gst[2] = setpoint.i; gst[3] = setpoint.j; gst[4] = setpoint.k; gst[5] = setpoint.w;
    gst[APOLLO_S_PROPELLANT] = prop; gst[APOLLO_S_RCS_PROPELLANT] = rcs_prop; gst[APOLLO_S_THRUST] = thrust;
    gst[9] = torque.x; gst[10] = torque.y; gst[11] = torque.z;
    gst[APOLLO_S_LANDED] = landed; gst[APOLLO_S_TOUCHDOWN_SPEED] = td_speed; gst[APOLLO_S_TOUCHDOWN_HSPEED] = td_hspeed;
    gst[APOLLO_S_PITCH] = pitch;
    ggd[APOLLO_G_LAST_THROTTLE] = last_throttle;
    ggd[1] = last_att.i; ggd[2] = last_att.j; ggd[3] = last_att.k; ggd[4] = last_att.w;
    ggd[APOLLO_G_LAST_RATE] = last_rate; ggd[APOLLO_G_FTP_LATCHED] = latched; ggd[APOLLO_G_RESULT_EMITTED] = emitted;
    gsc[0] = e_alt; gsc[1] = e_pitch; gsc[2] = e_n;
}

hipError_t launch_apollo(const ApolloParams& p, hipStream_t stream) {
    if (p.n == 0) return hipSuccess;
    hipLaunchKernelGGL(apollo_rollout_kernel, dim3((p.n + 63) / 64), dim3(64), 0, stream, p);
    return hipGetLastError();
}

}  // namespace sixdof
