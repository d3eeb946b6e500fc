/*
 * apollo_oracle.c — CPU restatement of the Apollo-lander rollout (BASELINE config 4).  TEST INFRASTRUCTURE ONLY.
 *
 * Follows, in the reference's operation order:
 *   examples/apollo-lander/sim.py:334-343   engine_response
 *   examples/apollo-lander/sim.py:368-378   attitude_control
 *   examples/apollo-lander/sim.py:345-366   mass_props
 *   examples/apollo-lander/sim.py:380-398   lunar_gravity | apply_main_thrust | apply_rcs_torque  (six_dof effectors)
 *   libs/nox-py/src/integrator/semi_implicit.rs:17-62 (el.Integrator.SemiImplicit, sim.py:523)
 *   examples/apollo-lander/sim.py:400-431   ground_contact
 *   examples/apollo-lander/sim.py:433-444   derive_telemetry
 *   pipe order sim.py:517-526: non_effectors | six_dof(effectors) | ground_contact | derive_telemetry
 * and the closed loop the reference runs OUTSIDE the tick, in a sidecar process over UDP:
 *   examples/apollo-lander/controller/src/main.rs:100-262  guidance law (command, ThrottleLogic, tilt caps)
 *   examples/apollo-lander/main.py:147-163,166-283        post_step: state packing, 3-deg attitude slew, result
 *
 *   libs/nox-py/src/impeller2_server.rs:553-678,790-791    server loop: ticks_per_telemetry (= 120 / 40 = 3) ticks per
 *                                                          batch, then post_step(end_tick = ticks completed - 1)
 *
 * PINNED (tests/test_apollo_reference_fixtures.py) against tests/golden/apollo_reference_runs.json: full descents flown by
 * the reference's OWN sim.py systems and main.py post_step, executed on numpy under tests/golden/refshim.py with the
 * server loop's batching restated from impeller2_server.rs.  NOT pinned by reference code: the guidance law itself
 * (controller/src/main.rs is Rust with no tests or vectors; the fixture generator carries a line-by-line Python port of
 * `command()` as the stand-in for the UDP bridge), and the lock-step timing of that bridge (assumed never to time out).
 * The visualisation-only systems (thrust_visualization, truth_playback) are not modelled.
 */
#include "apollo_oracle.h"

#include <math.h>
#include <string.h>

#define G0 9.80665
#define LUNAR_GRAVITY 1.622
#define R_MOON_M 1737400.0
#define DPS_MAX_THRUST_N 45040.0
#define DPS_MIN_THRUST_N 4670.0
#define THROTTLE_MIN (DPS_MIN_THRUST_N / DPS_MAX_THRUST_N)
#define THROTTLE_MAX 1.0
#define RCS_THRUST_N 445.0
#define RCS_ISP_S 290.0
#define RCS_MOMENT_ARM_M 2.0
#define RCS_AXIS_TORQUE_LIMIT_NM (4.0 * RCS_THRUST_N * RCS_MOMENT_ARM_M)
#define FOOTPAD_HEIGHT_M 2.40
#define SIM_TIME_STEP (1.0 / 120.0)
#define SOFT_VERTICAL_SPEED_MPS 3.0
#define SOFT_HORIZONTAL_SPEED_MPS 1.0
#define UPRIGHT_DOT_MIN 0.94

/* controller/src/main.rs:8-45 */
#define C_MIN_THROTTLE (4670.0 / 45040.0)
#define C_FTP_THROTTLE 0.925
#define C_EROSION_BAND_MIN 0.65
#define C_MAX_DESCENT_RATE 120.0
#define C_MIN_DESCENT_RATE 0.5
#define C_MIN_VERTICAL_ACCEL 0.05
#define C_MAX_TILT_BRAKING_DEG 82.0
#define C_MAX_TILT_APPROACH_DEG 30.0
#define C_TILT_BLEND_HI 150.0
#define C_TILT_BLEND_LO 40.0
#define C_HSPEED_GAIN 0.25
#define C_POSITION_AUTHORITY 0.5
#define C_RATE_TRACK_AUTHORITY 12.0
#define C_VERTICAL_FB_AUTHORITY 0.8
#define C_HSPEED_FB_AUTHORITY 0.8
#define C_TERMINAL_NULL_ALT 40.0

static double clampd(double x, double lo, double hi) { return fmin(fmax(x, lo), hi); }
static double to_radians(double d) { return d * (M_PI / 180.0); } /* f64::to_radians */

/* reference.py:164-175 interp (bisect_right) */
static double interp(double t, const double* xs, const double* ys, uint32_t n) {
    if (t <= xs[0]) return ys[0];
    if (t >= xs[n - 1]) return ys[n - 1];
    uint32_t lo = 0, hi = n; /* bisect_right */
    while (lo < hi) {
        uint32_t mid = (lo + hi) / 2;
        if (t < xs[mid]) hi = mid; else lo = mid + 1;
    }
    const uint32_t i = lo, a = i - 1, b = i;
    const double span = xs[b] - xs[a];
    if (span <= 0.0) return ys[a];
    const double frac = (t - xs[a]) / span;
    return ys[a] + (ys[b] - ys[a]) * frac;
}

/* controller main.rs:100-121 */
static void quat_from_body_z(const double* dir, double* q) {
    double d[3];
    const double n = sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    if (n < 1e-9) { d[0] = 0.0; d[1] = 0.0; d[2] = 1.0; }
    else { d[0] = dir[0] / n; d[1] = dir[1] / n; d[2] = dir[2] / n; }
    const double cr[3] = {-d[1], d[0], 0.0};
    const double dot = clampd(d[2], -1.0, 1.0);
    if (dot < -0.999999) { q[0] = 1.0; q[1] = 0.0; q[2] = 0.0; q[3] = 0.0; return; }
    const double v[4] = {cr[0], cr[1], cr[2], 1.0 + dot};
    const double m = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
    q[0] = v[0] / m; q[1] = v[1] / m; q[2] = v[2] / m; q[3] = v[3] / m;
}

/* main.py:141-163 */
static void normalize_quat(const double* q, double* o) {
    const double norm = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (norm < 1e-12) { o[0] = 0.0; o[1] = 0.0; o[2] = 0.0; o[3] = 1.0; return; }
    for (int i = 0; i < 4; i++) o[i] = q[i] / norm;
}
static void slew_quat(const double* current_in, const double* target_in, double max_deg, double* out) {
    double cur[4], tgt[4];
    normalize_quat(current_in, cur);
    normalize_quat(target_in, tgt);
    double dot = cur[0] * tgt[0] + cur[1] * tgt[1] + cur[2] * tgt[2] + cur[3] * tgt[3];
    if (dot < 0.0) { for (int i = 0; i < 4; i++) tgt[i] = -tgt[i]; dot = -dot; }
    dot = fmin(fmax(dot, -1.0), 1.0);
    const double angle = 2.0 * acos(dot);
    const double max_angle = max_deg * (M_PI / 180.0); /* math.radians */
    if (angle <= max_angle || angle < 1e-9) { memcpy(out, tgt, sizeof(tgt)); return; }
    const double frac = max_angle / angle;
    double bl[4];
    for (int i = 0; i < 4; i++) bl[i] = (1.0 - frac) * cur[i] + frac * tgt[i];
    normalize_quat(bl, out);
}

/* controller main.rs:188-262 command() with ThrottleLogic (:163-186) */
static void guidance_command(const apollo_world* w, uint64_t i, double t_s, double* gd /* guidance row */,
                             double* out_throttle, double* out_q, double* out_rate_cmd) {
    const double* P = w->params + APOLLO_N_PARAMS * i;
    const double* pos = w->world_pos + 7 * i;
    const double* vel = w->world_vel + 6 * i;
    const double altitude = w->altitude[i], vertical_speed = w->vertical_speed[i];
    const double wv[3] = {vel[3], vel[4], vel[5]};
    const double mass = P[APOLLO_P_DRY_MASS] + w->propellant[i] + w->rcs_propellant[i];
    const double gravity = LUNAR_GRAVITY * P[APOLLO_P_GRAVITY_SCALE];
    const double ref_alt = interp(t_s, w->ref_time, w->ref_altitude, w->n_ref);
    const double ref_rate = interp(t_s, w->ref_time, w->ref_rate, w->n_ref);
    const double ref_downrange = interp(t_s, w->ref_time, w->ref_downrange, w->n_ref);
    const double ref_hspeed = interp(t_s, w->ref_time, w->ref_hspeed, w->n_ref);
    const double ref_hdecel = ref_hspeed - interp(t_s + 1.0, w->ref_time, w->ref_hspeed, w->n_ref);
    const double track_gain = P[APOLLO_P_TRACK_GAIN], vertical_gain = P[APOLLO_P_VERTICAL_GAIN],
                 horizontal_gain = P[APOLLO_P_HORIZONTAL_GAIN], thrust_scale = P[APOLLO_P_THRUST_SCALE];

    const double h_speed = hypot(wv[0], wv[1]);
    const double g_eff = fmax(gravity - h_speed * h_speed / R_MOON_M, 0.05 * gravity);
    const double rate_track = clampd(track_gain * (ref_alt - altitude), -C_RATE_TRACK_AUTHORITY, C_RATE_TRACK_AUTHORITY);
    const double rate_cmd = clampd(ref_rate + rate_track, -C_MAX_DESCENT_RATE, -C_MIN_DESCENT_RATE);
    const double vertical_fb = clampd(vertical_gain * (rate_cmd - vertical_speed), -C_VERTICAL_FB_AUTHORITY, C_VERTICAL_FB_AUTHORITY);
    double vertical_accel = fmax(g_eff + vertical_fb, C_MIN_VERTICAL_ACCEL);

    const double position_gain = 0.01 * horizontal_gain;
    const double trim_fade = clampd((altitude - 30.0) / 120.0, 0.0, 1.0);
    const double trim_x = clampd(position_gain * (ref_downrange - pos[4]), -C_POSITION_AUTHORITY, C_POSITION_AUTHORITY) * trim_fade;
    const double trim_y = clampd(position_gain * (-pos[5]), -C_POSITION_AUTHORITY, C_POSITION_AUTHORITY) * trim_fade;
    double target_vx = ref_hspeed, target_decel = ref_hdecel;
    if (altitude < C_TERMINAL_NULL_ALT) { target_vx = 0.0; target_decel = 0.0; }
    const double hspeed_fb = clampd(C_HSPEED_GAIN * (target_vx - wv[0]), -C_HSPEED_FB_AUTHORITY, C_HSPEED_FB_AUTHORITY);
    double ax = -target_decel + hspeed_fb + trim_x;
    double ay = clampd(C_HSPEED_GAIN * (-wv[1]), -C_HSPEED_FB_AUTHORITY, C_HSPEED_FB_AUTHORITY) + trim_y;

    const double blend = clampd((h_speed - C_TILT_BLEND_LO) / (C_TILT_BLEND_HI - C_TILT_BLEND_LO), 0.0, 1.0);
    const double max_tilt_deg = C_MAX_TILT_APPROACH_DEG + (C_MAX_TILT_BRAKING_DEG - C_MAX_TILT_APPROACH_DEG) * blend;
    if (h_speed > C_TILT_BLEND_LO) {
        /* cap_tilt_preserve_magnitude, main.rs:128-142 */
        const double az = fmax(vertical_accel, C_MIN_VERTICAL_ACCEL);
        const double ah = hypot(ax, ay);
        vertical_accel = az;
        if (!(ah < 1e-9)) {
            const double max_tilt = to_radians(max_tilt_deg);
            if (!(atan2(ah, az) <= max_tilt)) {
                const double mag = sqrt(ah * ah + az * az);
                const double scale_h = mag * sin(max_tilt) / ah;
                ax = ax * scale_h; ay = ay * scale_h; vertical_accel = mag * cos(max_tilt);
            }
        }
    } else {
        /* clamp_horizontal, main.rs:147-156 */
        const double limit = fmax(vertical_accel, C_MIN_VERTICAL_ACCEL) * tan(to_radians(C_MAX_TILT_APPROACH_DEG));
        const double mag = hypot(ax, ay);
        if (!(mag <= limit || mag < 1e-9)) { const double sc = limit / mag; ax = ax * sc; ay = ay * sc; }
    }
    const double desired[3] = {ax, ay, vertical_accel};
    const double thrust_required = mass * sqrt(ax * ax + ay * ay + vertical_accel * vertical_accel);
    const double demand = clampd(thrust_required / fmax(DPS_MAX_THRUST_N * thrust_scale, 1.0), C_MIN_THROTTLE, C_FTP_THROTTLE);
    /* ThrottleLogic::apply */
    int latched = gd[APOLLO_G_FTP_LATCHED] > 0.5;
    if (latched && demand < 0.60) latched = 0;
    else if (!latched && demand > 0.80) latched = 1;
    double throttle;
    if (demand <= C_EROSION_BAND_MIN && !latched) throttle = fmax(demand, C_MIN_THROTTLE);
    else if (latched) throttle = C_FTP_THROTTLE;
    else throttle = C_EROSION_BAND_MIN;
    gd[APOLLO_G_FTP_LATCHED] = latched ? 1.0 : 0.0;
    *out_throttle = throttle;
    quat_from_body_z(desired, out_q);
    *out_rate_cmd = rate_cmd;
}

static void tick_one(apollo_world* w, uint64_t i) {
    const double* P = w->params + APOLLO_N_PARAMS * i;
    double* pos = w->world_pos + 7 * i;
    double* vel = w->world_vel + 6 * i;
    double* inertia = w->inertia + 7 * i;
    double* setpoint = w->attitude_setpoint + 4 * i;
    double* torque = w->rcs_torque + 3 * i;

    const double dry_mass = P[APOLLO_P_DRY_MASS];
    const double total_mass = dry_mass + P[APOLLO_P_PROPELLANT] + P[APOLLO_P_RCS_PROPELLANT];
    const double isp = P[APOLLO_P_ISP], thrust_scale = P[APOLLO_P_THRUST_SCALE];
    const double ag = P[APOLLO_P_ATTITUDE_GAIN] / 0.040;
    const double rcs_k[3] = {4500.0 * ag, 5500.0 * ag, 4500.0 * ag};
    const double rcs_d[3] = {19000.0, 21000.0, 19000.0};
    const double base_inertia[3] = {78000.0, 72000.0, 45000.0};
    const double response_alpha = fmin(fmax(P[APOLLO_P_THROTTLE_RESPONSE_HZ] * SIM_TIME_STEP, 0.0), 1.0);
    const double lunar_g = LUNAR_GRAVITY * P[APOLLO_P_GRAVITY_SCALE];
    const double landed = w->landed[i];

    /* engine_response */
    {
        const double cmd = clampd(w->throttle_cmd[i], THROTTLE_MIN, THROTTLE_MAX);
        double actual = w->throttle[i] + (cmd - w->throttle[i]) * response_alpha;
        const int active = (w->propellant[i] > 0.0) && (landed < 0.5);
        actual = active ? actual : 0.0;
        w->throttle[i] = actual;
        w->thrust[i] = actual * DPS_MAX_THRUST_N * thrust_scale;
    }
    /* attitude_control */
    {
        double qi[4], qerr[4], body_rate[3];
        orc_quat_inverse(pos, qi);
        orc_quat_mul(qi, setpoint, qerr);
        const double sign = (qerr[3] >= 0.0) ? 1.0 : -1.0;
        orc_quat_rotate(qi, vel, body_rate);
        for (int c = 0; c < 3; c++) {
            double t = sign * qerr[c] * rcs_k[c] - body_rate[c] * rcs_d[c];
            t = clampd(t, -RCS_AXIS_TORQUE_LIMIT_NM, RCS_AXIS_TORQUE_LIMIT_NM);
            torque[c] = (landed > 0.5) ? 0.0 : t;
        }
    }
    /* mass_props */
    {
        const double dps_burn = w->thrust[i] / (isp * G0) * SIM_TIME_STEP;
        const double rcs_force_equivalent = (fabs(torque[0]) + fabs(torque[1]) + fabs(torque[2])) / RCS_MOMENT_ARM_M;
        const double rcs_burn = rcs_force_equivalent / (RCS_ISP_S * G0) * SIM_TIME_STEP;
        const double next_prop = fmax(w->propellant[i] - dps_burn, 0.0);
        const double next_rcs = fmax(w->rcs_propellant[i] - rcs_burn, 0.0);
        const double mass = dry_mass + next_prop + next_rcs;
        const double inertia_scale = mass / total_mass;
        w->propellant[i] = next_prop;
        w->rcs_propellant[i] = next_rcs;
        for (int c = 0; c < 3; c++) inertia[c] = (landed > 0.5) ? 1.0e9 : base_inertia[c] * inertia_scale;
        inertia[3] = 0.0; inertia[4] = 0.0; inertia[5] = 0.0;
        inertia[6] = mass;
    }
    /* six_dof(lunar_gravity | apply_main_thrust | apply_rcs_torque), semi-implicit */
    {
        double F[6] = {0, 0, 0, 0, 0, 0}, A[6], r[3];
        const double v_h_sq = vel[3] * vel[3] + vel[4] * vel[4];
        const double g_eff = fmax(lunar_g - v_h_sq / R_MOON_M, 0.0);
        const double gdir[3] = {0.0, 0.0, -1.0};
        for (int c = 0; c < 3; c++) { F[c] = F[c] + 0.0; F[3 + c] = F[3 + c] + gdir[c] * g_eff * inertia[6]; }
        const double thrust_body[3] = {0.0, 0.0, w->thrust[i]};
        orc_quat_rotate(pos, thrust_body, r);
        for (int c = 0; c < 3; c++) { F[c] = F[c] + 0.0; F[3 + c] = F[3 + c] + r[c]; }
        orc_quat_rotate(pos, torque, r);
        for (int c = 0; c < 3; c++) { F[c] = F[c] + r[c]; F[3 + c] = F[3 + c] + 0.0; }
        orc_calc_accel(F, inertia, pos, A);
        const double dt = w->simulation_time_step;
        double dv[6], xn[7];
        for (int c = 0; c < 6; c++) vel[c] = vel[c] + dt * A[c];
        for (int c = 0; c < 6; c++) dv[c] = dt * vel[c];
        orc_transform_add_motion(pos, dv, xn);
        memcpy(pos, xn, sizeof(xn));
        memcpy(w->world_accel + 6 * i, A, sizeof(A));
        memcpy(w->force + 6 * i, F, sizeof(F));
    }
    /* ground_contact */
    {
        const double altitude = pos[6], vertical_speed = vel[5];
        const int contact = altitude <= FOOTPAD_HEIGHT_M;
        const int was_landed = landed > 0.5;
        const int landed_now = was_landed || contact;
        const int first_contact = !was_landed && contact;
        if (first_contact) {
            w->touchdown_speed[i] = fabs(vertical_speed);
            w->touchdown_horizontal_speed[i] = sqrt(vel[3] * vel[3] + vel[4] * vel[4]);
        }
        if (landed_now) {
            pos[6] = FOOTPAD_HEIGHT_M;
            for (int c = 0; c < 6; c++) vel[c] = 0.0;
        }
        w->landed[i] = landed_now ? 1.0 : 0.0;
    }
    /* derive_telemetry */
    {
        const double up[3] = {0.0, 0.0, 1.0};
        double body_up[3];
        orc_quat_rotate(pos, up, body_up);
        w->pitch[i] = acos(clampd(body_up[2], -1.0, 1.0)) * (180.0 / M_PI);
        w->altitude[i] = pos[6];
        w->vertical_speed[i] = vel[5];
        w->horizontal_speed[i] = sqrt(vel[3] * vel[3] + vel[4] * vel[4]);
    }
}

/* main.py:166-283 post_step for one rollout; `tick` = the end_tick the server loop hands over (ticks completed - 1) */
static void post_step_one(apollo_world* w, uint64_t i, uint64_t tick) {
    double* gd = w->guidance + APOLLO_N_GUIDANCE * i;
    double* sc = w->score + APOLLO_N_SCORE * i;
    double* res = w->result + APOLLO_N_RESULT * i;
    const double* pos = w->world_pos + 7 * i;
    const double t_s = (double)tick * SIM_TIME_STEP;
    const double altitude = w->altitude[i], pitch = w->pitch[i];
    const int landed = w->landed[i] > 0.5;
    const double real_altitude = interp(t_s, w->ref_time, w->ref_altitude, w->n_ref);
    const double truth_pitch_now = fabs(interp(t_s, w->ref_time, w->ref_pitch, w->n_ref));
    const double da = altitude - real_altitude, dp = pitch - truth_pitch_now;
    sc[0] += da * da;
    sc[1] += dp * dp;
    sc[2] += 1.0;

    if (tick % w->guidance_period == 0 && !landed) {
        double thr, tq[4], rc, sl[4];
        guidance_command(w, i, t_s, gd, &thr, tq, &rc);
        gd[APOLLO_G_LAST_THROTTLE] = thr;
        gd[APOLLO_G_LAST_RATE] = rc;
        slew_quat(gd + APOLLO_G_LAST_ATT, tq, 3.0, sl);
        memcpy(gd + APOLLO_G_LAST_ATT, sl, sizeof(sl));
    }
    w->throttle_cmd[i] = gd[APOLLO_G_LAST_THROTTLE];
    memcpy(w->attitude_setpoint + 4 * i, gd + APOLLO_G_LAST_ATT, 4 * sizeof(double));

    if (!(gd[APOLLO_G_RESULT_EMITTED] > 0.5) && (landed || tick >= w->max_ticks - 1)) {
        double td = w->touchdown_speed[i], tdh = w->touchdown_horizontal_speed[i];
        if (!landed) { td = fabs(w->vertical_speed[i]); tdh = w->horizontal_speed[i]; }
        const double n = fmax(sc[2], 1.0);
        const double upright_dot = cos(fabs(pitch) * (M_PI / 180.0));
        const double propellant = w->propellant[i];
        res[APOLLO_R_TOUCHDOWN_SPEED] = td;
        res[APOLLO_R_HORIZONTAL_SPEED] = tdh;
        res[APOLLO_R_FUEL_REMAINING] = propellant;
        res[APOLLO_R_RCS_FUEL_REMAINING] = w->rcs_propellant[i];
        res[APOLLO_R_TRAJ_RMSE] = sqrt(sc[0] / n);
        res[APOLLO_R_PITCH_RMSE] = sqrt(sc[1] / n);
        res[APOLLO_R_DOWNRANGE_MISS] = hypot(pos[4], pos[5]);
        res[APOLLO_R_UPRIGHT_DOT] = upright_dot;
        res[APOLLO_R_LANDED] = landed ? 1.0 : 0.0;
        res[APOLLO_R_SOFT_LANDING] = (landed && td <= SOFT_VERTICAL_SPEED_MPS && tdh <= SOFT_HORIZONTAL_SPEED_MPS &&
                                      upright_dot >= UPRIGHT_DOT_MIN && propellant > 0.0) ? 1.0 : 0.0;
        res[APOLLO_R_TICK] = (double)tick;
        gd[APOLLO_G_RESULT_EMITTED] = 1.0;
    }
}

int apollo_step(apollo_world* w, uint64_t n_ticks, int threads) {
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t i = 0; i < (int64_t)w->n; i++) {
        uint64_t tick = w->tick;
        const uint64_t tpt = w->ticks_per_telemetry ? w->ticks_per_telemetry : 3;
        for (uint64_t t = 0; t < n_ticks; t++) {
            tick += 1;   /* ticks completed */
            tick_one(w, (uint64_t)i);
            /* impeller2_server.rs:553-678: post_step once per batch of ticks_per_telemetry ticks (the last batch is cut
             * at max_ticks), with end_tick = batch start + batch - 1 = ticks completed - 1 */
            if (tick % tpt == 0 || tick == w->max_ticks) post_step_one(w, (uint64_t)i, tick - 1);
        }
    }
    w->tick += n_ticks;
    return 0;
}
