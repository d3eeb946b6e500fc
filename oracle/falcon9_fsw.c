/* falcon9_fsw.c — CPU restatement of the Falcon 9 example's FLIGHT SOFTWARE, ascent phases.  TEST INFRASTRUCTURE ONLY.
 *
 * The reference flies its Falcon 9 plant (examples/falcon9/sim.py) with an external Rust process
 * (examples/falcon9/controller/src/{main,math,profile}.rs) that receives a 49-double sensor packet every 10 ticks and
 * answers with a 27-double command packet (examples/falcon9/main.py:280-352).  There is no rustc in this image, so the
 * process cannot be built; this file restates it in C, statement by statement in the reference's operation order, for
 * the phases an ascent visits: PadPress -> VerticalRise -> PitchKick -> GravityTurn -> Meco (main.rs:384-533), with the
 * IMU + GPS (+ radar) navigator (main.rs:213-327), the recorded-profile reference trajectory (profile.rs:23-97) and the
 * vector / quaternion / WGS84 helpers (math.rs).  Phases from Flip on (main.rs:534-786: boostback, entry, landing) are
 * NOT restated: f9fsw_step reports F9FSW_BEYOND_ASCENT when the state machine would enter Flip, and the caller stops.
 *
 * Nothing under elodin_amd/ includes, links or calls this file.  It exists so that the closed-loop fixtures
 * (tests/golden/make_falcon9_closed_loop.py: the reference's OWN plant + sensor systems stepped by this flight software)
 * pin the product's generated kernel against something that is not the product's tracer.
 *
 * Parity of this restatement itself: PARITY UNPINNED by reference-held vectors — the Rust source has no tests and no vectors
 *  and cannot be built here, so the state machine and the navigator are pinned by reading only (every function cites
 * the lines it follows).  What of it can be checked against reference code run here is (tests/test_falcon9_fsw_oracle.py): the
 * WGS84 helpers against the answers of the example's frames.py, the profile resampling against reference.py's interpolation.
 * libm calls are the ones Rust's f64 methods lower to on Linux.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "falcon9_fsw.h"

/* ---- math.rs:5-9 ------------------------------------------------------------------------------------------------ */
#define WGS84_A 6378137.0
#define WGS84_F (1.0 / 298.257223563)
#define WGS84_E2 (WGS84_F * (2.0 - WGS84_F))
#define MU_EARTH 3.986004418e14
#define OMEGA_EARTH 7.292115e-5
#define PI 3.14159265358979323846264338327950288

/* main.rs:34 */
#define THROTTLE_MIN 0.57
/* main.rs:27-32 valve indices */
enum { V_HE_LOX = 0, V_HE_RP1 = 2, V_MAIN_LOX = 4, V_MAIN_RP1 = 5, V_TEATEB = 6, V_PURGE = 7 };

typedef struct { double v[3]; } v3;
typedef struct { double q[4]; } quat;

static double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); } /* f64::clamp (no NaN here) */
static double to_radians(double deg) { return deg * (PI / 180.0); }                              /* f64::to_radians */

/* math.rs:13-49 */
static v3 add(v3 a, v3 b) { return (v3){{a.v[0] + b.v[0], a.v[1] + b.v[1], a.v[2] + b.v[2]}}; }
static v3 sub(v3 a, v3 b) { return (v3){{a.v[0] - b.v[0], a.v[1] - b.v[1], a.v[2] - b.v[2]}}; }
static v3 scale(v3 a, double s) { return (v3){{a.v[0] * s, a.v[1] * s, a.v[2] * s}}; }
static double dot(v3 a, v3 b) { return a.v[0] * b.v[0] + a.v[1] * b.v[1] + a.v[2] * b.v[2]; }
static v3 cross(v3 a, v3 b) {
    return (v3){{a.v[1] * b.v[2] - a.v[2] * b.v[1], a.v[2] * b.v[0] - a.v[0] * b.v[2], a.v[0] * b.v[1] - a.v[1] * b.v[0]}};
}
static double norm(v3 a) { return sqrt(dot(a, a)); }
static v3 normalize(v3 a) {
    const double n = norm(a);
    if (n < 1e-12) return (v3){{0.0, 0.0, 0.0}};
    return scale(a, 1.0 / n);
}

/* math.rs:54-89 */
static const quat QUAT_IDENT = {{0.0, 0.0, 0.0, 1.0}};
static quat quat_mul(quat a, quat b) {
    const double ax = a.q[0], ay = a.q[1], az = a.q[2], aw = a.q[3];
    const double bx = b.q[0], by = b.q[1], bz = b.q[2], bw = b.q[3];
    return (quat){{aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                   aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz}};
}
static quat quat_conj(quat q) { return (quat){{-q.q[0], -q.q[1], -q.q[2], q.q[3]}}; }
static quat quat_normalize(quat q) {
    const double n = sqrt(q.q[0] * q.q[0] + q.q[1] * q.q[1] + q.q[2] * q.q[2] + q.q[3] * q.q[3]);
    if (n < 1e-12) return QUAT_IDENT;
    return (quat){{q.q[0] / n, q.q[1] / n, q.q[2] / n, q.q[3] / n}};
}
static v3 quat_rotate(quat q, v3 v) { /* v' = q v q*  (math.rs:78-83) */
    const v3 qv = {{q.q[0], q.q[1], q.q[2]}};
    const v3 t = scale(cross(qv, v), 2.0);
    return add(add(v, scale(t, q.q[3])), cross(qv, t));
}
static v3 quat_rotate_inv(quat q, v3 v) { return quat_rotate(quat_conj(q), v); }
static quat quat_from_axis_angle(v3 axis, double angle) { /* math.rs:90-94 */
    const v3 a = normalize(axis);
    const double s = sin(angle * 0.5), c = cos(angle * 0.5);
    return (quat){{a.v[0] * s, a.v[1] * s, a.v[2] * s, c}};
}
static quat quat_integrate(quat q, v3 omega_body, double dt) { /* math.rs:112-119: q <- q * exp(omega dt / 2) */
    const double angle = norm(omega_body) * dt;
    if (angle < 1e-12) return q;
    return quat_normalize(quat_mul(q, quat_from_axis_angle(omega_body, angle)));
}
static quat quat_between(v3 from, v3 to) { /* math.rs:122-138 */
    const double c = clampd(dot(from, to), -1.0, 1.0);
    if (c > 1.0 - 1e-12) return QUAT_IDENT;
    if (c < -1.0 + 1e-12) {
        const v3 axis = fabs(from.v[0]) < 0.9 ? normalize(cross(from, (v3){{1.0, 0.0, 0.0}})) : normalize(cross(from, (v3){{0.0, 1.0, 0.0}}));
        return quat_from_axis_angle(axis, PI);
    }
    return quat_from_axis_angle(normalize(cross(from, to)), acos(c));
}

/* math.rs:141-158: Bowring, four fixed iterations */
static void ecef_to_geodetic(v3 r, double* lat_out, double* lon_out, double* alt_out) {
    const double x = r.v[0], y = r.v[1], z = r.v[2];
    const double lon = atan2(y, x);
    const double p = hypot(x, y);
    const double b = WGS84_A * (1.0 - WGS84_F);
    const double ep2 = WGS84_E2 / (1.0 - WGS84_E2);
    double beta = atan2(z, (1.0 - WGS84_F) * p);
    double lat = beta;
    for (int i = 0; i < 4; i++) {
        const double sb = sin(beta), cb = cos(beta);
        lat = atan2(z + ep2 * b * (sb * sb * sb), p - WGS84_E2 * WGS84_A * (cb * cb * cb)); /* powi(3) = x*x*x */
        beta = atan((1.0 - WGS84_F) * tan(lat));
    }
    const double sin_lat = sin(lat);
    const double w = sqrt(1.0 - WGS84_E2 * sin_lat * sin_lat);
    *lat_out = lat;
    *lon_out = lon;
    *alt_out = p * cos(lat) + z * sin_lat - WGS84_A * w;
}
/* math.rs:170-178: rows north, east, down */
static void ned_basis(double lat, double lon, v3 out[3]) {
    const double sl = sin(lat), cl = cos(lat), so = sin(lon), co = cos(lon);
    out[0] = (v3){{-sl * co, -sl * so, cl}};
    out[1] = (v3){{-so, co, 0.0}};
    out[2] = (v3){{-cl * co, -cl * so, -sl}};
}
static v3 up_at(v3 pos) { /* `scale(ned_basis(lat, lon)[2], -1.0)`, main.rs:233-234,412-415 */
    double lat, lon, alt;
    v3 ned[3];
    ecef_to_geodetic(pos, &lat, &lon, &alt);
    ned_basis(lat, lon, ned);
    return scale(ned[2], -1.0);
}
static v3 gravity(v3 r) { /* math.rs:181-184 */
    const double n = norm(r);
    return scale(r, -MU_EARTH / (n * n * n));
}
static v3 frame_accel(v3 r, v3 v) { /* math.rs:187-190: -2 w x v - w x (w x r) */
    const v3 w = {{0.0, 0.0, OMEGA_EARTH}};
    return sub(scale(cross(w, v), -2.0), cross(w, cross(w, r)));
}
static double density(double alt_m) { /* math.rs:193-200 */
    const double h = alt_m > 0.0 ? alt_m : 0.0;   /* f64::max(0.0) */
    if (h < 25000.0) return 1.225 * exp(-h / 8440.0);
    return 0.0642 * exp(-(h - 25000.0) / 6580.0);
}

/* ---- profile.rs ----------------------------------------------------------------------------------------------------- */
static double interp(double x, const double* xs, const double* ys, size_t n) { /* profile.rs:23-38 */
    if (x <= xs[0]) return ys[0];
    if (x >= xs[n - 1]) return ys[n - 1];
    size_t lo = 0, hi = n - 1;
    while (hi - lo > 1) {
        const size_t mid = (lo + hi) / 2;
        if (xs[mid] <= x) lo = mid;
        else hi = mid;
    }
    const double f = (x - xs[lo]) / (xs[hi] - xs[lo]);
    return ys[lo] + f * (ys[hi] - ys[lo]);
}
static void smooth(const double* v, double* out, size_t n) { /* profile.rs:57-66: moving average over [i-4, i+5) */
    for (size_t i = 0; i < n; i++) {
        const size_t lo = i >= 4 ? i - 4 : 0, hi = i + 5 < n ? i + 5 : n;
        double s = 0.0;   /* iter().sum::<f64>() adds left to right starting from 0.0 */
        for (size_t k = lo; k < hi; k++) s += v[k];
        out[i] = s / (double)(hi - lo);
    }
}

struct f9fsw {
    /* Navigator, main.rs:198-211 */
    v3 nav_pos, nav_vel;
    quat nav_att;
    double last_gps_count, last_t;
    int initialized;
    v3 wind_est;
    double radar_alt_m;
    /* Fsw, main.rs:329-352 */
    int phase;
    double phase_t0;
    int meco_speed_reached;
    double purge_until;
    v3 pad_pos, up_pad, track_dir;
    double t_liftoff;
    /* AscentProfile, profile.rs:16-21 */
    size_t n_prof;
    double *prof_time, *prof_speed, *prof_alt, *prof_vspeed;
};

f9fsw* f9fsw_new(void) { /* Navigator::new main.rs:214-225, Fsw::new main.rs:355-381 */
    f9fsw* f = (f9fsw*)calloc(1, sizeof(f9fsw));
    if (!f) return NULL;
    f->nav_att = QUAT_IDENT;
    f->radar_alt_m = -1.0;
    f->phase = F9FSW_PAD_PRESS;
    f->purge_until = -1.0;
    f->up_pad = (v3){{0.0, 0.0, 1.0}};
    f->track_dir = (v3){{1.0, 0.0, 0.0}};
    f->t_liftoff = -1.0;
    return f;
}
void f9fsw_free(f9fsw* f) {
    if (!f) return;
    free(f->prof_time);
    free(f->prof_speed);
    free(f->prof_alt);
    free(f->prof_vspeed);
    free(f);
}

/* AscentProfile::load after the JSON parse (profile.rs:44-85): uniform 0.5 s grid, 9-point moving average, central
 * differences for the vertical speed.  `altitude_km` is the raw file's altitude column (km). */
int f9fsw_load_profile(f9fsw* f, const double* time, const double* velocity, const double* altitude_km, size_t n_raw) {
    if (!f || !time || !velocity || !altitude_km || n_raw == 0) return -1;
    const double t_end = time[n_raw - 1];
    const size_t n = (size_t)(t_end / 0.5) + 1;
    double* grid = (double*)malloc(n * sizeof(double));
    double* speed_raw = (double*)calloc(n, sizeof(double));
    double* alt_raw = (double*)calloc(n, sizeof(double));
    double* speed = (double*)malloc(n * sizeof(double));
    double* alt_m = (double*)malloc(n * sizeof(double));
    double* vspeed = (double*)malloc(n * sizeof(double));
    if (!grid || !speed_raw || !alt_raw || !speed || !alt_m || !vspeed) return -1;
    for (size_t i = 0; i < n; i++) grid[i] = (double)i * 0.5;
    for (size_t i = 0; i < n; i++) speed_raw[i] = interp(grid[i], time, velocity, n_raw);
    for (size_t i = 0; i < n; i++) alt_raw[i] = interp(grid[i], time, altitude_km, n_raw) * 1000.0;
    smooth(speed_raw, speed, n);
    smooth(alt_raw, alt_m, n);
    for (size_t i = 0; i < n; i++) {
        const size_t lo = i >= 1 ? i - 1 : 0, hi = i + 1 < n - 1 ? i + 1 : n - 1;
        vspeed[i] = hi == lo ? 0.0 : (alt_m[hi] - alt_m[lo]) / ((double)(hi - lo) * 0.5);
    }
    free(speed_raw);
    free(alt_raw);
    free(f->prof_time);
    free(f->prof_speed);
    free(f->prof_alt);
    free(f->prof_vspeed);
    f->n_prof = n;
    f->prof_time = grid;
    f->prof_speed = speed;
    f->prof_alt = alt_m;
    f->prof_vspeed = vspeed;
    return 0;
}
/* The resampled table itself (what f9fsw_profile_table returns, shipped as elodin_amd/data/falcon9_crs12_profile.csv and checked
 * bit for bit against f9fsw_load_profile's own resampling in tests/test_falcon9_fsw_oracle.py): for a host that has the table but
 * not the reference's raw data file (the GPU box). */
int f9fsw_load_table(f9fsw* f, const double* time, const double* speed, const double* alt_m, const double* vspeed, size_t n) {
    if (!f || !time || !speed || !alt_m || !vspeed || n == 0) return -1;
    double* c[4];
    const double* src[4] = {time, speed, alt_m, vspeed};
    for (int k = 0; k < 4; k++) {
        c[k] = (double*)malloc(n * sizeof(double));
        if (!c[k]) return -1;
        memcpy(c[k], src[k], n * sizeof(double));
    }
    free(f->prof_time);
    free(f->prof_speed);
    free(f->prof_alt);
    free(f->prof_vspeed);
    f->n_prof = n;
    f->prof_time = c[0];
    f->prof_speed = c[1];
    f->prof_alt = c[2];
    f->prof_vspeed = c[3];
    return 0;
}
size_t f9fsw_profile_table(const f9fsw* f, double* time, double* speed, double* alt_m, double* vspeed, size_t cap) {
    if (!f) return 0;
    for (size_t i = 0; i < f->n_prof && i < cap; i++) {
        if (time) time[i] = f->prof_time[i];
        if (speed) speed[i] = f->prof_speed[i];
        if (alt_m) alt_m[i] = f->prof_alt[i];
        if (vspeed) vspeed[i] = f->prof_vspeed[i];
    }
    return f->n_prof;
}

/* SensorPacket::parse, main.rs:131-174 (only what the ascent reads) */
typedef struct {
    double t;
    v3 imu_accel, imu_gyro, gps_pos, gps_vel;
    double gps_count, radar_range;
    double kick_deg, kick_start_s, kick_ramp_s, bucket_throttle, bucket_q_on_pa, meco_speed_mps, azimuth_deg, ascent_throttle,
        meco_fpa_deg, pitch_exp;
} packet;
static packet parse(const double* v) {
    packet s;
    s.t = v[0];
    s.imu_accel = (v3){{v[1], v[2], v[3]}};
    s.imu_gyro = (v3){{v[4], v[5], v[6]}};
    s.gps_pos = (v3){{v[7], v[8], v[9]}};
    s.gps_vel = (v3){{v[10], v[11], v[12]}};
    s.gps_count = v[13];
    s.radar_range = v[14];
    s.kick_deg = v[22];
    s.kick_start_s = v[23];
    s.kick_ramp_s = v[24];
    s.bucket_throttle = v[25];
    s.bucket_q_on_pa = v[26];
    s.meco_speed_mps = v[28];
    s.azimuth_deg = v[29];
    s.ascent_throttle = v[34];
    s.meco_fpa_deg = v[35];
    s.pitch_exp = v[36];
    return s;
}

/* Navigator::init main.rs:227-239, Navigator::step main.rs:241-295 */
static void nav_step(f9fsw* f, const packet* s) {
    if (!f->initialized) {
        if (s->gps_count > 0.0) {
            f->nav_pos = s->gps_pos;
            f->nav_vel = (v3){{0.0, 0.0, 0.0}};
            f->nav_att = quat_between((v3){{1.0, 0.0, 0.0}}, up_at(f->nav_pos));
            f->last_gps_count = s->gps_count;
            f->last_t = s->t;
            f->initialized = 1;
            f->wind_est = (v3){{0.0, 0.0, 0.0}};
            f->radar_alt_m = -1.0;
        }
        return;
    }
    const double dt = clampd(s->t - f->last_t, 0.0, 0.1);
    f->last_t = s->t;
    /* attitude: integrate the gyro minus the Earth rate it also measures */
    const v3 omega_e_body = quat_rotate_inv(f->nav_att, (v3){{0.0, 0.0, OMEGA_EARTH}});
    const v3 omega_frame = sub(s->imu_gyro, omega_e_body);
    f->nav_att = quat_integrate(f->nav_att, omega_frame, dt);
    /* translation: specific force + gravity + fictitious terms */
    const v3 f_e = quat_rotate(f->nav_att, s->imu_accel);
    const v3 a = add(add(f_e, gravity(f->nav_pos)), frame_accel(f->nav_pos, f->nav_vel));
    f->nav_vel = add(f->nav_vel, scale(a, dt));
    f->nav_pos = add(f->nav_pos, scale(f->nav_vel, dt));
    /* complementary GPS blend */
    if (s->gps_count > f->last_gps_count) {
        const v3 innov_v = sub(s->gps_vel, f->nav_vel);
        f->wind_est = add(scale(f->wind_est, 0.95), scale(innov_v, 0.05));
        f->nav_pos = add(f->nav_pos, scale(sub(s->gps_pos, f->nav_pos), 0.20));
        f->nav_vel = add(f->nav_vel, scale(sub(s->gps_vel, f->nav_vel), 0.50));
        f->last_gps_count = s->gps_count;
    }
    /* radar altimeter below 500 m */
    if (s->radar_range >= 0.0 && s->radar_range < 500.0) {
        double lat, lon, geo_alt;
        ecef_to_geodetic(f->nav_pos, &lat, &lon, &geo_alt);
        if (f->radar_alt_m < 0.0) f->radar_alt_m = s->radar_range;
        else f->radar_alt_m = 0.7 * f->radar_alt_m + 0.3 * s->radar_range;
        const v3 up = up_at(f->nav_pos);
        const double dh = f->radar_alt_m - geo_alt;
        if (fabs(dh) < 50.0) f->nav_pos = add(f->nav_pos, scale(up, 0.35 * dh));
    } else {
        f->radar_alt_m = -1.0;
    }
}
static double nav_altitude(const f9fsw* f) { /* main.rs:297-303 */
    if (f->radar_alt_m >= 0.0) return f->radar_alt_m;
    double lat, lon, alt;
    ecef_to_geodetic(f->nav_pos, &lat, &lon, &alt);
    return alt;
}

static void set_phase(f9fsw* f, int p, double t) { /* main.rs:372-378 */
    if (p != f->phase) {
        f->phase = p;
        f->phase_t0 = t;
    }
}
static void set_engines(double* cmd, double u) {
    for (int i = 0; i < 9; i++) cmd[i] = u;
}
static void set_attitude(double* cmd, quat q) {
    for (int i = 0; i < 4; i++) cmd[17 + i] = q.q[i];
}
static void open_feed(double* cmd) { /* the three `cmd.valves[...] = 1.0` lines every powered phase opens with */
    cmd[9 + V_MAIN_LOX] = 1.0;
    cmd[9 + V_MAIN_RP1] = 1.0;
    cmd[9 + V_TEATEB] = 1.0;
}

/* Fsw::step, main.rs:384-533, writing Command::pack's layout (main.rs:187-196):
 * engines 0..8 | valves 9..16 | attitude 17..20 | tvc_enable 21 | rcs_enable 22 | fins 23..25 | phase 26 */
int f9fsw_step(f9fsw* f, const double* state49, double* cmd27) {
    if (!f || !state49 || !cmd27) return -1;
    const packet s = parse(state49);
    nav_step(f, &s);
    memset(cmd27, 0, 27 * sizeof(double));
    set_attitude(cmd27, f->nav_att);
    cmd27[26] = (double)f->phase;   /* the phase the command was computed in: set BEFORE any transition */
    cmd27[9 + V_HE_LOX] = 1.0;
    cmd27[9 + V_HE_RP1] = 1.0;
    cmd27[9 + V_PURGE] = s.t < f->purge_until ? 1.0 : 0.0;   /* with the purge deadline as it stood BEFORE this step */
    if (!f->initialized) return 0;
    const double t = s.t;
    const double alt = nav_altitude(f);
    const double speed = norm(f->nav_vel);

    if (f->phase == F9FSW_PAD_PRESS && f->pad_pos.v[0] == 0.0 && f->pad_pos.v[1] == 0.0 && f->pad_pos.v[2] == 0.0) {
        f->pad_pos = f->nav_pos;
        double lat, lon, h;
        v3 ned[3];
        ecef_to_geodetic(f->pad_pos, &lat, &lon, &h);
        ned_basis(lat, lon, ned);
        f->up_pad = scale(ned[2], -1.0);
        const double az = to_radians(s.azimuth_deg);
        f->track_dir = normalize(add(scale(ned[0], cos(az)), scale(ned[1], sin(az))));
    }
    const v3 up_here = up_at(f->nav_pos);
    if (f->t_liftoff < 0.0 && dot(f->nav_vel, up_here) > 1.0) f->t_liftoff = t;   /* profile clock: first sustained climb */
    const v3 x_axis = {{1.0, 0.0, 0.0}};

    switch (f->phase) {
    case F9FSW_PAD_PRESS:
        open_feed(cmd27);
        set_attitude(cmd27, quat_between(x_axis, f->up_pad));
        cmd27[21] = 1.0;
        if (t >= 0.2) {
            set_engines(cmd27, s.ascent_throttle);
            set_phase(f, F9FSW_VERTICAL_RISE, t);
        }
        break;
    case F9FSW_VERTICAL_RISE:
        open_feed(cmd27);
        set_engines(cmd27, s.ascent_throttle);
        set_attitude(cmd27, quat_between(x_axis, f->up_pad));
        cmd27[21] = 1.0;
        if (t >= s.kick_start_s) set_phase(f, F9FSW_PITCH_KICK, t);
        break;
    case F9FSW_PITCH_KICK: {
        open_feed(cmd27);
        set_engines(cmd27, s.ascent_throttle);
        cmd27[21] = 1.0;
        const double fr = clampd((t - f->phase_t0) / s.kick_ramp_s, 0.0, 1.0);
        const double angle = fr * to_radians(s.kick_deg);
        const v3 dir = normalize(add(scale(f->up_pad, cos(angle)), scale(f->track_dir, sin(angle))));
        set_attitude(cmd27, quat_between(x_axis, dir));
        if (fr >= 1.0 && speed > 80.0) set_phase(f, F9FSW_GRAVITY_TURN, t);
        break;
    }
    case F9FSW_GRAVITY_TURN: {
        open_feed(cmd27);
        cmd27[21] = 1.0;
        /* the parametric lofted pitch program: flight-path angle as a function of speed (the fallback without a profile) */
        const double v0 = 90.0;
        const double fr = clampd((speed - v0) / (s.meco_speed_mps - v0), 0.0, 1.0);
        const double gamma = to_radians(90.0 - (90.0 - s.meco_fpa_deg) * pow(fr, s.pitch_exp));
        const v3 dir = normalize(add(scale(up_here, sin(gamma)), scale(f->track_dir, cos(gamma))));
        v3 dir_cmd = dir;
        double u = s.ascent_throttle;
        if (f->n_prof && f->t_liftoff >= 0.0) {   /* profile-following ascent, main.rs:489-507 */
            const double t_ref = t - f->t_liftoff;
            const double v_ref = interp(t_ref, f->prof_time, f->prof_speed, f->n_prof);
            const double gamma_ref = asin(clampd(interp(t_ref, f->prof_time, f->prof_vspeed, f->n_prof) / (v_ref > 30.0 ? v_ref : 30.0), -1.0, 1.0));
            const double alt_err = interp(t_ref, f->prof_time, f->prof_alt, f->n_prof) - alt;
            const double gamma_cmd = clampd(gamma_ref + clampd(alt_err * 2.0e-4, -0.12, 0.12), 0.0, 1.55);
            dir_cmd = normalize(add(scale(up_here, sin(gamma_cmd)), scale(f->track_dir, cos(gamma_cmd))));
            u = clampd(s.ascent_throttle + (v_ref - speed) * 2.0e-3, 0.62, 1.0);
        }
        set_attitude(cmd27, quat_between(x_axis, dir_cmd));
        const double qbar = 0.5 * density(alt) * speed * speed;
        if (qbar > s.bucket_q_on_pa && speed < 500.0) u = u < s.bucket_throttle ? u : s.bucket_throttle;   /* f64::min */
        const double a_meas = norm(s.imu_accel);
        if (a_meas > 34.0) {
            const double lim = u * 34.0 / a_meas;
            u = lim > THROTTLE_MIN ? lim : THROTTLE_MIN;   /* f64::max */
        }
        set_engines(cmd27, u);
        if (speed >= s.meco_speed_mps) {
            f->meco_speed_reached = 1;
            set_engines(cmd27, 0.0);           /* cutoff_with_purge, main.rs:380-383 */
            f->purge_until = t + 5.0;
            set_phase(f, F9FSW_MECO, t);
        }
        break;
    }
    case F9FSW_MECO:
        cmd27[22] = 1.0;
        set_attitude(cmd27, quat_between(x_axis, normalize(f->nav_vel)));
        if (t - f->phase_t0 > 3.0) {
            set_phase(f, F9FSW_FLIP, t);
            return F9FSW_BEYOND_ASCENT;
        }
        break;
    default:
        return F9FSW_BEYOND_ASCENT;
    }
    return 0;
}

/* what the tests look at: phase | phase_t0 | purge_until | t_liftoff | initialized | last_gps_count | radar_alt_m |
 * nav pos(3) vel(3) att(4) | up_pad(3) | track_dir(3) = 23 doubles */
void f9fsw_peek(const f9fsw* f, double* out23) {
    if (!f || !out23) return;
    out23[0] = (double)f->phase;
    out23[1] = f->phase_t0;
    out23[2] = f->purge_until;
    out23[3] = f->t_liftoff;
    out23[4] = (double)f->initialized;
    out23[5] = f->last_gps_count;
    out23[6] = f->radar_alt_m;
    for (int i = 0; i < 3; i++) out23[7 + i] = f->nav_pos.v[i];
    for (int i = 0; i < 3; i++) out23[10 + i] = f->nav_vel.v[i];
    for (int i = 0; i < 4; i++) out23[13 + i] = f->nav_att.q[i];
    for (int i = 0; i < 3; i++) out23[17 + i] = f->up_pad.v[i];
    for (int i = 0; i < 3; i++) out23[20 + i] = f->track_dir.v[i];
}

/* the math.rs helpers on their own, for known-answer checks against the plant's (independently pinned) frames.py */
void f9fsw_ecef_to_geodetic(const double* r, double* lat_lon_alt) {
    ecef_to_geodetic((v3){{r[0], r[1], r[2]}}, &lat_lon_alt[0], &lat_lon_alt[1], &lat_lon_alt[2]);
}
void f9fsw_quat_between(const double* from, const double* to, double* q) {
    const quat r = quat_between((v3){{from[0], from[1], from[2]}}, (v3){{to[0], to[1], to[2]}});
    for (int i = 0; i < 4; i++) q[i] = r.q[i];
}
double f9fsw_density(double alt_m) { return density(alt_m); }
