/*
 * sixdof_apollo.h — column / parameter layout of the Apollo-lander rollout model (BASELINE config 4),
 * shared by the HIP model kernel (elodin_amd/csrc/apollo_kernels.hip), the C ABI and the CPU oracle.
 *
 * The model is the reference example examples/apollo-lander/sim.py:517-526 — `engine_response |
 * attitude_control | mass_props | six_dof(lunar_gravity | apply_main_thrust | apply_rcs_torque,
 * SemiImplicit) | ground_contact | derive_telemetry` — with the external guidance computer
 * (examples/apollo-lander/controller/src/main.rs, driven from main.py's post_step over UDP) moved
 * in-line.  One rollout = one row of every column; component names are the reference's.
 */
#ifndef SIXDOF_APOLLO_H
#define SIXDOF_APOLLO_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* per-rollout parameter row: the 17 variables of examples/apollo-lander/spec.toml in sorted-name
 * order (= column order of a plan table, sample.py:112) */
enum {
    APOLLO_P_ATTITUDE_GAIN = 0,
    APOLLO_P_DRY_MASS = 1,
    APOLLO_P_GRAVITY_SCALE = 2,
    APOLLO_P_HORIZONTAL_GAIN = 3,
    APOLLO_P_INIT_ALTITUDE = 4,
    APOLLO_P_INIT_CROSSRANGE_SPEED = 5,
    APOLLO_P_INIT_DOWNRANGE_OFFSET = 6,
    APOLLO_P_INIT_DOWNRANGE_SPEED = 7,
    APOLLO_P_INIT_PITCH_DEG = 8,
    APOLLO_P_INIT_VERTICAL_SPEED = 9,
    APOLLO_P_ISP = 10,
    APOLLO_P_PROPELLANT = 11,
    APOLLO_P_RCS_PROPELLANT = 12,
    APOLLO_P_THROTTLE_RESPONSE_HZ = 13,
    APOLLO_P_THRUST_SCALE = 14,
    APOLLO_P_TRACK_GAIN = 15,
    APOLLO_P_VERTICAL_GAIN = 16,
    APOLLO_N_PARAMS = 17
};

/* guidance row [n,8]: state main.py keeps in Python globals + the controller's throttle latch */
enum {
    APOLLO_G_LAST_THROTTLE = 0,
    APOLLO_G_LAST_ATT = 1, /* ..4 */
    APOLLO_G_LAST_RATE = 5,
    APOLLO_G_FTP_LATCHED = 6,
    APOLLO_G_RESULT_EMITTED = 7,
    APOLLO_N_GUIDANCE = 8
};

/* score accumulators [n,4]: altitude_error_sum, pitch_error_sum, error_samples, reserved */
enum { APOLLO_N_SCORE = 4 };

/* result row [n,12]: el.monte_carlo.result(...) of main.py:259-271 (+ the tick it was emitted on) */
enum {
    APOLLO_R_TOUCHDOWN_SPEED = 0,
    APOLLO_R_HORIZONTAL_SPEED = 1,
    APOLLO_R_FUEL_REMAINING = 2,
    APOLLO_R_RCS_FUEL_REMAINING = 3,
    APOLLO_R_TRAJ_RMSE = 4,
    APOLLO_R_PITCH_RMSE = 5,
    APOLLO_R_DOWNRANGE_MISS = 6,
    APOLLO_R_UPRIGHT_DOT = 7,
    APOLLO_R_LANDED = 8,
    APOLLO_R_SOFT_LANDING = 9,
    APOLLO_R_TICK = 10,
    APOLLO_N_RESULT = 12
};

/* packed per-rollout scalar state [n,16] (one column so a wave moves it as one slab):
 * the reference's scalar components of the `lander` entity, by name */
enum {
    APOLLO_S_THROTTLE = 0,
    APOLLO_S_THROTTLE_CMD = 1,
    APOLLO_S_ATTITUDE_SETPOINT = 2, /* ..5 */
    APOLLO_S_PROPELLANT = 6,
    APOLLO_S_RCS_PROPELLANT = 7,
    APOLLO_S_THRUST = 8,
    APOLLO_S_RCS_TORQUE = 9, /* ..11 */
    APOLLO_S_LANDED = 12,
    APOLLO_S_TOUCHDOWN_SPEED = 13,
    APOLLO_S_TOUCHDOWN_HSPEED = 14,
    APOLLO_S_PITCH = 15, /* derive_telemetry; altitude / vertical / horizontal speed are views of pos / vel */
    APOLLO_N_STATE = 16
};

/* shared descent reference profile (examples/apollo-lander/reference.py build_reference): 1-second grid */
typedef struct sixdof_apollo_tables {
    const double* time_s;
    const double* altitude_m;
    const double* descent_rate_mps;
    const double* pitch_deg;
    const double* horizontal_speed_mps;
    const double* downrange_m;
    uint32_t n;
    uint32_t guidance_period_ticks; /* round(120 / 24) = 5 */
    uint64_t max_ticks;             /* result is emitted at tick >= max_ticks - 1 if not landed */
    /* Telemetry batch of the server loop (libs/nox-py/src/impeller2_server.rs:553-678,790-791): the world runs
     * `ticks_per_telemetry` = simulation_rate / telemetry_rate = 120 / 40 = 3 ticks back to back, THEN main.py's post_step
     * is called once with end_tick = (ticks completed) - 1.  So the guidance exchange (end_tick % 5 == 0), the RMSE
     * samples and the result check happen at batch ends only, and the command columns change between batches only.
     * 0 = 3.  1 reproduces a post_step after every tick. */
    uint32_t ticks_per_telemetry;
    uint32_t reserved;
} sixdof_apollo_tables;

#ifdef __cplusplus
}
#endif
#endif
