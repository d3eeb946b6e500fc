/* apollo_oracle.h — CPU oracle of the Apollo-lander rollout.  TEST INFRASTRUCTURE ONLY (see apollo_oracle.c). */
#ifndef APOLLO_ORACLE_H
#define APOLLO_ORACLE_H
#include "../include/sixdof_apollo.h" /* shared layout enums only */
#include "sixdof_oracle.h"

typedef struct apollo_world {
    uint64_t n;
    double *world_pos, *world_vel, *world_accel, *force, *inertia;
    /* the lander's scalar components, one array each (reference component names) */
    double *throttle, *throttle_cmd, *attitude_setpoint /*[n,4]*/, *propellant, *rcs_propellant, *thrust,
        *rcs_torque /*[n,3]*/, *landed, *touchdown_speed, *touchdown_horizontal_speed;
    double *altitude, *vertical_speed, *horizontal_speed, *pitch;
    const double* params; /* [n,17] */
    double* guidance;     /* [n,8]  */
    double* score;        /* [n,4]  */
    double* result;       /* [n,12] */
    const double *ref_time, *ref_altitude, *ref_rate, *ref_pitch, *ref_hspeed, *ref_downrange;
    uint32_t n_ref;
    uint32_t guidance_period;
    uint64_t max_ticks;
    uint64_t tick;
    double simulation_time_step;
    uint32_t ticks_per_telemetry; /* 0 = 3 (simulation_rate 120 / telemetry_rate 40, main.py:274-283) */
} apollo_world;

int apollo_step(apollo_world* w, uint64_t n_ticks, int threads);
#endif
