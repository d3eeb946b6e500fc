/* sixdof_oracle.h — CPU oracle of the reference six_dof step.  TEST INFRASTRUCTURE ONLY:
 * nothing under elodin_amd/ or include/ may include, link or load it (see sixdof_oracle.c). */
#ifndef SIXDOF_ORACLE_H
#define SIXDOF_ORACLE_H
#include <stdint.h>
#include "../include/sixdof_hip.h" /* shared descriptor enums/structs only; no product code */

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_OPS 8

/* Columns in the reference's row-major layout (world.rs:23-45). Stepped in place. */
typedef struct orc_world {
    uint64_t n;
    double* world_pos;   /* [n,7] qx qy qz qw x y z */
    double* world_vel;   /* [n,6] wx wy wz vx vy vz */
    double* world_accel; /* [n,6] */
    double* force;       /* [n,6] tau f */
    double* inertia;     /* [n,7] Ixx Iyy Izz px py pz m */
    uint64_t tick;
    double simulation_time_step;
    double time_step;
    int32_t has_time_step;
    int32_t integrator;
    uint32_t n_ops;
    uint32_t pair_threads; /* OpenMP threads for the all-pairs fold (0/1 = serial); results are identical */
    sixdof_effector_op ops[ORC_MAX_OPS];
    const double* aux[ORC_MAX_OPS]; /* per-op [n,3] column or NULL */
    const uint32_t* edge_src;       /* resolved row indices, spawn order */
    const uint32_t* edge_dst;
    uint64_t n_edges;
} orc_world;

void orc_quat_mul(const double* l, const double* r, double* o);
void orc_quat_inverse(const double* q, double* o);
void orc_quat_normalize(const double* q, double* o);
void orc_quat_rotate(const double* q, const double* v, double* o);
void orc_quat_from_axis_angle(const double* axis, double angle, double* o);
void orc_quat_integrate_body(const double* q, const double* delta, double* o);
void orc_transform_add_motion(const double* x, const double* m, double* o);
void orc_transform_mul(const double* a, const double* b, double* o);
void orc_calc_accel(const double* F, const double* I, const double* x, double* a);
uint64_t orc_component_id(const char* name);
double orc_quantize_time_step(double rate_hz);
int orc_step(orc_world* w, uint64_t n_ticks);
int orc_step_omp(orc_world* w, uint64_t n_ticks, int threads);
int orc_resolve_edges(const uint64_t* body_ids, uint64_t n, const uint64_t* from_ids, const uint64_t* to_ids,
                      uint64_t n_edges, uint32_t* src_rows, uint32_t* dst_rows);
#ifdef __cplusplus
}
#endif
#endif
