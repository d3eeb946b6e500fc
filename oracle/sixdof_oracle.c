/*
 * sixdof_oracle.c — CPU restatement of the reference `six_dof` step.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * library.  The product (elodin_amd/, include/) never links, imports or calls it.
 *
 * Why a restatement: the reference path (nox-py six_dof -> Noxpr -> JAX -> StableHLO ->
 * cranelift-mlir JIT) needs rustc 1.98 + jax 0.10.0, neither present here; the arithmetic it
 * executes is fully in-tree and is restated below in the reference's exact operation order.
 * Third-party code on the reference's execution path (not under /root/reference): jax 0.10.0
 * (libs/nox-py/pyproject.toml:14), cranelift-codegen/jit 0.130.2 (Cargo.lock), libm.
 * All ops on this path are IEEE +,-,*,/,sqrt, so a no-FMA C build (-ffp-contract=off)
 * reproduces them up to summation order inside 3/4-element dots.
 *
 * PARITY PINNED by the reference's own golden data (tests/test_oracle_golden.py):
 *   G1 scripts/ci/baseline/three-body-csv  (RK4 + edge_fold gravity, 100 ticks)
 *   G2 scripts/ci/baseline/ball-csv        (RK4 + gravity + drag, 100 ticks)
 *   G3 scripts/ci/baseline/cube-sat-csv    (SemiImplicit, torque != 0, non-uniform inertia diagonal; 100 ticks of
 *      ore_sat + earth, teacher-forced with the recorded `force` rows: tests/test_oracle_semi_implicit_golden.py
 *      — pins semi_implicit_tick and the angular half of calc_accel to 5e-16)
 *   K1-K8 unit-test known answers (tests/test_oracle_kat.py)
 * NOT pinned by golden data: the softened n-body term at N>35.
 *
 * Each function cites the reference file:line it follows (paths relative to the reference).
 */
#include "sixdof_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- libs/nox/src/quaternion.rs, libs/nox/src/vector.rs ------------------------------- */

/* Vector::dot — sequential left-to-right accumulation. vector.rs:110-112 */
static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static double dot4(const double* a, const double* b) {
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
}

/* Hamilton product, scalar-last [i,j,k,w]. quaternion.rs:268-281 */
void orc_quat_mul(const double* l, const double* r, double* o) {
    const double li = l[0], lj = l[1], lk = l[2], lw = l[3];
    const double ri = r[0], rj = r[1], rk = r[2], rw = r[3];
    const double i = lw * ri + li * rw + lj * rk - lk * rj;
    const double j = lw * rj - li * rk + lj * rw + lk * ri;
    const double k = lw * rk + li * rj - lj * ri + lk * rw;
    const double w = lw * rw - li * ri - lj * rj - lk * rk;
    o[0] = i; o[1] = j; o[2] = k; o[3] = w;
}

/* inverse = conjugate / norm_squared. quaternion.rs:141-155 */
void orc_quat_inverse(const double* q, double* o) {
    const double d = dot4(q, q);
    o[0] = -q[0] / d; o[1] = -q[1] / d; o[2] = -q[2] / d; o[3] = q[3] / d;
}

/* normalize = q / sqrt(q.q). quaternion.rs:147-149, vector.rs:114-122 */
void orc_quat_normalize(const double* q, double* o) {
    const double n = sqrt(dot4(q, q));
    o[0] = q[0] / n; o[1] = q[1] / n; o[2] = q[2] / n; o[3] = q[3] / n;
}

/* q * v = (q (x) [v,0] (x) inverse(q)).xyz — inverse recomputed per call. quaternion.rs:283-305 */
void orc_quat_rotate(const double* q, const double* v, double* o) {
    double vq[4] = {v[0], v[1], v[2], 0.0}, inv[4], t[4], r[4];
    orc_quat_inverse(q, inv);
    orc_quat_mul(q, vq, t);
    orc_quat_mul(t, inv, r);
    o[0] = r[0]; o[1] = r[1]; o[2] = r[2];
}

/* Quaternion::from_axis_angle: normalise axis, [axis*sin(a/2), cos(a/2)]. quaternion.rs:157-169 */
void orc_quat_from_axis_angle(const double* axis, double angle, double* o) {
    const double n = sqrt(dot3(axis, axis));
    const double half = angle / 2.0;
    const double s = sin(half), c = cos(half);
    o[0] = (axis[0] / n) * s; o[1] = (axis[1] / n) * s; o[2] = (axis[2] / n) * s; o[3] = c;
}

/* Quaternion::integrate_body: q + q (x) (delta/2, 0), normalised (RIGHT multiply). quaternion.rs:176-182 */
void orc_quat_integrate_body(const double* q, const double* delta, double* o) {
    double ho[4] = {delta[0] / 2.0, delta[1] / 2.0, delta[2] / 2.0, 0.0}, t[4], s[4];
    orc_quat_mul(q, ho, t);
    for (int i = 0; i < 4; i++) s[i] = q[i] + t[i];
    orc_quat_normalize(s, o);
}

/* ---- libs/nox/src/spatial.rs ---------------------------------------------------------- */

/* SpatialTransform + SpatialMotion: q' = normalize(q + (w/2,0) (x) q); p' = p + v. spatial.rs:530-549 */
void orc_transform_add_motion(const double* x, const double* m, double* o) {
    double ho[4] = {m[0] / 2.0, m[1] / 2.0, m[2] / 2.0, 0.0}, t[4], s[4];
    orc_quat_mul(ho, x, t);
    for (int i = 0; i < 4; i++) s[i] = x[i] + t[i];
    orc_quat_normalize(s, o);
    o[4] = x[4] + m[3]; o[5] = x[5] + m[4]; o[6] = x[6] + m[5];
}

/* SpatialTransform * SpatialTransform (only used by KAT K5). spatial.rs:131-143:
 * angular = a.q (x) b.q ; linear = a.lin + a.q * b.lin */
void orc_transform_mul(const double* a, const double* b, double* o) {
    double r[3];
    orc_quat_mul(a, b, o);
    orc_quat_rotate(a, b + 4, r);
    o[4] = a[4] + r[0]; o[5] = a[5] + r[1]; o[6] = a[6] + r[2];
}

/* calc_accel: six_dof.rs:137-146 with SpatialForce/SpatialInertia (spatial.rs:353-361) and
 * Quaternion * SpatialForce/Motion (spatial.rs:571-593). F=[tau,f], I=[Ixx,Iyy,Izz,px,py,pz,m]. */
void orc_calc_accel(const double* F, const double* I, const double* x, double* a) {
    double qi[4], bt[3], bf[3], ba_ang[3], ba_lin[3];
    orc_quat_inverse(x, qi);
    orc_quat_rotate(qi, F, bt);
    orc_quat_rotate(qi, F + 3, bf);
    ba_lin[0] = bf[0] / I[6]; ba_lin[1] = bf[1] / I[6]; ba_lin[2] = bf[2] / I[6];
    ba_ang[0] = bt[0] / I[0]; ba_ang[1] = bt[1] / I[1]; ba_ang[2] = bt[2] / I[2];
    orc_quat_rotate(x, ba_ang, a);
    orc_quat_rotate(x, ba_lin, a + 3);
}

/* ---- globals / ids -------------------------------------------------------------------- */

/* ComponentId::new: FNV-1a-64 & !(1<<63). impeller2/src/types.rs:39-44 (const-fnv1a-hash 1.1.0) */
uint64_t orc_component_id(const char* name) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (const unsigned char* p = (const unsigned char*)name; *p; ++p) {
        h ^= (uint64_t)*p;
        h *= 0x100000001b3ull;
    }
    return h & ~(1ull << 63);
}

/* Duration::from_secs_f64(1/rate).as_secs_f64(): ns-quantised. world_builder.rs:221, world.rs:185-191 */
double orc_quantize_time_step(double rate_hz) {
    const long double ns = nearbyintl((long double)(1.0 / rate_hz) * 1.0e9L);
    const uint64_t total = (uint64_t)ns;
    const uint64_t secs = total / 1000000000ull, nanos = total % 1000000000ull;
    return (double)secs + (double)nanos / 1.0e9;
}

/* ---- effectors (the `sys` of six_dof; run on every stage) ------------------------------ */

static void effectors(const orc_world* w, const double* xs, const double* vs, double* F) {
    const uint64_t n = w->n;
    for (uint32_t k = 0; k < w->n_ops; k++) {
        const sixdof_effector_op* op = &w->ops[k];
        const double* aux = w->aux[k];
        switch (op->kind) {
        case SIXDOF_EFF_CONST_WRENCH: /* test_all.py:353-356 */
            for (uint64_t i = 0; i < n; i++)
                for (int c = 0; c < 6; c++) F[6 * i + c] = F[6 * i + c] + op->p[c];
            break;
        case SIXDOF_EFF_UNIFORM_GRAVITY: /* examples/ball/sim.py:57-59: f + SpatialForce(linear=g*m) */
            for (uint64_t i = 0; i < n; i++) {
                const double m = w->inertia[7 * i + 6];
                for (int c = 0; c < 3; c++) {
                    F[6 * i + c] = F[6 * i + c] + 0.0;
                    F[6 * i + 3 + c] = F[6 * i + 3 + c] + op->p[c] * m;
                }
            }
            break;
        case SIXDOF_EFF_BODY_TORQUE: /* apollo-lander/sim.py:396-398: force + SpatialForce(torque=q @ t) */
            for (uint64_t i = 0; i < n; i++) {
                double r[3];
                orc_quat_rotate(xs + 7 * i, aux + 3 * i, r);
                for (int c = 0; c < 3; c++) {
                    F[6 * i + c] = F[6 * i + c] + r[c];
                    F[6 * i + 3 + c] = F[6 * i + 3 + c] + 0.0;
                }
            }
            break;
        case SIXDOF_EFF_BODY_FORCE: /* apollo-lander/sim.py:391-394 */
            for (uint64_t i = 0; i < n; i++) {
                double r[3];
                orc_quat_rotate(xs + 7 * i, aux + 3 * i, r);
                for (int c = 0; c < 3; c++) {
                    F[6 * i + c] = F[6 * i + c] + 0.0;
                    F[6 * i + 3 + c] = F[6 * i + 3 + c] + r[c];
                }
            }
            break;
        case SIXDOF_EFF_WORLD_TORQUE: /* force + SpatialForce(torque=t), t a per-entity world-frame column */
            for (uint64_t i = 0; i < n; i++)
                for (int c = 0; c < 3; c++) {
                    F[6 * i + c] = F[6 * i + c] + aux[3 * i + c];
                    F[6 * i + 3 + c] = F[6 * i + 3 + c] + 0.0;
                }
            break;
        case SIXDOF_EFF_WORLD_FORCE: /* force + SpatialForce(linear=f) */
            for (uint64_t i = 0; i < n; i++)
                for (int c = 0; c < 3; c++) {
                    F[6 * i + c] = F[6 * i + c] + 0.0;
                    F[6 * i + 3 + c] = F[6 * i + 3 + c] + aux[3 * i + c];
                }
            break;
        case SIXDOF_EFF_BALL_DRAG: /* examples/ball/sim.py:92-116 */
            for (uint64_t i = 0; i < n; i++) {
                double fl[3];
                for (int c = 0; c < 3; c++) fl[c] = aux[3 * i + c] - vs[6 * i + 3 + c];
                const double V = sqrt(fl[0] * fl[0] + fl[1] * fl[1] + fl[2] * fl[2]);
                const double drag = 0.5 * ((op->p[0] * op->p[1]) * (V * V) * op->p[2]);
                for (int c = 0; c < 3; c++) {
                    F[6 * i + c] = 0.0; /* el.SpatialForce(linear=...) has zero torque */
                    F[6 * i + 3 + c] = F[6 * i + 3 + c] + drag * (fl[c] / V);
                }
            }
            break;
        case SIXDOF_EFF_EDGE_GRAVITY_NEWTON:   /* examples/three-body/main.py:56-78 */
        case SIXDOF_EFF_EDGE_GRAVITY_SOFTENED: /* examples/n-body/sim.py:344-369 */
        {
            /* GraphQuery.edge_fold (graph.rs:239-361, __init__.py:454-557): per source in
             * ascending id, sequential left fold over its out-edges in spawn order, starting from
             * init_value (zero Force); the result REPLACES Force on the source rows. */
            uint8_t* is_src = (uint8_t*)calloc(n ? n : 1, 1);
            double* acc = (double*)calloc(6 * (n ? n : 1), sizeof(double));
            for (uint64_t e = 0; e < w->n_edges; e++) {
                const uint32_t a = w->edge_src[e], b = w->edge_dst[e];
                const double* pa = xs + 7 * a + 4;
                const double* pb = xs + 7 * b + 4;
                const double ma = w->inertia[7 * a + 6], mb = w->inertia[7 * b + 6];
                is_src[a] = 1;
                if (op->kind == SIXDOF_EFF_EDGE_GRAVITY_NEWTON) {
                    double r[3] = {pa[0] - pb[0], pa[1] - pb[1], pa[2] - pb[2]};
                    const double nrm = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
                    const double GMm = op->p[0] * mb * ma; /* G * M * m, M = b mass, m = a mass */
                    const double den = nrm * nrm * nrm;
                    for (int c = 0; c < 3; c++) {
                        const double f = GMm * r[c] / den;
                        acc[6 * a + 3 + c] = acc[6 * a + 3 + c] - f;
                        acc[6 * a + c] = 0.0; /* el.Force(linear=...) */
                    }
                } else {
                    double r[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
                    const double d2 = (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]) + op->p[1];
                    const double inv = 1.0 / sqrt(d2);
                    const double inv3 = inv * inv * inv;
                    const double s = op->p[0] * ma * mb * inv3;
                    for (int c = 0; c < 3; c++) {
                        acc[6 * a + c] = acc[6 * a + c] + 0.0;
                        acc[6 * a + 3 + c] = acc[6 * a + 3 + c] + s * r[c];
                    }
                }
            }
            for (uint64_t i = 0; i < n; i++)
                if (is_src[i]) memcpy(F + 6 * i, acc + 6 * i, 6 * sizeof(double));
            free(acc);
            free(is_src);
            break;
        }
        case SIXDOF_EFF_ALLPAIRS_GRAVITY_SOFTENED: /* complete graph in n-body spawn order */
            /* sources are independent; each source's fold over its targets stays sequential (the reference order) */
#pragma omp parallel for schedule(static) num_threads(w->pair_threads > 0 ? w->pair_threads : 1)
            for (int64_t a_ = 0; a_ < (int64_t)n; a_++) {
                const uint64_t a = (uint64_t)a_;
                double acc[6] = {0, 0, 0, 0, 0, 0};
                const double* pa = xs + 7 * a + 4;
                const double ma = w->inertia[7 * a + 6];
                for (uint64_t b = 0; b < n; b++) {
                    if (b == a) continue;
                    const double* pb = xs + 7 * b + 4;
                    const double mb = w->inertia[7 * b + 6];
                    double r[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
                    const double d2 = (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]) + op->p[1];
                    const double inv = 1.0 / sqrt(d2);
                    const double inv3 = inv * inv * inv;
                    const double s = op->p[0] * ma * mb * inv3;
                    for (int c = 0; c < 3; c++) acc[3 + c] = acc[3 + c] + s * r[c];
                }
                if (n > 1) memcpy(F + 6 * a, acc, sizeof(acc));
            }
            break;
        default:
            break;
        }
    }
}

/* clear_forces | effectors | calc_accel  (six_dof.rs:148-150,184-186) */
static void pipe(const orc_world* w, const double* xs, const double* vs, double* F, double* A) {
    const uint64_t n = w->n;
    memset(F, 0, 6 * n * sizeof(double));
    effectors(w, xs, vs, F);
    for (uint64_t i = 0; i < n; i++) orc_calc_accel(F + 6 * i, w->inertia + 7 * i, xs + 7 * i, A + 6 * i);
}

/* ---- integrators ---------------------------------------------------------------------- */

/* Rk4::compile, integrator/rk4.rs:87-135.  Quirks kept: stage positions advance with v0
 * (WorldVel is in both U and DU and is reset to v0 by init_u.insert_into_builder, :110-121);
 * stage offsets use the GLOBAL simulation_time_step (:96-100), final combination uses the
 * six_dof(time_step=) override when given (:93,129). */
static void rk4_tick(orc_world* w, double* scratch) {
    const uint64_t n = w->n;
    double* xs = scratch;            /* [n,7] */
    double* vs = xs + 7 * n;         /* [n,6] */
    double* F = vs + 6 * n;          /* [n,6] */
    double* A[4];                    /* stage accelerations */
    double* V[4];                    /* stage velocities */
    double* p = F + 6 * n;
    for (int s = 0; s < 4; s++) { A[s] = p; p += 6 * n; V[s] = p; p += 6 * n; }
    static const double C[4] = {0.0, 0.5, 0.5, 1.0};
    const double dt_g = w->simulation_time_step;
    const double dt = w->has_time_step ? w->time_step : dt_g;

    for (int s = 0; s < 4; s++) {
        const double h = dt_g * C[s];
        const double* du_a = (s == 0) ? w->world_accel : A[s - 1];
        for (uint64_t i = 0; i < n; i++) {
            double hv[6];
            for (int c = 0; c < 6; c++) hv[c] = h * w->world_vel[6 * i + c];
            orc_transform_add_motion(w->world_pos + 7 * i, hv, xs + 7 * i);
            for (int c = 0; c < 6; c++) vs[6 * i + c] = w->world_vel[6 * i + c] + h * du_a[6 * i + c];
        }
        pipe(w, xs, vs, F, A[s]);
        memcpy(V[s], vs, 6 * n * sizeof(double));
    }
    const double g = dt * (1.0 / 6.0);
    for (uint64_t i = 0; i < n; i++) {
        double sv[6], sa[6];
        for (int c = 0; c < 6; c++) {
            const uint64_t j = 6 * i + c;
            sv[c] = g * (V[0][j] + 2.0 * V[1][j] + 2.0 * V[2][j] + V[3][j]);
            sa[c] = g * (A[0][j] + 2.0 * A[1][j] + 2.0 * A[2][j] + A[3][j]);
        }
        double xn[7];
        orc_transform_add_motion(w->world_pos + 7 * i, sv, xn);
        memcpy(w->world_pos + 7 * i, xn, sizeof(xn));
        for (int c = 0; c < 6; c++) w->world_vel[6 * i + c] = w->world_vel[6 * i + c] + sa[c];
    }
    memcpy(w->world_accel, A[3], 6 * n * sizeof(double));
    memcpy(w->force, F, 6 * n * sizeof(double));
}

/* semi_implicit_euler[_with_dt], integrator/semi_implicit.rs:17-62, pipe six_dof.rs:176-180 */
static void semi_implicit_tick(orc_world* w, double* scratch) {
    const uint64_t n = w->n;
    double* F = scratch;
    double* A = F + 6 * n;
    const double dt = w->has_time_step ? w->time_step : w->simulation_time_step;
    pipe(w, w->world_pos, w->world_vel, F, A);
    for (uint64_t i = 0; i < n; i++) {
        double dv[6], xn[7];
        for (int c = 0; c < 6; c++) w->world_vel[6 * i + c] = w->world_vel[6 * i + c] + dt * A[6 * i + c];
        for (int c = 0; c < 6; c++) dv[c] = dt * w->world_vel[6 * i + c];
        orc_transform_add_motion(w->world_pos + 7 * i, dv, xn);
        memcpy(w->world_pos + 7 * i, xn, sizeof(xn));
    }
    memcpy(w->world_accel, A, 6 * n * sizeof(double));
    memcpy(w->force, F, 6 * n * sizeof(double));
}

/* increment_sim_tick | six_dof   (globals.rs:42-44, world_builder.rs:1762) */
int orc_step(orc_world* w, uint64_t n_ticks) {
    const uint64_t n = w->n ? w->n : 1;
    double* scratch = (double*)malloc((7 + 6 + 6 + 8 * 6) * n * sizeof(double));
    if (!scratch) return -1;
    for (uint64_t t = 0; t < n_ticks; t++) {
        w->tick += 1;
        if (w->n == 0) continue;
        if (w->integrator == SIXDOF_INTEGRATOR_RK4) rk4_tick(w, scratch);
        else semi_implicit_tick(w, scratch);
    }
    free(scratch);
    return 0;
}

/* Entity-id -> row resolution for edges (query.rs:599-621: constant u32 gather indices).
 * Returns 0, or -1 if an id is not a Body row. */
int orc_resolve_edges(const uint64_t* body_ids, uint64_t n, const uint64_t* from_ids, const uint64_t* to_ids,
                      uint64_t n_edges, uint32_t* src_rows, uint32_t* dst_rows) {
    for (uint64_t e = 0; e < n_edges; e++) {
        int64_t a = -1, b = -1;
        for (uint64_t i = 0; i < n; i++) {
            if (body_ids[i] == from_ids[e]) a = (int64_t)i;
            if (body_ids[i] == to_ids[e]) b = (int64_t)i;
        }
        if (a < 0 || b < 0) return -1;
        src_rows[e] = (uint32_t)a;
        dst_rows[e] = (uint32_t)b;
    }
    return 0;
}

/* Multi-core timing variant for the bench's cpu_baseline leg: entities are independent when no
 * pair effector is present, so the world is split into contiguous row blocks, each stepped by
 * the scalar code above on its own thread (the reference tick itself is single-threaded:
 * libs/cranelift-mlir/ARCHITECTURE.md:25-27).  Results are identical to orc_step. */
int orc_step_omp(orc_world* w, uint64_t n_ticks, int threads) {
    for (uint32_t k = 0; k < w->n_ops; k++)
        if (SIXDOF_EFF_IS_PAIR(w->ops[k].kind)) return orc_step(w, n_ticks);
    if (threads < 1) threads = 1;
    int rc = 0;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int t = 0; t < threads; t++) {
        const uint64_t lo = w->n * (uint64_t)t / (uint64_t)threads;
        const uint64_t hi = w->n * (uint64_t)(t + 1) / (uint64_t)threads;
        orc_world s = *w;
        s.n = hi - lo;
        s.world_pos += 7 * lo; s.world_vel += 6 * lo; s.world_accel += 6 * lo;
        s.force += 6 * lo; s.inertia += 7 * lo;
        for (uint32_t k = 0; k < s.n_ops; k++) if (s.aux[k]) s.aux[k] += 3 * lo;
        if (orc_step(&s, n_ticks) != 0) rc = -1;
    }
    w->tick += n_ticks;
    return rc;
}
