// sixdof_kernels.hip — fused per-entity six_dof step for gfx950 (MI355X).
//
// Replaces, for worlds whose effectors are per-entity, the whole compiled tick of the reference
//   clear_forces | effectors | calc_accel            libs/nox-py/src/six_dof.rs:137-150,184-203
//   Rk4::compile (4 stages + combination)            libs/nox-py/src/integrator/rk4.rs:87-135
//   semi_implicit_euler                              libs/nox-py/src/integrator/semi_implicit.rs:17-62
// with ONE kernel: one lane = one entity, the whole RK4 tick (or n_ticks of them) in VGPRs.
//
// Data layout in HBM is the reference's column layout (world.rs:23-45): row-major [n,7]/[n,6]
// rows.  A workgroup of 256 lanes owns 256 consecutive rows = one contiguous slab per column
// (14,336 / 12,288 bytes in f64).  Slabs move HBM<->LDS as 16-byte-per-lane coalesced transfers
// (1 KiB per wave instruction); each lane then picks its own row out of LDS.  The strided
// 56-byte-row accesses therefore never reach the memory system.  LDS reads of 56-B rows with
// ds_read_b64 are conflict-free (lane*14 mod 64 is a permutation of the even banks per 32-lane
// group); 48-B rows are 2-way.
//
// Roofline: HBM-bound at n_ticks == 1 (360 algorithmic bytes per entity-step in f64: read
// pos 56 + vel 48 + inertia 56, write pos 56 + vel 48 + accel 48 + force 48); with n_ticks > 1
// the state stays in registers and the kernel is f64-VALU bound.
//
// RK4 quirks of the reference kept on purpose:
// stage positions advance with the INITIAL velocity v0; stage offsets use the global dt, the
// final combination uses the six_dof(time_step=) override.  Consequence used here: stages 1 and
// 2 see the same transform, so when no effector reads the stage velocity their force and
// acceleration are bit-identical and are computed once.
// The reference multiplies the incoming world_accel column by 0 in stage 0 (rk4.rs:96-100);
// that column is therefore not read (finite input assumed).
#include "kernels.hpp"
#include "spatial.hpp"
#include "../../include/sixdof_hip.h"

namespace sixdof {

// ---- slab movement ---------------------------------------------------------------------------------

// Copy `count` elements (count = rows*ROW) between a 16-byte aligned global slab and LDS, 16 bytes per
// lane per iteration, tail element-wise.
template <class T>
__device__ __forceinline__ void slab_to_lds(const T* __restrict__ g, T* __restrict__ l, uint32_t count) {
    constexpr uint32_t V = 16 / sizeof(T);
    using Vec = typename std::conditional<sizeof(T) == 8, double2, float4>::type;
    const uint32_t nvec = count / V;
    const Vec* gv = reinterpret_cast<const Vec*>(g);
    Vec* lv = reinterpret_cast<Vec*>(l);
    for (uint32_t c = threadIdx.x; c < nvec; c += kBlock) lv[c] = gv[c];
    const uint32_t tail = nvec * V + threadIdx.x;
    if (tail < count) l[tail] = g[tail];
}
template <class T>
__device__ __forceinline__ void lds_to_slab(const T* __restrict__ l, T* __restrict__ g, uint32_t count) {
    constexpr uint32_t V = 16 / sizeof(T);
    using Vec = typename std::conditional<sizeof(T) == 8, double2, float4>::type;
    const uint32_t nvec = count / V;
    const Vec* lv = reinterpret_cast<const Vec*>(l);
    Vec* gv = reinterpret_cast<Vec*>(g);
    for (uint32_t c = threadIdx.x; c < nvec; c += kBlock) gv[c] = lv[c];
    const uint32_t tail = nvec * V + threadIdx.x;
    if (tail < count) g[tail] = l[tail];
}

// ---- per-entity effectors (the `sys` of six_dof) -----------------------------------------------------

template <class T>
struct Body {
    Quat<T> q;     // stage attitude (unit)
    Vec3<T> p;     // stage position
    Spatial<T> v;  // stage velocity
    T mass;
};

template <class T>
__device__ __forceinline__ void apply_ops(const StepParams& P, const Vec3<T> (&aux)[kMaxOps], const Body<T>& b,
                                          Spatial<T>& F) {
#pragma unroll
    for (int k = 0; k < kMaxOps; k++) {
        if (k >= (int)P.n_ops) break;  // wave-uniform
        const DevOp& op = P.ops[k];
        switch (op.kind) {  // wave-uniform (kernel argument)
        case SIXDOF_EFF_CONST_WRENCH:
            F.ang = F.ang + Vec3<T>{T(op.p[0]), T(op.p[1]), T(op.p[2])};
            F.lin = F.lin + Vec3<T>{T(op.p[3]), T(op.p[4]), T(op.p[5])};
            break;
        case SIXDOF_EFF_UNIFORM_GRAVITY:
            F.lin = axpy(b.mass, Vec3<T>{T(op.p[0]), T(op.p[1]), T(op.p[2])}, F.lin);
            break;
        case SIXDOF_EFF_BODY_TORQUE:
            F.ang = F.ang + rotate(b.q, aux[k]);
            break;
        case SIXDOF_EFF_BODY_FORCE:
            F.lin = F.lin + rotate(b.q, aux[k]);
            break;
        case SIXDOF_EFF_BALL_DRAG: {
            const Vec3<T> fl = aux[k] - b.v.lin;
            const T v2 = dot(fl, fl);
            const T V = fast_sqrt(v2);
            const T drag = T(0.5) * ((T(op.p[0]) * T(op.p[1])) * v2 * T(op.p[2]));
            F.ang = Vec3<T>{T(0), T(0), T(0)};
            F.lin = axpy(drag / V, fl, F.lin);
            break;
        }
        default:
            break;
        }
    }
}

// calc_accel (six_dof.rs:137-146): alpha = q * ((q^-1 * tau) / I_diag), a = f / m.
// The linear half of the reference is q*((q^-1*f)/m), which is f/m exactly in real arithmetic.
template <class T>
__device__ __forceinline__ Spatial<T> calc_accel(const Quat<T>& q, const Spatial<T>& F, const Vec3<T>& inv_I,
                                                 T inv_m) {
    Spatial<T> a;
    a.ang = rotate(q, hadamard(rotate_inv(q, F.ang), inv_I));
    a.lin = inv_m * F.lin;
    return a;
}

template <class T, int INTEGRATOR>
__global__ __launch_bounds__(kBlock) void sixdof_step_kernel(const StepParams P) {
    // pos | vel | inertia on the way in (20 elems/entity), pos | vel | accel | force on the way out (25)
    __shared__ __attribute__((aligned(16))) T lds[kBlock * 25];
    T* const l_pos = lds;
    T* const l_vel = lds + kBlock * 7;
    T* const l_c = lds + kBlock * 13;  // inertia (in) / accel (out)
    T* const l_force = lds + kBlock * 19;

    const uint32_t row0 = blockIdx.x * kBlock;
    const uint32_t rows = min((uint32_t)kBlock, P.n - row0);
    const uint32_t t = threadIdx.x;

    T* const g_pos = static_cast<T*>(P.pos) + (size_t)row0 * 7;
    T* const g_vel = static_cast<T*>(P.vel) + (size_t)row0 * 6;
    T* const g_accel = static_cast<T*>(P.accel) + (size_t)row0 * 6;
    T* const g_force = static_cast<T*>(P.force) + (size_t)row0 * 6;
    const T* const g_inertia = static_cast<const T*>(P.inertia) + (size_t)row0 * 7;

    slab_to_lds(g_pos, l_pos, rows * 7);
    slab_to_lds(g_vel, l_vel, rows * 6);
    slab_to_lds(g_inertia, l_c, rows * 7);

    // per-entity effector columns: 24-byte rows, read once per launch straight from global
    Vec3<T> aux[kMaxOps];
#pragma unroll
    for (int k = 0; k < kMaxOps; k++) {
        aux[k] = Vec3<T>{T(0), T(0), T(0)};
        if (k < (int)P.n_ops && P.ops[k].aux != nullptr && t < rows) {
            const T* a = static_cast<const T*>(P.ops[k].aux) + (size_t)(row0 + t) * 3;
            aux[k] = Vec3<T>{a[0], a[1], a[2]};
        }
    }
    __syncthreads();

    Quat<T> q0;
    Vec3<T> p0, inv_I;
    Spatial<T> v0, A_out, F_out;
    T mass = T(1), inv_m = T(1);
    const bool active = t < rows;
    if (active) {
        const T* r = l_pos + t * 7;
        q0 = Quat<T>{r[0], r[1], r[2], r[3]};
        p0 = Vec3<T>{r[4], r[5], r[6]};
        const T* s = l_vel + t * 6;
        v0.ang = Vec3<T>{s[0], s[1], s[2]};
        v0.lin = Vec3<T>{s[3], s[4], s[5]};
        const T* m = l_c + t * 7;
        inv_I = Vec3<T>{T(1) / m[0], T(1) / m[1], T(1) / m[2]};
        mass = m[6];
        inv_m = T(1) / mass;

        const T dt_g = T(P.dt_g), dt = T(P.dt);
        for (uint32_t tick = 0; tick < P.n_ticks; tick++) {
            if constexpr (INTEGRATOR == kRk4) {
                const T h1 = dt_g * T(0.5), h3 = dt_g;
                Body<T> b;
                b.mass = mass;
                Spatial<T> F, A, sv, sa;
                const Spatial<T> zero = {{T(0), T(0), T(0)}, {T(0), T(0), T(0)}};
                // stage 0 (c = 0): x0 (+) 0 only renormalises the quaternion
                b.q = normalized(q0);
                b.p = p0;
                b.v = v0;
                F = zero;
                apply_ops(P, aux, b, F);
                A = calc_accel(b.q, F, inv_I, inv_m);
                sv = v0;
                sa = A;
                // stage 1 (c = 1/2): position advanced with v0 (reference quirk), velocity with A0
                b.q = integrate_world(q0, h1 * v0.ang);
                b.p = axpy(h1, v0.lin, p0);
                b.v = axpy(h1, A, v0);
                sv = axpy(T(2), b.v, sv);
                F = zero;
                apply_ops(P, aux, b, F);
                A = calc_accel(b.q, F, inv_I, inv_m);
                sa = axpy(T(2), A, sa);
                // stage 2 (c = 1/2): same transform as stage 1
                b.v = axpy(h1, A, v0);
                sv = axpy(T(2), b.v, sv);
                if (!P.vel_independent) {
                    F = zero;
                    apply_ops(P, aux, b, F);
                    A = calc_accel(b.q, F, inv_I, inv_m);
                }
                sa = axpy(T(2), A, sa);
                // stage 3 (c = 1)
                b.q = integrate_world(q0, h3 * v0.ang);
                b.p = axpy(h3, v0.lin, p0);
                b.v = axpy(h3, A, v0);
                sv = sv + b.v;
                F = zero;
                apply_ops(P, aux, b, F);
                A = calc_accel(b.q, F, inv_I, inv_m);
                sa = sa + A;
                // u' = u + (dt/6)(k1 + 2k2 + 2k3 + k4)
                const T g = dt * T(1.0 / 6.0);
                q0 = integrate_world(q0, g * sv.ang);
                p0 = axpy(g, sv.lin, p0);
                v0 = axpy(g, sa, v0);
                A_out = A;
                F_out = F;
            } else {
                // semi-implicit: a = calc_accel(F(x0,v0)); v' = v0 + dt a; x' = x0 (+) dt v'
                Body<T> b;
                b.mass = mass;
                const T n2 = q0.i * q0.i + q0.j * q0.j + q0.k * q0.k + q0.w * q0.w;
                const T rn = fast_rsqrt(n2);
                b.q = Quat<T>{q0.i * rn, q0.j * rn, q0.k * rn, q0.w * rn};  // rotations are scale-invariant
                b.p = p0;
                b.v = v0;
                Spatial<T> F = {{T(0), T(0), T(0)}, {T(0), T(0), T(0)}};
                apply_ops(P, aux, b, F);
                const Spatial<T> A = calc_accel(b.q, F, inv_I, inv_m);
                v0 = axpy(dt, A, v0);
                q0 = integrate_world(q0, dt * v0.ang);
                p0 = axpy(dt, v0.lin, p0);
                A_out = A;
                F_out = F;
            }
        }
    }
    __syncthreads();  // everyone has consumed the input slabs
    if (active && P.n_ticks > 0) {
        T* r = l_pos + t * 7;
        r[0] = q0.i; r[1] = q0.j; r[2] = q0.k; r[3] = q0.w; r[4] = p0.x; r[5] = p0.y; r[6] = p0.z;
        T* s = l_vel + t * 6;
        s[0] = v0.ang.x; s[1] = v0.ang.y; s[2] = v0.ang.z; s[3] = v0.lin.x; s[4] = v0.lin.y; s[5] = v0.lin.z;
        T* a = l_c + t * 6;
        a[0] = A_out.ang.x; a[1] = A_out.ang.y; a[2] = A_out.ang.z;
        a[3] = A_out.lin.x; a[4] = A_out.lin.y; a[5] = A_out.lin.z;
        T* f = l_force + t * 6;
        f[0] = F_out.ang.x; f[1] = F_out.ang.y; f[2] = F_out.ang.z;
        f[3] = F_out.lin.x; f[4] = F_out.lin.y; f[5] = F_out.lin.z;
    }
    __syncthreads();
    if (P.n_ticks > 0) {
        lds_to_slab(l_pos, g_pos, rows * 7);
        lds_to_slab(l_vel, g_vel, rows * 6);
        lds_to_slab(l_c, g_accel, rows * 6);
        lds_to_slab(l_force, g_force, rows * 6);
    }
}

hipError_t launch_step(const StepParams& p, int integrator, int dtype, hipStream_t stream) {
    if (p.n == 0) return hipSuccess;
    const dim3 grid((p.n + kBlock - 1) / kBlock), block(kBlock);
    if (dtype == SIXDOF_F64) {
        if (integrator == kRk4) hipLaunchKernelGGL((sixdof_step_kernel<double, kRk4>), grid, block, 0, stream, p);
        else hipLaunchKernelGGL((sixdof_step_kernel<double, kSemiImplicit>), grid, block, 0, stream, p);
    } else {
        if (integrator == kRk4) hipLaunchKernelGGL((sixdof_step_kernel<float, kRk4>), grid, block, 0, stream, p);
        else hipLaunchKernelGGL((sixdof_step_kernel<float, kSemiImplicit>), grid, block, 0, stream, p);
    }
    return hipGetLastError();
}

}  // namespace sixdof
