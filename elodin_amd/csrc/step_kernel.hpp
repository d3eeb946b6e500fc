// step_kernel.hpp — the fused per-entity six_dof step kernel template for gfx950 (MI355X).
//
// Included by sixdof_kernels.hip (built-in effector pipes) and by run-time generated translation units
// (elodin_amd/codegen.py: user-defined effector pipes), so both instantiate the SAME kernel.
//
// Replaces, for worlds whose effectors are per-entity, the whole compiled tick of the reference
//   clear_forces | effectors | calc_accel            libs/nox-py/src/six_dof.rs:137-150,184-203
//   Rk4::compile (4 stages + combination)            libs/nox-py/src/integrator/rk4.rs:87-135
//   semi_implicit_euler                              libs/nox-py/src/integrator/semi_implicit.rs:17-62
// with ONE kernel: one lane = one entity, the whole tick (or n_ticks of them) in VGPRs.
//
// Memory plan.  Columns stay in HBM in the reference's row-major layout (world.rs:23-45): [n,7] /
// [n,6] rows of 56 / 48 bytes.  A workgroup is ONE wavefront and owns 64 consecutive rows, i.e. one
// contiguous 3,584- / 3,072-byte slab per column.  Slabs are pulled HBM -> LDS with
// `global_load_lds_dwordx4` (LDS-DMA: 16 B per lane, 1 KiB per wave instruction, no VGPR round
// trip, all of a wave's ~10 KiB in flight at once); each lane then reads its own row from LDS
// (56-B rows with ds_read_b64 are bank-conflict free: lane*14 mod 64 permutes the even banks of a
// 32-lane group).  Results go rows -> LDS -> 16-B-per-lane coalesced stores.  The strided row
// accesses therefore never reach the memory system, and PMC traffic equals the algorithmic bytes
// (profiles/).  Single-wave workgroups need no cross-wave barrier and let 8+ independent waves per
// CU overlap their load / compute / store phases.
//
// Roofline: HBM-bound at n_ticks == 1 (f64: read pos 56 + vel 48 + inertia 56 [+24 per [n,3]
// effector column], write pos 56 + vel 48 + accel 48 + force 48 = 360 B per entity-step);
// with n_ticks > 1 the state stays in registers and the kernel is f64-VALU bound.
//
// RK4 quirks of the reference kept on purpose: stage positions advance with the
// INITIAL velocity v0; stage offsets use the global dt, the final combination uses the
// six_dof(time_step=) override.  Consequence used here: stages 1 and 2 see the same transform, so
// when no effector reads the stage velocity their wrench and acceleration are bit-identical and are
// computed once.  The reference multiplies the incoming world_accel column by 0 in stage 0
// (rk4.rs:96-100); that column is therefore not read (finite input assumed).
#pragma once
#include <type_traits>

#include "effectors.hpp"
#include "kernels.hpp"
#include "spatial.hpp"

namespace sixdof {

constexpr int kWave = 64;

typedef __attribute__((address_space(1))) const void* global_cptr;
typedef __attribute__((address_space(3))) void* lds_ptr;

// ---- slab movement ---------------------------------------------------------------------------------

// Whole-wave slab of BYTES bytes (multiple of 16), global -> LDS by LDS-DMA.
// NT = non-temporal cache policy (aux = 2) for worlds far larger than the 256 MiB Infinity Cache, where every
// byte is touched exactly once per tick and retaining it only evicts useful lines.
template <int BYTES, int NT>
__device__ __forceinline__ void slab_dma_in(const char* __restrict__ g, char* l, uint32_t lane) {
    constexpr int kFull = BYTES / 1024, kRem = (BYTES % 1024) / 16;
    constexpr int kAux = NT == 1 ? 2 : 0;
#pragma unroll
    for (int i = 0; i < kFull; i++)
        __builtin_amdgcn_global_load_lds((global_cptr)(g + i * 1024 + lane * 16), (lds_ptr)(l + i * 1024), 16, 0,
                                         kAux);
    if (kRem && lane < (uint32_t)kRem)
        __builtin_amdgcn_global_load_lds((global_cptr)(g + kFull * 1024 + lane * 16), (lds_ptr)(l + kFull * 1024), 16,
                                         0, kAux);
}

// Whole-wave slab, LDS -> global: read every chunk first, then issue the stores back to back.
typedef float vfloat4 __attribute__((ext_vector_type(4)));
// Store policy: 0 = plain (lines stay dirty in the XCD's L2 until the kernel-end write-back), 1 = non-temporal,
// 2 = write-through (`sc1`): the bytes leave for memory as the store issues, so the end of the kernel has nothing
// left to flush (MI355X_MICROARCH.md "publish-large").
template <int NT>
__device__ __forceinline__ void store16(char* g, vfloat4 v) {
    if constexpr (NT == 1) __builtin_nontemporal_store(v, reinterpret_cast<vfloat4*>(g));
    else if constexpr (NT == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(g), "v"(v) : "memory");
    else *reinterpret_cast<vfloat4*>(g) = v;
}
template <int BYTES, int NT>
__device__ __forceinline__ void slab_out(const char* l, char* __restrict__ g, uint32_t lane) {
    constexpr int kFull = BYTES / 1024, kRem = (BYTES % 1024) / 16;
    vfloat4 tmp[kFull + 1];
#pragma unroll
    for (int i = 0; i < kFull; i++) tmp[i] = *reinterpret_cast<const vfloat4*>(l + i * 1024 + lane * 16);
    if (kRem && lane < (uint32_t)kRem) tmp[kFull] = *reinterpret_cast<const vfloat4*>(l + kFull * 1024 + lane * 16);
#pragma unroll
    for (int i = 0; i < kFull; i++) store16<NT>(g + i * 1024 + lane * 16, tmp[i]);
    if (kRem && lane < (uint32_t)kRem) store16<NT>(g + kFull * 1024 + lane * 16, tmp[kFull]);
}

// Ragged last wave (rows < 64): element-wise.
template <class T>
__device__ __forceinline__ void slab_in_tail(const T* __restrict__ g, T* l, uint32_t count, uint32_t lane) {
    for (uint32_t e = lane; e < count; e += kWave) l[e] = g[e];
}
template <class T>
__device__ __forceinline__ void slab_out_tail(const T* l, T* __restrict__ g, uint32_t count, uint32_t lane) {
    for (uint32_t e = lane; e < count; e += kWave) g[e] = l[e];
}

// ---- the kernel --------------------------------------------------------------------------------------

template <class T, int INTEGRATOR, class PIPE, int NT>
__global__ __launch_bounds__(kWave) void sixdof_step_kernel(const StepParams P) {
    // pos | vel | inertia on the way in (20 elems/entity); pos | vel | accel | force on the way out (25)
    __shared__ __attribute__((aligned(16))) T lds[kWave * 25];
    T* const l_pos = lds;
    T* const l_vel = lds + kWave * 7;
    T* const l_c = lds + kWave * 13;  // inertia (in) / accel (out)
    T* const l_force = lds + kWave * 19;

    const uint32_t row0 = blockIdx.x * kWave;
    const uint32_t rows = min((uint32_t)kWave, P.n - row0);
    const uint32_t t = threadIdx.x;
    const bool full = rows == kWave;  // wave-uniform

    T* const g_pos = static_cast<T*>(P.pos) + (size_t)row0 * 7;
    T* const g_vel = static_cast<T*>(P.vel) + (size_t)row0 * 6;
    T* const g_accel = static_cast<T*>(P.accel) + (size_t)row0 * 6;
    T* const g_force = static_cast<T*>(P.force) + (size_t)row0 * 6;
    const T* const g_inertia = static_cast<const T*>(P.inertia) + (size_t)row0 * 7;

    if (full) {
        slab_dma_in<kWave * 7 * sizeof(T), NT>(reinterpret_cast<const char*>(g_pos), reinterpret_cast<char*>(l_pos), t);
        slab_dma_in<kWave * 6 * sizeof(T), NT>(reinterpret_cast<const char*>(g_vel), reinterpret_cast<char*>(l_vel), t);
        slab_dma_in<kWave * 7 * sizeof(T), NT>(reinterpret_cast<const char*>(g_inertia), reinterpret_cast<char*>(l_c), t);
    } else {
        slab_in_tail(g_pos, l_pos, rows * 7, t);
        slab_in_tail(g_vel, l_vel, rows * 6, t);
        slab_in_tail(g_inertia, l_c, rows * 7, t);
    }

    // per-entity effector columns: 24-byte rows, read once per launch straight from global
    const bool active = t < rows;
    Vec3<T> aux[kMaxOps];
#pragma unroll
    for (int k = 0; k < kMaxOps; k++) aux[k] = Vec3<T>{T(0), T(0), T(0)};
    auto load_aux = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr (PIPE::template uses_aux<k>()) {
            if (k < (int)P.n_ops && P.ops[k].aux != nullptr && active) {
                const int w = P.ops[k].aux_width ? P.ops[k].aux_width : 3;  // row width of the column (1..3), wave-uniform
                const T* a = static_cast<const T*>(P.ops[k].aux) + (size_t)(row0 + t) * w;
                aux[k] = Vec3<T>{a[0], w > 1 ? a[1] : T(0), w > 2 ? a[2] : T(0)};
            }
        }
    };
    load_aux(std::integral_constant<int, 0>{});
    load_aux(std::integral_constant<int, 1>{});
    load_aux(std::integral_constant<int, 2>{});
    load_aux(std::integral_constant<int, 3>{});

    typename PIPE::template Regs<T> regs;   // component columns of a generated program, one row per lane
    if constexpr (PIPE::kHasModel) PIPE::load(P, row0 + t, active, regs);

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // LDS-DMA data has landed
    __syncthreads();

    // Row state.  Inactive lanes of a ragged last wave carry a harmless identity body so the whole wave can
    // run the tick loop (and its wave-level barriers) uniformly.
    const Spatial<T> zero6 = {{T(0), T(0), T(0)}, {T(0), T(0), T(0)}};
    Quat<T> q0 = {T(0), T(0), T(0), T(1)};
    Vec3<T> p0 = {T(0), T(0), T(0)}, inv_I = {T(1), T(1), T(1)}, I_diag = {T(1), T(1), T(1)};
    Spatial<T> v0 = zero6, A_out = zero6, F_out = zero6;
    T mass = T(1), inv_m = T(1);
    if (active) {
        const T* r = l_pos + t * 7;
        q0 = Quat<T>{r[0], r[1], r[2], r[3]};
        p0 = Vec3<T>{r[4], r[5], r[6]};
        const T* s = l_vel + t * 6;
        v0.ang = Vec3<T>{s[0], s[1], s[2]};
        v0.lin = Vec3<T>{s[3], s[4], s[5]};
        const T* m = l_c + t * 7;
        I_diag = Vec3<T>{m[0], m[1], m[2]};
        inv_I = Vec3<T>{T(1) / m[0], T(1) / m[1], T(1) / m[2]};
        mass = m[6];
        inv_m = T(1) / mass;
        if constexpr (INTEGRATOR == kNone) {   // no six_dof in the pipe: world_accel / force pass through untouched
            const T* a = g_accel + (size_t)t * 6;
            const T* f = g_force + (size_t)t * 6;
            A_out = Spatial<T>{{a[0], a[1], a[2]}, {a[3], a[4], a[5]}};
            F_out = Spatial<T>{{f[0], f[1], f[2]}, {f[3], f[4], f[5]}};
        }
    }
    if (P.n_ticks == 0) return;
    __syncthreads();  // every lane has consumed the input slabs; LDS is the output staging area from here on

    auto stage_rows = [&]() {
        if (active) {
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
    };
    // rows in LDS -> the four output columns at `base` pointers (live columns or one history slot)
    auto flush_rows = [&](T* o_pos, T* o_vel, T* o_accel, T* o_force, auto nt) {
        constexpr int kNt = decltype(nt)::value;
        if (full) {
            slab_out<kWave * 7 * sizeof(T), kNt>(reinterpret_cast<const char*>(l_pos), reinterpret_cast<char*>(o_pos), t);
            slab_out<kWave * 6 * sizeof(T), kNt>(reinterpret_cast<const char*>(l_vel), reinterpret_cast<char*>(o_vel), t);
            slab_out<kWave * 6 * sizeof(T), kNt>(reinterpret_cast<const char*>(l_c), reinterpret_cast<char*>(o_accel), t);
            slab_out<kWave * 6 * sizeof(T), kNt>(reinterpret_cast<const char*>(l_force), reinterpret_cast<char*>(o_force), t);
        } else {
            slab_out_tail(l_pos, o_pos, rows * 7, t);
            slab_out_tail(l_vel, o_vel, rows * 6, t);
            slab_out_tail(l_c, o_accel, rows * 6, t);
            slab_out_tail(l_force, o_force, rows * 6, t);
        }
    };

    const bool record = P.hist_pos != nullptr;  // wave-uniform: stream every tick's outputs to the history ring
    const T dt_g = T(P.dt_g), dt = T(P.dt);
    Body<T> b;
    b.mass = mass;
    b.I = I_diag;
    Wrench<T> F = zero_wrench<T>();
    for (uint32_t tick = 0; tick < P.n_ticks; tick++) {
        if constexpr (PIPE::kHasModel) {   // user systems piped in front of six_dof (may rewrite inertia, pose, velocity)
            PIPE::pre(P, P.tick0 + tick + 1, regs, q0, p0, v0, I_diag, mass);
            if constexpr (PIPE::kWritesInertia) {
                inv_I = Vec3<T>{T(1) / I_diag.x, T(1) / I_diag.y, T(1) / I_diag.z};
                inv_m = T(1) / mass;
                b.mass = mass;
                b.I = I_diag;
            }
        }
        if constexpr (INTEGRATOR == kRk4) {
            const T h1 = dt_g * T(0.5), h3 = dt_g;
            Spatial<T> A, sv, sa;
            // stage 0 (c = 0): x0 (+) 0 only renormalises the quaternion
            b.q = normalized(q0);
            b.p = p0;
            b.v = v0;
            F = zero_wrench<T>();
            PIPE::apply(P, aux, regs, b, F);
            A = calc_accel<PIPE>(b.q, F, inv_I, inv_m);
            sv = v0;
            sa = A;
            // stage 1 (c = 1/2): position advanced with v0 (reference quirk), velocity with A0
            b.q = integrate_world(q0, h1 * v0.ang);
            b.p = axpy(h1, v0.lin, p0);
            b.v = axpy(h1, A, v0);
            sv = axpy(T(2), b.v, sv);
            F = zero_wrench<T>();
            PIPE::apply(P, aux, regs, b, F);
            A = calc_accel<PIPE>(b.q, F, inv_I, inv_m);
            sa = axpy(T(2), A, sa);
            // stage 2 (c = 1/2): same transform as stage 1
            b.v = axpy(h1, A, v0);
            sv = axpy(T(2), b.v, sv);
            if (!PIPE::vel_independent(P)) {
                F = zero_wrench<T>();
                PIPE::apply(P, aux, regs, b, F);
                A = calc_accel<PIPE>(b.q, F, inv_I, inv_m);
            }
            sa = axpy(T(2), A, sa);
            // stage 3 (c = 1)
            b.q = integrate_world(q0, h3 * v0.ang);
            b.p = axpy(h3, v0.lin, p0);
            b.v = axpy(h3, A, v0);
            sv = sv + b.v;
            F = zero_wrench<T>();
            PIPE::apply(P, aux, regs, b, F);
            A = calc_accel<PIPE>(b.q, F, inv_I, inv_m);
            sa = sa + A;
            // u' = u + (dt/6)(k1 + 2k2 + 2k3 + k4)
            const T g = dt * T(1.0 / 6.0);
            q0 = integrate_world(q0, g * sv.ang);
            p0 = axpy(g, sv.lin, p0);
            v0 = axpy(g, sa, v0);
            A_out = A;
        } else if constexpr (INTEGRATOR == kNone) {
            // systems only (`World.build(system)` without six_dof): the pre / post hooks are the whole tick
        } else {
            // semi-implicit: a = calc_accel(F(x0,v0)); v' = v0 + dt a; x' = x0 (+) dt v'
            b.q = normalized(q0);  // q * v is scale-invariant; user data may not be unit on tick 0
            b.p = p0;
            b.v = v0;
            F = zero_wrench<T>();
            PIPE::apply(P, aux, regs, b, F);
            const Spatial<T> A = calc_accel<PIPE>(b.q, F, inv_I, inv_m);
            v0 = axpy(dt, A, v0);
            q0 = integrate_world(q0, dt * v0.ang);
            p0 = axpy(dt, v0.lin, p0);
            A_out = A;
        }
        if constexpr (PIPE::kHasModel) PIPE::post(P, P.tick0 + tick + 1, regs, q0, p0, v0, I_diag, mass, A_out);   // A_out: this tick's world_accel
        if (record) {
            // telemetry: this tick's world_pos / world_vel / world_accel / force rows -> ring slot, in the
            // reference's row layout, write-once (non-temporal); the stores drain under the next tick's math
            if constexpr (INTEGRATOR != kNone) F_out = world_wrench<PIPE>(b.q, F);
            stage_rows();
            __syncthreads();
            const size_t slot = (size_t)((P.hist_slot0 + tick) % P.hist_ring);
            const size_t r7 = (slot * P.n + row0) * 7, r6 = (slot * P.n + row0) * 6;
            flush_rows(static_cast<T*>(P.hist_pos) + r7, static_cast<T*>(P.hist_vel) + r6,
                       static_cast<T*>(P.hist_accel) + r6, static_cast<T*>(P.hist_force) + r6, std::integral_constant<int, 1>{});
            if constexpr (PIPE::kHasModel)
                if (active) PIPE::record(P, slot, row0 + t, regs);   // component columns of a generated program
            __syncthreads();
        }
    }
    if constexpr (PIPE::kHasModel) {
        if (active) {
            PIPE::store(P, row0 + t, regs);
            if constexpr (PIPE::kWritesInertia) {   // a system returned el.Inertia: the column is an output
                T* gi = static_cast<T*>(const_cast<void*>(P.inertia)) + (size_t)(row0 + t) * 7;
                gi[0] = I_diag.x; gi[1] = I_diag.y; gi[2] = I_diag.z; gi[6] = mass;
            }
        }
    }
    if constexpr (INTEGRATOR != kNone) F_out = world_wrench<PIPE>(b.q, F);  // wrench of the last stage evaluated, world frame
    stage_rows();
    __syncthreads();
    flush_rows(g_pos, g_vel, g_accel, g_force, std::integral_constant<int, NT>{});
}

// ---- launch helpers (shared with generated translation units) ----------------------------------------------

template <class T, class PIPE, int NT>
inline void launch_i(const StepParams& p, int integrator, dim3 grid, hipStream_t s) {
    if (integrator == kRk4) hipLaunchKernelGGL((sixdof_step_kernel<T, kRk4, PIPE, NT>), grid, dim3(kWave), 0, s, p);
    else if (integrator == kNone) {
        if constexpr (PIPE::kHasModel) hipLaunchKernelGGL((sixdof_step_kernel<T, kNone, PIPE, NT>), grid, dim3(kWave), 0, s, p);
    } else hipLaunchKernelGGL((sixdof_step_kernel<T, kSemiImplicit, PIPE, NT>), grid, dim3(kWave), 0, s, p);
}

template <class T, class PIPE>
inline void launch_t(const StepParams& p, int integrator, dim3 grid, hipStream_t s) {
    if (p.streaming == 1) launch_i<T, PIPE, 1>(p, integrator, grid, s);
    else if (p.streaming == 2) launch_i<T, PIPE, 2>(p, integrator, grid, s);
    else launch_i<T, PIPE, 0>(p, integrator, grid, s);
}

template <class PIPE>
inline void launch_p(const StepParams& p, int integrator, int dtype, dim3 grid, hipStream_t s) {
    if (dtype == SIXDOF_F64) launch_t<double, PIPE>(p, integrator, grid, s);
    else launch_t<float, PIPE>(p, integrator, grid, s);
}

}  // namespace sixdof
