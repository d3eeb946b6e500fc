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
// (rk4.rs:96-100); the column is read only on the first tick after an upload, where it can hold non-finite host data
// (0 * NaN = NaN poisons that tick like the reference's); later ticks' a_in is already folded into v0.
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

// Cache policy of a launch, one template parameter POL = load policy * 8 + store policy:
//   loads  0 = default, 1 = non-temporal (LDS-DMA aux = 2: a byte that is touched once per tick is not worth a line)
//   stores 0 = plain (lines stay dirty in the XCD's L2 until the kernel-end write-back), 1 = non-temporal,
//          2 = write-through `sc1`, 3 = `sc0 sc1`, 4 = `sc1 nt`
// profiles/r02_step_ab_*.txt hold the A/B of these at 65,536 / 131,072 / 4.2M bodies.
constexpr int kPolPlain = 0, kPolNtStores = 1, kPolSc1Stores = 2, kPolNt = 9;
constexpr int pol_ld(int pol) { return pol / 8; }
constexpr int pol_st(int pol) { return pol % 8; }

// Whole-wave slab of BYTES bytes (multiple of 16), global -> LDS by LDS-DMA.
template <int BYTES, int POL>
__device__ __forceinline__ void slab_dma_in(const char* __restrict__ g, char* l, uint32_t lane) {
    constexpr int kFull = BYTES / 1024, kRem = (BYTES % 1024) / 16;
    constexpr int kAux = pol_ld(POL) == 1 ? 2 : 0;
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
template <int POL>
__device__ __forceinline__ void store16(char* g, vfloat4 v) {
    constexpr int ST = pol_st(POL);
    if constexpr (ST == 1) __builtin_nontemporal_store(v, reinterpret_cast<vfloat4*>(g));
    else if constexpr (ST == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(g), "v"(v) : "memory");
    else if constexpr (ST == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(g), "v"(v) : "memory");
    else if constexpr (ST == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(g), "v"(v) : "memory");
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

// One element of a generated program's component column, per lane, under the launch's cache policy: at one tick per launch
// every column byte is touched once per tick — non-temporal accesses keep a stream of 1.7 KB per rollout from evicting what
// little is re-read (A/B: profiles/r04_falcon9_k1_policy_ab.txt).
template <int POL, class T>
__device__ __forceinline__ T col_ld(const T* p) {
    if constexpr (pol_ld(POL) == 1) return __builtin_nontemporal_load(p);
    else return *p;
}
template <int POL, class T>
__device__ __forceinline__ void col_st(T* p, T v) {
    if constexpr (pol_st(POL) == 1) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// ---- the kernel --------------------------------------------------------------------------------------

// CHECK: the instantiation a launch with StepParams::accel_in_check uses (first RK4 launch after an upload, once): its
// first tick reads the incoming world_accel row for stage 0.  A kernel of its own so that the everyday kernels carry
// neither the branch nor a third copy of the tick body (measured: +0.25 us per one-tick launch at 65,536 bodies when it
// lived in the same kernel, profiles/r02_step_taint_check_ab.txt).  Generated programs (kHasModel) keep it in their one kernel.
// ROWS: entities per wave.  64 = one per lane.  32 = half-filled waves, twice as many of them (lanes 32..63 idle): at sizes
// where a full-width launch is one wave per SIMD the two waves a SIMD then holds overlap each other's load -> math -> store
// chain (profiles/r03_step_half_waves_ab.txt).
template <class T, int INTEGRATOR, class PIPE, int POL, bool CHECK = false, int ROWS = kWave>
__global__ __launch_bounds__(kWave) void sixdof_step_kernel(const StepParams P) {
    // pos | vel | inertia on the way in (20 elems/entity); pos | vel | accel | force on the way out (25)
    __shared__ __attribute__((aligned(16))) T lds[ROWS * 25];
    T* const l_pos = lds;
    T* const l_vel = lds + ROWS * 7;
    T* const l_c = lds + ROWS * 13;  // inertia (in) / accel (out)
    T* const l_force = lds + ROWS * 19;

    const uint32_t row0 = blockIdx.x * ROWS;
    const uint32_t rows = min((uint32_t)ROWS, P.n - row0);
    const uint32_t t = threadIdx.x;
    const bool full = rows == ROWS;  // wave-uniform

    T* const g_pos = static_cast<T*>(P.pos) + (size_t)row0 * 7;
    T* const g_vel = static_cast<T*>(P.vel) + (size_t)row0 * 6;
    T* const g_accel = static_cast<T*>(P.accel) + (size_t)row0 * 6;
    T* const g_force = static_cast<T*>(P.force) + (size_t)row0 * 6;
    const T* const g_inertia = static_cast<const T*>(P.inertia) + (size_t)row0 * 7;

    // systems only and none of them touches a Body column (NoModel::kBodyDead): the Body slabs stay where they are
    constexpr bool kDead = INTEGRATOR == kNone && PIPE::kBodyDead;
    if constexpr (!kDead) {
        if (full) {
            slab_dma_in<ROWS * 7 * sizeof(T), POL>(reinterpret_cast<const char*>(g_pos), reinterpret_cast<char*>(l_pos), t);
            slab_dma_in<ROWS * 6 * sizeof(T), POL>(reinterpret_cast<const char*>(g_vel), reinterpret_cast<char*>(l_vel), t);
            slab_dma_in<ROWS * 7 * sizeof(T), POL>(reinterpret_cast<const char*>(g_inertia), reinterpret_cast<char*>(l_c), t);
        } else {
            slab_in_tail(g_pos, l_pos, rows * 7, t);
            slab_in_tail(g_vel, l_vel, rows * 6, t);
            slab_in_tail(g_inertia, l_c, rows * 7, t);
        }
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
    // each lane reads / writes its own rows straight from global memory: for programs with dozens of narrow columns that
    // beats staging their slabs through LDS (profiles/r02_generated_io_ab.txt: Falcon 9 at one tick per launch, 1M
    // rollouts: 287 us per-lane vs 360 us with LDS-DMA slabs, double-buffered) — everything is in flight at once
    if constexpr (PIPE::kHasModel) PIPE::template load<T, POL>(P, row0 + t, active, regs);

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // LDS-DMA data has landed
    __syncthreads();

    // Row state.  Inactive lanes of a ragged last wave carry a harmless identity body so the whole wave can
    // run the tick loop (and its wave-level barriers) uniformly.
    const Spatial<T> zero6 = {{T(0), T(0), T(0)}, {T(0), T(0), T(0)}};
    Quat<T> q0 = {T(0), T(0), T(0), T(1)};
    Vec3<T> p0 = {T(0), T(0), T(0)}, inv_I = {T(1), T(1), T(1)}, I_diag = {T(1), T(1), T(1)};
    Spatial<T> v0 = zero6, A_out = zero6, F_out = zero6;
    T mass = T(1), inv_m = T(1);
    if (active && !kDead) {
        const T* r = l_pos + t * 7;
        q0 = Quat<T>{r[0], r[1], r[2], r[3]};
        p0 = Vec3<T>{r[4], r[5], r[6]};
        const T* s = l_vel + t * 6;
        v0.ang = Vec3<T>{s[0], s[1], s[2]};
        v0.lin = Vec3<T>{s[3], s[4], s[5]};
        const T* m = l_c + t * 7;
        I_diag = Vec3<T>{m[0], m[1], m[2]};
        inv_I = Vec3<T>{recip(m[0]), recip(m[1]), recip(m[2])};
        mass = m[6];
        inv_m = recip(mass);
        if constexpr (INTEGRATOR == kNone || PIPE::kPreReadsAccel) {
            // no six_dof in the pipe: world_accel / force pass through untouched.  With six_dof: a system piped in front of
            // it that reads world_accel sees what the previous tick left there (the reference's column semantics) — later
            // ticks of this launch find it in A_out.
            const T* a = g_accel + (size_t)t * 6;
            A_out = Spatial<T>{{a[0], a[1], a[2]}, {a[3], a[4], a[5]}};
        }
        if constexpr (INTEGRATOR == kNone) {
            const T* f = g_force + (size_t)t * 6;
            F_out = Spatial<T>{{f[0], f[1], f[2]}, {f[3], f[4], f[5]}};
        }
    }
    if (P.n_ticks == 0) return;
    __syncthreads();  // every lane has consumed the input slabs; LDS is the output staging area from here on

    // One output column at a time: rows -> LDS staging area -> 16-B-per-lane coalesced stores.  Each column has its
    // own LDS region, so a column can leave as soon as its rows exist (see `early` below).
    auto stage_pos = [&](const Quat<T>& q, const Vec3<T>& p) {
        if (active) {
            T* r = l_pos + t * 7;
            r[0] = q.i; r[1] = q.j; r[2] = q.k; r[3] = q.w; r[4] = p.x; r[5] = p.y; r[6] = p.z;
        }
    };
    auto stage6 = [&](T* l, const Spatial<T>& m) {
        if (active) {
            T* s = l + t * 6;
            s[0] = m.ang.x; s[1] = m.ang.y; s[2] = m.ang.z; s[3] = m.lin.x; s[4] = m.lin.y; s[5] = m.lin.z;
        }
    };
    auto flush7 = [&](const T* l, T* o, auto pol) {
        if (full) slab_out<ROWS * 7 * sizeof(T), decltype(pol)::value>(reinterpret_cast<const char*>(l), reinterpret_cast<char*>(o), t);
        else slab_out_tail(l, o, rows * 7, t);
    };
    auto flush6 = [&](const T* l, T* o, auto pol) {
        if (full) slab_out<ROWS * 6 * sizeof(T), decltype(pol)::value>(reinterpret_cast<const char*>(l), reinterpret_cast<char*>(o), t);
        else slab_out_tail(l, o, rows * 6, t);
    };
    constexpr std::integral_constant<int, POL> kLive{};   // cache policy of the live columns
    constexpr std::integral_constant<int, 1> kRing{};     // history ring slots are write-once: non-temporal stores

    const bool record = P.hist_pos != nullptr && P.hist_ring != 0;  // wave-uniform: stream every tick's outputs to the history ring
    // At one tick per launch the launch is a latency chain (dispatch -> loads -> math -> stores -> write-back), so on
    // the LAST tick of a launch every output column is staged and stored the moment its rows exist instead of all four
    // after the tick: world_pos is known before the last stage's force evaluation (the stage positions never see a
    // stage velocity, rk4.rs:110-121, and sum(v_s) needs only A of stage 2), force before calc_accel, world_accel
    // before the velocity update.  Same arithmetic in the same order -> same bits as the late flush.
    // Not with a generated program's post hook (it may still rewrite the pose) nor while recording.
    // (StepParams::streaming bit 8 turns it off: the A/B knob of tools/step_ab.py.)
    const bool early_ok = !PIPE::kHasModel && !record && !(P.streaming & 256u);
    const T dt_g = T(P.dt_g), dt = T(P.dt);
    Body<T> b;
    b.mass = mass;
    b.I = I_diag;
    Wrench<T> F = zero_wrench<T>();
    bool flushed = false;   // wave-uniform: the live columns already left on the early path
    // One tick.  `early` is a compile-time flag so that the per-tick body of a fused launch (n_ticks > 1) carries none of
    // the early-store code: only the last tick of a launch is instantiated with it.  The copies compute the same bits
    // because contraction is per source expression (kernels.hpp: #pragma clang fp contract(on)).
    // `check` (first tick of the first launch after an upload only, StepParams::accel_in_check) reads the incoming
    // world_accel row for stage 0; also compile-time, so no other tick carries the branch or the load.
#ifdef SIXDOF_TICK_OUT_OF_LINE
    // last-resort build of a generated program (codegen.py): the tick body as a real function.  Its captures (the whole
    // register image) are then reached through the closure in the lane's private memory and the body gets a register budget
    // of its own — what LLVM did by itself in round 2 whenever a body was instantiated more than once and too big to inline.
    auto one_tick = [&](auto early_tag, auto check_tag, uint32_t tick) __attribute__((noinline)) {
#else
    auto one_tick = [&](auto early_tag, auto check_tag, uint32_t tick) {
#endif
        constexpr bool early = decltype(early_tag)::value;
        // hand-written pipes: compile-time.  Generated programs: decided at run time (a wave-uniform branch around one
        // load), so that their kernel has exactly ONE call site of the tick body — a second copy of a 40,000-instruction
        // body doubles the code the instruction cache has to stream, and once the body is too big to inline twice the
        // compiler turns it into a real function whose call spills the whole register state to scratch (4 KB per lane
        // per tick on the Falcon 9 program).
        const bool check = decltype(check_tag)::value || (PIPE::kHasModel && P.accel_in_check && tick == 0);
        if constexpr (PIPE::kHasModel) {   // user systems piped in front of six_dof (may rewrite inertia, pose, velocity)
            PIPE::pre(P, P.tick0 + tick + 1, regs, q0, p0, v0, I_diag, mass, A_out);
            if constexpr (PIPE::kWritesInertia) {
                inv_I = Vec3<T>{recip(I_diag.x), recip(I_diag.y), recip(I_diag.z)};
                inv_m = recip(mass);
                b.mass = mass;
                b.I = I_diag;
            }
        }
        if constexpr (INTEGRATOR == kRk4) {
            const T h1 = dt_g * T(0.5), h3 = dt_g;
            Spatial<T> A, sv, sa;
            // stage 0 (c = 0): x0 (+) 0 only renormalises the quaternion; v_s = v0 + 0 * a_in.  a_in is the world_accel
            // COLUMN: from the second tick on it is this kernel's own A of the previous tick, which already sits inside
            // v0 (a non-finite A made v0 non-finite), so 0 * a_in adds nothing; only host data handed over by an upload
            // can be non-finite on its own — then the reference's tick turns NaN through exactly this product
            // (rk4.rs:96-100), and so does this one: the column is read once, on the first tick after an upload.
            b.q = normalized(q0);
            b.p = p0;
            b.v = v0;
            if (check && active) {
                const T* a = g_accel + (size_t)t * 6;
                b.v.ang = b.v.ang + T(0) * Vec3<T>{a[0], a[1], a[2]};
                b.v.lin = b.v.lin + T(0) * Vec3<T>{a[3], a[4], a[5]};
            }
            F = zero_wrench<T>();
            PIPE::apply(P, aux, regs, b, F);
            A = calc_accel<PIPE>(b.q, F, inv_I, inv_m, T(0));
            sv = b.v;   // = v0 (+ 0 * a_in on the first tick after an upload)
            sa = A;
            // stage 1 (c = 1/2): position advanced with v0 (reference quirk), velocity with A0
            b.q = integrate_world(q0, h1 * v0.ang);
            b.p = axpy(h1, v0.lin, p0);
            b.v = axpy(h1, A, v0);
            sv = axpy(T(2), b.v, sv);
            F = zero_wrench<T>();
            PIPE::apply(P, aux, regs, b, F);
            A = calc_accel<PIPE>(b.q, F, inv_I, inv_m, T(0));
            sa = axpy(T(2), A, sa);
            // stage 2 (c = 1/2): same transform as stage 1
            b.v = axpy(h1, A, v0);
            sv = axpy(T(2), b.v, sv);
            if (!PIPE::vel_independent(P)) {
                F = zero_wrench<T>();
                PIPE::apply(P, aux, regs, b, F);
                A = calc_accel<PIPE>(b.q, F, inv_I, inv_m, T(0));
            }
            sa = axpy(T(2), A, sa);
            // stage 3 (c = 1)
            T n3;
            b.q = integrate_world(q0, h3 * v0.ang, &n3);
            const T taint = accel_taint(n3);   // NaN when q0 or v0.ang is not finite (effectors.hpp), +-0 otherwise
            b.p = axpy(h3, v0.lin, p0);
            b.v = axpy(h3, A, v0);
            sv = sv + b.v;
            // x' = x0 (+) (dt/6) sum(v_s): complete here, before the last force evaluation
            const T g = dt * T(1.0 / 6.0);
            const Quat<T> q_new = integrate_world(q0, g * sv.ang);
            const Vec3<T> p_new = axpy(g + taint, sv.lin, p0);
            if constexpr (early) {
                stage_pos(q_new, p_new);
                __syncthreads();
                flush7(l_pos, g_pos, kLive);
            }
            F = zero_wrench<T>();
            PIPE::apply(P, aux, regs, b, F);
            if constexpr (early) {
                stage6(l_force, world_wrench<PIPE>(b.q, F));
                __syncthreads();
                flush6(l_force, g_force, kLive);
            }
            A = calc_accel<PIPE>(b.q, F, inv_I, inv_m + taint, taint);
            if constexpr (early) {
                stage6(l_c, A);
                __syncthreads();
                flush6(l_c, g_accel, kLive);
            }
            sa = sa + A;
            // v' = v0 + (dt/6)(k1 + 2k2 + 2k3 + k4)
            v0 = axpy(g, sa, v0);
            q0 = q_new;
            p0 = p_new;
            A_out = A;
            if constexpr (early) {
                stage6(l_vel, v0);
                __syncthreads();
                flush6(l_vel, g_vel, kLive);
                flushed = true;
            }
        } else if constexpr (INTEGRATOR == kNone) {
            // systems only (`World.build(system)` without six_dof): the pre / post hooks are the whole tick
        } else {
            // semi-implicit: a = calc_accel(F(x0,v0)); v' = v0 + dt a; x' = x0 (+) dt v'
            T n0;
            b.q = normalized(q0, &n0);  // q * v is scale-invariant; user data may not be unit on tick 0
            const T taint = accel_taint(n0);
            b.p = p0;
            b.v = v0;
            F = zero_wrench<T>();
            PIPE::apply(P, aux, regs, b, F);
            if constexpr (early) {
                stage6(l_force, world_wrench<PIPE>(b.q, F));
                __syncthreads();
                flush6(l_force, g_force, kLive);
            }
            const Spatial<T> A = calc_accel<PIPE>(b.q, F, inv_I, inv_m + taint, taint);
            if constexpr (early) {
                stage6(l_c, A);
                __syncthreads();
                flush6(l_c, g_accel, kLive);
            }
            v0 = axpy(dt, A, v0);
            if constexpr (early) {
                stage6(l_vel, v0);
                __syncthreads();
                flush6(l_vel, g_vel, kLive);
            }
            q0 = integrate_world(q0, dt * v0.ang);
            p0 = axpy(dt, v0.lin, p0);
            A_out = A;
            if constexpr (early) {
                stage_pos(q0, p0);
                __syncthreads();
                flush7(l_pos, g_pos, kLive);
                flushed = true;
            }
        }
        if constexpr (PIPE::kHasModel) PIPE::post(P, P.tick0 + tick + 1, regs, q0, p0, v0, I_diag, mass, A_out);   // A_out: this tick's world_accel
        if (record) {
            // telemetry: this tick's world_pos / world_vel / world_accel / force rows -> ring slot, in the
            // reference's row layout, write-once (non-temporal); the stores drain under the next tick's math
            if constexpr (INTEGRATOR != kNone) F_out = world_wrench<PIPE>(b.q, F);
            const size_t slot = (size_t)((P.hist_slot0 + tick) % P.hist_ring);
            if constexpr (!kDead) {      // (a program without Body state records its component columns only)
                stage_pos(q0, p0);
                stage6(l_vel, v0);
                stage6(l_c, A_out);
                stage6(l_force, F_out);
                __syncthreads();
                const size_t r7 = (slot * P.n + row0) * 7, r6 = (slot * P.n + row0) * 6;
                flush7(l_pos, static_cast<T*>(P.hist_pos) + r7, kRing);
                flush6(l_vel, static_cast<T*>(P.hist_vel) + r6, kRing);
                flush6(l_c, static_cast<T*>(P.hist_accel) + r6, kRing);
                flush6(l_force, static_cast<T*>(P.hist_force) + r6, kRing);
            }
            if constexpr (PIPE::kHasModel)
                if (active) PIPE::record(P, slot, row0 + t, regs);   // component columns of a generated program
            __syncthreads();
        }
    };
    {
        constexpr std::false_type no{};
        constexpr std::true_type yes{};
        uint32_t tick = 0;
        if constexpr (PIPE::kHasModel) {
            for (; tick < P.n_ticks; tick++) one_tick(no, no, tick);        // the one call site (see `check` above)
        } else {
            if constexpr (INTEGRATOR == kRk4 && CHECK)
                if (P.accel_in_check && P.n_ticks) one_tick(no, yes, tick++);   // once per upload: late flush, speed is no concern
            for (; tick + 1 < P.n_ticks; tick++) one_tick(no, no, tick);
            if (tick < P.n_ticks) {
                if (early_ok) one_tick(yes, no, tick);
                else one_tick(no, no, tick);
            }
        }
    }
    if constexpr (PIPE::kHasModel) {
        if (active) {
            PIPE::template store<T, POL>(P, row0 + t, regs);
            if constexpr (PIPE::kWritesInertia) {   // a system returned el.Inertia: the column is an output
                T* gi = static_cast<T*>(const_cast<void*>(P.inertia)) + (size_t)(row0 + t) * 7;
                gi[0] = I_diag.x; gi[1] = I_diag.y; gi[2] = I_diag.z; gi[6] = mass;
            }
        }
    }
    if (flushed || kDead) return;
    if constexpr (INTEGRATOR != kNone) F_out = world_wrench<PIPE>(b.q, F);  // wrench of the last stage evaluated, world frame
    stage_pos(q0, p0);
    stage6(l_vel, v0);
    stage6(l_c, A_out);
    stage6(l_force, F_out);
    __syncthreads();
    flush7(l_pos, g_pos, kLive);
    flush6(l_vel, g_vel, kLive);
    flush6(l_c, g_accel, kLive);
    flush6(l_force, g_force, kLive);
}

// ---- launch helpers (shared with generated translation units) ----------------------------------------------

template <class T, class PIPE, int POL>
inline void launch_i(const StepParams& p, int integrator, dim3 grid, hipStream_t s) {
#ifdef SIXDOF_AB_BUILD
    if constexpr (!PIPE::kHasModel) {
        if ((p.streaming & 512u) && integrator == kRk4 && !p.accel_in_check) {      // A/B: half-filled waves (ROWS = 32)
            hipLaunchKernelGGL((sixdof_step_kernel<T, kRk4, PIPE, POL, false, 32>), dim3((p.n + 31) / 32), dim3(kWave), 0, s, p);
            return;
        }
    }
#endif
    if (integrator == kRk4) {
        if constexpr (!PIPE::kHasModel) {
            if (p.accel_in_check) {   // one launch per upload: a single cache policy is plenty
                hipLaunchKernelGGL((sixdof_step_kernel<T, kRk4, PIPE, kPolPlain, true>), grid, dim3(kWave), 0, s, p);
                return;
            }
        }
        hipLaunchKernelGGL((sixdof_step_kernel<T, kRk4, PIPE, POL>), grid, dim3(kWave), 0, s, p);
    } else if (integrator == kNone) {
        if constexpr (PIPE::kHasModel) hipLaunchKernelGGL((sixdof_step_kernel<T, kNone, PIPE, POL>), grid, dim3(kWave), 0, s, p);
    } else hipLaunchKernelGGL((sixdof_step_kernel<T, kSemiImplicit, PIPE, POL>), grid, dim3(kWave), 0, s, p);
}

// StepParams::streaming is the cache-policy code (load * 8 + store).  Every pipe has the three shipped policies
// (plain, nt stores, nt both ways) and the product library has nothing else; an A/B build (-DSIXDOF_AB_BUILD) adds the
// rest of the matrix (SWEEP, tools/step_ab.py) on the pipes that ask for it.
template <class T, class PIPE, bool SWEEP>
inline void launch_t(const StepParams& p, int integrator, dim3 grid, hipStream_t s) {
    const uint32_t pol = p.streaming & 255u;   // bit 8 = late flush (A/B knob), see the kernel
    switch (pol) {
    case kPolNt: return launch_i<T, PIPE, kPolNt>(p, integrator, grid, s);
    case kPolNtStores: return launch_i<T, PIPE, kPolNtStores>(p, integrator, grid, s);
    default: break;
    }
#ifdef SIXDOF_AB_BUILD
    // A/B library only (make ab -> libsixdof_hip_ab.so, never shipped).  kPolSc1Stores is UNSAFE across launches
    // (the next launch reads stale rows, profiles/r02_sc1_store_policy_is_unsafe.txt): it exists to demonstrate that.
    if (pol == kPolSc1Stores) return launch_i<T, PIPE, kPolSc1Stores>(p, integrator, grid, s);
    if constexpr (SWEEP) {
        switch (pol) {
        case 3: return launch_i<T, PIPE, 3>(p, integrator, grid, s);
        case 4: return launch_i<T, PIPE, 4>(p, integrator, grid, s);
        case 8: return launch_i<T, PIPE, 8>(p, integrator, grid, s);
        case 10: return launch_i<T, PIPE, 10>(p, integrator, grid, s);
        case 11: return launch_i<T, PIPE, 11>(p, integrator, grid, s);
        case 12: return launch_i<T, PIPE, 12>(p, integrator, grid, s);
        default: break;
        }
    }
#endif
    launch_i<T, PIPE, kPolPlain>(p, integrator, grid, s);
}

template <class PIPE, bool SWEEP = false>
inline void launch_p(const StepParams& p, int integrator, int dtype, dim3 grid, hipStream_t s) {
    if (dtype == SIXDOF_F64) launch_t<double, PIPE, SWEEP>(p, integrator, grid, s);
    else launch_t<float, PIPE, false>(p, integrator, grid, s);
}

}  // namespace sixdof
