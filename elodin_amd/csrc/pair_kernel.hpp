// pair_kernel.hpp — pairwise (GraphQuery.edge_fold) path for gfx950, as templates over a PAIR functor.
//
// Included by nbody_kernels.hip (the built-in gravity folds) and by generated translation units
// (elodin_amd/codegen.py: user-written fold functions), so both run the same kernels.
//
// Reference semantics (libs/nox-py/src/graph.rs:239-361, python twin elodin/__init__.py:454-557): for
// every source entity, a sequential left fold over its out-edges in spawn order,
//     acc = f(acc, source components, target components),
// whose result REPLACES `Force` on the source rows; it runs inside the six_dof pipe, i.e. on every
// RK4 stage.  The reference materialises gathered operands [S, D, dim] — O(N^2) memory for a
// complete graph — which is why its n-body example stops at 35 bodies.
//
// What this file does instead.  The reference's RK4 advances stage POSITIONS with the initial
// velocity only (rk4.rs:95-121: x_s = x0 (+) c*dt*v0), and gravity reads positions and masses only,
// so all stage forces of a tick are known from (x0, v0) before any acceleration is: F(c=0),
// F(c=1/2) (shared by stages 1 and 2, bit-identical) and F(c=1).  One tick is therefore
//   1. pair_pack_kernel      per source: p(c=0), p(c=1/2), p(c=1), mass -> pack[n,10]   (FIRST tick of a batch only)
//   2. allpairs_kernel       LDS-tiled all-pairs: every (target, source) pair is visited ONCE and
//                            accumulates the three stage forces together (3 independent FMA chains)
//   3. pair_integrate_kernel per entity: reduce the source splits in fixed order, calc_accel on each
//                            stage, RK4 combination, write pos / vel / accel / force — and the NEXT tick's pack row
//                            from the state it has just formed, so a batch of ticks is pack + 2 launches per tick;
//   edge lists (CSR):        2 + 3 in ONE launch per tick (pair_tick_fused_kernel, 3b: the lane folds its out-edges while its
//                            state slabs land; pack rows double-buffered), hub sources folded by whole waves in front of it (2c).
// instead of four dependent all-pairs sweeps.  Kernels 1 and 3 use the step kernel's memory plan (step_kernel.hpp):
// single-wave workgroups own 64 consecutive rows, whole slabs move HBM <-> LDS 16 B per lane (LDS-DMA on the way in), a lane
// reads / writes its own row in LDS — no 56- / 48-byte-strided global access is left on the path.  Bound: f64 vector ALU (about 21 instructions per pair
// evaluation, of which one v_rsq_f64 + refinement); bytes are negligible.  MFMA is not used: gfx950's
// f64 MFMA rate equals its f64 vector rate, and the |ri|^2+|rj|^2-2 ri.rj form a dense tile would
// need cancels catastrophically for close pairs, breaking the 1e-9 parity bar.
//
// Summation order: the reference folds each source's targets sequentially; here a target range is
// split over `splits` workgroup columns (partials reduced in fixed split order) — a different
// association of the same sum (~1e-16 * sqrt(N) relative), deterministic run to run.
#pragma once
#include "effectors.hpp"
#include "kernels.hpp"
#include "spatial.hpp"
#include "step_kernel.hpp"      // slab_dma_in / slab_out / the ragged-tail movers

namespace sixdof {

constexpr int kTile = 256;  // sources staged per LDS tile = targets per workgroup
constexpr uint32_t kSmallEdgeCache = 2048;  // edges of a small graph (n <= 256) cached in LDS by the one-launch kernel

// ---- PAIR functors: one directed edge a -> b folded into a's Force accumulator [tau(3), f(3)] ----------------
// fold(acc, pa, ma, pb, mb, p0, p1): pa / pb = the two bodies' positions at ONE stage, ma / mb their masses.
// kAdditive: fold(acc, ...) = acc + g(a, b) component by component (or the constant 0), so partial
// accumulators over disjoint edge subsets may be summed — what lets a hub's out-edges be folded by many lanes (2c).
// Both built-in folds form 1/|r|^3 from ONE v_rsq_f64 + cubic correction (spatial.hpp rsqrt_pos: full f64 accuracy in 5
// instructions) where the reference's text divides three components by norm^3 (three IEEE divides and a sqrt, ~80 dependent f64
// instructions per stage and edge): with one wave per SIMD a sparse fold is a chain of dependent arithmetic, not of gathers
// (profiles/r06_pair_kernels.md: batching the gathers changed nothing, this did).  Algebraically the same value, O(1e-16) per edge
// against the divide-exact oracle — inside the 1e-9 bar like every other reciprocal of this backend.
struct PairNewton {   // examples/three-body/main.py:61-70: r = a - b; f = G*M*m*r / |r|^3; Force(linear = acc.f - f)
    static constexpr bool kAdditive = true;
    static constexpr int kEdgeBatch = 4;      // targets fetched per round trip (edge_accumulate_range)
    __device__ static __forceinline__ void fold(double (&acc)[6], const double* pa, double ma, const double* pb,
                                                double mb, double p0, double) {
        const double rx = pa[0] - pb[0], ry = pa[1] - pb[1], rz = pa[2] - pb[2];
        const double inv = rsqrt_pos(rx * rx + ry * ry + rz * rz);   // coincident bodies: rsqrt(0) = inf, inf * 0 = NaN like the reference's 0 / 0
        const double sc = (p0 * mb * ma) * (inv * inv * inv);
        acc[0] = 0.0; acc[1] = 0.0; acc[2] = 0.0;   // el.Force(linear=...) carries zero torque
        acc[3] -= sc * rx;
        acc[4] -= sc * ry;
        acc[5] -= sc * rz;
    }
};
struct PairSoftened {   // examples/n-body/sim.py:356-361: acc + SpatialForce(linear = K ma mb inv^3 r), r = b - a
    static constexpr bool kAdditive = true;
    static constexpr int kEdgeBatch = 4;
    __device__ static __forceinline__ void fold(double (&acc)[6], const double* pa, double ma, const double* pb,
                                                double mb, double p0, double p1) {
        const double rx = pb[0] - pa[0], ry = pb[1] - pa[1], rz = pb[2] - pa[2];
        const double inv = rsqrt_pos((rx * rx + ry * ry + rz * rz) + p1);
        const double sc = p0 * ma * mb * (inv * inv * inv);
        acc[3] += sc * rx;
        acc[4] += sc * ry;
        acc[5] += sc * rz;
    }
};

// ---- 1. pack ------------------------------------------------------------------------------------------
// One pack row from a body's position, linear velocity and mass — the one expression every path shares (this kernel, the
// integrate kernel's next-tick rows, the one-launch small-graph kernel), so they agree bit for bit.
__device__ __forceinline__ void pack_row(double* o, const double (&x)[3], const double (&v)[3], double mass, double h1, double h3) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
        o[c] = x[c];
        o[3 + c] = x[c] + h1 * v[c];  // linear half of SpatialTransform + SpatialMotion (spatial.rs:546)
        o[6 + c] = x[c] + h3 * v[c];
    }
    o[9] = mass;
}

// Single-wave workgroups, 64 rows each: pos / vel slabs by LDS-DMA, the pack rows leave as one 5,120-byte slab.  (The mass is
// one double of a 56-byte inertia row: read per lane — a slab would move seven times the bytes for it.)
__global__ __launch_bounds__(kWave) void pair_pack_kernel(const double* __restrict__ pos, const double* __restrict__ vel,
                                                          const double* __restrict__ inertia, double* __restrict__ pack,
                                                          uint32_t n, double h1, double h3) {
    __shared__ __attribute__((aligned(16))) double lds[kWave * (7 + 6)];
    double* const l_pos = lds;
    double* const l_vel = lds + kWave * 7;
    const uint32_t row0 = blockIdx.x * kWave, t = threadIdx.x;
    const uint32_t rows = min((uint32_t)kWave, n - row0);
    const bool full = rows == kWave, active = t < rows;
    if (full) {
        slab_dma_in<kWave * 7 * 8, kPolPlain>(reinterpret_cast<const char*>(pos + (size_t)row0 * 7), reinterpret_cast<char*>(l_pos), t);
        slab_dma_in<kWave * 6 * 8, kPolPlain>(reinterpret_cast<const char*>(vel + (size_t)row0 * 6), reinterpret_cast<char*>(l_vel), t);
    } else {
        slab_in_tail(pos + (size_t)row0 * 7, l_pos, rows * 7, t);
        slab_in_tail(vel + (size_t)row0 * 6, l_vel, rows * 6, t);
    }
    const double mass = active ? inertia[(size_t)(row0 + t) * 7 + 6] : 0.0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    double x[3] = {0, 0, 0}, v[3] = {0, 0, 0};
    if (active) {
#pragma unroll
        for (int c = 0; c < 3; c++) { x[c] = l_pos[t * 7 + 4 + c]; v[c] = l_vel[t * 6 + 3 + c]; }
    }
    __syncthreads();      // the input slabs are consumed: the same LDS stages the pack rows (10 <= 13 doubles per row)
    if (active) pack_row(lds + t * kPackWidth, x, v, mass, h1, h3);
    __syncthreads();
    if (full) slab_out<kWave * kPackWidth * 8, kPolPlain>(reinterpret_cast<const char*>(lds), reinterpret_cast<char*>(pack + (size_t)row0 * kPackWidth), t);
    else slab_out_tail(lds, pack + (size_t)row0 * kPackWidth, rows * kPackWidth, t);
}

// ---- 2a. all-pairs, softened (examples/n-body/sim.py:344-369) -------------------------------------------
// acc_s += mb * inv3 * r with r = pb - pa, inv = rsqrt(r.r + eps); the common factor K*ma is applied
// once per target at the end (the reference applies it per pair: ((K*ma)*mb)*inv3).
template <int NS, bool CHECK_SELF>
__device__ __forceinline__ void tile_accumulate(const double* __restrict__ tile, int count, uint32_t j0, uint32_t i,
                                                const double (&pi)[3][3], double eps, double (&acc)[3][3]) {
#pragma unroll 2
    for (int jj = 0; jj < count; jj++) {
        const double* s = tile + jj * kPackWidth;  // same address in every lane: LDS broadcast
        const double mj = s[9];
#pragma unroll
        for (int st = 0; st < NS; st++) {
            const double rx = s[3 * st + 0] - pi[st][0];
            const double ry = s[3 * st + 1] - pi[st][1];
            const double rz = s[3 * st + 2] - pi[st][2];
            const double d2 = fma(rz, rz, fma(ry, ry, fma(rx, rx, eps)));
            const double inv = rsqrt_pos(d2);
            double sc = mj * (inv * inv * inv);
            if (CHECK_SELF) sc = (j0 + jj == i) ? 0.0 : sc;  // i == j is not an edge (sim.py:333-337)
            acc[st][0] = fma(sc, rx, acc[st][0]);
            acc[st][1] = fma(sc, ry, acc[st][1]);
            acc[st][2] = fma(sc, rz, acc[st][2]);
        }
    }
}

template <int NS>
__global__ __launch_bounds__(kTile) void allpairs_kernel(const double* __restrict__ pack, double* __restrict__ partial,
                                                         uint32_t n, uint32_t splits, double K, double eps) {
    __shared__ __attribute__((aligned(16))) double tile[kTile * kPackWidth];
    const uint32_t i = blockIdx.x * kTile + threadIdx.x;
    const uint32_t split = blockIdx.y;
    const uint32_t tiles_total = (n + kTile - 1) / kTile;
    const uint32_t tiles_per_split = (tiles_total + splits - 1) / splits;
    const uint32_t tile_lo = split * tiles_per_split;
    const uint32_t tile_hi = min(tiles_total, tile_lo + tiles_per_split);

    double pi[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    double mi = 0.0;
    if (i < n) {
        const double* s = pack + (size_t)i * kPackWidth;
#pragma unroll
        for (int st = 0; st < NS; st++)
            for (int c = 0; c < 3; c++) pi[st][c] = s[3 * st + c];
        mi = s[9];
    }
    double acc[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (uint32_t tl = tile_lo; tl < tile_hi; tl++) {
        const uint32_t j0 = tl * kTile;
        const int count = (int)min((uint32_t)kTile, n - j0);
        __syncthreads();
        {   // contiguous slab of count*10 doubles, 16 B per lane per iteration
            const double2* g = reinterpret_cast<const double2*>(pack + (size_t)j0 * kPackWidth);
            double2* l = reinterpret_cast<double2*>(tile);
            for (int c = threadIdx.x; c < count * (kPackWidth / 2); c += kTile) l[c] = g[c];
        }
        __syncthreads();
        if (tl == blockIdx.x) tile_accumulate<NS, true>(tile, count, j0, i, pi, eps, acc);
        else tile_accumulate<NS, false>(tile, count, j0, i, pi, eps, acc);
    }
    if (i < n) {
        double* o = partial + ((size_t)split * n + i) * kPartialForce;
        const double kmi = K * mi;
#pragma unroll
        for (int st = 0; st < NS; st++)
            for (int c = 0; c < 3; c++) o[3 * st + c] = kmi * acc[st][c];
    }
}

// ---- 2b. explicit edge list, CSR by source (three-body and sparse graphs) ---------------------------------
// One source's left fold over its out-edges (CSR range), spawn order, for the NS stage positions.
// A lane's edges are independent GATHERS feeding one dependent fold.  PAIR::kEdgeBatch > 1 fetches the targets' pack rows that
// many edges at a time — all their loads in flight together — and folds them in order afterwards: same operations in the same
// order, the bits do not change.  Measured (profiles/r06_pair_kernels.md): nothing for the stand-alone fold kernel of the
// two-kernel tick (20.0 vs 19.5 us), 17.0 -> 13.0 us for the fused fold-and-integrate launch, where the fold sits on the wave's
// critical path beside its slab loads.  The built-in folds take 4; a GENERATED fold function keeps the plain loop (kEdgeBatch = 1):
// the row buffers cost registers a large function does not have (a fuzz-generated fold spilled 8 VGPRs with them and was refused).
template <int NS, class PAIR>
__device__ __forceinline__ void edge_accumulate_range(const double* pack, uint32_t e0, uint32_t e1, const uint32_t* dst,
                                                      uint32_t i, double p0, double p1, double (&acc)[3][6]) {
    const double* a = pack + (size_t)i * kPackWidth;
    const double ma = a[9];
    uint32_t e = e0;
    if constexpr (PAIR::kEdgeBatch > 1) {
        constexpr int B = PAIR::kEdgeBatch;
        for (; e + B <= e1; e += B) {
            double rows[B][kPackWidth];
#pragma unroll
            for (int u = 0; u < B; u++) {
                const double* b = pack + (size_t)dst[e + u] * kPackWidth;
#pragma unroll
                for (int k = 0; k < kPackWidth; k++) rows[u][k] = (k < 3 * NS || k == 9) ? b[k] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < B; u++)
#pragma unroll
                for (int st = 0; st < NS; st++) PAIR::fold(acc[st], a + 3 * st, ma, rows[u] + 3 * st, rows[u][9], p0, p1);
        }
    }
    for (; e < e1; e++) {  // spawn order inside a source
        const double* b = pack + (size_t)dst[e] * kPackWidth;
        const double mb = b[9];
#pragma unroll
        for (int st = 0; st < NS; st++) PAIR::fold(acc[st], a + 3 * st, ma, b + 3 * st, mb, p0, p1);
    }
}

template <int NS, class PAIR>
__device__ __forceinline__ void edge_accumulate(const double* pack, const uint32_t* __restrict__ row_start,
                                                const uint32_t* __restrict__ dst, uint32_t i, double p0, double p1,
                                                double (&acc)[3][6]) {
    edge_accumulate_range<NS, PAIR>(pack, row_start[i], row_start[i + 1], dst, i, p0, p1, acc);
}

// ---- 2c. hub sources ----------------------------------------------------------------------------------------
// One lane per source serialises on a source's out-degree: a hub with 10^5 out-edges would hold its wave for 10^5
// dependent gathers while every other lane idles.  The reference buckets sources by out-degree for the same reason
// (graph.rs:290-328).  Here, for additive folds, a hub's edge range is cut into chunks of kHubChunk edges and ONE WAVE folds
// a chunk: lane l takes edges l, l+64, ... (each lane its own accumulator, in edge order), the 64 accumulators are summed
// by a fixed shuffle tree, and a second small kernel (one wave per hub) adds the hub's chunk sums the same way.  Same sum, different
// association than the sequential fold (~1e-16 * sqrt(degree) relative), identical from run to run.
template <int NS, class PAIR>
__global__ __launch_bounds__(64) void edge_hub_chunk_kernel(const double* __restrict__ pack, const uint32_t* __restrict__ row_start,
                                                            const uint32_t* __restrict__ dst, const uint32_t* __restrict__ chunk_e0,
                                                            const uint32_t* __restrict__ chunk_row, double* __restrict__ chunk_partial,
                                                            double p0, double p1) {
    const uint32_t c = blockIdx.x, lane = threadIdx.x;
    const uint32_t i = chunk_row[c];
    const uint32_t e0 = chunk_e0[c], e1 = min(e0 + kHubChunk, row_start[i + 1]);
    const double* a = pack + (size_t)i * kPackWidth;
    const double ma = a[9];
    double acc[3][6] = {{0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}};
    for (uint32_t e = e0 + lane; e < e1; e += 64) {
        const double* b = pack + (size_t)dst[e] * kPackWidth;
        const double mb = b[9];
#pragma unroll
        for (int st = 0; st < NS; st++) PAIR::fold(acc[st], a + 3 * st, ma, b + 3 * st, mb, p0, p1);
    }
#pragma unroll
    for (int st = 0; st < NS; st++)
#pragma unroll
        for (int k = 0; k < 6; k++) {
            double v = acc[st][k];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0) chunk_partial[(size_t)c * kPartialWidth + 6 * st + k] = v;
        }
}

// One wave per hub: lane l adds chunk sums l, l+64, ... in chunk order, then the same shuffle tree.
template <int NS>
__global__ __launch_bounds__(64) void edge_hub_reduce_kernel(const uint32_t* __restrict__ hub_rows, const uint32_t* __restrict__ hub_chunk_start,
                                                             const double* __restrict__ chunk_partial, double* __restrict__ partial) {
    const uint32_t hb = blockIdx.x, lane = threadIdx.x;
    double acc[NS * 6];
#pragma unroll
    for (int k = 0; k < NS * 6; k++) acc[k] = 0.0;
    for (uint32_t c = hub_chunk_start[hb] + lane; c < hub_chunk_start[hb + 1]; c += 64)
#pragma unroll
        for (int k = 0; k < NS * 6; k++) acc[k] += chunk_partial[(size_t)c * kPartialWidth + k];
    double* o = partial + (size_t)hub_rows[hb] * kPartialWidth;
#pragma unroll
    for (int k = 0; k < NS * 6; k++) {
        double v = acc[k];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) o[k] = v;
    }
}

// ---- 3. integrate ---------------------------------------------------------------------------------------
// Per-entity half of the tick: same stage structure as sixdof_step_kernel, with the pair forces
// taken from `pf`.  Per-entity ops that precede the pair op in the pipe only survive on
// rows that are not edge sources.
struct EntityState {
    Quat<double> q0;
    Vec3<double> p0;
    Spatial<double> v0;
    Vec3<double> I, inv_I;
    double mass, inv_m;
};

// `A` comes in holding the world_accel column of this entity (a_in): RK4 stage 0 forms v_s = v0 + 0 * a_in like the
// reference (rk4.rs:96-100), so non-finite input poisons the tick the same way; it goes out as this tick's acceleration.
template <int INTEGRATOR>
__device__ __forceinline__ void pair_integrate_entity(const PairParams& P, const StepParams& SP,
                                                      const Vec3<double> (&aux)[kMaxOps], const double (&pf)[3][6],
                                                      bool is_source, EntityState& e, Spatial<double>& A,
                                                      Spatial<double>& Fw) {
    using T = double;
    using PIPE = PipeGeneric;
    Quat<T>& q0 = e.q0;
    Vec3<T>& p0 = e.p0;
    Spatial<T>& v0 = e.v0;
    const Vec3<T> inv_I = e.inv_I;
    const T inv_m = e.inv_m;
    Body<T> b;
    b.mass = e.mass;
    b.I = e.I;
    Wrench<T> F;
    auto stage_force = [&](int st) {
        F = zero_wrench<T>();
        if (P.n_ops) PIPE::apply(SP, aux, NoModel::Regs<T>{}, b, F);
        if (is_source) {  // the edge_fold output replaces Force on source rows
            F = zero_wrench<T>();
            F.tau_w = Vec3<T>{pf[st][0], pf[st][1], pf[st][2]};
            F.f = Vec3<T>{pf[st][3], pf[st][4], pf[st][5]};
        }
    };
    const T dt_g = P.dt_g, dt = P.dt;
    if constexpr (INTEGRATOR == kRk4) {
        const T h1 = dt_g * 0.5, h3 = dt_g;
        Spatial<T> sv, sa;
        b.q = normalized(q0); b.p = p0;
        b.v = Spatial<T>{v0.ang + T(0) * A.ang, v0.lin + T(0) * A.lin};
        stage_force(0);
        A = calc_accel<PIPE>(b.q, F, inv_I, inv_m, T(0));
        sv = b.v; sa = A;
        b.q = integrate_world(q0, h1 * v0.ang); b.p = axpy(h1, v0.lin, p0); b.v = axpy(h1, A, v0);
        sv = axpy(T(2), b.v, sv);
        stage_force(1);
        A = calc_accel<PIPE>(b.q, F, inv_I, inv_m, T(0));
        sa = axpy(T(2), A, sa);
        b.v = axpy(h1, A, v0);
        sv = axpy(T(2), b.v, sv);
        stage_force(1);
        A = calc_accel<PIPE>(b.q, F, inv_I, inv_m, T(0));
        sa = axpy(T(2), A, sa);
        T n3;
        b.q = integrate_world(q0, h3 * v0.ang, &n3); b.p = axpy(h3, v0.lin, p0); b.v = axpy(h3, A, v0);
        const T taint = accel_taint(n3);   // effectors.hpp: the last stage carries a non-finite attitude into A, v' and x'
        sv = sv + b.v;
        stage_force(2);
        A = calc_accel<PIPE>(b.q, F, inv_I, inv_m + taint, taint);
        sa = sa + A;
        const T g = dt * T(1.0 / 6.0);
        q0 = integrate_world(q0, g * sv.ang);
        p0 = axpy(g + taint, sv.lin, p0);
        v0 = axpy(g, sa, v0);
    } else {
        T n0;
        b.q = normalized(q0, &n0); b.p = p0; b.v = v0;
        const T taint = accel_taint(n0);
        stage_force(0);
        A = calc_accel<PIPE>(b.q, F, inv_I, inv_m + taint, taint);
        v0 = axpy(dt, A, v0);
        q0 = integrate_world(q0, dt * v0.ang);
        p0 = axpy(dt, v0.lin, p0);
    }
    Fw = world_wrench<PIPE>(b.q, F);
}

__device__ __forceinline__ void load_entity(const PairParams& P, uint32_t i, EntityState& e, StepParams& SP,
                                            Vec3<double> (&aux)[kMaxOps]) {
    const double* pos = static_cast<const double*>(P.pos) + (size_t)i * 7;
    const double* vel = static_cast<const double*>(P.vel) + (size_t)i * 6;
    const double* in = static_cast<const double*>(P.inertia) + (size_t)i * 7;
    e.q0 = {pos[0], pos[1], pos[2], pos[3]};
    e.p0 = {pos[4], pos[5], pos[6]};
    e.v0 = {{vel[0], vel[1], vel[2]}, {vel[3], vel[4], vel[5]}};
    e.I = {in[0], in[1], in[2]};
    e.inv_I = {1.0 / in[0], 1.0 / in[1], 1.0 / in[2]};
    e.mass = in[6];
    e.inv_m = 1.0 / in[6];
    SP.n_ops = P.n_ops;   // view of the per-entity ops for the shared effector code
    SP.vel_independent = 0;
#pragma unroll
    for (int k = 0; k < kMaxOps; k++) SP.ops[k] = P.ops[k];
#pragma unroll
    for (int k = 0; k < kMaxOps; k++) {
        aux[k] = Vec3<double>{0, 0, 0};
        if (k < (int)P.n_ops && P.ops[k].aux != nullptr) {
            const double* a = static_cast<const double*>(P.ops[k].aux) + (size_t)i * 3;
            aux[k] = Vec3<double>{a[0], a[1], a[2]};
        }
    }
}

__device__ __forceinline__ void store_entity(const PairParams& P, uint32_t i, const EntityState& e,
                                             const Spatial<double>& A, const Spatial<double>& Fw) {
    double* pos = static_cast<double*>(P.pos) + (size_t)i * 7;
    double* vel = static_cast<double*>(P.vel) + (size_t)i * 6;
    pos[0] = e.q0.i; pos[1] = e.q0.j; pos[2] = e.q0.k; pos[3] = e.q0.w; pos[4] = e.p0.x; pos[5] = e.p0.y; pos[6] = e.p0.z;
    vel[0] = e.v0.ang.x; vel[1] = e.v0.ang.y; vel[2] = e.v0.ang.z;
    vel[3] = e.v0.lin.x; vel[4] = e.v0.lin.y; vel[5] = e.v0.lin.z;
    double* ac = static_cast<double*>(P.accel) + (size_t)i * 6;
    ac[0] = A.ang.x; ac[1] = A.ang.y; ac[2] = A.ang.z; ac[3] = A.lin.x; ac[4] = A.lin.y; ac[5] = A.lin.z;
    double* fo = static_cast<double*>(P.force) + (size_t)i * 6;
    fo[0] = Fw.ang.x; fo[1] = Fw.ang.y; fo[2] = Fw.ang.z; fo[3] = Fw.lin.x; fo[4] = Fw.lin.y; fo[5] = Fw.lin.z;
}

// The per-entity ops' view of PairParams and this row's effector columns (24-byte rows, read per lane like the step kernel).
__device__ __forceinline__ void load_ops(const PairParams& P, uint32_t i, bool active, StepParams& SP, Vec3<double> (&aux)[kMaxOps]) {
    SP.n_ops = P.n_ops;
    SP.vel_independent = 0;
#pragma unroll
    for (int k = 0; k < kMaxOps; k++) SP.ops[k] = P.ops[k];
#pragma unroll
    for (int k = 0; k < kMaxOps; k++) {
        aux[k] = Vec3<double>{0, 0, 0};
        if (active && k < (int)P.n_ops && P.ops[k].aux != nullptr) {
            const double* a = static_cast<const double*>(P.ops[k].aux) + (size_t)i * 3;
            aux[k] = Vec3<double>{a[0], a[1], a[2]};
        }
    }
}

// The integrate half of an ALL-PAIRS tick (edge lists fold and integrate in one launch: 3b).  Single-wave workgroups, 64 rows
// each (the step kernel's memory plan): pos / vel / inertia / world_accel come in as slabs by LDS-DMA; pos / vel / accel / force
// and the NEXT tick's pack rows leave as slabs; the `splits` partial force sums (72 B each) are added per lane in fixed order.  The arithmetic is load_entity / pair_integrate_entity / pack_row, i.e. exactly the one-launch small-graph
// kernel's: the two paths stay bit-identical (tests/test_gpu_parity.py::test_small_graph_single_launch_path_is_bit_identical).
// (1.0 / x stays an IEEE divide here — four per entity and launch — because recip() may differ from it in the last bit.)
template <int INTEGRATOR, bool PACK_NEXT>
__global__ __launch_bounds__(kWave) void pair_integrate_kernel(const PairParams P) {
    // in: pos 7 | vel 6 | inertia 7 | accel 6 = 26 doubles per row; out: pos 7 | vel 6 | accel 6 | force 6 | pack 10 = 35
    __shared__ __attribute__((aligned(16))) double lds[kWave * 35];
    double* const l_pos = lds;
    double* const l_vel = lds + kWave * 7;
    double* const l_in = lds + kWave * 13;
    double* const l_acc = lds + kWave * 20;
    const uint32_t row0 = blockIdx.x * kWave, t = threadIdx.x, i = row0 + t;
    const uint32_t rows = min((uint32_t)kWave, P.n - row0);
    const bool full = rows == kWave, active = t < rows;
    double* const g_pos = static_cast<double*>(P.pos) + (size_t)row0 * 7;
    double* const g_vel = static_cast<double*>(P.vel) + (size_t)row0 * 6;
    double* const g_acc = static_cast<double*>(P.accel) + (size_t)row0 * 6;
    double* const g_force = static_cast<double*>(P.force) + (size_t)row0 * 6;
    const double* const g_in = static_cast<const double*>(P.inertia) + (size_t)row0 * 7;
    if (full) {
        slab_dma_in<kWave * 7 * 8, kPolPlain>(reinterpret_cast<const char*>(g_pos), reinterpret_cast<char*>(l_pos), t);
        slab_dma_in<kWave * 6 * 8, kPolPlain>(reinterpret_cast<const char*>(g_vel), reinterpret_cast<char*>(l_vel), t);
        slab_dma_in<kWave * 7 * 8, kPolPlain>(reinterpret_cast<const char*>(g_in), reinterpret_cast<char*>(l_in), t);
        slab_dma_in<kWave * 6 * 8, kPolPlain>(reinterpret_cast<const char*>(g_acc), reinterpret_cast<char*>(l_acc), t);
    } else {
        slab_in_tail(g_pos, l_pos, rows * 7, t);
        slab_in_tail(g_vel, l_vel, rows * 6, t);
        slab_in_tail(g_in, l_in, rows * 7, t);
        slab_in_tail(g_acc, l_acc, rows * 6, t);
    }
    StepParams SP;
    Vec3<double> aux[kMaxOps];
    load_ops(P, i, active, SP, aux);
    constexpr int NS = INTEGRATOR == kRk4 ? 3 : 1;
    double pf[3][6] = {{0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}};
    const bool is_source = active && P.n > 1;
    if (active) {      // `splits` partial sums of 72 bytes each, added in fixed order (deterministic)
        for (uint32_t sp = 0; sp < P.splits; sp++) {
            const double* part = P.partial + ((size_t)sp * P.n + i) * kPartialForce;
#pragma unroll
            for (int st = 0; st < NS; st++)
                for (int c = 0; c < 3; c++) pf[st][3 + c] += part[3 * st + c];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // LDS-DMA data has landed
    __syncthreads();
    EntityState e;
    Spatial<double> A = {{0, 0, 0}, {0, 0, 0}}, Fw = {{0, 0, 0}, {0, 0, 0}};
    if (active) {
        const double* pos = l_pos + t * 7;
        const double* vel = l_vel + t * 6;
        const double* in = l_in + t * 7;
        e.q0 = {pos[0], pos[1], pos[2], pos[3]};
        e.p0 = {pos[4], pos[5], pos[6]};
        e.v0 = {{vel[0], vel[1], vel[2]}, {vel[3], vel[4], vel[5]}};
        e.I = {in[0], in[1], in[2]};
        e.inv_I = {1.0 / in[0], 1.0 / in[1], 1.0 / in[2]};
        e.mass = in[6];
        e.inv_m = 1.0 / in[6];
        const double* ac = l_acc + t * 6;
        A = Spatial<double>{{ac[0], ac[1], ac[2]}, {ac[3], ac[4], ac[5]}};
    }
    __syncthreads();  // every lane has consumed the input slabs; LDS is the output staging area from here on
    if (active) pair_integrate_entity<INTEGRATOR>(P, SP, aux, pf, is_source, e, A, Fw);
    double* const o_pos = lds;
    double* const o_vel = lds + kWave * 7;
    double* const o_acc = lds + kWave * 13;
    double* const o_force = lds + kWave * 19;
    double* const o_pack = lds + kWave * 25;
    if (active) {
        double* r = o_pos + t * 7;
        r[0] = e.q0.i; r[1] = e.q0.j; r[2] = e.q0.k; r[3] = e.q0.w; r[4] = e.p0.x; r[5] = e.p0.y; r[6] = e.p0.z;
        double* v = o_vel + t * 6;
        v[0] = e.v0.ang.x; v[1] = e.v0.ang.y; v[2] = e.v0.ang.z; v[3] = e.v0.lin.x; v[4] = e.v0.lin.y; v[5] = e.v0.lin.z;
        double* a = o_acc + t * 6;
        a[0] = A.ang.x; a[1] = A.ang.y; a[2] = A.ang.z; a[3] = A.lin.x; a[4] = A.lin.y; a[5] = A.lin.z;
        double* f = o_force + t * 6;
        f[0] = Fw.ang.x; f[1] = Fw.ang.y; f[2] = Fw.ang.z; f[3] = Fw.lin.x; f[4] = Fw.lin.y; f[5] = Fw.lin.z;
        if constexpr (PACK_NEXT) {
            const double x[3] = {e.p0.x, e.p0.y, e.p0.z}, vl[3] = {e.v0.lin.x, e.v0.lin.y, e.v0.lin.z};
            pack_row(o_pack + t * kPackWidth, x, vl, e.mass, P.dt_g * 0.5, P.dt_g);
        }
    }
    __syncthreads();
    double* const g_pack = P.pack + (size_t)row0 * kPackWidth;
    if (full) {
        slab_out<kWave * 7 * 8, kPolPlain>(reinterpret_cast<const char*>(o_pos), reinterpret_cast<char*>(g_pos), t);
        slab_out<kWave * 6 * 8, kPolPlain>(reinterpret_cast<const char*>(o_vel), reinterpret_cast<char*>(g_vel), t);
        slab_out<kWave * 6 * 8, kPolPlain>(reinterpret_cast<const char*>(o_acc), reinterpret_cast<char*>(g_acc), t);
        slab_out<kWave * 6 * 8, kPolNtStores>(reinterpret_cast<const char*>(o_force), reinterpret_cast<char*>(g_force), t);   // written, never read back
        if constexpr (PACK_NEXT) slab_out<kWave * kPackWidth * 8, kPolPlain>(reinterpret_cast<const char*>(o_pack), reinterpret_cast<char*>(g_pack), t);
    } else {
        slab_out_tail(o_pos, g_pos, rows * 7, t);
        slab_out_tail(o_vel, g_vel, rows * 6, t);
        slab_out_tail(o_acc, g_acc, rows * 6, t);
        slab_out_tail(o_force, g_force, rows * 6, t);
        if constexpr (PACK_NEXT) slab_out_tail(o_pack, g_pack, rows * kPackWidth, t);
    }
}

// ---- 3b. edge lists: fold + integrate in ONE launch (hub sources' sums come from the two hub launches in front of it) -------------
// The edge fold reads only the PACK rows (of the source and of its targets); the integrate half reads the source's own state.
// With the pack rows double-buffered — this tick reads `pack`, writes the next tick's rows to `pack_next` — nothing a wave reads
// is written by another wave of the same launch, so the two halves need no launch boundary between them: a tick is one kernel,
// the [n, 18] partial rows (144 B written and read back per entity and tick) never exist, and the state slabs' LDS-DMA is in
// flight while the lane folds its edges.  Same device functions as the two-kernel path in the same order: identical bits.
template <int INTEGRATOR, class PAIR>
__global__ __launch_bounds__(kWave) void pair_tick_fused_kernel(const PairParams P) {
    __shared__ __attribute__((aligned(16))) double lds[kWave * 35];      // in: pos 7 | vel 6 | inertia 7 | accel 6; out: + force 6 | pack 10
    double* const l_pos = lds;
    double* const l_vel = lds + kWave * 7;
    double* const l_in = lds + kWave * 13;
    double* const l_acc = lds + kWave * 20;
    const uint32_t row0 = blockIdx.x * kWave, t = threadIdx.x, i = row0 + t;
    const uint32_t rows = min((uint32_t)kWave, P.n - row0);
    const bool full = rows == kWave, active = t < rows;
    double* const g_pos = static_cast<double*>(P.pos) + (size_t)row0 * 7;
    double* const g_vel = static_cast<double*>(P.vel) + (size_t)row0 * 6;
    double* const g_acc = static_cast<double*>(P.accel) + (size_t)row0 * 6;
    double* const g_force = static_cast<double*>(P.force) + (size_t)row0 * 6;
    const double* const g_in = static_cast<const double*>(P.inertia) + (size_t)row0 * 7;
    if (full) {
        slab_dma_in<kWave * 7 * 8, kPolPlain>(reinterpret_cast<const char*>(g_pos), reinterpret_cast<char*>(l_pos), t);
        slab_dma_in<kWave * 6 * 8, kPolPlain>(reinterpret_cast<const char*>(g_vel), reinterpret_cast<char*>(l_vel), t);
        slab_dma_in<kWave * 7 * 8, kPolPlain>(reinterpret_cast<const char*>(g_in), reinterpret_cast<char*>(l_in), t);
        slab_dma_in<kWave * 6 * 8, kPolPlain>(reinterpret_cast<const char*>(g_acc), reinterpret_cast<char*>(l_acc), t);
    } else {
        slab_in_tail(g_pos, l_pos, rows * 7, t);
        slab_in_tail(g_vel, l_vel, rows * 6, t);
        slab_in_tail(g_in, l_in, rows * 7, t);
        slab_in_tail(g_acc, l_acc, rows * 6, t);
    }
    StepParams SP;
    Vec3<double> aux[kMaxOps];
    load_ops(P, i, active, SP, aux);
    constexpr int NS = INTEGRATOR == kRk4 ? 3 : 1;
    double pf[3][6] = {{0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}};
    bool is_source = false;
    if (active) {      // the fold, while the slabs land
        const uint32_t e0 = P.row_start[i], e1 = P.row_start[i + 1];
        is_source = e1 > e0;
        if (PAIR::kAdditive && P.n_hubs && e1 - e0 >= kHubDegree) {
            // a hub source: its edges were folded by whole waves in front of this launch (2c), the sums wait in its partial row
            const double* part = P.partial + (size_t)i * kPartialWidth;
#pragma unroll
            for (int st = 0; st < NS; st++)
                for (int c = 0; c < 6; c++) pf[st][c] = part[6 * st + c];
        } else {
            edge_accumulate_range<NS, PAIR>(P.pack, e0, e1, P.dst, i, P.p0, P.p1, pf);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    EntityState e;
    Spatial<double> A = {{0, 0, 0}, {0, 0, 0}}, Fw = {{0, 0, 0}, {0, 0, 0}};
    if (active) {
        const double* pos = l_pos + t * 7;
        const double* vel = l_vel + t * 6;
        const double* in = l_in + t * 7;
        e.q0 = {pos[0], pos[1], pos[2], pos[3]};
        e.p0 = {pos[4], pos[5], pos[6]};
        e.v0 = {{vel[0], vel[1], vel[2]}, {vel[3], vel[4], vel[5]}};
        e.I = {in[0], in[1], in[2]};
        e.inv_I = {1.0 / in[0], 1.0 / in[1], 1.0 / in[2]};
        e.mass = in[6];
        e.inv_m = 1.0 / in[6];
        const double* ac = l_acc + t * 6;
        A = Spatial<double>{{ac[0], ac[1], ac[2]}, {ac[3], ac[4], ac[5]}};
    }
    __syncthreads();
    if (active) pair_integrate_entity<INTEGRATOR>(P, SP, aux, pf, is_source, e, A, Fw);
    double* const o_pos = lds;
    double* const o_vel = lds + kWave * 7;
    double* const o_acc = lds + kWave * 13;
    double* const o_force = lds + kWave * 19;
    double* const o_pack = lds + kWave * 25;
    if (active) {
        double* r = o_pos + t * 7;
        r[0] = e.q0.i; r[1] = e.q0.j; r[2] = e.q0.k; r[3] = e.q0.w; r[4] = e.p0.x; r[5] = e.p0.y; r[6] = e.p0.z;
        double* v = o_vel + t * 6;
        v[0] = e.v0.ang.x; v[1] = e.v0.ang.y; v[2] = e.v0.ang.z; v[3] = e.v0.lin.x; v[4] = e.v0.lin.y; v[5] = e.v0.lin.z;
        double* a = o_acc + t * 6;
        a[0] = A.ang.x; a[1] = A.ang.y; a[2] = A.ang.z; a[3] = A.lin.x; a[4] = A.lin.y; a[5] = A.lin.z;
        double* f = o_force + t * 6;
        f[0] = Fw.ang.x; f[1] = Fw.ang.y; f[2] = Fw.ang.z; f[3] = Fw.lin.x; f[4] = Fw.lin.y; f[5] = Fw.lin.z;
        const double x[3] = {e.p0.x, e.p0.y, e.p0.z}, vl[3] = {e.v0.lin.x, e.v0.lin.y, e.v0.lin.z};
        pack_row(o_pack + t * kPackWidth, x, vl, e.mass, P.dt_g * 0.5, P.dt_g);
    }
    __syncthreads();
    double* const g_pack = P.pack_next + (size_t)row0 * kPackWidth;
    if (full) {
        slab_out<kWave * 7 * 8, kPolPlain>(reinterpret_cast<const char*>(o_pos), reinterpret_cast<char*>(g_pos), t);
        slab_out<kWave * 6 * 8, kPolPlain>(reinterpret_cast<const char*>(o_vel), reinterpret_cast<char*>(g_vel), t);
        slab_out<kWave * 6 * 8, kPolPlain>(reinterpret_cast<const char*>(o_acc), reinterpret_cast<char*>(g_acc), t);
        slab_out<kWave * 6 * 8, kPolNtStores>(reinterpret_cast<const char*>(o_force), reinterpret_cast<char*>(g_force), t);
        slab_out<kWave * kPackWidth * 8, kPolPlain>(reinterpret_cast<const char*>(o_pack), reinterpret_cast<char*>(g_pack), t);
    } else {
        slab_out_tail(o_pos, g_pos, rows * 7, t);
        slab_out_tail(o_vel, g_vel, rows * 6, t);
        slab_out_tail(o_acc, g_acc, rows * 6, t);
        slab_out_tail(o_force, g_force, rows * 6, t);
        slab_out_tail(o_pack, g_pack, rows * kPackWidth, t);
    }
}

// ---- small graphs: the whole tick (and n_ticks of them) in ONE single-workgroup launch --------------------------
// Three-body / solar-system sized worlds (n <= 256) are launch-bound, not math-bound: pack, fold and integrate
// run in one workgroup with the packed sources in LDS and the entity state in registers across ticks.  Same device
// functions as the multi-kernel path, so results are bit-identical to it.
template <int INTEGRATOR, class PAIR>
__global__ __launch_bounds__(kTile) void pair_small_kernel(const PairParams P, uint32_t n_ticks) {
    __shared__ __attribute__((aligned(16))) double pack[kTile * kPackWidth];
    __shared__ uint32_t l_dst[kSmallEdgeCache];   // the edge list of a small graph lives in LDS for the whole launch
    constexpr int NS = INTEGRATOR == kRk4 ? 3 : 1;
    const uint32_t i = threadIdx.x;
    const bool active = i < P.n;
    EntityState e;
    StepParams SP;
    Vec3<double> aux[kMaxOps];
    if (active) load_entity(P, i, e, SP, aux);
    const bool allpairs = P.pair_kind == SIXDOF_EFF_ALLPAIRS_GRAVITY_SOFTENED;
    // CSR range of this source in registers, targets in LDS: the per-tick fold then touches no global memory
    uint32_t e0 = 0, e1 = 0;
    const bool cached = !allpairs && P.n_edges <= kSmallEdgeCache;
    if (!allpairs) {
        if (active) { e0 = P.row_start[i]; e1 = P.row_start[i + 1]; }
        if (cached)
            for (uint32_t k = threadIdx.x; k < P.n_edges; k += blockDim.x) l_dst[k] = P.dst[k];
    }
    const uint32_t* const dst = cached ? l_dst : P.dst;
    const bool is_source = active && (allpairs ? (P.n > 1) : (e1 > e0));
    const double h1 = P.dt_g * 0.5, h3 = P.dt_g;
    Spatial<double> A = {{0, 0, 0}, {0, 0, 0}}, Fw = {{0, 0, 0}, {0, 0, 0}};
    if (active) {   // a_in of the first tick = the world_accel column; of later ticks = the previous tick's A, in registers
        const double* ac = static_cast<const double*>(P.accel) + (size_t)i * 6;
        A = Spatial<double>{{ac[0], ac[1], ac[2]}, {ac[3], ac[4], ac[5]}};
    }
    for (uint32_t t = 0; t < n_ticks; t++) {
        if (active) {
            const double x[3] = {e.p0.x, e.p0.y, e.p0.z}, v[3] = {e.v0.lin.x, e.v0.lin.y, e.v0.lin.z};
            pack_row(pack + i * kPackWidth, x, v, e.mass, h1, h3);
        }
        __syncthreads();
        double pf[3][6] = {{0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}};
        if (active) {
            if (allpairs) {
                double acc[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
                double pi[3][3];
                const double* s = pack + i * kPackWidth;
#pragma unroll
                for (int st = 0; st < 3; st++)
                    for (int c = 0; c < 3; c++) pi[st][c] = st < NS ? s[3 * st + c] : 0.0;
                tile_accumulate<NS, true>(pack, (int)P.n, 0, i, pi, P.p1, acc);
                const double kmi = P.p0 * e.mass;
#pragma unroll
                for (int st = 0; st < NS; st++)
                    for (int c = 0; c < 3; c++) pf[st][3 + c] += kmi * acc[st][c];
            } else {
                double acc[3][6] = {{0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}};
                edge_accumulate_range<NS, PAIR>(pack, e0, e1, dst, i, P.p0, P.p1, acc);
#pragma unroll
                for (int st = 0; st < NS; st++)
                    for (int c = 0; c < 6; c++) pf[st][c] += acc[st][c];
            }
        }
        __syncthreads();   // every lane has read the packed sources before the next tick overwrites them
        if (active) pair_integrate_entity<INTEGRATOR>(P, SP, aux, pf, is_source, e, A, Fw);
    }
    if (active && n_ticks) store_entity(P, i, e, A, Fw);
}

// ---- launch helpers (shared with generated translation units) ------------------------------------------------

inline uint32_t pair_splits_for_n(uint32_t n) {
    const uint32_t tblocks = (n + kTile - 1) / kTile;
    if (tblocks == 0) return 1;
    uint32_t s = (1024 + tblocks - 1) / tblocks;  // aim for >= 4 workgroups per CU
    if (s > tblocks) s = tblocks;                 // at least one tile per split
    return s ? s : 1;
}

// n <= kPairSmallMax: one single-workgroup launch for n_ticks ticks.
// ONLY: -1 instantiates both integrators (the product library); a generated object built for one executor names the integrator
// it will be launched with and carries that kernel alone — half the device code to compile (codegen.generate_pair_source).
// Launching such an object with the other integrator is an error, not a silent substitution.
template <class PAIR, int ONLY = -1>
inline hipError_t launch_pair_small_t(const PairParams& p, int integrator, uint32_t n_ticks, hipStream_t stream,
                                      uint64_t* launches) {
    if (p.n == 0 || n_ticks == 0) return hipSuccess;
    if (ONLY >= 0 && integrator != ONLY) return hipErrorInvalidValue;
    const dim3 block(p.n <= 64 ? 64 : kTile);   // one wave when it suffices: its barriers cost nothing
    if constexpr (ONLY != kSemiImplicit) {
        if (integrator == kRk4) hipLaunchKernelGGL((pair_small_kernel<kRk4, PAIR>), dim3(1), block, 0, stream, p, n_ticks);
    }
    if constexpr (ONLY != kRk4) {
        if (integrator != kRk4) hipLaunchKernelGGL((pair_small_kernel<kSemiImplicit, PAIR>), dim3(1), block, 0, stream, p, n_ticks);
    }
    if (launches) *launches += 1;
    return hipGetLastError();
}

// A batch of ticks: pack once (unless the caller says `pack` already holds the rows of the current state: `packed`), then per
// tick fold -> integrate, the integrate kernel writing the next tick's pack rows.  ALLPAIRS selects the tiled complete-graph
// kernel (softened gravity).  `last_packs`: the final tick also leaves its pack rows (so a following batch may skip the pack).
template <class PAIR, bool ALLPAIRS, int ONLY = -1>
inline hipError_t launch_pair_ticks_t(const PairParams& p, int integrator, uint32_t n_ticks, bool packed, hipStream_t stream, uint64_t* launches) {
    if (p.n == 0 || n_ticks == 0) return hipSuccess;
    if (ONLY >= 0 && integrator != ONLY) return hipErrorInvalidValue;
    const uint32_t waves = (p.n + kWave - 1) / kWave;
    const double h1 = p.dt_g * 0.5, h3 = p.dt_g;
    const bool fused_path = !ALLPAIRS && p.pack_next != nullptr;      // (its buffers alternate: every batch packs)
    if (!packed || fused_path) {
        hipLaunchKernelGGL(pair_pack_kernel, dim3(waves), dim3(kWave), 0, stream, static_cast<const double*>(p.pos),
                           static_cast<const double*>(p.vel), static_cast<const double*>(p.inertia), p.pack, p.n, h1, h3);
        if (launches) *launches += 1;
    }
    const bool rk4 = integrator == kRk4;
    if constexpr (!ALLPAIRS) {
        // with a second pack buffer: ONE launch per tick (+ the two hub launches in front of it when there are hub sources)
        if (p.pack_next != nullptr) {
            PairParams q = p;
            const uint32_t hubs = PAIR::kAdditive ? p.n_hubs : 0u;
            for (uint32_t t = 0; t < n_ticks; t++) {
                if (hubs) {
                    if constexpr (ONLY != kSemiImplicit) if (rk4) {
                        hipLaunchKernelGGL((edge_hub_chunk_kernel<3, PAIR>), dim3(q.n_hub_chunks), dim3(64), 0, stream, q.pack, q.row_start, q.dst, q.chunk_e0, q.chunk_row, q.chunk_partial, q.p0, q.p1);
                        hipLaunchKernelGGL(edge_hub_reduce_kernel<3>, dim3(hubs), dim3(64), 0, stream, q.hub_rows, q.hub_chunk_start, q.chunk_partial, q.partial);
                    }
                    if constexpr (ONLY != kRk4) if (!rk4) {
                        hipLaunchKernelGGL((edge_hub_chunk_kernel<1, PAIR>), dim3(q.n_hub_chunks), dim3(64), 0, stream, q.pack, q.row_start, q.dst, q.chunk_e0, q.chunk_row, q.chunk_partial, q.p0, q.p1);
                        hipLaunchKernelGGL(edge_hub_reduce_kernel<1>, dim3(hubs), dim3(64), 0, stream, q.hub_rows, q.hub_chunk_start, q.chunk_partial, q.partial);
                    }
                    if (launches) *launches += 2;
                }
                if constexpr (ONLY != kSemiImplicit) { if (rk4) hipLaunchKernelGGL((pair_tick_fused_kernel<kRk4, PAIR>), dim3(waves), dim3(kWave), 0, stream, q); }
                if constexpr (ONLY != kRk4) { if (!rk4) hipLaunchKernelGGL((pair_tick_fused_kernel<kSemiImplicit, PAIR>), dim3(waves), dim3(kWave), 0, stream, q); }
                double* const cur = q.pack;      // the rows just written are the next tick's
                q.pack = q.pack_next;
                q.pack_next = cur;
                if (launches) *launches += 1;
            }
            return hipGetLastError();
        }
    }
    if constexpr (ALLPAIRS) {
        for (uint32_t t = 0; t < n_ticks; t++) {
            const dim3 grid((p.n + kTile - 1) / kTile, p.splits);
            if constexpr (ONLY != kSemiImplicit) { if (rk4) hipLaunchKernelGGL(allpairs_kernel<3>, grid, dim3(kTile), 0, stream, p.pack, p.partial, p.n, p.splits, p.p0, p.p1); }
            if constexpr (ONLY != kRk4) { if (!rk4) hipLaunchKernelGGL(allpairs_kernel<1>, grid, dim3(kTile), 0, stream, p.pack, p.partial, p.n, p.splits, p.p0, p.p1); }
            // every tick writes the next tick's pack rows (5 KB per wave beside the 12.5 KB of state it writes anyway)
            if constexpr (ONLY != kSemiImplicit) { if (rk4) hipLaunchKernelGGL((pair_integrate_kernel<kRk4, true>), dim3(waves), dim3(kWave), 0, stream, p); }
            if constexpr (ONLY != kRk4) { if (!rk4) hipLaunchKernelGGL((pair_integrate_kernel<kSemiImplicit, true>), dim3(waves), dim3(kWave), 0, stream, p); }
            if (launches) *launches += 2;
        }
        return hipGetLastError();
    }
    return hipErrorInvalidValue;      // an edge list without the second pack buffer (PairParams::pack_next): not a launch this library makes
}

// One tick on its own (pack, then fold + integrate).
template <class PAIR, bool ALLPAIRS, int ONLY = -1>
inline hipError_t launch_pair_tick_t(const PairParams& p, int integrator, hipStream_t stream, uint64_t* launches) {
    return launch_pair_ticks_t<PAIR, ALLPAIRS, ONLY>(p, integrator, 1, false, stream, launches);
}

}  // namespace sixdof
