// kernels.hpp — launch interface between the C-ABI host layer (sixdof_capi.cpp) and the gfx950
// kernels (sixdof_kernels.hip, nbody_kernels.hip).  Internal; not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Floating-point contraction is decided per SOURCE EXPRESSION (a * b + c written as one expression becomes one FMA) and
// nowhere else.  hipcc's default for device code ("fast") also fuses a multiply and an add that merely end up in the
// same basic block after inlining — so two inlined copies of the same tick body (loop body vs last tick, one launch
// shape vs another) could round differently.  With "on" every instantiation of the kernels computes the same bits:
// results do not depend on ticks_per_launch, graph replay, cache policy or which copy of the body ran a tick
// (tests/test_gpu_parity.py::test_fused_ticks_match_single_tick_launches).  Applies to everything that includes this
// header, generated translation units included.
#pragma clang fp contract(on)

namespace sixdof {

constexpr int kMaxOps = 4;       // per-entity effector ops fused into the step kernel
constexpr int kMaxModelCols = 128; // component columns of a generated program (kernarg: 2 pointers each; 128 keep StepParams at 2.4 KB of the 4 KB limit)
constexpr int kBlock = 256;      // threads per workgroup = entities per workgroup (4 waves of 64)

// One effector op as the kernel sees it (sixdof_effector_op with the aux column resolved).
struct DevOp {
    int32_t kind;
    int32_t aux_width;  // row width of `aux` (1..3); 0 = 3
    const void* aux;    // device [n,aux_width] column (element type = state dtype) or nullptr
    double p[6];
};

// Kernel argument block of the fused per-entity step.  Columns are device pointers in the
// reference's row-major layout: pos [n,7], vel [n,6], accel [n,6], force [n,6], inertia [n,7].
struct StepParams {
    void* pos;
    void* vel;
    void* accel;
    void* force;
    const void* inertia;
    uint32_t n;
    uint32_t n_ticks;         // ticks fused into this launch
    double dt_g;              // globals simulation_time_step (RK4 stage offsets)
    double dt;                // six_dof(time_step=) override or dt_g (final combination / semi-implicit)
    uint32_t n_ops;
    uint32_t vel_independent; // 1: no op reads the stage velocity -> RK4 stages 1 and 2 share F and A
    uint32_t streaming;       // cache-policy code of the launch: load policy * 8 + store policy (step_kernel.hpp)
    uint32_t hist_ring;       // history ring length in ticks (0 = off)
    uint64_t hist_slot0;      // ring slot index (before modulo) of the first tick of this launch
    void* hist_pos;           // [ring][n,7]   per-tick outputs, reference row layout; nullptr = no recording
    void* hist_vel;           // [ring][n,6]
    void* hist_accel;         // [ring][n,6]
    void* hist_force;         // [ring][n,6]
    uint64_t tick0;           // tick count before this launch (generated systems may read the tick)
    uint32_t accel_in_check;  // 1: the world_accel column holds HOST data (first launch after an upload): RK4 stage 0 reads it
    uint32_t reserved0;       //    the way the reference does, v_s = v0 + 0 * a_in (rk4.rs:96-100), so a non-finite row poisons the tick
    DevOp ops[kMaxOps];
    // generated programs only — behind everything the hand-written kernels read, so their kernarg loads stay within the
    // first 400 bytes whatever kMaxModelCols is
    void* model_cols[kMaxModelCols];  // device [n,w] component columns, read and written
    void* model_hist[kMaxModelCols];  // their history rings [ring][n,w] (nullptr = not recorded)
};

enum : int { kRk4 = 0, kSemiImplicit = 1, kNone = 2 };   // kNone: a pipe of systems without six_dof (generated programs only)

// Fused clear_forces | effectors | calc_accel | integrator over n entities, n_ticks ticks.
// dtype: 0 = f64, 1 = f32.  Returns hipGetLastError() of the launch.
hipError_t launch_step(const StepParams& p, int integrator, int dtype, hipStream_t stream);

// Entry points of a run-time generated effector pipe (elodin_amd/codegen.py), resolved with dlsym.
using CustomAbiFn = unsigned (*)();                                               // sizeof(StepParams) it was built with
using CustomLayoutFn = unsigned (*)();                                            // n_aux | n_model_cols << 8 | writes_inertia << 16
using CustomLaunchFn = int (*)(const StepParams*, int integrator, int dtype, void* stream);  // returns hipError_t

// ---- pairwise (edge_fold) path -----------------------------------------------------------------------
// One tick = pack -> accumulate -> integrate (3 launches).  See nbody_kernels.hip.
constexpr int kPackWidth = 10;   // per source: p(c=0) p(c=1/2) p(c=1) mass
constexpr int kPartialForce = 9;  // all-pairs gravity writes forces only
constexpr int kPartialWidth = 18; // per target and source split: 3 stage positions x Force [tau(3), f(3)]

struct PairParams {
    void* pos;            // [n,7]
    void* vel;            // [n,6]
    void* accel;          // [n,6]
    void* force;          // [n,6]
    const void* inertia;  // [n,7]
    uint32_t n;
    double dt_g, dt;
    double* pack;         // [n,10] scratch
    double* pack_next;    // [n,10] the pack rows the one-launch sparse tick writes for the NEXT tick (pair_kernel.hpp 3b); nullptr: not provided
    double* partial;      // [splits,n,partial_width] scratch
    uint32_t splits;      // source-range splits of the all-pairs kernel (1 for edge lists)
    uint32_t partial_width;  // kPartialForce (all-pairs: 3 stages x force) or kPartialWidth (3 stages x [tau, f])
    // edge list in CSR-by-source form (spawn order preserved inside a source), device
    const uint32_t* row_start;  // [n+1]
    const uint32_t* dst;        // [n_edges]
    uint32_t n_edges;
    // hub sources (out-degree >= kHubDegree), folded by whole waves instead of one lane each (pair_kernel.hpp 2c):
    // their edge ranges cut into chunks of <= kHubChunk edges
    uint32_t n_hubs, n_hub_chunks;
    const uint32_t* hub_rows;          // [n_hubs] source rows
    const uint32_t* hub_chunk_start;   // [n_hubs+1] first chunk of each hub
    const uint32_t* chunk_e0;          // [n_hub_chunks] first edge of each chunk (it ends kHubChunk later or with its source)
    const uint32_t* chunk_row;         // [n_hub_chunks] the chunk's source row
    double* chunk_partial;             // [n_hub_chunks, kPartialWidth] scratch
    int32_t pair_kind;    // sixdof_effector_kind 6,7,8
    double p0, p1;        // G | K, eps
    uint32_t n_ops;       // per-entity ops applied BEFORE the pair op (pipe order); they survive only on
    DevOp ops[kMaxOps];   // rows that are not edge sources (edge_fold replaces Force on source rows)
    // 1: `pack` already holds the rows of the current state (the previous tick's integrate kernel wrote them, and nothing has
    // touched pos / vel / inertia since): the batch starts without a pack launch.  Set by the caller inside ONE step call only —
    // between calls a host may write the columns (upload, sixdof_device_column), so every call packs once.
    uint32_t packed;
};
// Picks the number of source splits for n targets so the all-pairs grid fills 256 CUs.
uint32_t pair_splits_for(uint32_t n);
// n_ticks ticks: one pack launch (unless p.packed), then fold + integrate per tick (the integrate kernel writes the next pack rows).
hipError_t launch_pair_ticks(const PairParams& p, int integrator, uint32_t n_ticks, hipStream_t stream, uint64_t* launches);
// n <= kPairSmallMax: pack, fold and integrate n_ticks ticks in one single-workgroup launch (bit-identical results).
constexpr uint32_t kPairSmallMax = 256;
constexpr uint32_t kHubDegree = 32;    // a source with this many out-edges or more is a hub
constexpr uint32_t kHubChunk = 256;    // edges one wave folds (4 per lane: a short dependent-gather chain)
hipError_t launch_pair_small(const PairParams& p, int integrator, uint32_t n_ticks, hipStream_t stream,
                             uint64_t* launches);
// Entry points of a generated pair-fold translation unit (codegen.py: generate_pair_source).
using CustomPairAbiFn = unsigned (*)();
using CustomPairLaunchFn = int (*)(const PairParams* p, int integrator, uint32_t n_ticks, int small, void* stream,
                                   uint64_t* launches);

// ---- joins (query.rs:599-725): gather / scatter of rows by constant u32 indices -----------------------------
hipError_t launch_gather_rows(void* dst, const void* src, const uint32_t* rows, uint32_t m, uint32_t w, size_t elem,
                              hipStream_t s);
hipError_t launch_scatter_rows(void* dst, const void* src, const uint32_t* rows, uint32_t m, uint32_t w, size_t elem,
                               hipStream_t s);

hipError_t launch_nonfinite(const void* pos, const void* vel, uint32_t n, size_t elem, uint8_t* flags,
                            unsigned long long* count, hipStream_t s);

// ---- Apollo-lander rollout model (include/sixdof_apollo.h) ---------------------------------------------
struct ApolloParams {
    double *pos, *vel, *accel, *force, *inertia;  // Body columns
    double* state;           // [n,16]
    const double* params;    // [n,17]
    double* guidance;        // [n,8]
    double* score;           // [n,4]
    double* result;          // [n,12]
    const double* tick_refs; // [n_ticks,8] per-tick reference profile values, device
    uint32_t n;
    uint32_t n_ticks;
    uint64_t tick0;          // tick count before this launch
    uint64_t max_ticks;
    uint32_t guidance_period;
    uint32_t ticks_per_telemetry;  // post_step runs when a batch of this many ticks has completed (impeller2_server.rs:553-678)
    double dt;               // globals simulation_time_step
};
hipError_t launch_apollo(const ApolloParams& p, hipStream_t stream);

}  // namespace sixdof
