/*
 * sixdof_hip.h — C ABI of the MI355X-native 6DOF rigid-body integrator.
 *
 * This is the drop-in boundary for ONE path of elodin-sys/elodin: the `six_dof`
 * system (RK4 / semi-implicit) executed over ECS columns.  Every entry point
 * names the reference interface it replaces (paths relative to the reference
 * checkout).  Plain pointers and sizes only; no C++/torch types cross the ABI.
 *
 *   reference boundary                                        this header
 *   --------------------------------------------------------  -----------------------
 *   CraneliftExec::new            cranelift_exec.rs:54-127    sixdof_create + sixdof_bind_columns
 *   CraneliftExec::invoke_batch   cranelift_exec.rs:129-195   sixdof_step
 *   JaxExec H2D / D2H per batch   jax_exec.rs:130-177         sixdof_upload / sixdof_download
 *   TickFn(inputs**, outputs**)   cranelift_exec.rs:11        sixdof_tick (+ sixdof_tick_bind)
 *   ExecMetadata slot tables      exec.rs:17-29               sixdof_tick_slots
 *   Error -> PyErr                error.rs                    negative sixdof_status + sixdof_last_error
 *   ComponentId::new              impeller2/src/types.rs:39-44 sixdof_component_id
 *   validate_rates dt quantise    world_builder.rs:211-243    sixdof_quantize_time_step
 *   six_dof(time_step, sys, integrator)  six_dof.rs:161-203   sixdof_desc + sixdof_set_effectors
 *   GraphQuery edges              graph.rs:17-41,113-175      sixdof_set_edges
 *   JIT of user systems           cranelift_compile.rs:13-162 sixdof_set_custom_pipe / sixdof_set_custom_pair
 *   World / Column / spawn        world.rs:23-45,193-229      sixdof_world_* + sixdof_bind_world
 *   commit_world_head per batch   impeller2_server.rs:390-438 sixdof_download_async / _wait, sixdof_set_history
 *   failure detection / resume    (process exit, DB replay)   sixdof_count_nonfinite, sixdof_get/set_tick
 *   campaign fan-out / fan-in     monte-carlo/src/lib.rs:1066-1140,2083  sixdof_comm_*, sixdof_campaign_broadcast / _gather
 *
 * Column byte layout is the reference's (`World.host`, world.rs:23-45): row r of a
 * component occupies bytes [r*size, (r+1)*size), little-endian, row-major, rows in
 * spawn order, `entity_ids[r]` u64 LE.  The same bytes are kept resident in HBM.
 *
 * Threading: one handle = one GPU = one caller thread (mirrors the reference, where the
 * exec is moved once into the sim thread, cranelift_exec.rs:31-51).  Handles on
 * different GPUs may be driven from different threads.  Nothing throws across the ABI.
 */
#ifndef SIXDOF_HIP_H
#define SIXDOF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SIXDOF_ABI_VERSION 2   /* 2: sixdof_timings.graph_launches, sixdof_apollo_tables.ticks_per_telemetry, effector kinds 10-11 */

typedef struct sixdof_handle sixdof_handle;

/* 0 = ok, negative = error.  Mirrors nox-py `Error` (libs/nox-py/src/error.rs). */
typedef enum sixdof_status {
    SIXDOF_OK = 0,
    SIXDOF_ERR_INVALID_ARGUMENT = -1,
    SIXDOF_ERR_COMPONENT_NOT_FOUND = -2,  /* Error::ComponentNotFound   */
    SIXDOF_ERR_VALUE_SIZE_MISMATCH = -3,  /* Error::ValueSizeMismatch   */
    SIXDOF_ERR_BACKEND = -4,              /* Error::CraneliftBackend(String) analogue: HIP runtime failure */
    SIXDOF_ERR_NO_DEVICE = -5,
    SIXDOF_ERR_UNSUPPORTED = -6,
    SIXDOF_ERR_ENTITY_MISMATCH = -7,      /* Body columns do not share one entity-id vector */
    SIXDOF_ERR_TIME_TRAVEL = -8,          /* elodin-db Error::TimeTravel: a sample older than the pair's last one */
    /* The exception barrier (csrc/abi_guard.hpp): a C++ exception never unwinds into the caller's frames (Rust `extern "C"`,
     * cranelift_exec.rs:11,163-165).  Every allocating entry point catches, leaves the text in *_last_error and returns: */
    SIXDOF_ERR_OUT_OF_MEMORY = -9,        /* std::bad_alloc / std::length_error: a host allocation failed (absurd row count?) */
    SIXDOF_ERR_INTERNAL = -10             /* any other exception: Error::Unknown / PyErr analogue (error.rs:12-51) */
} sixdof_status;

/* integrator/mod.rs:7-10 */
typedef enum sixdof_integrator {
    SIXDOF_INTEGRATOR_RK4 = 0,
    SIXDOF_INTEGRATOR_SEMI_IMPLICIT = 1,
    /* no six_dof stage: a pipe of per-entity systems only (`World.build(system)`, Query::map query.rs:504-545);
     * valid only with a generated program (sixdof_set_custom_pipe); world_accel / force pass through unchanged */
    SIXDOF_INTEGRATOR_NONE = 2
} sixdof_integrator;

/* Arithmetic type of the state columns.  The reference six_dof is f64 only
 * (six_dof.rs:12-14,133-135); F32 is an extension for BASELINE config 5. */
typedef enum sixdof_dtype {
    SIXDOF_F64 = 0,
    SIXDOF_F32 = 1
} sixdof_dtype;

/* impeller2 PrimType subset used by this path */
typedef enum sixdof_prim {
    SIXDOF_PRIM_F64 = 0,
    SIXDOF_PRIM_U64 = 1,
    SIXDOF_PRIM_F32 = 2
} sixdof_prim;

/*
 * Effector pipeline: the `sys` argument of six_dof(), i.e. what runs between
 * clear_forces and calc_accel on every integrator stage (six_dof.rs:184-203).
 * The reference takes arbitrary JAX; this backend takes a list of built-in ops
 * applied in pipe order.  Each op cites the example effector it restates.
 */
typedef enum sixdof_effector_kind {
    /* F += [tau(3), f(3)] = p[0..6), world frame.  test_all.py:342-364 constant_force */
    SIXDOF_EFF_CONST_WRENCH = 1,
    /* f += (p[0],p[1],p[2]) * mass.  examples/ball/sim.py:57-59 */
    SIXDOF_EFF_UNIFORM_GRAVITY = 2,
    /* tau += q * aux[i][0..3)  (body-frame torque column).  examples/apollo-lander/sim.py:396-398 */
    SIXDOF_EFF_BODY_TORQUE = 3,
    /* f += q * aux[i][0..3)  (body-frame force column).  examples/apollo-lander/sim.py:391-394 */
    SIXDOF_EFF_BODY_FORCE = 4,
    /* F = Force(linear = f + drag) with wind = aux[i][0..3), p = {Cd, rho, area}.
     * NOTE: clears the torque, like the reference.  examples/ball/sim.py:96-116 */
    SIXDOF_EFF_BALL_DRAG = 5,
    /* edge_fold over sixdof_set_edges(): acc.f -= p[0]*M*m*r/|r|^3, r = a-b (p[0] = G).
     * Result REPLACES Force for source rows.  examples/three-body/main.py:56-78 */
    SIXDOF_EFF_EDGE_GRAVITY_NEWTON = 6,
    /* edge_fold: acc.f += (p[0]*ma*mb*inv3)*r, r = b-a, inv = 1/sqrt(r.r+p[1]) (p = {K, eps}).
     * examples/n-body/sim.py:344-369 */
    SIXDOF_EFF_EDGE_GRAVITY_SOFTENED = 7,
    /* Same fold as 7 over the complete graph (for i: for j != i, ascending), without
     * materialising edges.  examples/n-body/sim.py:333-337 spawn order */
    SIXDOF_EFF_ALLPAIRS_GRAVITY_SOFTENED = 8,
    /* edge_fold whose fold function is generated code (sixdof_set_custom_pair); never passed to
     * sixdof_set_effectors directly.  graph.rs:177-282 edge_fold with an arbitrary `fn` */
    SIXDOF_EFF_EDGE_CUSTOM = 9,
    /* per-entity ops again (kinds 6..9 are the pair ops): */
    /* tau += aux[i][0..3)  (world-frame torque column): `force + el.SpatialForce(torque=c)` with c a per-entity
     * component, the shape of every effector that applies an externally computed load; it is also what a
     * teacher-forced replay of a recorded `force` column (scripts/ci/baseline/cube-sat-csv) feeds the step */
    SIXDOF_EFF_WORLD_TORQUE = 10,
    /* f += aux[i][0..3)  (world-frame force column).  Same pattern, linear half */
    SIXDOF_EFF_WORLD_FORCE = 11
} sixdof_effector_kind;
/* kinds 6..9 fold over edges / pairs; every other kind is a per-entity op */
#define SIXDOF_EFF_IS_PAIR(k) ((k) >= SIXDOF_EFF_EDGE_GRAVITY_NEWTON && (k) <= SIXDOF_EFF_EDGE_CUSTOM)

typedef struct sixdof_effector_op {
    int32_t kind;               /* sixdof_effector_kind */
    int32_t reserved;
    uint64_t aux_component_id;  /* per-entity [n,3] f64 column for kinds 3,4,5,10,11; else 0 */
    double p[6];
} sixdof_effector_op;

typedef struct sixdof_desc {
    uint32_t struct_size;        /* = sizeof(sixdof_desc) */
    int32_t device_ordinal;      /* HIP device */
    int32_t integrator;          /* sixdof_integrator */
    int32_t dtype;               /* sixdof_dtype */
    uint64_t n_entities;         /* size of the joined Body entity set; 0 = take it from the bound columns */
    double simulation_time_step; /* globals column value (ns-quantised, see sixdof_quantize_time_step) */
    double time_step;            /* six_dof(time_step=...) override, used iff has_time_step */
    int32_t has_time_step;
    uint32_t ticks_per_launch;   /* ticks fused into one kernel launch (>=1). 1 = every tick
                                    materialises all output columns in HBM (reference tick semantics);
                                    K>1 = the reference's ticks_per_telemetry batch, state in registers */
    uint32_t flags;              /* SIXDOF_FLAG_* */
    uint32_t reserved;
} sixdof_desc;

#define SIXDOF_FLAG_USE_GRAPH 1u /* replay sixdof_step batches of >= 4 launches from captured hipGraphs (chains of 32 + one
                                    chain of the remainder, cached per length) */
#define SIXDOF_FLAG_TIME_EACH_LAUNCH 2u /* profiling: bracket every launch of sixdof_step with its own HIP
                                           event pair (<= 4096 launches per call; disables graph replay) */
#define SIXDOF_FLAG_ASYNC_STEP 4u /* sixdof_step only enqueues and returns (no stream sync): pair it with
                                     sixdof_download_async so the telemetry copy of batch i overlaps batch i+1;
                                     kernel_device_ms then reports the previous finished batch.  sixdof_sync joins. */

/* One ECS column as the reference holds it (world.rs:26-30 + ExecSlotMetadata exec.rs:17-22). */
typedef struct sixdof_column {
    uint64_t component_id;       /* sixdof_component_id("world_pos") ... */
    int32_t prim_type;           /* sixdof_prim */
    uint32_t ndim;               /* 0 (scalar), 1 */
    uint64_t dims[2];            /* e.g. {7} */
    uint64_t n_rows;
    const uint64_t* entity_ids;  /* [n_rows] */
    void* host_ptr;              /* caller-owned row-major bytes; borrowed until destroy/rebind */
} sixdof_column;

/* profile.rs TickTimings analogue (ms) */
typedef struct sixdof_timings {
    double h2d_upload_ms;
    double kernel_invoke_ms;   /* host wall time of the step call incl. stream sync */
    double d2h_download_ms;
    double kernel_device_ms;   /* HIP-event time of the launches of the last step, on the handle's stream */
    uint64_t launches;         /* kernel launches issued by the last step */
    uint64_t ticks;
    double kernel_sum_ms;      /* SIXDOF_FLAG_TIME_EACH_LAUNCH: sum of the per-launch event times (no gaps) */
    uint64_t graph_launches;   /* how many of `launches` were replayed from a captured hipGraph (SIXDOF_FLAG_USE_GRAPH) */
} sixdof_timings;

typedef struct sixdof_slot {
    uint64_t component_id;
    uint64_t bytes;
} sixdof_slot;

/* download masks */
#define SIXDOF_COL_WORLD_POS 1u
#define SIXDOF_COL_WORLD_VEL 2u
#define SIXDOF_COL_WORLD_ACCEL 4u
#define SIXDOF_COL_FORCE 8u
#define SIXDOF_COL_INERTIA 16u
#define SIXDOF_COL_ALL 31u

uint32_t sixdof_abi_version(void);
/* FNV-1a-64(name) & ~(1<<63): impeller2/src/types.rs:39-44 */
uint64_t sixdof_component_id(const char* name);
/* Duration::from_secs_f64(1/rate).as_secs_f64(): world_builder.rs:221, world.rs:185-191 */
double sixdof_quantize_time_step(double simulation_rate_hz);
int sixdof_device_count(void);

int sixdof_create(const sixdof_desc* desc, sixdof_handle** out);
void sixdof_destroy(sixdof_handle* h);
const char* sixdof_last_error(const sixdof_handle* h); /* h may be NULL: last create error */

int sixdof_bind_columns(sixdof_handle* h, const sixdof_column* cols, size_t n_cols);
int sixdof_set_effectors(sixdof_handle* h, const sixdof_effector_op* ops, size_t n_ops);
/* Edges as (from entity id, to entity id) in spawn order; resolved to row indices against the
 * bound Body entity ids (query.rs:599-621 gathers by constant u32 indices). */
int sixdof_set_edges(sixdof_handle* h, const uint64_t* from_ids, const uint64_t* to_ids, size_t n_edges);
/* Join tables: row of joined entity j inside column `component_id` — the constant u32 gather indices the
 * reference bakes at compile time (query.rs:599-621).  six_dof runs over the INTERSECTION of the Body columns'
 * entity ids in ascending id order (query.rs:136-208); when every Body column already is that set the order is
 * the columns' own (fast path query.rs:673,702) and rows[j] = j.  rows == NULL just returns the joined count. */
int sixdof_get_join_rows(const sixdof_handle* h, uint64_t component_id, uint32_t* rows, size_t cap, size_t* n_out);
/* Read back the resolved u32 row-index tables (bit-exact integer parity surface). */
int sixdof_get_edge_rows(const sixdof_handle* h, uint32_t* src_rows, uint32_t* dst_rows, size_t cap, size_t* n_out);

int sixdof_upload(sixdof_handle* h);
int sixdof_step(sixdof_handle* h, uint64_t n_ticks, sixdof_timings* timings /* may be NULL */);
/* SIXDOF_FLAG_USE_GRAPH: capture (and cache) the launch chains a later sixdof_step(h, n_ticks) replays, without stepping —
 * the one-off cost CraneliftExec::new pays at build time (cranelift_exec.rs:54-127), kept out of the first timed batch. */
int sixdof_prepare_step(sixdof_handle* h, uint64_t n_ticks);
int sixdof_download(sixdof_handle* h, uint32_t column_mask);
/* Telemetry commit without the per-batch stall (the step either side of the path: commit_world_head after
 * every batch, impeller2_server.rs:390-438; JaxExec blocks on copy_to_host there, jax_exec.rs:150-178).
 * download_async snapshots the selected columns on the compute stream (device-to-device, microseconds) and
 * copies the snapshot into the bound host buffers on a second stream; the host buffers are page-locked on
 * first use (hipHostRegister) so the copy is a real DMA.  The next sixdof_step may be issued immediately.
 * download_wait blocks until the host buffers hold that snapshot.  One snapshot in flight: a second
 * download_async first waits (on the device) for the previous copy to have left the snapshot buffers. */
int sixdof_download_async(sixdof_handle* h, uint32_t column_mask);
int sixdof_download_wait(sixdof_handle* h);
int sixdof_sync(sixdof_handle* h);   /* both streams idle */
int sixdof_get_tick(const sixdof_handle* h, uint64_t* tick);
int sixdof_set_tick(sixdof_handle* h, uint64_t tick);
int sixdof_set_ticks_per_launch(sixdof_handle* h, uint32_t k);
int sixdof_set_flags(sixdof_handle* h, uint32_t flags);
/* Device-resident column (reference byte layout) for zero-copy consumers; NULL if unknown. */
void* sixdof_device_column(sixdof_handle* h, uint64_t component_id);
/* hipStream_t the handle launches on, as an opaque pointer */
void* sixdof_stream(sixdof_handle* h);

/* TickFn-compatible entry (cranelift_exec.rs:11): upload -> 1 tick -> download through the
 * slot tables below.  sixdof_tick_bind selects the handle for the calling thread. */
int sixdof_tick_bind(sixdof_handle* h);
void sixdof_tick(const uint8_t* const* inputs, uint8_t* const* outputs);
/* inputs: order of first use; outputs: ascending ComponentId (system.rs:139-153). */
int sixdof_tick_slots(const sixdof_handle* h, sixdof_slot* inputs, size_t in_cap, size_t* n_in,
                      sixdof_slot* outputs, size_t out_cap, size_t* n_out);

/* ---- health / profiling ------------------------------------------------------------------------------------------
 * Per-rollout failure sentinel (a campaign's replacement for "the sim process exited non-zero",
 * libs/monte-carlo/src/lib.rs:2083-2379): counts joined rows whose world_pos or world_vel holds a NaN/Inf and,
 * if `row_flags` is non-NULL, writes one byte per joined row (1 = non-finite). */
int sixdof_count_nonfinite(sixdof_handle* h, uint64_t* count, uint8_t* row_flags /* [n] or NULL */);
/* Timings of the most recent upload / step / download (profile.rs:14-59 phases). */
int sixdof_last_timings(const sixdof_handle* h, sixdof_timings* out);

/* ---- host-side ECS column store: the `World` of libs/nox-py/src/world.rs:23-45,174-229 ---------------------------
 * Per component a growing row-major buffer + the entity id of each row, rows in spawn order, ids sequential,
 * entity 0 = "Globals" (tick u64, simulation_time_step f64).  Pure host code: needs no GPU. */
typedef struct sixdof_world sixdof_world;
sixdof_world* sixdof_world_create(void);
void sixdof_world_destroy(sixdof_world* w);
const char* sixdof_world_last_error(const sixdof_world* w);
uint64_t sixdof_world_spawn(sixdof_world* w);                 /* World::spawn: next sequential EntityId */
uint64_t sixdof_world_entity_len(const sixdof_world* w);
/* World::insert_with_id for one component: append `row` (n_bytes = prod(dims) * sizeof(prim)) and the entity id. */
int sixdof_world_insert(sixdof_world* w, uint64_t entity, const char* component, int prim_type, const uint64_t* dims,
                        uint32_t ndim, const void* row, size_t n_bytes);
/* World::column_by_id: a view (pointers stay valid until the next insert into that component). */
int sixdof_world_column(sixdof_world* w, uint64_t component_id, sixdof_column* out);
size_t sixdof_world_components(const sixdof_world* w, uint64_t* ids, size_t cap); /* ascending ComponentId */
/* validate_rates + set_globals (world_builder.rs:211-243): telemetry_rate_hz = 0 means "same as simulation". */
int sixdof_world_set_rates(sixdof_world* w, double simulation_rate_hz, double telemetry_rate_hz);
double sixdof_world_time_step(const sixdof_world* w);
uint64_t sixdof_world_ticks_per_telemetry(const sixdof_world* w);
uint64_t sixdof_world_tick(const sixdof_world* w);
void sixdof_world_advance_tick(sixdof_world* w, uint64_t n);
/* Bind every 1-D f64/f32 component column of the world to the backend handle (CraneliftExec::new walks
 * world.column_by_id the same way, cranelift_exec.rs:101-106).  Follow with sixdof_upload. */
int sixdof_bind_world(sixdof_handle* h, sixdof_world* w);

/* ---- the commit path's hand-off: per (entity, component) pair one time series ------------------------------------------
 * What commit_world_head_for_world (libs/nox-py/src/impeller2_server.rs:398-438) writes after every batch and
 * copy_db_to_world (:320-364) reads back before the next: PairId = ComponentId::from_pair(entity, component)
 * (impeller2/src/types.rs:54-59), a series = an index of i64 microsecond timestamps + the samples' bytes
 * (libs/db/src/time_series.rs:201-230: a push older than the last timestamp is SIXDOF_ERR_TIME_TRAVEL).  The database
 * behind it (files, subscriptions, wire protocol) is out of scope; pure host code, needs no GPU. */
typedef struct sixdof_sink sixdof_sink;
uint64_t sixdof_pair_id(const char* entity, const char* component);               /* ComponentId::from_pair */
sixdof_sink* sixdof_sink_create(void);
void sixdof_sink_destroy(sixdof_sink* s);
const char* sixdof_sink_last_error(const sixdof_sink* s);
int sixdof_sink_register(sixdof_sink* s, uint64_t pair_id, uint32_t elem_bytes, const char* name);
int sixdof_sink_push(sixdof_sink* s, uint64_t pair_id, int64_t timestamp_us, const void* buf, uint32_t bytes);   /* TimeSeries::push_buf */
uint64_t sixdof_sink_sample_count(const sixdof_sink* s, uint64_t pair_id);
size_t sixdof_sink_pairs(const sixdof_sink* s, uint64_t* ids, size_t cap);        /* ascending PairId */
int sixdof_sink_latest(const sixdof_sink* s, uint64_t pair_id, int64_t* timestamp_us, void* out, uint32_t bytes);
/* the sample with the greatest timestamp <= timestamp_us (past the last write: the latest; before the first:
 * SIXDOF_ERR_COMPONENT_NOT_FOUND) — StepContext.read_component(timestamp=), elodin.pyi:63-88 */
int sixdof_sink_at(const sixdof_sink* s, uint64_t pair_id, int64_t timestamp_us, int64_t* found_us, void* out, uint32_t bytes);
/* the two append logs of a pair, as the database would persist them (valid until the next push) */
int sixdof_sink_series(const sixdof_sink* s, uint64_t pair_id, const int64_t** timestamps, const uint8_t** data, uint64_t* n,
                       uint32_t* elem_bytes);
void sixdof_sink_truncate(sixdof_sink* s);
/* commit_world_head for one column: row i of `rows` ([n_rows, row_bytes], e.g. a bound host column after sixdof_download /
 * sixdof_download_wait) -> the series of pair_ids[i] (0 or unregistered: skipped, like an entity without metadata). */
int sixdof_sink_commit_rows(sixdof_sink* s, const uint64_t* pair_ids, const void* rows, uint32_t n_rows, uint32_t row_bytes,
                            int64_t timestamp_us);
/* copy_db_to_world for one column: latest samples -> rows; *changed = 1 when a byte differed (follow with sixdof_upload). */
int sixdof_sink_copy_to_rows(const sixdof_sink* s, const uint64_t* pair_ids, void* rows, uint32_t n_rows, uint32_t row_bytes,
                             int* changed);

/* ---- effector front-end: run-time generated pipes -------------------------------------------------------------
 * The reference JIT-compiles whatever effector graph the user wrote (cranelift_compile.rs:13-162).  The analogue
 * here: elodin_amd/codegen.py turns an effector pipe written against a jax.numpy-like tracer into HIP source that
 * instantiates the SAME fused step kernel (csrc/step_kernel.hpp) with the user's code as its effector stage,
 * builds it with hipcc for gfx950 and hands the shared object to this call.  `aux_component_ids` name the bound
 * per-entity columns the generated code uses, in the order it indexes them: first the read-only effector
 * columns (row width 1..3, <= 4), then — for whole programs `pre | six_dof(effectors) | post` — the component
 * columns its systems read AND write (row width 1..64 — a small matrix component is its row-major flattening —, <= 128 columns; fetch them back with sixdof_download_column).
 * A WINDOW column (a wide component such as the rocket example's 480 x 3 sample buffer, examples/rocket/main.py:91-98)
 * is bound like any other ([n, rows*width] elements) but stays in HBM, in the layout the generated object was built for: entity-major
 * (the reference's rows; small executors) or ELEMENT-major — buffer[e * n + entity] for e < rows*width, so that lane-adjacent
 * entities read adjacent addresses; executors of 32,768 entities or more —, used as a ring whose head
 * (physical index of the oldest row) lives in a hidden [n,1] column `<name>#head` — transpose + un-rotate after a download
 * (elodin_amd/exec.py HipExec.component does).  The generated object exports the row width it was built for per column; sixdof_step refuses columns bound
 * with another width.  A program may hold stand-alone folds between its systems (graph.rs:239-361): the generated launch
 * entry then issues a chain of kernels per tick (systems | fold | systems | six_dof | ...) over the same columns; the fold's
 * scratch rows are one more program column `<out>#fold<k>`.  Replaces the built-in op list (sixdof_set_effectors) for the
 * per-entity path.  An object built from a whole-world StableHLO tick with one lane per entity (cranelift_compile.rs:47-68's module,
 * `python -m elodin_amd.stablehlo`, manifest "rows_per_world") exchanges data between the entities of a world inside the
 * wavefront and exports the rows a world occupies; a handle whose n_entities is not a multiple of it is refused with
 * SIXDOF_ERR_INVALID_ARGUMENT. */
int sixdof_set_custom_pipe(sixdof_handle* h, const char* so_path, const uint64_t* aux_component_ids, size_t n_aux);
/* Same idea for GraphQuery.edge_fold (graph.rs:177-282) with a user-written fold function over
 * (acc: Force, a: (WorldPos, Inertia), b: (WorldPos, Inertia)): the generated object instantiates the pair kernels
 * (csrc/pair_kernel.hpp) with that function.  Appends the fold as the LAST op of the pipe set by
 * sixdof_set_effectors (per-entity built-in ops before it are kept); edges come from sixdof_set_edges. */
int sixdof_set_custom_pair(sixdof_handle* h, const char* so_path);

/* ---- telemetry: device-side history ring (the commit step either side of the path) ------------------------
 * The reference commits every output column to its DB after each batch (exec.rs:110-172,
 * impeller2_server.rs:390-438) and `exec.history()` reads it back.  With a ring enabled, sixdof_step writes
 * world_pos / world_vel / world_accel / force of EVERY tick into slot (tick-1) % ring_ticks of a device ring
 * (reference row layout, one contiguous [n,w] block per tick and column) from inside the fused kernel, so
 * ticks_per_launch > 1 no longer drops intermediate ticks.  With a generated program installed
 * (sixdof_set_custom_pipe) every component column of the program is recorded the same way and can be read
 * back by its component id.  Worlds stepped by the pair (edge_fold / all-pairs) kernels or by a rollout model are
 * recorded too: with a ring enabled they run one tick per launch and the four output columns are copied into the
 * tick's slot on the device (the model's own columns are not recorded).  ring_ticks = 0 disables and frees the ring. */
int sixdof_set_history(sixdof_handle* h, uint32_t ring_ticks);
/* Copy the [n,w] block of `component_id` as it was after `tick` ticks into host_dst.  Fails with
 * SIXDOF_ERR_INVALID_ARGUMENT if that tick is not (or no longer) in the ring. */
int sixdof_history_read(sixdof_handle* h, uint64_t component_id, uint64_t tick, void* host_dst);
/* Stream a run of recorded ticks to the host without stalling the stepper: the [n_ticks, n, w] blocks of
 * world_pos / world_vel / world_accel / force for ticks first_tick .. first_tick + n_ticks - 1 are copied, on the
 * copy stream, into host_dst[0..3] (a NULL entry skips that column; buffers are page-locked on first use).  Returns
 * once the copies are enqueued; sixdof_download_wait blocks until they have landed.  With SIXDOF_FLAG_ASYNC_STEP and a
 * ring of at least two batches the copy of batch i overlaps the compute of batch i+1: a later sixdof_step only waits
 * (on the device) when it is about to overwrite ring slots that are still being read.  The host buffers must stay
 * allocated until sixdof_sync, which also releases their page locks. */
int sixdof_history_stream(sixdof_handle* h, uint64_t first_tick, uint64_t n_ticks, void* const host_dst[4]);

/* ---- rollout models: systems piped AROUND six_dof, fused with it (the pipes of examples/<name>/sim.py) ------------- */
struct sixdof_apollo_tables; /* include/sixdof_apollo.h */
/* Select the Apollo-lander rollout model (examples/apollo-lander/sim.py:517-526 + the guidance sidecar
 * controller/src/main.rs).  Needs, besides the Body columns, the columns "apollo_state" [n,16],
 * "apollo_params" [n,17], "apollo_guidance" [n,8], "apollo_score" [n,4], "apollo_result" [n,12] bound
 * with sixdof_bind_columns; integrator must be SEMI_IMPLICIT, dtype F64.  Tables are copied. */
int sixdof_set_model_apollo(sixdof_handle* h, const struct sixdof_apollo_tables* tables);
/* D2H of any bound column by id (model columns are not covered by the sixdof_download mask). */
int sixdof_download_column(sixdof_handle* h, uint64_t component_id);
/* H2D of ONE bound column from its host buffer: how an external write to a component (StepContext.write_component,
 * libs/nox-py/src/step_context.rs; copy_db_to_world's per-component copy, impeller2_server.rs:320-362) reaches the device between
 * batches without re-uploading the columns the host never downloaded.  Needs a prior sixdof_upload. */
int sixdof_upload_column(sixdof_handle* h, uint64_t component_id);

/* ---- campaign collectives: Monte-Carlo rollouts sharded over the GPUs of one node ---------------------------------------
 * The reference runs one OS process per rollout with nothing exchanged between them (libs/monte-carlo/src/lib.rs:2083): the
 * plan row reaches a run through its context file and result.json comes back through the file system.  With rollouts as
 * column rows on several GPUs the two movements are ONE broadcast of the plan table from rank 0 and ONE gather of the
 * result rows, in run-id order (row = idx, run_id = run_%07d, seed = idx + 1: sample.py:149) — RCCL over xGMI, loaded
 * with dlopen (a single-GPU host never needs it).  No exchange per step.  One communicator = one rank = one GPU = one
 * caller thread, like a handle.  The 128-byte id is RCCL's ncclUniqueId: rank 0 makes it, the host ships it to the other
 * ranks by whatever channel it already has (the runner's job file, an env var, a socket). */
typedef struct sixdof_comm sixdof_comm;
#define SIXDOF_COMM_ID_BYTES 128
int sixdof_comm_unique_id(uint8_t id[SIXDOF_COMM_ID_BYTES]);                       /* SIXDOF_ERR_UNSUPPORTED: no RCCL here */
int sixdof_comm_init(sixdof_comm** out, const uint8_t id[SIXDOF_COMM_ID_BYTES], int world, int rank, int device_ordinal);
void sixdof_comm_destroy(sixdof_comm* c);
const char* sixdof_comm_last_error(const sixdof_comm* c);                          /* c may be NULL: last init error */
/* Rank r's contiguous block [lo, hi) of n_rows run ids (blocks differ by at most one row). */
void sixdof_shard_range(uint64_t n_rows, int world, int rank, uint64_t* lo, uint64_t* hi);
/* The gather's packing without a transport (pure host code): equal blocks of sixdof_gather_block_rows(n_total, world) =
 * ceil(n_total / world) rows, rank r's rows first and zeros after; `blocks` = the `world` blocks in rank order.  What
 * sixdof_campaign_gather runs around its one ncclAllGather, and what a host with its own transport runs around its. */
uint64_t sixdof_gather_block_rows(uint64_t n_total, int world);
int sixdof_gather_pack(const double* local_rows, uint64_t n_local, uint64_t width, uint64_t n_total, int world, int rank, double* block);
int sixdof_gather_unpack(const double* blocks, uint64_t width, uint64_t n_total, int world, double* all_rows);
/* `root`'s host buffer of n_bytes (the plan table [n_runs, n_params] f64, a reference profile ...) -> every rank's. */
int sixdof_campaign_broadcast(sixdof_comm* c, void* table, uint64_t n_bytes, int root);
/* Every rank contributes its block's result rows [n_local, width] f64 (n_local = its sixdof_shard_range of n_total);
 * every rank receives all [n_total, width] rows in run-id order. */
int sixdof_campaign_gather(sixdof_comm* c, const double* local_rows, uint64_t n_local, uint64_t width, double* all_rows,
                           uint64_t n_total);

#ifdef __cplusplus
}
#endif
#endif /* SIXDOF_HIP_H */
