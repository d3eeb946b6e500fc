// sixdof_capi.cpp — host side of the C ABI declared in include/sixdof_hip.h.
//
// Mirrors the reference's backend object (CraneliftExec / JaxExec,
// libs/nox-py/src/cranelift_exec.rs:13-195, jax_exec.rs:118-185): it owns the slot tables and
// the device-resident copies of the ECS columns, runs batches of ticks, and copies columns
// back on request.  Differences that are the point of this backend: columns stay resident in
// HBM between batches (the JAX backend re-uploads every input and downloads every output per
// batch), and a batch is one or a few kernel launches instead of a per-tick host loop.
//
// No CPU fallback exists: every entry point that needs the GPU fails with SIXDOF_ERR_NO_DEVICE /
// SIXDOF_ERR_BACKEND when HIP is unavailable.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/sixdof_hip.h"
#include "../../include/sixdof_apollo.h"
#include "kernels.hpp"
#include "abi_guard.hpp"

using namespace sixdof;

namespace {

struct Column {
    uint64_t id = 0;
    int prim = SIXDOF_PRIM_F64;
    uint64_t width = 1;  // elements per row
    uint64_t n_rows = 0;
    size_t elem = 8;
    size_t bytes = 0;
    std::vector<uint64_t> ids;
    void* host = nullptr;  // borrowed
    void* dev = nullptr;   // owned: the full column, reference byte layout
    // join state: rows of this column that belong to the joined entity set, in joined order.  `live` is what
    // the kernels read and write: == dev when the column IS the joined set, else an owned compact [m,w] copy.
    std::vector<uint32_t> rows;
    uint32_t* d_rows = nullptr;
    void* compact = nullptr;
    void* live = nullptr;
    bool joined = false;   // join resolved for the current binding
    void* snap = nullptr;  // owned: device snapshot the async telemetry copy reads (sixdof_download_async)
    bool host_pinned = false;   // host buffer page-locked by us (hipHostRegister)
};

thread_local std::string g_create_error;

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

uint64_t cid(const char* s) { return sixdof_component_id(s); }

}  // namespace

struct sixdof_handle {
    sixdof_desc desc{};
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t evp0 = nullptr, evp1 = nullptr;  // the pair the PREVIOUS asynchronous step recorded (two pairs alternate)
    bool prev_pending = false;
    hipStream_t copy_stream = nullptr;          // telemetry D2H, overlaps the compute stream
    hipEvent_t ev_snap = nullptr, ev_copied = nullptr;
    bool copy_pending = false;
    uint64_t stream_lo = 0, stream_hi = 0;      // ticks of the history run whose copy may still be in flight
    std::vector<void*> pinned_user;             // host buffers page-locked by sixdof_history_stream
    bool step_pending = false;                  // SIXDOF_FLAG_ASYNC_STEP: ev1 of the last step not yet read
    std::vector<hipEvent_t> launch_events;  // SIXDOF_FLAG_TIME_EACH_LAUNCH: 2 per launch
    std::map<uint64_t, Column> cols;  // ascending ComponentId = reference BTreeMap order
    std::vector<sixdof_effector_op> ops;
    // edges
    std::vector<uint32_t> edge_src, edge_dst;       // resolved rows, spawn order
    std::vector<uint32_t> csr_start, csr_dst;       // by source, spawn order kept inside a source
    uint32_t* d_csr_start = nullptr;
    uint32_t* d_csr_dst = nullptr;
    // hub sources of the edge list (out-degree >= kHubDegree): one device block [hub_rows | hub_chunk_start | chunk_e0 | chunk_row]
    uint32_t* d_hub = nullptr;
    double* d_chunk_partial = nullptr;
    uint32_t n_hubs = 0, n_hub_chunks = 0;
    // pair-path scratch
    void* d_scratch = nullptr;
    size_t scratch_bytes = 0;
    std::vector<uint64_t> joined_ids;  // intersection of the Body columns' entity ids (query.rs:136-208)
    bool identity_join = true;         // every Body column already is the joined set (query.rs:673,702 fast path)
    uint64_t tick = 0;
    bool bound = false;
    bool resident = false;             // columns uploaded at least once since the last bind
    sixdof_timings last{};   // most recent upload / step / download
    // run-time generated effector pipe
    void* custom_dl = nullptr;
    CustomLaunchFn custom_launch = nullptr;
    void* pair_dl = nullptr;               // generated edge_fold function (sixdof_set_custom_pair)
    CustomPairLaunchFn pair_launch = nullptr;
    std::vector<uint64_t> custom_aux;      // read-only [n,1..3] columns of a generated effector pipe
    std::vector<uint64_t> custom_model;    // read/write [n,1..16] component columns of a generated program
    bool custom_tick_free = false;          // the generated program never looks at the absolute tick (layout bit 17): replayable
    std::vector<hipStream_t> split_streams;  // side streams of a replay graph split into row-block chains (SIXDOF_GRAPH_SPLIT)
    std::vector<hipEvent_t> split_joins;
    hipEvent_t split_fork = nullptr;
    int pair_only_small = -1;               // what the installed pair object was generated for (-1: both launch shapes)
    unsigned custom_rows_multiple = 1;      // rows a world of the generated program occupies (lane mode): the joined row count must be a multiple
    // telemetry ring
    uint32_t hist_ring = 0;
    uint64_t hist_first_tick = 0;   // first tick (1-based count) recorded since the ring was enabled
    void* d_hist[4] = {nullptr, nullptr, nullptr, nullptr};  // pos, vel, accel, force
    std::vector<void*> d_model_hist;      // one ring per component column of a generated program (same order as custom_model)
    std::vector<unsigned> custom_model_width;   // what the generated code expects per column (0 = unknown), bit 31 = window
    // rollout model (0 = none, 1 = Apollo lander)
    int model = 0;
    std::vector<double> ap_time, ap_alt, ap_rate, ap_pitch, ap_hspeed, ap_downrange;
    bool accel_is_host_data = false;   // set by an upload: the next RK4 launch reads world_accel once (see sixdof_step)
    uint32_t ap_ticks_per_telemetry = 3;
    uint32_t ap_guidance_period = 5;
    uint64_t ap_max_ticks = 0;
    double* d_tick_refs = nullptr;
    size_t tick_refs_cap = 0;
    // graph cache for long batches
    std::map<uint32_t, hipGraphExec_t> graphs;   // replay graphs by chain length (launches per replay), one StepParams signature
    uint32_t graph_k = 0;
    uint64_t graph_sig = 0;
    mutable std::string err;

    uint64_t id_pos, id_vel, id_accel, id_force, id_inertia, id_tick, id_dt;

    int fail(int code, const std::string& msg) const {
        err = msg;
        return code;
    }
    int hip_fail(hipError_t e, const char* what) const {
        err = std::string(what) + ": " + hipGetErrorString(e);
        return SIXDOF_ERR_BACKEND;
    }
    Column* col(uint64_t id) {
        auto it = cols.find(id);
        return it == cols.end() ? nullptr : &it->second;
    }
    const Column* col(uint64_t id) const {
        auto it = cols.find(id);
        return it == cols.end() ? nullptr : &it->second;
    }
    size_t elem_size() const { return desc.dtype == SIXDOF_F32 ? 4 : 8; }
    int state_prim() const { return desc.dtype == SIXDOF_F32 ? SIXDOF_PRIM_F32 : SIXDOF_PRIM_F64; }
    void free_join(Column& c) {
        if (c.d_rows) hipFree(c.d_rows), c.d_rows = nullptr;
        if (c.compact) hipFree(c.compact), c.compact = nullptr;
        c.rows.clear();
        c.live = nullptr;
        c.joined = false;
    }
    void drop_graph() {
        for (auto& kv : graphs) hipGraphExecDestroy(kv.second);
        graphs.clear();
    }
    bool has_pair_op() const {
        for (auto& o : ops)
            if (SIXDOF_EFF_IS_PAIR(o.kind)) return true;
        return false;
    }
};

// where the exception barrier (abi_guard.hpp) leaves its message: the handle's error, or the creation error without a handle
std::string* err_of(const sixdof_handle* h) { return h ? &const_cast<sixdof_handle*>(h)->err : &g_create_error; }

#define HIP_TRY(h, call)                                      \
    do {                                                      \
        hipError_t e_ = (call);                               \
        if (e_ != hipSuccess) return (h)->hip_fail(e_, #call); \
    } while (0)

// Map the joined entity set onto one column: `rows[j]` = row of joined entity j in this column.  Identity ->
// kernels work on the column itself; otherwise on a compact [m,w] copy (gathered on upload, scattered on download).
static int resolve_join(sixdof_handle* h, Column* c) {
    if (c->joined) return SIXDOF_OK;
    const size_t m = h->joined_ids.size();
    if (c->ids == h->joined_ids) {
        c->live = c->dev;
        c->joined = true;
        return SIXDOF_OK;
    }
    std::unordered_map<uint64_t, uint32_t> row_of;
    row_of.reserve(c->ids.size() * 2);
    for (size_t r = 0; r < c->ids.size(); r++) row_of.emplace(c->ids[r], static_cast<uint32_t>(r));
    c->rows.resize(m);
    for (size_t j = 0; j < m; j++) {
        auto it = row_of.find(h->joined_ids[j]);
        if (it == row_of.end())
            return h->fail(SIXDOF_ERR_ENTITY_MISMATCH, "join: a column does not cover the joined Body entity set");
        c->rows[j] = it->second;
    }
    if (m) {
        HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&c->d_rows), m * sizeof(uint32_t)));
        HIP_TRY(h, hipMemcpy(c->d_rows, c->rows.data(), m * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIP_TRY(h, hipMalloc(&c->compact, m * c->width * c->elem));
    }
    c->live = c->compact;
    c->joined = true;
    return SIXDOF_OK;
}

extern "C" {

uint32_t sixdof_abi_version(void) { return SIXDOF_ABI_VERSION; }

// sixdof_component_id / sixdof_quantize_time_step: pure host code, in world.cpp (so the host layer links without HIP: `make asan`)

int sixdof_device_count(void) try {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
} SIXDOF_ABI_CATCH(&g_create_error)

const char* sixdof_last_error(const sixdof_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

void sixdof_destroy(sixdof_handle* h);

int sixdof_create(const sixdof_desc* d, sixdof_handle** out) try {
    if (!d || !out) {
        g_create_error = "sixdof_create: null argument";
        return SIXDOF_ERR_INVALID_ARGUMENT;
    }
    *out = nullptr;
    if (d->struct_size != sizeof(sixdof_desc)) {
        g_create_error = "sixdof_create: struct_size mismatch (ABI version skew)";
        return SIXDOF_ERR_INVALID_ARGUMENT;
    }
    if (d->integrator != SIXDOF_INTEGRATOR_RK4 && d->integrator != SIXDOF_INTEGRATOR_SEMI_IMPLICIT &&
        d->integrator != SIXDOF_INTEGRATOR_NONE) {
        g_create_error = "sixdof_create: unknown integrator";
        return SIXDOF_ERR_INVALID_ARGUMENT;
    }
    if (d->dtype != SIXDOF_F64 && d->dtype != SIXDOF_F32) {
        g_create_error = "sixdof_create: unknown dtype";
        return SIXDOF_ERR_INVALID_ARGUMENT;
    }
    if (d->n_entities > 0xFFFFFFF0ull) {
        g_create_error = "sixdof_create: n_entities exceeds u32 row index range";
        return SIXDOF_ERR_INVALID_ARGUMENT;
    }
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev == 0) {
        g_create_error = std::string("sixdof_create: no HIP device (") + hipGetErrorString(e) +
                         "); this backend has no CPU fallback";
        return SIXDOF_ERR_NO_DEVICE;
    }
    if (d->device_ordinal < 0 || d->device_ordinal >= n_dev) {
        g_create_error = "sixdof_create: device_ordinal out of range";
        return SIXDOF_ERR_INVALID_ARGUMENT;
    }
    auto* h = new sixdof_handle();
    h->desc = *d;
    if (h->desc.ticks_per_launch == 0) h->desc.ticks_per_launch = 1;
    h->device = d->device_ordinal;
    h->id_pos = cid("world_pos");
    h->id_vel = cid("world_vel");
    h->id_accel = cid("world_accel");
    h->id_force = cid("force");
    h->id_inertia = cid("inertia");
    h->id_tick = cid("tick");
    h->id_dt = cid("simulation_time_step");
    if ((e = hipSetDevice(h->device)) != hipSuccess || (e = hipStreamCreate(&h->stream)) != hipSuccess ||
        (e = hipEventCreate(&h->ev0)) != hipSuccess || (e = hipEventCreate(&h->ev1)) != hipSuccess ||
        (e = hipEventCreate(&h->evp0)) != hipSuccess || (e = hipEventCreate(&h->evp1)) != hipSuccess) {
        g_create_error = std::string("sixdof_create: ") + hipGetErrorString(e);
        sixdof_destroy(h);   // releases whatever part was created
        return SIXDOF_ERR_BACKEND;
    }
    *out = h;
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(&g_create_error)

void sixdof_destroy(sixdof_handle* h) try {
    if (!h) return;
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    h->drop_graph();
    if (h->copy_stream) hipStreamSynchronize(h->copy_stream);
    for (auto& kv : h->cols) {
        h->free_join(kv.second);
        if (kv.second.dev) hipFree(kv.second.dev);
        if (kv.second.snap) hipFree(kv.second.snap);
        if (kv.second.host_pinned) hipHostUnregister(kv.second.host);
    }
    for (void* p : h->pinned_user) (void)hipHostUnregister(p);
    if (h->copy_stream) hipStreamDestroy(h->copy_stream);
    for (auto st : h->split_streams) hipStreamDestroy(st);
    for (auto ev : h->split_joins) hipEventDestroy(ev);
    if (h->split_fork) hipEventDestroy(h->split_fork);
    if (h->ev_snap) hipEventDestroy(h->ev_snap);
    if (h->ev_copied) hipEventDestroy(h->ev_copied);
    if (h->d_csr_start) hipFree(h->d_csr_start);
    if (h->d_csr_dst) hipFree(h->d_csr_dst);
    if (h->d_hub) hipFree(h->d_hub);
    if (h->d_chunk_partial) hipFree(h->d_chunk_partial);
    if (h->d_scratch) hipFree(h->d_scratch);
    if (h->d_tick_refs) hipFree(h->d_tick_refs);
    for (void* p : h->d_hist) if (p) hipFree(p);
    for (void* p : h->d_model_hist) if (p) hipFree(p);
    if (h->custom_dl) dlclose(h->custom_dl);
    if (h->pair_dl) dlclose(h->pair_dl);
    for (hipEvent_t e : h->launch_events) hipEventDestroy(e);
    if (h->ev0) hipEventDestroy(h->ev0);
    if (h->ev1) hipEventDestroy(h->ev1);
    if (h->evp0) hipEventDestroy(h->evp0);
    if (h->evp1) hipEventDestroy(h->evp1);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
} SIXDOF_ABI_CATCH_VALUE(err_of(h), )

int sixdof_bind_columns(sixdof_handle* h, const sixdof_column* cols, size_t n_cols) try {
    if (!h || (!cols && n_cols)) return SIXDOF_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    h->drop_graph();
    for (size_t i = 0; i < n_cols; i++) {
        const sixdof_column& c = cols[i];
        if (c.ndim > 1) return h->fail(SIXDOF_ERR_UNSUPPORTED, "bind_columns: only scalar / 1-D components");
        if (!c.host_ptr && c.n_rows) return h->fail(SIXDOF_ERR_INVALID_ARGUMENT, "bind_columns: null host_ptr");
        Column col;
        col.id = c.component_id;
        col.prim = c.prim_type;
        col.width = c.ndim == 0 ? 1 : c.dims[0];
        col.n_rows = c.n_rows;
        col.elem = c.prim_type == SIXDOF_PRIM_F32 ? 4 : 8;
        col.bytes = static_cast<size_t>(col.width * col.n_rows) * col.elem;
        col.host = c.host_ptr;
        if (c.entity_ids) col.ids.assign(c.entity_ids, c.entity_ids + c.n_rows);
        Column* old = h->col(c.component_id);
        if (old) {
            h->free_join(*old);
            if (old->dev) {
                if (old->bytes == col.bytes) col.dev = old->dev;
                else hipFree(old->dev);
            }
            // async-commit state of the previous binding: the snapshot is sized per binding, the page lock belongs to
            // the previous host buffer
            if (h->copy_stream) (void)hipStreamSynchronize(h->copy_stream);
            if (old->snap) hipFree(old->snap);
            if (old->host_pinned) (void)hipHostUnregister(old->host);
        }
        if (!col.dev && col.bytes) HIP_TRY(h, hipMalloc(&col.dev, col.bytes));
        h->cols[c.component_id] = std::move(col);
    }
    // The Body archetype (six_dof.rs:152-159): five columns.  six_dof's queries run over the INTERSECTION of
    // their entity ids in ascending id order (query.rs:136-208); rows outside it are never touched.
    const struct { uint64_t id; uint64_t width; const char* name; } body[5] = {
        {h->id_pos, 7, "world_pos"}, {h->id_vel, 6, "world_vel"}, {h->id_accel, 6, "world_accel"},
        {h->id_force, 6, "force"},   {h->id_inertia, 7, "inertia"}};
    const Column* first = nullptr;
    bool identical = true;
    for (auto& b : body) {
        const Column* c = h->col(b.id);
        if (!c) return h->fail(SIXDOF_ERR_COMPONENT_NOT_FOUND, std::string("bind_columns: missing Body column ") + b.name);
        if (c->prim != h->state_prim())
            return h->fail(SIXDOF_ERR_VALUE_SIZE_MISMATCH, std::string("bind_columns: dtype mismatch on ") + b.name);
        if (c->width != b.width)
            return h->fail(SIXDOF_ERR_VALUE_SIZE_MISMATCH, std::string("bind_columns: shape mismatch on ") + b.name);
        if (c->ids.size() != c->n_rows)
            return h->fail(SIXDOF_ERR_INVALID_ARGUMENT, std::string("bind_columns: entity_ids missing on ") + b.name);
        if (!first) first = c;
        else if (c->ids != first->ids) identical = false;
    }
    if (identical) {
        h->joined_ids = first->ids;   // fast path: column order as is (query.rs:673,702)
    } else {
        std::vector<uint64_t> acc(first->ids);
        std::sort(acc.begin(), acc.end());
        for (auto& b : body) {
            std::vector<uint64_t> ids(h->col(b.id)->ids), out;
            std::sort(ids.begin(), ids.end());
            std::set_intersection(acc.begin(), acc.end(), ids.begin(), ids.end(), std::back_inserter(out));
            acc.swap(out);
        }
        h->joined_ids.swap(acc);      // ascending entity id
    }
    h->identity_join = identical;
    if (h->desc.n_entities == 0) h->desc.n_entities = h->joined_ids.size();
    // a generated program installed BEFORE the join was sized (n_entities == 0 then: its whole-worlds check passed on nothing)
    if (h->custom_rows_multiple > 1 && h->desc.n_entities % h->custom_rows_multiple != 0)
        return h->fail(SIXDOF_ERR_INVALID_ARGUMENT, "bind_columns: the installed program lays a world out as " + std::to_string(h->custom_rows_multiple) +
                       " consecutive rows; the joined " + std::to_string(h->desc.n_entities) + " rows are not a whole number of worlds");
    if (h->joined_ids.size() != h->desc.n_entities)
        return h->fail(SIXDOF_ERR_VALUE_SIZE_MISMATCH, "bind_columns: n_entities does not match the joined Body entity set");
    for (auto& kv : h->cols) h->free_join(kv.second);
    for (auto& b : body) {
        int rc = resolve_join(h, h->col(b.id));
        if (rc != SIXDOF_OK) return rc;
    }
    h->bound = true;
    h->resident = false;
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(h))

int sixdof_bind_world(sixdof_handle* h, sixdof_world* w) try {
    if (!h || !w) return SIXDOF_ERR_INVALID_ARGUMENT;
    std::vector<uint64_t> ids(sixdof_world_components(w, nullptr, 0));
    sixdof_world_components(w, ids.data(), ids.size());
    std::vector<sixdof_column> cols;
    for (uint64_t id : ids) {
        sixdof_column c{};
        if (sixdof_world_column(w, id, &c) != SIXDOF_OK) continue;
        if (id == h->id_tick || id == h->id_dt) continue;   // globals travel in the descriptor / handle
        if (c.prim_type != h->state_prim() || c.ndim > 1) continue;
        cols.push_back(c);
    }
    h->desc.simulation_time_step = sixdof_world_time_step(w);
    h->tick = sixdof_world_tick(w);
    return sixdof_bind_columns(h, cols.data(), cols.size());
} SIXDOF_ABI_CATCH(err_of(h))

int sixdof_set_effectors(sixdof_handle* h, const sixdof_effector_op* ops, size_t n_ops) try {
    if (!h || (!ops && n_ops)) return SIXDOF_ERR_INVALID_ARGUMENT;
    size_t n_entity_ops = 0, n_pair = 0;
    for (size_t i = 0; i < n_ops; i++) {
        const int k = ops[i].kind;
        if (k < SIXDOF_EFF_CONST_WRENCH || k > SIXDOF_EFF_WORLD_FORCE || k == SIXDOF_EFF_EDGE_CUSTOM)
            return h->fail(SIXDOF_ERR_INVALID_ARGUMENT, "set_effectors: unknown effector kind");
        if (SIXDOF_EFF_IS_PAIR(k)) {
            n_pair++;
            if (i + 1 != n_ops)
                return h->fail(SIXDOF_ERR_UNSUPPORTED, "set_effectors: a pair (edge_fold) op must be last in the pipe");
        } else {
            n_entity_ops++;
        }
    }
    if (n_entity_ops > static_cast<size_t>(kMaxOps))
        return h->fail(SIXDOF_ERR_UNSUPPORTED, "set_effectors: at most 4 per-entity ops");
    if (n_pair > 1) return h->fail(SIXDOF_ERR_UNSUPPORTED, "set_effectors: at most one pair op");
    h->ops.assign(ops, ops + n_ops);
    h->drop_graph();
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(h))

int sixdof_set_edges(sixdof_handle* h, const uint64_t* from_ids, const uint64_t* to_ids, size_t n_edges) try {
    if (!h || ((!from_ids || !to_ids) && n_edges)) return SIXDOF_ERR_INVALID_ARGUMENT;
    if (!h->bound) return h->fail(SIXDOF_ERR_COMPONENT_NOT_FOUND, "set_edges: bind Body columns first");
    HIP_TRY(h, hipSetDevice(h->device));
    std::unordered_map<uint64_t, uint32_t> row_of;   // entity id -> row of the joined Body set
    row_of.reserve(h->joined_ids.size() * 2);
    for (size_t r = 0; r < h->joined_ids.size(); r++) row_of.emplace(h->joined_ids[r], static_cast<uint32_t>(r));
    std::vector<uint32_t> src(n_edges), dst(n_edges);
    for (size_t e = 0; e < n_edges; e++) {
        auto a = row_of.find(from_ids[e]), b = row_of.find(to_ids[e]);
        if (a == row_of.end() || b == row_of.end())
            return h->fail(SIXDOF_ERR_COMPONENT_NOT_FOUND, "set_edges: edge endpoint is not a Body entity");
        src[e] = a->second;
        dst[e] = b->second;
    }
    const uint32_t n = static_cast<uint32_t>(h->desc.n_entities);
    // CSR by source; a stable counting sort keeps each source's out-edges in spawn order, which is
    // the fold order of GraphQuery::edge_fold (graph.rs:113-175,239-361)
    std::vector<uint32_t> start(n + 1, 0), cdst(n_edges);
    for (size_t e = 0; e < n_edges; e++) start[src[e] + 1]++;
    for (uint32_t i = 0; i < n; i++) start[i + 1] += start[i];
    std::vector<uint32_t> cursor(start.begin(), start.end() - 1);
    for (size_t e = 0; e < n_edges; e++) cdst[cursor[src[e]]++] = dst[e];
    if (h->d_csr_start) hipFree(h->d_csr_start), h->d_csr_start = nullptr;
    if (h->d_csr_dst) hipFree(h->d_csr_dst), h->d_csr_dst = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&h->d_csr_start), (n + 1) * sizeof(uint32_t)));
    HIP_TRY(h, hipMemcpy(h->d_csr_start, start.data(), (n + 1) * sizeof(uint32_t), hipMemcpyHostToDevice));
    if (n_edges) {
        HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&h->d_csr_dst), n_edges * sizeof(uint32_t)));
        HIP_TRY(h, hipMemcpy(h->d_csr_dst, cdst.data(), n_edges * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    // hub sources: the fold kernels give them whole waves (pair_kernel.hpp 2c)
    if (h->d_hub) hipFree(h->d_hub), h->d_hub = nullptr;
    if (h->d_chunk_partial) hipFree(h->d_chunk_partial), h->d_chunk_partial = nullptr;
    std::vector<uint32_t> hub_rows, hub_chunk_start{0}, chunk_e0, chunk_row;
    const char* no_hubs = std::getenv("SIXDOF_NO_HUBS");   // A/B knob: "1" folds every source with one lane
    for (uint32_t i = 0; i < n && !(no_hubs && no_hubs[0] == '1'); i++) {
        const uint32_t deg = start[i + 1] - start[i];
        if (deg < kHubDegree) continue;
        hub_rows.push_back(i);
        for (uint32_t e = start[i]; e < start[i + 1]; e += kHubChunk) {
            chunk_e0.push_back(e);
            chunk_row.push_back(i);
        }
        hub_chunk_start.push_back(static_cast<uint32_t>(chunk_e0.size()));
    }
    h->n_hubs = static_cast<uint32_t>(hub_rows.size());
    h->n_hub_chunks = static_cast<uint32_t>(chunk_e0.size());
    if (h->n_hubs) {
        std::vector<uint32_t> blob;
        blob.insert(blob.end(), hub_rows.begin(), hub_rows.end());
        blob.insert(blob.end(), hub_chunk_start.begin(), hub_chunk_start.end());
        blob.insert(blob.end(), chunk_e0.begin(), chunk_e0.end());
        blob.insert(blob.end(), chunk_row.begin(), chunk_row.end());
        HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&h->d_hub), blob.size() * sizeof(uint32_t)));
        HIP_TRY(h, hipMemcpy(h->d_hub, blob.data(), blob.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&h->d_chunk_partial), static_cast<size_t>(h->n_hub_chunks) * kPartialWidth * sizeof(double)));
    }
    h->edge_src.swap(src);
    h->edge_dst.swap(dst);
    h->csr_start.swap(start);
    h->csr_dst.swap(cdst);
    h->drop_graph();
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(h))

int sixdof_get_join_rows(const sixdof_handle* h, uint64_t component_id, uint32_t* rows, size_t cap, size_t* n_out) try {
    if (!h || !n_out) return SIXDOF_ERR_INVALID_ARGUMENT;
    const Column* c = h->col(component_id);
    if (!c) return h->fail(SIXDOF_ERR_COMPONENT_NOT_FOUND, "get_join_rows: unknown component");
    const size_t m = h->joined_ids.size();
    *n_out = m;
    if (!rows) return SIXDOF_OK;
    if (cap < m) return h->fail(SIXDOF_ERR_VALUE_SIZE_MISMATCH, "get_join_rows: buffer too small");
    if (!c->joined) return h->fail(SIXDOF_ERR_INVALID_ARGUMENT, "get_join_rows: column is not part of the join yet");
    for (size_t j = 0; j < m; j++) rows[j] = c->rows.empty() ? static_cast<uint32_t>(j) : c->rows[j];
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(h))

int sixdof_get_edge_rows(const sixdof_handle* h, uint32_t* src_rows, uint32_t* dst_rows, size_t cap, size_t* n_out) try {
    if (!h || !n_out) return SIXDOF_ERR_INVALID_ARGUMENT;
    *n_out = h->edge_src.size();
    if (cap < h->edge_src.size()) return h->fail(SIXDOF_ERR_VALUE_SIZE_MISMATCH, "get_edge_rows: buffer too small");
    if (src_rows) std::memcpy(src_rows, h->edge_src.data(), h->edge_src.size() * sizeof(uint32_t));
    if (dst_rows) std::memcpy(dst_rows, h->edge_dst.data(), h->edge_dst.size() * sizeof(uint32_t));
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(h))

int prepare_graph(sixdof_handle* h);

int sixdof_upload(sixdof_handle* h) try {
    if (!h) return SIXDOF_ERR_INVALID_ARGUMENT;
    if (!h->bound) return h->fail(SIXDOF_ERR_COMPONENT_NOT_FOUND, "upload: no columns bound");
    HIP_TRY(h, hipSetDevice(h->device));
    const double t_up = now_ms();
    for (auto& kv : h->cols) {
        Column& c = kv.second;
        if (c.bytes) HIP_TRY(h, hipMemcpyAsync(c.dev, c.host, c.bytes, hipMemcpyHostToDevice, h->stream));
        if (c.joined && c.compact) {
            hipError_t e = launch_gather_rows(c.compact, c.dev, c.d_rows, static_cast<uint32_t>(h->joined_ids.size()),
                                              static_cast<uint32_t>(c.width), c.elem, h->stream);
            if (e != hipSuccess) return h->hip_fail(e, "gather_rows");
        }
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->last.h2d_upload_ms = now_ms() - t_up;
    h->resident = true;
    h->accel_is_host_data = true;
    return prepare_graph(h);   // SIXDOF_FLAG_USE_GRAPH: capture now, not inside the first long step call
} SIXDOF_ABI_CATCH(err_of(h))

// joined rows -> their places in the full column (before any D2H of that column)
static int scatter_back(sixdof_handle* h, Column* c) {
    if (c->joined && c->compact) {
        hipError_t e = launch_scatter_rows(c->dev, c->compact, c->d_rows, static_cast<uint32_t>(h->joined_ids.size()),
                                           static_cast<uint32_t>(c->width), c->elem, h->stream);
        if (e != hipSuccess) return h->hip_fail(e, "scatter_rows");
    }
    return SIXDOF_OK;
}

int sixdof_download(sixdof_handle* h, uint32_t mask) try {
    if (!h) return SIXDOF_ERR_INVALID_ARGUMENT;
    if (!h->bound) return h->fail(SIXDOF_ERR_COMPONENT_NOT_FOUND, "download: no columns bound");
    HIP_TRY(h, hipSetDevice(h->device));
    const double t_dn = now_ms();
    const struct { uint32_t bit; uint64_t id; } sel[5] = {{SIXDOF_COL_WORLD_POS, h->id_pos},
                                                        {SIXDOF_COL_WORLD_VEL, h->id_vel},
                                                        {SIXDOF_COL_WORLD_ACCEL, h->id_accel},
                                                        {SIXDOF_COL_FORCE, h->id_force},
                                                        {SIXDOF_COL_INERTIA, h->id_inertia}};
    for (auto& s : sel) {
        if (!(mask & s.bit)) continue;
        Column* c = h->col(s.id);
        if (!c) continue;
        int rc = scatter_back(h, c);
        if (rc != SIXDOF_OK) return rc;
        if (c->bytes) HIP_TRY(h, hipMemcpyAsync(c->host, c->dev, c->bytes, hipMemcpyDeviceToHost, h->stream));
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->last.d2h_download_ms = now_ms() - t_dn;
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(h))

int sixdof_download_async(sixdof_handle* h, uint32_t mask) try {
    if (!h) return SIXDOF_ERR_INVALID_ARGUMENT;
    if (!h->bound) return h->fail(SIXDOF_ERR_COMPONENT_NOT_FOUND, "download_async: no columns bound");
    HIP_TRY(h, hipSetDevice(h->device));
    if (!h->copy_stream) {
        HIP_TRY(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
        HIP_TRY(h, hipEventCreateWithFlags(&h->ev_snap, hipEventDisableTiming));
        HIP_TRY(h, hipEventCreateWithFlags(&h->ev_copied, hipEventDisableTiming));
    }
    const struct { uint32_t bit; uint64_t id; } sel[5] = {{SIXDOF_COL_WORLD_POS, h->id_pos},
                                                        {SIXDOF_COL_WORLD_VEL, h->id_vel},
                                                        {SIXDOF_COL_WORLD_ACCEL, h->id_accel},
                                                        {SIXDOF_COL_FORCE, h->id_force},
                                                        {SIXDOF_COL_INERTIA, h->id_inertia}};
    // the previous copy must have drained the snapshot buffers before they are overwritten (device-side wait)
    if (h->copy_pending) HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_copied, 0));
    Column* picked[5];
    int n_picked = 0;
    for (auto& s : sel) {
        if (!(mask & s.bit)) continue;
        Column* c = h->col(s.id);
        if (!c || !c->bytes) continue;
        int rc = scatter_back(h, c);
        if (rc != SIXDOF_OK) return rc;
        if (!c->snap) HIP_TRY(h, hipMalloc(&c->snap, c->bytes));
        HIP_TRY(h, hipMemcpyAsync(c->snap, c->dev, c->bytes, hipMemcpyDeviceToDevice, h->stream));
        if (!c->host_pinned) {   // page-lock once; if the range cannot be locked the copy below still works, staged
            if (hipHostRegister(c->host, c->bytes, hipHostRegisterDefault) == hipSuccess) c->host_pinned = true;
            else (void)hipGetLastError();
        }
        picked[n_picked++] = c;
    }
    HIP_TRY(h, hipEventRecord(h->ev_snap, h->stream));
    HIP_TRY(h, hipStreamWaitEvent(h->copy_stream, h->ev_snap, 0));
    for (int k = 0; k < n_picked; k++)
        HIP_TRY(h, hipMemcpyAsync(picked[k]->host, picked[k]->snap, picked[k]->bytes, hipMemcpyDeviceToHost, h->copy_stream));
    HIP_TRY(h, hipEventRecord(h->ev_copied, h->copy_stream));
    h->copy_pending = true;
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(h))

int sixdof_download_wait(sixdof_handle* h) try {
    if (!h) return SIXDOF_ERR_INVALID_ARGUMENT;
    if (!h->copy_pending) return SIXDOF_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    const double t0 = now_ms();
    HIP_TRY(h, hipEventSynchronize(h->ev_copied));
    h->last.d2h_download_ms = now_ms() - t0;     // the part of the copy the host actually waited for
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(h))

int sixdof_sync(sixdof_handle* h) try {
    if (!h) return SIXDOF_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->copy_stream) HIP_TRY(h, hipStreamSynchronize(h->copy_stream));
    float ms0 = 0.f;
    if (h->step_pending && hipEventElapsedTime(&ms0, h->ev0, h->ev1) == hipSuccess) h->last.kernel_device_ms = ms0;
    h->step_pending = h->prev_pending = false;
    h->copy_pending = false;
    h->stream_lo = 1, h->stream_hi = 0;
    // page locks taken on the caller's history buffers end here: the caller may free them after sixdof_sync
    for (void* p : h->pinned_user) (void)hipHostUnregister(p);
    h->pinned_user.clear();
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(h))

int sixdof_get_tick(const sixdof_handle* h, uint64_t* tick) try {
    if (!h || !tick) return SIXDOF_ERR_INVALID_ARGUMENT;
    *tick = h->tick;
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(h))
int sixdof_set_tick(sixdof_handle* h, uint64_t tick) try {
    if (!h) return SIXDOF_ERR_INVALID_ARGUMENT;
    h->tick = tick;
    h->hist_first_tick = tick + 1;
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(h))
int sixdof_set_ticks_per_launch(sixdof_handle* h, uint32_t k) try {
    if (!h || k == 0) return SIXDOF_ERR_INVALID_ARGUMENT;
    h->desc.ticks_per_launch = k;
    h->drop_graph();
    return h->resident ? prepare_graph(h) : SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(h))

int sixdof_set_flags(sixdof_handle* h, uint32_t flags) try {
    if (!h) return SIXDOF_ERR_INVALID_ARGUMENT;
    h->desc.flags = flags;
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(h))

void* sixdof_device_column(sixdof_handle* h, uint64_t component_id) try {
    if (!h) return nullptr;
    Column* c = h->col(component_id);
    return c ? (c->live ? c->live : c->dev) : nullptr;
} SIXDOF_ABI_CATCH_VALUE(err_of(h), nullptr)
void* sixdof_stream(sixdof_handle* h) { return h ? static_cast<void*>(h->stream) : nullptr; }

}  // extern "C"

// ---- step --------------------------------------------------------------------------------------------------

namespace {

int build_dev_ops(sixdof_handle* h, DevOp* out, uint32_t* n_out, uint32_t* vel_independent) {
    uint32_t n = 0;
    *vel_independent = 1;
    if (h->custom_launch) {   // generated pipe: slots are just the columns its code reads
        for (uint64_t id : h->custom_aux) {
            Column* c = h->col(id);
            if (!c) return h->fail(SIXDOF_ERR_COMPONENT_NOT_FOUND, "step: column read by the generated pipe is not bound");
            if (c->width < 1 || c->width > 3 || c->prim != h->state_prim())
                return h->fail(SIXDOF_ERR_VALUE_SIZE_MISMATCH, "step: generated-pipe columns must be [n,1..3] of the state dtype");
            if (!c->joined) {
                int rc = resolve_join(h, c);
                if (rc != SIXDOF_OK) return rc;
                if (c->compact) {
                    hipError_t e = launch_gather_rows(c->compact, c->dev, c->d_rows, static_cast<uint32_t>(h->joined_ids.size()),
                                                      static_cast<uint32_t>(c->width), c->elem, h->stream);
                    if (e != hipSuccess) return h->hip_fail(e, "gather_rows");
                }
            }
            DevOp d{};
            d.aux = c->live;
            d.aux_width = static_cast<int32_t>(c->width);
            out[n++] = d;
        }
        *n_out = n;
        return SIXDOF_OK;
    }
    for (auto& o : h->ops) {
        if (SIXDOF_EFF_IS_PAIR(o.kind)) continue;
        DevOp d{};
        d.kind = o.kind;
        std::memcpy(d.p, o.p, sizeof(d.p));
        if (o.kind == SIXDOF_EFF_BODY_TORQUE || o.kind == SIXDOF_EFF_BODY_FORCE || o.kind == SIXDOF_EFF_BALL_DRAG ||
            o.kind == SIXDOF_EFF_WORLD_TORQUE || o.kind == SIXDOF_EFF_WORLD_FORCE) {
            Column* c = h->col(o.aux_component_id);
            if (!c) return h->fail(SIXDOF_ERR_COMPONENT_NOT_FOUND, "step: effector aux column not bound");
            if (c->width != 3 || c->prim != h->state_prim())
                return h->fail(SIXDOF_ERR_VALUE_SIZE_MISMATCH, "step: effector aux column must be [n,3] of the state dtype");
            if (!c->joined) {   // first use after binding: join it onto the Body set and bring its rows over
                int rc = resolve_join(h, c);
                if (rc != SIXDOF_OK) return rc;
                if (c->compact) {
                    hipError_t e = launch_gather_rows(c->compact, c->dev, c->d_rows, static_cast<uint32_t>(h->joined_ids.size()), 3, c->elem, h->stream);
                    if (e != hipSuccess) return h->hip_fail(e, "gather_rows");
                }
            }
            d.aux = c->live;
        }
        if (o.kind == SIXDOF_EFF_BALL_DRAG) *vel_independent = 0;
        out[n++] = d;
    }
    *n_out = n;
    return SIXDOF_OK;
}

int fill_step_params(sixdof_handle* h, StepParams* P) {
    std::memset(P, 0, sizeof(*P));
    P->pos = h->col(h->id_pos)->live;
    P->vel = h->col(h->id_vel)->live;
    P->accel = h->col(h->id_accel)->live;
    P->force = h->col(h->id_force)->live;
    P->inertia = h->col(h->id_inertia)->live;
    P->n = static_cast<uint32_t>(h->desc.n_entities);
    P->dt_g = h->desc.simulation_time_step;
    P->dt = h->desc.has_time_step ? h->desc.time_step : h->desc.simulation_time_step;
    // Cache policy of the launch (step_kernel.hpp: load * 8 + store, bit 8 = one flush after the tick instead of columns
    // stored as they complete), by working-set size, from the A/B matrices in profiles/r02_step_ab_load_x_store_policy.txt
    // and r02_step_ab_policy_by_size.txt (f64 body counts in brackets):
    //   <= 768 MiB [<= 3.1M]      plain loads, nt stores, early     65,536: 5.48 -> 4.93 us   262,144: 16.6 -> 16.3 us
    //                                                               1,048,576: 61.9 -> 60.0   2,097,152: 136 -> 123 us
    //   beyond                    nt loads,    nt stores, early     4,194,304: 300 -> 263 us
    // (inputs are re-read next tick: keep them cacheable while the 256 MiB Infinity Cache can hold them; outputs are
    // written once per tick: never worth a line).  SIXDOF_STREAMING=<code> overrides it for A/B runs (tools/step_ab.py).
    const char* force_nt = std::getenv("SIXDOF_STREAMING");
    size_t row_elems = 32;      // the Body columns: pos 7 + vel 6 + accel 6 + force 6 + inertia 7
    for (size_t k = 0; k < h->custom_model.size(); k++)      // + the component columns of a generated program (Falcon 9: 201 values)
        if (const Column* c = h->col(h->custom_model[k])) row_elems += c->width;
    const size_t state_bytes = static_cast<size_t>(h->desc.n_entities) * row_elems * h->elem_size();
    // NOT the write-through (`sc1`) store policy, although it measured 7 % faster between 48 and 192 MiB of state: the next
    // launch can read STALE rows after it (262,144 bodies: ~10 % of the rows differ from the fused run, differently every
    // run — tools/debug_midsize_determinism.py, profiles/r02_sc1_store_policy_is_unsafe.txt).  It is compiled into the A/B
    // library only (`make ab`, -DSIXDOF_AB_BUILD); the product ignores any SIXDOF_STREAMING code outside {0, 1, 9}.
    uint32_t policy = 9u;
    if (state_bytes <= (768ull << 20)) policy = 1u;
    if (force_nt) {
        const uint32_t code = static_cast<uint32_t>(std::atoi(force_nt));
#ifdef SIXDOF_AB_BUILD
        policy = code;
#else
        // the product library carries the three safe policies only (plain / nt stores / nt both ways, + the late-flush bit)
        const uint32_t pol = code & 255u;
        if ((code & ~0x1ffu) == 0 && (pol == 0u || pol == 1u || pol == 9u)) policy = code;
#endif
    }
    P->streaming = policy;
    P->hist_ring = h->hist_ring;
    if (h->hist_ring) {
        P->hist_pos = h->d_hist[0];
        P->hist_vel = h->d_hist[1];
        P->hist_accel = h->d_hist[2];
        P->hist_force = h->d_hist[3];
    }
    for (size_t k = 0; k < h->custom_model.size(); k++) {
        Column* c = h->col(h->custom_model[k]);
        if (!c) return h->fail(SIXDOF_ERR_COMPONENT_NOT_FOUND, "step: component column of the generated program is not bound");
        const unsigned expect = k < h->custom_model_width.size() ? h->custom_model_width[k] : 0u;
        const bool window = (expect >> 31) != 0;
        const size_t want = expect & 0x1fffffffu;     // bit 30: built for the element-major window layout; bit 29: element-major register columns
        if (c->prim != h->state_prim() || c->width < 1 || (!window && c->width > 64) || (want && c->width != want))
            return h->fail(SIXDOF_ERR_VALUE_SIZE_MISMATCH,
                           "step: program columns must be of the state dtype and as wide as the generated code expects "
                           "([n,1..64]; a window column [n, rows*width])");
        if (!c->joined) {
            int rc = resolve_join(h, c);
            if (rc != SIXDOF_OK) return rc;
            if (c->compact) {
                if ((expect >> 29) & 1u)      // the gather / scatter kernels move [n,w] rows
                    return h->fail(SIXDOF_ERR_UNSUPPORTED, "step: a program built for element-major columns needs every column on "
                                                           "the executor's own entity set (no entity-set join)");
                hipError_t e = launch_gather_rows(c->compact, c->dev, c->d_rows, static_cast<uint32_t>(h->joined_ids.size()),
                                                  static_cast<uint32_t>(c->width), c->elem, h->stream);
                if (e != hipSuccess) return h->hip_fail(e, "gather_rows");
            }
        }
        P->model_cols[k] = c->live;
        P->model_hist[k] = (h->hist_ring && k < h->d_model_hist.size()) ? h->d_model_hist[k] : nullptr;
    }
    P->tick0 = h->tick;
    return build_dev_ops(h, P->ops, &P->n_ops, &P->vel_independent);
}

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// built-in pipes or the handle's generated pipe
hipError_t launch_any(sixdof_handle* h, const StepParams& P) {
    if (h->desc.integrator == SIXDOF_INTEGRATOR_NONE && !h->custom_launch) return hipErrorInvalidValue;
    if (h->custom_launch) return static_cast<hipError_t>(h->custom_launch(&P, h->desc.integrator, h->desc.dtype, h->stream));
    return launch_step(P, h->desc.integrator, h->desc.dtype, h->stream);
}

int fill_pair_params(sixdof_handle* h, PairParams* P) {
    std::memset(P, 0, sizeof(*P));
    const size_t n = h->desc.n_entities;
    const size_t es = h->elem_size();
    const sixdof_effector_op& pop = h->ops.back();
    const uint32_t splits = pop.kind == SIXDOF_EFF_ALLPAIRS_GRAVITY_SOFTENED ? pair_splits_for(static_cast<uint32_t>(n)) : 1;
    (void)es;
    // scratch layout: pack[n,10] | partial[splits,n,width]   (f64)
    const bool allpairs = pop.kind == SIXDOF_EFF_ALLPAIRS_GRAVITY_SOFTENED;
    const size_t pwidth = allpairs ? kPartialForce : kPartialWidth;
    const size_t pack_bytes = align_up(sizeof(double) * kPackWidth * n, 256);
    const size_t partial_bytes = align_up(sizeof(double) * pwidth * n * splits, 256);
    const size_t total = pack_bytes + partial_bytes + (allpairs ? 0 : pack_bytes);      // edge lists: a second pack buffer (the one-launch tick ping-pongs)
    if (total > h->scratch_bytes) {
        if (h->d_scratch) hipFree(h->d_scratch), h->d_scratch = nullptr;
        HIP_TRY(h, hipMalloc(&h->d_scratch, total ? total : 256));
        h->scratch_bytes = total;
    }
    char* base = static_cast<char*>(h->d_scratch);
    P->pos = h->col(h->id_pos)->live;
    P->vel = h->col(h->id_vel)->live;
    P->accel = h->col(h->id_accel)->live;
    P->force = h->col(h->id_force)->live;
    P->inertia = h->col(h->id_inertia)->live;
    P->n = static_cast<uint32_t>(n);
    P->dt_g = h->desc.simulation_time_step;
    P->dt = h->desc.has_time_step ? h->desc.time_step : h->desc.simulation_time_step;
    P->pack = reinterpret_cast<double*>(base);
    P->partial = reinterpret_cast<double*>(base + pack_bytes);
    P->pack_next = allpairs ? nullptr : reinterpret_cast<double*>(base + pack_bytes + partial_bytes);
    P->splits = splits;
    P->partial_width = static_cast<uint32_t>(pwidth);
    P->pair_kind = pop.kind;
    P->p0 = pop.p[0];
    P->p1 = pop.p[1];
    if (pop.kind != SIXDOF_EFF_ALLPAIRS_GRAVITY_SOFTENED) {
        if (!h->d_csr_start) return h->fail(SIXDOF_ERR_COMPONENT_NOT_FOUND, "step: edge effector without sixdof_set_edges");
        P->row_start = h->d_csr_start;
        P->dst = h->d_csr_dst;
        P->n_edges = static_cast<uint32_t>(h->edge_src.size());
        P->n_hubs = h->n_hubs;
        P->n_hub_chunks = h->n_hub_chunks;
        if (h->n_hubs) {
            P->hub_rows = h->d_hub;
            P->hub_chunk_start = h->d_hub + h->n_hubs;
            P->chunk_e0 = h->d_hub + h->n_hubs + (h->n_hubs + 1);
            P->chunk_row = P->chunk_e0 + h->n_hub_chunks;
            P->chunk_partial = h->d_chunk_partial;
        }
    }
    uint32_t vi = 0;
    return build_dev_ops(h, P->ops, &P->n_ops, &vi);
}

// reference.py:164-175 (bisect_right interpolation), evaluated once per tick on the host: every rollout
// of a campaign sees the same reference profile at a given tick.
double ref_interp(double t, const std::vector<double>& xs, const std::vector<double>& ys) {
    const size_t n = xs.size();
    if (t <= xs[0]) return ys[0];
    if (t >= xs[n - 1]) return ys[n - 1];
    size_t lo = 0, hi = n;
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (t < xs[mid]) hi = mid; else lo = mid + 1;
    }
    const size_t a = lo - 1, b = lo;
    const double span = xs[b] - xs[a];
    if (span <= 0.0) return ys[a];
    const double frac = (t - xs[a]) / span;
    return ys[a] + (ys[b] - ys[a]) * frac;
}

// Telemetry ring for the paths whose kernels do not record in-line (pair / edge_fold ticks, the Apollo model): with a
// ring enabled those paths are stepped ONE tick per launch and the four live output columns are copied, device to
// device on the compute stream, into the tick's ring slot — every tick is there, in the layout sixdof_history_read /
// _stream expect.  (The fused per-entity kernel records from registers instead, step_kernel.hpp.)
int snapshot_tick_to_ring(sixdof_handle* h, uint64_t ticks_done) {
    if (!h->hist_ring) return SIXDOF_OK;
    const size_t n = h->desc.n_entities, es = h->elem_size();
    const size_t slot = static_cast<size_t>((ticks_done - 1) % h->hist_ring);
    const uint64_t ids[4] = {h->id_pos, h->id_vel, h->id_accel, h->id_force};
    const size_t widths[4] = {7, 6, 6, 6};
    for (int k = 0; k < 4; k++) {
        const size_t block = n * widths[k] * es;
        if (!block) continue;
        HIP_TRY(h, hipMemcpyAsync(static_cast<char*>(h->d_hist[k]) + slot * block, h->col(ids[k])->live, block,
                                  hipMemcpyDeviceToDevice, h->stream));
    }
    return SIXDOF_OK;
}

int step_apollo(sixdof_handle* h, uint64_t n_ticks, uint64_t* launches) {
    const char* names[5] = {"apollo_state", "apollo_params", "apollo_guidance", "apollo_score", "apollo_result"};
    const uint64_t widths[5] = {APOLLO_N_STATE, APOLLO_N_PARAMS, APOLLO_N_GUIDANCE, APOLLO_N_SCORE, APOLLO_N_RESULT};
    Column* c[5];
    for (int k = 0; k < 5; k++) {
        c[k] = h->col(cid(names[k]));
        if (!c[k]) return h->fail(SIXDOF_ERR_COMPONENT_NOT_FOUND, std::string("step: Apollo model column not bound: ") + names[k]);
        if (c[k]->width != widths[k] || c[k]->prim != SIXDOF_PRIM_F64)
            return h->fail(SIXDOF_ERR_VALUE_SIZE_MISMATCH, std::string("step: bad shape for ") + names[k]);
        if (!c[k]->joined) {
            int rc = resolve_join(h, c[k]);
            if (rc != SIXDOF_OK) return rc;
            if (c[k]->compact) {
                hipError_t e = launch_gather_rows(c[k]->compact, c[k]->dev, c[k]->d_rows, static_cast<uint32_t>(h->joined_ids.size()),
                                                  static_cast<uint32_t>(c[k]->width), c[k]->elem, h->stream);
                if (e != hipSuccess) return h->hip_fail(e, "gather_rows");
            }
        }
    }
    ApolloParams P{};
    P.pos = static_cast<double*>(h->col(h->id_pos)->live);
    P.vel = static_cast<double*>(h->col(h->id_vel)->live);
    P.accel = static_cast<double*>(h->col(h->id_accel)->live);
    P.force = static_cast<double*>(h->col(h->id_force)->live);
    P.inertia = static_cast<double*>(h->col(h->id_inertia)->live);
    P.state = static_cast<double*>(c[0]->live);
    P.params = static_cast<const double*>(c[1]->live);
    P.guidance = static_cast<double*>(c[2]->live);
    P.score = static_cast<double*>(c[3]->live);
    P.result = static_cast<double*>(c[4]->live);
    P.n = static_cast<uint32_t>(h->desc.n_entities);
    P.max_ticks = h->ap_max_ticks;
    P.guidance_period = h->ap_guidance_period;
    P.ticks_per_telemetry = h->ap_ticks_per_telemetry;
    P.dt = h->desc.simulation_time_step;
    const uint32_t K = h->hist_ring ? 1u : h->desc.ticks_per_launch;   // recording: one tick per launch, see snapshot_tick_to_ring
    if (K > h->tick_refs_cap) {
        if (h->d_tick_refs) hipFree(h->d_tick_refs), h->d_tick_refs = nullptr;
        HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&h->d_tick_refs), static_cast<size_t>(K) * 8 * sizeof(double)));
        h->tick_refs_cap = K;
    }
    std::vector<double> refs(static_cast<size_t>(K) * 8);
    uint64_t done = 0;
    while (done < n_ticks) {
        const uint32_t k = static_cast<uint32_t>(std::min<uint64_t>(K, n_ticks - done));
        for (uint32_t j = 0; j < k; j++) {
            // post_step's t_s = end_tick * SIM_TIME_STEP with end_tick = ticks completed - 1 (impeller2_server.rs:566,671)
            const double t_s = static_cast<double>(h->tick + done + j) * (1.0 / 120.0);
            double* r = &refs[static_cast<size_t>(j) * 8];
            r[0] = ref_interp(t_s, h->ap_time, h->ap_alt);
            r[1] = ref_interp(t_s, h->ap_time, h->ap_rate);
            r[2] = std::fabs(ref_interp(t_s, h->ap_time, h->ap_pitch));
            r[3] = ref_interp(t_s, h->ap_time, h->ap_hspeed);
            r[4] = ref_interp(t_s, h->ap_time, h->ap_downrange);
            r[5] = r[3] - ref_interp(t_s + 1.0, h->ap_time, h->ap_hspeed);
            r[6] = r[7] = 0.0;
        }
        // stream-ordered: the previous launch has consumed the buffer before this copy executes
        HIP_TRY(h, hipMemcpyAsync(h->d_tick_refs, refs.data(), static_cast<size_t>(k) * 8 * sizeof(double),
                                  hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));  // `refs` is reused by the next chunk
        P.tick_refs = h->d_tick_refs;
        P.n_ticks = k;
        P.tick0 = h->tick + done;
        hipError_t e = launch_apollo(P, h->stream);
        if (e != hipSuccess) return h->hip_fail(e, "launch_apollo");
        (*launches)++;
        done += k;
        if (h->hist_ring) {
            int rc = snapshot_tick_to_ring(h, h->tick + done);
            if (rc != SIXDOF_OK) return rc;
        }
    }
    return SIXDOF_OK;
}

}  // namespace

extern "C" {

int sixdof_set_model_apollo(sixdof_handle* h, const sixdof_apollo_tables* t) try {
    if (!h || !t || t->n < 2 || !t->time_s || !t->altitude_m || !t->descent_rate_mps || !t->pitch_deg ||
        !t->horizontal_speed_mps || !t->downrange_m)
        return SIXDOF_ERR_INVALID_ARGUMENT;
    if (h->desc.integrator != SIXDOF_INTEGRATOR_SEMI_IMPLICIT || h->desc.dtype != SIXDOF_F64)
        return h->fail(SIXDOF_ERR_UNSUPPORTED, "set_model_apollo: the example uses el.Integrator.SemiImplicit in f64 (sim.py:523)");
    h->ap_time.assign(t->time_s, t->time_s + t->n);
    h->ap_alt.assign(t->altitude_m, t->altitude_m + t->n);
    h->ap_rate.assign(t->descent_rate_mps, t->descent_rate_mps + t->n);
    h->ap_pitch.assign(t->pitch_deg, t->pitch_deg + t->n);
    h->ap_hspeed.assign(t->horizontal_speed_mps, t->horizontal_speed_mps + t->n);
    h->ap_downrange.assign(t->downrange_m, t->downrange_m + t->n);
    h->ap_guidance_period = t->guidance_period_ticks ? t->guidance_period_ticks : 5;
    h->ap_ticks_per_telemetry = t->ticks_per_telemetry ? t->ticks_per_telemetry : 3;
    h->ap_max_ticks = t->max_ticks;
    h->model = 1;
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(h))

int sixdof_last_timings(const sixdof_handle* h, sixdof_timings* out) try {
    if (!h || !out) return SIXDOF_ERR_INVALID_ARGUMENT;
    *out = h->last;
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(h))

int sixdof_count_nonfinite(sixdof_handle* h, uint64_t* count, uint8_t* row_flags) try {
    if (!h || !count) return SIXDOF_ERR_INVALID_ARGUMENT;
    if (!h->bound) return h->fail(SIXDOF_ERR_COMPONENT_NOT_FOUND, "count_nonfinite: no columns bound");
    HIP_TRY(h, hipSetDevice(h->device));
    const uint32_t n = static_cast<uint32_t>(h->desc.n_entities);
    unsigned long long* d_count = nullptr;
    uint8_t* d_flags = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&d_count), sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d_count, 0, sizeof(unsigned long long), h->stream);
    if (e == hipSuccess && row_flags && n) e = hipMalloc(reinterpret_cast<void**>(&d_flags), n);
    if (e == hipSuccess)
        e = launch_nonfinite(h->col(h->id_pos)->live, h->col(h->id_vel)->live, n, h->elem_size(), d_flags, d_count, h->stream);
    unsigned long long host_count = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&host_count, d_count, sizeof(host_count), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess && d_flags) e = hipMemcpyAsync(row_flags, d_flags, n, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    hipFree(d_count);
    if (d_flags) hipFree(d_flags);
    if (e != hipSuccess) return h->hip_fail(e, "count_nonfinite");
    *count = host_count;
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(h))

int sixdof_set_custom_pipe(sixdof_handle* h, const char* so_path, const uint64_t* aux_ids, size_t n_aux) try {
    if (!h || !so_path || (!aux_ids && n_aux)) return SIXDOF_ERR_INVALID_ARGUMENT;
    void* dl = dlopen(so_path, RTLD_NOW | RTLD_LOCAL);
    if (!dl) return h->fail(SIXDOF_ERR_BACKEND, std::string("set_custom_pipe: dlopen failed: ") + dlerror());
    auto abi = reinterpret_cast<CustomAbiFn>(dlsym(dl, "sixdof_custom_abi"));
    auto layout = reinterpret_cast<CustomLayoutFn>(dlsym(dl, "sixdof_custom_layout"));
    auto launch = reinterpret_cast<CustomLaunchFn>(dlsym(dl, "sixdof_custom_launch"));
    if (!abi || !layout || !launch || abi() != sizeof(StepParams)) {
        dlclose(dl);
        return h->fail(SIXDOF_ERR_BACKEND, "set_custom_pipe: not a generated pipe for this library build (StepParams layout differs)");
    }
    auto col_widths = reinterpret_cast<void (*)(unsigned*)>(dlsym(dl, "sixdof_custom_column_widths"));
    // a program that exchanges data between the entities of a world inside the wavefront (a whole-world StableHLO tick with one lane
    // per entity: elodin_amd/stablehlo.py, manifest "rows_per_world") lays a world out as that many consecutive rows
    auto rows_multiple = reinterpret_cast<unsigned (*)()>(dlsym(dl, "sixdof_custom_rows_multiple"));
    if (rows_multiple && rows_multiple() > 1 && h->desc.n_entities % rows_multiple() != 0) {
        const unsigned m = rows_multiple();
        dlclose(dl);
        return h->fail(SIXDOF_ERR_INVALID_ARGUMENT, "set_custom_pipe: the program lays a world out as " + std::to_string(m) +
                       " consecutive rows; " + std::to_string(h->desc.n_entities) + " rows are not a whole number of worlds");
    }
    const unsigned lay = layout();
    const size_t k_aux = lay & 0xff, k_model = (lay >> 8) & 0xff;
    if (k_aux > static_cast<size_t>(kMaxOps) || k_model > static_cast<size_t>(kMaxModelCols) || k_aux + k_model != n_aux) {
        dlclose(dl);
        return h->fail(SIXDOF_ERR_VALUE_SIZE_MISMATCH, "set_custom_pipe: column list does not match the generated code's layout");
    }
    if (h->custom_dl) dlclose(h->custom_dl);
    h->custom_dl = dl;
    h->custom_launch = launch;
    h->custom_rows_multiple = rows_multiple ? rows_multiple() : 1;      // checked again when the join is (re)sized: sixdof_bind_columns
    h->custom_aux.assign(aux_ids, aux_ids + k_aux);
    h->custom_model.assign(aux_ids + k_aux, aux_ids + k_aux + k_model);
    h->custom_tick_free = ((lay >> 17) & 1u) != 0;
    // row widths the generated code was built for (bit 31: a window column — memory-resident, any width, not recorded)
    h->custom_model_width.assign(k_model, 0u);
    if (col_widths && k_model) col_widths(h->custom_model_width.data());
    h->ops.clear();
    h->drop_graph();
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(h))

int sixdof_set_custom_pair(sixdof_handle* h, const char* so_path) try {
    if (!h || !so_path) return SIXDOF_ERR_INVALID_ARGUMENT;
    if (h->custom_launch) return h->fail(SIXDOF_ERR_UNSUPPORTED, "set_custom_pair: a generated per-entity pipe is installed; pair folds combine with built-in ops only");
    void* dl = dlopen(so_path, RTLD_NOW | RTLD_LOCAL);
    if (!dl) return h->fail(SIXDOF_ERR_BACKEND, std::string("set_custom_pair: dlopen failed: ") + dlerror());
    auto abi = reinterpret_cast<CustomPairAbiFn>(dlsym(dl, "sixdof_custom_pair_abi"));
    auto launch = reinterpret_cast<CustomPairLaunchFn>(dlsym(dl, "sixdof_custom_pair_launch"));
    if (!abi || !launch || abi() != sizeof(PairParams)) {
        dlclose(dl);
        return h->fail(SIXDOF_ERR_BACKEND, "set_custom_pair: not a generated pair fold for this library build (PairParams layout differs)");
    }
    if (h->pair_dl) dlclose(h->pair_dl);
    h->pair_dl = dl;
    h->pair_launch = launch;
    // an object generated for one launch shape only (codegen.build_pair(small=...)) says which: step follows the object
    auto only_small = reinterpret_cast<int (*)()>(dlsym(dl, "sixdof_custom_pair_only_small"));
    h->pair_only_small = only_small ? only_small() : -1;
    while (!h->ops.empty() && SIXDOF_EFF_IS_PAIR(h->ops.back().kind)) h->ops.pop_back();
    sixdof_effector_op op{};
    op.kind = SIXDOF_EFF_EDGE_CUSTOM;
    h->ops.push_back(op);
    h->drop_graph();
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(h))

int sixdof_set_history(sixdof_handle* h, uint32_t ring_ticks) try {
    if (!h) return SIXDOF_ERR_INVALID_ARGUMENT;
    if (!h->bound) return h->fail(SIXDOF_ERR_COMPONENT_NOT_FOUND, "set_history: bind Body columns first");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (void*& p : h->d_hist) {
        if (p) hipFree(p);
        p = nullptr;
    }
    for (void* p : h->d_model_hist) if (p) hipFree(p);
    h->d_model_hist.clear();
    h->hist_ring = 0;
    h->drop_graph();
    if (ring_ticks == 0) return SIXDOF_OK;
    const size_t n = h->desc.n_entities, es = h->elem_size();
    const size_t widths[4] = {7, 6, 6, 6};
    for (int k = 0; k < 4; k++) {
        const size_t bytes = static_cast<size_t>(ring_ticks) * n * widths[k] * es;
        hipError_t e = hipMalloc(&h->d_hist[k], bytes ? bytes : 16);
        if (e != hipSuccess) {
            for (void*& p : h->d_hist) {
                if (p) hipFree(p);
                p = nullptr;
            }
            return h->hip_fail(e, "set_history: hipMalloc of the ring");
        }
    }
    for (uint64_t id : h->custom_model) {      // component columns of a generated program are recorded too
        const Column* c = h->col(id);
        void* ring = nullptr;
        const size_t m_idx = h->d_model_hist.size();
        const bool window = m_idx < h->custom_model_width.size() && (h->custom_model_width[m_idx] >> 31);
        if (c && !window) {   // a window column is its own history (and far too wide to copy per tick)
            const size_t bytes = static_cast<size_t>(ring_ticks) * n * c->width * es;
            hipError_t e = hipMalloc(&ring, bytes ? bytes : 16);
            if (e != hipSuccess) return h->hip_fail(e, "set_history: hipMalloc of a component ring");
        }
        h->d_model_hist.push_back(ring);
    }
    h->hist_ring = ring_ticks;
    h->hist_first_tick = h->tick + 1;
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(h))

int sixdof_history_read(sixdof_handle* h, uint64_t component_id, uint64_t tick, void* host_dst) try {
    if (!h || !host_dst) return SIXDOF_ERR_INVALID_ARGUMENT;
    if (!h->hist_ring) return h->fail(SIXDOF_ERR_INVALID_ARGUMENT, "history_read: no history ring (sixdof_set_history)");
    int k = -1;
    size_t w = 6;
    if (component_id == h->id_pos) k = 0, w = 7;
    else if (component_id == h->id_vel) k = 1;
    else if (component_id == h->id_accel) k = 2;
    else if (component_id == h->id_force) k = 3;
    const void* ring_base = k >= 0 ? h->d_hist[k] : nullptr;
    if (k < 0) {
        for (size_t m = 0; m < h->custom_model.size() && m < h->d_model_hist.size(); m++)
            if (h->custom_model[m] == component_id && h->d_model_hist[m]) {
                ring_base = h->d_model_hist[m];
                w = h->col(component_id)->width;
            }
        if (!ring_base)
            return h->fail(SIXDOF_ERR_COMPONENT_NOT_FOUND, "history_read: only world_pos / world_vel / world_accel / force and the component columns of a generated program are recorded");
    }
    if (tick < h->hist_first_tick || tick > h->tick || tick + h->hist_ring <= h->tick)
        return h->fail(SIXDOF_ERR_INVALID_ARGUMENT, "history_read: tick is not in the ring");
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t block = static_cast<size_t>(h->desc.n_entities) * w * h->elem_size();
    const size_t slot = static_cast<size_t>((tick - 1) % h->hist_ring);
    if (block) HIP_TRY(h, hipMemcpyAsync(host_dst, static_cast<const char*>(ring_base) + slot * block, block, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(h))

int sixdof_history_stream(sixdof_handle* h, uint64_t first_tick, uint64_t n_ticks, void* const host_dst[4]) try {
    if (!h || !host_dst) return SIXDOF_ERR_INVALID_ARGUMENT;
    if (!h->hist_ring) return h->fail(SIXDOF_ERR_INVALID_ARGUMENT, "history_stream: no history ring (sixdof_set_history)");
    if (n_ticks == 0) return SIXDOF_OK;
    const uint64_t last = first_tick + n_ticks - 1;
    if (first_tick < h->hist_first_tick || last > h->tick || first_tick + h->hist_ring <= h->tick || n_ticks > h->hist_ring)
        return h->fail(SIXDOF_ERR_INVALID_ARGUMENT, "history_stream: ticks are not (all) in the ring");
    HIP_TRY(h, hipSetDevice(h->device));
    if (!h->copy_stream) {
        HIP_TRY(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
        HIP_TRY(h, hipEventCreateWithFlags(&h->ev_snap, hipEventDisableTiming));
        HIP_TRY(h, hipEventCreateWithFlags(&h->ev_copied, hipEventDisableTiming));
    }
    HIP_TRY(h, hipEventRecord(h->ev_snap, h->stream));            // everything recorded so far is in the ring after this
    HIP_TRY(h, hipStreamWaitEvent(h->copy_stream, h->ev_snap, 0));
    const size_t n = h->desc.n_entities, es = h->elem_size();
    for (int k = 0; k < 4; k++) {
        if (!host_dst[k]) continue;
        const size_t block = n * (k == 0 ? 7 : 6) * es;
        if (!block) continue;
        if (std::find(h->pinned_user.begin(), h->pinned_user.end(), host_dst[k]) == h->pinned_user.end()) {
            if (hipHostRegister(host_dst[k], block * n_ticks, hipHostRegisterDefault) == hipSuccess) h->pinned_user.push_back(host_dst[k]);
            else (void)hipGetLastError();
        }
        // the run is contiguous in the ring except where it wraps: at most two copies per column
        const size_t slot0 = static_cast<size_t>((first_tick - 1) % h->hist_ring);
        const size_t head = std::min<size_t>(n_ticks, h->hist_ring - slot0);
        char* dst = static_cast<char*>(host_dst[k]);
        const char* ring = static_cast<const char*>(h->d_hist[k]);
        HIP_TRY(h, hipMemcpyAsync(dst, ring + slot0 * block, head * block, hipMemcpyDeviceToHost, h->copy_stream));
        if (head < n_ticks)
            HIP_TRY(h, hipMemcpyAsync(dst + head * block, ring, (n_ticks - head) * block, hipMemcpyDeviceToHost, h->copy_stream));
    }
    HIP_TRY(h, hipEventRecord(h->ev_copied, h->copy_stream));
    h->copy_pending = true;
    h->stream_lo = first_tick;
    h->stream_hi = last;
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(h))

int sixdof_download_column(sixdof_handle* h, uint64_t component_id) try {
    if (!h) return SIXDOF_ERR_INVALID_ARGUMENT;
    Column* c = h->col(component_id);
    if (!c) return h->fail(SIXDOF_ERR_COMPONENT_NOT_FOUND, "download_column: unknown component");
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = scatter_back(h, c);
    if (rc != SIXDOF_OK) return rc;
    if (c->bytes) HIP_TRY(h, hipMemcpyAsync(c->host, c->dev, c->bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(h))

// H2D of ONE bound column: an external write to a component (StepContext.write_component, copy_db_to_world's per-component copy,
// impeller2_server.rs:320-362) reaches the device without re-uploading — and so clobbering — the columns the host never downloaded.
int sixdof_upload_column(sixdof_handle* h, uint64_t component_id) try {
    if (!h) return SIXDOF_ERR_INVALID_ARGUMENT;
    Column* c = h->col(component_id);
    if (!c) return h->fail(SIXDOF_ERR_COMPONENT_NOT_FOUND, "upload_column: unknown component");
    if (!h->resident) return h->fail(SIXDOF_ERR_INVALID_ARGUMENT, "upload_column: nothing is resident yet (sixdof_upload first)");
    HIP_TRY(h, hipSetDevice(h->device));
    if (c->bytes) HIP_TRY(h, hipMemcpyAsync(c->dev, c->host, c->bytes, hipMemcpyHostToDevice, h->stream));
    if (c->joined && c->compact) {
        hipError_t e = launch_gather_rows(c->compact, c->dev, c->d_rows, static_cast<uint32_t>(h->joined_ids.size()),
                                          static_cast<uint32_t>(c->width), c->elem, h->stream);
        if (e != hipSuccess) return h->hip_fail(e, "gather_rows");
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (component_id == h->id_accel) h->accel_is_host_data = true;
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(h))

// Launches per replayed chain.  A long chain amortises the gap between two replays (4,096 launches: 4.96 -> 4.83 us each with
// 128-launch chains) but starts later (100 launches as one chain: 8 % slower than 32 + 32 + 32 + 4), so a batch OPENS with a
// 32-launch chain and, when at least four fit, continues with 128-launch ones (profiles/r02_graph_len_ab.txt).  SIXDOF_GRAPH_LONG=<n> overrides the
// long length for A/B runs (n <= 32: short chains only).
constexpr uint32_t kGraphLen = 32;
static const uint32_t kGraphLong = [] { const char* e = std::getenv("SIXDOF_GRAPH_LONG"); const int v = e ? std::atoi(e) : 128; return v > 32 ? static_cast<uint32_t>(v) : 0u; }();

bool graph_eligible(const sixdof_handle* h) {
    return (h->desc.flags & SIXDOF_FLAG_USE_GRAPH) && !(h->desc.flags & SIXDOF_FLAG_TIME_EACH_LAUNCH) && !h->hist_ring &&
           (h->custom_model.empty() || h->custom_tick_free) && h->model == 0 && !h->has_pair_op();
}

constexpr uint32_t kGraphMinLen = 4;   // shorter chains are launched eagerly (a replay costs ~10-16 us of host time)
constexpr size_t kGraphCacheMax = 8;

// Everything a captured launch bakes in: the whole argument block (column pointers, n, both time steps, effector ops
// and their column pointers, cache policy) plus integrator and dtype.  Any change re-captures — e.g. sixdof_tick
// overwriting simulation_time_step from its input slot, or a rebind.  tick0 / hist_slot0 differ per launch but are
// only read by paths that are not graph-eligible.
uint64_t step_signature(const sixdof_handle* h, StepParams P, uint32_t K) {
    P.tick0 = 0;
    P.hist_slot0 = 0;
    P.n_ticks = K;
    uint64_t sig = 0xcbf29ce484222325ull;
    auto mix = [&](const void* p, size_t n) {
        const unsigned char* b = static_cast<const unsigned char*>(p);
        for (size_t i = 0; i < n; i++) sig = (sig ^ b[i]) * 0x100000001b3ull;
    };
    mix(&P, sizeof(P));
    const int32_t extra[2] = {h->desc.integrator, h->desc.dtype};
    mix(extra, sizeof(extra));
    return sig;
}

// Row blocks a replayed chain is split into (ensure_graph): SIXDOF_GRAPH_SPLIT=<S>, default 1.  Hand-written pipes only (a
// generated program's columns have widths of their own), no recording, and at least 4,096 rows per block.
static const uint32_t kGraphSplit = [] { const char* e = std::getenv("SIXDOF_GRAPH_SPLIT"); const int v = e ? std::atoi(e) : 1; return static_cast<uint32_t>(v < 1 ? 1 : (v > 16 ? 16 : v)); }();

uint32_t graph_split_for(const sixdof_handle* h, const StepParams& P) {
    if (kGraphSplit <= 1 || h->custom_launch || P.hist_ring || P.n < kGraphSplit * 4096u) return 1;
    return kGraphSplit;
}

// The argument block of rows [row0, row0 + rows) of a launch: every column pointer moved to the block's first row.
StepParams row_block(const sixdof_handle* h, const StepParams& P, uint32_t row0, uint32_t rows) {
    StepParams Q = P;
    const size_t es = h->elem_size();
    auto at = [&](const void* p, size_t width) { return p ? static_cast<const char*>(p) + static_cast<size_t>(row0) * width * es : nullptr; };
    Q.pos = const_cast<char*>(at(P.pos, 7));
    Q.vel = const_cast<char*>(at(P.vel, 6));
    Q.accel = const_cast<char*>(at(P.accel, 6));
    Q.force = const_cast<char*>(at(P.force, 6));
    Q.inertia = at(P.inertia, 7);
    for (uint32_t k = 0; k < P.n_ops && k < static_cast<uint32_t>(kMaxOps); k++)
        Q.ops[k].aux = at(P.ops[k].aux, P.ops[k].aux_width ? P.ops[k].aux_width : 3);
    Q.n = rows;
    return Q;
}

// An executable graph of `len` identical launches of the step kernel (cached per chain length; all cached graphs share
// one signature and are dropped together when it changes).
int ensure_graph(sixdof_handle* h, const StepParams& P, uint32_t K, uint32_t len, hipGraphExec_t* out) {
    const uint64_t sig = step_signature(h, P, K);
    if (h->graph_sig != sig || h->graph_k != K) h->drop_graph();
    auto it = h->graphs.find(len);
    if (it != h->graphs.end()) {
        *out = it->second;
        return SIXDOF_OK;
    }
    if (h->graphs.size() >= kGraphCacheMax) {   // many distinct batch lengths: keep the long chain, drop the rest
        for (auto g = h->graphs.begin(); g != h->graphs.end();) {
            if (g->first == kGraphLen || g->first == kGraphLong) { ++g; continue; }
            hipGraphExecDestroy(g->second);
            g = h->graphs.erase(g);
        }
    }
    hipGraph_t g = nullptr;
    const uint32_t S = graph_split_for(h, P);
    if (S > 1) {      // the side streams and the fork / join events exist before the capture starts
        while (h->split_streams.size() < S - 1) {
            hipStream_t s = nullptr;
            HIP_TRY(h, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            h->split_streams.push_back(s);
            hipEvent_t e = nullptr;
            HIP_TRY(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            h->split_joins.push_back(e);
        }
        if (!h->split_fork) HIP_TRY(h, hipEventCreateWithFlags(&h->split_fork, hipEventDisableTiming));
    }
    HIP_TRY(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    hipError_t le = hipSuccess;
    if (S <= 1) {
        for (uint32_t i = 0; i < len && le == hipSuccess; i++) le = launch_any(h, P);
    } else {
        // ROW-BLOCK CHAINS.  Rows are independent and a launch of this path exchanges nothing, so tick k + 1 of a row block
        // depends on tick k of THAT block only: the replayed graph is S parallel chains of `len` launches, one per contiguous
        // row block, forked from and joined back into the handle's stream once per replay.  Same kernel, same rows per wave,
        // same arithmetic -> the same bits; what changes is that one block's end-of-kernel write-back and dispatch gap overlap
        // another block's loads and math (profiles/r06_k1_floor.md).
        const uint32_t block = ((P.n + S - 1) / S + 255u) & ~255u;      // whole 256-row groups: every wave shape divides it
        le = hipEventRecord(h->split_fork, h->stream);
        for (uint32_t s = 0; s < S && le == hipSuccess; s++) {
            const uint32_t row0 = s * block;
            if (row0 >= P.n) break;
            hipStream_t st = s == 0 ? h->stream : h->split_streams[s - 1];
            if (s > 0) le = hipStreamWaitEvent(st, h->split_fork, 0);
            const StepParams Ps = row_block(h, P, row0, std::min(block, P.n - row0));
            for (uint32_t i = 0; i < len && le == hipSuccess; i++) le = launch_step(Ps, h->desc.integrator, h->desc.dtype, st);
            if (s > 0 && le == hipSuccess) le = hipEventRecord(h->split_joins[s - 1], st);
            if (s > 0 && le == hipSuccess) le = hipStreamWaitEvent(h->stream, h->split_joins[s - 1], 0);
        }
    }
    hipError_t ce = hipStreamEndCapture(h->stream, &g);
    if (le != hipSuccess) return h->hip_fail(le, "launch_step (capture)");
    if (ce != hipSuccess) return h->hip_fail(ce, "hipStreamEndCapture");
    hipGraphExec_t exec = nullptr;
    hipError_t ie = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
    hipGraphDestroy(g);
    if (ie != hipSuccess) return h->hip_fail(ie, "hipGraphInstantiate");
    (void)hipGraphUpload(exec, h->stream);   // move the one-off device-side setup out of the first replay
    h->graphs[len] = exec;
    h->graph_k = K;
    h->graph_sig = sig;
    *out = exec;
    return SIXDOF_OK;
}

// Build the replay graph ahead of the first long batch (called when the columns become resident and when the batch
// shape changes), so that no step call pays for capture + instantiation.
int prepare_graph(sixdof_handle* h) {
    if (!h->bound || !graph_eligible(h)) return SIXDOF_OK;
    StepParams P;
    int rc = fill_step_params(h, &P);
    if (rc != SIXDOF_OK) return SIXDOF_OK;     // not steppable yet (columns missing): the step call will report it
    P.n_ticks = h->desc.ticks_per_launch;
    hipGraphExec_t unused = nullptr;
    rc = ensure_graph(h, P, h->desc.ticks_per_launch, kGraphLen, &unused);
    if (rc == SIXDOF_OK) (void)hipStreamSynchronize(h->stream);
    return rc;
}

int sixdof_prepare_step(sixdof_handle* h, uint64_t n_ticks) try {
    if (!h) return SIXDOF_ERR_INVALID_ARGUMENT;
    if (!h->bound || !h->resident) return h->fail(SIXDOF_ERR_COMPONENT_NOT_FOUND, "prepare_step: upload the columns first");
    if (!graph_eligible(h)) return SIXDOF_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    StepParams P;
    int rc = fill_step_params(h, &P);
    if (rc != SIXDOF_OK) return rc;
    const uint32_t K = h->desc.ticks_per_launch;
    P.n_ticks = K;
    // the chains a batch of `full` K-tick launches replays (the same arithmetic as sixdof_step)
    auto prepare = [&](uint64_t full) -> int {
        hipGraphExec_t unused = nullptr;
        int prc;
        if (full >= kGraphLen && (prc = ensure_graph(h, P, K, kGraphLen, &unused)) != SIXDOF_OK) return prc;
        if (kGraphLong && full >= kGraphLen + 4 * kGraphLong) {
            if ((prc = ensure_graph(h, P, K, kGraphLong, &unused)) != SIXDOF_OK) return prc;
            full = (full - kGraphLen) % kGraphLong;     // what the opening chain and the long ones leave
        }
        const uint32_t tail_len = static_cast<uint32_t>(full % kGraphLen);
        if (full >= kGraphMinLen && tail_len >= kGraphMinLen && (prc = ensure_graph(h, P, K, tail_len, &unused)) != SIXDOF_OK) return prc;
        return SIXDOF_OK;
    };
    const uint64_t full = n_ticks / K;
    if ((rc = prepare(full)) != SIXDOF_OK) return rc;
    // the first RK4 step after an upload takes one eager launch out of `full` (sixdof_step: accel_in_check), so that
    // batch replays `full - 1` launches; whether the batch being prepared is that one the library cannot know (a warm-up
    // may come first), so both shapes are captured
    if (h->accel_is_host_data && h->desc.integrator == SIXDOF_INTEGRATOR_RK4 && full > 0 && (rc = prepare(full - 1)) != SIXDOF_OK) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(h))

int sixdof_step(sixdof_handle* h, uint64_t n_ticks, sixdof_timings* tm) try {
    if (!h) return SIXDOF_ERR_INVALID_ARGUMENT;
    if (!h->bound) return h->fail(SIXDOF_ERR_COMPONENT_NOT_FOUND, "step: no columns bound");
    HIP_TRY(h, hipSetDevice(h->device));
    const double t0 = now_ms();
    uint64_t launches = 0;
    const bool async_step = (h->desc.flags & SIXDOF_FLAG_ASYNC_STEP) != 0;
    if (h->step_pending || h->prev_pending) {
        // asynchronous batches alternate between two event pairs, so the host may enqueue batch i+1 while batch i
        // still computes; the pair about to be re-recorded belongs to batch i-1 (long finished): read its time
        std::swap(h->ev0, h->evp0);
        std::swap(h->ev1, h->evp1);
        std::swap(h->step_pending, h->prev_pending);
        if (h->step_pending) {
            float ms_prev = 0.f;
            HIP_TRY(h, hipEventSynchronize(h->ev1));
            if (hipEventElapsedTime(&ms_prev, h->ev0, h->ev1) == hipSuccess) h->last.kernel_device_ms = ms_prev;
            h->step_pending = false;
        }
    }
    if (h->hist_ring && h->copy_pending && h->stream_hi >= h->stream_lo && n_ticks) {
        // ticks tick+1 .. tick+n land in slots (t-1) % ring: hold the compute stream back only if that range reaches a
        // slot the copy stream may still be reading
        const uint64_t ring = h->hist_ring, in_flight = h->stream_hi - h->stream_lo + 1;
        const uint64_t a = h->tick % ring, b = (h->stream_lo - 1) % ring;       // first slot written / first slot read
        const uint64_t gap_ab = (b + ring - a) % ring, gap_ba = (a + ring - b) % ring;
        const bool overlap = n_ticks + in_flight > ring || gap_ab < std::min<uint64_t>(n_ticks, ring) || gap_ba < in_flight;
        if (overlap) HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_copied, 0));
    }
    HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
    if (h->desc.integrator == SIXDOF_INTEGRATOR_NONE && (!h->custom_launch || h->model != 0 || h->has_pair_op()))
        return h->fail(SIXDOF_ERR_UNSUPPORTED, "step: SIXDOF_INTEGRATOR_NONE runs generated system programs only (sixdof_set_custom_pipe)");
    if (h->model == 1) {
        int rc = step_apollo(h, n_ticks, &launches);
        if (rc != SIXDOF_OK) return rc;
    } else if (h->has_pair_op()) {
        if (h->desc.dtype != SIXDOF_F64) return h->fail(SIXDOF_ERR_UNSUPPORTED, "step: pair effectors are f64 only");
        PairParams P;
        int rc = fill_pair_params(h, &P);
        if (rc != SIXDOF_OK) return rc;
        const char* no_small = std::getenv("SIXDOF_PAIR_SMALL");   // "0": force the multi-kernel path (tests)
        if (P.pair_kind == SIXDOF_EFF_EDGE_CUSTOM) {
            if (!h->pair_launch) return h->fail(SIXDOF_ERR_BACKEND, "step: custom pair op without sixdof_set_custom_pair");
            const bool small = h->pair_only_small >= 0 ? h->pair_only_small == 1 : (P.n <= kPairSmallMax && !(no_small && no_small[0] == '0'));
            if (small && P.n > kPairSmallMax)
                return h->fail(SIXDOF_ERR_INVALID_ARGUMENT, "step: the pair object was generated for the one-launch small-graph kernel (<= " +
                               std::to_string(kPairSmallMax) + " rows) but the joined graph has " + std::to_string(P.n) + " rows");
            const uint32_t K = h->hist_ring ? 1u : (small ? h->desc.ticks_per_launch : 1u << 20);
            for (uint64_t done = 0; done < n_ticks;) {
                const uint32_t k = static_cast<uint32_t>(std::min<uint64_t>(K, n_ticks - done));
                hipError_t e = static_cast<hipError_t>(h->pair_launch(&P, h->desc.integrator, k, small ? 1 : 0, h->stream, &launches));
                if (e != hipSuccess) return h->hip_fail(e, "custom pair launch");
                done += k;
                P.packed = 1;      // the integrate kernel left the next tick's pack rows (this call only: PairParams::packed)
                if (int src = snapshot_tick_to_ring(h, h->tick + done); src != SIXDOF_OK) return src;
            }
        } else if (P.n <= kPairSmallMax && !(no_small && no_small[0] == '0')) {   // small graphs: ticks_per_launch ticks per launch
            const uint32_t K = h->hist_ring ? 1u : h->desc.ticks_per_launch;
            for (uint64_t done = 0; done < n_ticks;) {
                const uint32_t k = static_cast<uint32_t>(std::min<uint64_t>(K, n_ticks - done));
                hipError_t e = launch_pair_small(P, h->desc.integrator, k, h->stream, &launches);
                if (e != hipSuccess) return h->hip_fail(e, "launch_pair_small");
                done += k;
                if (int src = snapshot_tick_to_ring(h, h->tick + done); src != SIXDOF_OK) return src;
            }
        } else {
            // a batch is ONE pack launch, then fold + integrate per tick (the integrate kernel writes the next tick's pack rows);
            // with a telemetry ring the batch is cut at every tick for the snapshot, the pack rows carry over all the same
            const uint32_t K = h->hist_ring ? 1u : 1u << 20;
            for (uint64_t done = 0; done < n_ticks;) {
                const uint32_t k = static_cast<uint32_t>(std::min<uint64_t>(K, n_ticks - done));
                hipError_t e = launch_pair_ticks(P, h->desc.integrator, k, h->stream, &launches);
                if (e != hipSuccess) return h->hip_fail(e, "launch_pair_ticks");
                done += k;
                P.packed = 1;
                if (int src = snapshot_tick_to_ring(h, h->tick + done); src != SIXDOF_OK) return src;
            }
        }
    } else {
        StepParams P;
        int rc = fill_step_params(h, &P);
        if (rc != SIXDOF_OK) return rc;
        const uint32_t K = h->desc.ticks_per_launch;
        uint64_t full = n_ticks / K;
        const uint32_t rem = static_cast<uint32_t>(n_ticks % K);
        uint32_t rem_left = rem;
        P.n_ticks = K;
        // Long batches of identical launches replay from a hipGraph (launch-bound regime:
        // a 65,536-entity tick is a few microseconds of device time).
        const bool time_each = (h->desc.flags & SIXDOF_FLAG_TIME_EACH_LAUNCH) != 0;
        if (time_each) {
            const uint64_t need = 2 * (full + (rem ? 1 : 0));
            if (need > 8192) return h->fail(SIXDOF_ERR_INVALID_ARGUMENT, "step: TIME_EACH_LAUNCH supports <= 4096 launches per call");
            while (h->launch_events.size() < need) {
                hipEvent_t e = nullptr;
                HIP_TRY(h, hipEventCreate(&e));
                h->launch_events.push_back(e);
            }
        }
        uint64_t ticks_issued = 0;   // history slot of a launch's first tick = ticks done before it
        uint64_t graph_launches = 0;
        if (h->accel_is_host_data && n_ticks > 0) {
            // First launch after an upload: the world_accel column holds whatever the host put there.  The reference's RK4
            // forms v_s = v0 + 0 * a_in on stage 0 (rk4.rs:96-100), so a non-finite row poisons that tick; this one launch
            // reads the column to do the same (step_kernel.hpp).  Every later a_in is this kernel's own output and is
            // already folded into v0.  A launch of its own, eager, so the replay graphs never carry the flag.
            h->accel_is_host_data = false;
            if (h->desc.integrator == SIXDOF_INTEGRATOR_RK4) {
                StepParams P1 = P;
                P1.accel_in_check = 1;
                P1.n_ticks = static_cast<uint32_t>(std::min<uint64_t>(K, n_ticks));
                P1.hist_slot0 = h->tick;
                P1.tick0 = h->tick;
                if (time_each) HIP_TRY(h, hipEventRecord(h->launch_events[0], h->stream));
                hipError_t e = launch_any(h, P1);
                if (e != hipSuccess) return h->hip_fail(e, "launch_step");
                if (time_each) HIP_TRY(h, hipEventRecord(h->launch_events[1], h->stream));
                launches = 1;
                if (n_ticks >= K) full -= 1;   // that launch was one of the K-tick launches ...
                else rem_left = 0;             // ... or the whole (short) batch
            }
        }
        if (graph_eligible(h) && full >= kGraphMinLen) {
            // long batches replay 32-launch chains; what is left (or a short batch as a whole, e.g. 20 ticks) replays as
            // ONE chain of exactly that length, captured on first use and cached — so a short timed region is
            // steady-state device work too, not eager launches racing the host.
            hipGraphExec_t big = nullptr, tail = nullptr, longer = nullptr;
            uint64_t n_long = 0;
            const uint64_t eager_before = launches;
            if (kGraphLong && full >= kGraphLen + 4 * kGraphLong) {   // open with one short chain, then long ones (long batches only:
                                                                    // 200 launches as 32 + 128 + 40 measured 5 % slower)
                int grc = ensure_graph(h, P, K, kGraphLong, &longer);
                if (grc != SIXDOF_OK) return grc;
                n_long = (full - kGraphLen) / kGraphLong;
            }
            const uint32_t tail_len = static_cast<uint32_t>((full - n_long * kGraphLong) % kGraphLen);
            if (full >= kGraphLen) {
                int grc = ensure_graph(h, P, K, kGraphLen, &big);
                if (grc != SIXDOF_OK) return grc;
            }
            if (tail_len >= kGraphMinLen) {
                int grc = ensure_graph(h, P, K, tail_len, &tail);
                if (grc != SIXDOF_OK) return grc;
            }
            // a capture may just have happened after ev0 was recorded: re-record so the pair brackets real work only
            HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
            if (n_long) {
                HIP_TRY(h, hipGraphLaunch(big, h->stream));
                full -= kGraphLen;
                launches += kGraphLen;
                for (uint64_t i = 0; i < n_long; i++) {
                    HIP_TRY(h, hipGraphLaunch(longer, h->stream));
                    full -= kGraphLong;
                    launches += kGraphLong;
                }
            }
            while (full >= kGraphLen) {
                HIP_TRY(h, hipGraphLaunch(big, h->stream));
                full -= kGraphLen;
                launches += kGraphLen;
            }
            if (tail) {
                HIP_TRY(h, hipGraphLaunch(tail, h->stream));
                full -= tail_len;
                launches += tail_len;
            }
            graph_launches = launches - eager_before;
        }
        h->last.graph_launches = graph_launches;
        ticks_issued = launches * K;
        for (uint64_t i = 0; i < full; i++) {
            if (time_each) HIP_TRY(h, hipEventRecord(h->launch_events[2 * launches], h->stream));
            P.hist_slot0 = h->tick + ticks_issued;
            P.tick0 = h->tick + ticks_issued;
            ticks_issued += K;
            hipError_t e = launch_any(h, P);
            if (e != hipSuccess) return h->hip_fail(e, "launch_step");
            if (time_each) HIP_TRY(h, hipEventRecord(h->launch_events[2 * launches + 1], h->stream));
            launches++;
        }
        if (rem_left) {
            P.n_ticks = rem_left;
            P.hist_slot0 = h->tick + ticks_issued;
            P.tick0 = h->tick + ticks_issued;
            if (time_each) HIP_TRY(h, hipEventRecord(h->launch_events[2 * launches], h->stream));
            hipError_t e = launch_any(h, P);
            if (e != hipSuccess) return h->hip_fail(e, "launch_step");
            if (time_each) HIP_TRY(h, hipEventRecord(h->launch_events[2 * launches + 1], h->stream));
            launches++;
        }
    }
    HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
    if (!async_step) {
        // short batches finish in tens of microseconds: poll for that long before paying a blocking wait's wake-up
        const double spin_until = now_ms() + 0.25;
        hipError_t q = hipEventQuery(h->ev1);
        while (q == hipErrorNotReady && now_ms() < spin_until) q = hipEventQuery(h->ev1);
        if (q != hipSuccess) {
            (void)hipGetLastError();
            HIP_TRY(h, hipStreamSynchronize(h->stream));
        }
    }
    h->tick += n_ticks;  // increment_sim_tick (globals.rs:42-44), once per tick
    h->step_pending = async_step;
    {
        if (!async_step) {
            float ms0 = 0.f;
            hipEventElapsedTime(&ms0, h->ev0, h->ev1);
            h->last.kernel_device_ms = ms0;
        }
        h->last.kernel_invoke_ms = now_ms() - t0;
        h->last.launches = launches;
        h->last.ticks = n_ticks;
    }
    if (tm) {
        tm->h2d_upload_ms = h->last.h2d_upload_ms;
        tm->d2h_download_ms = h->last.d2h_download_ms;
        tm->kernel_device_ms = h->last.kernel_device_ms;
        tm->kernel_invoke_ms = h->last.kernel_invoke_ms;
        tm->launches = launches;
        tm->ticks = n_ticks;
        tm->kernel_sum_ms = 0.0;
        tm->graph_launches = h->last.graph_launches;
        if ((h->desc.flags & SIXDOF_FLAG_TIME_EACH_LAUNCH) && !h->has_pair_op()) {
            double sum = 0.0;
            for (uint64_t i = 0; i < launches && 2 * i + 1 < h->launch_events.size(); i++) {
                float one = 0.f;
                if (hipEventElapsedTime(&one, h->launch_events[2 * i], h->launch_events[2 * i + 1]) == hipSuccess) sum += one;
            }
            tm->kernel_sum_ms = sum;
        }
    }
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(h))

// ---- TickFn-compatible shim (cranelift_exec.rs:11,129-195) -----------------------------------------------------

static thread_local sixdof_handle* g_tick_handle = nullptr;

int sixdof_tick_bind(sixdof_handle* h) try {
    g_tick_handle = h;
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(h))

static void tick_slot_ids(const sixdof_handle* h, std::vector<uint64_t>* in, std::vector<uint64_t>* out) {
    // inputs: first-use order of `increment_sim_tick | six_dof(sys)` (system.rs:172-200); then effector columns
    *in = {h->id_tick, h->id_force, h->id_inertia, h->id_pos, h->id_dt, h->id_vel, h->id_accel};
    for (auto& o : h->ops)
        if (o.aux_component_id) in->push_back(o.aux_component_id);
    // outputs: every variable of the builder in ascending ComponentId (BTreeMap, system.rs:139-153)
    std::map<uint64_t, int> ordered;
    for (uint64_t id : *in) ordered[id] = 1;
    out->clear();
    for (auto& kv : ordered) out->push_back(kv.first);
}

static uint64_t slot_bytes(const sixdof_handle* h, uint64_t id) {
    if (id == h->id_tick || id == h->id_dt) return 8;
    const Column* c = h->col(id);
    return c ? c->bytes : 0;
}

int sixdof_tick_slots(const sixdof_handle* h, sixdof_slot* inputs, size_t in_cap, size_t* n_in, sixdof_slot* outputs,
                      size_t out_cap, size_t* n_out) try {
    if (!h || !n_in || !n_out) return SIXDOF_ERR_INVALID_ARGUMENT;
    std::vector<uint64_t> in, out;
    tick_slot_ids(h, &in, &out);
    *n_in = in.size();
    *n_out = out.size();
    if (in_cap < in.size() || out_cap < out.size())
        return h->fail(SIXDOF_ERR_VALUE_SIZE_MISMATCH, "tick_slots: buffer too small");
    for (size_t i = 0; inputs && i < in.size(); i++) inputs[i] = {in[i], slot_bytes(h, in[i])};
    for (size_t i = 0; outputs && i < out.size(); i++) outputs[i] = {out[i], slot_bytes(h, out[i])};
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(h))

void sixdof_tick(const uint8_t* const* inputs, uint8_t* const* outputs) try {
    sixdof_handle* h = g_tick_handle;
    if (!h || !h->bound || !inputs || !outputs) return;  // TickFn cannot fail (cranelift_exec.rs:163-165)
    if (hipSetDevice(h->device) != hipSuccess) return;
    std::vector<uint64_t> in, out;
    tick_slot_ids(h, &in, &out);
    uint64_t tick = 0;
    for (size_t i = 0; i < in.size(); i++) {
        if (in[i] == h->id_tick) std::memcpy(&tick, inputs[i], 8);
        else if (in[i] == h->id_dt) std::memcpy(&h->desc.simulation_time_step, inputs[i], 8);
        else if (Column* c = h->col(in[i])) {
            if (c->bytes) hipMemcpyAsync(c->dev, inputs[i], c->bytes, hipMemcpyHostToDevice, h->stream);
            if (in[i] == h->id_accel) h->accel_is_host_data = true;   // a_in is host data on every TickFn call (rk4.rs:96-100)
            if (c->joined && c->compact)
                launch_gather_rows(c->compact, c->dev, c->d_rows, static_cast<uint32_t>(h->joined_ids.size()),
                                   static_cast<uint32_t>(c->width), c->elem, h->stream);
        }
    }
    h->tick = tick;
    const uint32_t k = h->desc.ticks_per_launch;
    h->desc.ticks_per_launch = 1;
    sixdof_step(h, 1, nullptr);
    h->desc.ticks_per_launch = k;
    for (size_t i = 0; i < out.size(); i++) {
        if (out[i] == h->id_tick) std::memcpy(outputs[i], &h->tick, 8);
        else if (out[i] == h->id_dt) std::memcpy(outputs[i], &h->desc.simulation_time_step, 8);
        else if (Column* c = h->col(out[i])) {
            scatter_back(h, c);
            if (c->bytes) hipMemcpyAsync(outputs[i], c->dev, c->bytes, hipMemcpyDeviceToHost, h->stream);
        }
    }
    hipStreamSynchronize(h->stream);
} SIXDOF_ABI_CATCH_VALUE(err_of(g_tick_handle), )      // TickFn returns nothing: the message is on the bound handle

}  // extern "C"
