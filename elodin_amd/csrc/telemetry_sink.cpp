// telemetry_sink.cpp — the hand-off of the commit path: per (entity, component) pair one time series, the record layout
// the reference's server loop commits into after every batch of ticks.
//
// Reference: commit_world_head_for_world (libs/nox-py/src/impeller2_server.rs:398-438) walks every component column of the
// world, and for every row pushes `column.buffer[row * size .. (row + 1) * size]` with the batch's end timestamp into the
// time series of PairId = ComponentId::from_pair(entity name, component name) (impeller2/src/types.rs:54-59: the hash of
// "entity.component"); copy_db_to_world (:320-364) goes the other way before a batch — the latest sample of every pair
// overwrites the world's row, which is how a value written by a pre_step / post_step callback (StepContext.write_component) or
// an external controller reaches the simulation.  A time series (libs/db/src/time_series.rs:201-230) is two append logs: an
// index of i64 little-endian timestamps (microseconds) and the samples' bytes; a push older than the last timestamp is
// refused (Error::TimeTravel), equal timestamps append.  Reads: latest(), and the sample with the greatest timestamp <= t
// (StepContext.read_component(timestamp=), elodin.pyi:63-88; past the last write -> the latest).
//
// The database itself (persistence, subscriptions, the wire protocol) is out of scope; this is the in-memory shape its
// writer sees, so that the GPU path's double-buffered column download (sixdof_download_async) has somewhere faithful to land.
// Pure host code (no HIP): usable and tested without a GPU.
#include <algorithm>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/sixdof_hip.h"
#include "abi_guard.hpp"

struct Series {
    std::string name;
    uint32_t elem_bytes = 0;
    std::vector<int64_t> index;    // AppendLog<Timestamp>
    std::vector<uint8_t> data;     // AppendLog<u64> of the samples' bytes
};

struct sixdof_sink {
    std::map<uint64_t, Series> series;
    std::string err;
    uint64_t commits = 0;
};

static std::string* err_of(const sixdof_sink* s) { return s ? &const_cast<sixdof_sink*>(s)->err : nullptr; }

static int fail(sixdof_sink* s, int code, const std::string& msg) {
    if (s) s->err = msg;
    return code;
}

extern "C" {

uint64_t sixdof_pair_id(const char* entity, const char* component) try {
    // ComponentId::from_pair: the component id of "entity.component" (types.rs:54-59)
    const std::string joined = std::string(entity ? entity : "") + "." + (component ? component : "");
    return sixdof_component_id(joined.c_str());
} SIXDOF_ABI_CATCH_VALUE(nullptr, 0)

sixdof_sink* sixdof_sink_create(void) { return new (std::nothrow) sixdof_sink(); }
void sixdof_sink_destroy(sixdof_sink* s) { delete s; }
const char* sixdof_sink_last_error(const sixdof_sink* s) { return s ? s->err.c_str() : "null sink"; }

int sixdof_sink_register(sixdof_sink* s, uint64_t pair_id, uint32_t elem_bytes, const char* name) try {
    if (!s || elem_bytes == 0) return fail(s, SIXDOF_ERR_INVALID_ARGUMENT, "sink_register: null sink or empty element");
    auto it = s->series.find(pair_id);
    if (it != s->series.end()) {
        if (it->second.elem_bytes != elem_bytes)
            return fail(s, SIXDOF_ERR_VALUE_SIZE_MISMATCH, "sink_register: pair " + it->second.name + " exists with another element size");
        return SIXDOF_OK;
    }
    Series fresh;                                   // filled before it enters the map: a failed allocation leaves no half-made pair
    fresh.name = name ? name : "";
    fresh.elem_bytes = elem_bytes;
    s->series.emplace(pair_id, std::move(fresh));
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(s))

int sixdof_sink_push(sixdof_sink* s, uint64_t pair_id, int64_t timestamp_us, const void* buf, uint32_t bytes) try {
    if (!s || !buf) return fail(s, SIXDOF_ERR_INVALID_ARGUMENT, "sink_push: null argument");
    auto it = s->series.find(pair_id);
    if (it == s->series.end()) return fail(s, SIXDOF_ERR_COMPONENT_NOT_FOUND, "sink_push: pair is not registered");
    Series& t = it->second;
    if (bytes != t.elem_bytes) return fail(s, SIXDOF_ERR_VALUE_SIZE_MISMATCH, "sink_push: " + t.name + ": sample size differs from the pair's element");
    if (!t.index.empty() && t.index.back() > timestamp_us)      // time_series.rs:206-222
        return fail(s, SIXDOF_ERR_TIME_TRAVEL, "sink_push: " + t.name + ": time travel (timestamp older than the last sample)");
    const uint8_t* b = static_cast<const uint8_t*>(buf);
    t.index.reserve(t.index.size() + 1);            // whatever can throw throws before a byte changes: a failed push leaves the series as it was
    t.data.insert(t.data.end(), b, b + bytes);      // data first, index last (consistent reads, :224-228)
    t.index.push_back(timestamp_us);
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(s))

uint64_t sixdof_sink_sample_count(const sixdof_sink* s, uint64_t pair_id) try {
    if (!s) return 0;
    auto it = s->series.find(pair_id);
    return it == s->series.end() ? 0 : it->second.index.size();
} SIXDOF_ABI_CATCH_VALUE(err_of(s), 0)

size_t sixdof_sink_pairs(const sixdof_sink* s, uint64_t* ids, size_t cap) try {
    if (!s) return 0;
    size_t n = 0;
    for (const auto& kv : s->series) {
        if (ids && n < cap) ids[n] = kv.first;
        n++;
    }
    return n;
} SIXDOF_ABI_CATCH_VALUE(err_of(s), 0)

int sixdof_sink_latest(const sixdof_sink* s, uint64_t pair_id, int64_t* timestamp_us, void* out, uint32_t bytes) try {
    if (!s) return SIXDOF_ERR_INVALID_ARGUMENT;
    auto it = s->series.find(pair_id);
    if (it == s->series.end() || it->second.index.empty()) return SIXDOF_ERR_COMPONENT_NOT_FOUND;
    const Series& t = it->second;
    if (out && bytes != t.elem_bytes) return SIXDOF_ERR_VALUE_SIZE_MISMATCH;
    const size_t k = t.index.size() - 1;
    if (timestamp_us) *timestamp_us = t.index[k];
    if (out) std::memcpy(out, t.data.data() + k * t.elem_bytes, t.elem_bytes);
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(s))

int sixdof_sink_at(const sixdof_sink* s, uint64_t pair_id, int64_t timestamp_us, int64_t* found_us, void* out, uint32_t bytes) try {
    if (!s) return SIXDOF_ERR_INVALID_ARGUMENT;
    auto it = s->series.find(pair_id);
    if (it == s->series.end() || it->second.index.empty()) return SIXDOF_ERR_COMPONENT_NOT_FOUND;
    const Series& t = it->second;
    if (out && bytes != t.elem_bytes) return SIXDOF_ERR_VALUE_SIZE_MISMATCH;
    // the sample with the greatest timestamp <= the requested one (the LAST of equals); past the end: the latest
    auto up = std::upper_bound(t.index.begin(), t.index.end(), timestamp_us);
    if (up == t.index.begin()) return SIXDOF_ERR_COMPONENT_NOT_FOUND;      // before the first sample: nothing to hold
    const size_t k = static_cast<size_t>(up - t.index.begin()) - 1;
    if (found_us) *found_us = t.index[k];
    if (out) std::memcpy(out, t.data.data() + k * t.elem_bytes, t.elem_bytes);
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(s))

int sixdof_sink_series(const sixdof_sink* s, uint64_t pair_id, const int64_t** timestamps, const uint8_t** data, uint64_t* n,
                       uint32_t* elem_bytes) try {
    if (!s) return SIXDOF_ERR_INVALID_ARGUMENT;
    auto it = s->series.find(pair_id);
    if (it == s->series.end()) return SIXDOF_ERR_COMPONENT_NOT_FOUND;
    const Series& t = it->second;
    if (timestamps) *timestamps = t.index.data();
    if (data) *data = t.data.data();
    if (n) *n = t.index.size();
    if (elem_bytes) *elem_bytes = t.elem_bytes;
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(s))

void sixdof_sink_truncate(sixdof_sink* s) try {      // TimeSeries::truncate: samples go, the schema stays
    if (!s) return;
    for (auto& kv : s->series) {
        kv.second.index.clear();
        kv.second.data.clear();
    }
} SIXDOF_ABI_CATCH_VALUE(err_of(s), )

int sixdof_sink_commit_rows(sixdof_sink* s, const uint64_t* pair_ids, const void* rows, uint32_t n_rows, uint32_t row_bytes,
                            int64_t timestamp_us) try {
    // commit_world_head_for_world for ONE column: row i -> the series of pair_ids[i]; pair id 0 = that entity has no name
    // in the metadata (the reference skips it, :418-420); an unregistered pair is skipped too (:432-434)
    if (!s || !pair_ids || (!rows && n_rows)) return fail(s, SIXDOF_ERR_INVALID_ARGUMENT, "sink_commit_rows: null argument");
    const uint8_t* b = static_cast<const uint8_t*>(rows);
    for (uint32_t i = 0; i < n_rows; i++) {
        if (pair_ids[i] == 0) continue;
        auto it = s->series.find(pair_ids[i]);
        if (it == s->series.end()) continue;
        int rc = sixdof_sink_push(s, pair_ids[i], timestamp_us, b + static_cast<size_t>(i) * row_bytes, row_bytes);
        if (rc != SIXDOF_OK) return rc;
    }
    s->commits++;
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(s))

int sixdof_sink_copy_to_rows(const sixdof_sink* s, const uint64_t* pair_ids, void* rows, uint32_t n_rows, uint32_t row_bytes,
                             int* changed) try {
    // copy_db_to_world for ONE column: the latest sample of every pair overwrites its row; *changed = some byte differed
    // (the reference marks the component dirty then, :353-361)
    if (!s || !pair_ids || (!rows && n_rows)) return SIXDOF_ERR_INVALID_ARGUMENT;
    uint8_t* b = static_cast<uint8_t*>(rows);
    int diff = 0;
    for (uint32_t i = 0; i < n_rows; i++) {
        if (pair_ids[i] == 0) continue;
        auto it = s->series.find(pair_ids[i]);
        if (it == s->series.end() || it->second.index.empty() || it->second.elem_bytes != row_bytes) continue;
        const Series& t = it->second;
        const uint8_t* head = t.data.data() + (t.index.size() - 1) * t.elem_bytes;
        uint8_t* dst = b + static_cast<size_t>(i) * row_bytes;
        if (std::memcmp(dst, head, row_bytes) != 0) {
            diff = 1;
            std::memcpy(dst, head, row_bytes);
        }
    }
    if (changed) *changed = diff;
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(s))

}  // extern "C"
