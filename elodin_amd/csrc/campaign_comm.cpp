// campaign_comm.cpp — campaign-level collectives of a Monte-Carlo run behind the C ABI, on RCCL over xGMI.
//
// The reference runs a campaign as one OS process per rollout and moves nothing between them (libs/monte-carlo/src/
// lib.rs:2083): rank 0 reads plan.csv, every worker gets its row through a context file, result.json files are collected
// at the end.  With rollouts as rows of a GPU column the same two movements are a broadcast of the plan table from rank 0
// and a gather of the result rows in run-id order — once per campaign, no exchange per step.  These entry points let a
// host that is not Python (the Rust runner `north_star` describes) do both without torch: RCCL is loaded with dlopen,
// so the library has no link-time dependency on it and single-GPU hosts never touch it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/sixdof_hip.h"
#include "abi_guard.hpp"

namespace {

// the handful of rccl.h declarations used (ABI-stable since NCCL 2.x): kept local so building needs no RCCL headers
constexpr int kNcclSuccess = 0, kNcclFloat64 = 8, kNcclUint8 = 1;
struct NcclUniqueId { char internal[128]; };
using ncclComm_t = void*;
using GetUniqueIdFn = int (*)(NcclUniqueId*);
using CommInitRankFn = int (*)(ncclComm_t*, int, NcclUniqueId, int);
using CommDestroyFn = int (*)(ncclComm_t);
using BroadcastFn = int (*)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t);
using AllGatherFn = int (*)(const void*, void*, size_t, int, ncclComm_t, hipStream_t);
using ErrorStringFn = const char* (*)(int);

struct Rccl {
    void* dl = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    BroadcastFn broadcast = nullptr;
    AllGatherFn all_gather = nullptr;
    ErrorStringFn error_string = nullptr;
    std::string error;
};

Rccl& rccl() {
    static Rccl r = [] {
        Rccl x;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names)
            if ((x.dl = dlopen(n, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
        if (!x.dl) {
            x.error = std::string("RCCL not found (dlopen librccl.so): ") + (dlerror() ? dlerror() : "");
            return x;
        }
        x.get_unique_id = reinterpret_cast<GetUniqueIdFn>(dlsym(x.dl, "ncclGetUniqueId"));
        x.comm_init_rank = reinterpret_cast<CommInitRankFn>(dlsym(x.dl, "ncclCommInitRank"));
        x.comm_destroy = reinterpret_cast<CommDestroyFn>(dlsym(x.dl, "ncclCommDestroy"));
        x.broadcast = reinterpret_cast<BroadcastFn>(dlsym(x.dl, "ncclBroadcast"));
        x.all_gather = reinterpret_cast<AllGatherFn>(dlsym(x.dl, "ncclAllGather"));
        x.error_string = reinterpret_cast<ErrorStringFn>(dlsym(x.dl, "ncclGetErrorString"));
        if (!x.get_unique_id || !x.comm_init_rank || !x.comm_destroy || !x.broadcast || !x.all_gather)
            x.error = "librccl.so lacks an expected symbol";
        return x;
    }();
    return r;
}

thread_local std::string g_comm_error;

}  // namespace

struct sixdof_comm {
    int world = 1, rank = 0, device = 0;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    void* d_buf = nullptr;      // staging, grown on demand
    size_t d_cap = 0;
    std::string error;

    int fail(int code, const std::string& msg) {
        error = msg;
        g_comm_error = msg;
        return code;
    }
    int hip(hipError_t e, const char* what) {
        return fail(SIXDOF_ERR_BACKEND, std::string(what) + ": " + hipGetErrorString(e));
    }
    int nccl(int rc, const char* what) {
        const Rccl& r = rccl();
        return fail(SIXDOF_ERR_BACKEND, std::string(what) + ": " + (r.error_string ? r.error_string(rc) : "RCCL error") +
                                            " (" + std::to_string(rc) + ")");
    }
    int reserve(size_t bytes) {
        if (bytes <= d_cap) return SIXDOF_OK;
        if (d_buf) hipFree(d_buf), d_buf = nullptr, d_cap = 0;
        hipError_t e = hipMalloc(&d_buf, bytes);
        if (e != hipSuccess) return hip(e, "hipMalloc (staging)");
        d_cap = bytes;
        return SIXDOF_OK;
    }
};

static std::string* err_of(const sixdof_comm* c) { return c ? &const_cast<sixdof_comm*>(c)->error : &g_comm_error; }

extern "C" {

void sixdof_shard_range(uint64_t n_rows, int world, int rank, uint64_t* lo, uint64_t* hi) {
    // contiguous blocks in run-id order, sizes differing by at most one (row = idx, run_id = run_%07d, sample.py:149)
    if (world < 1) world = 1;
    rank = std::min(std::max(rank, 0), world - 1);
    const uint64_t base = n_rows / static_cast<uint64_t>(world), extra = n_rows % static_cast<uint64_t>(world);
    const uint64_t r = static_cast<uint64_t>(rank);
    const uint64_t l = r * base + std::min(r, extra);
    if (lo) *lo = l;
    if (hi) *hi = l + base + (r < extra ? 1 : 0);
}

// ---- the gather's packing, free of any transport: what sixdof_campaign_gather does around its ONE ncclAllGather, and what a
// host with another transport (torch.distributed over gloo in the CPU tests, elodin_amd/shard.py) calls around its own ----
uint64_t sixdof_gather_block_rows(uint64_t n_total, int world) {
    const uint64_t w = static_cast<uint64_t>(world < 1 ? 1 : world);
    return (n_total + w - 1) / w;      // the largest shard: blocks differ by at most one row
}

int sixdof_gather_pack(const double* local_rows, uint64_t n_local, uint64_t width, uint64_t n_total, int world, int rank, double* block) {
    if (world < 1 || rank < 0 || rank >= world || (!local_rows && n_local * width) || (!block && n_total * width)) return SIXDOF_ERR_INVALID_ARGUMENT;
    uint64_t lo = 0, hi = 0;
    sixdof_shard_range(n_total, world, rank, &lo, &hi);
    if (hi - lo != n_local) return SIXDOF_ERR_VALUE_SIZE_MISMATCH;
    const uint64_t pad_rows = sixdof_gather_block_rows(n_total, world);
    if (n_local * width) std::memcpy(block, local_rows, n_local * width * sizeof(double));
    if ((pad_rows - n_local) * width) std::memset(block + n_local * width, 0, (pad_rows - n_local) * width * sizeof(double));
    return SIXDOF_OK;
}

int sixdof_gather_unpack(const double* blocks, uint64_t width, uint64_t n_total, int world, double* all_rows) {
    if (world < 1 || ((!blocks || !all_rows) && n_total * width)) return SIXDOF_ERR_INVALID_ARGUMENT;
    const uint64_t pad_rows = sixdof_gather_block_rows(n_total, world);
    for (int r = 0; r < world; r++) {   // drop the padding: rank r's rows go to [lo_r, hi_r) in run-id order
        uint64_t l = 0, h = 0;
        sixdof_shard_range(n_total, world, r, &l, &h);
        if ((h - l) * width)
            std::memcpy(all_rows + l * width, blocks + static_cast<uint64_t>(r) * pad_rows * width, (h - l) * width * sizeof(double));
    }
    return SIXDOF_OK;
}

int sixdof_comm_unique_id(uint8_t id[SIXDOF_COMM_ID_BYTES]) try {
    if (!id) return SIXDOF_ERR_INVALID_ARGUMENT;
    Rccl& r = rccl();
    if (!r.error.empty()) {
        g_comm_error = r.error;
        return SIXDOF_ERR_UNSUPPORTED;
    }
    NcclUniqueId u;
    const int rc = r.get_unique_id(&u);
    if (rc != kNcclSuccess) {
        g_comm_error = std::string("ncclGetUniqueId: ") + (r.error_string ? r.error_string(rc) : "error");
        return SIXDOF_ERR_BACKEND;
    }
    static_assert(sizeof(u) == SIXDOF_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    std::memcpy(id, &u, sizeof(u));
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(&g_comm_error)

int sixdof_comm_init(sixdof_comm** out, const uint8_t id[SIXDOF_COMM_ID_BYTES], int world, int rank, int device_ordinal) try {
    if (!out || world < 1 || rank < 0 || rank >= world || (world > 1 && !id)) return SIXDOF_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    auto* c = new sixdof_comm();
    c->world = world;
    c->rank = rank;
    c->device = device_ordinal;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device_ordinal < 0 || device_ordinal >= count) {
        (void)hipGetLastError();
        g_comm_error = "comm_init: no such HIP device";
        delete c;
        return SIXDOF_ERR_NO_DEVICE;
    }
    hipError_t e = hipSetDevice(device_ordinal);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        g_comm_error = std::string("comm_init: ") + hipGetErrorString(e);
        delete c;
        return SIXDOF_ERR_BACKEND;
    }
    // a single rank needs no communicator (and no RCCL on the machine); SIXDOF_COMM_FORCE_RCCL=1 builds a one-rank RCCL
    // communicator anyway, so the RCCL code path can be exercised on a one-GPU box (tests)
    const char* force = std::getenv("SIXDOF_COMM_FORCE_RCCL");
    if (world > 1 || (force && force[0] == '1')) {
        Rccl& r = rccl();
        if (!r.error.empty()) {
            g_comm_error = r.error;
            hipStreamDestroy(c->stream);
            delete c;
            return SIXDOF_ERR_UNSUPPORTED;
        }
        NcclUniqueId u;
        if (id) std::memcpy(&u, id, sizeof(u));
        else if (r.get_unique_id(&u) != kNcclSuccess) {   // world == 1 only (checked above): make our own
            g_comm_error = "ncclGetUniqueId failed";
            hipStreamDestroy(c->stream);
            delete c;
            return SIXDOF_ERR_BACKEND;
        }
        const int rc = r.comm_init_rank(&c->comm, world, u, rank);
        if (rc != kNcclSuccess) {
            g_comm_error = std::string("ncclCommInitRank: ") + (r.error_string ? r.error_string(rc) : "error");
            hipStreamDestroy(c->stream);
            delete c;
            return SIXDOF_ERR_BACKEND;
        }
    }
    *out = c;
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(&g_comm_error)

void sixdof_comm_destroy(sixdof_comm* c) try {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->comm) rccl().comm_destroy(c->comm);
    if (c->d_buf) hipFree(c->d_buf);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
} SIXDOF_ABI_CATCH_VALUE(err_of(c), )

const char* sixdof_comm_last_error(const sixdof_comm* c) { return c ? c->error.c_str() : g_comm_error.c_str(); }

int sixdof_campaign_broadcast(sixdof_comm* c, void* table, uint64_t n_bytes, int root) try {
    if (!c || (!table && n_bytes) || root < 0 || root >= c->world) return SIXDOF_ERR_INVALID_ARGUMENT;
    if (c->comm == nullptr || n_bytes == 0) return SIXDOF_OK;   // single rank without a communicator: nothing to move
    hipError_t e = hipSetDevice(c->device);
    if (e != hipSuccess) return c->hip(e, "hipSetDevice");
    int rc = c->reserve(n_bytes);
    if (rc != SIXDOF_OK) return rc;
    if (c->rank == root && (e = hipMemcpyAsync(c->d_buf, table, n_bytes, hipMemcpyHostToDevice, c->stream)) != hipSuccess)
        return c->hip(e, "hipMemcpyAsync (H2D)");
    const int nrc = rccl().broadcast(c->d_buf, c->d_buf, n_bytes, kNcclUint8, root, c->comm, c->stream);
    if (nrc != kNcclSuccess) return c->nccl(nrc, "ncclBroadcast");
    if (c->rank != root && (e = hipMemcpyAsync(table, c->d_buf, n_bytes, hipMemcpyDeviceToHost, c->stream)) != hipSuccess)
        return c->hip(e, "hipMemcpyAsync (D2H)");
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return c->hip(e, "hipStreamSynchronize");
    return SIXDOF_OK;
} SIXDOF_ABI_CATCH(err_of(c))

int sixdof_campaign_gather(sixdof_comm* c, const double* local_rows, uint64_t n_local, uint64_t width, double* all_rows,
                           uint64_t n_total) try {
    if (!c || !all_rows || (!local_rows && n_local)) return SIXDOF_ERR_INVALID_ARGUMENT;
    uint64_t lo = 0, hi = 0;
    sixdof_shard_range(n_total, c->world, c->rank, &lo, &hi);
    if (hi - lo != n_local)
        return c->fail(SIXDOF_ERR_VALUE_SIZE_MISMATCH, "campaign_gather: n_local is not this rank's block of n_total rows (sixdof_shard_range)");
    if (c->comm == nullptr) {   // single rank without a communicator
        if (n_local * width) std::memcpy(all_rows, local_rows, n_local * width * sizeof(double));
        return SIXDOF_OK;
    }
    // equal, padded blocks so ONE ncclAllGather moves everything (sixdof_gather_pack / _unpack: the same packing a host
    // with another transport uses, unit-tested on the CPU for uneven blocks)
    const uint64_t pad_rows = sixdof_gather_block_rows(n_total, c->world);
    const size_t block = static_cast<size_t>(pad_rows * width) * sizeof(double);
    if (block == 0) return SIXDOF_OK;
    hipError_t e = hipSetDevice(c->device);
    if (e != hipSuccess) return c->hip(e, "hipSetDevice");
    int rc = c->reserve(block * (static_cast<size_t>(c->world) + 1));
    if (rc != SIXDOF_OK) return rc;
    char* send = static_cast<char*>(c->d_buf);
    char* recv = send + block;
    std::vector<double> host(static_cast<size_t>(pad_rows * width) * static_cast<size_t>(c->world));
    if ((rc = sixdof_gather_pack(local_rows, n_local, width, n_total, c->world, c->rank, host.data())) != SIXDOF_OK)
        return c->fail(rc, "campaign_gather: sixdof_gather_pack refused the block");
    if ((e = hipMemcpyAsync(send, host.data(), block, hipMemcpyHostToDevice, c->stream)) != hipSuccess) return c->hip(e, "hipMemcpyAsync (H2D)");
    const int nrc = rccl().all_gather(send, recv, pad_rows * width, kNcclFloat64, c->comm, c->stream);
    if (nrc != kNcclSuccess) return c->nccl(nrc, "ncclAllGather");
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return c->hip(e, "hipStreamSynchronize");      // `host` was the H2D source
    if ((e = hipMemcpyAsync(host.data(), recv, block * static_cast<size_t>(c->world), hipMemcpyDeviceToHost, c->stream)) != hipSuccess)
        return c->hip(e, "hipMemcpyAsync (D2H)");
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return c->hip(e, "hipStreamSynchronize");
    return sixdof_gather_unpack(host.data(), width, n_total, c->world, all_rows);
} SIXDOF_ABI_CATCH(err_of(c))

}  // extern "C"
