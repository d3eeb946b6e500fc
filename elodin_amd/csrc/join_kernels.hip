// join_kernels.hip — device gather / scatter by constant u32 row indices.
//
// Reference semantics (libs/nox-py/src/query.rs:136-208,599-725): a query over several components
// iterates the INTERSECTION of their entity-id sets in ascending id order; each component is gathered
// with a constant u32 index vector baked at compile time and results are scattered back by row
// (`dynamic_update_slice`).  Entities that carry only some of the components (static scene objects with a
// world_pos but no Body, examples/apollo-lander/sim.py:312-332) are never touched by six_dof.
// Here the joined rows live in compact [m,w] device columns that the step kernels update in place;
// gather runs after an upload, scatter before a download, so the per-tick path pays nothing.
#include "kernels.hpp"

namespace sixdof {

template <class E>
__global__ __launch_bounds__(256) void gather_rows_kernel(E* __restrict__ dst, const E* __restrict__ src,
                                                          const uint32_t* __restrict__ rows, uint32_t m, uint32_t w) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)m * w) return;
    const uint32_t r = (uint32_t)(i / w), c = (uint32_t)(i % w);
    dst[i] = src[(uint64_t)rows[r] * w + c];
}

template <class E>
__global__ __launch_bounds__(256) void scatter_rows_kernel(E* __restrict__ dst, const E* __restrict__ src,
                                                           const uint32_t* __restrict__ rows, uint32_t m, uint32_t w) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)m * w) return;
    const uint32_t r = (uint32_t)(i / w), c = (uint32_t)(i % w);
    dst[(uint64_t)rows[r] * w + c] = src[i];
}

hipError_t launch_gather_rows(void* dst, const void* src, const uint32_t* rows, uint32_t m, uint32_t w, size_t elem,
                              hipStream_t s) {
    if (m == 0 || w == 0) return hipSuccess;
    const dim3 grid((unsigned)(((uint64_t)m * w + 255) / 256));
    if (elem == 8) hipLaunchKernelGGL(gather_rows_kernel<uint64_t>, grid, dim3(256), 0, s, (uint64_t*)dst, (const uint64_t*)src, rows, m, w);
    else hipLaunchKernelGGL(gather_rows_kernel<uint32_t>, grid, dim3(256), 0, s, (uint32_t*)dst, (const uint32_t*)src, rows, m, w);
    return hipGetLastError();
}

hipError_t launch_scatter_rows(void* dst, const void* src, const uint32_t* rows, uint32_t m, uint32_t w, size_t elem,
                               hipStream_t s) {
    if (m == 0 || w == 0) return hipSuccess;
    const dim3 grid((unsigned)(((uint64_t)m * w + 255) / 256));
    if (elem == 8) hipLaunchKernelGGL(scatter_rows_kernel<uint64_t>, grid, dim3(256), 0, s, (uint64_t*)dst, (const uint64_t*)src, rows, m, w);
    else hipLaunchKernelGGL(scatter_rows_kernel<uint32_t>, grid, dim3(256), 0, s, (uint32_t*)dst, (const uint32_t*)src, rows, m, w);
    return hipGetLastError();
}

// ---- failure sentinel: rows whose pose or velocity is no longer finite ------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void nonfinite_kernel(const T* __restrict__ pos, const T* __restrict__ vel, uint32_t n,
                                                        uint8_t* __restrict__ flags, unsigned long long* __restrict__ count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool bad = false;
    if (i < n) {
        for (int c = 0; c < 7; c++) bad |= !isfinite(pos[(size_t)i * 7 + c]);
        for (int c = 0; c < 6; c++) bad |= !isfinite(vel[(size_t)i * 6 + c]);
        if (flags) flags[i] = bad ? 1 : 0;
    }
    const unsigned long long m = __ballot(bad);   // 64-lane wave: one atomic per wave
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(count, (unsigned long long)__popcll(m));
}

hipError_t launch_nonfinite(const void* pos, const void* vel, uint32_t n, size_t elem, uint8_t* flags,
                            unsigned long long* count, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const dim3 grid((n + 255) / 256);
    if (elem == 8) hipLaunchKernelGGL(nonfinite_kernel<double>, grid, dim3(256), 0, s, (const double*)pos, (const double*)vel, n, flags, count);
    else hipLaunchKernelGGL(nonfinite_kernel<float>, grid, dim3(256), 0, s, (const float*)pos, (const float*)vel, n, flags, count);
    return hipGetLastError();
}

}  // namespace sixdof
