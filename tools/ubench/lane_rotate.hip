// Micro-benchmark behind the n-body design note ("Newton's third law"): what does it cost to move ONE f64
// per lane to the neighbouring lane, against the f64 VALU work a symmetric pair evaluation would save?
//   (a) v_mov_b32 dpp wave_ror:1 x2 (the only full-wave rotate on gfx9-family hardware)
//   (b) ds_bpermute_b32 x2 (LDS crossbar, no LDS memory)
//   (c) ds_write_b64 + ds_read_b64 through LDS (read-modify-write of a reaction accumulator in a tile)
// each interleaved with independent v_fma_f64 so the issue cost, not the latency, is what is measured.
// hipcc --offload-arch=gfx950 -O3 lane_rotate.hip -o lane_rotate && ./lane_rotate
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ double rot_dpp(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x13C, 0xF, 0xF, false);   // wave_ror:1
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x13C, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rot_bperm(double x, int src_lane4) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_ds_bpermute(src_lane4, lo);
    hi = __builtin_amdgcn_ds_bpermute(src_lane4, hi);
    return __hiloint2double(hi, lo);
}

template <int MODE, int MOVES>   // MOVES f64 values moved per 17 FMAs (one pair-stage evaluation's worth of math)
__global__ __launch_bounds__(256) void k(double* out, int iters, double seed) {
    __shared__ double lds[256 * 9];
    double a[17], m[9];
    for (int i = 0; i < 17; i++) a[i] = seed + threadIdx.x * 1e-3 + i;
    for (int i = 0; i < 9; i++) m[i] = seed * 2 + threadIdx.x + i;
    const double b = seed * 0.5 + 1.0, c = seed * 0.25 + 1e-3;
    const int src4 = (((threadIdx.x & 63) + 1) & 63) * 4;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 17; i++) a[i] = fma(a[i], b, c);
#pragma unroll
        for (int i = 0; i < MOVES; i++) {
            if (MODE == 1) m[i] = rot_dpp(m[i]);
            else if (MODE == 2) m[i] = rot_bperm(m[i], src4);
            else if (MODE == 3) {
                double* p = lds + ((threadIdx.x + it) & 255) * 9 + i;
                *p = *p + m[i];
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 17; i++) s += a[i];
    for (int i = 0; i < 9; i++) s += m[i];
    if (MODE == 3) s += lds[threadIdx.x];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int MOVES>
void run(const char* name) {
    const int blocks = 256 * 4, iters = 2048;
    double* d;
    hipMalloc(&d, blocks * 256 * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, MOVES>), dim3(blocks), dim3(256), 0, 0, d, 16, 1.5);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, MOVES>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %.3f ms  %.1f ns per (17 FMA + %d moves) per wave per SIMD\n", name, ms, ms * 1e6 / (4.0 * iters), MOVES);
    hipFree(d);
}
int main() {
    run<0, 0>("17 v_fma_f64 alone");
    run<1, 3>("+ 3 f64 rotated by DPP wave_ror");
    run<1, 9>("+ 9 f64 rotated by DPP wave_ror");
    run<2, 3>("+ 3 f64 rotated by ds_bpermute");
    run<2, 9>("+ 9 f64 rotated by ds_bpermute");
    run<3, 3>("+ 3 f64 LDS read-modify-write");
    run<3, 9>("+ 9 f64 LDS read-modify-write");
    return 0;
}
