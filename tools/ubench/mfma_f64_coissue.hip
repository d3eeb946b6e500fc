// Micro-benchmark for VERDICT r2 item 8: can the n-body force kernel move its three accumulation FMAs per pair-stage
// onto v_mfma_f64_16x16x4_f64 and co-issue them beside the remaining f64 VALU work?
//
//   acc_i = sum_j s_ij p_j - p_i sum_j s_ij :  a [16 targets x 4 sources] . [4 sources x (px, py, pz, 1, 12 x 0)] product
//   per MFMA = 64 pairs, of whose 16 result columns 4 are useful.
//
// Measures, per SIMD with 4 waves resident (like the force kernel): (a) 13 dependent-free v_fma_f64 per "64-pair step"
// (the arithmetic that stays on the VALU: 3 sub, 3 fma, rsq refine, 3 mul), (b) one v_mfma_f64_16x16x4_f64 per step alone,
// (c) both in one loop (co-issue), (d) the current kernel's 17 VALU per step.  ns per step per wave.
// hipcc --offload-arch=gfx950 -O3 mfma_f64_coissue.hip -o mfma_f64_coissue && ./mfma_f64_coissue
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double double4_t __attribute__((ext_vector_type(4)));

template <int N_VALU, bool MFMA>
__global__ __launch_bounds__(256) void k(double* out, int iters, double seed) {
    double a[17];
    for (int i = 0; i < 17; i++) a[i] = seed + threadIdx.x * 1e-3 + i;
    const double b = seed * 0.5 + 1.0, c = seed * 0.25 + 1e-3;
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
    double ma = seed + (threadIdx.x & 15), mb = (threadIdx.x & 15) < 4 ? seed : 0.0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < N_VALU; i++) a[i] = fma(a[i], b, c);
        if (MFMA) {
            // A: this lane's s_ij (depends on the VALU chain like the real kernel: s comes out of the rsq refinement)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(N_VALU ? a[0] : ma, mb, acc, 0, 0, 0);
        }
    }
    double s = acc[0] + acc[1] + acc[2] + acc[3];
    for (int i = 0; i < 17; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int N_VALU, bool MFMA>
double run(const char* name) {
    const int blocks = 256 * 4, iters = 4096;   // 4 blocks of 256 per CU = 4 waves per SIMD
    double* d;
    hipMalloc(&d, blocks * 256 * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<N_VALU, MFMA>), dim3(blocks), dim3(256), 0, 0, d, 16, 1.5);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<N_VALU, MFMA>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / (4.0 * iters);      // per step per wave slot of a SIMD (4 waves interleave)
    printf("%-52s %.3f ms  %7.2f ns per 64-pair step per SIMD\n", name, ms, ns);
    hipFree(d);
    return ns;
}

int main() {
    const double v17 = run<17, false>("17 x v_fma_f64 (today's pair-stage on the VALU)");
    const double v13 = run<13, false>("13 x v_fma_f64 (what stays on the VALU)");
    const double m = run<0, true>("1 x v_mfma_f64_16x16x4_f64 alone");
    const double both = run<13, true>("13 x v_fma_f64 + 1 x v_mfma_f64_16x16x4_f64 (co-issue)");
    printf("\nper pair-stage of 64 pairs: VALU-only %.1f ns; VALU+MFMA %.1f ns -> %.2fx (ideal max(13 VALU, MFMA) = %.1f ns)\n",
           v17, both, v17 / both, v13 > m ? v13 : m);
    printf("note: of the MFMA's 16 result columns 4 carry (sum s px, sum s py, sum s pz, sum s); 12 are padding.\n");
    return 0;
}
