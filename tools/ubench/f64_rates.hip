// Micro-benchmark: issue cost (cycles per wave-instruction per SIMD) of the f64 VALU ops the n-body and step
// kernels are made of, on gfx950.  hipcc --offload-arch=gfx950 -O3 f64_rates.hip -o f64_rates && ./f64_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int OP>
__global__ __launch_bounds__(256) void k(double* out, int iters, double seed) {
    double a[8];
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 1e-3 + i;
    const double b = seed * 0.5 + 1.0, c = seed * 0.25 + 1e-3;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) a[i] = fma(a[i], b, c);
            else if (OP == 1) a[i] = a[i] * b;
            else if (OP == 2) a[i] = a[i] + c;
            else if (OP == 3) a[i] = __builtin_amdgcn_rsq(a[i]) + 2.0;      // rsq + add
            else if (OP == 4) a[i] = __builtin_amdgcn_rcp(a[i]) + 2.0;      // rcp + add
            else if (OP == 5) a[i] = __builtin_amdgcn_sqrt(a[i]) + 2.0;     // v_sqrt_f64 + add
            else if (OP == 6) a[i] = 1.0 / a[i] + 2.0;                       // IEEE divide + add
            else if (OP == 7) a[i] = sqrt(a[i]) + 2.0;                       // IEEE sqrt + add
        }
    }
    double s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// Does a wave whose upper 32 lanes are masked off issue an f64 op in half the cycles?  (Decides whether 32-row
// waves, two per SIMD, could overlap their phases for free in the 65,536-body step launch.)
__global__ __launch_bounds__(256) void k_half(double* out, int iters, double seed, int active_lanes) {
    double a[8];
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 1e-3 + i;
    const double b = seed * 0.5 + 1.0, c = seed * 0.25 + 1e-3;
    if ((int)(threadIdx.x & 63) < active_lanes) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] = fma(a[i], b, c);
        }
    }
    double s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
void run_half(int active) {
    const int blocks = 256 * 4, iters = 4096;
    double* d;
    hipMalloc(&d, blocks * 256 * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_half, dim3(blocks), dim3(256), 0, 0, d, 16, 1.5, active);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_half, dim3(blocks), dim3(256), 0, 0, d, iters, 1.5, active);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("v_fma_f64, %2d of 64 lanes active  %.3f ms  %.2f ns per wave-op per SIMD\n", active, ms, ms * 1e6 / (4.0 * iters * 8));
    hipFree(d);
}
template <int OP>
double run(const char* name, int extra_adds) {
    const int blocks = 256 * 4, iters = 4096;   // 4 blocks of 256 per CU = 4 waves per SIMD
    double* d;
    hipMalloc(&d, blocks * 256 * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 16, 1.5);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: 4 waves/SIMD * iters * 8
    const double winstr = 4.0 * iters * 8;
    const double ns_per = ms * 1e6 / winstr;
    printf("%-28s %.3f ms  %.2f ns per wave-op per SIMD (= %.1f cycles @2.4GHz)\n", name, ms, ns_per, ns_per * 2.4);
    hipFree(d);
    return ns_per;
}
int main() {
    run<0>("v_fma_f64", 0);
    run<1>("v_mul_f64", 0);
    run<2>("v_add_f64", 0);
    run<3>("v_rsq_f64 + add", 1);
    run<4>("v_rcp_f64 + add", 1);
    run<5>("v_sqrt_f64 + add", 1);
    run<6>("IEEE 1.0/x + add", 1);
    run<7>("IEEE sqrt(x) + add", 1);
    run_half(64); run_half(32); run_half(16);
    return 0;
}
