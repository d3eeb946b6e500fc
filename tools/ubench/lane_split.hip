// Micro-benchmark behind the round-4 design note ("splitting one rollout over lanes or waves"): can the two
// halves of ONE rollout's tick — say aerodynamics and the engine cluster, independent given the state — run side by side?
//   (a) one wave, every lane runs chain A then chain B                         = today's generated kernel (one lane, one rollout)
//   (b) one wave, lanes 0..31 run chain A, lanes 32..63 run chain B           = "split a rollout over 2 lanes": the branches are
//       DIFFERENT code, so the wave executes both with half its lanes masked   -> expected: same time as (a)
//   (c) one wave, lanes 0..31 active only, A then B                           = a half-filled wave (32 rollouts): same time as (a)
//   (d) two waves of one workgroup, wave 0 runs A, wave 1 runs B, results exchanged through LDS + s_barrier every `sync`
//       chain steps                                                           = "split a rollout over 2 waves": the only split
//       that shortens the tick; its price is the exchange
// A and B are dependent f32 FMA chains with different constants (so the compiler cannot merge the branches).
// hipcc --offload-arch=gfx950 -O3 lane_split.hip -o lane_split && ./lane_split
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kChain = 256;   // dependent FMAs per chain per iteration (~ a quarter of a Falcon 9 tick's critical path)

__device__ __forceinline__ float chain_a(float x, float k) {
#pragma unroll
    for (int i = 0; i < kChain; i++) x = fmaf(x, 0.999f, k);
    return x;
}
__device__ __forceinline__ float chain_b(float x, float k) {
#pragma unroll
    for (int i = 0; i < kChain; i++) x = fmaf(x, 1.001f, -k);
    return x;
}

template <int MODE>
__global__ __launch_bounds__(128) void k(float* out, int iters, float seed) {
    __shared__ float xch[2][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float a = seed + lane * 1e-3f, b = seed * 0.5f + lane * 1e-3f;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {                       // (a)
            a = chain_a(a, b * 1e-6f);
            b = chain_b(b, a * 1e-6f);
        } else if (MODE == 1) {                // (b) divergent halves
            if (lane < 32) a = chain_a(a, b * 1e-6f);
            else b = chain_b(b, a * 1e-6f);
        } else if (MODE == 2) {                // (c) half-filled wave
            if (lane < 32) {
                a = chain_a(a, b * 1e-6f);
                b = chain_b(b, a * 1e-6f);
            }
        } else {                               // (d) two waves, one chain each, exchange through LDS
            if (wave == 0) a = chain_a(a, b * 1e-6f);
            else b = chain_b(b, a * 1e-6f);
            xch[wave][lane] = wave == 0 ? a : b;
            __syncthreads();
            if (wave == 0) b = xch[1][lane];
            else a = xch[0][lane];
            __syncthreads();
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b;
}

template <int MODE>
void run(const char* name, int threads) {
    const int blocks = 512, iters = 4096;      // 512 workgroups: at most one per CU pair, no two waves share a SIMD
    float* d;
    hipMalloc(&d, blocks * 128 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(threads), 0, 0, d, 16, 1.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(threads), 0, 0, d, iters, 1.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-78s %.3f ms  %.1f ns per iteration (A: %d FMA, B: %d FMA)\n", name, ms, ms * 1e6 / iters, kChain, kChain);
    hipFree(d);
}
int main() {
    run<0>("(a) one wave, all 64 lanes run A then B", 64);
    run<1>("(b) one wave, lanes 0-31 run A, lanes 32-63 run B (divergent)", 64);
    run<2>("(c) one wave, lanes 0-31 only, A then B (half-filled wave)", 64);
    run<3>("(d) two waves of a workgroup, wave 0 runs A, wave 1 runs B, LDS exchange + 2 barriers", 128);
    return 0;
}
