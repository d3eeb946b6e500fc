// What does a lone wave per SIMD wait for?  The generated Falcon 9 kernel spends 20 % of its wave cycles in SQ_WAIT_ANY with
// 8 memory instructions per tick (profiles/r04_falcon9_instruction_mix.md).  Six loops of register-only code, 512 single-wave
// workgroups each (one wave on half the SIMDs, like the 32,768-rollout campaign), for a rocprofv3 --pmc pass
// (SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU):
//   dep_fma      one dependent chain of v_fma_f32                       (issue every 4 clocks, result latency hidden?)
//   ilp_fma      four independent chains interleaved
//   cmp_select   v_cmp (writes an SGPR pair) -> v_cndmask reading it    (VALU -> SGPR -> VALU hazard)
//   trans        dependent v_rcp_f32 + v_mul chains                      (quarter-rate transcendental unit)
//   mask_logic   v_cmp, v_cmp, s_and_b64, v_cndmask                      (bool logic on the scalar unit between vector ops)
//   any_branch   v_cmp, s_cbranch on the wave-wide ballot every 32 instructions (taken: refetch)
// hipcc --offload-arch=gfx950 -O3 issue_mix.hip -o issue_mix && ./issue_mix
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kIters = 20000;

__global__ __launch_bounds__(64) void dep_fma(float* out, float k) {
    float x = out[threadIdx.x];
    for (int i = 0; i < kIters; i++) {
#pragma unroll
        for (int j = 0; j < 64; j++) x = fmaf(x, 0.999f, k);
    }
    out[blockIdx.x * 64 + threadIdx.x] = x;
}
__global__ __launch_bounds__(64) void ilp_fma(float* out, float k) {
    float a = out[threadIdx.x], b = a + 1.0f, c = a + 2.0f, d = a + 3.0f;
    for (int i = 0; i < kIters; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) { a = fmaf(a, 0.999f, k); b = fmaf(b, 0.998f, k); c = fmaf(c, 0.997f, k); d = fmaf(d, 0.996f, k); }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d;
}
__global__ __launch_bounds__(64) void cmp_select(float* out, float k) {
    float x = out[threadIdx.x], y = x + 1.0f;
    for (int i = 0; i < kIters; i++) {
#pragma unroll
        for (int j = 0; j < 32; j++) { const bool c = x < y; const float t = c ? x + k : y; y = x; x = t; }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x + y;
}
__global__ __launch_bounds__(64) void trans(float* out, float k) {
    float x = out[threadIdx.x] + 2.0f;
    for (int i = 0; i < kIters; i++) {
#pragma unroll
        for (int j = 0; j < 32; j++) x = __builtin_amdgcn_rcpf(x) * k + 1.5f;
    }
    out[blockIdx.x * 64 + threadIdx.x] = x;
}
__global__ __launch_bounds__(64) void mask_logic(float* out, float k) {
    float x = out[threadIdx.x], y = x + 1.0f, z = x - 1.0f;
    for (int i = 0; i < kIters; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) { const bool c = (x < y) && (z < x) || (y < k); const float t = c ? x + k : z; z = y; y = x; x = t; }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x + y + z;
}
__global__ __launch_bounds__(64) void any_branch(float* out, float k) {
    float x = out[threadIdx.x];
    for (int i = 0; i < kIters; i++) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            if (__any(x > 1.0e30f)) x = sqrtf(x) * k;      // never true: the branch around it is taken every time
#pragma unroll
            for (int m = 0; m < 31; m++) x = fmaf(x, 0.999f, k);
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x;
}

int main() {
    float* d;
    hipMalloc(&d, 512 * 64 * sizeof(float));
    hipMemset(d, 0, 512 * 64 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto kern, int per_iter) {
        hipLaunchKernelGGL(kern, dim3(512), dim3(64), 0, 0, d, 1.0e-3f);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(512), dim3(64), 0, 0, d, 1.0e-3f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-11s %8.3f ms  %6.2f ns per loop body of ~%d vector instructions = %5.2f ns each\n", name, ms, ms * 1e6 / kIters, per_iter,
               ms * 1e6 / kIters / per_iter);
    };
    run("dep_fma", dep_fma, 64); run("ilp_fma", ilp_fma, 64); run("cmp_select", cmp_select, 96); run("trans", trans, 64);
    run("mask_logic", mask_logic, 96); run("any_branch", any_branch, 64);
    return 0;
}
