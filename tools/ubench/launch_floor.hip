// Where do the ~2.8 us of fixed per-launch time of a small step launch go?  Graph-replayed chains of
// (a) an empty kernel, (b) empty kernel with a 700-byte argument block, (c) one wave per SIMD doing the step
// kernel's LDS-DMA loads only, (d) loads + stores of the same 384/200 B per lane, no math.
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { double v[88]; };
__global__ void k_empty() {}
__global__ void k_bigarg(Big b, double* out) { if (b.v[3] == 12345.678) out[0] = b.v[7]; }
typedef __attribute__((address_space(1))) const void* gptr;
typedef __attribute__((address_space(3))) void* lptr;
template <bool STORE>
__global__ __launch_bounds__(64) void k_move(const char* in, char* out, Big b) {
    __shared__ __attribute__((aligned(16))) char lds[64 * 200];
    const size_t base = (size_t)blockIdx.x * 64 * 200;
    for (int i = 0; i < 12; i++)   // 12 KiB per wave ~ pos+vel+inertia+aux slabs
        __builtin_amdgcn_global_load_lds((gptr)(in + base + i * 1024 + threadIdx.x * 16), (lptr)(lds + i * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (STORE) {
        for (int i = 0; i < 12; i++) {
            float4 v = *reinterpret_cast<float4*>(lds + i * 1024 + threadIdx.x * 16);
            *reinterpret_cast<float4*>(out + base + i * 1024 + threadIdx.x * 16) = v;
        }
    } else if (b.v[1] == 4.25) out[threadIdx.x] = lds[threadIdx.x];
}
template <class F>
double chain(const char* name, F launch, int len = 64, int reps = 50) {
    hipStream_t s; hipStreamCreate(&s);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < len; i++) launch(s);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int r = 0; r < reps; r++) hipGraphLaunch(ge, s);
    hipEventRecord(e1, s); hipStreamSynchronize(s);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %.3f us per launch\n", name, ms * 1e3 / (len * reps));
    return ms;
}
int main() {
    const int waves = 1024;
    char *in, *out; double* d;
    hipMalloc(&in, (size_t)waves * 64 * 200); hipMalloc(&out, (size_t)waves * 64 * 200); hipMalloc(&d, 4096);
    hipMemset(in, 0, (size_t)waves * 64 * 200);
    Big b{}; b.v[3] = 1.0;
    chain("empty kernel, 1 block", [&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); });
    chain("empty kernel, 1024 blocks", [&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(waves), dim3(64), 0, s); });
    chain("700 B kernarg, 1024 blocks", [&](hipStream_t s) { hipLaunchKernelGGL(k_bigarg, dim3(waves), dim3(64), 0, s, b, d); });
    chain("LDS-DMA loads only (12 KiB/wave)", [&](hipStream_t s) { hipLaunchKernelGGL(k_move<false>, dim3(waves), dim3(64), 0, s, in, out, b); });
    chain("loads + stores (12+12 KiB/wave)", [&](hipStream_t s) { hipLaunchKernelGGL(k_move<true>, dim3(waves), dim3(64), 0, s, in, out, b); });
    chain("loads + stores, in place", [&](hipStream_t s) { hipLaunchKernelGGL(k_move<true>, dim3(waves), dim3(64), 0, s, in, in, b); });
    return 0;
}
