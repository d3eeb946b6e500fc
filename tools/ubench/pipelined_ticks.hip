// Can consecutive ticks of the step launch overlap?  Tick t+1's block b needs only tick t's block b.  Two streams
// alternate ticks (stream A: even, B: odd); inside a stream kernels stay ordered; across streams block b of tick t waits
// on a per-block flag written by block b of tick t-1 (sc0 sc1 payload stores + vmcnt(0) + relaxed agent flag store;
// relaxed agent poll + sc0 sc1 LDS-DMA loads).  Payload geometry = the 65,536-body step launch (12 KiB in, 12.5 KiB out
// per wave) with ~300 dependent f64 FMAs in between.  Checks every word against the sequential result.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(1))) const void* gptr;
typedef __attribute__((address_space(3))) void* lptr;
typedef float vfloat4 __attribute__((ext_vector_type(4)));
constexpr int kBytes = 12288;   // per wave, in place

template <bool PIPE>
__global__ __launch_bounds__(64) void k_tick(char* data, unsigned* flags, unsigned tick, unsigned* err) {
    __shared__ __attribute__((aligned(16))) char lds[kBytes];
    const unsigned b = blockIdx.x, t = threadIdx.x;
    char* base = data + (size_t)b * kBytes;
    if (PIPE) {
        if (t == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(flags + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < tick - 1) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 14)) { atomicAdd(err, 1u); break; }
            }
        }
        __syncthreads();
    }
    for (int i = 0; i < kBytes / 1024; i++)
        __builtin_amdgcn_global_load_lds((gptr)(base + i * 1024 + t * 16), (lptr)(lds + i * 1024), 16, 0, PIPE ? 0x11 : 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    double* mine = reinterpret_cast<double*>(lds) + t * (kBytes / 8 / 64);   // 24 doubles per lane
    double acc[24];
    for (int k = 0; k < 24; k++) acc[k] = mine[k];
    for (int r = 0; r < 12; r++)                                             // ~300 dependent-ish f64 FMAs
        for (int k = 0; k < 24; k++) acc[k] = fma(acc[k], 1.0000001, 1e-7 * (k + 1));
    for (int k = 0; k < 24; k++) mine[k] = acc[k];
    __syncthreads();
    for (int i = 0; i < kBytes / 1024; i++) {
        vfloat4 v = *reinterpret_cast<vfloat4*>(lds + i * 1024 + t * 16);
        char* g = base + i * 1024 + t * 16;
        if (PIPE) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(g), "v"(v) : "memory");
        else *reinterpret_cast<vfloat4*>(g) = v;
    }
    if (PIPE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) __hip_atomic_store(flags + b, tick, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main() {
    const int waves = 1024, ticks = 64, reps = 10;
    const size_t bytes = (size_t)waves * kBytes;
    char* d; unsigned *flags, *err;
    hipMalloc(&d, bytes); hipMalloc(&flags, waves * 4); hipMalloc(&err, 4);
    std::vector<double> h(bytes / 8), ref(bytes / 8);
    for (size_t i = 0; i < h.size(); i++) h[i] = ref[i] = 1.0 + 1e-3 * (i % 977);
    hipStream_t sa, sb; hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    hipEvent_t e0, e1, fork, join; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreateWithFlags(&fork, hipEventDisableTiming); hipEventCreateWithFlags(&join, hipEventDisableTiming);
    for (int mode = 0; mode < 4; mode++) {   // 0: one stream, plain (the product today); 1: one stream, flag protocol; 2: two streams, flag protocol
        hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
        hipMemset(flags, 0, waves * 4); hipMemset(err, 0, 4);
        if (mode == 3) {   // eager launches alternating between two real streams, flags never reset (absolute tick numbers)
            hipEventRecord(e0, sa);
            for (int t = 1; t <= (1 + reps) * ticks; t++)
                hipLaunchKernelGGL(k_tick<true>, dim3(waves), dim3(64), 0, (t & 1) ? sa : sb, d, flags, (unsigned)t, err);
            hipStreamSynchronize(sb);
            hipEventRecord(e1, sa); hipStreamSynchronize(sa);
            float ms3; hipEventElapsedTime(&ms3, e0, e1);
            unsigned herr3 = 0; hipMemcpy(&herr3, err, 4, hipMemcpyDeviceToHost);
            std::vector<double> out3(bytes / 8);
            hipMemcpy(out3.data(), d, bytes, hipMemcpyDeviceToHost);
            size_t bad3 = 0;
            for (size_t i = 0; i < out3.size(); i += 97) {
                double v = h[i];
                const int k = (int)(i % 24);
                for (int n = 0; n < (1 + reps) * ticks * 12; n++) v = fma(v, 1.0000001, 1e-7 * (k + 1));
                if (v != out3[i]) bad3++;
            }
            printf("mode 3 (eager, two streams): %.3f us per tick, spin timeouts %u, wrong words %zu\n", ms3 * 1e3 / ((1 + reps) * ticks), herr3, bad3);
            fflush(stdout);
            continue;
        }
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(sa, hipStreamCaptureModeThreadLocal);
        if (mode == 2) { hipEventRecord(fork, sa); hipStreamWaitEvent(sb, fork, 0); }
        // flags hold the tick count relative to this graph launch: reset at the start of every replay
        if (mode) hipMemsetAsync(flags, 0, waves * 4, sa);
        if (mode == 2) { hipEventRecord(fork, sa); hipStreamWaitEvent(sb, fork, 0); }
        for (int t = 1; t <= ticks; t++) {
            hipStream_t s = (mode == 2 && (t & 1) == 0) ? sb : sa;
            if (mode == 0) hipLaunchKernelGGL(k_tick<false>, dim3(waves), dim3(64), 0, s, d, flags, (unsigned)t, err);
            else hipLaunchKernelGGL(k_tick<true>, dim3(waves), dim3(64), 0, s, d, flags, (unsigned)t, err);
        }
        if (mode == 2) { hipEventRecord(join, sb); hipStreamWaitEvent(sa, join, 0); }
        hipStreamEndCapture(sa, &g);
        if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { printf("instantiate failed\n"); return 1; }
        hipGraphLaunch(ge, sa); hipStreamSynchronize(sa);      // 1 replay = `ticks` ticks
        hipEventRecord(e0, sa);
        for (int r = 0; r < reps; r++) hipGraphLaunch(ge, sa);
        hipEventRecord(e1, sa); hipStreamSynchronize(sa);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned herr = 0; hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
        std::vector<double> out(bytes / 8);
        hipMemcpy(out.data(), d, bytes, hipMemcpyDeviceToHost);
        // sequential reference of (1 + reps) * ticks ticks on a sample of words
        size_t bad = 0;
        for (size_t i = 0; i < out.size(); i += 97) {
            double v = h[i];
            const int k = (int)(i % 24);
            for (int n = 0; n < (1 + reps) * ticks * 12; n++) v = fma(v, 1.0000001, 1e-7 * (k + 1));
            if (v != out[i]) bad++;
        }
        printf("mode %d: %.3f us per tick, spin timeouts %u, wrong words %zu of %zu sampled\n", mode, ms * 1e3 / (reps * ticks), herr, bad, out.size() / 97 + 1);
        fflush(stdout);
    }
    return 0;
}
