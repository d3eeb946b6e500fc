// What does a packed single-precision instruction cost a lone wave per SIMD?  The generated Falcon 9 campaign kernel is bound by
// VALU issue (profiles/r04_falcon9_instruction_mix.md: 1,500 VALU per tick, 81 % of wave cycles issuing); gfx950's
// v_pk_{fma,mul,add}_f32 do two f32 operations per lane.  If one packed instruction takes the issue slot of one scalar
// instruction, every pair of independent same-op nodes the generator (or LLVM's SLP vectoriser) packs saves a slot.
// Register-only loops, one single-wave workgroup per SIMD (1,024) and per half of them (512, the 32,768-rollout campaign):
//   scalar_dep     one dependent chain of v_fma_f32
//   scalar_pair    TWO independent chains of v_fma_f32 interleaved       (what a pair of nodes costs unpacked)
//   packed_dep     one dependent chain of v_pk_fma_f32                   (the same pair as one instruction)
//   packed_pair    two independent chains of v_pk_fma_f32                (four scalar chains' worth)
//   the same for mul and add; `mixed` = pk_fma whose operands were last written by scalar v_fma_f32 (the move / hazard cost of
//   going in and out of a packed pair: operands of a packed instruction are 64-bit aligned register pairs)
// hipcc --offload-arch=gfx950 -O3 pk_f32.hip -o pk_f32 && ./pk_f32
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float float2_ __attribute__((ext_vector_type(2)));
constexpr int kIters = 20000;
constexpr int kUnroll = 64;

#define SCALAR_OP(name, stmt)                                                                      \
    __global__ __launch_bounds__(64) void name##_dep(float* out, float k) {                       \
        float x = out[threadIdx.x] + 1.0f;                                                        \
        for (int i = 0; i < kIters; i++) {                                                        \
            _Pragma("unroll") for (int j = 0; j < kUnroll; j++) { stmt(x) }                       \
        }                                                                                          \
        out[blockIdx.x * 64 + threadIdx.x] = x;                                                   \
    }                                                                                              \
    __global__ __launch_bounds__(64) void name##_pair(float* out, float k) {                      \
        float x = out[threadIdx.x] + 1.0f, y = x + 0.5f;                                          \
        for (int i = 0; i < kIters; i++) {                                                        \
            _Pragma("unroll") for (int j = 0; j < kUnroll / 2; j++) { stmt(x) stmt(y) }           \
        }                                                                                          \
        out[blockIdx.x * 64 + threadIdx.x] = x + y;                                               \
    }
#define PACKED_OP(name, stmt)                                                                      \
    __global__ __launch_bounds__(64) void name##_dep(float* out, float k) {                       \
        float2_ x = {out[threadIdx.x] + 1.0f, out[threadIdx.x] + 1.5f};                           \
        const float2_ c = {0.999f, 0.998f}, kk = {k, k};                                          \
        for (int i = 0; i < kIters; i++) {                                                        \
            _Pragma("unroll") for (int j = 0; j < kUnroll; j++) { stmt(x) }                       \
        }                                                                                          \
        out[blockIdx.x * 64 + threadIdx.x] = x.x + x.y;                                           \
    }                                                                                              \
    __global__ __launch_bounds__(64) void name##_pair(float* out, float k) {                      \
        float2_ x = {out[threadIdx.x] + 1.0f, out[threadIdx.x] + 1.5f}, y = x + 0.25f;            \
        const float2_ c = {0.999f, 0.998f}, kk = {k, k};                                          \
        for (int i = 0; i < kIters; i++) {                                                        \
            _Pragma("unroll") for (int j = 0; j < kUnroll / 2; j++) { stmt(x) stmt(y) }           \
        }                                                                                          \
        out[blockIdx.x * 64 + threadIdx.x] = x.x + x.y + y.x + y.y;                               \
    }

// inline asm keeps the compiler from re-associating, re-packing or un-packing the chains
#define S_FMA(v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(0.999f), "v"(k));
#define S_MUL(v) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v) : "v"(0.9999f));
#define S_ADD(v) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v) : "v"(k));
#define P_FMA(v) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(c), "v"(kk));
#define P_MUL(v) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v) : "v"(c));
#define P_ADD(v) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v) : "v"(kk));
SCALAR_OP(s_fma, S_FMA)
SCALAR_OP(s_mul, S_MUL)
SCALAR_OP(s_add, S_ADD)
PACKED_OP(p_fma, P_FMA)
PACKED_OP(p_mul, P_MUL)
PACKED_OP(p_add, P_ADD)

// a packed instruction fed by scalar results: lo / hi halves written by two v_fma_f32, then one v_pk_fma_f32 over the pair
__global__ __launch_bounds__(64) void mixed(float* out, float k) {
    float2_ x = {out[threadIdx.x] + 1.0f, out[threadIdx.x] + 1.5f};
    const float2_ c = {0.999f, 0.998f}, kk = {k, k};
    for (int i = 0; i < kIters; i++) {
#pragma unroll
        for (int j = 0; j < kUnroll / 4; j++) {
            float lo = x.x, hi = x.y;
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(lo) : "v"(0.999f), "v"(k));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(hi) : "v"(0.998f), "v"(k));
            x.x = lo; x.y = hi;
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(kk));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(kk));
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x.x + x.y;
}

int main() {
    float* d;
    hipMalloc(&d, 2048 * 64 * sizeof(float));
    hipMemset(d, 0, 2048 * 64 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    printf("device clock attribute: %d kHz; %d iterations x %d instructions per wave\n", clk_khz, kIters, kUnroll);
    printf("%-14s %6s %10s %12s %14s %16s\n", "kernel", "waves", "ms", "ns/instr", "clk/instr@attr", "f32 ops/instr");
    auto run = [&](const char* name, auto kern, int waves, int flops_per_instr) {
        hipLaunchKernelGGL(kern, dim3(waves), dim3(64), 0, 0, d, 1.0e-3f);
        hipDeviceSynchronize();
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(waves), dim3(64), 0, 0, d, 1.0e-3f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        const double ns = best * 1e6 / (double(kIters) * kUnroll);
        printf("%-14s %6d %10.3f %12.3f %14.2f %16d\n", name, waves, best, ns, ns * clk_khz * 1e-6, flops_per_instr);
    };
    for (int waves : {512, 1024, 2048}) {
        run("s_fma_dep", s_fma_dep, waves, 1);   run("s_fma_pair", s_fma_pair, waves, 1);
        run("p_fma_dep", p_fma_dep, waves, 2);   run("p_fma_pair", p_fma_pair, waves, 2);
        run("s_mul_dep", s_mul_dep, waves, 1);   run("s_mul_pair", s_mul_pair, waves, 1);
        run("p_mul_dep", p_mul_dep, waves, 2);   run("p_mul_pair", p_mul_pair, waves, 2);
        run("s_add_dep", s_add_dep, waves, 1);   run("s_add_pair", s_add_pair, waves, 1);
        run("p_add_dep", p_add_dep, waves, 2);   run("p_add_pair", p_add_pair, waves, 2);
        run("mixed", mixed, waves, 0);
    }
    return 0;
}
