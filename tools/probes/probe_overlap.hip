// Do MFMA and VALU work of DIFFERENT waves on one SIMD overlap on gfx950?  Two waves per SIMD (512-thread blocks, one block
// per CU): the first four waves run `mfma` back-to-back MFMAs, the last four `valu` v_exp_f32 / v_fma_f32 (or both kinds do
// the same work).  Also the same mix inside ONE wave (independent instructions interleaved).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/probe_overlap.hip -o probe_overlap && ./probe_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define REP8(x) x x x x x x x x

__device__ __forceinline__ void mfma_block(f4 (&c)[8], h8 a, h8 b) {          // 32 MFMAs
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[i], 0, 0, 0);
}
template <int KIND>
__device__ __forceinline__ void valu_block(float (&x)[8]) {                   // 32 VALU instructions
    if (KIND == 0) { REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));) }
    else { REP8(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                             : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : "v"(x[4]), "v"(x[5]));) }
}

// MODE 0: waves 0-3 MFMA, waves 4-7 idle; 1: waves 0-3 idle, 4-7 VALU; 2: 0-3 MFMA + 4-7 VALU; 3: every wave alternates 32 MFMA / 32 VALU (independent);
//      4: every wave MFMA only (2 waves/SIMD); 5: every wave VALU only
template <int MODE, int KIND>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
    f4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = f4{0.f, 0.f, 0.f, 0.f};
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = -0.001f * (threadIdx.x + i);
    const bool do_m = MODE == 0 ? wave < 4 : MODE == 1 ? false : MODE == 2 ? wave < 4 : MODE == 5 ? false : true;
    const bool do_v = MODE == 0 ? false : MODE == 1 ? wave >= 4 : MODE == 2 ? wave >= 4 : MODE == 4 ? false : true;
    for (int it = 0; it < iters; ++it) {
        if (do_m) mfma_block(c, a, b);
        if (do_v) valu_block<KIND>(x);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][3] + x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// fine-grained: G MFMAs then G v_exp_f32 (independent registers), 32 of each per block; ACTIVE = waves per SIMD doing it (1 or 2)
template <int G, int ACTIVE>
__global__ __launch_bounds__(512) void kf(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
    f4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = f4{0.f, 0.f, 0.f, 0.f};
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = -0.001f * (threadIdx.x + i);
    if (ACTIVE == 1 && wave >= 4) { out[blockIdx.x * blockDim.x + threadIdx.x] = 0.f; return; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 32 / G; ++g) {
#pragma unroll
            for (int j = 0; j < G; ++j) { const int i = (g * G + j) & 7; c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[i], 0, 0, 0); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < G; ++j) { const int i = (g * G + j) & 7; asm volatile("v_exp_f32 %0, %0" : "+v"(x[i])); }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][3] + x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    const int iters = 20000;
    auto run = [&](const char* name, auto kern) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, out, 100);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, out, iters);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-72s %8.3f ms  = %7.1f ns per (32 MFMA | 32 VALU) block\n", name, ms, ms * 1e6 / iters);
    };
    run("one wave/SIMD MFMA, the other idle", k<0, 0>);
    run("one wave/SIMD v_exp_f32, the other idle", k<1, 0>);
    run("one wave/SIMD MFMA + the other v_exp_f32", k<2, 0>);
    run("one wave/SIMD v_fma_f32, the other idle", k<1, 1>);
    run("one wave/SIMD MFMA + the other v_fma_f32", k<2, 1>);
    run("both waves: 32 MFMA then 32 v_exp_f32, alternating", k<3, 0>);
    run("both waves: 32 MFMA then 32 v_fma_f32, alternating", k<3, 1>);
    run("both waves MFMA only", k<4, 0>);
    run("both waves v_exp_f32 only", k<5, 0>);
    run("both waves v_fma_f32 only", k<5, 1>);
    run("ONE wave/SIMD: 1 MFMA, 1 exp, 1 MFMA, ...", kf<1, 1>);
    run("ONE wave/SIMD: 2 MFMA, 2 exp, ...", kf<2, 1>);
    run("ONE wave/SIMD: 4 MFMA, 4 exp, ...", kf<4, 1>);
    run("ONE wave/SIMD: 8 MFMA, 8 exp, ...", kf<8, 1>);
    run("both waves: 1 MFMA, 1 exp, ...", kf<1, 2>);
    run("both waves: 2 MFMA, 2 exp, ...", kf<2, 2>);
    run("both waves: 4 MFMA, 4 exp, ...", kf<4, 2>);
    run("both waves: 8 MFMA, 8 exp, ...", kf<8, 2>);
    run("both waves: 16 MFMA, 16 exp, ...", kf<16, 2>);
    return 0;
}
