// Would the head_dim-40 self-attention kernel (attention_pipe.hip) gain from taking S^T = K Q^T on v_mfma_f32_32x32x16_f16 — k padded
// 40 -> 48 (three k steps of 16) instead of 40 -> 64 (two k steps of 32 on v_mfma_f32_16x16x32_f16)?  (probe first.)
// A wave's key tile is 64 keys x 32 queries:
//   A  today:     16 x mfma_16x16x32 (4 key blocks x 2 query blocks x 2 k steps)  + the 12 PV MFMAs (16x16x32)
//   B  proposed:   6 x mfma_32x32x16 (2 key blocks x 1 query block x 3 k steps)   + the 12 PV MFMAs (16x16x32)
// MFMA-only cost of one tile per SIMD with 1 / 2 / 3 waves per SIMD (register operands, no LDS, no softmax), and the same with the
// tile's 32 v_exp_f32 + 16 v_cvt_pk + 16 v_max3 interleaved one per MFMA (the real kernel's VALU mix).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/probe_qk32.hip -o probe_qk32 && ./probe_qk32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int MODE, bool VALU>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
    f4 s16[8], o[6];
    f16v s32[2];
    for (int i = 0; i < 8; ++i) s16[i] = f4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 6; ++i) o[i] = f4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) s32[i][j] = 0.f;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = -0.001f * (threadIdx.x + i);
    auto valu = [&](int n) __attribute__((always_inline)) {      // n "slots": 2 exp + 1 cvt-like + 1 max3-like per slot
        if (!VALU) return;
#pragma unroll
        for (int j = 0; j < n; ++j)
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_cvt_pk_f16_f32 %2, %0, %1\n v_max3_f32 %3, %3, %0, %1" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
    };
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int m = 0; m < 16; ++m) { s16[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, s16[m & 7], 0, 0, 0); valu(1); __builtin_amdgcn_sched_barrier(0); }
        } else {
#pragma unroll
            for (int m = 0; m < 6; ++m) {
                s32[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, s32[m & 1], 0, 0, 0);
                valu(m < 4 ? 3 : 2);                              // the same 16 slots spread over 6 MFMAs
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int m = 0; m < 12; ++m) { o[m % 6] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, o[m % 6], 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += s16[i][0] + x[i];
    for (int i = 0; i < 6; ++i) s += o[i][1];
    s += s32[0][3] + s32[1][7];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float* out; (void)hipMalloc(&out, 256 * 4 * 256 * 4);
    const int iters = 20000;
    auto run = [&](const char* name, auto kern, int bpc) {
        hipLaunchKernelGGL(kern, dim3(256 * bpc), dim3(256), 0, 0, out, 100);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(256 * bpc), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-66s %d wave(s)/SIMD: %8.3f ms  %7.1f ns per key tile per wave, %7.1f ns per tile per SIMD\n", name, bpc, ms, ms * 1e6 / iters, ms * 1e6 / iters / bpc);
    };
    for (int bpc : {1, 2, 3}) {
        run("A 16 x 16x16x32 (k 64) + 12 PV, MFMA only", k<0, false>, bpc);
        run("B  6 x 32x32x16 (k 48) + 12 PV, MFMA only", k<1, false>, bpc);
        run("A ... + 32 exp / 16 cvt / 16 max3 interleaved", k<0, true>, bpc);
        run("B ... + 32 exp / 16 cvt / 16 max3 interleaved", k<1, true>, bpc);
    }
    return 0;
}
