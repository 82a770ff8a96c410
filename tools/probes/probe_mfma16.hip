// issue cost of v_mfma_f32_16x16x16_f16 vs v_mfma_f32_16x16x32_f16 on gfx950 (is the k = 16 form half the price?)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/probe_mfma16.hip -o probe_mfma16 && ./probe_mfma16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    h8 a8, b8; h4 a4, b4;
    for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(0.001f * (threadIdx.x + i)); b8[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
    for (int i = 0; i < 4; ++i) { a4[i] = a8[i]; b4[i] = b8[i]; }
    f4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = f4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c[i], 0, 0, 0);
                else if (MODE == 1) c[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c[i], 0, 0, 0);
                else { if (i & 1) c[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c[i], 0, 0, 0); else c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c[i], 0, 0, 0); }
            }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    const int iters = 20000;
    auto run = [&](const char* name, auto kern, int bpc) {
        hipLaunchKernelGGL(kern, dim3(256 * bpc), dim3(256), 0, 0, out, 100);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(256 * bpc), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double n = 32.0 * iters * bpc;        // MFMAs per SIMD
        printf("%-40s %d wave(s)/SIMD: %8.3f ms  %6.2f ns per MFMA per SIMD\n", name, bpc, ms, ms * 1e6 / n);
    };
    for (int bpc : {1, 2}) {
        run("v_mfma_f32_16x16x32_f16", k<0>, bpc);
        run("v_mfma_f32_16x16x16_f16", k<1>, bpc);
        run("alternating x32 / x16", k<2>, bpc);
    }
    return 0;
}
