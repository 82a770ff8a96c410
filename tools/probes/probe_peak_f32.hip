// On-box ceiling of the fp32 matrix cores for gemm32's roofline (bench.py --workload dift): v_mfma_f32_16x16x4_f32 and
// v_mfma_f32_32x32x2_f32 on register operands, no memory or LDS traffic, 1 / 2 / 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/probe_peak_f32.hip -o probe_peak_f32 && ./probe_peak_f32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(256) void mfma_kernel(float* out, int iters) {
    float a = 0.001f * threadIdx.x, b = 0.002f * (threadIdx.x + 3);
    float s = 0.f;
    if (SHAPE == 16) {
        f4 c[8];
        for (int i = 0; i < 8; ++i) c[i] = f4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c[i]) : "v"(a), "v"(b));
        for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    } else {
        f16v c[4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c[i]) : "v"(a), "v"(b));
        for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][5];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float* out; hipMalloc(&out, 256 * 16 * 256 * 4);
    for (int shape : {16, 32})
        for (int bpc = 1; bpc <= 4; bpc *= 2)
            for (int iters : {500, 5000}) {
                auto k = shape == 16 ? mfma_kernel<16> : mfma_kernel<32>;
                hipLaunchKernelGGL(k, dim3(256 * bpc), dim3(256), 0, 0, out, 50);
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(k, dim3(256 * bpc), dim3(256), 0, 0, out, iters);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double per_iter = shape == 16 ? 32 * 2.0 * 16 * 16 * 4 : 16 * 2.0 * 32 * 32 * 2;
                const double flops = per_iter * iters * 4.0 * 256.0 * bpc;
                printf("MFMA f32 %s: %d wave(s)/SIMD, %5d iters: %8.3f ms  %6.1f TFLOP/s\n", shape == 16 ? "16x16x4" : "32x32x2", bpc, iters, ms, flops / ms / 1e9);
            }
    return 0;
}
