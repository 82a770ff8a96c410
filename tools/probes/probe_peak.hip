// On-box peaks for the roofline denominators: dense fp16 MFMA rate with no memory traffic,
// and HBM stream-copy / read-only / write-only bandwidth.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/probe_peak.hip -o probe_peak && ./probe_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void mfma_kernel(float* out, int iters) {
    h8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(0.001f * (threadIdx.x + k)); b[k] = (_Float16)(0.002f * (threadIdx.x - k)); }
    f4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = f4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[i]) : "v"(a), "v"(b));
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void copy_kernel(const u4* __restrict__ src, u4* __restrict__ dst, size_t n, int mode) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    u4 acc = u4{0, 0, 0, 0};
    for (; i < n; i += stride) {
        if (mode == 0) dst[i] = src[i];
        else if (mode == 1) { const u4 v = src[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
        else dst[i] = u4{(unsigned)i, 1u, 2u, 3u};
    }
    if (mode == 1 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) dst[0] = acc;
}

int main() {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float* out; hipMalloc(&out, 256 * 16 * 256 * 4);
    for (int blocks_per_cu = 1; blocks_per_cu <= 4; blocks_per_cu *= 2) {
        for (int iters : {2000, 20000}) {
            hipLaunchKernelGGL(mfma_kernel, dim3(256 * blocks_per_cu), dim3(256), 0, 0, out, 200);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(mfma_kernel, dim3(256 * blocks_per_cu), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = 2.0 * 16 * 16 * 32 * 32.0 * iters * 4.0 * 256.0 * blocks_per_cu;   // 32 MFMAs/iter, 4 waves/block
            printf("MFMA 16x16x32 f16: %d wave(s)/SIMD, %6d iters: %8.3f ms  %7.1f TFLOP/s\n", blocks_per_cu, iters, ms, flops / ms / 1e9);
        }
    }
    const size_t bytes = (size_t)4 << 30;
    u4 *src, *dst; hipMalloc(&src, bytes); hipMalloc(&dst, bytes);
    hipMemset(src, 1, bytes); hipMemset(dst, 2, bytes);
    const char* names[3] = {"copy (read + write)", "read only", "write only"};
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(copy_kernel, dim3(256 * 16), dim3(256), 0, 0, src, dst, bytes / 16, mode);
        hipEventRecord(e0, 0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(copy_kernel, dim3(256 * 16), dim3(256), 0, 0, src, dst, bytes / 16, mode);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double moved = (mode == 0 ? 2.0 : 1.0) * bytes * 5;
        printf("HBM %-20s 4 GiB x5: %8.3f ms  %6.2f TB/s\n", names[mode], ms, moved / ms / 1e9);
    }
    return 0;
}
