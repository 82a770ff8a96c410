// Throughput probe: cycles per wave-instruction for v_exp_f32, v_fma_f32, v_pk_fma_f32, v_max3_f32,
// v_cvt_pk_f16_f32 on gfx950, one and two waves per SIMD.   hipcc --offload-arch=gfx950 -O2 probe_valu.hip -o probe_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ void k(float* out, long long* cyc, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 1) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 2) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %2, %2, %3, %0\n v_pk_fma_f32 %3, %3, %0, %1"
                                          : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6));) }
        if (OP == 3) { REP16(asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %0\n v_max3_f32 %3, %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 4) { REP16(asm volatile("v_cvt_pk_f16_f32 %0, %1, %2\n v_cvt_pk_f16_f32 %1, %2, %3\n v_cvt_pk_f16_f32 %2, %3, %0\n v_cvt_pk_f16_f32 %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 6) {
            typedef _Float16 h8 __attribute__((ext_vector_type(8)));
            typedef float f4 __attribute__((ext_vector_type(4)));
            static_assert(sizeof(h8) == 16, "");
            h8 x = {1, 2, 3, 4, 5, 6, 7, 8};
            f4 c0 = {a0, a1, a2, a3}, c1 = c0, c2 = c0, c3 = c0;
            REP16(c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, x, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, x, c1, 0, 0, 0);
                  c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, x, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, x, c3, 0, 0, 0);)
            a0 += c0[0] + c1[1] + c2[2] + c3[3];
        }
        if (OP == 5) { REP16(asm volatile("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int OP> void run(const char* name, int waves_per_simd) {
    float* out; long long* cyc; hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 8);
    const int iters = 2000;
    // one block per CU, 256 threads = 1 wave per SIMD; 512 threads = 2 waves per SIMD
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256 * waves_per_simd), 0, 0, out, cyc, iters);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256 * waves_per_simd), 0, 0, out, cyc, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-18s waves/SIMD=%d: %.2f ticks per wave-instr, %.2f per SIMD-instr; kernel %.3f ms -> %.3f ticks/ns, %.2f ns per SIMD-instr\n", name,
           waves_per_simd, (double)c / (iters * 64.0), (double)c / (iters * 64.0) / waves_per_simd, ms, c / (ms * 1e6),
           ms * 1e6 / (iters * 64.0) / waves_per_simd);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int w = 1; w <= 4; ++w) {
        run<0>("v_exp_f32", w); run<5>("v_exp_f16", w); run<1>("v_fma_f32", w); run<2>("v_pk_fma_f32", w); run<3>("v_max3_f32", w);
        run<4>("v_cvt_pk_f16_f32", w); run<6>("mfma_16x16x32_f16", w);
    }
    return 0;
}
