// Slab probe (r04): the short-K layers (GEGLU 320 -> 2560 and friends) spend 6.2 of 15.4 us per 256 x 320 tile in an epilogue that
// nothing overlaps.  r03's probe_pingpong.hip showed that splitting the BLOCK (two wave sets, two blocks per CU,
// specialised waves) costs more in the k step than the overlap returns.  This probe prices the remaining structure: overlap INSIDE one
// wave by software pipelining, which needs two accumulator sets and therefore small accumulators —
//   SLAB : 8 waves, a block owns 128 activation rows x K = 320 RESIDENT in LDS (80 KiB, fetched once per row block) and walks the NS = 16
//          channel slabs of 160 GEMM channels; a slab = 5 k steps on 20 KiB weight stages (ring of three, counted vmcnt, one barrier per
//          step); wave tile 32 px x 80 ch = 20 MFMA + 14 fragment reads per step, accumulators 40 registers, TWO sets:
//          mode 0: no epilogue work; 1: the slab's 20 GELUs per lane after its k loop (serial); 2: the GELUs of slab j - 1 issued
//          between the MFMA groups of slab j (two accumulator quads per k step).
// Synthetic activity mix as probe_pingpong.hip (real LDS-DMA, fragment reads, MFMAs, the real erf-GELU sequence; no real data flow).
// Prints ns per 256 x 320 x 320 tile equivalent (= 4 slab units) per CU, next to probe_pingpong's BASE (15409 ns with / 9187 without
// the epilogue on the shipped structure).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/probe_slab.hip -o probe_slab && ./probe_slab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float gelu_erf(float x) {
    const float ax = __builtin_fabsf(x);
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f * 0.70710678118654752440f, ax, 1.0f));
    float poly = __builtin_fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f);
    poly = __builtin_fmaf(t, poly, 0.5f * 1.421413741f);
    poly = __builtin_fmaf(t, poly, 0.5f * -0.284496736f);
    poly = __builtin_fmaf(t, poly, 0.5f * 0.254829592f);
    poly *= t;
    const float e = __builtin_amdgcn_exp2f((x * x) * -0.72134752044448170368f);
    return __builtin_fmaxf(x, 0.f) - ax * (poly * e);
}
// one accumulator quad (h0, h1, g0, g1) -> two GEGLU outputs, as the real epilogue: bias, round, gelu, multiply, round
__device__ __forceinline__ float geglu_quad(const f4 a, float b) {
    const _Float16 h0 = (_Float16)(a[0] + b), h1 = (_Float16)(a[1] + b), g0 = (_Float16)(a[2] + b), g1 = (_Float16)(a[3] + b);
    const _Float16 q0 = (_Float16)gelu_erf((float)g0), q1 = (_Float16)gelu_erf((float)g1);
    return (float)(_Float16)((float)h0 * (float)q0) + (float)(_Float16)((float)h1 * (float)q1);
}

constexpr int ROWS = 128, KTOT = 320, NKS = KTOT / 64, SLABC = 160, NST = 3;
constexpr int A_BYTES = NKS * ROWS * 128, W_STAGE = SLABC * 128, LDS_BYTES = A_BYTES + NST * W_STAGE;      // 80 KiB + 3 x 20 KiB

template <int MODE>
__global__ __launch_bounds__(512, 1) void slab_kernel(const _Float16* __restrict__ Wp, const _Float16* __restrict__ X, int nrb, int NS, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const wring = smem + A_BYTES;
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lrow = lane >> 3, lchunk = ((lane & 7) ^ lrow) * 8;
    const int l15 = lane & 15, lg = lane >> 4;
    const int wp = wid >> 1, wc = wid & 1;
    const int boff = (wp * 32 + l15) * 128 + ((lg ^ (l15 & 7)) << 4);          // activation fragment (B operand) inside a k slab of A
    const int aoff = (wc * 80 + l15) * 128 + ((lg ^ (l15 & 7)) << 4);          // weight fragment (A operand) inside a stage
    const int npw = wid < 4 ? 3 : 2;                                            // this wave's LDS-DMA pieces per weight stage (20 in all)
    f4 accA[10], accB[10];
    for (int i = 0; i < 10; ++i) { accA[i] = f4{0.f, 0.f, 0.f, 0.f}; accB[i] = f4{0.f, 0.f, 0.f, 0.f}; }
    float out = 0.f;
    int g = 0;                                                                  // global k step of the weight stream: stage g % 3
    auto issue_w = [&](int slab, int kt, int stage) __attribute__((always_inline)) {
        for (int r = 0; r < 3; ++r) {
            const int i = wid + 8 * r;
            if (i < 20) {
                const _Float16* src = Wp + (size_t)((slab % 16) * SLABC + i * 8 + lrow) * KTOT + kt * 64 + lchunk;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(wring + stage * W_STAGE + i * 1024), 16, 0, 0);
            }
        }
    };
    // one slab: five k steps on `cur`, the epilogue work of the previous slab (`prev`) interleaved when MODE == 2
    auto slab = [&](f4 (&cur)[10], f4 (&prev)[10], int s, bool has_prev) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 10; ++i) cur[i] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NKS; ++kt, ++g) {
            const char* wst = wring + (g % NST) * W_STAGE;
            const char* ast = smem + kt * ROWS * 128;
            // weights of stream step g + 2 (two ahead): next slab's first steps once this slab runs out
            {
                const int nkt = kt + 2, ns = s + (nkt >= NKS ? 1 : 0);
                issue_w(ns, nkt % NKS, (g + 2) % NST);
            }
            h8 fb[2], fa;
#pragma unroll
            for (int hk = 0; hk < 2; ++hk) {
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const h8*>(ast + boff + j * 2048 + hk * 64);
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    fa = *reinterpret_cast<const h8*>(wst + aoff + i * 2048 + hk * 64);
#pragma unroll
                    for (int j = 0; j < 2; ++j) cur[i * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb[j], cur[i * 2 + j], 0, 0, 0);
                    if (MODE == 2 && has_prev && hk == 0 && (i == 1 || i == 3)) out += geglu_quad(prev[kt * 2 + (i >> 1)], (float)kt);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // stream step g + 1 must have landed (this wave's pieces of it were issued one step ago); step g + 2's stay in flight
            if (npw == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
        }
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 10; ++i) out += geglu_quad(cur[i], (float)i);
        }
    };
    for (int rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
        // the row block's activations: 80 pieces of 1 KiB, ten per wave, once per NS slabs
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        for (int r = 0; r < 10; ++r) {
            const int i = wid + 8 * r;                                          // piece: k slab i / 16, rows (i % 16) * 8 ..
            const _Float16* src = X + (size_t)(rb * ROWS + (i % 16) * 8 + lrow) * KTOT + (i / 16) * 64 + lchunk;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + i * 1024), 16, 0, 0);
        }
        issue_w(0, 0, g % NST);
        issue_w(0, 1, (g + 1) % NST);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        for (int s = 0; s < NS; s += 2) {
            slab(accA, accB, s, s > 0);
            slab(accB, accA, s + 1, true);
        }
        if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 10; ++i) out += geglu_quad(accB[i], (float)i);      // the last slab's epilogue has nothing to hide under
        }
    }
    for (int i = 0; i < 10; ++i) out += accA[i][0] + accB[i][3];
    if (out == 1.2345f) sink[0] = out;
}

int main() {
    const int Cout = 2560, M = 655360;
    _Float16 *W, *X; float* sink;
    hipMalloc(&W, (size_t)Cout * KTOT * 2); hipMalloc(&X, (size_t)M * KTOT * 2); hipMalloc(&sink, 64);
    hipMemset(W, 0, (size_t)Cout * KTOT * 2); hipMemset(X, 0, (size_t)M * KTOT * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int NS = Cout / SLABC;                       // 16 slabs
    const int nrb = 256 * 16;                          // 16 row blocks of 128 rows per CU = 64 tile equivalents of 256 x 320 per CU
    auto time = [&](auto launch) {
        launch();
        hipEventRecord(e0, 0);
        for (int r = 0; r < 3; ++r) launch();
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        return ms / 3;
    };
    auto report = [&](const char* name, float ms) {
        const double units = (double)nrb * NS / 256.0;          // slab units (128 x 160 x 320) per CU
        const double tile_ns = ms * 1e6 / (units / 4.0);
        const double tf = 2.0 * 128 * 160 * 320.0 * nrb * NS / (ms * 1e-3) / 1e12;
        printf("%-72s %7.3f ms  %7.0f ns per 256x320x320 tile equivalent  %7.1f TF/s\n", name, ms, tile_ns, tf);
    };
#define SETLDS(k) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES)
    SETLDS(slab_kernel<0>); SETLDS(slab_kernel<1>); SETLDS(slab_kernel<2>);
    report("SLAB  no epilogue work", time([&] { hipLaunchKernelGGL(slab_kernel<0>, dim3(256), dim3(512), LDS_BYTES, 0, W, X, nrb, NS, sink); }));
    report("SLAB  GEGLU epilogue after each slab (serial)", time([&] { hipLaunchKernelGGL(slab_kernel<1>, dim3(256), dim3(512), LDS_BYTES, 0, W, X, nrb, NS, sink); }));
    report("SLAB  GEGLU epilogue of slab j-1 between the MFMAs of slab j", time([&] { hipLaunchKernelGGL(slab_kernel<2>, dim3(256), dim3(512), LDS_BYTES, 0, W, X, nrb, NS, sink); }));
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("error: %s\n", hipGetErrorString(e)); return 1; }
    return 0;
}
