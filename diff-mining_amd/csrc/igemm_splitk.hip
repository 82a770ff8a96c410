// igemm_splitk.hip — split-K instantiation of the 128x320 implicit-GEMM tile kernel plus its reduction /
// epilogue kernel, for the small-M layers of the SDv1.5 U-Net (8x8 latents: M = 10 240 rows at the bench
// batch -> 320 tiles for 256 CUs, 62 % of a second round idle).  With the k range cut in four, each of
// the 1280 blocks does a quarter of the k steps and the chip is evenly loaded; the fp32 partial tiles
// (M x Cout x 4 B per part) are summed by a bandwidth-trivial kernel that applies the fused epilogue's
// arithmetic in the same order (bias, fp16; + time embedding, fp16; + residual, fp16).
// Own translation unit: co-compiling instantiations perturbs the register allocation of the main kernel.
#include "igemm_tile.h"

namespace dm {

namespace {

__global__ void splitk_reduce_kernel(const float* __restrict__ partial, int parts, long long MN, int M, int Cout, int OHW,
                                     const f16* __restrict__ bias, const f16* __restrict__ temb, int temb_ld,
                                     const f16* __restrict__ res, int ldres, f16* __restrict__ Y, int ldy) {
    const long long i8 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i8 >= MN) return;
    const int m = (int)(i8 / Cout), c = (int)(i8 - (long long)m * Cout);
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < parts; ++s) {
        const floatx4 lo = *reinterpret_cast<const floatx4*>(partial + (size_t)s * MN + i8);
        const floatx4 hi = *reinterpret_cast<const floatx4*>(partial + (size_t)s * MN + i8 + 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) { a[r] += lo[r]; a[4 + r] += hi[r]; }
    }
    half8 o;
    if (bias) {
        const half8 bv = *reinterpret_cast<const half8*>(bias + c);
#pragma unroll
        for (int r = 0; r < 8; ++r) a[r] += (float)bv[r];
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) o[r] = (f16)a[r];
    if (temb) {
        const half8 tv = *reinterpret_cast<const half8*>(temb + (size_t)(m / OHW) * temb_ld + c);
#pragma unroll
        for (int r = 0; r < 8; ++r) o[r] = (f16)((float)o[r] + (float)tv[r]);
    }
    if (res) {
        const half8 rv = *reinterpret_cast<const half8*>(res + (size_t)m * ldres + c);
#pragma unroll
        for (int r = 0; r < 8; ++r) o[r] = (f16)((float)o[r] + (float)rv[r]);
    }
    *reinterpret_cast<half8*>(Y + (size_t)m * ldy + c) = o;
}

}  // namespace

hipError_t launch_splitk_reduce(const IGemmParams& p, hipStream_t s);

hipError_t launch_igemm_splitk(const IGemmParams& p, hipStream_t s) {
    constexpr int WC = 4, NI = 5, TP = 128, TC = 16 * NI * WC;
    const int ntaps = (p.mode == IG_DENSE) ? 1 : 9;
    const int nk = ntaps * (p.Cin / BK);
    if (p.ksplit < 2 || !p.partial || p.epi != EPI_PLAIN || p.Cout % TC != 0 || nk % p.ksplit != 0) return hipErrorInvalidValue;
    constexpr size_t lds = 2 * (size_t)(TP + TC) * 128 + 1024;
    const int tiles = ((p.M + TP - 1) / TP) * (p.Cout / TC);
    static std::atomic<uint64_t> attr_seen{0};      // hipFuncSetAttribute is per DEVICE, not per process
    if (first_use_on_device(attr_seen)) {
        (void)hipFuncSetAttribute((const void*)igemm_kernel<WC, EPI_PLAIN, NI, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    launch_timed((igemm_kernel<WC, EPI_PLAIN, NI, true>), dim3(tiles * p.ksplit), dim3(128 * WC), lds, s, p);
    return launch_splitk_reduce(p, s);
}

// partial[ksplit][M][Cout] fp32 -> Y with the fused epilogue's arithmetic (also behind the persistent split-K kernel)
hipError_t launch_splitk_reduce(const IGemmParams& p, hipStream_t s) {
    const long long MN = (long long)p.M * p.Cout;
    const int OHW = p.OH * p.OW;
    launch_timed(splitk_reduce_kernel, dim3((unsigned)((MN / 8 + 255) / 256)), dim3(256), 0, s, p.partial, p.ksplit, MN, p.M,
                       p.Cout, OHW > 0 ? OHW : 1, p.bias, p.temb, p.temb_ld, p.res, p.ldres, p.Y, p.ldy);
    return hipGetLastError();
}

}  // namespace dm
