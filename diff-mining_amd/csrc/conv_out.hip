// conv_out.hip — conv_out (3x3, 320 -> 4 channels) + the eps-MSE of `SD.compute_loss` (diffmining/typicality/compute.py:100-101:
// `unet(...).sample`, `mse_loss(..., reduction='none')`) with the input rows staged ONCE in LDS and walked by all nine taps (r05).
//
// conv_out_kernel (misc.hip) gathers every pixel's nine taps through L1 / L2 — 23 KB of 16-byte reads per pixel from lines that
// more waves want than the L1 holds: 0.44 ms for one 419 MB tensor at the bench batch, 4x its HBM time; the same gather feeding the
// matrix cores read 0.47 ms (misc.hip's note): the bound is the gather, not the arithmetic.  Here a block owns a strip of TR image
// rows of one sample and walks the 320 channels in five 64-channel slabs:
//   * a slab of the strip's TR + 2 image rows, one halo pixel left and right ((TR + 2) x (W + 2) LDS rows of 128 bytes), goes
//     L2 / HBM -> LDS by LDS-DMA (global_load_lds_dwordx4: 1 KiB pieces of 8 rows, the XOR swizzle on the per-lane source chunk),
//     the next slab in flight while this one is computed — every input byte is requested (TR + 2) / TR times instead of nine;
//   * the four output channels are rows 0..3 of a 16x16x32 MFMA's A operand (lanes of rows 4..15 hold zeros; the weights of the
//     whole layer, 23 KB, sit in LDS), sixteen pixels its columns: tap (dy, dx) of a pixel is the same B-fragment read shifted by
//     dy (W + 2) + dx LDS rows, 18 MFMAs per slab and 16-pixel fragment;
//   * lanes 0..15 end with the four channels of their pixel: + bias, fp16 (the reference's rounding of `.sample`), squared error
//     against eps in fp32, stores of 64 contiguous bytes per channel.
// The fp32 sums run in another order than conv_out_kernel's (MFMA k blocks of 32 instead of eight dot2 lanes): equal to fp32
// rounding, not bit-identical; a pixel's bits do not depend on the batch (the strip geometry is a function of H and W only).
#include "dm_kernels.h"

namespace dm {

typedef _Float16 co_half8 __attribute__((ext_vector_type(8)));
typedef float co_floatx4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* co_gptr_t;
typedef __attribute__((address_space(3))) void* co_lptr_t;

namespace {

constexpr int CO_C = 320;                  // input channels
constexpr int CO_K = 9 * CO_C;
constexpr int CO_SLABS = CO_C / 64;
constexpr int CO_WBYTES = 23 * 1024;       // 4 x 2880 halfs = 23 040 bytes, rounded to whole KiB (piece bases stay 1 KiB aligned)
constexpr int CO_NW = 8;                   // waves per block (two per SIMD: one wave's fragment reads fly under the other's MFMAs)
constexpr int CO_MAXF = 3;                 // 16-pixel fragments per wave: a strip has at most 8 waves x 3 x 16 = 384 pixels
constexpr int CO_MAXP = 9;                 // LDS-DMA pieces per wave and slab: a stage has at most 72 KiB

__device__ __attribute__((aligned(256))) unsigned char g_co_zero[256];

// NF: fragments per wave = ceil(strip pixels / 128) — compile-time, so the fragment loops carry no branch and the compiler batches the
// B-fragment reads of a (tap, k step) in front of its MFMAs (with a run-time fragment count every MFMA waited for its own read: 735 us
// against 430 for the gather kernel); fragments beyond the strip's pixels recompute its last pixel and store nothing
template <typename TE, int NF>
__global__ __launch_bounds__(64 * CO_NW)
void conv_out_rows_kernel(const f16* __restrict__ Xn, const f16* __restrict__ w, const f16* __restrict__ bias, const TE* __restrict__ eps,
                          int B, int H, int W, int TR, int nstrips, float* __restrict__ loss, f16* __restrict__ pred, int eps_rows,
                          int out_group, int out_stride, int out_off) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kg = lane >> 4;
    const int b = blockIdx.x / nstrips, strip = blockIdx.x - b * nstrips;
    const int oh0 = strip * TR;
    const int rows = (H - oh0) < TR ? (H - oh0) : TR;
    const int PITCH = W + 2;
    const int SR = (TR + 2) * PITCH;
    const int NP = (SR + 7) >> 3;
    const int SB = NP * 1024;
    char* const ws = smem;
    char* const st0 = smem + CO_WBYTES;

    // ---- LDS-DMA of slab `sl` into stage `stg`: lane (lr, lc) of piece i lands at LDS row 8 i + lr, 16-byte slot lc, and fetches the
    //      chunk lc ^ (row & 7) of its pixel (zero page: halo / outside the image / rows beyond the stage)
    const int lr = lane >> 3, lc = lane & 7;
    auto issue_slab = [&](int sl, int stg) {
#pragma unroll
        for (int j = 0; j < CO_MAXP; ++j) {
            const int piece = wid + CO_NW * j;
            if (piece < NP) {
                const int R = piece * 8 + lr;
                const int ir = R / PITCH;
                const int iw = R - ir * PITCH - 1;
                const int ih = oh0 - 1 + ir;
                const bool ok = R < SR && iw >= 0 && iw < W && ih >= 0 && ih < H;
                const f16* src = ok ? Xn + (((size_t)b * H + ih) * W + iw) * CO_C + sl * 64 + ((lc ^ (R & 7)) << 3)
                                    : reinterpret_cast<const f16*>(g_co_zero) + (lc << 3);
                __builtin_amdgcn_global_load_lds((co_gptr_t)src, (co_lptr_t)(st0 + stg * SB + piece * 1024), 16, 0, 0);
            }
        }
    };
    issue_slab(0, 0);
    // the layer's weights [4][2880] -> LDS (once per block; they stay in L2)
    for (int i = tid * 8; i < 4 * CO_K; i += 64 * CO_NW * 8)
        *reinterpret_cast<co_half8*>(ws + i * 2) = *reinterpret_cast<const co_half8*>(w + i);

    // ---- this wave's fragments: pixel (r, c) of the strip -> LDS row of its tap (0, 0)
    const int npx = rows * W;
    int r0[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        int px = (wid + CO_NW * f) * 16 + l15;
        px = px < npx ? px : npx - 1;
        const int r = px / W;
        r0[f] = r * PITCH + (px - r * W);
    }
    co_floatx4 acc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) acc[f] = co_floatx4{0.f, 0.f, 0.f, 0.f};

    for (int sl = 0; sl < CO_SLABS; ++sl) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                              // slab sl has landed (every wave's pieces); everyone is done reading the other stage
        if (sl + 1 < CO_SLABS) issue_slab(sl + 1, (sl + 1) & 1);
        const char* const xs = st0 + (sl & 1) * SB;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                co_half8 a = co_half8{0, 0, 0, 0, 0, 0, 0, 0};
                if (l15 < 4) a = *reinterpret_cast<const co_half8*>(ws + ((l15 * CO_K + tap * CO_C + sl * 64 + ks * 32 + kg * 8) << 1));
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const int R = r0[f] + dy * PITCH + dx;
                    const co_half8 bv = *reinterpret_cast<const co_half8*>(xs + R * 128 + (((ks * 4 + kg) ^ (R & 7)) << 4));
                    acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bv, acc[f], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue: lanes 0..15 hold channels 0..3 of pixel (fragment, l15)
    if (kg == 0) {
        const int HW = H * W;
        const int orow = (b / out_group) * out_stride + out_off + b % out_group;
        float bs[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) bs[r] = (float)bias[r];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int px = (wid + CO_NW * f) * 16 + l15;
            if (px < npx) {
                const int rem = oh0 * W + px;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const f16 pr = (f16)(acc[f][r] + bs[r]);
                    const size_t oidx = ((size_t)orow * 4 + r) * HW + rem;
                    if (eps) {
                        const float d = (float)pr - (float)eps[((size_t)(b % eps_rows) * 4 + r) * HW + rem];
                        loss[oidx] = d * d;
                    }
                    if (pred) pred[oidx] = pr;
                }
            }
        }
    }
}

}  // namespace

// strip height for an H x W latent: the tallest strip whose pixels fit the block's 20 fragments and whose two stages fit LDS next
// to the weights; 0 = this geometry stays on conv_out_kernel.  A function of (H, W) only, never of the batch.
int conv_out_rows_strip(int H, int W, int C0) {
    if (C0 != CO_C || H < 1 || W < 1) return 0;
    const int pitch = W + 2;
    int tr = 0;
    for (int t = 1; t <= H && t <= 64; ++t) {
        const int sr = (t + 2) * pitch, np = (sr + 7) / 8;
        if (t * W > CO_NW * CO_MAXF * 16 || np > CO_NW * CO_MAXP || CO_WBYTES + 2 * np * 1024 > 156 * 1024) break;
        tr = t;
    }
    return tr;
}

hipError_t launch_conv_out_rows(const f16* Xn, const f16* w, const f16* bias, const void* eps, int eps_f32, int B, int H, int W, int C0,
                                float* loss, f16* pred, int eps_rows, int out_group, int out_stride, int out_off, hipStream_t s) {
    const int tr = conv_out_rows_strip(H, W, C0);
    if (tr <= 0) return hipErrorInvalidValue;
    const int nstrips = (H + tr - 1) / tr;
    const int np = ((tr + 2) * (W + 2) + 7) / 8;
    const size_t lds = (size_t)CO_WBYTES + 2 * (size_t)np * 1024;
    const int nf = (tr * W + CO_NW * 16 - 1) / (CO_NW * 16);
    static std::atomic<uint64_t> attr_seen{0};      // hipFuncSetAttribute is per DEVICE, not per process
    if (first_use_on_device(attr_seen)) {
        (void)hipFuncSetAttribute((const void*)conv_out_rows_kernel<float, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_out_rows_kernel<float, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_out_rows_kernel<float, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_out_rows_kernel<f16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_out_rows_kernel<f16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_out_rows_kernel<f16, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    const dim3 grid((unsigned)(B * nstrips)), block(64 * CO_NW);
#define DM_CO_LAUNCH(TE, NF_) hipLaunchKernelGGL((conv_out_rows_kernel<TE, NF_>), grid, block, lds, s, Xn, w, bias, (const TE*)eps, B, H, W, tr, nstrips, \
                                                 loss, pred, eps_rows, out_group, out_stride, out_off)
    if (eps_f32) { if (nf == 1) DM_CO_LAUNCH(float, 1); else if (nf == 2) DM_CO_LAUNCH(float, 2); else DM_CO_LAUNCH(float, 3); }
    else { if (nf == 1) DM_CO_LAUNCH(f16, 1); else if (nf == 2) DM_CO_LAUNCH(f16, 2); else DM_CO_LAUNCH(f16, 3); }
#undef DM_CO_LAUNCH
    return hipGetLastError();
}

}  // namespace dm
