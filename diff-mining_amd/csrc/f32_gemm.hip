// f32_gemm.hip — fp32 implicit GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), gfx950.
//
// The arithmetic of the reference's DIFT featuriser: `SDFeaturizer` builds its pipeline without torch_dtype and calls the U-Net
// without autocast (diffmining/typicality/dift.py:197-199, 191), i.e. every convolution / linear is an fp32 GEMM.  Same operator
// as igemm_tile.h (3x3 convolution stride 1 / 2 / after nearest up-sampling, 1x1 convolution, linear; two concatenated sources;
// bias + per-sample time embedding + residual in the epilogue) on fp32 tensors.
//
// Roofline: the fp32 MFMA rate is 256 FLOP/clk/CU (157.3 TFLOP/s; tools/probes/probe_peak_f32.hip reaches 154-156 with the 32x32x2
// instruction from one wave per SIMD) — 1/16 of the fp16 rate —, so at this block tile (128 pixels x 160 channels x 32 k per step:
// 80 MFMAs of 64 cycles per wave and step against 24 ds_read_b128, 9 ds_write_b128 and 9 16-byte global loads per lane) the
// kernel is bound by the matrix pipe; nothing of the fp16 tile's LDS-DMA machinery is needed.  What does matter (measured r04,
// DESIGN 4e): the two blocks of a CU run in lock step (their waves share a SIMD's matrix pipe one to one), so every non-MFMA
// instruction a wave issues OUTSIDE the shadow of its own MFMAs idles the pipe — the first version (per-step address arithmetic
// of ~400 VALU / SALU instructions in front of the MFMAs) ran at 0.73 of peak.  Hence:
//   * per-lane source POINTERS, advanced by 32 floats per step; the row -> pixel arithmetic runs only when the tap or the
//     concatenated source changes (every >= 5 steps); out-of-image rows and channels beyond Cout read a zero page (no branches)
//   * the step's LDS stores (stage t+1) and global loads (step t+2) are issued BETWEEN its MFMA groups
//   * fragments of k group g+1 are read under the MFMAs of group g
// Geometry: block = 4 waves, wave w = pixels 32 w .. + 31 x all 160 channels = 5 accumulator tiles of 32 x 32 (80 VGPRs);
// MFMA operands A = weights [channel][k], B = activations [k][pixel] -> D[channel][pixel]: a lane holds 4 consecutive channels of
// one pixel per register quad, so the epilogue stores 16 bytes per lane into the NHWC row.  LDS: two stages of (160 + 128) rows x
// 32 floats (72 KiB, two blocks per CU), 16-byte chunks XOR-swizzled by row.
#include "f32_kernels.h"
#include <atomic>

namespace dm32 {
namespace {

constexpr int BM = 128, BN = 160, BK = 32, NT = 256;
constexpr int CH = BK / 4;                               // 16-byte chunks per row and k step
constexpr int XI = BM * CH / NT, WI = BN * CH / NT;      // 4 activation + 5 weight chunks per lane and step
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __attribute__((aligned(256))) float g_zero_page32[BK] = {0};

template <bool UP4>
__global__ __launch_bounds__(NT, 2) void gemm32_kernel(GemmParams p, int tiles_m, int tiles_n, int per_xcd, const float* zero_page) {
    __shared__ __attribute__((aligned(16))) float Ws[2][BN][BK];
    __shared__ __attribute__((aligned(16))) float Xs[2][BM][BK];
    // consecutive block ids land on different XCDs (8 L2s): give each XCD a contiguous range of tiles (channel tiles fastest: the
    // blocks of one XCD share their activation rows in its L2)
    const int tile = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (tile >= tiles_m * tiles_n) return;
    int tm = tile / tiles_n;
    const int tn = tile - tm * tiles_n;
    // mode 5 (r04): Upsample2D (nearest, exact 2x) + conv3x3 as four 2x2 convolutions on the SOURCE grid, one per output parity class
    // ph = py * 2 + px (igemm_pers_tile.h, UP4): H x W = OH x OW = the source grid, M = rows per class, tiles_m = 4 x tiles per class,
    // Wp [4][Cout][4 Cin] with the 3x3 taps that read the same source pixel pre-summed; the epilogue scatters rows into [N][2H][2W].
    int ph = 0;
    if constexpr (UP4) { const int tmp = tiles_m >> 2; ph = tm / tmp; tm -= ph * tmp; }
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int c32 = lane & 31, h32 = lane >> 5;
    const int K = (UP4 ? 4 : p.mode == 0 ? 1 : 9) * p.Cin;
    const int OHW = p.OH * p.OW;
    const int lrow = tid / CH, lchunk = tid % CH;         // loader: rows lrow + 32 i, chunk lchunk
    const float* zero = zero_page + 4 * lchunk;

    int xn[XI], xoy[XI], xox[XI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int m = m0 + lrow + (NT / CH) * i;
        if (m < p.M) {
            if (p.mode == 0) { xn[i] = 0; xoy[i] = 0; xox[i] = m; }
            else { const int n = m / OHW, rem = m - n * OHW; xn[i] = n; xoy[i] = rem / p.OW; xox[i] = rem - xoy[i] * p.OW; }
        } else { xn[i] = -1; xoy[i] = 0; xox[i] = 0; }
    }
    const float sh = (float)p.H / (float)p.OH, sw = (float)p.W / (float)p.OW;

    const float* xsrc[XI]; int xinc[XI];
    const float* wsrc[WI]; int winc[WI];
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int ch = n0 + lrow + (NT / CH) * i;
        const bool ok = ch < p.Cout;
        wsrc[i] = ok ? p.Wp + (UP4 ? (size_t)ph * p.Cout + ch : (size_t)ch) * K + 4 * lchunk : zero;
        winc[i] = ok ? BK : 0;
    }
    int ld_tap = 0, ld_c = 0;                              // loader position: tap and channel offset of the NEXT step to load
    // pointers of the activation rows for the step at (ld_tap, ld_c): runs at a tap start and where the concatenated source changes
    auto set_rows = [&]() {
        const bool second = ld_c >= p.C1;
        const float* base = second ? p.X2 : p.X;
        const int cs = second ? p.Cin - p.C1 : p.C1;
        const int cc = (second ? ld_c - p.C1 : ld_c) + 4 * lchunk;
        const int dy = UP4 ? ld_tap >> 1 : ld_tap / 3, dx = UP4 ? ld_tap & 1 : ld_tap - dy * 3;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            long long off = -1;
            if (xn[i] >= 0) {
                if (UP4) {                           // 2x2 tap (dy, dx) of parity class ph reads source pixel (y - 1 + py + dy, x - 1 + px + dx)
                    const int ih = xoy[i] + dy - 1 + (ph >> 1), iw = xox[i] + dx - 1 + (ph & 1);
                    if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) off = ((long long)xn[i] * p.H + ih) * p.W + iw;
                }
                else if (p.mode == 0) off = xox[i];
                else if (p.mode == 3) {              // F.interpolate(mode="nearest") to (OH, OW), then the 3x3 convolution (pad 1)
                    const int uh = xoy[i] + dy - 1, uw = xox[i] + dx - 1;
                    if (uh >= 0 && uh < p.OH && uw >= 0 && uw < p.OW) {
                        int ih = (int)floorf((float)uh * sh); ih = ih < p.H - 1 ? ih : p.H - 1;
                        int iw = (int)floorf((float)uw * sw); iw = iw < p.W - 1 ? iw : p.W - 1;
                        off = ((long long)xn[i] * p.H + ih) * p.W + iw;
                    }
                } else {
                    const int st = (p.mode == 1) ? 1 : 2, pd = (p.mode == 4) ? 0 : 1;
                    const int ih = xoy[i] * st + dy - pd, iw = xox[i] * st + dx - pd;
                    if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) off = ((long long)xn[i] * p.H + ih) * p.W + iw;
                }
            }
            xsrc[i] = (off >= 0) ? base + off * cs + cc : zero;
            xinc[i] = (off >= 0) ? BK : 0;
        }
    };
    auto next_pos = [&]() {                                 // after a step's loads were issued: move the loader on
        ld_c += BK;
        if (ld_c == p.Cin) { ld_c = 0; ++ld_tap; set_rows(); }
        else if (ld_c == p.C1) set_rows();
    };
    // LDS rows hold 32 floats as 8 chunks of 16 bytes; chunk q of row r sits at position q ^ (r / 2 % 8): the 16 lanes of a
    // ds_read_b128 group then touch 16 distinct 16-byte bank columns
    auto pos = [&](int row, int q) { return q ^ ((row >> 1) & (CH - 1)); };

    v4f acc[5][4];          // [channel tile of 32][row group r / 4 of the 32 x 32 accumulator]
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK;
    v4f xr[XI], wr[WI];
    auto gload_x = [&](int i) { xr[i] = *reinterpret_cast<const v4f*>(xsrc[i]); xsrc[i] += xinc[i]; };
    auto gload_w = [&](int i) { wr[i] = *reinterpret_cast<const v4f*>(wsrc[i]); wsrc[i] += winc[i]; };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < XI; ++i) { const int r = lrow + (NT / CH) * i; *reinterpret_cast<v4f*>(&Xs[buf][r][4 * pos(r, lchunk)]) = xr[i]; }
#pragma unroll
        for (int i = 0; i < WI; ++i) { const int r = lrow + (NT / CH) * i; *reinterpret_cast<v4f*>(&Ws[buf][r][4 * pos(r, lchunk)]) = wr[i]; }
    };
    // A lane (c32, h32) = weights[channel c32][k = h32], B = activations[k = h32][pixel c32]; a 16-byte fragment read gives floats
    // 4 h32 .. + 3 of an 8-float k group, register jj pairs k = 4 h32 + jj in both operands
    auto frag = [&](int buf, int k8, v4f (&a)[5], v4f& b) {
#pragma unroll
        for (int ct = 0; ct < 5; ++ct) { const int r = ct * 32 + c32; a[ct] = *reinterpret_cast<const v4f*>(&Ws[buf][r][4 * pos(r, 2 * k8 + h32)]); }
        const int r = wid * 32 + c32;
        b = *reinterpret_cast<const v4f*>(&Xs[buf][r][4 * pos(r, 2 * k8 + h32)]);
    };

    set_rows();
#pragma unroll
    for (int i = 0; i < XI; ++i) gload_x(i);
#pragma unroll
    for (int i = 0; i < WI; ++i) gload_w(i);
    next_pos();
    lstore(0);
    if (nk > 1) {
#pragma unroll
        for (int i = 0; i < XI; ++i) gload_x(i);
#pragma unroll
        for (int i = 0; i < WI; ++i) gload_w(i);
        next_pos();
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        const bool st = kt + 1 < nk, ld = kt + 2 < nk;        // uniform
        v4f fa[5], fb;
        frag(buf, 0, fa, fb);
#pragma unroll
        for (int k8 = 0; k8 < BK / 8; ++k8) {
            v4f an[5], bn;
            if (k8 + 1 < BK / 8) frag(buf, k8 + 1, an, bn);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
                for (int ct = 0; ct < 5; ++ct) {
                    v16f& d = *reinterpret_cast<v16f*>(&acc[ct][0]);
                    d = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ct][jj], fb[jj], d, 0, 0, 0);
                }
                // the step's memory work, spread over the shadows of its 16 MFMA groups (stage t+1 was last read in step t-1, which
                // every wave left through the barrier; the registers it is stored from were loaded during step t-1)
                if (k8 == 0 && jj == 0 && st) lstore(buf ^ 1);
                if (ld) {
                    const int slot = k8 * 4 + jj - 2;             // slots 0..8 = the nine loads of step t+2, one per MFMA group
                    if (slot >= 0 && slot < XI) gload_x(slot);
                    if (slot >= XI && slot < XI + WI) gload_w(slot - XI);
                    if (slot == XI + WI) next_pos();
                }
            }
            if (k8 + 1 < BK / 8) {
#pragma unroll
                for (int ct = 0; ct < 5; ++ct) fa[ct] = an[ct];
                fb = bn;
            }
        }
        __syncthreads();
    }

    // epilogue — 32 x 32 accumulator: lane (c32, h32), registers 4 q .. 4 q + 3 = channels 8 q + 4 h32 .. + 3 of pixel c32
    const int m = m0 + wid * 32 + c32;
    if (m < p.M) {
        const float* tb = p.temb ? p.temb + (size_t)(m / OHW) * p.temb_ld : nullptr;
        size_t orow = (size_t)m;
        if constexpr (UP4) {
            const int n = m / OHW, rem = m - n * OHW, oy = rem / p.OW, ox = rem - oy * p.OW;
            orow = ((size_t)n * (2 * p.OH) + (2 * oy + (ph >> 1))) * (size_t)(2 * p.OW) + (2 * ox + (ph & 1));
        }
#pragma unroll
        for (int ct = 0; ct < 5; ++ct)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ch = n0 + ct * 32 + 8 * q + 4 * h32;
                if (ch >= p.Cout) continue;
                v4f v = acc[ct][q];
                if (p.bias) v += *reinterpret_cast<const v4f*>(p.bias + ch);
                if (p.epi == 1) {          // GEGLU: the quad is (h0, h1, g0, g1); erf GELU as F.gelu
                    v2f y;
                    y[0] = v[0] * (0.5f * v[2] * (1.0f + erff(v[2] * 0.70710678118654752440f)));
                    y[1] = v[1] * (0.5f * v[3] * (1.0f + erff(v[3] * 0.70710678118654752440f)));
                    *reinterpret_cast<v2f*>(p.Y + orow * p.ldy + (ch >> 1)) = y;
                    continue;
                }
                if (tb) v += *reinterpret_cast<const v4f*>(tb + ch);
                if (p.res) v += *reinterpret_cast<const v4f*>(p.res + (size_t)m * p.ldres + ch);
                *reinterpret_cast<v4f*>(p.Y + orow * p.ldy + ch) = v;
            }
    }
}

}  // namespace

hipError_t launch_gemm(const GemmParams& p, hipStream_t s) {
    const int taps = p.mode == 0 ? 1 : p.mode == 5 ? 4 : 9;
    if (p.mode == 5 && (p.X2 || p.C1 != p.Cin || p.temb || p.res || p.epi || p.OH != p.H || p.OW != p.W || p.M != (p.M / (p.H * p.W)) * p.H * p.W))
        return hipErrorInvalidValue;
    if (p.M <= 0 || p.Cout <= 0 || p.Cin % BK != 0 || p.C1 % BK != 0 || p.C1 <= 0 || p.Cout % 4 != 0 || (p.C1 < p.Cin && !p.X2) ||
        (long long)taps * p.Cin > (1LL << 30) || p.ldy % (p.epi == 1 ? 2 : 4) != 0 || (p.epi == 1 && (p.temb || p.res)) || (p.res && p.ldres % 4 != 0) || (p.temb && p.temb_ld % 4 != 0))
        return hipErrorInvalidValue;
    const int tiles_m = (p.mode == 5 ? 4 : 1) * ((p.M + BM - 1) / BM), tiles_n = (p.Cout + BN - 1) / BN;
    const long long tiles = (long long)tiles_m * tiles_n;
    const int per_xcd = (int)((tiles + 7) / 8);
    // device address of the zero page: one symbol lookup per device ordinal
    static std::atomic<float*> zero_pages[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    float* zp = zero_pages[dev & 63].load(std::memory_order_relaxed);
    if (!zp) {
        hipError_t e = hipGetSymbolAddress((void**)&zp, HIP_SYMBOL(g_zero_page32));
        if (e != hipSuccess) return e;
        zero_pages[dev & 63].store(zp, std::memory_order_relaxed);
    }
    if (p.mode == 5) hipLaunchKernelGGL(gemm32_kernel<true>, dim3((unsigned)(per_xcd * 8)), dim3(NT), 0, s, p, tiles_m, tiles_n, per_xcd, (const float*)zp);
    else hipLaunchKernelGGL(gemm32_kernel<false>, dim3((unsigned)(per_xcd * 8)), dim3(NT), 0, s, p, tiles_m, tiles_n, per_xcd, (const float*)zp);
    return hipGetLastError();
}

}  // namespace dm32
