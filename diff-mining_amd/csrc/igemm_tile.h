// igemm_tile.h — the 128-pixel implicit-GEMM tile kernel, shared by igemm.hip (U-Net: waves of
// 64 px x 80 ch, NI = 5) and igemm64.hip (VAE encoder: waves of 64 px x 64 ch, NI = 4, channel counts
// 128/256/512).  Each instantiation lives in its own translation unit on purpose: co-compiling large
// kernels perturbs the register allocation of both (measured).
#pragma once
#include "dm_kernels.h"
#include <cstdlib>
#include <type_traits>

namespace dm {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 ln_half2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int BK = 64;           // k per step (one tap, 64 input channels)

// erf-GELU  x * Phi(x) = max(x, 0) - |x| * T(|x|),  T(a) = 0.5 * (1 - erf(a / sqrt 2)) by Abramowitz-Stegun 7.1.26
// (|erf error| < 1.5e-7; measured over all 63 488 finite fp16 inputs: max |error| 3.3e-7, i.e. far below the fp16
// rounding of the result): 1 rcp + 1 exp2 + 12 plain VALU instead of libm erff (~40 instructions), which dominated the
// GEGLU epilogue (40 calls per lane per 256 x 320 tile).  The tail T is formed directly (no 1 - 1 cancellation for
// negative x), the 0.5 lives in the coefficients and 1 / sqrt 2 in the constants, and the max / |x| form needs no
// compare-and-select (r02: -2..3 % on the GEGLU launches against the x * (x < 0 ? T : 1 - T) form; a fused-multiply-add
// form and an inline-asm max measured the same).
__device__ __forceinline__ float gelu_erf(float x) {
    const float ax = __builtin_fabsf(x);
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f * 0.70710678118654752440f, ax, 1.0f));
    float poly = __builtin_fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f);
    poly = __builtin_fmaf(t, poly, 0.5f * 1.421413741f);
    poly = __builtin_fmaf(t, poly, 0.5f * -0.284496736f);
    poly = __builtin_fmaf(t, poly, 0.5f * 0.254829592f);
    poly *= t;
    const float e = __builtin_amdgcn_exp2f((x * x) * -0.72134752044448170368f);        // exp(-x^2 / 2)
    return __builtin_fmaxf(x, 0.f) - ax * (poly * e);
}

// Epilogue staged through LDS: fragments (bias / time-embedding / GEGLU applied, rounded to fp16) are
// written to an LDS tile [TP px][TCO ch] (row stride padded by 8 B: conflict-free ds_write_b64 /
// b32), then copied out as whole rows with 16-byte stores (+ the residual read the same way).
// The direct fragment stores write 8-byte (GEGLU: 4-byte) pieces of 16 different 128-byte lines per
// instruction; on the wide, short-K linears that partial-line traffic bound the whole kernel.
// Memory operations are batched and the optional operands are resolved once (uniform branches outside
// the fragment loops): under load a global access costs several thousand cycles, and vmcnt is
// in-order, so a per-fragment "load, wait, use" chain (or a load queued behind the previous stores)
// made this epilogue as long as ten k steps.  The bias of the tile's channels is staged in LDS before
// the main loop (`bias_lds`, zeros when there is no bias).
template <int EPI, int NTH, int TP, int TC, int NI, bool LN = false>
__device__ __forceinline__ void epilogue_lds(const IGemmParams& p, floatx4 (&acc)[NI][4], char* smem, const char* bias_lds,
                                             int p0, int c0out, int wc, int wp, int l15, int lg, int OHW) {
    constexpr int TCO = (EPI == EPI_GEGLU) ? TC / 2 : TC;     // output channels of the tile
    constexpr int ROWB = TCO * 2 + 8;
    // plain: bz = bias.  folded LayerNorm: bz = ln_t, sz = ln_s (fp32, staged behind the bias slot) and the
    // per-row (mean, rstd) of this lane's four pixel rows.
    float bz[NI][4], sz[LN ? NI : 1][4], mu[4], rs[4];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int cl = wc * (16 * NI) + 16 * i + 4 * lg;
        if (LN) {
            const floatx4 tv = *reinterpret_cast<const floatx4*>(bias_lds + 1024 + TC * 4 + cl * 4);
            const floatx4 sv = *reinterpret_cast<const floatx4*>(bias_lds + 1024 + cl * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) { bz[i][r] = tv[r]; sz[i][r] = sv[r]; }
        } else {
            const half4 bv = *reinterpret_cast<const half4*>(bias_lds + cl * 2);
#pragma unroll
            for (int r = 0; r < 4; ++r) bz[i][r] = (float)bv[r];
        }
    }
    if (LN) {          // (mean, rstd) of the tile's rows were staged in LDS before the k loop
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float2 st = *reinterpret_cast<const float2*>(bias_lds + 1024 + 8 * TC + (wp * 64 + 16 * j + l15) * 8);
            mu[j] = st.x; rs[j] = st.y;
        }
    }
    __syncthreads();                                           // every wave is done with the operand tiles
    auto stage = [&](auto has_temb) __attribute__((always_inline)) {
        constexpr bool TEMB = decltype(has_temb)::value;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pr = wp * 64 + 16 * j + l15;
            const int m = p0 + pr;
            half4 tv[NI];
            if (TEMB) {
                const int n = (m < p.M) ? (m / OHW) : 0;
                const f16* tp = p.temb + (size_t)n * p.temb_ld + c0out + wc * (16 * NI) + 4 * lg;
#pragma unroll
                for (int i = 0; i < NI; ++i) tv[i] = *reinterpret_cast<const half4*>(tp + 16 * i);     // batched
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int cl = wc * (16 * NI) + 16 * i + 4 * lg;              // tile-local channel
                float v0, v1, v2, v3;
                if (LN) {          // fma(rstd, acc, fma(-rstd mean, s, t)): the same form as igemm_pers_tile.h
                    const float nrm = -(rs[j] * mu[j]);
                    v0 = __builtin_fmaf(rs[j], acc[i][j][0], __builtin_fmaf(nrm, sz[i][0], bz[i][0]));
                    v1 = __builtin_fmaf(rs[j], acc[i][j][1], __builtin_fmaf(nrm, sz[i][1], bz[i][1]));
                    v2 = __builtin_fmaf(rs[j], acc[i][j][2], __builtin_fmaf(nrm, sz[i][2], bz[i][2]));
                    v3 = __builtin_fmaf(rs[j], acc[i][j][3], __builtin_fmaf(nrm, sz[i][3], bz[i][3]));
                } else {
                    v0 = acc[i][j][0] + bz[i][0]; v1 = acc[i][j][1] + bz[i][1];
                    v2 = acc[i][j][2] + bz[i][2]; v3 = acc[i][j][3] + bz[i][3];
                }
                if (EPI == EPI_GEGLU) {
                    const f16 h0 = (f16)v0, h1 = (f16)v1, g0 = (f16)v2, g1 = (f16)v3;
                    const f16 q0 = (f16)gelu_erf((float)g0), q1 = (f16)gelu_erf((float)g1);
                    typedef _Float16 half2_ __attribute__((ext_vector_type(2)));
                    const half2_ o = half2_{(f16)((float)h0 * (float)q0), (f16)((float)h1 * (float)q1)};
                    const int ol = (cl >> 4) * 8 + 2 * lg;
                    *reinterpret_cast<half2_*>(smem + pr * ROWB + ol * 2) = o;
                } else {
                    half4 o = half4{(f16)v0, (f16)v1, (f16)v2, (f16)v3};
                    if (TEMB) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = (f16)((float)o[r] + (float)tv[i][r]);
                    }
                    *reinterpret_cast<half4*>(smem + pr * ROWB + cl * 2) = o;
                }
            }
        }
    };
    if (EPI != EPI_GEGLU && p.temb) stage(std::true_type{}); else stage(std::false_type{});
    __syncthreads();
    constexpr int CPR = TCO / 8;                               // 16-byte chunks per row
    constexpr int NIT = (TP * CPR) / NTH;                      // chunks per thread (exact)
    static_assert((TP * CPR) % NTH == 0, "whole number of chunks per thread");
    constexpr int UB = (NIT % 5 == 0) ? 5 : ((NIT % 4 == 0) ? 4 : 1);
    const int c0o = (EPI == EPI_GEGLU) ? c0out / 2 : c0out;
    auto copy_out = [&](auto has_res) __attribute__((always_inline)) {
        constexpr bool RES = decltype(has_res)::value;
#pragma unroll
        for (int it0 = 0; it0 < NIT; it0 += UB) {
            half8 o[UB], rv[UB];
            int mrow[UB], chn[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int idx = threadIdx.x + (it0 + u) * NTH;
                const int row = idx / CPR, ch = idx - row * CPR;
                mrow[u] = p0 + row; chn[u] = ch;
                const char* src = smem + row * ROWB + ch * 16;
                const half4 lo = *reinterpret_cast<const half4*>(src);
                const half4 hi = *reinterpret_cast<const half4*>(src + 8);
                o[u] = half8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                if (RES) {
                    const int mr = mrow[u] < p.M ? mrow[u] : p.M - 1;
                    rv[u] = *reinterpret_cast<const half8*>(p.res + (size_t)mr * p.ldres + c0o + ch * 8);
                }
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                if (RES) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) o[u][r] = (f16)((float)o[u][r] + (float)rv[u][r]);
                }
                if (mrow[u] < p.M) *reinterpret_cast<half8*>(p.Y + (size_t)mrow[u] * p.ldy + c0o + chn[u] * 8) = o[u];
            }
        }
    };
    if (EPI != EPI_GEGLU && p.res) copy_out(std::true_type{}); else copy_out(std::false_type{});
}

__device__ __attribute__((aligned(256))) unsigned char g_zero_page[256];

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// WS: per-sample weights and bias row (GroupNorm folded into a 1x1 convolution), the 128-row partner of igemm_pers_tile.h's WS
// variant: the same arithmetic (y = fp16(acc + t), the folded-LayerNorm epilogue with (mean, rstd) = (0, 1)).
// SC: a ResNet block's conv_shortcut folded into its conv2 as extra k steps on a second tensor pair (igemm_pers_tile.h): the
// same k order and arithmetic as the persistent tile's SC variant -> bit-identical tiles.
// KO: the (dy, 64-channel slab, dx) k order of a 3x3 stride-1 convolution (weights still [Cout][tap][Cin], visited in that order):
// the 128-row partner of igemm_pers_tr.h, whose three dx steps of a (dy, slab) share one activation stage -> bit-identical tiles.
template <int WC, int EPI, int NI, bool SPLIT = false, bool LN = false, bool WS = false, bool SC = false, bool KO = false>
__global__ __launch_bounds__(128 * WC, 2)
void igemm_kernel(IGemmParams p) {
    constexpr int WP = 2;
    constexpr int NW = WP * WC;
    constexpr int TP = 64 * WP, TC = 16 * NI * WC;
    constexpr int WBYTES = TC * 128, XBYTES = TP * 128, STAGE = WBYTES + XBYTES;
    constexpr int WG = TC / 8, XG = TP / 8;
    constexpr int WI = WG / NW, XI = XG / NW;               // exact: 5 + 4 (WC=2), 5 + 2 (WC=4)
    static_assert(WG % NW == 0 && XG % NW == 0, "uniform LDS-DMA count per wave required");
    constexpr int NL = WI + XI;
    static_assert(NL <= 2 * NI, "one LDS-DMA piece per MFMA group");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wid % WC;
    const int wp = wid / WC;

    const int tiles_c = p.Cout / TC;
    const int nblk = gridDim.x;
    int v;
    {
        const int b = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = b & 7, loc = b >> 3;
        v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    int split = 0;
    if (SPLIT) { split = v % p.ksplit; v = v / p.ksplit; }    // the parts of a tile sit next to each other (same XCD)
    const int pt = v / tiles_c;
    const int ct = v - pt * tiles_c;
    const int p0 = pt * TP;
    const int c0out = ct * TC;

    const int C1 = p.C1;
    const int C2 = p.Cin - C1;
    const int ntaps = (p.mode == IG_DENSE) ? 1 : 9;
    const int cpt = p.Cin / BK;
    const int cpt_sc = SC ? p.Csc / BK : 0;
    const int nk_all = ntaps * cpt + cpt_sc;
    const int kt_begin = SPLIT ? split * (nk_all / p.ksplit) : 0;          // ksplit divides nk_all (launcher)
    const int nk = SPLIT ? nk_all / p.ksplit : nk_all;
    const int Ktot = ntaps * p.Cin + (SC ? p.Csc : 0);
    const int OHW = p.OH * p.OW;

    const int lrow = lane >> 3;
    const int lchunk = ((lane & 7) ^ lrow) * 8;

    const f16* wsrc[WI];
#pragma unroll
    for (int k = 0; k < WI; ++k)
        wsrc[k] = p.Wp + (size_t)(c0out + (wid + k * NW) * 8 + lrow) * Ktot + lchunk + (SPLIT ? kt_begin * BK : 0) +
                  (WS ? (size_t)(p0 / p.rows_per_sample) * (size_t)p.w_sample_stride : (size_t)0);
    int xn[XI], xoh[XI], xow[XI];
#pragma unroll
    for (int k = 0; k < XI; ++k) {
        const int m = p0 + (wid + k * NW) * 8 + lrow;
        if (m < p.M) {
            const int n = m / OHW;
            const int rem = m - n * OHW;
            const int oh = rem / p.OW;
            xn[k] = n; xoh[k] = oh; xow[k] = rem - oh * p.OW;
        } else { xn[k] = -1; xoh[k] = 0; xow[k] = 0; }
    }
    const float sh = (float)p.H / (float)p.OH;
    const float sw = (float)p.W / (float)p.OW;
    const f16* zero = reinterpret_cast<const f16*>(g_zero_page) + lchunk;

    // per-lane source pointer of each activation row for the tile about to be loaded; advanced by
    // BK per tile inside a tap (0 for zero-page rows), recomputed when the tap or the source changes
    const f16* xsrc[XI];
    int xinc[XI];
    long long xpix[XI];
    auto set_tap = [&](int tap) {
        const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
        for (int k = 0; k < XI; ++k) {
            long long off = -1;
            if (xn[k] >= 0) {
                if (p.mode == IG_DENSE) {
                    off = (long long)(p0 + (wid + k * NW) * 8 + lrow);
                } else if (p.mode == IG_CONV3 || p.mode == IG_CONV3_S2 || (NI == 4 && p.mode == IG_CONV3_S2P0)) {
                    const bool p0s2 = (NI == 4 && p.mode == IG_CONV3_S2P0);     // pad (0,1,0,1), stride 2 (VAE)
                    const int st = (p.mode == IG_CONV3_S2 || p0s2) ? 2 : 1;
                    const int pd = p0s2 ? 0 : 1;
                    const int ih = xoh[k] * st + dy - pd, iw = xow[k] * st + dx - pd;
                    if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) off = ((long long)xn[k] * p.H + ih) * p.W + iw;
                } else {
                    const int uh = xoh[k] + dy - 1, uw = xow[k] + dx - 1;
                    if (uh >= 0 && uh < p.OH && uw >= 0 && uw < p.OW) {
                        int ih = (int)floorf((float)uh * sh); ih = ih < p.H - 1 ? ih : p.H - 1;
                        int iw = (int)floorf((float)uw * sw); iw = iw < p.W - 1 ? iw : p.W - 1;
                        off = ((long long)xn[k] * p.H + ih) * p.W + iw;
                    }
                }
            }
            xpix[k] = off;
            xsrc[k] = (off >= 0) ? (p.X + off * C1 + lchunk) : zero;
            xinc[k] = (off >= 0) ? BK : 0;
        }
    };
    int ld_tap = SPLIT ? kt_begin / cpt : 0, ld_cc = SPLIT ? kt_begin % cpt : 0;
    bool first_prepare = true;
    int ld_dx = 0, xlr = 0, w_adv = BK;                      // KO: ld_tap counts dy; xpix[] = centre pixel (dx = 1) of the row's dy line
    auto set_rows = [&](int dy) {                            // KO: centre pixels of the line dy, validity of the left / right neighbour
        xlr = 0;
#pragma unroll
        for (int k = 0; k < XI; ++k) {
            long long pix = -1;
            if (xn[k] >= 0) {
                const int ih = xoh[k] + dy - 1, iw = xow[k];            // in output (= up-sampled) coordinates
                if (ih >= 0 && ih < p.OH) {
                    // IG_CONV3_UP (exact 2x): the row of the source image; the column is added per dx (neighbours are not adjacent)
                    pix = (p.mode == IG_CONV3_UP) ? ((long long)xn[k] * p.H + (ih >> 1)) * p.W : ((long long)xn[k] * p.H + ih) * p.W + iw;
                    xlr |= ((iw >= 1) ? 1 : 0) << (2 * k);
                    xlr |= ((iw + 1 < p.OW) ? 2 : 0) << (2 * k);
                }
            }
            xpix[k] = pix;
        }
    };
    auto prepare_ko = [&]() {
        if (ld_cc == 0 && ld_dx == 0) set_rows(ld_tap);
        const int ch = ld_cc * BK;
        const bool second = ch >= C1;
        const f16* base = second ? p.X2 : p.X;
        const long long cs = second ? C2 : C1;
        const int cho = second ? ch - C1 : ch;
#pragma unroll
        for (int k = 0; k < XI; ++k) {
            const bool ok = xpix[k] >= 0 && (ld_dx == 1 || ((xlr >> (2 * k + (ld_dx >> 1))) & 1) != 0);
            const long long px = (p.mode == IG_CONV3_UP) ? xpix[k] + ((xow[k] + ld_dx - 1) >> 1) : xpix[k] + ld_dx - 1;
            xsrc[k] = ok ? base + px * cs + cho + lchunk : zero;
            xinc[k] = 0;
        }
        // weight offset of the NEXT k step relative to this one: dx + 1 = the next tap (+ Cin); after dx = 2 back to the first
        // tap of the line at the next slab (+ 64 - 2 Cin); after the last slab the next line starts 64 further
        w_adv = (ld_dx < 2) ? p.Cin : ((ld_cc == cpt - 1) ? BK : BK - 2 * p.Cin);
        if (++ld_dx == 3) { ld_dx = 0; if (++ld_cc == cpt) { ld_cc = 0; ++ld_tap; } }
    };
    auto prepare = [&]() {                                   // pointers for the next tile to load
        if constexpr (KO) { prepare_ko(); return; }
        if constexpr (SC) {
            if (ld_tap == ntaps) {                           // the folded second GEMM: centre tap (dense: the row) on cat([X3, X4])
                if (ld_cc == 0) {
                    set_tap(p.mode == IG_DENSE ? 0 : 4);
#pragma unroll
                    for (int k = 0; k < XI; ++k) if (xpix[k] >= 0) xsrc[k] = p.X3 + xpix[k] * p.C3 + lchunk;
                } else if (ld_cc * BK == p.C3) {
#pragma unroll
                    for (int k = 0; k < XI; ++k) if (xpix[k] >= 0) xsrc[k] = p.X4 + xpix[k] * (p.Csc - p.C3) + lchunk;
                }
                first_prepare = false;
                if (++ld_cc == cpt_sc) { ld_cc = 0; ++ld_tap; }
                return;
            }
        }
        if (SPLIT && first_prepare && ld_cc != 0) {
            // a part may start in the middle of a tap: set the tap, then move to the right channel slab
            set_tap(ld_tap);
            const bool second = ld_cc * BK >= C1;
#pragma unroll
            for (int k = 0; k < XI; ++k)
                if (xpix[k] >= 0) xsrc[k] = second ? (p.X2 + xpix[k] * C2 + (ld_cc * BK - C1) + lchunk) : (xsrc[k] + ld_cc * BK);
        } else if (ld_cc == 0) set_tap(ld_tap);
        else if (ld_cc * BK == C1) {
#pragma unroll
            for (int k = 0; k < XI; ++k) if (xpix[k] >= 0) xsrc[k] = p.X2 + xpix[k] * C2 + lchunk;
        }
        first_prepare = false;
        if (++ld_cc == cpt) { ld_cc = 0; ++ld_tap; }
    };
    auto load_piece = [&](int buf, int idx) {                // idx in [0, NL): W pieces first, then X
        char* wt = smem + buf * STAGE;
        if (idx < WI) {
            __builtin_amdgcn_global_load_lds((gptr_t)wsrc[idx], (lptr_t)(wt + (wid + idx * NW) * 1024), 16, 0, 0);
            wsrc[idx] += KO ? w_adv : BK;
        } else {
            const int k = idx - WI;
            __builtin_amdgcn_global_load_lds((gptr_t)xsrc[k], (lptr_t)(wt + WBYTES + (wid + k * NW) * 1024), 16, 0, 0);
            xsrc[k] += xinc[k];
        }
    };

    floatx4 acc[NI][4];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int l15 = lane & 15;
    const int lg = lane >> 4;
    const int a_row_off = (wc * (16 * NI) + l15) * 128;
    const int b_row_off = (wp * 64 + l15) * 128;
    const int koff0 = ((lg ^ (l15 & 7)) << 4), koff1 = (((4 + lg) ^ (l15 & 7)) << 4);

    // bias of this tile's channels -> LDS behind the operand stages (zeros without a bias); read back in
    // the epilogue, visible after the first barrier of the k loop
    if (tid < TC / 4) {
        half4 bv = half4{0, 0, 0, 0};
        if (p.bias) bv = *reinterpret_cast<const half4*>(p.bias + c0out + tid * 4);
        *reinterpret_cast<half4*>(smem + 2 * STAGE + tid * 8) = bv;
        if (LN) {      // ln_s, ln_t of the tile's channels (fp32) behind the 1 KB bias slot
            const float* lt = p.ln_t + (WS ? (size_t)(p0 / p.rows_per_sample) * p.Cout : (size_t)0);      // WS: the sample's bias row
            *reinterpret_cast<floatx4*>(smem + 2 * STAGE + 1024 + tid * 16) = *reinterpret_cast<const floatx4*>((WS ? lt : p.ln_s) + c0out + tid * 4);
            *reinterpret_cast<floatx4*>(smem + 2 * STAGE + 1024 + TC * 4 + tid * 16) = *reinterpret_cast<const floatx4*>(lt + c0out + tid * 4);
        }
    }
    if (LN && WS && tid < TP) *reinterpret_cast<float2*>(smem + 2 * STAGE + 1024 + 8 * TC + tid * 8) = float2{0.f, 1.f};
    if (LN && !WS && p.ln_stats && tid < TP) {      // per-row (mean, rstd) of the tile's rows behind them
        int m = p0 + tid;
        m = m < p.M ? m : p.M - 1;
        *reinterpret_cast<float2*>(smem + 2 * STAGE + 1024 + 8 * TC + tid * 8) = *reinterpret_cast<const float2*>(p.ln_stats + 2 * (size_t)m);
    }
    // without a statistics kernel (p.ln_stats == nullptr): thread pair (2r, 2r + 1) accumulates row r of the tile from the k
    // loop's LDS tiles — the same numbers in the same order as igemm_pers_tile.h (bit-identical statistics)
    const bool ln_ink = LN && !WS && p.ln_stats == nullptr && tid < 2 * TP;
    float ln_s1 = 0.f, ln_s2 = 0.f;
    prepare();
#pragma unroll
    for (int i = 0; i < NL; ++i) load_piece(0, i);

    auto step = [&](int cur, bool more) {
        const char* wt = smem + cur * STAGE;
        const char* xt = wt + WBYTES;
        half8 a0[NI], b0[4], a1[NI], b1[4];
#pragma unroll
        for (int i = 0; i < NI; ++i) a0[i] = *reinterpret_cast<const half8*>(wt + a_row_off + i * 2048 + koff0);
#pragma unroll
        for (int j = 0; j < 4; ++j) b0[j] = *reinterpret_cast<const half8*>(xt + b_row_off + j * 2048 + koff0);
        if (more) prepare();
        // 2*NI groups of 4 MFMAs; one LDS-DMA piece after each of the first NL groups
#pragma unroll
        for (int g = 0; g < 2 * NI; ++g) {
            const int i = g % NI;
            if (g == 2) {
#pragma unroll
                for (int ii = 0; ii < NI; ++ii) a1[ii] = *reinterpret_cast<const half8*>(wt + a_row_off + ii * 2048 + koff1);
#pragma unroll
                for (int j = 0; j < 4; ++j) b1[j] = *reinterpret_cast<const half8*>(xt + b_row_off + j * 2048 + koff1);
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = (DM_MFMA_SNAKE && (g & 1)) ? 3 - jj : jj;        // snake order: one operand changes per MFMA (igemm_pers_tile.h)
                acc[i][j] = (g < NI) ? __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[i], b0[j], acc[i][j], 0, 0, 0)
                                    : __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[i], b1[j], acc[i][j], 0, 0, 0);
            }
            if (more && g < NL) load_piece(cur ^ 1, g);
            if (LN && g >= 2 && g < 6) {
                if (ln_ink) {
                    const int row = tid >> 1, hf = tid & 1;
                    const half8 xv = *reinterpret_cast<const half8*>(xt + row * 128 + (((4 * hf + (g - 2)) ^ (row & 7)) << 4));
#pragma unroll
                    for (int k = 0; k < 8; k += 2) {          // v_dot2_f32_f16: two exact fp16 products + the fp32 accumulator per instruction
                        const ln_half2 v2 = ln_half2{xv[k], xv[k + 1]};
                        ln_s1 = __builtin_amdgcn_fdot2(v2, ln_half2{(_Float16)1.0f, (_Float16)1.0f}, ln_s1, false);
                        ln_s2 = __builtin_amdgcn_fdot2(v2, v2, ln_s2, false);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    for (int kt = 0; kt < nk - 1; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        step(kt & 1, true);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    step((nk - 1) & 1, false);
    if (SPLIT) {
        // fp32 partial tile straight from the fragments (16 bytes per lane): partial[split][m][c]
        float* base = p.partial + (size_t)split * p.M * p.Cout;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = p0 + wp * 64 + 16 * j + l15;
            if (m >= p.M) continue;
#pragma unroll
            for (int i = 0; i < NI; ++i)
                *reinterpret_cast<floatx4*>(base + (size_t)m * p.Cout + c0out + wc * (16 * NI) + 16 * i + 4 * lg) = acc[i][j];
        }
        return;
    }
    if (LN && !WS && p.ln_stats == nullptr) {
        const float t1 = ln_s1 + __shfl_xor(ln_s1, 1), t2 = ln_s2 + __shfl_xor(ln_s2, 1);
        const float mean = t1 / (float)p.Cin;
        float var = t2 / (float)p.Cin - mean * mean;
        var = var > 0.f ? var : 0.f;
        if (ln_ink && mean * mean > LN_REDO_RATIO2 * var) {      // cancellation: exact second pass over this row (dm_kernels.h)
            int m = p0 + (tid >> 1);
            m = m < p.M ? m : p.M - 1;
            const int half_c = p.Cin >> 1;
            const f16* xr = p.X + (size_t)m * p.Cin + (tid & 1) * half_c;
            float q = 0.f;
            for (int c = 0; c < half_c; c += 8) {
                const half8 v = *reinterpret_cast<const half8*>(xr + c);
#pragma unroll
                for (int k = 0; k < 8; ++k) { const float d = (float)v[k] - mean; q = __builtin_fmaf(d, d, q); }
            }
            var = (q + __shfl_xor(q, 1)) / (float)p.Cin;
        }
        if (ln_ink && (tid & 1) == 0)
            *reinterpret_cast<float2*>(smem + 2 * STAGE + 1024 + 8 * TC + (tid >> 1) * 8) = float2{mean, rsqrtf(var + p.ln_eps)};
        __syncthreads();
    }
    epilogue_lds<EPI, 128 * WC, TP, TC, NI, LN>(p, acc, smem, smem + 2 * STAGE, p0, c0out, wc, wp, l15, lg, OHW);
}

}  // namespace

template <int WC, int NI, bool LN = false, bool WS = false, bool SC = false, bool KO = false>
static hipError_t launch_t(const IGemmParams& p, hipStream_t s) {
    constexpr int TP = 128, TC = 16 * NI * WC;
    constexpr size_t lds = 2 * (size_t)(TP + TC) * 128 + 1024 + (LN ? 8 * TC + 8 * TP : 0);      // operand stages + bias (+ ln_s, ln_t, row stats)
    const int tiles_p = (p.M + TP - 1) / TP;
    const int tiles_c = p.Cout / TC;
    dim3 grid(tiles_p * tiles_c), block(128 * WC);
    static std::atomic<uint64_t> attr_seen{0};      // hipFuncSetAttribute is per DEVICE, not per process
    if (first_use_on_device(attr_seen)) {
        (void)hipFuncSetAttribute((const void*)igemm_kernel<WC, EPI_PLAIN, NI, false, LN, WS, SC, KO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)igemm_kernel<WC, EPI_GEGLU, NI, false, LN, WS, SC, KO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (p.epi == EPI_GEGLU)
        launch_timed((igemm_kernel<WC, EPI_GEGLU, NI, false, LN, WS, SC, KO>), grid, block, lds, s, p);
    else
        launch_timed((igemm_kernel<WC, EPI_PLAIN, NI, false, LN, WS, SC, KO>), grid, block, lds, s, p);
    return hipGetLastError();
}

}  // namespace dm
