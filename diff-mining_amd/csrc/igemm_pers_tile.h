// igemm_pers_tile.h — persistent form of the 256 px x 320 ch implicit-GEMM tile (formulation, operand layout and
// swizzle: igemm.hip).  One block per CU walks a strided list of tiles; the k loop is continuous across tiles:
//
//   * while the last k step of tile i runs, the LDS-DMA of tile i+1's first k step goes into the other stage, and
//     right after it (one barrier) the second k step goes into the stage just consumed — the epilogue below needs
//     no LDS, so both stages prefetch under it: no prologue bubble per tile (≈ 4-5 k cycles of 20-50 k for the
//     K = 320..1280 linears);
//   * the epilogue writes straight from the accumulators.  A lane of the MFMA C layout holds 4 consecutive
//     channels of one pixel per 16x16 block (8 B after fp16 packing); v_permlane16_swap + v_permlane32_swap
//     transpose the four lane groups against four channel blocks, after which a lane holds 16 consecutive
//     channels of its pixel = two 16-byte stores (the store tail is issue-bound per instruction, not per byte).
//     No staging pass, no epilogue barriers; the residual is read in the same layout, one unit ahead of the
//     stores (vmcnt retires in order: a load queued behind stores would wait for them);
//   * the stores are never waited for inside the epilogue: the next tile's first two k steps were fetched BEFORE
//     them, so `s_waitcnt vmcnt(<stores per wave>)` at the top of the next tile proves the operands landed while
//     the stores drain under two k steps.  For the count to be exact every wave issues every store: rows beyond M
//     are redirected to a sink page instead of being predicated off;
//   * bias / time-embedding row / folded-LayerNorm vectors of a tile reach LDS by LDS-DMA too (double-buffered
//     8 KB slots behind the stages), issued with the prefetch of the tile that needs them.
//
// Arithmetic (k order, MFMA sequence, rounding points of bias -> fp16, + time embedding -> fp16, + residual -> fp16,
// GEGLU, folded LayerNorm) is identical to igemm_big_tile.h / igemm_tile.h: results are bit-identical, so a
// sample's output does not depend on which tile geometry its batch size selects.
#pragma once
#include "dm_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace dm {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2_ __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned uintx2 __attribute__((ext_vector_type(2)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK = 64;

// erf-GELU x * Phi(x): see igemm_big_tile.h (Abramowitz-Stegun 7.1.26, |erf error| < 1.5e-7)
__device__ __forceinline__ float gelu_erf(float x) {
    const float ax = __builtin_fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ax, 1.0f));
    float poly = __builtin_fmaf(t, 1.061405429f, -1.453152027f);
    poly = __builtin_fmaf(t, poly, 1.421413741f);
    poly = __builtin_fmaf(t, poly, -0.284496736f);
    poly = __builtin_fmaf(t, poly, 0.254829592f);
    poly *= t;
    const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.44269504088896340736f);
    const float half_tail = 0.5f * poly * e;
    const float phi = (x < 0.f) ? half_tail : 1.0f - half_tail;
    return x * phi;
}

__device__ __attribute__((aligned(256))) unsigned char g_zero_page_pers[1024];
__device__ __attribute__((aligned(256))) unsigned char g_store_sink[8 * 64 * 16];      // one 16-byte slot per (wave, lane)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void swap16(unsigned& a, unsigned& b) {      // a.row1 <-> b.row0, a.row3 <-> b.row2 (rows of 16 lanes)
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0]; b = r[1];
}
__device__ __forceinline__ void swap32(unsigned& a, unsigned& b) {      // a.lanes[32..63] <-> b.lanes[0..31]
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0]; b = r[1];
}
// 4 x 4 transpose of NWD-dword elements between the four 16-lane groups and four registers:
// in: R[i] on lane group g = element (block i, quarter g); out: R[q] on lane group g = element (block g, quarter q)
template <int NWD>
__device__ __forceinline__ void transpose4(unsigned (&R)[4][NWD]) {
#pragma unroll
    for (int w = 0; w < NWD; ++w) { swap16(R[0][w], R[1][w]); swap16(R[2][w], R[3][w]); }
#pragma unroll
    for (int w = 0; w < NWD; ++w) { swap32(R[0][w], R[2][w]); swap32(R[1][w], R[3][w]); }
}

// LDS slot of a tile's per-channel / per-row vectors (filled by LDS-DMA, 1 KiB pieces)
constexpr int AUX_BIAS = 0, AUX_TEMB = 1024, AUX_LNS = 2048, AUX_LNT = 4096, AUX_STATS = 6144, AUX_BYTES = 8192;

template <int EPI, bool LN>
__global__ __launch_bounds__(512, 2)
void igemm_pers_kernel(IGemmParams p, int ntiles) {
    constexpr int WC = 2, CH = 2, NW = 8;
    constexpr int TP = 256, TC = 320;
    constexpr int WBYTES = TC * 128, XBYTES = TP * 128, STAGE = WBYTES + XBYTES;
    constexpr int WI = TC / 8 / NW, XI = TP / 8 / NW;          // 5 + 4 LDS-DMA pieces per wave per k step
    constexpr int NL = WI + XI;
    // stores per wave per tile (every wave issues all of them: rows beyond M go to the sink page)
    constexpr int NSTORE = (EPI == EPI_GEGLU) ? 16 : 24;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const aux0 = smem + 2 * STAGE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wid % WC;
    const int wp = wid / WC;

    // ---- this block's tiles: XCD x (= block % 8) owns a contiguous range; its blocks stride through it ----------
    const int tiles_c = p.Cout / TC;
    int tile, tend, tstride;
    {
        const int nblk = gridDim.x;
        const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
        const int q = ntiles >> 3, r = ntiles & 7;
        const int tstart = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        tend = tstart + q + (xcd < r ? 1 : 0);
        tstride = (nblk - xcd + 7) >> 3;
        tile = tstart + loc;
    }
    if (tile >= tend) return;

    const int C1 = p.C1;
    const int C2 = p.Cin - C1;
    const int ntaps = (p.mode == IG_DENSE) ? 1 : 9;
    const int cpt = p.Cin / BK;
    const int nk = ntaps * cpt;                                  // >= 3 (launcher)
    const int Ktot = ntaps * p.Cin;
    const int OHW = p.OH * p.OW;
    const bool temb_lds = p.temb && (OHW % TP == 0);             // a tile lies inside one sample: its temb row goes through LDS

    const int lrow = lane >> 3;
    const int lchunk = ((lane & 7) ^ lrow) * 8;
    const f16* zero = reinterpret_cast<const f16*>(g_zero_page_pers) + lchunk;

    // ---- load-side state of the tile whose operands are being fetched ---------------------------------------------
    int lp0 = 0, lc0 = 0;
    int xohw[XI], xnb[XI], xoff[XI];
    unsigned woff = 0;
    const unsigned wstride = (unsigned)(NW * 8) * (unsigned)Ktot;
    const f16* xbase = p.X;
    int ld_tap = 0, ld_cc = 0;
    const float sh = (float)p.H / (float)p.OH;
    const float sw = (float)p.W / (float)p.OW;

    auto set_tile = [&](int tl) __attribute__((always_inline)) {
        const int pt = tl / tiles_c;
        lp0 = pt * TP;
        lc0 = (tl - pt * tiles_c) * TC;
#pragma unroll
        for (int k = 0; k < XI; ++k) {
            const int m = lp0 + (wid + k * NW) * 8 + lrow;
            if (m < p.M) {
                if (p.mode == IG_DENSE) { xohw[k] = 0; xnb[k] = m; }
                else {
                    const int n = m / OHW;
                    const int rem = m - n * OHW;
                    const int oh = rem / p.OW;
                    xohw[k] = (oh << 16) | (rem - oh * p.OW);
                    xnb[k] = n * p.H * p.W;
                }
            } else { xohw[k] = -1; xnb[k] = 0; }
        }
        woff = (unsigned)((size_t)(lc0 + wid * 8 + lrow) * Ktot + lchunk);
        ld_tap = 0; ld_cc = 0;
    };
    auto src_pixel = [&](int k, int dy, int dx) __attribute__((always_inline)) -> int {
        if (xohw[k] < 0) return -1;
        const int oh = xohw[k] >> 16, ow = xohw[k] & 0xffff;
        if (p.mode == IG_DENSE) return xnb[k];
        if (p.mode == IG_CONV3 || p.mode == IG_CONV3_S2) {
            const int st = (p.mode == IG_CONV3_S2) ? 2 : 1;
            const int ih = oh * st + dy - 1, iw = ow * st + dx - 1;
            return (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) ? xnb[k] + ih * p.W + iw : -1;
        }
        const int uh = oh + dy - 1, uw = ow + dx - 1;              // conv on the nearest-upsampled image
        if (uh < 0 || uh >= p.OH || uw < 0 || uw >= p.OW) return -1;
        int ih = (int)floorf((float)uh * sh); ih = ih < p.H - 1 ? ih : p.H - 1;
        int iw = (int)floorf((float)uw * sw); iw = iw < p.W - 1 ? iw : p.W - 1;
        return xnb[k] + ih * p.W + iw;
    };
    auto set_src = [&](int tap, int cs) __attribute__((always_inline)) {
        const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
        for (int k = 0; k < XI; ++k) {
            const int pix = src_pixel(k, dy, dx);
            xoff[k] = (pix >= 0) ? (pix * cs + lchunk) : -1;
        }
    };
    auto prepare = [&]() __attribute__((always_inline)) {    // sources of the next k tile to load
        if (ld_cc == 0) { xbase = p.X; set_src(ld_tap, C1); }
        else if (ld_cc * BK == C1) { xbase = p.X2; set_src(ld_tap, C2); }
        if (++ld_cc == cpt) { ld_cc = 0; ++ld_tap; }
    };
    auto load_piece = [&](int buf, int idx) __attribute__((always_inline)) {   // idx in [0, NL): W pieces, then X
        char* wt = smem + buf * STAGE;
        if (idx < WI) {
            lptr_t dst = (lptr_t)(wt + (wid + idx * NW) * 1024);
            __builtin_amdgcn_global_load_lds((gptr_t)(p.Wp + (size_t)(woff + (unsigned)idx * wstride)), dst, 16, 0, 0);
            if (idx == WI - 1) woff += BK;
        } else {
            const int k = idx - WI;
            lptr_t dst = (lptr_t)(wt + WBYTES + (wid + k * NW) * 1024);
            const f16* a = (xoff[k] >= 0) ? (xbase + (size_t)(unsigned)xoff[k]) : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)a, dst, 16, 0, 0);
            if (xoff[k] >= 0) xoff[k] += BK;
        }
    };
    // per-tile vectors -> LDS slot by LDS-DMA (lane-linear 1 KiB pieces; lanes past the vector read the zero page)
    auto load_aux = [&](int slot) __attribute__((always_inline)) {
        char* ax = aux0 + slot * AUX_BYTES;
        const char* zp = reinterpret_cast<const char*>(g_zero_page_pers) + lane * 16;
        if (wid == 0) {
            const char* s = (p.bias && lane < TC / 8) ? reinterpret_cast<const char*>(p.bias + lc0) + lane * 16 : zp;
            __builtin_amdgcn_global_load_lds((gptr_t)s, (lptr_t)(ax + AUX_BIAS), 16, 0, 0);
        } else if (wid == 1) {
            if (temb_lds) {
                const int n = lp0 / OHW;
                const char* s = (lane < TC / 8) ? reinterpret_cast<const char*>(p.temb + (size_t)n * p.temb_ld + lc0) + lane * 16 : zp;
                __builtin_amdgcn_global_load_lds((gptr_t)s, (lptr_t)(ax + AUX_TEMB), 16, 0, 0);
            }
        } else if (LN && wid < 8) {
            // wid 2,3: ln_s halves; 4,5: ln_t halves; 6,7: row statistics halves (1 KiB = 256 floats / 128 rows each)
            const int part = (wid - 2) >> 1, half = (wid - 2) & 1;
            const char* s;
            if (part < 2) {
                const float* v = (part == 0 ? p.ln_s : p.ln_t) + lc0;
                const int f = half * 256 + lane * 4;
                s = (f < TC) ? reinterpret_cast<const char*>(v + f) : zp;
            } else {
                int row = lp0 + half * 128 + lane * 2;               // two (mean, rstd) pairs per lane
                row = row < p.M - 1 ? row : p.M - 2;
                s = reinterpret_cast<const char*>(p.ln_stats + 2 * (size_t)row);
            }
            __builtin_amdgcn_global_load_lds((gptr_t)s, (lptr_t)(ax + AUX_LNS + (wid - 2) * 1024), 16, 0, 0);
        }
    };

    floatx4 acc[CH][5][4];
    const int l15 = lane & 15;
    const int lg = lane >> 4;
    const int a_row_off = (wc * 80 * CH + l15) * 128;
    const int b_row_off = (wp * 64 + l15) * 128;
    const int koff0 = ((lg ^ (l15 & 7)) << 4), koff1 = (((4 + lg) ^ (l15 & 7)) << 4);

    // one k step on stage `cur`; issue == 1: LDS-DMA of the following k tile into the other stage, interleaved
    auto step = [&](int cur, bool issue) __attribute__((always_inline)) {
        const char* wt = smem + cur * STAGE;
        const char* xt = wt + WBYTES;
        half8 b0[4], b1[4], a[5];
#pragma unroll
        for (int j = 0; j < 4; ++j) b0[j] = *reinterpret_cast<const half8*>(xt + b_row_off + j * 2048 + koff0);
#pragma unroll
        for (int i = 0; i < 5; ++i) a[i] = *reinterpret_cast<const half8*>(wt + a_row_off + i * 2048 + koff0);
        if (issue) prepare();
        int piece = 0;
#pragma unroll
        for (int q = 0; q < 2 * CH; ++q) {
            const int sidx = q / CH, h = q % CH;
            if (q == CH - 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) b1[j] = *reinterpret_cast<const half8*>(xt + b_row_off + j * 2048 + koff1);
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[h][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], sidx ? b1[j] : b0[j], acc[h][i][j], 0, 0, 0);
                if (q + 1 < 2 * CH) {          // fragment i of the next quarter replaces the one just consumed
                    const int nq = q + 1, ns = nq / CH, nh = nq % CH;
                    a[i] = *reinterpret_cast<const half8*>(wt + a_row_off + nh * (80 * 128) + i * 2048 + (ns ? koff1 : koff0));
                }
                const int g = q * 5 + i;
                if (issue && (g & 1) == 0 && piece < NL) { load_piece(cur ^ 1, piece); ++piece; }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // ---- epilogue: straight from the accumulators ---------------------------------------------------------------
    auto epilogue = [&](int p0, int c0out, int slot) __attribute__((always_inline)) {
        const char* ax = aux0 + slot * AUX_BYTES;
        constexpr int OCH = (EPI == EPI_GEGLU) ? 40 : 80;          // output channels of one (wave, h) sub-tile
        const int c0o = (EPI == EPI_GEGLU) ? c0out / 2 : c0out;
        f16* const sink = reinterpret_cast<f16*>(g_store_sink) + (wid * 64 + lane) * 8;
        // residual of unit u = (h, j), in the layout of the stores: 16 channels of block lg (two half8) + block 4's quarter
        half8 rlo, rhi; half4 r4;
        auto load_res = [&](int h, int j) __attribute__((always_inline)) {
            int m = p0 + wp * 64 + 16 * j + l15;
            m = m < p.M ? m : p.M - 1;
            const f16* rp = p.res + (size_t)m * p.ldres + c0o + wc * (OCH * CH) + h * OCH;
            rlo = *reinterpret_cast<const half8*>(rp + 16 * lg);
            rhi = *reinterpret_cast<const half8*>(rp + 16 * lg + 8);
            r4 = *reinterpret_cast<const half4*>(rp + 64 + 4 * lg);
        };
        const bool has_res = (EPI != EPI_GEGLU) && p.res;
        if (has_res) load_res(0, 0);
#pragma unroll
        for (int h = 0; h < CH; ++h) {
            float bz[5][4], sz[LN ? 5 : 1][4];
            half4 tv[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int ct = wc * 80 * CH + h * 80 + 16 * i + 4 * lg;          // tile-local GEMM channel
                if (LN) {
                    const floatx4 t4 = *reinterpret_cast<const floatx4*>(ax + AUX_LNT + ct * 4);
                    const floatx4 s4 = *reinterpret_cast<const floatx4*>(ax + AUX_LNS + ct * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { bz[i][r] = t4[r]; sz[i][r] = s4[r]; }
                } else {
                    const half4 bv = *reinterpret_cast<const half4*>(ax + AUX_BIAS + ct * 2);
#pragma unroll
                    for (int r = 0; r < 4; ++r) bz[i][r] = (float)bv[r];
                }
                if (EPI != EPI_GEGLU && temb_lds) tv[i] = *reinterpret_cast<const half4*>(ax + AUX_TEMB + ct * 2);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int pr = wp * 64 + 16 * j + l15;
                const int m = p0 + pr;
                float mu = 0.f, rs = 1.f;
                if (LN) {
                    const float2 st = *reinterpret_cast<const float2*>(ax + AUX_STATS + pr * 8);
                    mu = st.x; rs = st.y;
                }
                if (EPI != EPI_GEGLU && p.temb && !temb_lds) {          // tile straddles samples: per-row time-embedding loads
                    const int n = (m < p.M) ? (m / OHW) : 0;
                    const f16* tp = p.temb + (size_t)n * p.temb_ld + c0out + wc * 80 * CH + h * 80 + 4 * lg;
#pragma unroll
                    for (int i = 0; i < 5; ++i) tv[i] = *reinterpret_cast<const half4*>(tp + 16 * i);
                }
                constexpr int NWD = (EPI == EPI_GEGLU) ? 1 : 2;
                unsigned R[4][NWD], R4[NWD];
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        v[r] = LN ? rs * (acc[h][i][j][r] - mu * sz[i][r]) + bz[i][r] : acc[h][i][j][r] + bz[i][r];
                    unsigned w0, w1 = 0;
                    if (EPI == EPI_GEGLU) {
                        const f16 h0 = (f16)v[0], h1 = (f16)v[1], g0 = (f16)v[2], g1 = (f16)v[3];
                        const f16 q0 = (f16)gelu_erf((float)g0), q1 = (f16)gelu_erf((float)g1);
                        const half2_ o = half2_{(f16)((float)h0 * (float)q0), (f16)((float)h1 * (float)q1)};
                        w0 = __builtin_bit_cast(unsigned, o);
                    } else {
                        half4 o = half4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                        if (p.temb) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) o[r] = (f16)((float)o[r] + (float)tv[i][r]);
                        }
                        const uintx2 u = __builtin_bit_cast(uintx2, o);
                        w0 = u[0]; w1 = u[1];
                    }
                    if (i < 4) { R[i][0] = w0; if (NWD == 2) R[i][NWD - 1] = w1; }
                    else { R4[0] = w0; if (NWD == 2) R4[NWD - 1] = w1; }
                }
                transpose4<NWD>(R);            // lane group lg now holds quarters 0..3 of channel block lg
                f16* yp = (m < p.M) ? p.Y + (size_t)m * p.ldy + c0o + wc * (OCH * CH) + h * OCH : nullptr;
                if (EPI == EPI_GEGLU) {
                    const uintx4 o8 = uintx4{R[0][0], R[1][0], R[2][0], R[3][0]};           // 8 output channels of block lg
                    *reinterpret_cast<uintx4*>(yp ? yp + 8 * lg : sink) = o8;
                    *reinterpret_cast<unsigned*>(yp ? yp + 32 + 2 * lg : sink) = R4[0];
                } else {
                    half8 lo = __builtin_bit_cast(half8, uintx4{R[0][0], R[0][NWD - 1], R[1][0], R[1][NWD - 1]});
                    half8 hi = __builtin_bit_cast(half8, uintx4{R[2][0], R[2][NWD - 1], R[3][0], R[3][NWD - 1]});
                    half4 o4 = __builtin_bit_cast(half4, uintx2{R4[0], R4[NWD - 1]});
                    if (has_res) {
                        const half8 clo = rlo, chi = rhi; const half4 c4 = r4;
                        // the next unit's residual is requested BEFORE this unit's stores (in-order vmcnt)
                        if (j + 1 < 4) load_res(h, j + 1); else if (h + 1 < CH) load_res(h + 1, 0);
#pragma unroll
                        for (int r = 0; r < 8; ++r) { lo[r] = (f16)((float)lo[r] + (float)clo[r]); hi[r] = (f16)((float)hi[r] + (float)chi[r]); }
#pragma unroll
                        for (int r = 0; r < 4; ++r) o4[r] = (f16)((float)o4[r] + (float)c4[r]);
                    }
                    *reinterpret_cast<half8*>(yp ? yp + 16 * lg : sink) = lo;
                    *reinterpret_cast<half8*>(yp ? yp + 16 * lg + 8 : sink) = hi;
                    *reinterpret_cast<half4*>(yp ? yp + 64 + 4 * lg : sink) = o4;
                }
            }
        }
    };

    // ---- prologue: first tile's k steps 0 and 1 into stages 0 and 1, its vectors into slot 0 ----------------------
    set_tile(tile);
    load_aux(0);
    prepare();
#pragma unroll
    for (int i = 0; i < NL; ++i) load_piece(0, i);
    prepare();
#pragma unroll
    for (int i = 0; i < NL; ++i) load_piece(1, i);

    int base = 0;                     // stage of the current tile's k step 0
    int slot = 0;
    bool first = true;
    while (true) {
        const int pt = tile / tiles_c;
        const int p0 = pt * TP, c0out = (tile - pt * tiles_c) * TC;
        const int next = tile + tstride;
        const bool has_next = next < tend;
#pragma unroll
        for (int h = 0; h < CH; ++h)
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[h][i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

        // One copy of the k step for the whole stream (instruction-cache footprint); per k step kt of this tile:
        //   kt = 0      both of the tile's first stages were requested BEFORE the previous epilogue's stores, so "at most
        //               NSTORE outstanding" proves that they (and the tile's vectors) have landed; nothing to issue;
        //   kt = 1      operands landed with step 0's; every wave is done with stage `base`: refill it with k step 2;
        //   kt >= 2     wait for everything (this is where the previous tile's stores must have drained);
        //   last        the stream continues with the next tile's k step 0 into the other stage, then (one barrier later)
        //               its k step 1 into the stage just consumed.
        for (int kt = 0; kt < nk; ++kt) {
            const bool lastk = (kt == nk - 1);
            if (kt == 0 && !first) {
                if (NSTORE == 24) asm volatile("s_waitcnt vmcnt(24)\n\ts_barrier" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
            } else if (kt == 1) asm volatile("s_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            if (lastk && has_next) { set_tile(next); load_aux(slot ^ 1); }
            step((base + kt) & 1, kt >= 1 && (!lastk || has_next));
        }
        first = false;
        const int last = (base + nk - 1) & 1;
        if (has_next) {
            asm volatile("s_barrier" ::: "memory");            // every wave is done with stage `last`
            prepare();
#pragma unroll
            for (int i = 0; i < NL; ++i) load_piece(last, i);
        }
        epilogue(p0, c0out, slot);
        if (!has_next) break;
        tile = next;
        base = last ^ 1;
        slot ^= 1;
    }
}

}  // namespace

template <bool LN>
static hipError_t launch_igemm_pers_t(const IGemmParams& p, hipStream_t s) {
    constexpr int TP = 256, TC = 320;
    constexpr size_t lds = 2 * (size_t)(TP + TC) * 128 + 2 * AUX_BYTES;           // 160 KiB: two operand stages + two vector slots
    const int ntiles = ((p.M + TP - 1) / TP) * (p.Cout / TC);
    static int n_cu[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (n_cu[dev & 63] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n_cu[dev & 63] = v;
    }
    const int grid = ntiles < n_cu[dev & 63] ? ntiles : n_cu[dev & 63];
    static std::atomic<uint64_t> attr_seen{0};      // hipFuncSetAttribute is per DEVICE, not per process
    if (first_use_on_device(attr_seen)) {
        (void)hipFuncSetAttribute((const void*)igemm_pers_kernel<EPI_PLAIN, LN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)igemm_pers_kernel<EPI_GEGLU, LN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (p.epi == EPI_GEGLU) hipLaunchKernelGGL((igemm_pers_kernel<EPI_GEGLU, LN>), dim3(grid), dim3(512), lds, s, p, ntiles);
    else hipLaunchKernelGGL((igemm_pers_kernel<EPI_PLAIN, LN>), dim3(grid), dim3(512), lds, s, p, ntiles);
    return hipGetLastError();
}

}  // namespace dm
