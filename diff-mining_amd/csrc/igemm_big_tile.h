// igemm_big_tile.h — the 256 px x 320 ch tile of the implicit-GEMM (see igemm.hip for the formulation, the
// operand layout, the swizzle and the epilogues), shared by igemm_big.hip (plain) and igemm_big_ln.hip
// (LayerNorm folded into the epilogue).  Each instantiation set lives in its own translation unit on
// purpose: co-compiling large kernels perturbs the register allocation of both.
#pragma once
#include "dm_kernels.h"
#include <cstdio>
#include <type_traits>
#include <cstdlib>
#include <cstring>

namespace dm {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK = 64;

// erf-GELU  x * Phi(x),  Phi via Abramowitz-Stegun 7.1.26 (|erf error| < 1.5e-7, far below the fp16
// rounding of the result): 1 rcp + 1 exp2 + 7 FMAs instead of libm erff (~40 instructions), which
// dominated the GEGLU epilogue (40 calls per lane per tile).  The negative tail is formed as
// 0.5*poly*e directly (no 1 - 1 cancellation).
__device__ __forceinline__ float gelu_erf(float x) {
    const float ax = __builtin_fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ax, 1.0f));
    float poly = __builtin_fmaf(t, 1.061405429f, -1.453152027f);
    poly = __builtin_fmaf(t, poly, 1.421413741f);
    poly = __builtin_fmaf(t, poly, -0.284496736f);
    poly = __builtin_fmaf(t, poly, 0.254829592f);
    poly *= t;
    const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.44269504088896340736f);
    const float half_tail = 0.5f * poly * e;                 // = 0.5 * (1 - erf(|x|/sqrt2))
    const float phi = (x < 0.f) ? half_tail : 1.0f - half_tail;
    return x * phi;
}

// Epilogue staged through LDS: fragments (bias / time-embedding / GEGLU applied, rounded to fp16) are
// written to an LDS tile [TP px][TCO ch] (row stride padded by 8 B: conflict-free ds_write_b64 /
// b32), then copied out as whole rows with 16-byte stores (+ the residual read the same way).
// The direct fragment stores write 8-byte (GEGLU: 4-byte) pieces of 16 different 128-byte lines per
// instruction; on the wide, short-K linears that partial-line traffic bound the whole kernel.
// With CH > 1 channel sub-tiles per wave the block tile is written in CH passes: pass h stages the
// sub-tile h of every wave ([TP][TC] with TC = 80 * channel-waves) and maps staged column blocks of
// OB channels back to global channel  c0 + (col / OB) * OB * CH + h * OB + col % OB.
#ifdef DM_IGEMM_TIMING
__device__ long long g_igemm_dbg[16];
#endif

template <int EPI, int NTH, int TP, int TC, int CH, bool LN = false>
__device__ __forceinline__ void epilogue_lds(const IGemmParams& p, floatx4 (&acc)[5][4], char* smem, int p0,
                                             int c0out, int wc, int wp, int l15, int lg, int OHW, int h, const char* bias_lds
#ifdef DM_IGEMM_TIMING
                                             , long long* dbg, long long& tlast
#define ETICK(i) do { const long long _n = (long long)__builtin_readcyclecounter(); dbg[i] += _n - tlast; tlast = _n; } while (0)
#else
#define ETICK(i) do {} while (0)
#endif
                                             ) {
    constexpr int TCO = (EPI == EPI_GEGLU) ? TC / 2 : TC;     // output channels of the staged tile
    constexpr int OB = (EPI == EPI_GEGLU) ? 40 : 80;
    constexpr int ROWB = TCO * 2 + 8;
    // memory operations are batched and the optional operands resolved once, outside the fragment loops
    // (see igemm_tile.h: a per-fragment load-wait-use chain made this epilogue as long as ten k steps)
    // plain: bz = bias.  folded LayerNorm: bz = ln_t, sz = ln_s (fp32, staged behind the bias slot) and the
    // per-row (mean, rstd) of this lane's four pixel rows (see igemm_tile.h)
    constexpr int TCT = TC * CH;                               // channels of the whole tile
    float bz[5][4], sz[LN ? 5 : 1][4], mu[4], rs[4];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int ct = wc * 80 * CH + h * 80 + 16 * i + 4 * lg;
        if (LN) {
            const floatx4 tv = *reinterpret_cast<const floatx4*>(bias_lds + 1024 + TCT * 4 + ct * 4);
            const floatx4 sv = *reinterpret_cast<const floatx4*>(bias_lds + 1024 + ct * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) { bz[i][r] = tv[r]; sz[i][r] = sv[r]; }
        } else {
            const half4 bv = *reinterpret_cast<const half4*>(bias_lds + ct * 2);
#pragma unroll
            for (int r = 0; r < 4; ++r) bz[i][r] = (float)bv[r];
        }
    }
    if (LN) {          // (mean, rstd) of the tile's rows were staged in LDS before the k loop
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float2 st = *reinterpret_cast<const float2*>(bias_lds + 1024 + 8 * TCT + (wp * 64 + 16 * j + l15) * 8);
            mu[j] = st.x; rs[j] = st.y;
        }
    }
    // waits for the LDS reads of the previous pass only: no fence, so the stores of that pass stay in flight
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    ETICK(4);
    auto stage = [&](auto has_temb) __attribute__((always_inline)) {
        constexpr bool TEMB = decltype(has_temb)::value;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pr = wp * 64 + 16 * j + l15;
            const int m = p0 + pr;
            half4 tv[5];
            if (TEMB) {
                const int n = (m < p.M) ? (m / OHW) : 0;
                const f16* tp = p.temb + (size_t)n * p.temb_ld + c0out + wc * 80 * CH + h * 80 + 4 * lg;
#pragma unroll
                for (int i = 0; i < 5; ++i) tv[i] = *reinterpret_cast<const half4*>(tp + 16 * i);
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int cl = wc * 80 + 16 * i + 4 * lg;              // staged-tile channel
                float v0, v1, v2, v3;
                if (LN) {
                    v0 = rs[j] * (acc[i][j][0] - mu[j] * sz[i][0]) + bz[i][0]; v1 = rs[j] * (acc[i][j][1] - mu[j] * sz[i][1]) + bz[i][1];
                    v2 = rs[j] * (acc[i][j][2] - mu[j] * sz[i][2]) + bz[i][2]; v3 = rs[j] * (acc[i][j][3] - mu[j] * sz[i][3]) + bz[i][3];
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
    ETICK(5);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    ETICK(6);
    constexpr int CPR = TCO / 8;                               // 16-byte chunks per row
    constexpr int NIT = (TP * CPR) / NTH;
    static_assert((TP * CPR) % NTH == 0, "whole number of chunks per thread");
    constexpr int UB = (NIT % 5 == 0) ? 5 : ((NIT % 4 == 0) ? 4 : 1);
    const int c0o = (EPI == EPI_GEGLU) ? c0out / 2 : c0out;
    auto copy_out = [&](auto has_res) __attribute__((always_inline)) {
        constexpr bool RES = decltype(has_res)::value;
#pragma unroll
        for (int it0 = 0; it0 < NIT; it0 += UB) {
            half8 o[UB], rv[UB];
            int mrow[UB], gcol[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int idx = threadIdx.x + (it0 + u) * NTH;
                const int row = idx / CPR, ch = idx - row * CPR;
                mrow[u] = p0 + row;
                const int col = ch * 8;
                gcol[u] = c0o + (col / OB) * (OB * CH) + h * OB + col % OB;
                const char* src = smem + row * ROWB + ch * 16;
                const half4 lo = *reinterpret_cast<const half4*>(src);
                const half4 hi = *reinterpret_cast<const half4*>(src + 8);
                o[u] = half8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                if (RES) {
                    const int mr = mrow[u] < p.M ? mrow[u] : p.M - 1;
                    rv[u] = *reinterpret_cast<const half8*>(p.res + (size_t)mr * p.ldres + gcol[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                if (RES) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) o[u][r] = (f16)((float)o[u][r] + (float)rv[u][r]);
                }
                if (mrow[u] < p.M) *reinterpret_cast<half8*>(p.Y + (size_t)mrow[u] * p.ldy + gcol[u]) = o[u];
            }
        }
    };
    if (EPI != EPI_GEGLU && p.res) copy_out(std::true_type{}); else copy_out(std::false_type{});
    ETICK(7);
}

__device__ __attribute__((aligned(256))) unsigned char g_zero_page_big[256];

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// ---------------------------------------------------------------------------------------------------
// 256 px x 320 ch variant (WP=4 x WC=2 waves, CH=2 channel sub-tiles of 80 per wave: 160 accumulator
// VGPRs).  13.8 instead of 21.9 LDS-DMA bytes per kMAC and the activation tile is read once per 320
// output channels.  Register bound: 32-bit source offsets instead of pointers, one A-fragment set that
// is refilled fragment by fragment right after its MFMAs, epilogue in CH passes.  Used when the
// launch still has >= 2 tiles per CU; faster than 128x320 on the K >= 1280 linears, the 640-channel
// 3x3 convs and the concat convs (+8..25 %), equal or slower elsewhere.
// ---------------------------------------------------------------------------------------------------
template <int WP, int WC, int CH, int EPI, bool LN = false>
__global__ __launch_bounds__(64 * WP * WC, 2)
void igemm_big_kernel(IGemmParams p) {
    constexpr int NW = WP * WC;
    constexpr int TP = 64 * WP, TC = 80 * WC * CH;
    constexpr int WBYTES = TC * 128, XBYTES = TP * 128, STAGE = WBYTES + XBYTES;
    constexpr int WG = TC / 8, XG = TP / 8;
    constexpr int WI = WG / NW, XI = XG / NW;               // exact for every instantiated shape
    static_assert(WG % NW == 0 && XG % NW == 0, "uniform LDS-DMA count per wave required");
    constexpr int NL = WI + XI;
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef DM_IGEMM_TIMING
    long long dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = (long long)__builtin_readcyclecounter();
#define ITICK(i) do { const long long _n = (long long)__builtin_readcyclecounter(); dbg[i] += _n - tlast; tlast = _n; } while (0)
#else
#define ITICK(i) do {} while (0)
#endif

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
    const int pt = v / tiles_c;
    const int ct = v - pt * tiles_c;
    const int p0 = pt * TP;
    const int c0out = ct * TC;

    const int C1 = p.C1;
    const int C2 = p.Cin - C1;
    const int ntaps = (p.mode == IG_DENSE) ? 1 : 9;
    const int cpt = p.Cin / BK;
    const int nk = ntaps * cpt;
    const int Ktot = ntaps * p.Cin;
    const int OHW = p.OH * p.OW;

    const int lrow = lane >> 3;
    const int lchunk = ((lane & 7) ^ lrow) * 8;

    // ---- LDS-DMA source bookkeeping.
    // Per activation row of this lane: (oh << 16 | ow) or -1, and the image base pixel n*H*W.
    int xohw[XI], xnb[XI];
#pragma unroll
    for (int k = 0; k < XI; ++k) {
        const int m = p0 + (wid + k * NW) * 8 + lrow;
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
    const float sh = (float)p.H / (float)p.OH;
    const float sw = (float)p.W / (float)p.OW;
    const f16* zero = reinterpret_cast<const f16*>(g_zero_page_big) + lchunk;
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
    // SLIM (the register-bound 256x320 shape): 32-bit element offsets, addresses formed at issue.
    // otherwise: ready-made 64-bit pointers (cheaper per LDS-DMA, ~30 more VGPRs).
    constexpr bool SLIM = (CH > 1);
    unsigned woff = (unsigned)((size_t)(c0out + wid * 8 + lrow) * Ktot + lchunk);
    const unsigned wstride = (unsigned)(NW * 8) * (unsigned)Ktot;
    int xoff[XI];
    const f16* xbase = p.X;
    const f16* wsrc[SLIM ? 1 : WI];
    const f16* xsrc[SLIM ? 1 : XI];
    int xinc[SLIM ? 1 : XI];
    if (!SLIM) {
#pragma unroll
        for (int k = 0; k < WI; ++k) wsrc[k] = p.Wp + (size_t)(c0out + (wid + k * NW) * 8 + lrow) * Ktot + lchunk;
    }
    auto set_src = [&](int tap, const f16* base, int cs) __attribute__((always_inline)) {
        const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
        for (int k = 0; k < XI; ++k) {
            const int pix = src_pixel(k, dy, dx);
            if (SLIM) xoff[k] = (pix >= 0) ? (pix * cs + lchunk) : -1;
            else {
                xsrc[k] = (pix >= 0) ? (base + (size_t)pix * cs + lchunk) : zero;
                xinc[k] = (pix >= 0) ? BK : 0;
            }
        }
    };
    int ld_tap = 0, ld_cc = 0;
    auto prepare = [&]() __attribute__((always_inline)) {    // sources of the next tile to load
        if (ld_cc == 0) { xbase = p.X; set_src(ld_tap, p.X, C1); }
        else if (ld_cc * BK == C1) { xbase = p.X2; set_src(ld_tap, p.X2, C2); }
        if (++ld_cc == cpt) { ld_cc = 0; ++ld_tap; }
    };
    auto load_piece = [&](int buf, int idx) __attribute__((always_inline)) {   // idx in [0, NL): W pieces, then X
        char* wt = smem + buf * STAGE;
        if (idx < WI) {
            lptr_t dst = (lptr_t)(wt + (wid + idx * NW) * 1024);
            if (SLIM) {
                __builtin_amdgcn_global_load_lds((gptr_t)(p.Wp + (size_t)(woff + (unsigned)idx * wstride)), dst, 16, 0, 0);
                if (idx == WI - 1) woff += BK;
            } else {
                __builtin_amdgcn_global_load_lds((gptr_t)wsrc[idx], dst, 16, 0, 0);
                wsrc[idx] += BK;
            }
        } else {
            const int k = idx - WI;
            lptr_t dst = (lptr_t)(wt + WBYTES + (wid + k * NW) * 1024);
            if (SLIM) {
                const f16* a = (xoff[k] >= 0) ? (xbase + (size_t)(unsigned)xoff[k]) : zero;
                __builtin_amdgcn_global_load_lds((gptr_t)a, dst, 16, 0, 0);
                if (xoff[k] >= 0) xoff[k] += BK;
            } else {
                __builtin_amdgcn_global_load_lds((gptr_t)xsrc[k], dst, 16, 0, 0);
                xsrc[k] += xinc[k];
            }
        }
    };

    floatx4 acc[CH][5][4];
#pragma unroll
    for (int h = 0; h < CH; ++h)
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[h][i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int l15 = lane & 15;
    const int lg = lane >> 4;
    const int a_row_off = (wc * 80 * CH + l15) * 128;
    const int b_row_off = (wp * 64 + l15) * 128;
    const int koff0 = ((lg ^ (l15 & 7)) << 4), koff1 = (((4 + lg) ^ (l15 & 7)) << 4);

    // bias of this tile's TC channels -> LDS (read back in the epilogue; visible after the first barrier)
    if (tid < TC / 4) {
        half4 bv = half4{0, 0, 0, 0};
        if (p.bias) bv = *reinterpret_cast<const half4*>(p.bias + c0out + tid * 4);
        *reinterpret_cast<half4*>(smem + 2 * STAGE + tid * 8) = bv;
        if (LN) {      // ln_s, ln_t of the tile's channels (fp32) behind the 1 KB bias slot
            *reinterpret_cast<floatx4*>(smem + 2 * STAGE + 1024 + tid * 16) = *reinterpret_cast<const floatx4*>(p.ln_s + c0out + tid * 4);
            *reinterpret_cast<floatx4*>(smem + 2 * STAGE + 1024 + TC * 4 + tid * 16) = *reinterpret_cast<const floatx4*>(p.ln_t + c0out + tid * 4);
        }
    }
    if (LN && tid < TP) {      // per-row (mean, rstd) of the tile's rows behind them
        int m = p0 + tid;
        m = m < p.M ? m : p.M - 1;
        *reinterpret_cast<float2*>(smem + 2 * STAGE + 1024 + 8 * TC + tid * 8) = *reinterpret_cast<const float2*>(p.ln_stats + 2 * (size_t)m);
    }
    prepare();
#pragma unroll
    for (int i = 0; i < NL; ++i) load_piece(0, i);

    auto step = [&](int cur, bool more) __attribute__((always_inline)) {
        const char* wt = smem + cur * STAGE;
        const char* xt = wt + WBYTES;

        // 2*CH quarter-steps (s, h) of 5 groups x 4 MFMAs; the B fragments of s are shared by its CH
        // quarters; one LDS-DMA piece of the next tile after every other group until all NL are out
        half8 b0[4], b1[4], a[5];
#pragma unroll
        for (int j = 0; j < 4; ++j) b0[j] = *reinterpret_cast<const half8*>(xt + b_row_off + j * 2048 + koff0);
#pragma unroll
        for (int i = 0; i < 5; ++i) a[i] = *reinterpret_cast<const half8*>(wt + a_row_off + i * 2048 + koff0);
        if (more) prepare();
        int piece = 0;
#pragma unroll
        for (int q = 0; q < 2 * CH; ++q) {
            const int sidx = q / CH, h = q % CH;
            if (q == CH - 1 || (CH == 1 && q == 0)) {
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
                if (more && (CH == 1 || (g & 1) == 0) && piece < NL) { load_piece(cur ^ 1, piece); ++piece; }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    ITICK(0);                                   // setup + first DMA issue
    for (int kt = 0; kt < nk - 1; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        ITICK(1);                               // waits at the top of a k step
        step(kt & 1, true);
        ITICK(2);                               // k step body
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ITICK(1);
    step((nk - 1) & 1, false);
    ITICK(2);
#pragma unroll
    for (int h = 0; h < CH; ++h)
        epilogue_lds<EPI, 64 * NW, TP, 80 * WC, CH, LN>(p, acc[h], smem, p0, c0out, wc, wp, l15, lg, OHW, h, smem + 2 * STAGE
#ifdef DM_IGEMM_TIMING
                                                    , dbg, tlast
#endif
                                                    );
#ifdef DM_IGEMM_TIMING
    ITICK(3);                                   // epilogue
    if (blockIdx.x == gridDim.x / 2 + 8 && (threadIdx.x & 63) == 0 && wid < 2)
        for (int i = 0; i < 8; ++i) g_igemm_dbg[wid * 8 + i] = dbg[i];
#endif
}


}  // namespace

template <bool LN>
static hipError_t launch_igemm_big_t(const IGemmParams& p, hipStream_t s) {
    constexpr int TP = 256, TC = 320;
    constexpr size_t lds = 2 * (size_t)(TP + TC) * 128 + 1024 + (LN ? 8 * TC + 8 * TP : 0);     // + bias (+ ln_s, ln_t, row stats)
    dim3 grid(((p.M + TP - 1) / TP) * (p.Cout / TC)), block(512);
    static std::atomic<uint64_t> attr_seen{0};      // hipFuncSetAttribute is per DEVICE, not per process
    if (first_use_on_device(attr_seen)) {
        (void)hipFuncSetAttribute((const void*)igemm_big_kernel<4, 2, 2, EPI_PLAIN, LN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)igemm_big_kernel<4, 2, 2, EPI_GEGLU, LN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (p.epi == EPI_GEGLU) hipLaunchKernelGGL((igemm_big_kernel<4, 2, 2, EPI_GEGLU, LN>), grid, block, lds, s, p);
    else hipLaunchKernelGGL((igemm_big_kernel<4, 2, 2, EPI_PLAIN, LN>), grid, block, lds, s, p);
    return hipGetLastError();
}

}  // namespace dm
