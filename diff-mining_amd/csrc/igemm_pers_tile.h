// igemm_pers_tile.h — persistent form of the 256 px x 320 ch implicit-GEMM tile (formulation, operand layout and
// swizzle: igemm.hip).  One block per CU walks a strided list of tiles; the k loop is continuous across tiles:
//
//   * while the last k step of tile i runs, the LDS-DMA of tile i+1's first k step goes into the other stage (the
//     stream of "compute stage s, fetch the next k tile into stage s^1" simply continues into the next tile): no
//     prologue bubble per tile (≈ 4-5 k cycles of 20-50 k for the K = 320..1280 linears), and one loop body;
//   * the epilogue writes straight from the accumulators.  A lane of the MFMA C layout holds 4 consecutive
//     channels of one pixel per 16x16 block (8 B after fp16 packing); v_permlane16_swap + v_permlane32_swap
//     transpose the four lane groups against four channel blocks, after which a lane holds 16 consecutive
//     channels of its pixel = two 16-byte stores (the store tail is issue-bound per instruction, not per byte).
//     No staging pass, no epilogue barriers; the residual is read in the same layout, one unit ahead of the
//     stores (vmcnt retires in order: a load queued behind stores would wait for them);
//   * the stores are never waited for inside the epilogue: the next tile's first k step was fetched BEFORE them, so
//     `s_waitcnt vmcnt(<stores per wave>)` at the top of the next tile proves the operands landed while the stores
//     drain under that k step.  For the count to be exact every wave issues every store: rows beyond M are
//     redirected to a sink page instead of being predicated off;
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
typedef _Float16 ln_half2 __attribute__((ext_vector_type(2)));
typedef unsigned uintx2 __attribute__((ext_vector_type(2)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK = 64;
#ifndef DM_TILE_SWZ
#define DM_TILE_SWZ 8
#endif
constexpr int TILE_SWZ_G = DM_TILE_SWZ;       // pixel tiles per group of the wide layers' tile order (0 = channel-minor everywhere)

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

__device__ __attribute__((aligned(256))) unsigned char g_zero_page_pers[1024];
__device__ __attribute__((aligned(256))) unsigned char g_store_sink[8 * 64 * 16];      // one 16-byte slot per (wave, lane)
// Dynamic tile hand-out: one counter per XCD (its blocks share an L2, so an XCD keeps its contiguous tile range) +
// one completion counter; the last block to finish resets them, so a launch finds and leaves them at zero.  Static
// striding lost 3-5 % to the slowest CU.  The counters belong to the caller (IGemmParams::tile_ctr: an engine passes its
// own buffer, and an engine's launches are ordered on one stream, so two kernels never share live counters).  Without
// one (operator-level entry points) a launch takes the next of CSETS process-wide sets per device, round robin at enqueue
// time: that only separates launches that overlap if the device's launches are serialised on one stream — which is what
// those entry points document.
constexpr int CSETS = 64;
constexpr int CTR_DONE = 8 * 32;                       // index of the completion counter
__device__ int g_tile_ctr[CSETS][IGEMM_TILE_CTR_INTS];  // [set][xcd * 32] (128 bytes apart), [set][CTR_DONE]
#ifdef DM_IGEMM_TIMING
// phase timers of one block (tools/igemm_timing.py): [0] k-step bodies, [1] waits at the top of k steps, [2] epilogue,
// [3] tile switch (zeroing, first wait), [4] tiles, in shader cycles of wave 0
__device__ long long g_pers_dbg[8];
#define PTICK(i) do { const long long _n = (long long)__builtin_readcyclecounter(); dbg[i] += _n - tlast; tlast = _n; } while (0)
#else
#define PTICK(i) do {} while (0)
#endif

// device address of this translation unit's zero page (cached per device; a failed lookup is NOT cached and is reported by the
// launchers as its hipError_t: with a null zero page the halo / out-of-range LDS-DMA lanes would read from address 0, a GPU fault
// instead of an error return — ADVICE r04)
static inline hipError_t zero_page_addr(const void** out) {
    static std::atomic<const void*> cache[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    const void* z = cache[dev & 63].load(std::memory_order_relaxed);
    if (!z) {
        void* a = nullptr;
        const hipError_t r = hipGetSymbolAddress(&a, HIP_SYMBOL(g_zero_page_pers));
        if (r != hipSuccess || !a) { *out = nullptr; return r != hipSuccess ? r : hipErrorInvalidSymbol; }
        z = a;
        cache[dev & 63].store(z, std::memory_order_relaxed);
    }
    *out = z;
    return hipSuccess;
}
// IGemmParams::zero_page of this translation unit's zero page; the launchers call zero_page_addr() first and fail with its error
static inline IGemmParams with_zero_page(const IGemmParams& p) {
    IGemmParams q = p;
    (void)zero_page_addr(&q.zero_page);
    return q;
}
#define DM_REQUIRE_ZERO_PAGE() do { const void* _z; const hipError_t _r = zero_page_addr(&_z); if (_r != hipSuccess) return _r; } while (0)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Lane id straight from the hardware.  `asm volatile` on purpose: the k loop runs at 256 registers, and any per-lane
// constant derived once at kernel entry is either kept live across it or spilled — and a spill reload waits on
// vmcnt, i.e. behind the LDS-DMA in flight and the previous tile's stores.  Re-deriving the handful of lane
// constants where they are used costs a few VALU instructions per k step instead.
__device__ __forceinline__ int hw_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

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

// EXTRA: 0 = bias only, 1 = + time-embedding row (every tile inside one sample: OHW % 256 == 0), 2 = + residual.
// Compile-time, because runtime-optional loads in the epilogue make the compiler place their `s_waitcnt vmcnt` on the
// common path, where they wait for the prefetched LDS-DMA and the previous stores instead.
enum { PX_NONE = 0, PX_TEMB = 1, PX_RES = 2 };
// EPI value of the split-K form (small-M 3x3 convolutions): the tile stream walks (tile, k part) units, a part is a
// whole number of taps, and the epilogue stores the fp32 accumulators to partial[part][M][Cout]; the reduction kernel of
// igemm_splitk.hip then applies the fused epilogue's arithmetic.
constexpr int EPI_PARTIAL = 2;

// 8 waves (2 per SIMD), wave tile 64 px x 160 ch in two 80-channel halves, <= 256 registers.  (A 16-wave form of the
// same block tile — 64 x 80 wave tiles, <= 128 registers — was measured 1-5 % slower in r02: DESIGN.md §4b.)
// WS (r03): per-SAMPLE weights and bias row — GroupNorm folded into a 1x1 convolution (IGemmParams::w_sample_stride): a tile
// lies inside one sample n (rows_per_sample % 256 == 0), reads its weight rows at Wp + n * w_sample_stride and its fp32 bias
// row at ln_t + n * Cout, and the epilogue is the folded-LayerNorm one with (mean, rstd) = (0, 1): y = fp16(acc + t).  A
// compile-time variant in a translation unit of its own (igemm_pers_ws.hip): the other instantiations do not change by a byte.
// SC (r03): a second GEMM on another tensor folded into the k loop — after its own taps on X the loop runs Csc / 64 more steps of a
// 1x1 convolution on cat([X3, X4]) (IGemmParams::X3).  Two users: a ResNet block's `conv_shortcut` inside its conv2 (the shortcut
// tensor is never written or read back as a residual, its MACs run at the convolution's rate, the sum is rounded once), and the
// transformer blocks' `ff.net.2` + residual + `proj_out` chain as one GEMM (dense mode: (Wp W2) ff + Wp t2 + x).  Template variant in a
// translation unit of its own (igemm_pers_sc.hip).
// UP4 (r04): Upsample2D (nearest 2x) + its 3x3 convolution as FOUR 2x2 convolutions on the low-resolution source, one per output
// parity class (py, px): output pixel (2 y + py, 2 x + px) sees source rows y - 1 + py + {0, 1} and columns x - 1 + px + {0, 1}, and
// the taps of the 3x3 kernel that land on one source pixel are pre-summed (engine.hip fold_upconv_weights): 4 instead of 9 k taps.
// mode IG_CONV2_UP4, H x W = OH x OW = the SOURCE grid, M = N H W rows PER PHASE; the tile stream walks phase-major
// (4 x tiles-per-phase tiles), weights Wp [4][Cout][4 Cin], the epilogue scatters a row to its pixel of the [N][2H][2W] output.
// Template variant in a translation unit of its own (igemm_pers_up.hip).
template <int EPI, bool LN, int EXTRA, bool WS = false, bool SC = false, bool UP4 = false>
__global__ __launch_bounds__(512, 2)
void igemm_pers_kernel(IGemmParams p, int ntiles, int cset) {
    constexpr int WC = 2, CH = 2, NW = 8;
    constexpr int TP = 256, TC = 320;
    constexpr int WBYTES = TC * 128, XBYTES = TP * 128, STAGE = WBYTES + XBYTES;
    constexpr int WI = TC / 8 / NW, XI = TP / 8 / NW;          // 5 + 4 LDS-DMA pieces (8 rows x 128 B) per wave per k step
    constexpr int NL = WI + XI;
    // stores per wave per tile (every wave issues all of them: rows beyond M go to the sink page)
    constexpr bool PART = (EPI == EPI_PARTIAL);
    constexpr int NSTORE = PART ? 40 : (EPI == EPI_GEGLU) ? 16 : (EXTRA == PX_TEMB) ? 34 : 24;      // time-embedding launches: + 10 GroupNorm block-sum stores
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const aux0 = smem + 2 * STAGE;

    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wc = wid % WC;
    const int wp = wid / WC;

    // ---- this block's tiles: XCD x (= block % 8) owns a contiguous range; its blocks stride through it ----------
    const int tiles_c = p.Cout / TC;
    // the first tile of a block is static (tstart + its index on the XCD); further ones come from the XCD's counter
    int tile, tend, tdyn;
    const int xcd = blockIdx.x & 7;
    {
        const int nblk = gridDim.x;
        const int loc = blockIdx.x >> 3;
        const int q = ntiles >> 3, r = ntiles & 7;
        const int tstart = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        tend = tstart + q + (xcd < r ? 1 : 0);
        tdyn = tstart + ((nblk - xcd + 7) >> 3);       // first dynamically handed-out tile of this XCD
        tile = tstart + loc;
    }
    int* const ctr = p.tile_ctr ? p.tile_ctr : &g_tile_ctr[cset][0];
    auto finish = [&]() __attribute__((always_inline)) {
        if (threadIdx.x == 0) {
            if (atomicAdd(&ctr[CTR_DONE], 1) == (int)gridDim.x - 1) {
#pragma unroll
                for (int x = 0; x < 8; ++x) ctr[x * 32] = 0;
                ctr[CTR_DONE] = 0;
                __threadfence();
            }
        }
    };
    if (tile >= tend) { finish(); return; }

    const int C1 = p.C1;
    const int C2 = p.Cin - C1;
    const int ntaps = UP4 ? 4 : (p.mode == IG_DENSE) ? 1 : 9;
    const int cpt = p.Cin / BK;
    const int tpp = UP4 ? ntiles >> 2 : 0;                      // UP4: tiles per output parity class (the launch has 4 x tpp)
    const int KSP = PART ? p.ksplit : 1;                         // k parts (each ntaps / KSP taps); `ntiles` counts units = tiles * KSP
    const int cpt_sc = SC ? p.Csc / BK : 0;                      // SC: k steps of the folded shortcut ("tap 9")
    const int nk = ntaps * cpt / KSP + cpt_sc;                   // k steps of one unit, >= 4 (igemm_pers_ok)
    const int Ktot = ntaps * p.Cin + (SC ? p.Csc : 0);
    const int OHW = p.OH * p.OW;
    constexpr bool temb_lds = (EXTRA == PX_TEMB);                // a tile lies inside one sample: its temb row goes through LDS


    // ---- load-side state of the tile whose operands are being fetched ---------------------------------------------
    // (kept minimal: the k loop runs at 256 registers; the pixel coordinates of a lane's rows are re-derived from the
    // row index at every tap change instead of being held)
    int lp0 = 0, lc0 = 0;
    int xpk[XI], xoff[XI];            // xpk: sample << 18 | oh << 9 | ow of this lane's activation rows, -1 beyond M (n < 8192, oh / ow < 512)
    unsigned woff = 0;
    const unsigned wstride = (unsigned)(NW * 8) * (unsigned)Ktot;
    const f16* xbase = p.X;
    int ld_tap = 0, ld_cc = 0;
    int ld_ph = 0;                    // UP4: output parity class (py << 1 | px) of the tile being fetched

    // tile index -> (pixel tile, channel tile).  Channel-minor order makes the 32 CUs of an XCD, which work on consecutive tiles, share
    // ONE pixel tile and pull min(32, tiles_c) weight tiles between them — for the wide layers (GEGLU projections: tiles_c = 8 / 16 / 32,
    // q/k/v: 3 / 6 / 12) that re-streams the weight matrix once per round through a 4 MB L2 (profiles/r05_final_pmc_shapes.txt: 3.5 / 10.8 /
    // 33 x the algorithmic bytes).  r05: the LayerNorm-folded instantiations (GEGLU, q/k/v, to_q) walk groups of TILE_SWZ_G = 8 pixel
    // tiles x tiles_c channel tiles, pixel-minor inside a group: 32 consecutive tiles share 8 pixel tiles and 4 weight tiles (8.5 instead of
    // 26.9 MB per round at N = 10 240).  Same-box round-robin of G = 4 / 8 / 16 / 32 (profiles/r05_ab_tile_swz.txt): 8 is best, 640 -> 5120
    // -3.5 %, 1280 -> 10240 -1 %, a step -0.6 ms.  For tiles_c <= 4 the 32 tiles of a round are the same set as before.  The order of the
    // tiles never changes a tile's arithmetic (same checksum).  A channel-group-major order (4 weight tiles x ALL pixel tiles, hoping the weights
    // stay in the L2) fetched exactly the same bytes — the activations and outputs streaming through a 4 MB L2 evict 1.6 - 3.3 MB of weights
    // every round anyway — and measured -0.1 ms: not kept (profiles/r05_ab_tile_swz.txt).
    auto decode_tile = [&](int tl, int& pt, int& ct) __attribute__((always_inline)) {
        if constexpr ((EPI == EPI_GEGLU || LN) && !WS && !PART && !UP4 && TILE_SWZ_G > 0) {
            constexpr int G = TILE_SWZ_G;
            const int gsz = G * tiles_c;
            const int grp = tl / gsz, r = tl - grp * gsz;
            const int tiles_p = (p.M + TP - 1) / TP;
            const int left = tiles_p - grp * G;
            const int gp = left < G ? left : G;          // ragged last group
            ct = r / gp;
            pt = grp * G + (r - ct * gp);
            return;
        }
        pt = tl / tiles_c;
        ct = tl - pt * tiles_c;
    };
    auto pack_row = [&](int m) __attribute__((always_inline)) -> int {
        if (m >= p.M) return -1;
        if (p.mode == IG_DENSE) return m;
        const int n = m / OHW;
        const int rem = m - n * OHW;
        const int oh = rem / p.OW;
        return (n << 18) | (oh << 9) | (rem - oh * p.OW);
    };
    auto set_tile = [&](int unit) __attribute__((always_inline)) {
        int tl = PART ? unit / KSP : unit;
        const int tap0 = PART ? (unit - tl * KSP) * (ntaps / KSP) : 0;
        if constexpr (UP4) { ld_ph = tl / tpp; tl -= ld_ph * tpp; }
        int pt, ct;
        decode_tile(tl, pt, ct);
        lp0 = pt * TP;
        lc0 = ct * TC;
        const int ln = hw_lane();
        const int lrow = ln >> 3, lchunk = ((ln & 7) ^ lrow) * 8;
        woff = (unsigned)((size_t)(lc0 + wid * 8 + lrow) * Ktot + lchunk) + (unsigned)(tap0 * p.Cin);
        if constexpr (WS) woff += (unsigned)((long long)(lp0 / p.rows_per_sample) * p.w_sample_stride);
        if constexpr (UP4) woff += (unsigned)ld_ph * (unsigned)p.Cout * (unsigned)Ktot;
        ld_tap = tap0; ld_cc = 0;
#pragma unroll
        for (int k = 0; k < XI; ++k) xpk[k] = pack_row(lp0 + (wid + k * NW) * 8 + lrow);
    };
    auto src_pixel = [&](int k, int dy, int dx) __attribute__((always_inline)) -> int {
        const int pk = xpk[k];
        if (pk < 0) return -1;
        if (p.mode == IG_DENSE) return pk;
        const int oh = (pk >> 9) & 511, ow = pk & 511;
        const int xnb = (pk >> 18) * (p.H * p.W);
        if constexpr (UP4) {                                       // (dy, dx) = the 2x2 tap (a, b); rows are pixels of the source grid
            const int ih = oh + dy - 1 + (ld_ph >> 1), iw = ow + dx - 1 + (ld_ph & 1);
            return (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) ? xnb + ih * p.W + iw : -1;
        }
        if (p.mode == IG_CONV3 || p.mode == IG_CONV3_S2) {
            const int st = (p.mode == IG_CONV3_S2) ? 2 : 1;
            const int ih = oh * st + dy - 1, iw = ow * st + dx - 1;
            return (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) ? xnb + ih * p.W + iw : -1;
        }
        const int uh = oh + dy - 1, uw = ow + dx - 1;              // conv on the nearest-upsampled image
        if (uh < 0 || uh >= p.OH || uw < 0 || uw >= p.OW) return -1;
        int ih, iw;
        if (p.OH == 2 * p.H && p.OW == 2 * p.W) { ih = uh >> 1; iw = uw >> 1; }      // exact 2x: floor(u * 0.5), no float math
        else {                                                                      // F.interpolate(size=...), odd latent sizes
            const float sh = (float)p.H / (float)p.OH, sw = (float)p.W / (float)p.OW;
            ih = (int)floorf((float)uh * sh); ih = ih < p.H - 1 ? ih : p.H - 1;
            iw = (int)floorf((float)uw * sw); iw = iw < p.W - 1 ? iw : p.W - 1;
        }
        return xnb + ih * p.W + iw;
    };
    auto set_src = [&](int tap, int cs) __attribute__((always_inline)) {
        const int dy = UP4 ? tap >> 1 : tap / 3, dx = UP4 ? tap & 1 : tap - dy * 3;
        const int ln = hw_lane();
        const int lrow = ln >> 3, lchunk = ((ln & 7) ^ lrow) * 8;
#pragma unroll
        for (int k = 0; k < XI; ++k) {
            const int pix = src_pixel(k, dy, dx);
            xoff[k] = (pix >= 0) ? (pix * cs + lchunk) : -1;
        }
    };
    auto prepare = [&]() __attribute__((always_inline)) {    // sources of the next k tile to load
        if constexpr (SC) {
            if (ld_tap == ntaps) {                           // the folded second GEMM: centre tap (dense: the row itself) on cat([X3, X4])
                const int ctap = (p.mode == IG_DENSE) ? 0 : 4;
                if (ld_cc == 0) { xbase = p.X3; set_src(ctap, p.C3); }
                else if (ld_cc * BK == p.C3) { xbase = p.X4; set_src(ctap, p.Csc - p.C3); }
                if (++ld_cc == cpt_sc) { ld_cc = 0; ++ld_tap; }
                return;
            }
        }
        if (ld_cc == 0) { xbase = p.X; set_src(ld_tap, C1); }
        else if (ld_cc * BK == C1) { xbase = p.X2; set_src(ld_tap, C2); }
        if (++ld_cc == cpt) { ld_cc = 0; ++ld_tap; }
    };
    auto load_piece = [&](int buf, int idx, int lchunk) __attribute__((always_inline)) {   // idx in [0, NL): W pieces, then X
        char* wt = smem + buf * STAGE;
        if (idx < WI) {
            lptr_t dst = (lptr_t)(wt + (wid + idx * NW) * 1024);
            __builtin_amdgcn_global_load_lds((gptr_t)(p.Wp + (size_t)(woff + (unsigned)idx * wstride)), dst, 16, 0, 0);
            if (idx == WI - 1) woff += BK;
        } else {
            const int k = idx - WI;
            lptr_t dst = (lptr_t)(wt + WBYTES + (wid + k * NW) * 1024);
            const f16* a = (xoff[k] >= 0) ? (xbase + (size_t)(unsigned)xoff[k]) : reinterpret_cast<const f16*>(p.zero_page) + lchunk;
            __builtin_amdgcn_global_load_lds((gptr_t)a, dst, 16, 0, 0);
            if (xoff[k] >= 0) xoff[k] += BK;
        }
    };
    // per-tile vectors -> LDS slot by LDS-DMA (lane-linear 1 KiB pieces; lanes past the vector read the zero page)
    auto load_aux = [&](int slot) __attribute__((always_inline)) {
        char* ax = aux0 + slot * AUX_BYTES;
        const int lane = hw_lane();
        const char* zp = reinterpret_cast<const char*>(p.zero_page) + lane * 16;
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
                if constexpr (WS) v = p.ln_t + (size_t)(lp0 / p.rows_per_sample) * p.Cout + lc0;     // the sample's bias row (also as the unused s)
                const int f = half * 256 + lane * 4;
                s = (f < TC) ? reinterpret_cast<const char*>(v + f) : zp;
            } else {
                int row = lp0 + half * 128 + lane * 2;               // two (mean, rstd) pairs per lane
                row = row < p.M - 1 ? row : p.M - 2;
                s = (p.ln_stats && !WS) ? reinterpret_cast<const char*>(p.ln_stats + 2 * (size_t)row) : zp;     // in-kernel statistics: the slot is written after the k loop
            }
            __builtin_amdgcn_global_load_lds((gptr_t)s, (lptr_t)(ax + AUX_LNS + (wid - 2) * 1024), 16, 0, 0);
        }
    };

    floatx4 acc[CH][5][4];
    // folded LayerNorm without a statistics kernel (p.ln_stats == nullptr): K = C, so a tile's k steps carry every channel of
    // its 256 rows through LDS once.  Lane pair (2r, 2r + 1) of the block owns row r: per k step each lane adds its 32
    // channels (logical chunks 4 half .. 4 half + 3 of the 64-channel slab, ascending) to (sum, sum of squares) in fp32 —
    // VALU work in the shadow of the MFMAs — and after the k loop the pair is combined into (mean, rstd) in the tile's
    // vector slot.  The order depends on the row only, never on the tile geometry or the batch (igemm_tile.h adds the same
    // numbers in the same order).
    const bool ln_ink = LN && !WS && p.ln_stats == nullptr;
    float ln_s1 = 0.f, ln_s2 = 0.f;

    // one k step on stage `cur`, with the LDS-DMA of the following k tile of the stream into the other stage interleaved
    // (its sources were prepared after the previous step's MFMAs, when no fragment registers are live: the address
    // arithmetic of a tap change needs ~20 temporaries)
    auto step = [&](int cur) __attribute__((always_inline)) {
        const char* wt = smem + cur * STAGE;
        const char* xt = wt + WBYTES;
        const int ln = hw_lane();
        const int l15 = ln & 15, lg = ln >> 4;
        const int a_row_off = (wc * 80 * CH + l15) * 128;
        const int b_row_off = (wp * 64 + l15) * 128;
        const int koff0 = ((lg ^ (l15 & 7)) << 4), koff1 = (((4 + lg) ^ (l15 & 7)) << 4);
        const int lchunk = ((ln & 7) ^ (ln >> 3)) * 8;
        half8 b0[4], b1[4], a[5];
#pragma unroll
        for (int j = 0; j < 4; ++j) b0[j] = *reinterpret_cast<const half8*>(xt + b_row_off + j * 2048 + koff0);
#pragma unroll
        for (int i = 0; i < 5; ++i) a[i] = *reinterpret_cast<const half8*>(wt + a_row_off + i * 2048 + koff0);
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
                for (int jj = 0; jj < 4; ++jj) {
                    // snake order over the pixel fragments (DM_MFMA_SNAKE): between consecutive MFMAs exactly ONE operand changes — at a
                    // group change the weight fragment, inside a group the pixel fragment — instead of both at every group change.  Every
                    // accumulator gets the same products in the same k order (bit-identical); the matrix cores are power-bound on dense
                    // fp16 streams and operand toggling is a visible share of it (tools/probes/probe_order.hip: +1.7 % on the MFMA stream alone)
                    const int j = (DM_MFMA_SNAKE && ((q * 5 + i) & 1)) ? 3 - jj : jj;
                    acc[h][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], sidx ? b1[j] : b0[j], acc[h][i][j], 0, 0, 0);
                }
                if (q + 1 < 2 * CH) {          // fragment i of the next quarter replaces the one just consumed
                    const int nq = q + 1, ns = nq / CH, nh = nq % CH;
                    a[i] = *reinterpret_cast<const half8*>(wt + a_row_off + nh * (80 * 128) + i * 2048 + (ns ? koff1 : koff0));
                }
                // two LDS-DMA pieces per group: all nine out within the first quarter of the step (measured against one
                // piece per group / per other group: best by 0-7 % per shape; the later the last piece, the longer the wait)
                if (piece < NL) { load_piece(cur ^ 1, piece, lchunk); ++piece; }
                if (piece < NL) { load_piece(cur ^ 1, piece, lchunk); ++piece; }
                if (LN && q >= 1 && i < 2 && q < 3) {       // four 16-byte chunks of this lane's row half, one per MFMA group, after the DMA issue
                    if (ln_ink) {
                        const int c4 = (q - 1) * 2 + i;
                        const int row = (wid * 64 + ln) >> 1, hf = ln & 1;
                        const half8 xv = *reinterpret_cast<const half8*>(xt + row * 128 + (((4 * hf + c4) ^ (row & 7)) << 4));
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
        }
    };

    // ---- split-K epilogue: the fp32 accumulators of this k part, 16 bytes (4 channels of a pixel) per store ------------
    auto epilogue_partial = [&](int p0, int c0out, int part) __attribute__((always_inline)) {
        const int eln = hw_lane();
        const int e15 = eln & 15, eg = eln >> 4;
        float* const sink = reinterpret_cast<float*>(g_store_sink) + (wid * 64 + eln) * 4;
        float* const base = p.partial + (size_t)part * p.M * p.Cout + c0out + wc * 80 * CH + 4 * eg;
#pragma unroll
        for (int h = 0; h < CH; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = p0 + wp * 64 + 16 * j + e15;
                float* const row = (m < p.M) ? base + (size_t)m * p.Cout + h * 80 : nullptr;
#pragma unroll
                for (int i = 0; i < 5; ++i) *reinterpret_cast<floatx4*>(row ? row + 16 * i : sink) = acc[h][i][j];
            }
    };

    // ---- epilogue: straight from the accumulators ---------------------------------------------------------------
    auto epilogue = [&](int p0, int c0out, int slot, int ph) __attribute__((always_inline)) {
        const char* ax = aux0 + slot * AUX_BYTES;
        constexpr int OCH = (EPI == EPI_GEGLU) ? 40 : 80;          // output channels of one (wave, h) sub-tile
        constexpr bool RES = (EXTRA == PX_RES);
        constexpr bool GNB = (EXTRA == PX_TEMB) && EPI == EPI_PLAIN;          // GroupNorm block sums (IGemmParams::gn_blocks)
        const int c0o = (EPI == EPI_GEGLU) ? c0out / 2 : c0out;
        // the lane id is re-read from the hardware inside every epilogue: anything derived from the kernel-wide `lane`
        // is hoisted out of the tile loop by the compiler and then lives (or spills) across the k loop, which runs
        // at 256 registers; a spill reload here would also sit behind the LDS-DMA just issued (vmcnt is in-order)
        const int eln = hw_lane();
        const int e15 = eln & 15, eg = eln >> 4;
        f16* const sink = reinterpret_cast<f16*>(g_store_sink) + (wid * 64 + eln) * 8;
        // residual of unit u = (h, j) in the layout of the stores: 16 channels of block eg (two half8) + block 4's quarter
        // Residual variant, two phases: (A) every unit is converted / transposed into packed registers while ALL residual
        // loads are issued (unit u+1's right after unit u's conversion freed its 20 accumulator registers), (B) add + store.
        // Every load is older than every store (vmcnt retires in order: a load queued behind stores would wait for them),
        // and the wait for the first residual — which also sits behind the LDS-DMA of the next tile's first k step —
        // overlaps the conversion work instead of preceding it.
        constexpr int NU = 4 * CH;
        half8 rlo[RES ? NU : 1], rhi[RES ? NU : 1], plo[RES ? NU : 1], phi[RES ? NU : 1];
        half4 r4[RES ? NU : 1], p4[RES ? NU : 1];
        f16* ypu[RES ? NU : 1];
        auto load_res = [&](int u) __attribute__((always_inline)) {
            const int h = u >> 2, j = u & 3;
            int m = p0 + wp * 64 + 16 * j + e15;
            m = m < p.M ? m : p.M - 1;
            const f16* rp = p.res + (size_t)m * p.ldres + c0o + wc * (OCH * CH) + h * OCH;
            rlo[RES ? u : 0] = *reinterpret_cast<const half8*>(rp + 16 * eg);
            rhi[RES ? u : 0] = *reinterpret_cast<const half8*>(rp + 16 * eg + 8);
            r4[RES ? u : 0] = *reinterpret_cast<const half4*>(rp + 64 + 4 * eg);
        };
        if (RES) load_res(0);
#pragma unroll
        for (int h = 0; h < CH; ++h) {
            // GroupNorm block sums of the (wave, h) sub-tile's 64 rows x 80 channels (time-embedding launches = conv1, norm2's input):
            // ten channel pairs per lane, accumulated over the four row blocks j, then the fixed 16-lane tree (dm_kernels.h)
            float gs[GNB ? 10 : 1], gq[GNB ? 10 : 1];
            if constexpr (GNB) {
#pragma unroll
                for (int k = 0; k < 10; ++k) { gs[k] = 0.f; gq[k] = 0.f; }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int u = h * 4 + j;
                const int pr = wp * 64 + 16 * j + e15;
                const int m = p0 + pr;
                // folded LayerNorm: y = rstd (acc - mean s) + t, evaluated as fma(rstd, acc, fma(-rstd mean, s, t)) — two fused
                // operations per value instead of mul / sub / mul / add (r03: 4 of ~37 VALU instructions per GEGLU quad), and one
                // rounding less; the 128-row tile (igemm_tile.h) uses the same form
                float nrm = 0.f, rs = 1.f;
                if (LN && !WS) {
                    const float2 st = *reinterpret_cast<const float2*>(ax + AUX_STATS + pr * 8);
                    rs = st.y; nrm = -(st.y * st.x);
                }
                constexpr int NWD = (EPI == EPI_GEGLU) ? 1 : 2;
                unsigned R[4][NWD], R4[NWD];
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    const int ct = wc * 80 * CH + h * 80 + 16 * i + 4 * eg;          // tile-local GEMM channel
                    float v[4];
                    if (LN) {
                        const floatx4 t4 = *reinterpret_cast<const floatx4*>(ax + AUX_LNT + ct * 4);
                        const floatx4 s4 = *reinterpret_cast<const floatx4*>(ax + AUX_LNS + ct * 4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(rs, acc[h][i][j][r], __builtin_fmaf(nrm, s4[r], t4[r]));
                    } else {
                        const half4 bv = *reinterpret_cast<const half4*>(ax + AUX_BIAS + ct * 2);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = acc[h][i][j][r] + (float)bv[r];
                    }
                    unsigned w0, w1 = 0;
                    if (EPI == EPI_GEGLU) {
                        const f16 h0 = (f16)v[0], h1 = (f16)v[1], g0 = (f16)v[2], g1 = (f16)v[3];
                        const f16 q0 = (f16)gelu_erf((float)g0), q1 = (f16)gelu_erf((float)g1);
                        const half2_ o = half2_{(f16)((float)h0 * (float)q0), (f16)((float)h1 * (float)q1)};
                        w0 = __builtin_bit_cast(unsigned, o);
                    } else {
                        half4 o = half4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                        if (temb_lds) {
                            const half4 tv = *reinterpret_cast<const half4*>(ax + AUX_TEMB + ct * 2);
#pragma unroll
                            for (int r = 0; r < 4; ++r) o[r] = (f16)((float)o[r] + (float)tv[r]);
                        }
                        const uintx2 uu = __builtin_bit_cast(uintx2, o);
                        w0 = uu[0]; w1 = uu[1];
                    }
                    if (i < 4) { R[i][0] = w0; if (NWD == 2) R[i][NWD - 1] = w1; }
                    else { R4[0] = w0; if (NWD == 2) R4[NWD - 1] = w1; }
                }
                transpose4<NWD>(R);            // lane group eg now holds quarters 0..3 of channel block eg
                size_t orow = (size_t)m;
                if constexpr (UP4) {                               // source-grid row -> its pixel of parity class ph in the [N][2H][2W] output
                    const int n = m / OHW, rem = m - n * OHW;
                    const int oh = rem / p.OW, ow = rem - oh * p.OW;
                    orow = ((size_t)n * (2 * p.OH) + (2 * oh + (ph >> 1))) * (size_t)(2 * p.OW) + (2 * ow + (ph & 1));
                }
                f16* yp = (m < p.M) ? p.Y + orow * p.ldy + c0o + wc * (OCH * CH) + h * OCH : nullptr;
                if (EPI == EPI_GEGLU) {
                    const uintx4 o8 = uintx4{R[0][0], R[1][0], R[2][0], R[3][0]};           // 8 output channels of block eg
                    *reinterpret_cast<uintx4*>(yp ? yp + 8 * eg : sink) = o8;
                    *reinterpret_cast<unsigned*>(yp ? yp + 32 + 2 * eg : sink) = R4[0];
                } else {
                    half8 lo = __builtin_bit_cast(half8, uintx4{R[0][0], R[0][NWD - 1], R[1][0], R[1][NWD - 1]});
                    half8 hi = __builtin_bit_cast(half8, uintx4{R[2][0], R[2][NWD - 1], R[3][0], R[3][NWD - 1]});
                    half4 o4 = __builtin_bit_cast(half4, uintx2{R4[0], R4[NWD - 1]});
                    if (RES) {
                        plo[RES ? u : 0] = lo; phi[RES ? u : 0] = hi; p4[RES ? u : 0] = o4; ypu[RES ? u : 0] = yp;
                        if (u + 1 < NU) load_res(u + 1);
                    } else {
                        *reinterpret_cast<half8*>(yp ? yp + 16 * eg : sink) = lo;
                        *reinterpret_cast<half8*>(yp ? yp + 16 * eg + 8 : sink) = hi;
                        *reinterpret_cast<half4*>(yp ? yp + 64 + 4 * eg : sink) = o4;
                        if constexpr (GNB) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                gn_pair_acc(gs[GNB ? 2 * i : 0], gq[GNB ? 2 * i : 0], R[i][0]);
                                gn_pair_acc(gs[GNB ? 2 * i + 1 : 0], gq[GNB ? 2 * i + 1 : 0], R[i][NWD - 1]);
                            }
                            gn_pair_acc(gs[GNB ? 8 : 0], gq[GNB ? 8 : 0], R4[0]);
                            gn_pair_acc(gs[GNB ? 9 : 0], gq[GNB ? 9 : 0], R4[NWD - 1]);
                        }
                    }
                }
            }
            if constexpr (GNB) {
#pragma unroll
                for (int k = 0; k < 10; ++k) { gs[k] = gn_row16_sum(gs[k]); gq[k] = gn_row16_sum(gq[k]); }
                const int r0 = p0 + wp * 64;
                // every lane issues the five stores (the vmcnt count at the next tile's top is per instruction): lanes other than e15 = 0,
                // blocks beyond M and launches without a block buffer write their slot of the sink page
                float* const fsink = reinterpret_cast<float*>(sink);
                float* const gb = (p.gn_blocks && e15 == 0 && r0 < p.M) ? p.gn_blocks + (size_t)(r0 >> 6) * p.Cout + c0o + wc * (OCH * CH) + h * OCH : nullptr;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<floatx4*>(gb ? gb + 16 * eg + 4 * i : fsink) = floatx4{gs[GNB ? 2 * i : 0], gq[GNB ? 2 * i : 0], gs[GNB ? 2 * i + 1 : 0], gq[GNB ? 2 * i + 1 : 0]};
                *reinterpret_cast<floatx4*>(gb ? gb + 64 + 4 * eg : fsink) = floatx4{gs[GNB ? 8 : 0], gq[GNB ? 8 : 0], gs[GNB ? 9 : 0], gq[GNB ? 9 : 0]};
            }
        }
        if (RES) {
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                half8 lo = plo[RES ? u : 0], hi = phi[RES ? u : 0];
                half4 o4 = p4[RES ? u : 0];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    lo[r] = (f16)((float)lo[r] + (float)rlo[RES ? u : 0][r]);
                    hi[r] = (f16)((float)hi[r] + (float)rhi[RES ? u : 0][r]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) o4[r] = (f16)((float)o4[r] + (float)r4[RES ? u : 0][r]);
                f16* yp = ypu[RES ? u : 0];
                *reinterpret_cast<half8*>(yp ? yp + 16 * eg : sink) = lo;
                *reinterpret_cast<half8*>(yp ? yp + 16 * eg + 8 : sink) = hi;
                *reinterpret_cast<half4*>(yp ? yp + 64 + 4 * eg : sink) = o4;
            }
        }
    };

    // ---- prologue: the first tile's k step 0 into stage 0, its vectors into slot 0; sources of k step 1 prepared ----
    set_tile(tile);
    load_aux(0);
    prepare();
    {
        const int ln = hw_lane();
        const int lchunk = ((ln & 7) ^ (ln >> 3)) * 8;
#pragma unroll
        for (int i = 0; i < NL; ++i) load_piece(0, i, lchunk);
    }
    prepare();

#ifdef DM_IGEMM_TIMING
    long long dbg[5] = {0, 0, 0, 0, 0};
    long long tlast = (long long)__builtin_readcyclecounter();
#endif
    int g = 0;                        // global k step of the stream: stage = g & 1
    int slot = 0;
    bool first = true;
    while (true) {
        int rtile = PART ? tile / KSP : tile;
        int cur_ph = 0;
        if constexpr (UP4) { cur_ph = rtile / tpp; rtile -= cur_ph * tpp; }
        int pt, ct_;
        decode_tile(rtile, pt, ct_);
        const int p0 = pt * TP, c0out = ct_ * TC;
#pragma unroll
        for (int h = 0; h < CH; ++h)
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[h][i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

        // The tile's k step 0 (and its vectors) were requested BEFORE the previous epilogue's stores, so "at most
        // NSTORE outstanding" proves they have landed; every later k step waits for everything, which is where the
        // previous tile's stores must have drained (they had the epilogue's own run time plus one k step).
        if (first) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        else if (NSTORE == 40) asm volatile("s_waitcnt vmcnt(40)\n\ts_barrier" ::: "memory");
        else if (NSTORE == 34) asm volatile("s_waitcnt vmcnt(34)\n\ts_barrier" ::: "memory");
        else if (NSTORE == 24) asm volatile("s_waitcnt vmcnt(24)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
        first = false;
        PTICK(3);
        // next tile of this block: asked for now (one returning atomic by thread 0), published through LDS (a spare word
        // of the current vector slot) after k step 1's top wait, read by everyone after k step nk - 2  (nk >= 4)
        int ticket = 0;
        if (threadIdx.x == 0) ticket = atomicAdd(&ctr[xcd * 32], 1);
        int next = 0;
        bool has_next = false;
        for (int kt = 0; kt < nk; ++kt) {
            step(g & 1);                                   // computes stream step g, requests stream step g + 1
            ++g;
            PTICK(0);
            // sources of stream step g + 1 (two ahead of the one just computed): from k step nk - 2 on they belong
            // to the next tile (without one the block re-requests its own first k steps: valid addresses, unused)
            if (kt == nk - 2) {
                next = tdyn + *reinterpret_cast<volatile int*>(aux0 + slot * AUX_BYTES + 1020);
                next = __builtin_amdgcn_readfirstlane(next);
                has_next = next < tend;
                set_tile(has_next ? next : tile);
                if (has_next) load_aux(slot ^ 1);
            }
            prepare();
            if (kt < nk - 1) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            PTICK(1);
            if (kt == 0 && threadIdx.x == 0) *reinterpret_cast<volatile int*>(aux0 + slot * AUX_BYTES + 1020) = ticket;
        }
        if (LN) {
            if (ln_ink) {
                const int ln2 = hw_lane();
                const float t1 = ln_s1 + __shfl_xor(ln_s1, 1), t2 = ln_s2 + __shfl_xor(ln_s2, 1);
                const float mean = t1 / (float)p.Cin;
                float var = t2 / (float)p.Cin - mean * mean;
                var = var > 0.f ? var : 0.f;
                if (mean * mean > LN_REDO_RATIO2 * var) {            // cancellation: exact second pass over this row (dm_kernels.h)
                    int m = p0 + ((wid * 64 + ln2) >> 1);
                    m = m < p.M ? m : p.M - 1;
                    const int half_c = p.Cin >> 1;
                    const f16* xr = p.X + (size_t)m * p.Cin + (ln2 & 1) * half_c;
                    float q = 0.f;
                    for (int c = 0; c < half_c; c += 8) {
                        const half8 v = *reinterpret_cast<const half8*>(xr + c);
#pragma unroll
                        for (int k = 0; k < 8; ++k) { const float d = (float)v[k] - mean; q = __builtin_fmaf(d, d, q); }
                    }
                    var = (q + __shfl_xor(q, 1)) / (float)p.Cin;
                }
                if ((ln2 & 1) == 0)
                    *reinterpret_cast<float2*>(aux0 + slot * AUX_BYTES + AUX_STATS + ((wid * 64 + ln2) >> 1) * 8) = float2{mean, rsqrtf(var + p.ln_eps)};
                ln_s1 = 0.f; ln_s2 = 0.f;
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
        }
        if constexpr (PART) epilogue_partial(p0, c0out, tile - rtile * KSP);
        else epilogue(p0, c0out, slot, cur_ph);
        PTICK(2);
#ifdef DM_IGEMM_TIMING
        dbg[4] += 1;
#endif
        if (!has_next) break;
        tile = next;
        slot ^= 1;
    }
    finish();
#ifdef DM_IGEMM_TIMING
    if (blockIdx.x == 77 && threadIdx.x == 0)
        for (int i = 0; i < 5; ++i) g_pers_dbg[i] = dbg[i];
#endif
}

}  // namespace

template <bool LN>
static hipError_t launch_igemm_pers_t(const IGemmParams& p, hipStream_t s) {
    DM_REQUIRE_ZERO_PAGE();
    constexpr int TP = 256, TC = 320;
    constexpr size_t lds = 2 * (size_t)(TP + TC) * 128 + 2 * AUX_BYTES;           // 160 KiB: two operand stages + two vector slots
    const int ntiles = ((p.M + TP - 1) / TP) * (p.Cout / TC);
    const int n_cu = device_cu_count();
    const int grid = ntiles < n_cu ? ntiles : n_cu;
    static std::atomic<uint64_t> attr_seen{0};      // hipFuncSetAttribute is per DEVICE, not per process
    if (first_use_on_device(attr_seen)) {
        (void)hipFuncSetAttribute((const void*)igemm_pers_kernel<EPI_PLAIN, LN, PX_NONE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)igemm_pers_kernel<EPI_GEGLU, LN, PX_NONE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if constexpr (!LN) {
            (void)hipFuncSetAttribute((const void*)igemm_pers_kernel<EPI_PLAIN, false, PX_TEMB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute((const void*)igemm_pers_kernel<EPI_PLAIN, false, PX_RES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        }
    }
    static std::atomic<unsigned> launch_no{0};
    const int cset = (int)(launch_no.fetch_add(1) % CSETS);
    const dim3 g(grid), b(512);
    if (p.epi == EPI_GEGLU) { launch_timed((igemm_pers_kernel<EPI_GEGLU, LN, PX_NONE>), g, b, lds, s, with_zero_page(p), ntiles, cset); return hipGetLastError(); }
    if constexpr (!LN) {        // the folded-LayerNorm layers never carry a time embedding or a residual (igemm_pers_ok)
        if (p.temb) { launch_timed((igemm_pers_kernel<EPI_PLAIN, false, PX_TEMB>), g, b, lds, s, with_zero_page(p), ntiles, cset); return hipGetLastError(); }
        if (p.res) { launch_timed((igemm_pers_kernel<EPI_PLAIN, false, PX_RES>), g, b, lds, s, with_zero_page(p), ntiles, cset); return hipGetLastError(); }
    }
    launch_timed((igemm_pers_kernel<EPI_PLAIN, LN, PX_NONE>), g, b, lds, s, with_zero_page(p), ntiles, cset);
    return hipGetLastError();
}

// the shortcut-in-conv2 variant; only igemm_pers_sc.hip defines DM_IGEMM_PERS_SC
#ifdef DM_IGEMM_PERS_SC
static hipError_t launch_igemm_pers_sc_t(const IGemmParams& p, hipStream_t s) {
    DM_REQUIRE_ZERO_PAGE();
    constexpr int TP = 256, TC = 320;
    constexpr size_t lds = 2 * (size_t)(TP + TC) * 128 + 2 * AUX_BYTES;
    if ((p.mode != IG_CONV3 && p.mode != IG_DENSE) || p.epi != EPI_PLAIN || p.ln_s || p.temb || p.X2 || !p.X3 || p.Csc <= 0 || p.Csc % BK ||
        p.C3 % BK || (p.C3 < p.Csc && !p.X4) || p.Cout % TC != 0 || p.ksplit > 1) return hipErrorInvalidValue;
    const int ntiles = ((p.M + TP - 1) / TP) * (p.Cout / TC);
    const int n_cu = device_cu_count();
    const int grid = ntiles < n_cu ? ntiles : n_cu;
    static std::atomic<uint64_t> attr_seen{0};
    if (first_use_on_device(attr_seen)) {
        (void)hipFuncSetAttribute((const void*)igemm_pers_kernel<EPI_PLAIN, false, PX_NONE, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)igemm_pers_kernel<EPI_PLAIN, false, PX_RES, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    static std::atomic<unsigned> launch_no{0};
    const int cset = (int)(launch_no.fetch_add(1) % CSETS);
    if (p.res) launch_timed((igemm_pers_kernel<EPI_PLAIN, false, PX_RES, false, true>), dim3(grid), dim3(512), lds, s, with_zero_page(p), ntiles, cset);
    else launch_timed((igemm_pers_kernel<EPI_PLAIN, false, PX_NONE, false, true>), dim3(grid), dim3(512), lds, s, with_zero_page(p), ntiles, cset);
    return hipGetLastError();
}
#endif

// the per-sample-weights variant (GroupNorm folded into proj_in); only igemm_pers_ws.hip defines DM_IGEMM_PERS_WS
#ifdef DM_IGEMM_PERS_WS
static hipError_t launch_igemm_pers_ws_t(const IGemmParams& p, hipStream_t s) {
    DM_REQUIRE_ZERO_PAGE();
    constexpr int TP = 256, TC = 320;
    constexpr size_t lds = 2 * (size_t)(TP + TC) * 128 + 2 * AUX_BYTES;
    if (p.mode != IG_DENSE || p.epi != EPI_PLAIN || !p.ln_t || p.w_sample_stride <= 0 || p.rows_per_sample <= 0 ||
        p.rows_per_sample % TP != 0 || p.M % p.rows_per_sample != 0 || p.Cout % TC != 0 || p.res || p.temb) return hipErrorInvalidValue;
    const int ntiles = (p.M / TP) * (p.Cout / TC);
    const int n_cu = device_cu_count();
    const int grid = ntiles < n_cu ? ntiles : n_cu;
    static std::atomic<uint64_t> attr_seen{0};
    if (first_use_on_device(attr_seen))
        (void)hipFuncSetAttribute((const void*)igemm_pers_kernel<EPI_PLAIN, true, PX_NONE, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    static std::atomic<unsigned> launch_no{0};
    const int cset = (int)(launch_no.fetch_add(1) % CSETS);
    launch_timed((igemm_pers_kernel<EPI_PLAIN, true, PX_NONE, true>), dim3(grid), dim3(512), lds, s, with_zero_page(p), ntiles, cset);
    return hipGetLastError();
}
#endif

// Upsample2D + conv as four 2x2 convolutions on the source grid (template parameter UP4); only igemm_pers_up.hip defines DM_IGEMM_PERS_UP.
// Always this kernel, for every batch size (no 128-row partner, no head / tail cut): a sample's bits cannot depend on its batch.
#ifdef DM_IGEMM_PERS_UP
static hipError_t launch_igemm_pers_up4_t(const IGemmParams& p, hipStream_t s) {
    DM_REQUIRE_ZERO_PAGE();
    constexpr int TP = 256, TC = 320;
    constexpr size_t lds = 2 * (size_t)(TP + TC) * 128 + 2 * AUX_BYTES;
    if (p.mode != IG_CONV2_UP4 || p.epi != EPI_PLAIN || p.ln_s || p.temb || p.res || p.X2 || p.X3 || p.ksplit > 1 || p.w_sample_stride ||
        p.Cout % TC != 0 || p.Cin % BK != 0 || p.C1 != p.Cin || p.OH != p.H || p.OW != p.W || p.H < 1 || p.W < 1 || p.H > 511 || p.W > 511 ||
        p.M < 2 || p.M != (p.M / (p.H * p.W)) * p.H * p.W || p.M / (p.H * p.W) > 8191) return hipErrorInvalidValue;
    // 32-bit element offsets: activations (row * channels + chunk), the four weight matrices, and 4 x tiles in an int
    if ((long long)p.M * p.Cin >= (1LL << 31) || 16LL * p.Cout * p.Cin >= (1LL << 32)) return hipErrorInvalidValue;
    const int tpp = ((p.M + TP - 1) / TP) * (p.Cout / TC);
    const int ntiles = 4 * tpp;
    const int n_cu = device_cu_count();
    const int grid = ntiles < n_cu ? ntiles : n_cu;
    static std::atomic<uint64_t> attr_seen{0};
    if (first_use_on_device(attr_seen))
        (void)hipFuncSetAttribute((const void*)igemm_pers_kernel<EPI_PLAIN, false, PX_NONE, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    static std::atomic<unsigned> launch_no{0};
    const int cset = (int)(launch_no.fetch_add(1) % CSETS);
    launch_timed((igemm_pers_kernel<EPI_PLAIN, false, PX_NONE, false, false, true>), dim3(grid), dim3(512), lds, s, with_zero_page(p), ntiles, cset);
    return hipGetLastError();
}
#endif

// split-K form: units = tiles * ksplit, fp32 partials (see EPI_PARTIAL); the caller runs the reduction afterwards
static hipError_t launch_igemm_pers_partial_t(const IGemmParams& p, hipStream_t s) {
    DM_REQUIRE_ZERO_PAGE();
    constexpr int TP = 256, TC = 320;
    constexpr size_t lds = 2 * (size_t)(TP + TC) * 128 + 2 * AUX_BYTES;
    const int units = ((p.M + TP - 1) / TP) * (p.Cout / TC) * p.ksplit;
    const int n_cu = device_cu_count();
    const int grid = units < n_cu ? units : n_cu;
    static std::atomic<uint64_t> attr_seen{0};
    if (first_use_on_device(attr_seen))
        (void)hipFuncSetAttribute((const void*)igemm_pers_kernel<EPI_PARTIAL, false, PX_NONE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    static std::atomic<unsigned> launch_no{0};
    const int cset = (int)(launch_no.fetch_add(1) % CSETS);
    launch_timed((igemm_pers_kernel<EPI_PARTIAL, false, PX_NONE>), dim3(grid), dim3(512), lds, s, with_zero_page(p), units, cset);
    return hipGetLastError();
}

}  // namespace dm
