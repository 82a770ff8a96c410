// igemm.hip — K1/K2/K3: implicit-GEMM on gfx950 MFMA for every matmul-shaped op of the SDv1.5
// U-Net the reference calls at diffmining/typicality/compute.py:100 / dift.py:191:
// 3x3 conv (stride 1, stride 2, nearest-upsampled input), 1x1 conv and Linear, with fused
// bias / time-embedding / residual / GEGLU epilogues.
//
// Formulation: Y[m][co] = sum_k X~[m][k] * Wp[co][k], m = output pixel (NHWC row), k = (tap, cin).
// The WEIGHT tile is the MFMA "A" operand (rows = output channels) and the ACTIVATION tile the "B"
// operand (cols = pixels), so each lane ends up holding 4 consecutive output channels of one
// pixel -> 8-byte channel-contiguous NHWC stores.
//
// Tile: 128 pixels x 160 channels x 64 k per step, 256 threads = 4 waves (2 channel halves x
// 2 pixel halves), each wave 80 channels x 64 pixels = 5x4 fragments of v_mfma_f32_16x16x32_f16.
// 160 divides every channel count of the network (320/640/1280/2560/5120/10240).
// Operand tiles are staged global -> registers -> LDS (XOR-swizzled 128-byte rows, conflict-free
// ds_read_b128), double buffered, one barrier per k step; the next tile's global loads are issued
// before the MFMA block and written to LDS after it.
#include "dm_kernels.h"
#include <cstdlib>

namespace dm {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BP = 128;          // pixels per block
constexpr int BC = 160;          // output channels per block
constexpr int BK = 64;           // k per step (one tap, 64 input channels)
constexpr int WT_BYTES = BC * BK * 2;
constexpr int XT_BYTES = BP * BK * 2;
constexpr int STAGE_BYTES = WT_BYTES + XT_BYTES;   // 36864
constexpr int NTHREADS = 256;

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

template <int EPI>
__device__ __forceinline__ void epilogue(const IGemmParams& p, floatx4 (&acc)[5][4], int p0, int c0out, int wc,
                                         int wp, int l15, int lg, int OHW) {
    // ---- epilogue: D[row = channel (lg*4 + r)][col = pixel l15] --------------------------------
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = p0 + wp * 64 + 16 * j + l15;
        if (m >= p.M) continue;
        const int n = (p.temb != nullptr) ? (m / OHW) : 0;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int c = c0out + wc * 80 + 16 * i + 4 * lg;
            float v0 = acc[i][j][0], v1 = acc[i][j][1], v2 = acc[i][j][2], v3 = acc[i][j][3];
            if (p.bias) {
                const half4 bv = *reinterpret_cast<const half4*>(p.bias + c);
                v0 += (float)bv[0]; v1 += (float)bv[1]; v2 += (float)bv[2]; v3 += (float)bv[3];
            }
            if (EPI == EPI_GEGLU) {
                // packed rows: [h0, h1, g0, g1]; out = fp16(h * fp16(gelu(g))) as fp16 autocast does
                const f16 h0 = (f16)v0, h1 = (f16)v1, g0 = (f16)v2, g1 = (f16)v3;
                const f16 q0 = (f16)gelu_erf((float)g0), q1 = (f16)gelu_erf((float)g1);
                const f16 o0 = (f16)((float)h0 * (float)q0), o1 = (f16)((float)h1 * (float)q1);
                const int oc = (c >> 4) * 8 + 2 * lg;
                typedef _Float16 half2_ __attribute__((ext_vector_type(2)));
                *reinterpret_cast<half2_*>(p.Y + (size_t)m * p.ldy + oc) = half2_{o0, o1};
            } else {
                half4 o = half4{(f16)v0, (f16)v1, (f16)v2, (f16)v3};
                if (p.temb) {
                    const half4 tv = *reinterpret_cast<const half4*>(p.temb + (size_t)n * p.temb_ld + c);
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (f16)((float)o[r] + (float)tv[r]);
                }
                if (p.res) {
                    const half4 rv = *reinterpret_cast<const half4*>(p.res + (size_t)m * p.ldres + c);
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (f16)((float)o[r] + (float)rv[r]);
                }
                *reinterpret_cast<half4*>(p.Y + (size_t)m * p.ldy + c) = o;
            }
        }
    }
}

// Epilogue staged through LDS: fragments (bias / time-embedding / GEGLU applied, rounded to fp16) are
// written to an LDS tile [TP px][TCO ch] (row stride padded by 8 B: conflict-free ds_write_b64 /
// b32), then copied out as whole rows with 16-byte stores (+ the residual read the same way).
// The direct fragment stores write 8-byte (GEGLU: 4-byte) pieces of 16 different 128-byte lines per
// instruction; on the wide, short-K linears that partial-line traffic bound the whole kernel.
template <int EPI, int NTH, int TP, int TC>
__device__ __forceinline__ void epilogue_lds(const IGemmParams& p, floatx4 (&acc)[5][4], char* smem, int p0,
                                             int c0out, int wc, int wp, int l15, int lg, int OHW) {
    constexpr int TCO = (EPI == EPI_GEGLU) ? TC / 2 : TC;     // output channels of the tile
    constexpr int ROWB = TCO * 2 + 8;
    __syncthreads();                                           // every wave is done with the operand tiles
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int pr = wp * 64 + 16 * j + l15;
        const int m = p0 + pr;
        const int n = (p.temb != nullptr && m < p.M) ? (m / OHW) : 0;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int cl = wc * 80 + 16 * i + 4 * lg;              // tile-local channel
            const int c = c0out + cl;
            float v0 = acc[i][j][0], v1 = acc[i][j][1], v2 = acc[i][j][2], v3 = acc[i][j][3];
            if (p.bias) {
                const half4 bv = *reinterpret_cast<const half4*>(p.bias + c);
                v0 += (float)bv[0]; v1 += (float)bv[1]; v2 += (float)bv[2]; v3 += (float)bv[3];
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
                if (p.temb && m < p.M) {
                    const half4 tv = *reinterpret_cast<const half4*>(p.temb + (size_t)n * p.temb_ld + c);
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (f16)((float)o[r] + (float)tv[r]);
                }
                *reinterpret_cast<half4*>(smem + pr * ROWB + cl * 2) = o;
            }
        }
    }
    __syncthreads();
    constexpr int CPR = TCO / 8;                               // 16-byte chunks per row
    const int c0o = (EPI == EPI_GEGLU) ? c0out / 2 : c0out;
    for (int idx = threadIdx.x; idx < TP * CPR; idx += NTH) {
        const int row = idx / CPR, ch = idx - row * CPR;
        const int m = p0 + row;
        if (m >= p.M) continue;
        const char* src = smem + row * ROWB + ch * 16;
        const half4 lo = *reinterpret_cast<const half4*>(src);
        const half4 hi = *reinterpret_cast<const half4*>(src + 8);
        half8 o = half8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        if (EPI != EPI_GEGLU && p.res) {
            const half8 rv = *reinterpret_cast<const half8*>(p.res + (size_t)m * p.ldres + c0o + ch * 8);
#pragma unroll
            for (int r = 0; r < 8; ++r) o[r] = (f16)((float)o[r] + (float)rv[r]);
        }
        *reinterpret_cast<half8*>(p.Y + (size_t)m * p.ldy + c0o + ch * 8) = o;
    }
}

template <int EPI>
__global__ __launch_bounds__(NTHREADS, 2)
void igemm_kernel(IGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wc = wid & 1;      // channel half of the block tile
    const int wp = wid >> 1;     // pixel half

    // ---- block -> tile, XCD aware: blocks that share a pixel tile run on the same XCD ----------
    const int tiles_c = p.Cout / BC;
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
    const int p0 = pt * BP;
    const int c0out = ct * BC;

    const int C1 = p.C1;
    const int C2 = p.Cin - C1;
    const int ntaps = (p.mode == IG_DENSE) ? 1 : 9;
    const int cpt = p.Cin / BK;            // k tiles per tap
    const int nk = ntaps * cpt;
    const int Ktot = ntaps * p.Cin;

    // ---- per-thread staging coordinates -------------------------------------------------------
    const int chunk = tid & 7;             // 16-byte chunk within the 128-byte k row
    const int r0 = tid >> 3;               // rows r0 + 32 i
    const int swz = (chunk ^ (r0 & 7)) << 4;

    int xn[4], xoh[4], xow[4];
    const int OHW = p.OH * p.OW;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = p0 + r0 + 32 * i;
        if (m < p.M) {
            const int n = m / OHW;
            const int rem = m - n * OHW;
            const int oh = rem / p.OW;
            xn[i] = n; xoh[i] = oh; xow[i] = rem - oh * p.OW;
        } else {
            xn[i] = -1; xoh[i] = 0; xow[i] = 0;
        }
    }
    const f16* wrow[5];
#pragma unroll
    for (int i = 0; i < 5; ++i)
        wrow[i] = p.Wp + (size_t)(c0out + r0 + 32 * i) * Ktot + chunk * 8;

    const float sh = (float)p.H / (float)p.OH;     // nearest-upsample source scale (mode IG_CONV3_UP)
    const float sw = (float)p.W / (float)p.OW;

    long long xpix[4];                     // source pixel linear index for the current tap, -1 = zero
    auto set_tap = [&](int tap) {
        const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            long long off = -1;
            if (xn[i] >= 0) {
                if (p.mode == IG_DENSE) {
                    off = (long long)(p0 + r0 + 32 * i);
                } else if (p.mode == IG_CONV3 || p.mode == IG_CONV3_S2) {
                    const int st = (p.mode == IG_CONV3_S2) ? 2 : 1;
                    const int ih = xoh[i] * st + dy - 1, iw = xow[i] * st + dx - 1;
                    if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W)
                        off = ((long long)xn[i] * p.H + ih) * p.W + iw;
                } else {   // conv on the nearest-upsampled image of size OH x OW
                    const int uh = xoh[i] + dy - 1, uw = xow[i] + dx - 1;
                    if (uh >= 0 && uh < p.OH && uw >= 0 && uw < p.OW) {
                        int ih = (int)floorf((float)uh * sh); ih = ih < p.H - 1 ? ih : p.H - 1;
                        int iw = (int)floorf((float)uw * sw); iw = iw < p.W - 1 ? iw : p.W - 1;
                        off = ((long long)xn[i] * p.H + ih) * p.W + iw;
                    }
                }
            }
            xpix[i] = off;
        }
    };

    u32x4 xr[4], wr[5];
    int ld_tap = 0, ld_cc = 0;             // (tap, channel-tile) of the NEXT tile to load
    auto load_regs = [&]() {
        if (ld_cc == 0) set_tap(ld_tap);
        const int c0 = ld_cc * BK;
        const f16* src; int cs, cb;
        if (c0 < C1) { src = p.X; cs = C1; cb = c0; } else { src = p.X2; cs = C2; cb = c0 - C1; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (xpix[i] >= 0)
                xr[i] = *reinterpret_cast<const u32x4*>(src + xpix[i] * cs + cb + chunk * 8);
            else
                xr[i] = u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            wr[i] = *reinterpret_cast<const u32x4*>(wrow[i]);
            wrow[i] += BK;
        }
        if (++ld_cc == cpt) { ld_cc = 0; ++ld_tap; }
    };
    auto store_lds = [&](int buf) {
        char* wt = smem + buf * STAGE_BYTES;
        char* xt = wt + WT_BYTES;
#pragma unroll
        for (int i = 0; i < 5; ++i)
            *reinterpret_cast<u32x4*>(wt + (r0 + 32 * i) * 128 + swz) = wr[i];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<u32x4*>(xt + (r0 + 32 * i) * 128 + swz) = xr[i];
    };

    floatx4 acc[5][4];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int l15 = lane & 15;
    const int lg = lane >> 4;
    const int a_row_off = (wc * 80 + l15) * 128;
    const int b_row_off = (wp * 64 + l15) * 128;

    auto compute = [&](int cur) {
        const char* wt = smem + cur * STAGE_BYTES;
        const char* xt = wt + WT_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int koff = (((4 * s + lg) ^ (l15 & 7)) << 4);
            half8 a[5], b[4];
#pragma unroll
            for (int i = 0; i < 5; ++i)
                a[i] = *reinterpret_cast<const half8*>(wt + a_row_off + i * 16 * 128 + koff);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                b[j] = *reinterpret_cast<const half8*>(xt + b_row_off + j * 16 * 128 + koff);
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };

    load_regs();
    store_lds(0);
    __syncthreads();
    for (int kt = 0; kt < nk - 1; ++kt) {
        const int cur = kt & 1;
        load_regs();                 // next tile: global loads in flight under the MFMA block
        compute(cur);
        store_lds(cur ^ 1);
        __syncthreads();
    }
    compute((nk - 1) & 1);

    epilogue<EPI>(p, acc, p0, c0out, wc, wp, l15, lg, OHW);
}


// ---------------------------------------------------------------------------------------------------
// glds variant: operand tiles go HBM/L2 -> LDS directly (global_load_lds_dwordx4, no VGPR staging,
// no ds_write).  The LDS image is lane-linear per wave instruction (8 rows x 128 B), so the XOR
// swizzle is applied on the per-lane SOURCE address (chunk ^= row&7) and again on the ds_read.
// Zero padding of the 3x3 halo = lanes pointed at a 128-byte zero page.
// Block tile = (64*WP pixels) x (80*WC channels); each wave still owns 64 px x 80 ch.
// ---------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(256))) unsigned char g_zero_page[256];

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// STAGES = 2: one tile of lookahead, plain barrier.  STAGES = 3: two tiles of lookahead; the wait at
// the top of a k step is a COUNTED vmcnt that leaves the newest stage's LDS-DMA in flight, and the
// barrier is a raw s_barrier (a __syncthreads() would drain vmcnt to 0).
template <int WP, int WC, int EPI, int STAGES>
__global__ __launch_bounds__(64 * WP * WC, 2)
void igemm_glds_kernel(IGemmParams p) {
    constexpr int NW = WP * WC;
    constexpr int TP = 64 * WP, TC = 80 * WC;
    constexpr int WBYTES = TC * 128, XBYTES = TP * 128, STAGE = WBYTES + XBYTES;
    constexpr int WG = TC / 8, XG = TP / 8;                 // 8-row groups (one glds instruction each)
    constexpr int WI = (WG + NW - 1) / NW, XI = (XG + NW - 1) / NW;
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

    const int lrow = lane >> 3;                              // row within the 8-row group
    const int lchunk = ((lane & 7) ^ lrow) * 8;              // swizzled source chunk (elements)

    // weight rows of this lane (group g = wid + k*NW)
    const f16* wsrc[WI];
#pragma unroll
    for (int k = 0; k < WI; ++k) {
        const int g = wid + k * NW;
        wsrc[k] = p.Wp + (size_t)(c0out + (g < WG ? g : 0) * 8 + lrow) * Ktot + lchunk;
    }
    // pixel rows of this lane
    int xn[XI], xoh[XI], xow[XI];
#pragma unroll
    for (int k = 0; k < XI; ++k) {
        const int g = wid + k * NW;
        const int m = p0 + g * 8 + lrow;
        if (g < XG && m < p.M) {
            const int n = m / OHW;
            const int rem = m - n * OHW;
            const int oh = rem / p.OW;
            xn[k] = n; xoh[k] = oh; xow[k] = rem - oh * p.OW;
        } else { xn[k] = -1; xoh[k] = 0; xow[k] = 0; }
    }
    const float sh = (float)p.H / (float)p.OH;
    const float sw = (float)p.W / (float)p.OW;
    long long xpix[XI];
    auto set_tap = [&](int tap) {
        const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
        for (int k = 0; k < XI; ++k) {
            long long off = -1;
            if (xn[k] >= 0) {
                if (p.mode == IG_DENSE) {
                    off = (long long)(p0 + (wid + k * NW) * 8 + lrow);
                } else if (p.mode == IG_CONV3 || p.mode == IG_CONV3_S2) {
                    const int st = (p.mode == IG_CONV3_S2) ? 2 : 1;
                    const int ih = xoh[k] * st + dy - 1, iw = xow[k] * st + dx - 1;
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
        }
    };

    const f16* zero = reinterpret_cast<const f16*>(g_zero_page) + lchunk;
    int ld_tap = 0, ld_cc = 0;
    auto issue = [&](int buf) {
        if (ld_cc == 0) set_tap(ld_tap);
        const int c0 = ld_cc * BK;
        const f16* src; int cs, cb;
        if (c0 < C1) { src = p.X; cs = C1; cb = c0; } else { src = p.X2; cs = C2; cb = c0 - C1; }
        char* wt = smem + buf * STAGE;
        char* xt = wt + WBYTES;
#pragma unroll
        for (int k = 0; k < WI; ++k) {
            const int g = wid + k * NW;
            if (g < WG) {
                __builtin_amdgcn_global_load_lds((gptr_t)wsrc[k], (lptr_t)(wt + g * 1024), 16, 0, 0);
            }
            wsrc[k] += BK;
        }
#pragma unroll
        for (int k = 0; k < XI; ++k) {
            const int g = wid + k * NW;
            if (g < XG) {
                const f16* a = (xpix[k] >= 0) ? (src + xpix[k] * cs + cb + lchunk) : zero;
                __builtin_amdgcn_global_load_lds((gptr_t)a, (lptr_t)(xt + g * 1024), 16, 0, 0);
            }
        }
        if (++ld_cc == cpt) { ld_cc = 0; ++ld_tap; }
    };

    floatx4 acc[5][4];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int l15 = lane & 15;
    const int lg = lane >> 4;
    const int a_row_off = (wc * 80 + l15) * 128;
    const int b_row_off = (wp * 64 + l15) * 128;

    // LDS-DMA instructions this wave issues per stage (wave-uniform; used for the counted wait)
    int my_loads = 0;
#pragma unroll
    for (int k = 0; k < WI; ++k) my_loads += (wid + k * NW < WG) ? 1 : 0;
#pragma unroll
    for (int k = 0; k < XI; ++k) my_loads += (wid + k * NW < XG) ? 1 : 0;
    my_loads = __builtin_amdgcn_readfirstlane(my_loads);

    issue(0);
    if (STAGES == 3 && nk > 1) issue(1);
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (STAGES == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();             // tile kt landed for every wave; buffer cur^1 is free again
            if (kt + 1 < nk) issue(cur ^ 1);
        } else {
            // outstanding: tile kt (+ tile kt+1 unless this is the last step)
            if (kt + 1 < nk) {
                if (my_loads == WI + XI) wait_vmcnt<WI + XI>();
                else if (my_loads == WI + XI - 1) wait_vmcnt<(WI + XI - 1 > 0 ? WI + XI - 1 : 0)>();
                else wait_vmcnt<(WI + XI - 2 > 0 ? WI + XI - 2 : 0)>();
            } else {
                wait_vmcnt<0>();
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + 2 < nk) issue(cur >= 1 ? cur - 1 : 2);       // (cur + 2) % 3
        }
        const char* wt = smem + cur * STAGE;
        const char* xt = wt + WBYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int koff = (((4 * s + lg) ^ (l15 & 7)) << 4);
            half8 a[5], b[4];
#pragma unroll
            for (int i = 0; i < 5; ++i)
                a[i] = *reinterpret_cast<const half8*>(wt + a_row_off + i * 16 * 128 + koff);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                b[j] = *reinterpret_cast<const half8*>(xt + b_row_off + j * 16 * 128 + koff);
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (STAGES == 2) cur ^= 1; else cur = (cur == 2) ? 0 : cur + 1;
    }
    epilogue<EPI>(p, acc, p0, c0out, wc, wp, l15, lg, OHW);
}


// ---------------------------------------------------------------------------------------------------
// Interleaved variant (128 px x 80*WC ch, 2*WC waves): same tiles, swizzle and epilogues as the glds
// kernel, but the LDS-DMA of the NEXT k tile is issued one instruction at a time BETWEEN groups of
// four MFMAs instead of in a burst after the barrier.  An LDS-DMA costs ~60-180 issue cycles; in a
// burst the matrix pipe idles behind it, interleaved it hides under the other wave's MFMAs.
// Every wave issues exactly WI + XI pieces per tile (no masks) - true for WC in {2, 4}.
// ---------------------------------------------------------------------------------------------------
template <int WC, int EPI>
__global__ __launch_bounds__(128 * WC, 2)
void igemm_il_kernel(IGemmParams p) {
    constexpr int WP = 2;
    constexpr int NW = WP * WC;
    constexpr int TP = 64 * WP, TC = 80 * WC;
    constexpr int WBYTES = TC * 128, XBYTES = TP * 128, STAGE = WBYTES + XBYTES;
    constexpr int WG = TC / 8, XG = TP / 8;
    constexpr int WI = WG / NW, XI = XG / NW;               // exact: 5 + 4 (WC=2), 5 + 2 (WC=4)
    static_assert(WG % NW == 0 && XG % NW == 0, "uniform LDS-DMA count per wave required");
    constexpr int NL = WI + XI;
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

    const f16* wsrc[WI];
#pragma unroll
    for (int k = 0; k < WI; ++k)
        wsrc[k] = p.Wp + (size_t)(c0out + (wid + k * NW) * 8 + lrow) * Ktot + lchunk;
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
                } else if (p.mode == IG_CONV3 || p.mode == IG_CONV3_S2) {
                    const int st = (p.mode == IG_CONV3_S2) ? 2 : 1;
                    const int ih = xoh[k] * st + dy - 1, iw = xow[k] * st + dx - 1;
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
    int ld_tap = 0, ld_cc = 0;
    auto prepare = [&]() {                                   // pointers for the next tile to load
        if (ld_cc == 0) set_tap(ld_tap);
        else if (ld_cc * BK == C1) {
#pragma unroll
            for (int k = 0; k < XI; ++k) if (xpix[k] >= 0) xsrc[k] = p.X2 + xpix[k] * C2 + lchunk;
        }
        if (++ld_cc == cpt) { ld_cc = 0; ++ld_tap; }
    };
    auto load_piece = [&](int buf, int idx) {                // idx in [0, NL): W pieces first, then X
        char* wt = smem + buf * STAGE;
        if (idx < WI) {
            __builtin_amdgcn_global_load_lds((gptr_t)wsrc[idx], (lptr_t)(wt + (wid + idx * NW) * 1024), 16, 0, 0);
            wsrc[idx] += BK;
        } else {
            const int k = idx - WI;
            __builtin_amdgcn_global_load_lds((gptr_t)xsrc[k], (lptr_t)(wt + WBYTES + (wid + k * NW) * 1024), 16, 0, 0);
            xsrc[k] += xinc[k];
        }
    };

    floatx4 acc[5][4];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int l15 = lane & 15;
    const int lg = lane >> 4;
    const int a_row_off = (wc * 80 + l15) * 128;
    const int b_row_off = (wp * 64 + l15) * 128;
    const int koff0 = ((lg ^ (l15 & 7)) << 4), koff1 = (((4 + lg) ^ (l15 & 7)) << 4);

    prepare();
#pragma unroll
    for (int i = 0; i < NL; ++i) load_piece(0, i);

    auto step = [&](int cur, bool more) {
        const char* wt = smem + cur * STAGE;
        const char* xt = wt + WBYTES;
        half8 a0[5], b0[4], a1[5], b1[4];
#ifdef DM_EXP_NODSREAD
#pragma unroll
        for (int i = 0; i < 5; ++i) { a0[i] = half8{1, 1, 1, 1, 1, 1, 1, 1}; a1[i] = a0[i]; asm volatile("" : "+v"(a0[i]), "+v"(a1[i])); }
#pragma unroll
        for (int j = 0; j < 4; ++j) { b0[j] = half8{1, 1, 1, 1, 1, 1, 1, 1}; b1[j] = b0[j]; asm volatile("" : "+v"(b0[j]), "+v"(b1[j])); }
        (void)wt; (void)xt;
#else
#pragma unroll
        for (int i = 0; i < 5; ++i) a0[i] = *reinterpret_cast<const half8*>(wt + a_row_off + i * 2048 + koff0);
#pragma unroll
        for (int j = 0; j < 4; ++j) b0[j] = *reinterpret_cast<const half8*>(xt + b_row_off + j * 2048 + koff0);
#endif
        if (more) prepare();
        // 10 groups of 4 MFMAs; one LDS-DMA piece after each of the first NL groups
#pragma unroll
        for (int g = 0; g < 10; ++g) {
            const int i = g % 5;
#ifndef DM_EXP_NODSREAD
            if (g == 2) {
#pragma unroll
                for (int ii = 0; ii < 5; ++ii) a1[ii] = *reinterpret_cast<const half8*>(wt + a_row_off + ii * 2048 + koff1);
#pragma unroll
                for (int j = 0; j < 4; ++j) b1[j] = *reinterpret_cast<const half8*>(xt + b_row_off + j * 2048 + koff1);
            }
#endif
#ifdef DM_EXP_NOMFMA
            if (g < 5) { asm volatile("" :: "v"(a0[i]), "v"(b0[0]), "v"(b0[1]), "v"(b0[2]), "v"(b0[3])); }
            else { asm volatile("" :: "v"(a1[i]), "v"(b1[0]), "v"(b1[1]), "v"(b1[2]), "v"(b1[3])); }
#else
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = (g < 5) ? __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[i], b0[j], acc[i][j], 0, 0, 0)
                                    : __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[i], b1[j], acc[i][j], 0, 0, 0);
#endif
#ifndef DM_EXP_NOGLDS
            if (more && g < NL) load_piece(cur ^ 1, g);
#endif
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
#ifdef DM_EXP_NOEPI
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" :: "v"(acc[i][j]));
    if (p.M < 0) epilogue_lds<EPI, 128 * WC, TP, TC>(p, acc, smem, p0, c0out, wc, wp, l15, lg, OHW);
#else
    epilogue_lds<EPI, 128 * WC, TP, TC>(p, acc, smem, p0, c0out, wc, wp, l15, lg, OHW);
#endif
}


// ---------------------------------------------------------------------------------------------------
// conv3x3 (stride 1) kernel with horizontal tap reuse.  LDS-DMA issue (~170 cycles per 1 KiB piece
// per SIMD, measured) is what bounds the generic kernel, so this one moves fewer bytes per MAC:
//   * tile 256 px x 160 ch (8 waves, each 64 px x 80 ch as before);
//   * k order (dy, cin-chunk, dx): the activation tile of (dy, chunk) is loaded ONCE as rows
//     [p0-1, p0+256] of the raster-ordered NHWC image and read at row offsets 0/1/2 for dx = 0/1/2
//     (lanes whose ow+dx-1 falls outside the image row get a zero fragment);
//   => 3 weight tiles + 1 activation tile per 3 k steps: 11.7 B/kMAC instead of 21.9.
// Same swizzle (source-side XOR on the DMA, XOR on the ds_read), same epilogues.
// ---------------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(512, 2)
void igemm_conv3_kernel(IGemmParams p) {
    constexpr int WP = 4, WC = 2, NW = 8;
    constexpr int TP = 256, TC = 160;
    constexpr int XROWS = TP + 8;                              // row 0 <-> pixel p0-1
    constexpr int XBYTES = XROWS * 128, WBYTES = TC * 128;
    constexpr int XG = XROWS / 8, WG = TC / 8;                 // 33, 20 pieces
    constexpr int XI = (XG + NW - 1) / NW, WI = (WG + NW - 1) / NW;   // 5, 3
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const xbuf0 = smem;                                  // [2][XBYTES]
    char* const wbuf0 = smem + 2 * XBYTES;                     // [2][WBYTES]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wid % WC;
    const int wp = wid / WC;

    const int tiles_c = p.Cout / TC;
    const int tiles_p = (p.M + TP - 1) / TP;
    const int nblk = gridDim.x;
    int v;
    {
        const int b = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = b & 7, loc = b >> 3;
        v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    // big weight matrices: neighbouring blocks share the weight tile; small ones: the pixel tile
    int pt, ct;
    if ((long long)p.Cout * p.Cin * 18 > (3ll << 20)) { ct = v / tiles_p; pt = v - ct * tiles_p; }
    else { pt = v / tiles_c; ct = v - pt * tiles_c; }
    const int p0 = pt * TP;
    const int c0out = ct * TC;

    const int C1 = p.C1, C2 = p.Cin - C1;
    const int cpt = p.Cin / BK;
    const int nk = 9 * cpt;
    const int Ktot = 9 * p.Cin;
    const int HW = p.H * p.W;

    const int lrow = lane >> 3;
    const int lchunk = ((lane & 7) ^ lrow) * 8;

    // weight pieces of this wave
    const f16* wbase[WI];
#pragma unroll
    for (int k = 0; k < WI; ++k) {
        const int g = wid + k * NW;
        wbase[k] = p.Wp + (size_t)(c0out + (g < WG ? g : 0) * 8 + lrow) * Ktot + lchunk;
    }
    // activation rows of this lane: tile row r = 8g + lrow <-> pixel m = p0 - 1 + r
    int xm[XI], xoh[XI];
#pragma unroll
    for (int k = 0; k < XI; ++k) {
        const int g = wid + k * NW;
        const int m = p0 - 1 + g * 8 + lrow;
        if (g < XG && m >= 0 && m < p.M) { xm[k] = m; xoh[k] = (m / p.W) % p.H; }
        else { xm[k] = -1; xoh[k] = 0; }
    }
    const f16* zero = reinterpret_cast<const f16*>(g_zero_page) + lchunk;

    auto load_w = [&](int buf, int tap, int cc) {
        char* wt = wbuf0 + buf * WBYTES;
        const int koffs = tap * p.Cin + cc * BK;
#pragma unroll
        for (int k = 0; k < WI; ++k) {
            const int g = wid + k * NW;
            if (g < WG) __builtin_amdgcn_global_load_lds((gptr_t)(wbase[k] + koffs), (lptr_t)(wt + g * 1024), 16, 0, 0);
        }
    };
    auto load_x = [&](int buf, int dy, int cc) {
        char* xt = xbuf0 + buf * XBYTES;
        const int c0 = cc * BK;
        const f16* src; int cs, cb;
        if (c0 < C1) { src = p.X; cs = C1; cb = c0; } else { src = p.X2; cs = C2; cb = c0 - C1; }
#pragma unroll
        for (int k = 0; k < XI; ++k) {
            const int g = wid + k * NW;
            if (g < XG) {
                const int ih = xoh[k] + dy - 1;
                const bool ok = (xm[k] >= 0) && (ih >= 0) && (ih < p.H);
                const long long pix = (long long)xm[k] + (long long)(dy - 1) * p.W;
                const f16* a = ok ? (src + pix * cs + cb + lchunk) : zero;
                __builtin_amdgcn_global_load_lds((gptr_t)a, (lptr_t)(xt + g * 1024), 16, 0, 0);
            }
        }
    };

    floatx4 acc[5][4];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int l15 = lane & 15;
    const int lg = lane >> 4;
    const int a_row_off = (wc * 80 + l15) * 128;
    const int koffA0 = ((lg ^ (l15 & 7)) << 4), koffA1 = (((4 + lg) ^ (l15 & 7)) << 4);
    // horizontal validity of this lane's 4 pixels (fragment j) for dx = 0 (needs ow >= 1) and dx = 2
    bool okL[4], okR[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = p0 + wp * 64 + 16 * j + l15;
        const int ow = m % p.W;
        okL[j] = ow >= 1; okR[j] = ow < p.W - 1;
    }
    (void)HW;

    // step kt -> (dy, cc, dx), dx innermost
    int ld_dy = 0, ld_cc = 0, ld_dx = 0;        // coordinates of the NEXT step to load
    int ld_xbuf = 0, ld_wbuf = 0;
    auto issue_next = [&]() {
        if (ld_dx == 0) { load_x(ld_xbuf, ld_dy, ld_cc); }
        load_w(ld_wbuf, ld_dy * 3 + ld_dx, ld_cc);
        ld_wbuf ^= 1;
        if (++ld_dx == 3) { ld_dx = 0; ld_xbuf ^= 1; if (++ld_cc == cpt) { ld_cc = 0; ++ld_dy; } }
    };

    issue_next();
    int dx = 0, xb = 0, wb = 0;
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) issue_next();
        const char* wt = wbuf0 + wb * WBYTES;
        const char* xt = xbuf0 + xb * XBYTES;
        // B rows: tile row = (wp*64 + 16j + l15) + dx ; swizzle term (row & 7) = (l15 + dx) & 7
        const int rsw = (l15 + dx) & 7;
        const int b_row_off = (wp * 64 + l15 + dx) * 128;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int koffB = (((4 * s + lg) ^ rsw) << 4);
            const int koffA = s ? koffA1 : koffA0;
            half8 a[5], b[4];
#pragma unroll
            for (int i = 0; i < 5; ++i) a[i] = *reinterpret_cast<const half8*>(wt + a_row_off + i * 2048 + koffA);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                b[j] = *reinterpret_cast<const half8*>(xt + b_row_off + j * 2048 + koffB);
                const bool ok = (dx == 1) || (dx == 0 ? okL[j] : okR[j]);
                if (!ok) b[j] = half8{0, 0, 0, 0, 0, 0, 0, 0};
            }
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        wb ^= 1;
        if (++dx == 3) { dx = 0; xb ^= 1; }
    }
    const int OHW = p.OH * p.OW;
    epilogue<EPI>(p, acc, p0, c0out, wc, wp, l15, lg, OHW);
}


// ---------------------------------------------------------------------------------------------------
// p3: 256 px x 160 ch tile, 8 waves (64 px x 80 ch each), THREE LDS stages, fragments double
// buffered in registers.  Per k step: ONE barrier, LDS-DMA of tile kt+2 interleaved with the MFMAs,
// and the ds_reads of the next half-step always issued under the current half-step's MFMAs:
//     block 1:  ds_read frags(s=1, tile kt)     || 20 MFMA on frags(s=0, tile kt)  || DMA(tile kt+2)
//     counted vmcnt (tile kt+1 landed, tile kt+2 may be in flight) ; lgkmcnt(0) ; s_barrier
//     block 2:  ds_read frags(s=0, tile kt+1)   || 20 MFMA on frags(s=1, tile kt)
// ---------------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(512, 2)
void igemm_p3_kernel(IGemmParams p) {
    constexpr int WP = 4, WC = 2, NW = 8;
    constexpr int TP = 256, TC = 160;
    constexpr int WBYTES = TC * 128, XBYTES = TP * 128, STAGE = WBYTES + XBYTES;
    constexpr int WG = TC / 8, XG = TP / 8;                  // 20, 32
    constexpr int WI = (WG + NW - 1) / NW, XI = XG / NW;     // 3 (waves 4..7 issue 2), 4
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wid % WC;
    const int wp = wid / WC;
    const bool w3 = (wid + 2 * NW < WG);                     // this wave issues a third weight piece

    const int tiles_c = p.Cout / TC;
    const int tiles_p = (p.M + TP - 1) / TP;
    const int nblk = gridDim.x;
    int v;
    {
        const int b = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = b & 7, loc = b >> 3;
        v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    int pt, ct;
    if ((long long)p.Cout * p.Cin * ((p.mode == IG_DENSE) ? 2 : 18) > (3ll << 20)) { ct = v / tiles_p; pt = v - ct * tiles_p; }
    else { pt = v / tiles_c; ct = v - pt * tiles_c; }
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

    const f16* wsrc[WI];
#pragma unroll
    for (int k = 0; k < WI; ++k) {
        const int g = wid + k * NW;
        wsrc[k] = p.Wp + (size_t)(c0out + (g < WG ? g : 0) * 8 + lrow) * Ktot + lchunk;
    }
    int xn[XI], xoh[XI], xow[XI];
#pragma unroll
    for (int k = 0; k < XI; ++k) {
        const int m = p0 + (wid + k * NW) * 8 + lrow;
        if (m < p.M) {
            if (p.mode == IG_DENSE) { xn[k] = 0; xoh[k] = 0; xow[k] = m; }
            else {
                const int n = m / OHW;
                const int rem = m - n * OHW;
                const int oh = rem / p.OW;
                xn[k] = n; xoh[k] = oh; xow[k] = rem - oh * p.OW;
            }
        } else { xn[k] = -1; xoh[k] = 0; xow[k] = 0; }
    }
    const float sh = (float)p.H / (float)p.OH;
    const float sw = (float)p.W / (float)p.OW;
    const f16* zero = reinterpret_cast<const f16*>(g_zero_page) + lchunk;

    const f16* xsrc[XI];
    int xinc[XI];
    long long xpix[XI];
    auto set_tap = [&](int tap) __attribute__((always_inline)) {
        const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
        for (int k = 0; k < XI; ++k) {
            long long off = -1;
            if (xn[k] >= 0) {
                if (p.mode == IG_DENSE) {
                    off = (long long)xow[k];
                } else if (p.mode == IG_CONV3 || p.mode == IG_CONV3_S2) {
                    const int st = (p.mode == IG_CONV3_S2) ? 2 : 1;
                    const int ih = xoh[k] * st + dy - 1, iw = xow[k] * st + dx - 1;
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
    int ld_tap = 0, ld_cc = 0;
    auto prepare = [&]() __attribute__((always_inline)) {
        if (ld_cc == 0) set_tap(ld_tap);
        else if (ld_cc * BK == C1) {
#pragma unroll
            for (int k = 0; k < XI; ++k) if (xpix[k] >= 0) xsrc[k] = p.X2 + xpix[k] * C2 + lchunk;
        }
        if (++ld_cc == cpt) { ld_cc = 0; ++ld_tap; }
    };
    // piece idx: 0..1 weight (always), 2..5 activation, 6 = third weight piece (waves 0..3 only)
    auto load_piece = [&](char* stage, int idx) __attribute__((always_inline)) {
        if (idx < 2) {
            __builtin_amdgcn_global_load_lds((gptr_t)wsrc[idx], (lptr_t)(stage + (wid + idx * NW) * 1024), 16, 0, 0);
            wsrc[idx] += BK;
        } else if (idx < 2 + XI) {
            const int k = idx - 2;
            __builtin_amdgcn_global_load_lds((gptr_t)xsrc[k], (lptr_t)(stage + WBYTES + (wid + k * NW) * 1024), 16, 0, 0);
            xsrc[k] += xinc[k];
        } else {
            if (w3) __builtin_amdgcn_global_load_lds((gptr_t)wsrc[2], (lptr_t)(stage + (wid + 2 * NW) * 1024), 16, 0, 0);
            wsrc[2] += BK;
        }
    };
    constexpr int NPIECE = 2 + XI + 1;                        // 7 issue slots (the last may be empty)

    floatx4 acc[5][4];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int l15 = lane & 15;
    const int lg = lane >> 4;
    const int a_row_off = (wc * 80 + l15) * 128;
    const int b_row_off = WBYTES + (wp * 64 + l15) * 128;
    const int koff0 = ((lg ^ (l15 & 7)) << 4), koff1 = (((4 + lg) ^ (l15 & 7)) << 4);

    half8 a0[5], b0[4], a1[5], b1[4];
    auto read_frags = [&](const char* stage, int koff, half8 (&a)[5], half8 (&b)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 5; ++i) a[i] = *reinterpret_cast<const half8*>(stage + a_row_off + i * 2048 + koff);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const half8*>(stage + b_row_off + j * 2048 + koff);
    };
    auto wait_landed = [&](bool newer_in_flight) __attribute__((always_inline)) {
        // all of this wave's pieces except those of the newest tile have landed
        if (!newer_in_flight) wait_vmcnt<0>();
        else if (w3) wait_vmcnt<7>();
        else wait_vmcnt<6>();
    };

    // ---- prologue: tiles 0 and 1 in flight, fragments (s=0) of tile 0 in registers -------------
    prepare();
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) load_piece(smem, i);
    if (nk > 1) {
        prepare();
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) load_piece(smem + STAGE, i);
    }
    wait_landed(nk > 1);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_frags(smem, koff0, a0, b0);

    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const int nxt = (cur == 2) ? 0 : cur + 1;
        const int nn = (nxt == 2) ? 0 : nxt + 1;
        const char* st = smem + cur * STAGE;
        char* st2 = smem + nn * STAGE;
        const bool more2 = (kt + 2 < nk);
        if (more2) prepare();
        // ---- block 1 ----
        read_frags(st, koff1, a1, b1);
#pragma unroll
        for (int i = 0; i < 5; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[i], b0[j], acc[i][j], 0, 0, 0);
            if (more2) { load_piece(st2, i); if (i + 5 < NPIECE) load_piece(st2, i + 5); }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- hand-off ----
        if (kt + 1 < nk) {
            wait_landed(more2);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            read_frags(smem + nxt * STAGE, koff0, a0, b0);
        }
        // ---- block 2 ----
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[i], b1[j], acc[i][j], 0, 0, 0);
        cur = nxt;
    }
    epilogue_lds<EPI, 512, TP, TC>(p, acc, smem, p0, c0out, wc, wp, l15, lg, OHW);
}

}  // namespace

template <int WP, int WC, int STAGES>
static hipError_t launch_glds(const IGemmParams& p, hipStream_t s) {
    constexpr int TP = 64 * WP, TC = 80 * WC;
    constexpr size_t lds = STAGES * (size_t)(TP + TC) * 128;
    const int tiles_p = (p.M + TP - 1) / TP;
    const int tiles_c = p.Cout / TC;
    dim3 grid(tiles_p * tiles_c), block(64 * WP * WC);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)igemm_glds_kernel<WP, WC, EPI_PLAIN, STAGES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)igemm_glds_kernel<WP, WC, EPI_GEGLU, STAGES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    if (p.epi == EPI_GEGLU)
        hipLaunchKernelGGL((igemm_glds_kernel<WP, WC, EPI_GEGLU, STAGES>), grid, block, lds, s, p);
    else
        hipLaunchKernelGGL((igemm_glds_kernel<WP, WC, EPI_PLAIN, STAGES>), grid, block, lds, s, p);
    return hipGetLastError();
}

template <int WC>
static hipError_t launch_il(const IGemmParams& p, hipStream_t s) {
    constexpr int TP = 128, TC = 80 * WC;
    constexpr size_t lds = 2 * (size_t)(TP + TC) * 128;
    const int tiles_p = (p.M + TP - 1) / TP;
    const int tiles_c = p.Cout / TC;
    dim3 grid(tiles_p * tiles_c), block(128 * WC);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)igemm_il_kernel<WC, EPI_PLAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)igemm_il_kernel<WC, EPI_GEGLU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    if (p.epi == EPI_GEGLU)
        hipLaunchKernelGGL((igemm_il_kernel<WC, EPI_GEGLU>), grid, block, lds, s, p);
    else
        hipLaunchKernelGGL((igemm_il_kernel<WC, EPI_PLAIN>), grid, block, lds, s, p);
    return hipGetLastError();
}

static hipError_t launch_conv3(const IGemmParams& p, hipStream_t s) {
    constexpr int TP = 256, TC = 160;
    constexpr size_t lds = 2 * (size_t)((TP + 8) + TC) * 128;
    const int tiles_p = (p.M + TP - 1) / TP;
    const int tiles_c = p.Cout / TC;
    dim3 grid(tiles_p * tiles_c), block(512);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)igemm_conv3_kernel<EPI_PLAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((igemm_conv3_kernel<EPI_PLAIN>), grid, block, lds, s, p);
    return hipGetLastError();
}

static hipError_t launch_p3(const IGemmParams& p, hipStream_t s) {
    constexpr int TP = 256, TC = 160;
    constexpr size_t lds = 3 * (size_t)(TP + TC) * 128;
    const int tiles_p = (p.M + TP - 1) / TP;
    const int tiles_c = p.Cout / TC;
    dim3 grid(tiles_p * tiles_c), block(512);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)igemm_p3_kernel<EPI_PLAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)igemm_p3_kernel<EPI_GEGLU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    if (p.epi == EPI_GEGLU) hipLaunchKernelGGL((igemm_p3_kernel<EPI_GEGLU>), grid, block, lds, s, p);
    else hipLaunchKernelGGL((igemm_p3_kernel<EPI_PLAIN>), grid, block, lds, s, p);
    return hipGetLastError();
}

int igemm_variant() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("DM_IGEMM"); v = e ? atoi(e) : 6; }
    return v;
}

hipError_t launch_igemm(const IGemmParams& p, hipStream_t s) {
    if (p.Cout % BC != 0 || p.Cin % BK != 0 || p.C1 % BK != 0 || p.M <= 0) return hipErrorInvalidValue;
    const int var = igemm_variant();
    if (var == 1) return launch_glds<2, 2, 2>(p, s);
    if (var == 2) return launch_glds<4, 2, 2>(p, s);
    if (var == 3 && p.Cout % 320 == 0) return launch_glds<2, 4, 2>(p, s);
    if (var == 3) return launch_glds<2, 2, 2>(p, s);
    if (var == 9) return launch_p3(p, s);
    if (var == 8 && p.mode == IG_CONV3 && p.epi == EPI_PLAIN && p.OH == p.H && p.OW == p.W) return launch_conv3(p, s);
    if ((var == 6 || var == 8) && p.Cout % 320 == 0) return launch_il<4>(p, s);
    if (var == 8) return launch_il<2>(p, s);
    if (var == 6 || var == 7) return launch_il<2>(p, s);
    if (var == 4) return launch_glds<4, 2, 3>(p, s);
    if (var == 5) return launch_glds<2, 2, 3>(p, s);
    const int tiles_p = (p.M + BP - 1) / BP;
    const int tiles_c = p.Cout / BC;
    dim3 grid(tiles_p * tiles_c), block(NTHREADS);
    const size_t lds = 2 * STAGE_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)igemm_kernel<EPI_PLAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)igemm_kernel<EPI_GEGLU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    if (p.epi == EPI_GEGLU)
        hipLaunchKernelGGL(igemm_kernel<EPI_GEGLU>, grid, block, lds, s, p);
    else
        hipLaunchKernelGGL(igemm_kernel<EPI_PLAIN>, grid, block, lds, s, p);
    return hipGetLastError();
}

}  // namespace dm
