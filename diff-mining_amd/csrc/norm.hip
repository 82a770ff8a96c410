// norm.hip — K6 GroupNorm (fp32 statistics, two-stage deterministic reduction; apply + optional
// SiLU) and K7 LayerNorm for the SDv1.5 U-Net (`ResnetBlock2D.norm1/2`, `Transformer2DModel.norm`,
// `BasicTransformerBlock.norm1/2/3`, `conv_norm_out`), reached from compute.py:100 / dift.py:191.
// Under the reference's fp16 autocast these ops are promoted to fp32; here inputs are the fp16
// residual stream (NHWC), statistics are fp32 per thread and combined in fp64 (no atomics: the
// scored outputs must be run-to-run deterministic), outputs are rounded to fp16 once, at the point
// where the reference's next conv/linear would cast them.
// All kernels are HBM-bound: 16-byte loads/stores, one pass for stats, one for apply.
#include "dm_kernels.h"
#include <cstdlib>

namespace dm {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int GN_PIX_MAX = 256;     // pixels per stats block (fewer when the image is small)

// grid (chunks, N); block = (C/8 column threads) x R row groups, <= 320 threads.
// The channel concat cat([X, X2]) of the up blocks is read in place (never materialised).
__global__ void gn_stats_partial(const f16* __restrict__ X, const f16* __restrict__ X2, int HW, int C, int C1,
                                 int G, int R, int GN_PIX, double* __restrict__ partial) {
    extern __shared__ float sh[];   // [R][2][C]: sums and sums of squares planar (a thread's 8 channels = two 16-byte stores;
                                    // interleaved (sum, sq) pairs put 64 B between threads: a 16-way bank conflict per store)
    const int cols = C >> 3;
    const int n = blockIdx.y;
    const int chunk = blockIdx.x;
    const int t = threadIdx.x;
    const int col = t % cols;
    const int rg = t / cols;
    const int px0 = chunk * GN_PIX;
    const int px1 = min(HW, px0 + GN_PIX);
    float s[8], q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s[k] = 0.f; q[k] = 0.f; }
    const int c = col * 8;
    const int C2 = C - C1;
    if (rg < R) {
        // four pixels per trip: the loads are independent, the accumulation order is unchanged (px ascending)
        const f16* base = (c < C1) ? (X + (size_t)n * HW * C1 + c) : (X2 + (size_t)n * HW * C2 + (c - C1));
        const size_t cs = (c < C1) ? (size_t)C1 : (size_t)C2;
        int px = px0 + rg;
        for (; px + 3 * R < px1; px += 4 * R) {
            half8 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const half8*>(base + (size_t)(px + u * R) * cs);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k) { const float f = (float)v[u][k]; s[k] += f; q[k] += f * f; }
        }
        for (; px < px1; px += R) {
            const half8 v = *reinterpret_cast<const half8*>(base + (size_t)px * cs);
#pragma unroll
            for (int k = 0; k < 8; ++k) { const float f = (float)v[k]; s[k] += f; q[k] += f * f; }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            sh[((size_t)rg * 2 + 0) * C + c + k] = s[k];
            sh[((size_t)rg * 2 + 1) * C + c + k] = q[k];
        }
    }
    __syncthreads();
    if (t < G) {
        const int cpg = C / G;
        double ds = 0.0, dq = 0.0;
        for (int r = 0; r < R; ++r)
            for (int k = 0; k < cpg; ++k) {
                ds += (double)sh[((size_t)r * 2 + 0) * C + t * cpg + k];
                dq += (double)sh[((size_t)r * 2 + 1) * C + t * cpg + k];
            }
        double* out = partial + (((size_t)n * gridDim.x + chunk) * G + t) * 2;
        out[0] = ds; out[1] = dq;
    }
}

// ---- GroupNorm statistics as per-(64-row block, channel pair) sums (r05; arithmetic: dm_kernels.h, launch_gn_blocks) ----------
// One wave = one 64-row block; a pass covers 64 channels: lane (e, lg) reads the 16 channels 64 pass + 16 lg of rows 16 j + e (two
// 16-byte loads, four lanes = one 128-byte line of the row).  The persistent GEMM epilogues compute the same sums from their
// packed output registers; this kernel serves the producers that do not (128-row tile, split-K, tails of a head / tail cut).
__global__ __launch_bounds__(256) void gn_blocks_kernel(const f16* __restrict__ X, int nblk, int C, int blk0, float* __restrict__ blocks) {
    const int lane = threadIdx.x & 63, e = lane & 15, lg = lane >> 4;
    const int b = blk0 + blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= nblk) return;
    const f16* xb = X + (size_t)b * 64 * C;
    float* ob = blocks + (size_t)b * C;
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int c = c0 + 16 * lg;
        const bool ok = c < C;                       // C % 16 == 0
        float sa[8], qa[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { sa[k] = 0.f; qa[k] = 0.f; }
        uint4 v[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint4* src = reinterpret_cast<const uint4*>(xb + (size_t)(16 * j + e) * C + (ok ? c : 0));
            v[j][0] = src[0]; v[j][1] = src[1];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            gn_pair_acc(sa[0], qa[0], v[j][0].x); gn_pair_acc(sa[1], qa[1], v[j][0].y);
            gn_pair_acc(sa[2], qa[2], v[j][0].z); gn_pair_acc(sa[3], qa[3], v[j][0].w);
            gn_pair_acc(sa[4], qa[4], v[j][1].x); gn_pair_acc(sa[5], qa[5], v[j][1].y);
            gn_pair_acc(sa[6], qa[6], v[j][1].z); gn_pair_acc(sa[7], qa[7], v[j][1].w);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { sa[k] = gn_row16_sum(sa[k]); qa[k] = gn_row16_sum(qa[k]); }
        if (e == 0 && ok) {
            float4* o = reinterpret_cast<float4*>(ob + c);
            o[0] = make_float4(sa[0], qa[0], sa[1], qa[1]);
            o[1] = make_float4(sa[2], qa[2], sa[3], qa[3]);
            o[2] = make_float4(sa[4], qa[4], sa[5], qa[5]);
            o[3] = make_float4(sa[6], qa[6], sa[7], qa[7]);
        }
    }
}

// blocks -> partial [N][1][G][2] fp64: thread (g, sl) adds the blocks sl, sl + S, .. of its sample (pairs of the group ascending
// inside a block), the S slices are then added in order — a fixed order per (C, HW, G), so a sample's bits do not depend on its batch.
__global__ __launch_bounds__(256) void gn_blocks_final_kernel(const float* __restrict__ blocks, int nb, int C, int G, double* __restrict__ partial) {
    __shared__ double sh[2 * 256];
    const int n = blockIdx.x, t = threadIdx.x;
    const int S = 256 / G, g = t % G, sl = t / G, cpg = C / G;
    double ds = 0.0, dq = 0.0;
    if (sl < S) {
        const float* in = blocks + (size_t)n * nb * C + (size_t)g * cpg;
        for (int b = sl; b < nb; b += S) {
            const float* r = in + (size_t)b * C;
            for (int k = 0; k < cpg; k += 2) { ds += (double)r[k]; dq += (double)r[k + 1]; }
        }
    }
    sh[2 * t] = ds; sh[2 * t + 1] = dq;
    __syncthreads();
    if (t < G) {
        ds = 0.0; dq = 0.0;
        for (int k = 0; k < S; ++k) { ds += sh[2 * (k * G + t)]; dq += sh[2 * (k * G + t) + 1]; }
        partial[((size_t)n * G + t) * 2] = ds;
        partial[((size_t)n * G + t) * 2 + 1] = dq;
    }
}

__global__ void gn_merge_skip_kernel(const double* __restrict__ px, const double* __restrict__ pskip, int Ns, int chunks, int G, int G1, int m,
                                     double* __restrict__ out) {
    const int n = blockIdx.x, g = threadIdx.x;
    if (g >= G) return;
    double ds = 0.0, dq = 0.0;
    if (g < G1) {
        const double* in = px + ((size_t)n * chunks * G1 + g) * 2;
        for (int c = 0; c < chunks; ++c) { ds += in[(size_t)c * G1 * 2]; dq += in[(size_t)c * G1 * 2 + 1]; }
    } else {
        const double* in = pskip + ((size_t)(n % Ns) * chunks * G + (size_t)(g - G1) * m) * 2;
        for (int c = 0; c < chunks; ++c)
            for (int j = 0; j < m; ++j) { ds += in[((size_t)c * G + j) * 2]; dq += in[((size_t)c * G + j) * 2 + 1]; }
    }
    out[((size_t)n * G + g) * 2] = ds;
    out[((size_t)n * G + g) * 2 + 1] = dq;
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

// grid (pixel chunks, N); thread = fixed 8-channel column (its affine lives in registers) x row group.
// Prologue (r03: the former gn_stats_final launch, 61 per forward): the first G threads combine the per-chunk partial
// sums of their group in fp64 in a fixed order (chunk ascending: deterministic, batch independent) into (mean, rstd) in
// LDS; every thread then forms the per-channel affine of its 8 channels,
//   y = x * a + b,  a = rstd * gamma[c],  b = beta[c] - mean * a      (fp64, rounded to fp32 once).
__global__ void gn_apply_kernel(const f16* __restrict__ X, const f16* __restrict__ X2, int HW, int C, int C1, int R,
                                int pix_per_block, const double* __restrict__ partial, int chunks, int G, double count, float eps,
                                const float* __restrict__ gamma, const float* __restrict__ beta, int silu, f16* __restrict__ Y) {
    __shared__ double sh_stat[2 * 64];          // (mean, rstd) per group, G <= 64
    __shared__ double sh_part[2 * 1024];        // one (sum, sum of squares) per thread: slice s = t / G of group t % G
    const int cols = C >> 3;
    const int n = blockIdx.y;
    const int t = threadIdx.x;
    {
        // all threads fetch (one round trip instead of `chunks` dependent ones): slice s of group g adds the chunks
        // s, s + S, ... (S = blockDim / G slices); the S slices are then added in order — a fixed tree per (C, HW)
        const int S = (int)blockDim.x / G, g = t % G, sl = t / G;
        double ds = 0.0, dq = 0.0;
        if (sl < S) {
            const double* in = partial + ((size_t)n * chunks * G + g) * 2;
            for (int c = sl; c < chunks; c += S) { ds += in[(size_t)c * G * 2]; dq += in[(size_t)c * G * 2 + 1]; }
            sh_part[2 * t] = ds; sh_part[2 * t + 1] = dq;
        }
        __syncthreads();
        if (t < G) {
            ds = 0.0; dq = 0.0;
            for (int k = 0; k < S; ++k) { ds += sh_part[2 * (k * G + t)]; dq += sh_part[2 * (k * G + t) + 1]; }
            const double mean = ds / count;
            double var = dq / count - mean * mean;
            var = var > 0.0 ? var : 0.0;
            sh_stat[2 * t] = mean;
            sh_stat[2 * t + 1] = 1.0 / sqrt(var + (double)eps);
        }
    }
    __syncthreads();
    const int col = t % cols;
    const int rg = t / cols;
    if (rg >= R) return;
    const int c = col * 8;
    const int C2 = C - C1;
    const int cpg = C / G;
    float a[8], b[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int g = (c + k) / cpg;
        const double ad = sh_stat[2 * g + 1] * (double)gamma[c + k];
        a[k] = (float)ad;
        b[k] = (float)((double)beta[c + k] - sh_stat[2 * g] * ad);
    }
    const int px0 = blockIdx.x * pix_per_block;
    const int px1 = min(HW, px0 + pix_per_block);
    for (int px = px0 + rg; px < px1; px += R) {
        const size_t pix = (size_t)n * HW + px;
        const f16* src = (c < C1) ? (X + pix * C1 + c) : (X2 + pix * C2 + (c - C1));
        const half8 v = *reinterpret_cast<const half8*>(src);
        half8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float y = fmaf((float)v[k], a[k], b[k]);
            if (silu) y = silu_f(y);
            o[k] = (f16)y;
        }
        *reinterpret_cast<half8*>(Y + pix * C + c) = o;
    }
}


// GroupNorm folded into the following 1x1 convolution (see launch_gn_fold in dm_kernels.h).  grid (Cout / ROWS, N), 256 threads.
// Prologue = gn_apply_kernel's (the same reduction order: identical (mean, rstd)); then a[c], b[c] of the sample in LDS, and
// per output row: the scaled weights (one rounding, like folding LayerNorm's gamma) and the fp32 bias row entry (wave-level
// sum in a fixed order).
constexpr int GNF_ROWS = 8;
__global__ __launch_bounds__(256)
void gn_fold_kernel(const double* __restrict__ partial, int chunks, int G, int C, double count, float eps,
                    const float* __restrict__ gamma, const float* __restrict__ beta, const f16* __restrict__ W,
                    const f16* __restrict__ bias, int Cout, f16* __restrict__ Wn, float* __restrict__ tn) {
    __shared__ double sh_stat[2 * 64];
    __shared__ double sh_part[2 * 256];
    __shared__ float sh_a[2560], sh_b[2560];
    __shared__ float sh_red[4][GNF_ROWS];
    const int n = blockIdx.y, t = threadIdx.x;
    {
        const int S = 256 / G, g = t % G, sl = t / G;
        double ds = 0.0, dq = 0.0;
        if (sl < S) {
            const double* in = partial + ((size_t)n * chunks * G + g) * 2;
            for (int c = sl; c < chunks; c += S) { ds += in[(size_t)c * G * 2]; dq += in[(size_t)c * G * 2 + 1]; }
            sh_part[2 * t] = ds; sh_part[2 * t + 1] = dq;
        }
        __syncthreads();
        if (t < G) {
            ds = 0.0; dq = 0.0;
            for (int k = 0; k < S; ++k) { ds += sh_part[2 * (k * G + t)]; dq += sh_part[2 * (k * G + t) + 1]; }
            const double mean = ds / count;
            double var = dq / count - mean * mean;
            var = var > 0.0 ? var : 0.0;
            sh_stat[2 * t] = mean;
            sh_stat[2 * t + 1] = 1.0 / sqrt(var + (double)eps);
        }
    }
    __syncthreads();
    const int cpg = C / G;
    for (int c = t; c < C; c += 256) {
        const int g = c / cpg;
        const double ad = sh_stat[2 * g + 1] * (double)gamma[c];
        sh_a[c] = (float)ad;
        sh_b[c] = (float)((double)beta[c] - sh_stat[2 * g] * ad);
    }
    __syncthreads();
    const int o0 = blockIdx.x * GNF_ROWS;
    const int lane = t & 63, wv = t >> 6;
    float acc[GNF_ROWS];
#pragma unroll
    for (int r = 0; r < GNF_ROWS; ++r) acc[r] = 0.f;
    for (int c8 = t * 8; c8 < C; c8 += 256 * 8) {
#pragma unroll
        for (int r = 0; r < GNF_ROWS; ++r) {
            const int o = o0 + r;
            if (o >= Cout) break;
            const half8 w = *reinterpret_cast<const half8*>(W + (size_t)o * C + c8);
            half8 wn;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float wf = (float)w[k];
                wn[k] = (f16)(wf * sh_a[c8 + k]);
                acc[r] = fmaf(wf, sh_b[c8 + k], acc[r]);
            }
            *reinterpret_cast<half8*>(Wn + ((size_t)n * Cout + o) * C + c8) = wn;
        }
    }
#pragma unroll
    for (int r = 0; r < GNF_ROWS; ++r) {
        float v = acc[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) sh_red[wv][r] = v;
    }
    __syncthreads();
    if (t < GNF_ROWS && o0 + t < Cout)
        tn[(size_t)n * Cout + o0 + t] = ((sh_red[0][t] + sh_red[1][t]) + (sh_red[2][t] + sh_red[3][t])) + (bias ? (float)bias[o0 + t] : 0.f);
}

// LayerNorm: one wavefront per row, the row lives in registers (C <= 1280 -> <= 3 vectors/lane),
// two-pass (mean, then centred variance) with wavefront xor-shuffles.
template <int NV>
__global__ void layernorm_kernel(const f16* __restrict__ X, int rows, int C, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, float eps, f16* __restrict__ Y) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const f16* x = X + (size_t)row * C;
    const int nvec = C >> 3;
    half8 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = lane + 64 * i;
        if (vi < nvec) {
            v[i] = *reinterpret_cast<const half8*>(x + vi * 8);
#pragma unroll
            for (int k = 0; k < 8; ++k) s += (float)v[i][k];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = lane + 64 * i;
        if (vi < nvec) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { const float d = (float)v[i][k] - mean; q += d * d; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q / (float)C + eps);
    f16* y = Y + (size_t)row * C;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = lane + 64 * i;
        if (vi < nvec) {
            half8 o;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                o[k] = (f16)(((float)v[i][k] - mean) * rstd * gamma[vi * 8 + k] + beta[vi * 8 + k]);
            *reinterpret_cast<half8*>(y + vi * 8) = o;
        }
    }
}

// LayerNorm statistics only (the normalisation itself is folded into the following GEMM's epilogue):
// same two-pass arithmetic as layernorm_kernel; a wavefront owns RPW consecutive rows and issues all
// their loads before the first reduction (read-only and latency bound at one row per wave).
template <int NV, int RPW>
__global__ void ln_stats_kernel(const f16* __restrict__ X, int rows, int C, float eps, float* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * RPW;
    if (row0 >= rows) return;
    const int nvec = C >> 3;
    half8 v[RPW][NV];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = row0 + r < rows ? row0 + r : rows - 1;
        const f16* x = X + (size_t)row * C;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int vi = lane + 64 * i;
            v[r][i] = (vi < nvec) ? *reinterpret_cast<const half8*>(x + vi * 8) : half8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int k = 0; k < 8; ++k) s += (float)v[r][i][k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s / (float)C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int vi = lane + 64 * i;
            if (vi < nvec) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { const float d = (float)v[r][i][k] - mean; q += d * d; }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        if (lane == 0 && row0 + r < rows) {
            stats[2 * (size_t)(row0 + r)] = mean;
            stats[2 * (size_t)(row0 + r) + 1] = rsqrtf(q / (float)C + eps);
        }
    }
}

// The same statistics with G lanes per row (C = 40 G: 320 / 640 / 1280 channels, five 16-byte chunks per lane) and
// 64 / G rows per load instruction, SETS row sets per wave: every lane carries data (one row per wave leaves 24 of 64
// lanes idle at C = 320), a row's chunks of one load instruction are G * 16 contiguous bytes, and the reduction crosses
// log2(G) lanes instead of six.  Two passes in registers as above (sum -> mean, then the centred squares).
template <int G, int SETS>
__global__ void ln_stats_g_kernel(const f16* __restrict__ X, int rows, float eps, float* __restrict__ stats) {
    constexpr int C = 40 * G, RW = 64 / G;
    const int lane = threadIdx.x & 63;
    const int g = lane / G, j = lane % G;
    const int row0 = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * (RW * SETS);
    if (row0 >= rows) return;
    half8 v[SETS][5];
#pragma unroll
    for (int t = 0; t < SETS; ++t) {
        int r = row0 + t * RW + g;
        r = r < rows ? r : rows - 1;
        const f16* x = X + (size_t)r * C + j * 8;
#pragma unroll
        for (int i = 0; i < 5; ++i) v[t][i] = *reinterpret_cast<const half8*>(x + i * G * 8);
    }
#pragma unroll
    for (int t = 0; t < SETS; ++t) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int k = 0; k < 8; ++k) s += (float)v[t][i][k];
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s / (float)C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int k = 0; k < 8; ++k) { const float d = (float)v[t][i][k] - mean; q += d * d; }
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) q += __shfl_xor(q, o);
        const int r = row0 + t * RW + g;
        if (j == 0 && r < rows) {
            stats[2 * (size_t)r] = mean;
            stats[2 * (size_t)r + 1] = rsqrtf(q / (float)C + eps);
        }
    }
}

}  // namespace

static int gn_pix(int HW) { int p = GN_PIX_MAX; while (p > 32 && HW / p < 16) p >>= 1; return p; }
int gn_stats_chunks(int HW) { const int p = gn_pix(HW); return (HW + p - 1) / p; }

static void gn_geometry(int C, int* R, int* threads) {
    const int cols = C / 8;
    int r = 320 / cols; if (r < 1) r = 1; if (r > 8) r = 8;
    *R = r; *threads = ((cols * r + 63) / 64) * 64;
}

hipError_t launch_gn_stats(const f16* X, const f16* X2, int N, int HW, int C, int C1, int G, double* partial, hipStream_t s) {
    if (C % 8 || C % G || C1 % 8 || G > 64) return hipErrorInvalidValue;
    int R, threads; gn_geometry(C, &R, &threads);
    if (threads > 1024) return hipErrorInvalidValue;
    const int chunks = gn_stats_chunks(HW);
    const size_t lds = (size_t)R * C * 2 * sizeof(float);
    hipLaunchKernelGGL(gn_stats_partial, dim3(chunks, N), dim3(threads), lds, s, X, X2 ? X2 : X, HW, C, C1, G, R, gn_pix(HW), partial);
    return hipGetLastError();
}

hipError_t launch_gn_merge_skip(const double* px, const double* pskip, int N, int Ns, int chunks, int G, int G1, int m, double* out, hipStream_t s) {
    if (G > 64 || G1 <= 0 || G1 >= G || m < 1 || (G - G1) * m != G || Ns < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gn_merge_skip_kernel, dim3(N), dim3(64), 0, s, px, pskip, Ns, chunks, G, G1, m, out);
    return hipGetLastError();
}

hipError_t launch_gn_blocks(const f16* X, int rows, int C, int row0, float* blocks, hipStream_t s) {
    if (rows % 64 || row0 % 64 || C % 16 || row0 > rows) return hipErrorInvalidValue;
    const int nblk = rows / 64, blk0 = row0 / 64;
    if (nblk == blk0) return hipSuccess;
    hipLaunchKernelGGL(gn_blocks_kernel, dim3((nblk - blk0 + 3) / 4), dim3(256), 0, s, X, nblk, C, blk0, blocks);
    return hipGetLastError();
}

hipError_t launch_gn_blocks_final(const float* blocks, int N, int HW, int C, int G, double* partial, hipStream_t s) {
    if (HW % 64 || C % G || (C / G) % 2 || G > 64 || 256 % G) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gn_blocks_final_kernel, dim3(N), dim3(256), 0, s, blocks, HW / 64, C, G, partial);
    return hipGetLastError();
}

hipError_t launch_gn_apply(const f16* X, const f16* X2, int N, int HW, int C, int C1, int G, float eps, const float* gamma,
                           const float* beta, const double* partial, int silu, f16* Y, hipStream_t s, int chunks) {
    if (C % 8 || C % G || C1 % 8 || G > 64) return hipErrorInvalidValue;
    int R, threads; gn_geometry(C, &R, &threads);
    if (threads < G) threads = 64;
    // enough blocks to fill the chip, few enough that the per-block statistics prologue amortises
    int ppb = 256;
    while (ppb > 8 && (long long)N * ((HW + ppb - 1) / ppb) < 2048) ppb >>= 1;
    hipLaunchKernelGGL(gn_apply_kernel, dim3((HW + ppb - 1) / ppb, N), dim3(threads), 0, s, X, X2 ? X2 : X, HW, C, C1, R,
                       ppb, partial, chunks > 0 ? chunks : gn_stats_chunks(HW), G, (double)HW * (double)(C / G), eps, gamma, beta, silu, Y);
    return hipGetLastError();
}

hipError_t launch_gn_fold(const double* partial, int N, int HW, int C, int G, float eps, const float* gamma, const float* beta,
                          const f16* W, const f16* bias, int Cout, f16* Wn, float* tn, hipStream_t s) {
    if (C % 8 || C % G || C > 2560 || G > 64 || 256 % G) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gn_fold_kernel, dim3((Cout + GNF_ROWS - 1) / GNF_ROWS, N), dim3(256), 0, s, partial, gn_stats_chunks(HW), G, C,
                       (double)HW * (double)(C / G), eps, gamma, beta, W, bias, Cout, Wn, tn);
    return hipGetLastError();
}

hipError_t launch_layernorm(const f16* X, int rows, int C, const float* gamma, const float* beta, float eps,
                            f16* Y, hipStream_t s) {
    if (C % 8 || C > 64 * 8 * 3) return hipErrorInvalidValue;
    const int wpb = 4;
    dim3 grid((rows + wpb - 1) / wpb), block(64 * wpb);
    const int nvec = C / 8;
    if (nvec <= 64) hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, s, X, rows, C, gamma, beta, eps, Y);
    else if (nvec <= 128) hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, s, X, rows, C, gamma, beta, eps, Y);
    else hipLaunchKernelGGL(layernorm_kernel<3>, grid, block, 0, s, X, rows, C, gamma, beta, eps, Y);
    return hipGetLastError();
}

hipError_t launch_ln_stats(const f16* X, int rows, int C, float eps, float* stats, hipStream_t s) {
    if (C % 8 || C > 64 * 8 * 3) return hipErrorInvalidValue;
    if (option(OPT_LN_STATS_G) && (C == 320 || C == 640 || C == 1280)) {        // the U-Net's token widths
        constexpr int SETS = 2, WPB = 4;
        const int rpw = (C == 320 ? 8 : C == 640 ? 4 : 2) * SETS;
        dim3 grid((rows + WPB * rpw - 1) / (WPB * rpw)), block(64 * WPB);
        if (C == 320) hipLaunchKernelGGL((ln_stats_g_kernel<8, SETS>), grid, block, 0, s, X, rows, eps, stats);
        else if (C == 640) hipLaunchKernelGGL((ln_stats_g_kernel<16, SETS>), grid, block, 0, s, X, rows, eps, stats);
        else hipLaunchKernelGGL((ln_stats_g_kernel<32, SETS>), grid, block, 0, s, X, rows, eps, stats);
        return hipGetLastError();
    }
    const int wpb = 4;
    const int nvec = C / 8;
    constexpr int RPW = 4;          // measured: 1 row/wave 145 us, 4 rows 104 us, 8 rows 98 us (C = 320, 655 360 rows)
    dim3 grid((rows + wpb * RPW - 1) / (wpb * RPW)), block(64 * wpb);
    if (nvec <= 64) hipLaunchKernelGGL((ln_stats_kernel<1, RPW>), grid, block, 0, s, X, rows, C, eps, stats);
    else if (nvec <= 128) hipLaunchKernelGGL((ln_stats_kernel<2, RPW>), grid, block, 0, s, X, rows, C, eps, stats);
    else hipLaunchKernelGGL((ln_stats_kernel<3, RPW>), grid, block, 0, s, X, rows, C, eps, stats);
    return hipGetLastError();
}

}  // namespace dm
