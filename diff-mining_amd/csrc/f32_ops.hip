// f32_ops.hip — the non-GEMM kernels of the fp32 U-Net path (reference: the DIFT featuriser's fp32 run, dift.py:191,197-199):
// flash attention on the fp32 matrix cores, GroupNorm (+ SiLU), LayerNorm, the time-step embedding, conv_in / conv_out and the
// layout kernels at the NCHW boundary.  All tensors fp32, NHWC inside.
#include "f32_kernels.h"
#include <math.h>

namespace dm32 {
namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

// 2^x for x <= 0 as ONE v_exp_f32 (r05): libm's exp2f wraps the instruction in a denormal rescue (compare, select, add, ldexp: six more
// VALU per call) for results below 2^-126 — far below the fp32 resolution of a softmax sum that is >= 1; -inf -> 0, NaN-free for finite input
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// ---- attention -----------------------------------------------------------------------------------------------------------------
// One block = 64 QT queries of one (sample, head): 4 waves x QT 16-query MFMA tiles (QT = 2; 1 for the VAE's head_dim 512).  Key tiles of KT keys are staged
// in LDS as fp32 rows: K rows of D + 2 floats, V rows of D + 4 (r05).  The fragment reads are 4-byte reads, which the LDS serves 32
// lanes at a time over 32 banks: a K read of lanes (c, g in {0, 1}) hits banks c * LDK + g, distinct for the 16 keys c iff LDK / 2 is
// odd (D + 2: 42 / 82 / 162 / 514); the r04 stride D + 4 put keys c and c + 8 on one bank — a two-way conflict on every K read, the
// 30.7 % of profiles/r04_final_dift_f32_pmc.json.  A V read (rows 4 g + r, column c) wants the two rows 4 floats * odd * 4 apart:
// D + 4 (rows r and r + 4 are 16 banks apart) is conflict-free and stays.
//   S^T[key][query] = K Q^T   : A = K rows  (lane (c, g): key c, d = 4 s + g),  B = Q^T (query c, d = 4 s + g), D/4 k steps — head_dim
//                                40 needs no padding on the k = 4 instruction
//   softmax over keys          : a lane holds keys 4 g + r of query c; the row maximum crosses the four g lanes by two shuffles
//   O^T[d][query] += V^T P^T   : B = P^T is the S^T accumulator AS IT LIES (register r <-> key 4 g + r, lane <-> query c) — P never
//                                leaves its lane —, A = V^T (lane (c, g): d = 16 dt + c, key 4 g + r)
// Online softmax (running maximum / sum, exp2 on pre-scaled scores), exact fp32 products and accumulation.
// head_dim <= 80: two LDS stages; tile t + 1 is written (from the registers its loads landed in) while tile t is being read, the loads
// of tile t + 2 fly under the MFMAs of tile t, ONE barrier per tile (r04: single stage, store + two barriers per tile).
template <int D, int KT, int QT>
__global__ __launch_bounds__(256) void attn32_kernel(AttnParams p) {
    constexpr int LDK = D + 2, LD = D + 4, DT = (D + 15) / 16, KS = D / 4, NKT = KT / 16;
    constexpr bool PF = D <= 80;           // register prefetch + two LDS stages
    constexpr int STG = KT * LDK + KT * LD + 16;      // floats per stage; + 16 of slack behind V: the last d tile of D = 40 reads columns 40..47
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;
    float* Vs = smem + KT * LDK;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, c = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * (64 * QT) + wid * (16 * QT);
    int kb = b;
    if (p.kv_slot) { kb = p.kv_slot[b]; kb = kb < 0 ? 0 : (kb >= p.n_slots ? p.n_slots - 1 : kb); }
    const float* Qb = p.Q + (long long)b * p.bsq + h * D;
    const float* Kb = p.K + (long long)kb * p.bsk + h * D;
    const float* Vb = p.V + (long long)kb * p.bsv + h * D;
    const float qscale = p.scale * 1.44269504088896340736f;

    float qf[QT][KS];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        int qi = q0 + qt * 16 + c; qi = qi < p.Tq ? qi : p.Tq - 1;
#pragma unroll
        for (int s = 0; s < KS; ++s) qf[qt][s] = Qb[(long long)qi * p.ldq + 4 * s + g] * qscale;
    }
    v4f o[DT][QT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) o[dt][qt] = v4f{0.f, 0.f, 0.f, 0.f};
    float mrun[QT], lrun[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) { mrun[qt] = -INFINITY; lrun[qt] = 0.f; }

    // K / V tiles go global -> registers -> LDS; where the registers allow (head_dim <= 80) the loads of tile t+1 are issued before the
    // MFMAs of tile t, so their latency is covered by the block's own matrix work instead of by other blocks' (r04: 98.1 -> 99.4 TFLOP/s at 4096 keys: the tile fetch was not what the kernel waits for)
    constexpr int NLD = (KT * (D / 4) + 255) / 256;
    v4f kreg[PF ? NLD : 1], vreg[PF ? NLD : 1];
    auto gload = [&](int k0) {
#pragma unroll
        for (int j = 0; j < (PF ? NLD : 0); ++j) {
            const int i = tid + 256 * j;
            const int r = i / (D / 4), c4 = i - r * (D / 4), key = k0 + r;
            kreg[j] = v4f{0.f, 0.f, 0.f, 0.f}; vreg[j] = v4f{0.f, 0.f, 0.f, 0.f};
            if (i < KT * (D / 4) && key < p.Tk) {
                kreg[j] = *reinterpret_cast<const v4f*>(Kb + (long long)key * p.ldk + 4 * c4);
                vreg[j] = *reinterpret_cast<const v4f*>(Vb + (long long)key * p.ldv + 4 * c4);
            }
        }
    };
    typedef float v2f __attribute__((ext_vector_type(2)));
    auto lstore = [&](int stage) {
#pragma unroll
        for (int j = 0; j < (PF ? NLD : 0); ++j) {
            const int i = tid + 256 * j;
            if (i < KT * (D / 4)) {
                const int r = i / (D / 4), c4 = i - r * (D / 4);
                float* kd = Ks + stage * STG + r * LDK + 4 * c4;           // K rows are 8-byte aligned only (LDK = D + 2)
                *reinterpret_cast<v2f*>(kd) = v2f{kreg[j][0], kreg[j][1]};
                *reinterpret_cast<v2f*>(kd + 2) = v2f{kreg[j][2], kreg[j][3]};
                *reinterpret_cast<v4f*>(Vs + stage * STG + r * LD + 4 * c4) = vreg[j];
            }
        }
    };
    if (PF) { gload(0); lstore(0); if (KT < p.Tk) gload(KT); __syncthreads(); }
    int stage = 0;
    for (int k0 = 0; k0 < p.Tk; k0 += KT) {
        if (!PF) {
            __syncthreads();
            for (int i = tid; i < KT * (D / 4); i += 256) {
                const int r = i / (D / 4), c4 = i - r * (D / 4), key = k0 + r;
                v4f kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
                if (key < p.Tk) {
                    kv = *reinterpret_cast<const v4f*>(Kb + (long long)key * p.ldk + 4 * c4);
                    vv = *reinterpret_cast<const v4f*>(Vb + (long long)key * p.ldv + 4 * c4);
                }
                *reinterpret_cast<v2f*>(Ks + r * LDK + 4 * c4) = v2f{kv[0], kv[1]};
                *reinterpret_cast<v2f*>(Ks + r * LDK + 4 * c4 + 2) = v2f{kv[2], kv[3]};
                *reinterpret_cast<v4f*>(Vs + r * LD + 4 * c4) = vv;
            }
            __syncthreads();
        } else if (k0 + KT < p.Tk) {
            lstore(stage ^ 1);                                  // tile t + 1 (its loads were issued one tile ago) -> the other stage
            if (k0 + 2 * KT < p.Tk) gload(k0 + 2 * KT);         // tile t + 2: in flight under this tile's MFMAs
        }
        const float* Kc = Ks + (PF ? stage * STG : 0);
        const float* Vc = Vs + (PF ? stage * STG : 0);
        v4f sacc[NKT][QT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) sacc[kt][qt] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                const float a = Kc[(kt * 16 + c) * LDK + 4 * s + g];
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) sacc[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, qf[qt][s], sacc[kt][qt], 0, 0, 0);
            }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = k0 + kt * 16 + 4 * g + r;
                    const float v = key < p.Tk ? sacc[kt][qt][r] : -INFINITY;
                    sacc[kt][qt][r] = v;
                    mx = fmaxf(mx, v);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mnew = fmaxf(mrun[qt], mx);          // finite: every key tile holds at least one real key
            const float alpha = fast_exp2(mrun[qt] - mnew);
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = fast_exp2(sacc[kt][qt][r] - mnew);
                    sacc[kt][qt][r] = pv;
                    sum += pv;
                }
            lrun[qt] = lrun[qt] * alpha + sum;
            mrun[qt] = mnew;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) o[dt][qt] *= alpha;
        }
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const float a = Vc[(kt * 16 + 4 * g + r) * LD + dt * 16 + c];
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) o[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, sacc[kt][qt][r], o[dt][qt], 0, 0, 0);
                }
        if (PF) { __syncthreads(); stage ^= 1; }                // tile t + 1 is complete in the other stage; everybody has left this one
    }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float l = lrun[qt];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const float inv = 1.0f / l;
        const int qi = q0 + qt * 16 + c;
        if (qi >= p.Tq) continue;
        float* orow = p.O + (long long)b * p.bso + (long long)qi * p.ldo + h * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d = dt * 16 + 4 * g;
            if (d < D) *reinterpret_cast<v4f*>(orow + d) = o[dt][qt] * inv;
        }
    }
}

template <int D, int KT, int QT>
hipError_t launch_attn_t(const AttnParams& p, hipStream_t s) {
    const size_t lds = (size_t)(KT * (D + 2) + KT * (D + 4) + 16) * (D <= 80 ? 2 : 1) * sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn32_kernel<D, KT, QT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((attn32_kernel<D, KT, QT>), dim3((p.Tq + 64 * QT - 1) / (64 * QT), p.heads, p.B), dim3(256), lds, s, p);
    return hipGetLastError();
}

// ---- GroupNorm -----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn32_stats_kernel(const float* X, const float* X2, int HW, int C, int C1, int G, float eps, float* stats) {
    const int n = blockIdx.x / G, g = blockIdx.x - n * G, cpg = C / G, ch0 = g * cpg, C2 = C - C1;
    double s = 0.0, ss = 0.0;
    const long long total = (long long)HW * cpg;
    for (long long e = threadIdx.x; e < total; e += 256) {
        const long long px = e / cpg;
        const int ch = ch0 + (int)(e - px * cpg);
        const float v = ch < C1 ? X[((long long)n * HW + px) * C1 + ch] : X2[((long long)n * HW + px) * C2 + (ch - C1)];
        s += (double)v; ss += (double)v * (double)v;
    }
    __shared__ double rs[256], rq[256];
    rs[threadIdx.x] = s; rq[threadIdx.x] = ss;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) { rs[threadIdx.x] += rs[threadIdx.x + w]; rq[threadIdx.x] += rq[threadIdx.x + w]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double mean = rs[0] / (double)total;
        double var = rq[0] / (double)total - mean * mean;
        var = var > 0.0 ? var : 0.0;
        stats[2 * blockIdx.x] = (float)mean;
        stats[2 * blockIdx.x + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

__global__ void gn32_apply_kernel(const float* X, const float* X2, long long rows, int HW, int C, int C1, int G, const float* gamma,
                                  const float* beta, const float* stats, int silu, float* Y) {
    const int c4n = C / 4, cpg = C / G, C2 = C - C1;
    const long long total = rows * c4n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / c4n;
        const int ch = 4 * (int)(i - row * c4n);
        const int n = (int)(row / HW);
        const v4f x = ch < C1 ? *reinterpret_cast<const v4f*>(X + row * C1 + ch) : *reinterpret_cast<const v4f*>(X2 + row * C2 + (ch - C1));
        v4f y;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int g = (ch + j) / cpg;
            const float mean = stats[2 * (n * G + g)], rstd = stats[2 * (n * G + g) + 1];
            float v = (x[j] - mean) * rstd * gamma[ch + j] + beta[ch + j];
            if (silu) v = v / (1.0f + expf(-v));
            y[j] = v;
        }
        *reinterpret_cast<v4f*>(Y + row * C + ch) = y;
    }
}

// ---- LayerNorm: one wave per row, two-pass statistics ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ln32_kernel(const float* X, int rows, int C, const float* gamma, const float* beta, float eps, float* Y) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* x = X + (long long)row * C;
    float s = 0.f;
    for (int i = lane; i < C; i += 64) s += x[i];
    for (int w = 32; w > 0; w >>= 1) s += __shfl_xor(s, w);
    const float mean = s / (float)C;
    float q = 0.f;
    for (int i = lane; i < C; i += 64) { const float d = x[i] - mean; q += d * d; }
    for (int w = 32; w > 0; w >>= 1) q += __shfl_xor(q, w);
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
    float* y = Y + (long long)row * C;
    for (int i = lane; i < C; i += 64) y[i] = (x[i] - mean) * rstd * gamma[i] + beta[i];
}

// ---- CLIP text tower glue in fp32 (`pipe.encode_prompt` of the featuriser's fp32 pipeline, dift.py:197-199, 222-226) ---------------------
// token_embedding[ids] + position_embedding
__global__ void clip32_embed_kernel(const int32_t* ids, const float* tok, const float* pos, int rows, int T, int C, int vocab, float* out) {
    const int row = blockIdx.x;
    if (row >= rows) return;
    int id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const float* a = tok + (size_t)id * C;
    const float* b = pos + (size_t)(row % T) * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) out[(size_t)row * C + c] = a[c] + b[c];
}

// causal self-attention over <= 80 tokens, heads of 64: one block per (prompt, head), one thread per query; qkv [n*T][3*heads*64];
// q is scaled by head_dim^-0.5 = 1/8 AFTER its projection (bias included), as CLIPAttention does (a power of two: exact)
constexpr int CLD = 64, CLT = 80;
__global__ __launch_bounds__(128) void clip32_attn_kernel(const float* qkv, int T, int heads, float* out) {
    __shared__ float Ks[CLT][CLD + 1], Vs[CLT][CLD + 1];
    const int h = blockIdx.x, n = blockIdx.y;
    const int C = heads * CLD;
    const float* base = qkv + (size_t)n * T * 3 * C + h * CLD;
    for (int i = threadIdx.x; i < T * CLD; i += blockDim.x) {
        const int t = i / CLD, d = i - t * CLD;
        Ks[t][d] = base[(size_t)t * 3 * C + C + d];
        Vs[t][d] = base[(size_t)t * 3 * C + 2 * C + d];
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t >= T) return;
    float q[CLD];
#pragma unroll
    for (int d = 0; d < CLD; ++d) q[d] = base[(size_t)t * 3 * C + d] * 0.125f;
    float sc[CLT];
    float m = -INFINITY;
    for (int k = 0; k <= t; ++k) {                 // causal: keys 0..t
        float a = 0.f;
#pragma unroll
        for (int d = 0; d < CLD; ++d) a = fmaf(q[d], Ks[k][d], a);
        sc[k] = a;
        m = fmaxf(m, a);
    }
    float l = 0.f;
    for (int k = 0; k <= t; ++k) { sc[k] = expf(sc[k] - m); l += sc[k]; }
    float o[CLD];
#pragma unroll
    for (int d = 0; d < CLD; ++d) o[d] = 0.f;
    for (int k = 0; k <= t; ++k) {
        const float pk = sc[k] / l;                // softmax output (fp32), then P V
#pragma unroll
        for (int d = 0; d < CLD; ++d) o[d] = fmaf(pk, Vs[k][d], o[d]);
    }
    float* dst = out + ((size_t)n * T + t) * C + h * CLD;
#pragma unroll
    for (int d = 0; d < CLD; ++d) dst[d] = o[d];
}

__global__ void quick_gelu32_kernel(float* x, long long n) {          // y * sigmoid(1.702 y)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float y = x[i];
        x[i] = y / (1.0f + expf(-1.702f * y));
    }
}

__global__ void silu32_kernel(const float* in, float* out, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = in[i];
        out[i] = v / (1.0f + expf(-v));
    }
}

__global__ void temb32_kernel(const int64_t* t, int B, int dim, float* out) {
    const int half = dim / 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * half) return;
    const int b = i / half, j = i - b * half;
    const float freq = expf(-logf(10000.0f) * (float)j / (float)half);
    const float a = (float)t[b] * freq;
    out[(long long)b * dim + j] = cosf(a);
    out[(long long)b * dim + half + j] = sinf(a);
}

__global__ void conv_in32_kernel(const float* x, const float* w, const float* bias, int B, int Cin, int H, int W, int Cout, float* y) {
    const long long total = (long long)B * H * W * Cout;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int co = (int)(i % Cout);
        const long long m = i / Cout;
        const int ox = (int)(m % W), oy = (int)((m / W) % H), n = (int)(m / ((long long)W * H));
        float acc = 0.f;
        for (int ci = 0; ci < Cin; ++ci)
            for (int dy = 0; dy < 3; ++dy) {
                const int iy = oy + dy - 1;
                if (iy < 0 || iy >= H) continue;
                for (int dx = 0; dx < 3; ++dx) {
                    const int ix = ox + dx - 1;
                    if (ix < 0 || ix >= W) continue;
                    acc += x[(((long long)n * Cin + ci) * H + iy) * W + ix] * w[((ci * 3 + dy) * 3 + dx) * Cout + co];
                }
            }
        y[i] = acc + bias[co];
    }
}

// one wave per output pixel: lanes over channels, Cout <= 8 accumulators, shuffle reduction
__global__ __launch_bounds__(256) void conv_out32_kernel(const float* x, const float* w, const float* bias, int B, int H, int W, int C, int Cout, float* y) {
    const long long m = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (m >= (long long)B * H * W) return;
    const int ox = (int)(m % W), oy = (int)((m / W) % H), n = (int)(m / ((long long)W * H));
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int tap = 0; tap < 9; ++tap) {
        const int iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        const float* xr = x + (((long long)n * H + iy) * W + ix) * C;
        for (int ch = lane; ch < C; ch += 64) {
            const float v = xr[ch];
            for (int co = 0; co < Cout; ++co) acc[co] += v * w[((long long)co * 9 + tap) * C + ch];
        }
    }
    for (int co = 0; co < Cout; ++co) {
        float v = acc[co];
        for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s);
        if (lane == 0) y[(((long long)n * Cout + co) * H + oy) * W + ox] = v + bias[co];
    }
}

__global__ void nhwc_to_nchw32_kernel(const float* X, int N, int HW, int C, float* Y) {
    const long long total = (long long)N * HW * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int px = (int)(i % HW);
        const long long nc = i / HW;
        const int ch = (int)(nc % C);
        const long long n = nc / C;
        Y[i] = X[(n * HW + px) * C + ch];
    }
}

__global__ void ensemble_mean32_kernel(const float* X, int groups, int ens, int HW, int C, float* Y) {
    const long long total = (long long)groups * HW * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int px = (int)(i % HW);
        const long long gc = i / HW;
        const int ch = (int)(gc % C);
        const long long gi = gc / C;
        float s = 0.f;
        for (int k = 0; k < ens; ++k) s += X[((gi * ens + k) * HW + px) * C + ch];
        Y[i] = s / (float)ens;
    }
}

// quant_conv (1x1, 8 -> 8) on Hm [B*HW][8] + DiagonalGaussianDistribution.sample() with the draw injected, times scaling_factor,
// all fp32: latent = (mean + exp(0.5 clamp(logvar, -30, 20)) * noise) * scaling; noise == nullptr -> the mode.  `draws` samples per
// image: output sample b * draws + d reads the moments of image b.  NCHW outputs.
__global__ void posterior32_kernel(const float* Hm, const float* qw, const float* qb, const float* noise, int B, int draws, int HW,
                                   float scaling, float* latent, float* moments) {
    const long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= (long long)B * draws * HW) return;
    const int bo = (int)(pix / HW), rem = (int)(pix - (long long)bo * HW), b = bo / draws;
    const float* hv = Hm + ((size_t)b * HW + rem) * 8;
    float m[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += qw[o * 8 + i] * hv[i];
        m[o] = acc + qb[o];
        if (moments && bo == b * draws) moments[((size_t)b * 8 + o) * HW + rem] = m[o];
    }
    if (!latent) return;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float v = m[c];
        if (noise) {
            float lv = m[4 + c];
            lv = lv < -30.f ? -30.f : (lv > 20.f ? 20.f : lv);
            v += expf(0.5f * lv) * noise[((size_t)bo * 4 + c) * HW + rem];
        }
        latent[((size_t)bo * 4 + c) * HW + rem] = v * scaling;
    }
}

// scheduler.add_noise in fp32 (compute.py:99 / dift.py:190): noisy[b] = sqrt(acp[t[b]]) * x[xi[b]] + sqrt(1 - acp[t[b]]) * eps[b]
__global__ void add_noise32_kernel(const float* x, const int32_t* x_index, const float* eps, const int64_t* t, const float* sa, const float* sb,
                                   int B, long long per, float* out) {
    const long long total = (long long)B * per;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / per);
        const long long r = i - (long long)b * per;
        const int xi = x_index ? x_index[b] : b;
        long long tt = t[b]; tt = tt < 0 ? 0 : (tt > 999 ? 999 : tt);
        out[i] = sa[tt] * x[(long long)xi * per + r] + sb[tt] * eps[i];
    }
}
// F.mse_loss(pred, eps, reduction='none') (compute.py:101)
__global__ void sqerr32_kernel(const float* pred, const float* eps, long long n, float* out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float d = pred[i] - eps[i];
        out[i] = d * d;
    }
}

inline unsigned grid_for(long long n, int block = 256) {
    long long g = (n + block - 1) / block;
    return (unsigned)(g < 1 ? 1 : (g > 65536 * 16 ? 65536 * 16 : g));
}

}  // namespace

hipError_t launch_attention(const AttnParams& p, hipStream_t s) {
    if (p.B <= 0 || p.Tq <= 0 || p.Tk <= 0 || (p.ldq | p.ldk | p.ldv | p.ldo) % 4 != 0) return hipErrorInvalidValue;
    switch (p.D) {
        case 40: return launch_attn_t<40, 64, 2>(p, s);
        case 80: return launch_attn_t<80, 64, 2>(p, s);
        case 160: return launch_attn_t<160, 64, 2>(p, s);
        case 512: return launch_attn_t<512, 16, 1>(p, s);       // AutoencoderKL mid-block attention (one head)
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_gn_stats(const float* X, const float* X2, int N, int HW, int C, int C1, int G, float eps, float* stats, hipStream_t s) {
    if (C % G != 0 || (C1 < C && !X2)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gn32_stats_kernel, dim3(N * G), dim3(256), 0, s, X, X2, HW, C, C1, G, eps, stats);
    return hipGetLastError();
}
hipError_t launch_gn_apply(const float* X, const float* X2, int N, int HW, int C, int C1, int G, const float* gamma, const float* beta,
                           const float* stats, int silu, float* Y, hipStream_t s) {
    if (C % 4 != 0 || C1 % 4 != 0) return hipErrorInvalidValue;
    const long long rows = (long long)N * HW;
    hipLaunchKernelGGL(gn32_apply_kernel, dim3(grid_for(rows * (C / 4))), dim3(256), 0, s, X, X2, rows, HW, C, C1, G, gamma, beta, stats, silu, Y);
    return hipGetLastError();
}
hipError_t launch_layernorm(const float* X, int rows, int C, const float* gamma, const float* beta, float eps, float* Y, hipStream_t s) {
    hipLaunchKernelGGL(ln32_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, X, rows, C, gamma, beta, eps, Y);
    return hipGetLastError();
}
hipError_t launch_clip_embed(const int32_t* ids, const float* tok, const float* pos, int rows, int T, int C, int vocab, float* out, hipStream_t s) {
    if (rows <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(clip32_embed_kernel, dim3(rows), dim3(256), 0, s, ids, tok, pos, rows, T, C, vocab, out);
    return hipGetLastError();
}
hipError_t launch_clip_attention(const float* qkv, int n, int T, int heads, float* out, hipStream_t s) {
    if (T <= 0 || T > CLT || n <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(clip32_attn_kernel, dim3(heads, n), dim3(128), 0, s, qkv, T, heads, out);
    return hipGetLastError();
}
hipError_t launch_quick_gelu(float* x, long long n, hipStream_t s) {
    hipLaunchKernelGGL(quick_gelu32_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, n);
    return hipGetLastError();
}
hipError_t launch_silu(const float* in, float* out, long long n, hipStream_t s) {
    hipLaunchKernelGGL(silu32_kernel, dim3(grid_for(n)), dim3(256), 0, s, in, out, n);
    return hipGetLastError();
}
hipError_t launch_timestep_embed(const int64_t* t, int B, int dim, float* out, hipStream_t s) {
    hipLaunchKernelGGL(temb32_kernel, dim3((B * (dim / 2) + 255) / 256), dim3(256), 0, s, t, B, dim, out);
    return hipGetLastError();
}
hipError_t launch_conv_in(const float* x, const float* w, const float* bias, int B, int Cin, int H, int W, int Cout, float* y, hipStream_t s) {
    hipLaunchKernelGGL(conv_in32_kernel, dim3(grid_for((long long)B * H * W * Cout)), dim3(256), 0, s, x, w, bias, B, Cin, H, W, Cout, y);
    return hipGetLastError();
}
hipError_t launch_conv_out(const float* x, const float* w, const float* bias, int B, int H, int W, int C, int Cout, float* y, hipStream_t s) {
    if (Cout > 8) return hipErrorInvalidValue;
    const long long px = (long long)B * H * W;
    hipLaunchKernelGGL(conv_out32_kernel, dim3((unsigned)((px + 3) / 4)), dim3(256), 0, s, x, w, bias, B, H, W, C, Cout, y);
    return hipGetLastError();
}
hipError_t launch_posterior(const float* Hm, const float* qw, const float* qb, const float* noise, int B, int draws, int HW, float scaling,
                            float* latent, float* moments, hipStream_t s) {
    const long long total = (long long)B * draws * HW;
    hipLaunchKernelGGL(posterior32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, Hm, qw, qb, noise, B, draws, HW, scaling, latent, moments);
    return hipGetLastError();
}
hipError_t launch_add_noise(const float* x, const int32_t* x_index, const float* eps, const int64_t* t, const float* sa, const float* sb,
                            int B, long long per, float* out, hipStream_t s) {
    hipLaunchKernelGGL(add_noise32_kernel, dim3(grid_for((long long)B * per)), dim3(256), 0, s, x, x_index, eps, t, sa, sb, B, per, out);
    return hipGetLastError();
}
hipError_t launch_sqerr(const float* pred, const float* eps, long long n, float* out, hipStream_t s) {
    hipLaunchKernelGGL(sqerr32_kernel, dim3(grid_for(n)), dim3(256), 0, s, pred, eps, n, out);
    return hipGetLastError();
}
hipError_t launch_nhwc_to_nchw(const float* X, int N, int HW, int C, float* Y, hipStream_t s) {
    hipLaunchKernelGGL(nhwc_to_nchw32_kernel, dim3(grid_for((long long)N * HW * C)), dim3(256), 0, s, X, N, HW, C, Y);
    return hipGetLastError();
}
hipError_t launch_ensemble_mean(const float* X, int groups, int ens, int HW, int C, float* Y, hipStream_t s) {
    hipLaunchKernelGGL(ensemble_mean32_kernel, dim3(grid_for((long long)groups * HW * C)), dim3(256), 0, s, X, groups, ens, HW, C, Y);
    return hipGetLastError();
}

}  // namespace dm32
