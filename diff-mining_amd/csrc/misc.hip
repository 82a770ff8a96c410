// misc.hip — the small HBM-bound kernels around the U-Net of diff-mining's scoring path:
//   K8  timestep-embedding row gather (sinusoid table precomputed on the host),
//   K9  scheduler.add_noise (compute.py:99) fused into conv_in (4 -> 320, 3x3),
//   K10 conv_out (320 -> 4, 3x3) fused with the per-element eps-MSE of compute.py:101
//       (wavefront-shuffle reduction over the 2880-long dot products),
//   K11 DIFT ensemble mean (dift.py:231) + NHWC->NCHW feature export,
//   and the consumers' typicality reductions (cluster.py:125-137, xray/compute.py:210-218).
#include "dm_kernels.h"

namespace dm {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

namespace {

__global__ void time_gather_kernel(const f16* __restrict__ table, const int64_t* __restrict__ t, int B, int dim,
                                   f16* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * dim) return;
    const int b = i / dim, d = i - b * dim;
    long long tt = t[b];
    tt = tt < 0 ? 0 : (tt > 999 ? 999 : tt);
    out[i] = table[tt * dim + d];
}

__global__ void silu_kernel(const f16* __restrict__ in, f16* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = (float)in[i];
    out[i] = (f16)(x / (1.0f + __expf(-x)));
}

// One thread = one pixel: forms the 36 (zero padded) noisy inputs of its 3x3x4 patch and writes one
// 128-byte im2col row.  Two dtype flows of `scheduler.add_noise` (compute.py:99):
//   T = float (DM_F32, what the reference's autocast run actually does: encode_vae returns fp32 because
//       exp() is promoted, so randn_like / add_noise are fp32 and the table stays fp32):
//       noisy = fp16( fp32(sa*x) + fp32(sb*eps) ),  sa = sqrtf(acp[t]), sb = sqrtf(1 - acp[t]);  the single
//       rounding to fp16 is autocast's cast of conv_in's input;
//   T = f16 (DM_F16, an fp16 latent handed to an fp16 scheduler: table cast to fp16 FIRST):
//       noisy = fp16(fp16(sa * x) + fp16(sb * eps)),  sa = fp16(sqrt(fp16 acp[t])), sb likewise.
template <typename T>
__global__ void im2col_in_kernel(const T* __restrict__ x, const int32_t* __restrict__ x_index,
                                 const T* __restrict__ eps, const int64_t* __restrict__ t,
                                 const T* __restrict__ sa_tab, const T* __restrict__ sb_tab, int B, int H, int W,
                                 f16* __restrict__ out) {
    const long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int HW = H * W;
    if (pix >= (long long)B * HW) return;
    const int b = (int)(pix / HW);
    const int rem = (int)(pix - (long long)b * HW);
    const int oh = rem / W, ow = rem - oh * W;
    const bool noise = (sa_tab != nullptr);
    T sa = (T)1.0f, sb = (T)0.0f;
    if (noise) {
        long long tt = t[b];
        tt = tt < 0 ? 0 : (tt > 999 ? 999 : tt);
        sa = sa_tab[tt]; sb = sb_tab[tt];
    }
    const int xb = x_index ? x_index[b] : b;
    f16 row[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) row[k] = (f16)0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ih = oh + dy - 1, iw = ow + dx - 1;
                T v = (T)0.f;
                if (ih >= 0 && ih < H && iw >= 0 && iw < W) {
                    v = x[((size_t)xb * 4 + c) * HW + ih * W + iw];
                    if (noise) {
                        const T e = eps[((size_t)b * 4 + c) * HW + ih * W + iw];
                        const T p1 = sa * v;            // multiply, rounded in T (-ffp-contract=off: no FMA)
                        const T p2 = sb * e;
                        v = p1 + p2;                    // add, rounded in T
                    }
                }
                row[c * 9 + dy * 3 + dx] = (f16)v;
            }
    half8* dst = reinterpret_cast<half8*>(out + pix * 64);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        half8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = row[k * 8 + j];
        dst[k] = o;
    }
}

// conv_out + eps-MSE.  8 lanes cooperate on one pixel: lane j of the group walks the 16-byte
// channel chunks j, j+8, ... of the 9 taps; the four 2880-long dot products are then reduced with
// xor-shuffles inside the wavefront.  Weights [4][9*C0] are staged in LDS once per block, and a block walks
// CONV_OUT_GROUPS groups of 32 pixels (one group per block spent as long fetching the 23 KB of weights as computing).
// The products go through v_dot2_f32_f16 (two fp16 x fp16 products, exact, + an fp32 accumulator per instruction):
// 16 VALU instructions per 16-byte chunk instead of 40 converts + 32 FMAs — the layer was VALU-bound at 7x its
// HBM time (0.63 ms for one 419 MB read at the bench batch; 0.44 ms now — four pixels per weight chunk read measured
// slower again, 0.50 ms, and so did the same layer on the matrix cores — four channels as rows 0..3 of a 16x16x32
// MFMA, sixteen pixels as columns, operands straight from NHWC: 0.47 ms, 99.7 % of the outputs bit-equal.  The loop is
// bound by the nine-fold 16-byte gather through L1 / L2 (lines of 128 bytes used 64 at a time by more waves than the L1
// holds), not by arithmetic or LDS; the fix would be input rows staged once in LDS and walked by all nine taps).
constexpr int CONV_OUT_GROUPS = 8;
typedef _Float16 half2x __attribute__((ext_vector_type(2)));
template <typename TE>
__global__ __launch_bounds__(256)
void conv_out_kernel(const f16* __restrict__ Xn, const f16* __restrict__ w, const f16* __restrict__ bias,
                     const TE* __restrict__ eps, int B, int H, int W, int C0, float* __restrict__ loss,
                     f16* __restrict__ pred, int eps_rows, int out_group, int out_stride, int out_off) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f16* ws = reinterpret_cast<f16*>(smem);
    const int K = 9 * C0;
    for (int i = threadIdx.x * 8; i < 4 * K; i += blockDim.x * 8)
        *reinterpret_cast<half8*>(ws + i) = *reinterpret_cast<const half8*>(w + i);
    __syncthreads();
    const int HW = H * W;
    const long long npix = (long long)B * HW;
    const int sub = threadIdx.x & 7;
    const int nch = C0 >> 3;
    for (int grp = 0; grp < CONV_OUT_GROUPS; ++grp) {
        const long long pix = ((long long)blockIdx.x * CONV_OUT_GROUPS + grp) * (blockDim.x >> 3) + (threadIdx.x >> 3);
        const bool valid = pix < npix;
        const long long pp = valid ? pix : 0;
        const int b = (int)(pp / HW);
        const int rem = (int)(pp - (long long)b * HW);
        const int oh = rem / W, ow = rem - oh * W;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - dy * 3;
            const int ih = oh + dy - 1, iw = ow + dx - 1;
            if (ih < 0 || ih >= H || iw < 0 || iw >= W) continue;
            const f16* src = Xn + (((size_t)b * H + ih) * W + iw) * C0;
            for (int ch = sub; ch < nch; ch += 8) {
                const half8 v = *reinterpret_cast<const half8*>(src + ch * 8);
                const int kb = tap * C0 + ch * 8;
                const half8 w0 = *reinterpret_cast<const half8*>(ws + kb);
                const half8 w1 = *reinterpret_cast<const half8*>(ws + K + kb);
                const half8 w2 = *reinterpret_cast<const half8*>(ws + 2 * K + kb);
                const half8 w3 = *reinterpret_cast<const half8*>(ws + 3 * K + kb);
#pragma unroll
                for (int k = 0; k < 8; k += 2) {
                    const half2x f = half2x{v[k], v[k + 1]};
                    a0 = __builtin_amdgcn_fdot2(f, half2x{w0[k], w0[k + 1]}, a0, false);
                    a1 = __builtin_amdgcn_fdot2(f, half2x{w1[k], w1[k + 1]}, a1, false);
                    a2 = __builtin_amdgcn_fdot2(f, half2x{w2[k], w2[k + 1]}, a2, false);
                    a3 = __builtin_amdgcn_fdot2(f, half2x{w3[k], w3[k + 1]}, a3, false);
                }
            }
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            a0 += __shfl_xor(a0, o); a1 += __shfl_xor(a1, o); a2 += __shfl_xor(a2, o); a3 += __shfl_xor(a3, o);
        }
        if (valid && sub < 4) {
            const float a = sub == 0 ? a0 : (sub == 1 ? a1 : (sub == 2 ? a2 : a3));
            const f16 pr = (f16)(a + (float)bias[sub]);
            const int orow = (b / out_group) * out_stride + out_off + b % out_group;
            const size_t oidx = ((size_t)orow * 4 + sub) * HW + rem;
            if (eps) {
                const float d = (float)pr - (float)eps[((size_t)(b % eps_rows) * 4 + sub) * HW + rem];
                loss[oidx] = d * d;
            }
            if (pred) pred[oidx] = pr;
        }
    }
}

__global__ void nhwc_to_nchw_kernel(const f16* __restrict__ X, int HW, int C, f16* __restrict__ Y) {
    __shared__ f16 tile[32][33];
    const int n = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        tile[r][tx] = (p < HW && c < C) ? X[((size_t)n * HW + p) * C + c] : (f16)0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        if (p < HW && c < C) Y[((size_t)n * C + c) * HW + p] = tile[tx][r];
    }
}

__global__ void ensemble_mean_kernel(const f16* __restrict__ X, int ens, int HW, int C, float* __restrict__ Y) {
    __shared__ float tile[32][33];
    const int g = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        float s = 0.f;
        if (p < HW && c < C)
            for (int e = 0; e < ens; ++e) s += (float)X[(((size_t)g * ens + e) * HW + p) * C + c];
        tile[r][tx] = s / (float)ens;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        if (p < HW && c < C) Y[((size_t)g * C + c) * HW + p] = tile[tx][r];
    }
}

// map[p] = mean_n( mean_c L[n, last, c, p] - mean_c L[n, 0, c, p] ); deterministic (fixed order).
// blockIdx.y = image; sample (image i, draw n, prompt k) is row i*s_img + n*s_draw + k*s_cond of L [rows][4][HW]
// (the reference's [N, n_cond] grid: s_draw = n_cond, s_cond = 1; dm_score_conds' cond-major rows: s_cond = n_img*N,
// s_img = N, s_draw = 1).
template <typename T>
__global__ void typicality_map_kernel(const T* __restrict__ L, int n_draws, int n_cond, int HW, float* __restrict__ map,
                                      long long s_img, long long s_draw, long long s_cond) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    L += (size_t)blockIdx.y * s_img * 4 * HW;
    map += (size_t)blockIdx.y * HW;
    float acc = 0.f;
    for (int n = 0; n < n_draws; ++n) {
        const T* l0 = L + ((size_t)n * s_draw) * 4 * HW;
        const T* l1 = L + ((size_t)n * s_draw + (size_t)(n_cond - 1) * s_cond) * 4 * HW;
        float m0 = 0.f, m1 = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) { m0 += (float)l0[(size_t)c * HW + p]; m1 += (float)l1[(size_t)c * HW + p]; }
        acc += (m1 - m0) * 0.25f;
    }
    map[p] = acc / (float)n_draws;
}

// Consumers' image-space reduction (cluster.py:125-137 `load_typicality`, xray/compute.py:210-218):
//   U = bilinear(D, (H, W), align_corners=False)  of the latent map D = mean_N(mean_C L_null - mean_C L_c)
//   out = AvgPool2d((kx, ky), stride 1)(U)                     (linear, so it commutes with the means)
// as two separable box-sum passes; pass 1 evaluates the bilinear sample on the fly.
__device__ __forceinline__ float bilinear_at(const float* __restrict__ D, int h, int w, float sy, float sx, int y, int x) {
    float fy = sy * ((float)y + 0.5f) - 0.5f; fy = fy < 0.f ? 0.f : fy;
    float fx = sx * ((float)x + 0.5f) - 0.5f; fx = fx < 0.f ? 0.f : fx;
    int y0 = (int)fy, x0 = (int)fx;
    y0 = y0 < h - 1 ? y0 : h - 1; x0 = x0 < w - 1 ? x0 : w - 1;
    const int y1 = y0 + 1 < h ? y0 + 1 : h - 1, x1 = x0 + 1 < w ? x0 + 1 : w - 1;
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float top = D[y0 * w + x0] * (1.f - lx) + D[y0 * w + x1] * lx;
    const float bot = D[y1 * w + x0] * (1.f - lx) + D[y1 * w + x1] * lx;
    return top * (1.f - ly) + bot * ly;
}

// tmp[y][x] = sum_{dx < ky} U[y][x + dx],  y in [0,H), x in [0, W - ky + 1)
__global__ void upsample_rowsum_kernel(const float* __restrict__ D, int h, int w, int H, int W, int ky,
                                       float* __restrict__ tmp) {
    const int OWd = W - ky + 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * OWd) return;
    const int y = i / OWd, x = i - y * OWd;
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    float acc = 0.f;
    for (int dx = 0; dx < ky; ++dx) acc += bilinear_at(D, h, w, sy, sx, y, x + dx);
    tmp[i] = acc;
}

// out[y][x] = (1/(kx*ky)) sum_{dy < kx} tmp[y + dy][x]
__global__ void colsum_kernel(const float* __restrict__ tmp, int H, int OWd, int kx, float inv, float* __restrict__ out) {
    const int OHd = H - kx + 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= OHd * OWd) return;
    const int y = i / OWd, x = i - y * OWd;
    float acc = 0.f;
    for (int dy = 0; dy < kx; ++dy) acc += tmp[(y + dy) * OWd + x];
    out[i] = acc * inv;
}

// Consumers' normalisations of an image-space typicality map (fp32, numpy semantics restated in fp32 IEEE arithmetic):
//   mode 1  `normalize(dm)` of cluster.py:32-47 as `load_typicality_norm` (cluster.py:112-123) calls it: negatives / |min|,
//           positives / max, then (dm + 1) / 2
//   mode 2  `dm / np.max(np.abs(dm))`: `d_compute` (utils.py:122-134) and utils.normalize (utils.py:14-20)
//   mode 3  positive_only: max(dm, 0) / max(max(dm, 0))  (cluster.py:39-42, utils.py:16-19)
//   mode 4  positive_only == 'split' (cluster.py:34-36): d = dm / |max(dm)|; out = clip(d, 0, 1), out2 = -clip(d, -1, 0)
// pass 1: min / max / max|.| of the map (order-independent, so any reduction tree gives numpy's value); pass 2: elementwise.
__global__ void map_minmax_kernel(const float* __restrict__ x, long long n, float* __restrict__ mm) {
    __shared__ float smin[1024], smax[1024];
    float lo = INFINITY, hi = -INFINITY;
    for (long long i = threadIdx.x; i < n; i += blockDim.x) { const float v = x[i]; lo = fminf(lo, v); hi = fmaxf(hi, v); }
    smin[threadIdx.x] = lo; smax[threadIdx.x] = hi;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            smin[threadIdx.x] = fminf(smin[threadIdx.x], smin[threadIdx.x + o]);
            smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + o]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { mm[0] = smin[0]; mm[1] = smax[0]; }
}

__global__ void map_normalize_kernel(const float* x, long long n, int mode, const float* __restrict__ mm,
                                     float* out, float* out2) {              // out may alias x (same index read, then written)
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float lo = mm[0], hi = mm[1];
    float v = x[i];
    if (mode == 1) {
        if (v < 0.f) v = __fdiv_rn(v, fabsf(lo));
        // numpy re-reads the maximum after the negatives were rescaled: positives are untouched by that, so it is `hi`
        if (v > 0.f) v = __fdiv_rn(v, hi);
        out[i] = __fdiv_rn(v + 1.0f, 2.0f);
    } else if (mode == 2) {
        out[i] = __fdiv_rn(v, fmaxf(fabsf(lo), fabsf(hi)));
    } else if (mode == 3) {
        out[i] = __fdiv_rn(fmaxf(v, 0.f), fmaxf(hi, 0.f));
    } else {
        const float d = __fdiv_rn(v, fabsf(hi));
        out[i] = fminf(fmaxf(d, 0.f), 1.f);
        out2[i] = -fminf(fmaxf(d, -1.f), 0.f);
    }
}

__global__ void mean_reduce_kernel(const float* __restrict__ map, int n, float* __restrict__ out) {
    __shared__ double sh[256];
    map += (size_t)blockIdx.x * n;
    out += blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += (double)map[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(sh[0] / (double)n);
}

}  // namespace

// DIFT patch descriptor: window mean of every channel of the ensemble-mean feature map, then L2
// normalisation across channels (cluster.py:291-299: emb[:, r0:r1, c0:c1].mean((1,2)); emb / ||emb||).
// One block per patch; a wavefront per channel (lanes stride over the window), means staged in LDS.
// An empty window gives NaN like numpy's mean of an empty slice.
__global__ __launch_bounds__(256)
void patch_embed_kernel(const float* __restrict__ feat, int C, int h, int w, const int32_t* __restrict__ boxes,
                        float* __restrict__ out) {
    extern __shared__ float mean_s[];            // [C] + 4 partial sums
    const int pch = blockIdx.x;
    int r0 = boxes[pch * 4 + 0], r1 = boxes[pch * 4 + 1], c0 = boxes[pch * 4 + 2], c1 = boxes[pch * 4 + 3];
    r0 = r0 < 0 ? 0 : r0; c0 = c0 < 0 ? 0 : c0; r1 = r1 > h ? h : r1; c1 = c1 > w ? w : c1;   // numpy slice clamping
    const int wh = r1 > r0 ? r1 - r0 : 0, ww = c1 > c0 ? c1 - c0 : 0;
    const int npx = wh * ww;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int c = wv; c < C; c += 4) {
        const float* f = feat + (size_t)c * h * w;
        float s = 0.f;
        for (int i = lane; i < npx; i += 64) {
            const int r = i / ww, col = i - r * ww;
            s += f[(r0 + r) * w + c0 + col];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) mean_s[c] = s / (float)npx;                 // 0/0 = NaN for an empty window
    }
    __syncthreads();
    float q = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) q += mean_s[c] * mean_s[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    if (lane == 0) mean_s[C + wv] = q;
    __syncthreads();
    const float inv = 1.0f / sqrtf(mean_s[C] + mean_s[C + 1] + mean_s[C + 2] + mean_s[C + 3]);
    for (int c = threadIdx.x; c < C; c += 256) out[(size_t)pch * C + c] = mean_s[c] * inv;
}

hipError_t launch_time_gather(const f16* table, const int64_t* t, int B, int dim, f16* out, hipStream_t s) {
    hipLaunchKernelGGL(time_gather_kernel, dim3((B * dim + 255) / 256), dim3(256), 0, s, table, t, B, dim, out);
    return hipGetLastError();
}

hipError_t launch_silu(const f16* in, f16* out, long long n, hipStream_t s) {
    hipLaunchKernelGGL(silu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, n);
    return hipGetLastError();
}

hipError_t launch_im2col_in(const void* x, const int32_t* x_index, const void* eps, const int64_t* t, const void* sa,
                            const void* sb, int latent_f32, int B, int H, int W, f16* out, hipStream_t s) {
    const long long total = (long long)B * H * W;
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (latent_f32)
        hipLaunchKernelGGL(im2col_in_kernel<float>, grid, block, 0, s, (const float*)x, x_index, (const float*)eps, t,
                           (const float*)sa, (const float*)sb, B, H, W, out);
    else
        hipLaunchKernelGGL(im2col_in_kernel<f16>, grid, block, 0, s, (const f16*)x, x_index, (const f16*)eps, t,
                           (const f16*)sa, (const f16*)sb, B, H, W, out);
    return hipGetLastError();
}

hipError_t launch_conv_out(const f16* Xn, const f16* w, const f16* bias, const void* eps, int eps_f32, int B, int H, int W,
                           int C0, float* loss, f16* pred, int eps_rows, int out_group, int out_stride, int out_off,
                           hipStream_t s) {
    if (option(OPT_CONV_OUT_ROWS) != 0 && conv_out_rows_strip(H, W, C0) > 0)
        return launch_conv_out_rows(Xn, w, bias, eps, eps_f32, B, H, W, C0, loss, pred, eps_rows, out_group, out_stride, out_off, s);
    const long long npix = (long long)B * H * W;
    const size_t lds = (size_t)4 * 9 * C0 * sizeof(f16);
    const dim3 grid((unsigned)((npix + 32 * CONV_OUT_GROUPS - 1) / (32 * CONV_OUT_GROUPS))), block(256);
    if (eps_f32)
        hipLaunchKernelGGL(conv_out_kernel<float>, grid, block, lds, s, Xn, w, bias, (const float*)eps,
                           B, H, W, C0, loss, pred, eps_rows, out_group, out_stride, out_off);
    else
        hipLaunchKernelGGL(conv_out_kernel<f16>, grid, block, lds, s, Xn, w, bias, (const f16*)eps,
                           B, H, W, C0, loss, pred, eps_rows, out_group, out_stride, out_off);
    return hipGetLastError();
}

hipError_t launch_nhwc_to_nchw(const f16* X, int N, int HW, int C, f16* Y, hipStream_t s) {
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((HW + 31) / 32, (C + 31) / 32, N), dim3(256), 0, s, X, HW, C, Y);
    return hipGetLastError();
}

hipError_t launch_ensemble_mean(const f16* X, int groups, int ens, int HW, int C, float* Y, hipStream_t s) {
    hipLaunchKernelGGL(ensemble_mean_kernel, dim3((HW + 31) / 32, (C + 31) / 32, groups), dim3(256), 0, s, X, ens,
                       HW, C, Y);
    return hipGetLastError();
}

hipError_t launch_typicality(const void* loss, int is_f16, int n_images, int n_draws, int n_cond, int HW, int cond_major,
                             float* map, float* scalar, hipStream_t s) {
    const long long s_img = cond_major ? n_draws : (long long)n_draws * n_cond;
    const long long s_draw = cond_major ? 1 : n_cond;
    const long long s_cond = cond_major ? (long long)n_images * n_draws : 1;
    const dim3 grid((HW + 255) / 256, n_images);
    if (is_f16)
        hipLaunchKernelGGL(typicality_map_kernel<f16>, grid, dim3(256), 0, s, (const f16*)loss,
                           n_draws, n_cond, HW, map, s_img, s_draw, s_cond);
    else
        hipLaunchKernelGGL(typicality_map_kernel<float>, grid, dim3(256), 0, s,
                           (const float*)loss, n_draws, n_cond, HW, map, s_img, s_draw, s_cond);
    if (scalar) hipLaunchKernelGGL(mean_reduce_kernel, dim3(n_images), dim3(256), 0, s, map, HW, scalar);
    return hipGetLastError();
}

hipError_t launch_typicality_image(const float* map, int h, int w, int H, int W, int kx, int ky, float* tmp,
                                   float* out, hipStream_t s) {
    const int OWd = W - ky + 1, OHd = H - kx + 1;
    if (OWd <= 0 || OHd <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(upsample_rowsum_kernel, dim3((H * OWd + 255) / 256), dim3(256), 0, s, map, h, w, H, W, ky, tmp);
    hipLaunchKernelGGL(colsum_kernel, dim3((OHd * OWd + 255) / 256), dim3(256), 0, s, tmp, H, OWd, kx,
                       1.0f / ((float)kx * (float)ky), out);
    return hipGetLastError();
}

hipError_t launch_map_normalize(const float* map, long long n, int mode, float* mm, float* out, float* out2, hipStream_t s) {
    if (n <= 0 || mode < 1 || mode > 4 || (mode == 4 && !out2)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(map_minmax_kernel, dim3(1), dim3(1024), 0, s, map, n, mm);
    hipLaunchKernelGGL(map_normalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, map, n, mode, mm, out, out2);
    return hipGetLastError();
}

hipError_t launch_patch_embed(const float* feat, int C, int h, int w, const int32_t* boxes, int P, float* out, hipStream_t s) {
    if (C <= 0 || h <= 0 || w <= 0 || P <= 0 || C > 8192) return hipErrorInvalidValue;
    hipLaunchKernelGGL(patch_embed_kernel, dim3(P), dim3(256), (size_t)(C + 4) * sizeof(float), s, feat, C, h, w, boxes, out);
    return hipGetLastError();
}

}  // namespace dm
