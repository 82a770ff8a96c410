// vae.hip — kernels specific to the SDv1.5 VAE encoder (`vae.encode(img).latent_dist.sample() *
// scaling_factor`, diffmining/typicality/compute.py:91-93,137; dift.py:187).
// The encoder's convolutions and GroupNorms run on the shared igemm / norm kernels; this file adds
//   * im2col_rgb_kernel  — NCHW RGB image -> im2col rows of `encoder.conv_in` (3x3, 3 ch -> k = 27,
//                          zero-padded to 64), so conv_in is one igemm;
//   * attn512_kernel     — the mid block's single-head attention, head_dim 512 (flash style);
//   * posterior_kernel   — `quant_conv` (1x1, 8 -> 8) + DiagonalGaussianDistribution.sample()
//                          with the draw injected, times scaling_factor.
#include "dm_kernels.h"

namespace dm {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace {

__global__ void im2col_rgb_kernel(const f16* __restrict__ x, int B, int H, int W, f16* __restrict__ out) {
    const long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int HW = H * W;
    if (pix >= (long long)B * HW) return;
    const int b = (int)(pix / HW);
    const int rem = (int)(pix - (long long)b * HW);
    const int oh = rem / W, ow = rem - oh * W;
    f16 row[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) row[k] = (f16)0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ih = oh + dy - 1, iw = ow + dx - 1;
                if (ih >= 0 && ih < H && iw >= 0 && iw < W)
                    row[c * 9 + dy * 3 + dx] = x[((size_t)b * 3 + c) * HW + (size_t)ih * W + iw];
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

// ------------------------------------------------------------------------------------------------
// Single-head attention, head_dim 512.  Block = 4 waves, 64 queries; keys in tiles of 32.
//   phase S : wave w computes S^T = K Q^T for its 16 queries (Q fragments live in registers for the
//             whole kernel: 16 k-steps x 16 B), online softmax in fp32, writes P (fp16) and the
//             rescale factors to LDS;
//   phase PV: wave w owns output columns [128 w, 128 w + 128) of all 64 queries:
//             O^T += V^T P^T with V^T staged transposed in LDS.
// ~2 % of the encoder's FLOPs: kept simple (plain loads, three barriers per key tile).
// ------------------------------------------------------------------------------------------------
constexpr int AD = 512, AQ = 64, AK = 32;
constexpr int KS_STRIDE = AD * 2 + 16;            // bytes per K row  (conflict-free 16-B reads)
constexpr int VT_STRIDE = AK * 2 + 16;            // bytes per V^T row
constexpr int PS_STRIDE = AK * 2 + 16;
constexpr int KS_BYTES = AK * KS_STRIDE, VT_BYTES = AD * VT_STRIDE, PS_BYTES = AQ * PS_STRIDE;
constexpr int ATTN512_LDS = KS_BYTES + VT_BYTES + PS_BYTES + 2 * AQ * 4;

__global__ __launch_bounds__(256, 1)
void attn512_kernel(const f16* __restrict__ Q, const f16* __restrict__ K, const f16* __restrict__ V,
                    f16* __restrict__ O, int T, int ld, int ldo, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vt = smem + KS_BYTES;
    char* Ps = Vt + VT_BYTES;
    float* alpha_s = reinterpret_cast<float*>(Ps + PS_BYTES);
    float* l_s = alpha_s + AQ;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int b = blockIdx.y;
    const int q0 = blockIdx.x * AQ;
    const size_t base = (size_t)b * T;

    // Q fragments (B operand of S^T = K Q^T): lane (l15 = query, lg) holds Q[q][32 kk + 8 lg .. +8]
    half8 qf[16];
    {
        int q = q0 + w * 16 + l15;
        q = q < T ? q : T - 1;
        const f16* qp = Q + (base + q) * ld + lg * 8;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) qf[kk] = *reinterpret_cast<const half8*>(qp + kk * 32);
    }
    floatx4 oacc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) oacc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    for (int k0 = 0; k0 < T; k0 += AK) {
        // ---- stage K (row major) and V (transposed) of this key tile ----
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int chunk = tid + 256 * i;
            {   // K: coalesced rows
                const int key = chunk >> 6, c8 = chunk & 63;
                half8 v = half8{0, 0, 0, 0, 0, 0, 0, 0};
                if (k0 + key < T) v = *reinterpret_cast<const half8*>(K + (base + k0 + key) * ld + c8 * 8);
                *reinterpret_cast<half8*>(Ks + key * KS_STRIDE + c8 * 16) = v;
            }
            {   // V: lanes vary in key so the transposing 2-byte writes spread over the banks
                const int key = chunk & 31, c8 = chunk >> 5;
                half8 v = half8{0, 0, 0, 0, 0, 0, 0, 0};
                if (k0 + key < T) v = *reinterpret_cast<const half8*>(V + (base + k0 + key) * ld + c8 * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) *reinterpret_cast<f16*>(Vt + (c8 * 8 + j) * VT_STRIDE + key * 2) = v[j];
            }
        }
        __syncthreads();

        // ---- phase S ----
        floatx4 sacc[2] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const half8 a = *reinterpret_cast<const half8*>(Ks + (blk * 16 + l15) * KS_STRIDE + (kk * 32 + lg * 8) * 2);
                sacc[blk] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, qf[kk], sacc[blk], 0, 0, 0);
            }
        }
        // lane holds scores of query l15 against keys blk*16 + lg*4 + r
        float sv[8];
        float mt = -INFINITY;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = k0 + blk * 16 + lg * 4 + r;
                const float s = (key < T) ? sacc[blk][r] * scale : -INFINITY;
                sv[blk * 4 + r] = s;
                mt = fmaxf(mt, s);
            }
        mt = fmaxf(mt, __shfl_xor(mt, 16));
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = __expf(m_run - m_new);
        float ls = 0.f;
        f16 pv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float pe = __expf(sv[i] - m_new);
            pv[i] = (f16)pe;
            ls += pe;
        }
        ls += __shfl_xor(ls, 16);
        ls += __shfl_xor(ls, 32);
        l_run = l_run * alpha + ls;
        m_run = m_new;
        {
            const int q = w * 16 + l15;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
                *reinterpret_cast<half4*>(Ps + q * PS_STRIDE + (blk * 16 + lg * 4) * 2) =
                    half4{pv[blk * 4 + 0], pv[blk * 4 + 1], pv[blk * 4 + 2], pv[blk * 4 + 3]};
            if (lg == 0) alpha_s[q] = alpha;
        }
        __syncthreads();

        // ---- phase PV: O^T[d][q] = alpha[q] O^T[d][q] + sum_key V^T[d][key] P[q][key] ----
        half8 pb[4];
        float al[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            pb[j] = *reinterpret_cast<const half8*>(Ps + (j * 16 + l15) * PS_STRIDE + lg * 16);
            al[j] = alpha_s[j * 16 + l15];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const half8 a = *reinterpret_cast<const half8*>(Vt + (w * 128 + i * 16 + l15) * VT_STRIDE + lg * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                floatx4 c = oacc[i][j];
                c[0] *= al[j]; c[1] *= al[j]; c[2] *= al[j]; c[3] *= al[j];
                oacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, pb[j], c, 0, 0, 0);
            }
        }
        __syncthreads();
    }

    if (lg == 0) l_s[w * 16 + l15] = l_run;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = q0 + j * 16 + l15;
        if (q >= T) continue;
        const float inv = 1.0f / l_s[j * 16 + l15];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const floatx4 c = oacc[i][j];
            const half4 o = half4{(f16)(c[0] * inv), (f16)(c[1] * inv), (f16)(c[2] * inv), (f16)(c[3] * inv)};
            *reinterpret_cast<half4*>(O + (base + q) * ldo + w * 128 + i * 16 + lg * 4) = o;
        }
    }
}

// moments = quant_conv(h) (fp16 out, fp32 accumulate), mean | logvar = chunk(moments, 2, C);
// latent = (mean + exp(0.5 clamp(logvar, -30, 20)) * noise) * scaling   (fp32; noise == nullptr -> mode)
// `draws` posterior samples per image: output sample bo = b * draws + d reads the moments of image b.
__global__ void posterior_kernel(const f16* __restrict__ Hm, int ldh, const f16* __restrict__ qw /* [8][8] */,
                                 const f16* __restrict__ qb, const f16* __restrict__ noise, int B, int draws, int HW,
                                 float scaling, f16* __restrict__ latent16, float* __restrict__ latent32,
                                 float* __restrict__ moments) {
    const long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= (long long)B * draws * HW) return;
    const int bo = (int)(pix / HW);
    const int rem = (int)(pix - (long long)bo * HW);
    const int b = bo / draws;
    const half8 hv = *reinterpret_cast<const half8*>(Hm + ((size_t)b * HW + rem) * ldh);
    float m[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += (float)qw[o * 8 + i] * (float)hv[i];
        acc += (float)qb[o];
        m[o] = (float)(f16)acc;
        if (moments && bo == b * draws) moments[((size_t)b * 8 + o) * HW + rem] = m[o];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float v = m[c];
        if (noise) {
            float lv = m[4 + c];
            lv = lv < -30.f ? -30.f : (lv > 20.f ? 20.f : lv);
            v += expf(0.5f * lv) * (float)noise[((size_t)bo * 4 + c) * HW + rem];
        }
        v *= scaling;
        const size_t o = ((size_t)bo * 4 + c) * HW + rem;
        if (latent16) latent16[o] = (f16)v;
        if (latent32) latent32[o] = v;
    }
}

}  // namespace

hipError_t launch_im2col_rgb(const f16* x, int B, int H, int W, f16* out, hipStream_t s) {
    const long long total = (long long)B * H * W;
    hipLaunchKernelGGL(im2col_rgb_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, B, H, W, out);
    return hipGetLastError();
}

hipError_t launch_attention512(const f16* Q, const f16* K, const f16* V, f16* O, int B, int T, int ld, int ldo,
                               float scale, hipStream_t s) {
    if (T <= 0 || B <= 0 || ld % 8 || ldo % 4) return hipErrorInvalidValue;
    static std::atomic<uint64_t> attr_seen{0};      // hipFuncSetAttribute is per DEVICE, not per process
    if (first_use_on_device(attr_seen)) {
        (void)hipFuncSetAttribute((const void*)attn512_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ATTN512_LDS);
    }
    launch_timed(attn512_kernel, dim3((T + AQ - 1) / AQ, B), dim3(256), ATTN512_LDS, s, Q, K, V, O, T, ld, ldo, scale);
    return hipGetLastError();
}

hipError_t launch_posterior(const f16* Hm, int ldh, const f16* qw, const f16* qb, const f16* noise, int B, int draws,
                            int HW, float scaling, f16* latent16, float* latent32, float* moments, hipStream_t s) {
    const long long total = (long long)B * draws * HW;
    hipLaunchKernelGGL(posterior_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, Hm, ldh, qw, qb, noise,
                       B, draws, HW, scaling, latent16, latent32, moments);
    return hipGetLastError();
}

}  // namespace dm
