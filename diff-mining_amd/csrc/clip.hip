// clip.hip — kernels specific to the CLIP ViT-L/14 text tower whose `last_hidden_state` is the prompt
// embedding c (`CategoryFeatures.embed`, diffmining/typicality/compute.py:39-51).
// It runs once per category set (n prompts x 77 tokens), so these are plain, exact kernels; the
// Linear layers and LayerNorms run on the shared igemm / norm kernels.
//   * clip_embed_kernel   — token_embedding[ids] + position_embedding (fp16 add, like the fp16 model)
//   * clip_attn_kernel    — causal self-attention, 12 heads of 64, <= 77 tokens, one block per
//                           (prompt, head); q arrives pre-scaled (d^-0.5 = 1/8 is folded into q_proj)
//   * quick_gelu_kernel   — y * sigmoid(1.702 y)
#include "dm_kernels.h"

namespace dm {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

namespace {

__global__ void clip_embed_kernel(const int32_t* __restrict__ ids, const f16* __restrict__ tok, const f16* __restrict__ pos,
                                  int rows, int T, int C, int vocab, f16* __restrict__ out) {
    const int row = blockIdx.x;
    if (row >= rows) return;
    int id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const f16* a = tok + (size_t)id * C;
    const f16* b = pos + (size_t)(row % T) * C;
    for (int c = threadIdx.x * 8; c < C; c += blockDim.x * 8) {
        const half8 x = *reinterpret_cast<const half8*>(a + c), y = *reinterpret_cast<const half8*>(b + c);
        *reinterpret_cast<half8*>(out + (size_t)row * C + c) = x + y;        // fp16 add, rounded
    }
}

constexpr int CD = 64, CT = 80;          // head_dim, max tokens (77) rounded up

__global__ __launch_bounds__(128)
void clip_attn_kernel(const f16* __restrict__ qkv, int T, int heads, f16* __restrict__ out) {
    __shared__ float Ks[CT][CD + 1], Vs[CT][CD + 1];
    const int h = blockIdx.x, n = blockIdx.y;
    const int C = heads * CD;
    const f16* base = qkv + (size_t)n * T * 3 * C + h * CD;
    for (int i = threadIdx.x; i < T * CD; i += blockDim.x) {
        const int t = i / CD, d = i - t * CD;
        Ks[t][d] = (float)base[(size_t)t * 3 * C + C + d];
        Vs[t][d] = (float)base[(size_t)t * 3 * C + 2 * C + d];
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t >= T) return;
    float q[CD];
#pragma unroll
    for (int d = 0; d < CD; ++d) q[d] = (float)base[(size_t)t * 3 * C + d];
    float s[CT];
    float m = -INFINITY;
    for (int k = 0; k <= t; ++k) {                 // causal: keys 0..t
        float a = 0.f;
#pragma unroll
        for (int d = 0; d < CD; ++d) a = fmaf(q[d], Ks[k][d], a);
        s[k] = a;
        m = fmaxf(m, a);
    }
    float l = 0.f;
    for (int k = 0; k <= t; ++k) { s[k] = expf(s[k] - m); l += s[k]; }
    const float inv = 1.0f / l;
    float o[CD];
#pragma unroll
    for (int d = 0; d < CD; ++d) o[d] = 0.f;
    for (int k = 0; k <= t; ++k) {
        const float pk = (float)(f16)(s[k] * inv);          // the fp16 model's softmax output
#pragma unroll
        for (int d = 0; d < CD; ++d) o[d] = fmaf(pk, Vs[k][d], o[d]);
    }
    f16* dst = out + ((size_t)n * T + t) * C + h * CD;
#pragma unroll
    for (int d = 0; d < CD; ++d) dst[d] = (f16)o[d];
}

__global__ void quick_gelu_kernel(f16* __restrict__ x, long long n8) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    half8 v = *reinterpret_cast<half8*>(x + i * 8);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float y = (float)v[k];
        v[k] = (f16)(y / (1.0f + expf(-1.702f * y)));
    }
    *reinterpret_cast<half8*>(x + i * 8) = v;
}

__global__ void f16_to_f32_kernel(const f16* __restrict__ x, float* __restrict__ y, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = (float)x[i];
}

}  // namespace

hipError_t launch_clip_embed(const int32_t* ids, const f16* tok, const f16* pos, int rows, int T, int C, int vocab, f16* out,
                             hipStream_t s) {
    if (C % 8 || rows <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(clip_embed_kernel, dim3(rows), dim3(96), 0, s, ids, tok, pos, rows, T, C, vocab, out);
    return hipGetLastError();
}

hipError_t launch_clip_attention(const f16* qkv, int n, int T, int heads, f16* out, hipStream_t s) {
    if (T <= 0 || T > CT || n <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(clip_attn_kernel, dim3(heads, n), dim3(128), 0, s, qkv, T, heads, out);
    return hipGetLastError();
}

hipError_t launch_quick_gelu(f16* x, long long n, hipStream_t s) {
    if (n % 8) return hipErrorInvalidValue;
    const long long n8 = n / 8;
    hipLaunchKernelGGL(quick_gelu_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, x, n8);
    return hipGetLastError();
}

hipError_t launch_f16_to_f32(const f16* x, float* y, long long n, hipStream_t s) {
    hipLaunchKernelGGL(f16_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, y, n);
    return hipGetLastError();
}

}  // namespace dm
