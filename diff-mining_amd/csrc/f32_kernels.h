// f32_kernels.h — launcher prototypes of the fp32 kernels (internal, C++): the arithmetic of the reference's DIFT path,
// which runs the U-Net WITHOUT autocast and without torch_dtype (diffmining/typicality/dift.py:197-199: plain fp32).
// fp32 operands, fp32 MFMA (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulation), fp32 tensors in HBM (NHWC).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dm32 {

// modes as dm::IGemmMode: 0 dense, 1 conv3x3 stride 1, 2 conv3x3 stride 2 (pad 1), 3 nearest-upsample to (OH, OW) then conv3x3,
// 4 conv3x3 stride 2 with pad (0,1,0,1) (VAE downsampler), 5 Upsample2D (nearest, exact 2x) + conv3x3 as four 2x2 convolutions on the source
// grid (H = OH, W = OW = the source, M = N H W rows per parity class, Wp [4][Cout][4 Cin] pre-summed taps, Y [N][2H][2W][Cout])
struct GemmParams {
    const float* X = nullptr;     // source 1, NHWC [N,H,W,C1] (dense: [M,C1])
    const float* X2 = nullptr;    // source 2 (channel concat), NHWC [N,H,W,Cin-C1], or nullptr
    const float* Wp = nullptr;    // packed weights [Cout][taps*Cin], k = (tap, cin)
    const float* bias = nullptr;  // [Cout] or nullptr
    const float* temb = nullptr;  // per-sample channel add [N][temb_ld] (already offset) or nullptr
    const float* res = nullptr;   // residual [M][ldres] or nullptr
    float* Y = nullptr;           // [M][ldy]
    int M = 0, Cout = 0, Cin = 0, C1 = 0;
    int H = 1, W = 1, OH = 1, OW = 1;
    int mode = 0;
    int ldy = 0, ldres = 0, temb_ld = 0;
    // epi = 1: GEGLU — the weight rows are packed in quads (h0, h1, g0, g1) (value rows 2q, 2q + 1 and gate rows F + 2q, F + 2q + 1 of
    // ff.net.0.proj [2F][C], F = Cout / 2), a lane's four channels are one quad, Y [M][F] gets (h0 gelu(g0), h1 gelu(g1)) at columns 2q, 2q + 1
    int epi = 0;
};
hipError_t launch_gemm(const GemmParams& p, hipStream_t s);

struct AttnParams {
    const float* Q; const float* K; const float* V; float* O;
    int ldq, ldk, ldv, ldo;            // row strides in elements
    long long bsq, bsk, bsv, bso;      // batch strides in elements
    const int32_t* kv_slot = nullptr;  // optional: K/V batch index per sample (prompt slot), clamped to [0, n_slots)
    int n_slots = 0;
    int B, heads, Tq, Tk, D;           // D in {40, 80, 160, 512}
    float scale;
};
hipError_t launch_attention(const AttnParams& p, hipStream_t s);

// GroupNorm over cat([X (C1 channels), X2 (C - C1)]) NHWC: stats [N*G][2] = (mean, rstd), then the apply (+ SiLU)
hipError_t launch_gn_stats(const float* X, const float* X2, int N, int HW, int C, int C1, int G, float eps, float* stats, hipStream_t s);
hipError_t launch_gn_apply(const float* X, const float* X2, int N, int HW, int C, int C1, int G, const float* gamma, const float* beta,
                           const float* stats, int silu, float* Y, hipStream_t s);
hipError_t launch_layernorm(const float* X, int rows, int C, const float* gamma, const float* beta, float eps, float* Y, hipStream_t s);
// CLIP text tower glue (fp32): token + position embedding; causal attention of <= 80 tokens x heads of 64 on the stacked q|k|v rows
// [n*T][3*heads*64] (q scaled by 1/8 inside); y * sigmoid(1.702 y) in place
hipError_t launch_clip_embed(const int32_t* ids, const float* tok, const float* pos, int rows, int T, int C, int vocab, float* out, hipStream_t s);
hipError_t launch_clip_attention(const float* qkv, int n, int T, int heads, float* out, hipStream_t s);
hipError_t launch_quick_gelu(float* x, long long n, hipStream_t s);
hipError_t launch_silu(const float* in, float* out, long long n, hipStream_t s);
// Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): out [B][dim] = [cos | sin]
hipError_t launch_timestep_embed(const int64_t* t, int B, int dim, float* out, hipStream_t s);
// conv_in: sample NCHW [B,Cin<=4,H,W] (x) 3x3 pad 1 -> NHWC [B,H,W,Cout]; w transposed [Cin*9][Cout] (k = (ci, dy, dx): consecutive
// threads = consecutive output channels read consecutive weights), bias [Cout]
hipError_t launch_conv_in(const float* x, const float* w, const float* bias, int B, int Cin, int H, int W, int Cout, float* y, hipStream_t s);
// conv_out: NHWC [B,H,W,C] (x) 3x3 pad 1 -> NCHW [B,Cout<=8,H,W]; w packed [Cout][9*C] k = (tap, c)
hipError_t launch_conv_out(const float* x, const float* w, const float* bias, int B, int H, int W, int C, int Cout, float* y, hipStream_t s);
// scheduler.add_noise in fp32: out[b] = sa[t[b]] * x[x_index ? x_index[b] : b] + sb[t[b]] * eps[b]; `per` elements per sample
hipError_t launch_add_noise(const float* x, const int32_t* x_index, const float* eps, const int64_t* t, const float* sa, const float* sb,
                            int B, long long per, float* out, hipStream_t s);
hipError_t launch_sqerr(const float* pred, const float* eps, long long n, float* out, hipStream_t s);      // (pred - eps)^2
hipError_t launch_nhwc_to_nchw(const float* X, int N, int HW, int C, float* Y, hipStream_t s);
// quant_conv (1x1, 8 -> 8) on Hm [B*HW][8] + `draws` posterior samples per image * scaling (NCHW outputs: latent [B*draws,4,HW], moments [B,8,HW])
hipError_t launch_posterior(const float* Hm, const float* qw, const float* qb, const float* noise, int B, int draws, int HW, float scaling,
                            float* latent, float* moments, hipStream_t s);
// mean over groups of `ens` consecutive samples, NHWC -> NCHW
hipError_t launch_ensemble_mean(const float* X, int groups, int ens, int HW, int C, float* Y, hipStream_t s);

}  // namespace dm32
