/*
 * dm_engine.h — C ABI of the MI355X typicality-scoring engine (libdm_engine.so).
 *
 * This is the drop-in boundary for diff-mining's hot path.  The reference has no FFI of its own:
 * its seam is the Python attribute `self.model.unet` / `self.pipe.unet` (a diffusers
 * `UNet2DConditionModel`), driven from
 *     diffmining/typicality/compute.py:95-102   SD.compute_loss   (add_noise -> unet -> mse_loss)
 *     diffmining/typicality/compute.py:134-160  D.compute_losses  ([N,2,4,h,w] fp16 grid)
 *     diffmining/typicality/dift.py:24-169      MyUNet2DConditionModel.forward (up_ft tap)
 *     diffmining/typicality/dift.py:214-232     SDFeaturizer.forward (ensemble mean)
 * Each entry point below names the reference call it replaces.  Signatures use plain pointers and
 * sizes only (no torch types); the Python shim in diff-mining_amd/engine.py binds them with ctypes
 * and INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, nonzero on failure; dm_last_error() gives the message.
 *   - *_dev pointers are device (HBM) addresses owned by the caller; `stream` is a hipStream_t
 *     (NULL = default stream).  Calls are asynchronous on that stream.
 *   - tensors at the boundary are NCHW like the reference; the engine is NHWC fp16 inside.
 *   - one engine per device, not re-entrant (the reference is single-stream, compute.py:215).
 */
#ifndef DM_ENGINE_H
#define DM_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dm_engine dm_engine;

enum { DM_F16 = 0, DM_F32 = 1 };

/* Library version / build info string (host only; usable without a GPU). */
const char* dm_version(void);

/* Host-only helpers, callable without a GPU (used by the CPU test tier).
 * dm_scheduler_alphas_cumprod: the `scaled_linear` ᾱ table of the SDv1.5 scheduler whose
 *   `add_noise` the reference calls at compute.py:99 / dift.py:190.  out[n] fp32.
 * dm_timestep_sinusoid: diffusers `Timesteps(320, flip_sin_to_cos=True, shift=0)` row for
 *   integer t (dift.py:84), out[dim] fp32 = [cos | sin]. */
int dm_scheduler_alphas_cumprod(int num_train_timesteps, float beta_start, float beta_end, float* out);
int dm_timestep_sinusoid(int t, int dim, float* out);

/* Create an engine for the SDv1.5 U-Net architecture on HIP device `device`.
 * Replaces: StableDiffusionPipeline.from_pretrained(...).unet (compute.py:65-73). */
int dm_engine_create(int device, dm_engine** out);
void dm_engine_destroy(dm_engine* e);
const char* dm_last_error(dm_engine* e);   /* e may be NULL: last create() error */

/* Hand one diffusers-named tensor (e.g. "down_blocks.0.resnets.0.conv1.weight") to the engine.
 * `host_ptr` is host memory in the PyTorch layout ([Cout,Cin,kh,kw] / [out,in] / [C]);
 * dtype is DM_F16 or DM_F32 (values are rounded to fp16: the reference loads the pipeline with
 * torch_dtype=float16, compute.py:69).  Replaces: from_pretrained's state-dict load. */
int dm_engine_load_weight(dm_engine* e, const char* name, const void* host_ptr, int dtype,
                          const int64_t* shape, int ndim);

/* Check that all 686 tensors arrived, pack them into the MFMA-friendly HBM layout, upload. */
int dm_engine_finalize(dm_engine* e);

/* Register the distinct prompt embeddings of the next calls and precompute the cross-attention
 * K/V of all 16 transformer blocks for them (K/V depend only on the prompt, not on (t, eps)).
 * ctx_dev: [n_prompts, 77, 768] fp16.  Replaces: the `c` argument of compute_loss
 * (compute.py:96,100) / `encoder_hidden_states` (dift.py:191), which the reference re-projects
 * for every sample. */
int dm_engine_set_prompts(dm_engine* e, const void* ctx_dev, int n_prompts, void* stream);

/* SD.compute_loss (compute.py:95-102), fused: noisy = add_noise(x[x_index[b]], eps[b], t[b]);
 * eps_hat = UNet(noisy, t, prompt[slot[b]]); loss = (float(eps_hat) - float(eps))^2.
 *   latent_dtype DM_F32 or DM_F16: the element type of x_dev / eps_dev AND the arithmetic of add_noise.
 *     DM_F32 is what the reference's run does: `encode_vae` returns an fp32 latent under autocast (the
 *     posterior's exp() is promoted, compute.py:91-93), so `randn_like(x)` (:116), the scheduler's
 *     sqrt(acp[t]) / sqrt(1-acp[t]) coefficients and the sum (:99) are fp32, autocast rounds the noisy latent
 *     to fp16 once at conv_in, and mse_loss (:101) sees the fp32 eps.
 *     DM_F16 is the flow of an fp16 latent: table cast to fp16 first, fp16 products and sum, fp16 eps in the loss.
 *   x_dev        [n_x,4,h,w] (n_x images; the reference broadcasts one x over 2B rows)
 *   x_index_dev  [batch] int32 row of x for sample b, or NULL for identity (n_x == batch)
 *   eps_dev      [batch,4,h,w]
 *   t_dev        [batch] int64 in [0,1000)
 *   slot_dev     [batch] int32 prompt slot in [0, n_prompts) (see dm_engine_set_prompts)
 *   loss_out_dev [batch,4,h,w] fp32 (NCHW, as F.mse_loss(reduction='none') returns)
 */
int dm_score(dm_engine* e, const void* x_dev, const int32_t* x_index_dev, const void* eps_dev,
             const int64_t* t_dev, const int32_t* slot_dev, int batch, int n_x, int h, int w, int latent_dtype,
             void* loss_out_dev, void* stream);

/* The same, for the way D.compute_losses actually calls it (compute.py:145-155): every one of
 * n_draws (x, eps, t) draws is scored under the n_cond prompts registered in slots 0..n_cond-1.
 * The part of the U-Net that cannot see the prompt (conv_in, down_blocks.0.resnets.0 and its
 * transformer up to the self-attention residual) is evaluated once per draw instead of once per
 * (draw, prompt) pair; results are bit-identical to dm_score on the tiled batch.
 *   x_index_dev [n_draws] or NULL; eps_dev [n_draws,4,h,w]; t_dev [n_draws]
 *   loss_out_dev [n_cond * n_draws, 4, h, w] fp32, cond-major: row k*n_draws + i = draw i, prompt k
 *   (exactly the layout `torch.split(loss, [B]*n_cond)` expects at compute.py:155). */
int dm_score_conds(dm_engine* e, const void* x_dev, const int32_t* x_index_dev, const void* eps_dev,
                   const int64_t* t_dev, int n_cond, int n_draws, int n_x, int h, int w, int latent_dtype,
                   void* loss_out_dev, void* stream);

/* dm_score_conds with a prompt-slot TABLE: slot_table_dev [n_cond][n_draws] int32, entry (k, i) = the registered prompt that
 * draw i is scored under in its k-th condition.  This is the batched form of the reference's work list (compute.py:284-290: one
 * `path,category` line per image, D.compute(country, path) scores each image under ITS OWN category and the shared null prompt,
 * :182-192): the draws of several images ride in one call, image j's draws carry (slot of category_j, slot of "") — the
 * prompt-independent prefix still runs once per draw, and every draw's bits equal those of a per-image dm_score_conds call.
 * slot_table_dev == NULL is dm_score_conds (prompt k for every draw).  Entries are clamped to the registered prompts on the device. */
int dm_score_conds_slots(dm_engine* e, const void* x_dev, const int32_t* x_index_dev, const void* eps_dev,
                         const int64_t* t_dev, const int32_t* slot_table_dev, int n_cond, int n_draws, int n_x, int h, int w,
                         int latent_dtype, void* loss_out_dev, void* stream);

/* `unet(sample, t, encoder_hidden_states).sample` (compute.py:100) without the fused
 * add_noise / loss: sample_dev [batch,4,h,w] fp16 -> out_dev [batch,4,h,w] fp16 (NCHW). */
int dm_unet_forward(dm_engine* e, const void* sample_dev, const int64_t* t_dev, const int32_t* slot_dev,
                    int batch, int h, int w, void* out_dev, void* stream);

/* MyUNet2DConditionModel.forward(latents_noisy, t, [up_ft_index], prompt_embeds) (dift.py:24-169):
 * early exit after up_blocks[up_ft_index]; feat_out_dev [batch, C_i, h_i, w_i] fp16 NCHW where
 * (C_i,h_i,w_i) = dm_dift_shape().  If mean_out_dev != NULL also writes the ensemble mean over
 * every consecutive group of `ensemble` samples (SDFeaturizer.forward, dift.py:231) as
 * [batch/ensemble, C_i, h_i, w_i] fp32. */
int dm_dift(dm_engine* e, const void* noisy_dev, const int64_t* t_dev, const int32_t* slot_dev,
            int batch, int h, int w, int up_ft_index, void* feat_out_dev,
            void* mean_out_dev, int ensemble, void* stream);
int dm_dift_shape(int h, int w, int up_ft_index, int* c_out, int* h_out, int* w_out);

/* Consumers' reductions of the loss grid (cluster.py:125-137, xray/compute.py:210-218) on device:
 * loss_dev [n_draws, n_cond, 4, h, w] fp32 (or fp16 when loss_is_f16) ->
 *   map_out_dev [h,w] fp32 = mean_N( mean_C(L[:,n_cond-1]) - mean_C(L[:,0]) ), and
 *   scalar_out_dev [1] fp32 = mean over pixels of the map (T(x|c)).  Either may be NULL. */
int dm_reduce_typicality(dm_engine* e, const void* loss_dev, int loss_is_f16, int n_draws, int n_cond,
                         int h, int w, void* map_out_dev, void* scalar_out_dev, void* stream);

/* The same for the grids of n_images images in one launch: maps_out_dev [n_images,h,w] fp32 (required),
 * scalars_out_dev [n_images] fp32 (optional).  cond_major = 0: loss_dev is [n_images, n_draws, n_cond, 4, h, w]
 * (the reference's grids back to back); cond_major = 1: [n_cond, n_images, n_draws, 4, h, w], i.e. the rows
 * dm_score_conds writes when its n_draws argument is n_images * n_draws (image-major draws) - no transpose needed. */
int dm_reduce_typicality_batched(dm_engine* e, const void* loss_dev, int loss_is_f16, int n_images, int n_draws,
                                 int n_cond, int h, int w, int cond_major, void* maps_out_dev, void* scalars_out_dev,
                                 void* stream);

/* Image-space form of the same reduction, as `Cluster.load_typicality` (cluster.py:125-137) and
 * `Typicallity.compute` (xray/compute.py:210-218) produce it: the latent map is resized with
 * bilinear interpolation (align_corners=False) to the image size (img_h, img_w) and averaged over
 * every kx x ky window (AvgPool2d((kx, ky), stride 1); kx = ky = 1 gives the per-pixel X-ray map).
 * out_dev [img_h-kx+1, img_w-ky+1] fp32; work_dev: scratch of (h*w + img_h*img_w) fp32. */
int dm_typicality_image(dm_engine* e, const void* loss_dev, int loss_is_f16, int n_draws, int n_cond,
                        int h, int w, int img_h, int img_w, int kx, int ky, void* work_dev, void* out_dev,
                        void* stream);

/* The consumers' normalisations of an image-space map (fp32 [n] on the device, e.g. dm_typicality_image's output with
 * kx = ky = 1), in numpy's fp32 arithmetic:
 *   DM_NORM_SIGNED    `normalize(dm)` of diffmining/typicality/cluster.py:32-47 as `Cluster.load_typicality_norm`
 *                     (cluster.py:112-123) calls it: negatives / |min|, positives / max, (dm + 1) / 2
 *   DM_NORM_MAXABS    `dm / np.max(np.abs(dm))`: `d_compute` (diffmining/typicality/utils.py:122-134) and utils.py:14-20
 *   DM_NORM_POSITIVE  positive_only=True (cluster.py:39-42, utils.py:16-19): max(dm, 0) / max(max(dm, 0))
 *   DM_NORM_SPLIT     positive_only='split' (cluster.py:34-36): d = dm / |max(dm)|; out = clip(d, 0, 1), out_neg = -clip(d, -1, 0)
 * out_dev (and out_neg_dev for DM_NORM_SPLIT) fp32 [n]; out_dev may alias map_dev.  work_dev: 2 floats of scratch. */
enum { DM_NORM_SIGNED = 1, DM_NORM_MAXABS = 2, DM_NORM_POSITIVE = 3, DM_NORM_SPLIT = 4 };
int dm_normalize_map(dm_engine* e, const void* map_dev, int64_t n, int mode, void* work_dev, void* out_dev,
                     void* out_neg_dev, void* stream);

/* Profiling support for bench.py: when enabled, every launch of the dominant (implicit-GEMM)
 * kernel is bracketed by hipEvents on the launch stream.  dm_prof_read synchronises and returns
 * the accumulated kernel milliseconds, launch count and algorithmic FLOPs since the last reset. */
int dm_prof_enable(dm_engine* e, int on);
/* Measurement only (bench.py's roofline; no product path calls it): the rate the matrix cores of this device sustain on
 * v_mfma_f32_16x16x32_f16 alone — the igemm tile's MFMA stream (8 waves per CU, 80 MFMAs per wave and step, 160 accumulator registers)
 * with nothing else in the loop, `steps` steps per CU, on random fp16 operands (zero_operands = 0) or zeros (1).  At the package power
 * cap the clock this returns is well below the nominal 2.4 GHz and depends on the operand statistics: it is the ceiling a
 * GEMM-shaped kernel can reach on real data (csrc/probe_peak.hip).  sclk_ghz (optional) = s_memtime ticks of one block per ns of wall time:
 * a diagnostic only — on the boxes of r06 it read ~0.5 x the clock amdsmi reports, so bench.py does not print it. */
int dm_measure_mfma_rate(void* stream, int steps, int zero_operands, double* tflops, double* sclk_ghz);
int dm_prof_read(dm_engine* e, double* igemm_ms, double* igemm_flops, int64_t* igemm_launches,
                 double* attn_ms, double* attn_flops, int64_t* attn_launches);
/* igemm_flops above counts the multiply-adds the launches EXECUTE.  With "up_fold" an Upsample2D + conv launch executes 4 / 9 of the
 * layer's definition (interpolate, then 9 taps: the count SURVEY 8d's 803.27 GFLOP per forward uses); this returns the difference
 * (definition minus executed) accumulated over the interval the last dm_prof_read closed, so a caller can quote either figure. */
int dm_prof_read_folded(dm_engine* e, double* igemm_flops_folded);

/* ---- VAE encoder -------------------------------------------------------------
 * Replaces `self.vae.encode(x).latent_dist.sample() * self.vae.config.scaling_factor`
 * (diffmining/typicality/compute.py:91-93, called at :137; dift.py:187) so the engine can start from
 * pixels.  Optional: an engine without VAE weights still scores latents.
 *
 * dm_engine_load_vae_weight: one tensor of `AutoencoderKL.state_dict()` by its diffusers name
 *   (`encoder.*`, `quant_conv.*`; an optional `vae.` prefix is stripped; `decoder.*` and
 *   `post_quant_conv.*` are accepted and ignored; the pre-0.15 attention names query/key/value/proj_attn
 *   and their [C,C,1,1] shapes are mapped to to_q/to_k/to_v/to_out.0).  Host memory, DM_F16 or DM_F32.
 * dm_engine_finalize_vae: checks the 108 encoder tensors (34,163,664 parameters), packs, uploads.
 * dm_vae_encode: image [batch,3,H,W] fp16 NCHW in [-1,1] (H, W >= 8; sizes that are not multiples of 8 floor at each of
 *   the three stride-2 stages like diffusers' Downsample2D(padding=0): the latent is floor(H/8) x floor(W/8)); `draws_per_image` D
 *   posterior samples per image from noise [batch*D,4,H/8,W/8] fp16 = the injected N(0,1) draws of
 *   `latent_dist.sample()` (NULL with D = 1 -> posterior mode, mean only).  D = 8 is the DIFT ensemble
 *   (dift.py:187,220 encode the same image 8 times; here the encoder runs once per image).
 *   Outputs (each optional, at least one): latents fp16 / fp32 [batch*D,4,H/8,W/8] (sample b*D+d belongs to
 *   image b) = (mean + exp(0.5 clamp(logvar,-30,20)) * noise) * scaling_factor, and the fp32 moments
 *   [batch,8,H/8,W/8] (mean | logvar) of `quant_conv(encoder(image))`.                               */
int dm_engine_load_vae_weight(dm_engine* e, const char* name, const void* host_ptr, int dtype,
                              const int64_t* shape, int ndim);
int dm_engine_finalize_vae(dm_engine* e);
int dm_vae_encode(dm_engine* e, const void* image_dev, const void* noise_dev, int batch, int draws_per_image,
                  int H, int W, float scaling_factor, void* latent_f16_dev, void* latent_f32_dev,
                  void* moments_f32_dev, void* stream);

/* ---- CLIP text tower ----------------------------------------------------------
 * Replaces `self.clip(tokens.to(self.device))[0]` of `CategoryFeatures.embed`
 * (diffmining/typicality/compute.py:51; the pipeline's `text_encoder`, compute.py:68): CLIP ViT-L/14 text
 * model, token ids -> last_hidden_state.  Optional.  Tokenisation (BPE vocabulary) stays on the host.
 *
 * dm_engine_load_clip_weight: one tensor of `CLIPTextModel.state_dict()` (optional `text_encoder.` /
 *   `text_model.` prefixes are stripped; `position_ids` buffers are ignored).  Host memory, DM_F16 / DM_F32.
 * dm_engine_finalize_clip: checks the 196 tensors (123,060,480 parameters), packs, uploads.
 * dm_clip_encode: input_ids [n_prompts][seq_len = 77] int32 on the device (tokenizer output with
 *   padding="max_length") -> last_hidden_state [n_prompts][77][768] as fp16 and/or fp32 (the reference
 *   takes `.float()`).  The fp16 output is directly the `ctx` of dm_engine_set_prompts.               */
int dm_engine_load_clip_weight(dm_engine* e, const char* name, const void* host_ptr, int dtype,
                               const int64_t* shape, int ndim);
int dm_engine_finalize_clip(dm_engine* e);
int dm_clip_encode(dm_engine* e, const int32_t* input_ids_dev, int n_prompts, int seq_len, void* out_f16_dev,
                   void* out_f32_dev, void* stream);

/* ---- DIFT patch descriptors ---------------------------------------------------
 * Replaces the per-patch tail of `Cluster.compute_embeddings` (diffmining/typicality/cluster.py:291-299):
 *   emb = emb[:, int(x0*H):int(x1*H), int(y0*W):int(y1*W)].mean(axis=(1,2)); emb / np.linalg.norm(emb)
 * for all patches of an image in one launch, from the ensemble-mean feature map dm_dift wrote
 * (feat [C,h,w] fp32).  boxes [n_patches][4] int32 = (r0, r1, c0, c1) in feature cells (numpy slice
 * semantics: clamped to the map; an empty window yields NaN).  out [n_patches][C] fp32.                 */
int dm_patch_embed(dm_engine* e, const void* feat_f32_dev, int C, int h, int w, const int32_t* boxes_dev,
                   int n_patches, void* out_f32_dev, void* stream);

/* Bytes of device memory currently held by the engine: packed weights (U-Net + optional VAE / CLIP slabs) and
 * the workspace arena (the per-prompt K/V cache, a few MB, is not included). */
int dm_engine_memory(dm_engine* e, size_t* weights_bytes, size_t* arena_bytes);

/* Steady-state contract ("no allocation on the steady-state path").  The engine allocates device memory in
 * three places only: weights at finalize, the per-prompt K/V cache when more prompts are registered than it holds
 * (capacity >= 16, doubling), and the workspace arena when a call needs more than it holds (exact peak from a dry run of
 * the schedule, done once per (batch, shape) and cached).  dm_engine_reserve pre-sizes the latter two for the largest
 * call the caller will make — a U-Net batch of up to `max_batch` samples (cut into DM_CHUNK-sized runs like the calls
 * themselves) of h x w latents scored under n_cond prompts per draw (n_cond <= 1: dm_score / dm_unet_forward / dm_dift),
 * and `max_prompts` registered prompts (call it before dm_engine_set_prompts: growing the cache drops its rows) — after
 * which no call within those bounds allocates.  dm_engine_stats reports how many device allocations and schedule dry runs
 * the engine has done so far, so a caller (tests/test_gpu_e2e.py::test_no_allocation_in_steady_state) can assert it, and how many
 * U-Net runs were replays of a captured hipGraph (option "graph" = 1: a run whose schedule key and pointer arguments repeat is
 * captured on its second occurrence and replayed from then on; needs a caller stream other than the legacy default stream and is
 * bypassed while dm_prof_enable is on, because a replay carries no per-launch events). */
int dm_engine_reserve(dm_engine* e, int max_batch, int max_h, int max_w, int n_cond, int max_prompts, void* stream);
int dm_engine_stats(dm_engine* e, int64_t* device_allocs, int64_t* schedule_dry_runs, int64_t* graph_launches);

/* ---- operator-level entry points ------------------------------------------------------------------
 * The individual gfx950 kernels behind the U-Net, exposed so that every op can be parity-tested
 * against the matching torch.nn.functional op.  All buffers are device
 * pointers; activations are NHWC fp16; weights are in the engine's packed layout:
 *   igemm : Y[m][co] = sum_k X~[m][k] Wp[co][k], k = (tap, cin); Wp [Cout][taps*Cin] fp16.
 *           mode 0 dense (1x1 conv / Linear), 1 conv3x3 s1 p1, 2 conv3x3 s2 p1,
 *           3 conv3x3 on the nearest-upsampled (OH x OW) image, 4 conv3x3 s2 on F.pad(x,(0,1,0,1))
 *           (VAE Downsample2D(padding=0)).  Cout must be a multiple of 160 (80-channel waves) or of
 *           128 (64-channel waves; VAE).  X2 = optional second source
 *           (channel concat cat([X, X2]), as the up blocks do).  epi 1 = GEGLU on quad-interleaved rows.
 *   attention : softmax(Q K^T * scale) V per head, head_dim D in {40, 80, 160}; strides in elements.
 *   groupnorm : (optionally concat) GroupNorm(G) (+SiLU), fp32 statistics; gamma/beta fp32.
 *   layernorm : per-row LayerNorm over C; gamma/beta fp32.                                        */
int dm_op_igemm(void* stream, const void* X, const void* X2, const void* Wp, const void* bias, const void* temb,
                const void* res, void* Y, int N, int H, int W, int C1, int C2, int Cout, int OH, int OW,
                int mode, int epi, int temb_ld);
/* Runtime switches for A/B measurements (process-wide; each defaults to the measured best and is initialised from the
 * environment variable DM_<NAME>).  Returns 1 for an unknown name and 2 for a value outside the switch's set (nothing changes then:
 * 0 / 1 / 2 for every switch, -1 too for "igemm_big", the list below for "attn_pipe"); an environment value outside the set is
 * reported on stderr and the default is kept.
 *   bit-neutral (the same arithmetic in the same order, asserted by tests/test_gpu_ops.py): "igemm_big" (-1 per shape /
 *     0 / 1: which tile geometry), "igemm_tail" (head / tail row split), "igemm_splitk" 0 vs 1 only for layers that
 *     do not split (a split layer sums its k parts in fp32 in a different order than the unsplit k loop);
 *   numerically equivalent but NOT bit-identical: "igemm_splitk" = 2 (four k parts instead of three tap-aligned ones),
 *     "ln_fold" (LayerNorm folded into the next GEMM: the rounding moves from the LN output to the folded weights),
 *     "attn_pipe" / "attn_cross" (pipelined / one-pass-softmax kernels vs the generic online-softmax kernel: different
 *     rescaling points), "ln_stats_g" (different lane order of the row reductions), "ln_inkernel" (0 / 1 / 2: LayerNorm
 *     statistics from a statistics kernel (two-pass) or inside the folded GEMM (one-pass fp32 sums; 1 = where cheaper));
 *   "gn_fold" (1 / 0): Transformer2D.norm folded into proj_in where the per-sample weights are <= 1/8 of a sample's activations (Cout * 8 <= H * W:
 *     the 320-channel level of latents >= 64x64, the 640-channel level of latents >= 144x144; per-sample weights; the rounding
 *     moves from the normalised activations to the scaled weights) — numerically equivalent, not bit-identical to 0;
 *   "sc_fold" (1 / 0): the ResNet blocks' conv_shortcut folded into conv2 as extra k steps wherever conv2 runs unsplit (one
 *     rounding of the sum instead of three) — numerically equivalent, not bit-identical to 0;
 *   "ff_fold" (1 / 0): ff.net.2 + residual + proj_out of a transformer block as one GEMM with the pre-multiplied weights Wp W2
 *     (the [tokens x C] intermediate and its fp16 rounding disappear) — numerically equivalent, not bit-identical to 0;
 *   "tap_reuse" (1 / 0 / 2): the 3x3 stride-1 convolutions of 64-pixel-wide images and the time-embedding ones (ResnetBlock2D.conv1)
 *     of 32-pixel-wide images (2: every eligible layer down to 16 pixels) run their k steps in the order
 *     (dy, 64-channel slab, dx) and fetch ONE activation stage per three horizontal taps (igemm_pers_tr.hip; the 128-row tile
 *     follows the same order, so the two tile kernels stay bit-identical) — numerically equivalent, not bit-identical to 0;
 *   "up_fold" (1 / 0): Upsample2D (nearest, exact 2x) + its 3x3 convolution as four 2x2 convolutions on the low-resolution source, one per
 *     output parity class, with the 3x3 taps that read the same source pixel pre-summed in fp32 and rounded to fp16 once (4 / 9 of the
 *     layer's MACs; the up-sampled tensor is never formed) — numerically equivalent, not bit-identical to 0; other sizes
 *     (F.interpolate(size=...) of odd latents) always run the unfolded layer;
 *   "q_once" (1 / 0 / 2): dm_score_conds — attn2.to_q of the first transformer block runs once per draw (its input is the same under every
 *     prompt) and the cross-attention reads the queries modulo the draw count; 2 also runs the two GEMMs that read the prefix's
 *     outputs as a residual once per prompt block against the per-draw rows, so two of the three stacking copies of the shared
 *     prefix disappear (measured +-0, hence not the default) — all bit-identical to 0;
 *   "attn_pipe" (1; 0 / 2 / 3 / 5 / 9 / 10 / 12 — any other value is refused): the head_dim-40 / 80 self-attention kernels: 1 = head_dim 40
 *     on attention_qk32.hip (r06: scores on 32x32x16 MFMAs, P moved to the PV layout by v_permlane16_swap), head_dim 80 on the
 *     software-pipelined kernel, and from 8192 keys the three-wave-set anti-phase kernel (attention_pp.hip, r05); 5 = attention_qk32.hip
 *     wherever it applies; 9 = the r04 pipelined kernels everywhere; 0 = the generic kernel; 2 = pipelined head_dim 40 only; 3 = head_dim 80
 *     with constant-chunk rows (A/B); 12 / 10 = the anti-phase kernel everywhere (three sets with / without static priorities).  9, 2, 10 and
 *     12 are bit-identical to each other for head_dim 40; 1 / 5 differ from them by the summation order of a score (k = 48 in one fp32
 *     chain instead of 64), at the same distance from fp32 SDPA.  (The two-set kernel and the timing-only ablation instantiations of
 *     r05 exist only in a -DDM_ATTN_PP_ABLATE debug build.)
 *   "gn_epi" (1 / 0): 1 = norm2's GroupNorm statistics as per-(64-row block, channel pair) sums written by conv1's epilogue where the
 *     persistent kernels run it and computed from conv1's output where they do not (bit-identical between the two, so independent of
 *     the batch); 0 = the statistics pass.  Numerically equivalent, not bit-identical to 0 (another summation order); -0.3 ms per step
 *     (DESIGN.md 4g);
 *   "conv_out_rows" (1 / 0): conv_out + eps-MSE with the input rows staged once in LDS and walked by all nine taps on the matrix cores
 *     (conv_out.hip) instead of the per-pixel gather (misc.hip) — equal to fp32 rounding, not bit-identical;
 *   "gn_skip" (1 / 0): 1 = the up path's norm1 over cat([x, skip]) sums x only and merges the skip's GroupNorm partial sums kept from the
 *     down path's norm1 of the same tensor (one read of the skip less, where the group widths nest) — numerically equivalent, not
 *     bit-identical to 0; a property of the network's channel counts, never of the batch;
 *   "graph" (0 / 1): replay whole U-Net runs as captured hipGraphs (bit-identical: the same kernels with the same arguments);
 *   "igemm_exp": experimental kernel paths of the current round (0 = shipped). */
int dm_set_option(const char* name, int value);
/* current value of a switch (bench.py prints the arithmetic rewrites a line was measured with); nonzero for an unknown name */
int dm_get_option(const char* name, int* value);

/* which tile geometry dm_op_igemm runs a shape on: 0 = 128-row tile (128x320 / 128x160), 1 = persistent 256x320 tile */
int dm_op_igemm_tile(int M, int Cin, int Cout, int mode);
/* rows [0, r) of such a launch run on the persistent 256x320 tile (whole rounds over the CUs), rows [r, M) on the
 * 128-row tile ("igemm_tail"; r = 0 / M: one kernel for all rows).  `spatial` = OH*OW of one sample (ignored for mode 0). */
int dm_op_igemm_head_rows(int M, int spatial, int Cin, int Cout, int mode);
int dm_op_attention(void* stream, const void* Q, const void* K, const void* V, void* O, int ldq, int ldk, int ldv,
                    int ldo, int64_t bsq, int64_t bsk, int64_t bsv, int64_t bso, const int32_t* kv_slot,
                    int B, int heads, int Tq, int Tk, int D, float scale);
/* LayerNorm folded into a Linear (the transformer blocks' LN -> to_q/k/v, LN -> to_q, LN -> GEGLU projection):
 * dm_op_ln_stats writes (mean, rstd) per row of X [rows][C]; dm_op_igemm_ln computes
 *   Y[m][c] = rstd_m * (sum_k X[m][k] Wp[c][k] - mean_m * ln_s[c]) + ln_t[c]     (then GEGLU when epi = 1)
 * with Wp = fp16(W * gamma), ln_s[c] = sum_k Wp[c][k], ln_t[c] = sum_k W[c][k] beta[k] + bias[c]  (fp32).   */
int dm_op_ln_stats(void* stream, const void* X, int rows, int C, float eps, void* stats_f32);
int dm_op_igemm_ln(void* stream, const void* X, const void* Wp_folded, const void* ln_s, const void* ln_t,
                   const void* stats, void* Y, int M, int Cin, int Cout, int epi);
/* igemm with the k range cut into `ksplit` parts (>= 2, dividing the k steps) through an fp32 workspace of
 * ksplit * M * Cout floats, followed by the reduction + epilogue; the engine uses it for the 8x8 layers.   */
int dm_op_igemm_splitk(void* stream, const void* X, const void* X2, const void* Wp, const void* bias, const void* temb,
                       const void* res, void* Y, int N, int H, int W, int C1, int C2, int Cout, int OH, int OW,
                       int mode, int temb_ld, int ksplit, void* workspace_f32);
/* single-head attention with head_dim 512 (VAE mid block): Q/K/V [B][T][ld], O [B][T][ldo] */
int dm_op_attention512(void* stream, const void* Q, const void* K, const void* V, void* O, int B, int T, int ld,
                       int ldo, float scale);
int dm_op_groupnorm(void* stream, const void* X, const void* X2, int N, int HW, int C, int C1, int G, float eps,
                    const float* gamma, const float* beta, int silu, void* Y);
int dm_op_layernorm(void* stream, const void* X, int rows, int C, const float* gamma, const float* beta, float eps,
                    void* Y);
/* GroupNorm statistics as per-(64-row block, channel pair) fp32 sums (r05): blocks [rows / 64][C], entry (b, 2 k + {0, 1}) = (sum, sum of
 * squares) of channels 2 k, 2 k + 1 over rows 64 b .. 64 b + 63.  dm_op_gn_blocks computes the blocks of rows [row0, rows) from X [rows][C];
 * dm_op_conv_temb_gn_blocks is ResnetBlock2D.conv1 (3x3, + bias, + time-embedding row temb [N][temb_ld]) whose persistent kernels write the
 * blocks of the first *rows_done rows of Y from their epilogue (bit-identical to dm_op_gn_blocks on Y; 0 rows when another tile kernel takes
 * the launch); dm_op_groupnorm_blocks = the fixed-order fp64 combine + the apply (+ SiLU) — the path norm2 takes in the engine
 * (option "gn_epi").  rows % 64 == 0, HW % 64 == 0, C % 16 == 0, C / G even. */
/* conv_out (3x3, C0 -> 4, + bias, rounded to fp16 = `unet(...).sample`) on Xn [B,H,W,C0] fp16 NHWC with w [4][9*C0] (k = (tap, cin)): pred
 * [B,4,H,W] fp16 (optional) and, with eps [B,4,H,W] fp32, loss [B,4,H,W] fp32 = (pred - eps)^2 — the last two steps of SD.compute_loss
 * (compute.py:100-101).  Option "conv_out_rows" selects the kernel (1: input rows staged in LDS, conv_out.hip; 0: per-pixel gather). */
int dm_op_conv_out(void* stream, const void* Xn, const void* w, const void* bias, const float* eps, int B, int H, int W, int C0, float* loss,
                   void* pred);
int dm_op_conv_temb_gn_blocks(void* stream, const void* X, const void* Wp, const void* bias, const void* temb, void* Y, int N, int H, int W,
                              int Cin, int Cout, int temb_ld, float* blocks, int* rows_done);
int dm_op_gn_blocks(void* stream, const void* X, int rows, int C, int row0, float* blocks);
int dm_op_groupnorm_blocks(void* stream, const void* X, const float* blocks, int N, int HW, int C, int G, float eps, const float* gamma,
                           const float* beta, int silu, void* Y);
/* A GEMM with a second GEMM on another tensor folded into its k loop: after its own taps on X [N,H,W,Cin] (mode 1: 3x3 stride 1;
 * mode 0: dense) the loop runs a 1x1 convolution on cat([X3 (C3 channels), X4 (C4 channels)]) (same N, H, W).  Wp [Cout][taps*Cin + C3 + C4]:
 * the first GEMM's row (k = (tap, cin)) followed by the second's; bias = the sum of both; res = optional residual [M][Cout].
 * Used for ResnetBlock2D (out = conv_shortcut(x) + conv2(h)) and for ff.net.2 + residual + proj_out as one GEMM
 * ((Wp W2) f + Wp t2 + x).  Cout % 160 == 0; Cin, C3, C4 multiples of 64. */
int dm_op_igemm_shortcut(void* stream, const void* X, const void* X3, const void* X4, const void* Wp, const void* bias, const void* res,
                         void* Y, int N, int H, int W, int Cin, int C3, int C4, int Cout, int mode);
/* Upsample2D + conv (diffusers Upsample2D.forward: F.interpolate(scale_factor=2, mode="nearest") then Conv2d(3x3, padding 1)) folded onto
 * the source grid.  dm_op_fold_upconv_weights (host): w [Cout][Cin][3][3] fp16 -> W4 [4 = py*2+px][Cout][(a*2+b)*Cin + ci] fp16, entry =
 * fp16(fp32 sum of the 3x3 taps that read source pixel (y-1+py+a, x-1+px+b) for output pixel (2y+py, 2x+px)).
 * dm_op_upconv_folded (device): X [N][H][W][Cin], W4, bias [Cout] -> Y [N][2H][2W][Cout].  Cout % 320 == 0, Cin % 64 == 0. */
int dm_op_fold_upconv_weights(const void* w_oihw_f16_host, int Cout, int Cin, void* out_f16_host);
int dm_op_upconv_folded(void* stream, const void* X, const void* W4, const void* bias, void* Y, int N, int H, int W, int Cin, int Cout);
/* GroupNorm(G, eps) (no activation) folded into the following 1x1 convolution W [Cout][C] + bias (Transformer2DModel.norm ->
 * proj_in): statistics of X [N][HW][C], per-sample weights fp16(W diag(a_n)) and fp32 bias rows W b_n + bias, then the GEMM on
 * the raw X — Y [N][HW][Cout] fp16.  HW must be a multiple of 128, C of 64, Cout of 160. */
int dm_op_groupnorm_conv1x1(void* stream, const void* X, int N, int HW, int C, int G, float eps, const float* gamma,
                            const float* beta, const void* W, const void* bias, int Cout, void* Y);

/* ---- fp32 U-Net: the arithmetic of the reference's DIFT featuriser ------------------------------------------------------
 * `SDFeaturizer.__init__` (diffmining/typicality/dift.py:197-199) loads the pipeline WITHOUT torch_dtype and `forward`
 * (dift.py:214-232 -> OneStepSDPipeline.__call__, :173-193 -> MyUNet2DConditionModel.forward, :24-169) runs WITHOUT autocast:
 * every tensor and product of the DIFT path is fp32, unlike the typicality path (compute.py:98, fp16 autocast) that dm_engine
 * serves.  dm_f32_net is that second arithmetic: fp32 weights (values given as fp16 are widened exactly), fp32 NHWC
 * activations, fp32 matrix-core GEMMs (v_mfma_f32_16x16x4_f32, exact fp32 products) and attention.  Same U-Net, same
 * diffusers-named state dict, same prompt-slot mechanism as dm_engine; all *_dev tensors at this boundary are fp32.
 *   dm_f32_load_weight / dm_f32_finalize   replace from_pretrained's state-dict load (dift.py:197-199)
 *   dm_f32_set_prompts   ctx_dev [n_prompts,77,768] fp32 = `prompt_embeds` (dift.py:222-227); K/V of the 16 blocks per prompt
 *   dm_f32_dift          MyUNet2DConditionModel.forward(latents_noisy, t, [up_ft_index], prompt_embeds) (dift.py:24-169,191):
 *                        feat_out_dev [batch,C_i,h_i,w_i] fp32 NCHW and / or the mean over consecutive groups of `ensemble`
 *                        samples [batch/ensemble,C_i,h_i,w_i] (dift.py:231); shapes from dm_dift_shape()
 *   dm_f32_unet_forward  `unet(sample, t, encoder_hidden_states).sample` in fp32: out_dev [batch,4,h,w] — the full-size fp32
 *                        ground truth on the GPU for the fp16 engine's deviation measurements
 *   dm_f32_prof_*        as dm_prof_*: HIP events around every GEMM / attention launch, for bench.py's roofline leg */
typedef struct dm_f32_net dm_f32_net;
int dm_f32_create(int device, dm_f32_net** out);
void dm_f32_destroy(dm_f32_net* e);
const char* dm_f32_last_error(dm_f32_net* e);   /* e may be NULL: last create() error */
int dm_f32_load_weight(dm_f32_net* e, const char* name, const void* host_ptr, int dtype, const int64_t* shape, int ndim);
int dm_f32_finalize(dm_f32_net* e);
int dm_f32_set_prompts(dm_f32_net* e, const void* ctx_dev, int n_prompts, void* stream);
int dm_f32_unet_forward(dm_f32_net* e, const void* sample_dev, const int64_t* t_dev, const int32_t* slot_dev, int batch,
                        int h, int w, void* out_dev, void* stream);
int dm_f32_dift(dm_f32_net* e, const void* noisy_dev, const int64_t* t_dev, const int32_t* slot_dev, int batch, int h, int w,
                int up_ft_index, void* feat_out_dev, void* mean_out_dev, int ensemble, void* stream);
/* SD.compute_loss (compute.py:95-102) with NO autocast: fp32 add_noise (x_dev [n_x,4,h,w], x_index_dev [batch] or NULL for identity,
 * eps_dev [batch,4,h,w], t_dev [batch]), fp32 U-Net, fp32 squared error -> loss_out_dev [batch,4,h,w] fp32: the exact-arithmetic
 * yardstick of dm_score (bench.py's `score_deviation`, tools/t_deviation_gpu.py). */
int dm_f32_score(dm_f32_net* e, const void* x_dev, const int32_t* x_index_dev, const void* eps_dev, const int64_t* t_dev,
                 const int32_t* slot_dev, int batch, int n_x, int h, int w, void* loss_out_dev, void* stream);
/* optional AutoencoderKL encoder in fp32: `pipe.vae.encode(img_tensor).latent_dist.sample() * scaling_factor` of the featuriser
 * (dift.py:187) — image_dev [batch,3,H,W] fp32 in [-1,1]; noise_dev [batch*draws_per_image,4,H/8,W/8] fp32 N(0,1) draws or NULL
 * (posterior mode); latent_dev [batch*draws_per_image,4,H/8,W/8] fp32 and / or moments_dev [batch,8,H/8,W/8] fp32 (either may be
 * NULL).  State-dict names as dm_engine_load_vae_weight (decoder tensors ignored, legacy attention names accepted). */
int dm_f32_load_vae_weight(dm_f32_net* e, const char* name, const void* host_ptr, int dtype, const int64_t* shape, int ndim);
int dm_f32_finalize_vae(dm_f32_net* e);
int dm_f32_vae_encode(dm_f32_net* e, const void* image_dev, const void* noise_dev, int batch, int draws_per_image, int H, int W,
                      float scaling_factor, void* latent_dev, void* moments_dev, void* stream);
/* optional CLIP ViT-L/14 text tower in fp32: `pipe.encode_prompt(prompt, ...)[0]` of the featuriser (dift.py:222-226) — its pipeline is built
 * with no torch_dtype (dift.py:197-199), so `text_encoder(input_ids)[0]` is fp32 there; replaces transformers' CLIPTextModel.forward for that
 * call (12 layers, causal attention over 77 tokens, quick-GELU, final LayerNorm; fp32 GEMMs on the fp32 matrix cores).  State-dict names as
 * dm_engine_load_clip_weight (196 tensors; `text_model.` / `text_encoder.` prefixes and `position_ids` accepted).  input_ids_dev
 * [n_prompts, 77] int32 (tokenizer output, padding="max_length"); out_f32_dev [n_prompts, 77, 768] fp32 last_hidden_state. */
int dm_f32_load_clip_weight(dm_f32_net* e, const char* name, const void* host_ptr, int dtype, const int64_t* shape, int ndim);
int dm_f32_finalize_clip(dm_f32_net* e);
int dm_f32_clip_encode(dm_f32_net* e, const int32_t* input_ids_dev, int n_prompts, int seq_len, void* out_f32_dev, void* stream);
int dm_f32_prof_enable(dm_f32_net* e, int on);
int dm_f32_prof_read(dm_f32_net* e, double* gemm_ms, double* gemm_flops, int64_t* gemm_launches, double* attn_ms,
                     double* attn_flops, int64_t* attn_launches);
int dm_f32_memory(dm_f32_net* e, size_t* weights_bytes, size_t* arena_bytes);
/* operator-level entry points of the fp32 kernels (parity tests): NHWC fp32 tensors; modes as dm_op_igemm (0 dense, 1 conv3x3,
 * 2 stride 2, 3 nearest-upsample to (OH, OW) + conv3x3, 4 stride 2 pad (0,1,0,1)); Wp [Cout][taps*Cin] k = (tap, cin); res [M][Cout] */
int dm_f32_op_gemm(void* stream, const void* X, const void* X2, const void* Wp, const void* bias, const void* temb, const void* res,
                   void* Y, int N, int H, int W, int OH, int OW, int Cin, int C1, int Cout, int mode, int temb_ld);
int dm_f32_op_attention(void* stream, const void* Q, const void* K, const void* V, void* O, int ldq, int ldk, int ldv, int ldo,
                        int64_t bsq, int64_t bsk, int64_t bsv, int64_t bso, const int32_t* kv_slot, int n_slots, int B, int heads,
                        int Tq, int Tk, int D, float scale);
int dm_f32_op_groupnorm(void* stream, const void* X, const void* X2, int N, int HW, int C, int C1, int G, float eps,
                        const float* gamma, const float* beta, int silu, void* stats_work, void* Y);
int dm_f32_op_layernorm(void* stream, const void* X, int rows, int C, const float* gamma, const float* beta, float eps, void* Y);

#ifdef __cplusplus
}
#endif
#endif /* DM_ENGINE_H */
