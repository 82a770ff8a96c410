// dm_kernels.h — launcher prototypes of the hand-written gfx950 kernels (internal, C++).
// The public boundary is include/dm_engine.h.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <atomic>
#include <vector>

// order of the four pixel-fragment MFMAs of a weight-fragment group in the igemm tiles: 1 = alternating 0..3 / 3..0 (one operand
// changes per MFMA), 0 = always 0..3 (r01-r05); same bits either way
#ifndef DM_MFMA_SNAKE
#define DM_MFMA_SNAKE 1
#endif

namespace dm {

typedef _Float16 f16;

// Function attributes (dynamic LDS size) are per device: true the first time the caller's launcher runs on the current
// device (`seen` = one bit per device ordinal), so a second engine on another GPU of the same process sets them too.
inline bool first_use_on_device(std::atomic<uint64_t>& seen) {
    int d = 0;
    (void)hipGetDevice(&d);
    const uint64_t bit = 1ull << (d & 63);
    return (seen.fetch_or(bit) & bit) == 0;
}

// compute units of the current device (cached per device ordinal)
inline int device_cu_count() {
    static std::atomic<int> n_cu[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    int v = n_cu[dev & 63].load(std::memory_order_relaxed);
    if (v == 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n_cu[dev & 63].store(v, std::memory_order_relaxed);
    }
    return v;
}

// Per-dispatch timing for bench.py's live roofline (dm_prof_enable).  While a LaunchTimer is installed on the calling thread, every
// dispatch of the igemm / attention launchers goes out through hipExtLaunchKernelGGL with its own (start, stop) event pair — the
// timestamps of the dispatch's own completion signal — instead of being bracketed by hipEventRecord calls, each of which is an extra
// barrier packet in the queue (r04: the bracketing cost 1.7 ms per 140 ms step).  The pairs of one launch_* call are
// collected in `pairs`; its duration is the sum of its dispatches' kernel times.  Events come from / return to the engine's pool.
struct LaunchTimer {
    std::vector<hipEvent_t>* pool;
    std::vector<hipEvent_t> pairs;      // start, stop, start, stop, ...
    hipError_t err = hipSuccess;
};
extern thread_local LaunchTimer* g_launch_timer;          // engine.hip
template <typename F, typename... Args>
inline void launch_timed(F kernel, const dim3& grid, const dim3& block, size_t lds, hipStream_t s, Args... args) {
    LaunchTimer* t = g_launch_timer;
    if (!t) { hipLaunchKernelGGL(kernel, grid, block, lds, s, args...); return; }
    hipEvent_t ev[2];
    for (hipEvent_t& h : ev) {
        if (!t->pool->empty()) { h = t->pool->back(); t->pool->pop_back(); }
        else { const hipError_t r = hipEventCreate(&h); if (r != hipSuccess) { t->err = r; hipLaunchKernelGGL(kernel, grid, block, lds, s, args...); return; } }
    }
    hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, s, ev[0], ev[1], 0, args...);
    t->pairs.push_back(ev[0]); t->pairs.push_back(ev[1]);
}

// Runtime switches (A/B measurements; every default is the measured best).  Initialised from the environment variable of
// the same name in upper case with a DM_ prefix (DM_IGEMM_PERSIST=0 ...), changeable through dm_set_option().
enum Option { OPT_IGEMM_BIG = 0, OPT_IGEMM_SPLITK, OPT_LN_FOLD, OPT_ATTN_PIPE, OPT_IGEMM_TAIL, OPT_ATTN_CROSS, OPT_LN_STATS_G, OPT_IGEMM_EXP, OPT_LN_INKERNEL, OPT_GRAPH, OPT_GN_FOLD, OPT_SC_FOLD, OPT_FF_FOLD, OPT_TAP_REUSE, OPT_UP_FOLD, OPT_Q_ONCE, OPT_GN_EPI, OPT_CONV_OUT_ROWS, OPT_GN_SKIP, OPT_COUNT };
int option(Option o);                       // engine.hip
unsigned options_epoch();                   // engine.hip: changes with every set_option() that changed a value
int set_option(const char* name, int value);   // 0 on success
int get_option(const char* name, int* value);  // 0 on success

// ---- K1/K2/K3: implicit-GEMM on MFMA (conv3x3 s1/s2/upsampled, 1x1 conv, linear) -------------
enum IGemmMode { IG_DENSE = 0, IG_CONV3 = 1, IG_CONV3_S2 = 2, IG_CONV3_UP = 3, IG_CONV3_S2P0 = 4, IG_CONV2_UP4 = 5 };
enum IGemmEpi { EPI_PLAIN = 0, EPI_GEGLU = 1 };

struct IGemmParams {
    const f16* X;        // source 1, NHWC [N,H,W,C1] (dense: [M,C1])
    const f16* X2;       // source 2 (channel concat), NHWC [N,H,W,Cin-C1], or nullptr
    const f16* Wp;       // packed weights [Cout][taps*Cin], K order = (tap, cin)
    const f16* bias;     // [Cout] or nullptr
    const f16* temb;     // per-sample channel add [N][temb_ld] (already offset), or nullptr
    const f16* res;      // residual [M][ldres] or nullptr
    f16* Y;              // output [M][ldy] (already offset to the channel slot)
    int M;               // output rows = N*OH*OW
    int Cout, Cin, C1;   // Cin = C1 + C2
    int H, W, OH, OW;    // source spatial dims / output spatial dims (dense: H=OH=1, W=OW=M)
    int mode, epi;
    int ldy, ldres, temb_ld;
    // split-K (small-M layers: too few tiles to fill the chip): > 1 -> the k range is cut into `ksplit` parts,
    // each block writes its fp32 partial tile to partial[split][M][Cout]; launch_igemm then runs the
    // reduction + epilogue (bias, time-embedding, residual, fp16 rounding points as in the fused epilogue)
    int ksplit = 0;
    float* partial = nullptr;
    // LayerNorm folded into the GEMM (transformer blocks: LN -> Linear): X is the un-normalised token matrix,
    // Wp = W * diag(gamma) (fp16), and the epilogue applies  y = rstd_r * (acc - mean_r * ln_s[c]) + ln_t[c]
    // with ln_s[c] = sum_k Wp[c][k], ln_t[c] = sum_k W[c][k] beta[k] + bias[c];  ln_stats [M][2] = (mean, rstd)
    const float* ln_stats = nullptr;    // nullptr with ln_s set: the kernel takes the row statistics itself from the k loop's LDS tiles
    const float* ln_s = nullptr;
    const float* ln_t = nullptr;
    float ln_eps = 1e-5f;
    // persistent 256 x 320 kernel: tile hand-out counters owned by the caller (IGEMM_TILE_CTR_INTS ints, zero between
    // launches; an engine passes its own, so engines / streams sharing a device never share counters).  nullptr: a
    // process-wide set per device, valid only while the launches of a device are serialised on one stream (the
    // operator-level entry points of the parity tests).
    int* tile_ctr = nullptr;
    // GroupNorm folded into a 1x1 convolution (r03, Transformer2D.norm -> proj_in): per-SAMPLE weights Wp + n * w_sample_stride
    // (= fp16(W diag(a_n)), a_n the sample's per-channel GroupNorm scale) and a per-sample fp32 bias row ln_t + n * Cout
    // (= W b_n + bias), n = row / rows_per_sample; dense mode only, every tile inside one sample, epilogue y = fp16(acc + t)
    long long w_sample_stride = 0;
    int rows_per_sample = 0;
    // ResNet shortcut folded into conv2 (r03): after the nine taps of the 3x3 convolution (source X, Cin channels) the k loop
    // continues with Csc / 64 steps of a 1x1 convolution on a SECOND tensor pair cat([X3, X4]) (the block's input, C3 + (Csc - C3)
    // channels, same spatial size) — weight rows are [9 * Cin | Csc], the bias is the sum of both.  mode IG_CONV3, no residual.
    const f16* X3 = nullptr;
    const f16* X4 = nullptr;
    int C3 = 0, Csc = 0;
    // persistent kernels: device address of the launcher's 1 KiB zero page (the source of halo / out-of-range LDS-DMA lanes), set by
    // the launcher.  As a kernel ARGUMENT it sits in an SGPR pair; taken from the global symbol inside the kernel it is a GOT load
    // (s_getpc + s_load_dwordx2 + s_waitcnt lgkmcnt(0) — which also waits for every fragment read in flight) that the compiler
    // re-materialises inside the k loop of the register-tight variants (r04: once per k step in the tap-reuse and folded-LayerNorm
    // kernels).
    const void* zero_page = nullptr;
    // GroupNorm block sums out of the epilogue (r05; the time-embedding launches = ResnetBlock2D.conv1, whose output is norm2's
    // input): [M / 64][Cout] fp32, entry (b, 2 k + {0, 1}) = (sum, sum of squares) of output channels 2 k, 2 k + 1 over rows
    // 64 b .. 64 b + 63, in the arithmetic gn_blocks_kernel (norm.hip) defines.  Needs M % 64 == 0; nullptr = not wanted.
    float* gn_blocks = nullptr;
    // host-side only (launch rules; no kernel reads it): the layer HAS a time embedding.  The engine's dry run walks the schedule with
    // data pointers that are bare arena offsets (temb may be null there for a real layer), and the rules that pick an allocation path
    // (igemm_gn_layer, tap_reuse_layer) must answer the same in both walks.  Last member: no other field moves.
    bool has_temb = false;
};
constexpr int IGEMM_TILE_CTR_INTS = 8 * 32 + 32;
// LayerNorm statistics taken inside the folded GEMM are one-pass fp32 sums (var = E[x^2] - mean^2): with |mean| >> std the
// subtraction cancels (measured r04: rel-L2 of the output 7.8e-4 instead of 2.9e-4 at |mean| / std = 100, equal at 30).  A row
// whose mean^2 exceeds LN_REDO_RATIO2 x var gets its variance re-taken exactly — sum of (x - mean)^2 over the row, read back
// from global memory (L2-hot: the tile just streamed it) — by the lane pair that owns it.  Rare path; the same code in
// igemm_pers_tile.h and igemm_tile.h, so the two tiles stay bit-identical.
constexpr float LN_REDO_RATIO2 = 256.0f;       // |mean| / std > 16
hipError_t launch_igemm(const IGemmParams& p, hipStream_t s);
// Upsample2D (nearest, exact 2x) + conv3x3 as four 2x2 convolutions on the source grid (igemm_pers_up.hip): X [N][H][W][Cin],
// Wp = fold_upconv_weights() [4][Cout][4 Cin], Y [N][2H][2W][Cout]; p.mode = IG_CONV2_UP4, p.H = p.OH = H, p.W = p.OW = W, p.M = N H W.
// hipErrorInvalidValue for shapes it does not take (Cout % 320, Cin % 64, 32-bit offsets): the caller keeps the unfolded layer.
hipError_t launch_igemm_pers_up4(const IGemmParams& p, hipStream_t s);
inline bool igemm_up4_ok(int N, int H, int W, int Cin, int Cout) {
    return Cout % 320 == 0 && Cin % 64 == 0 && H >= 1 && W >= 1 && H <= 511 && W <= 511 && N <= 8191 && (long long)N * H * W >= 2 &&
           (long long)N * H * W * Cin < (1LL << 31) && 16LL * Cout * Cin < (1LL << 32);
}
// host: conv weight [Cout][Cin][3][3] fp16 -> [4 = py * 2 + px][Cout][(a * 2 + b) * Cin + ci] fp16, each entry the fp32 sum of the 1, 2 or 4
// taps (dy, dx) of the 3x3 kernel that read source pixel (y - 1 + py + a, x - 1 + px + b), rounded to fp16 once
void fold_upconv_weights(const f16* w_oihw, int cout, int cin, f16* out);
// number of k parts for a layer with `spatial` output positions per sample (1 = no split); batch independent;
// the caller provides the workspace
int igemm_splitk_parts(const IGemmParams& p, int spatial);
int igemm_tile_choice(const IGemmParams& p);     // 0 = 128-row tile, 1 = 256 x 320 tile (for all rows or the leading full rounds)
int igemm_head_rows(const IGemmParams& p);       // rows [0, r) run on the 256 x 320 tile, rows [r, M) on the 128-row tile
// shapes the persistent 256 x 320 tile (igemm_pers_tile.h) takes; the others run on the 128-row tile of igemm_tile.h
// (bit-identical results): >= 4 k steps, a time embedding only when every 256-row tile lies
// inside one sample, never a time embedding and a residual together
inline bool igemm_pers_ok(const IGemmParams& p) {
    const int nk = ((p.mode == IG_DENSE) ? 1 : 9) * (p.Cin / 64);
    if (nk < 4 || p.Cout % 320 != 0 || p.M < 2) return false;
    {   // the kernel keeps activation element offsets (row * channels + chunk) in 32 bits: larger tensors take the 128-row tile
        long long cmax = p.C1 > p.Cin - p.C1 ? p.C1 : p.Cin - p.C1;
        if (p.X3) {                          // the folded shortcut's sources (third / fourth k-loop source) use the same 32-bit offsets
            const long long c3 = p.C3 > p.Csc - p.C3 ? p.C3 : p.Csc - p.C3;
            cmax = cmax > c3 ? cmax : c3;
        }
        const long long src_rows = (p.mode == IG_DENSE) ? (long long)p.M : (long long)(p.M / (p.OH * p.OW > 0 ? p.OH * p.OW : 1)) * p.H * p.W;
        if (src_rows * cmax >= (1LL << 31)) return false;
    }
    if (p.mode != IG_DENSE && (p.OH < 1 || p.OW < 1 || p.OH > 511 || p.OW > 511 || p.M / (p.OH * p.OW) > 8191)) return false;   // packed row coordinates
    if (p.temb && (p.res || p.epi == EPI_GEGLU || p.ln_s || p.OH * p.OW < 1 || (p.OH * p.OW) % 256 != 0)) return false;
    if (p.res && (p.epi == EPI_GEGLU || p.ln_s)) return false;
    if (p.ln_s && (p.M & 1)) return false;
    if (p.ln_s && (p.mode != IG_DENSE || p.C1 != p.Cin)) return false;
    return true;
}

// Layers whose k steps run (dy, 64-channel slab, dx) so that the persistent tile can reuse one activation stage for the three
// horizontal taps (igemm_pers_tr.hip; the 128-row tile has the same k order as its KO variant, igemm_ko.hip): plain 3x3 stride-1
// convolutions on 16 / 32 / 64 / 128 pixel wide images whose samples are whole 256-row tiles.  A property of the LAYER and the sample
// geometry, never of the batch: a sample's bits do not depend on the batch it rides in.
inline bool igemm_ko_layer(const IGemmParams& p) {
    const bool same = p.mode == IG_CONV3 && p.H == p.OH && p.W == p.OW;
    const bool up2 = p.mode == IG_CONV3_UP && p.OH == 2 * p.H && p.OW == 2 * p.W && !p.temb && !p.res;     // nearest 2x, then the convolution
    return (same || up2) && p.epi == EPI_PLAIN && !p.ln_s && p.ksplit <= 1 && !p.X3 && !p.w_sample_stride &&
           (p.OW == 16 || p.OW == 32 || p.OW == 64 || p.OW == 128) && (p.OH * p.OW) % 256 == 0 &&
           p.Cout % 160 == 0 && p.Cin % 64 == 0 && p.C1 % 64 == 0;
}

// ---- K4/K5: flash attention (self and cross), head_dim 40/80/160 ------------------------------
struct AttnParams {
    const f16* Q; const f16* K; const f16* V; f16* O;
    int ldq, ldk, ldv, ldo;          // row strides in elements
    long long bsq, bsk, bsv, bso;    // batch strides in elements
    const int32_t* kv_slot;          // optional: K/V batch index per sample (prompt slot)
    int slot_div;                    // if > 0 (and kv_slot == nullptr): K/V batch index = sample / slot_div
    int n_slots = 0;                 // > 0: K/V batch indices are clamped to [0, n_slots) on the device (prompt cache rows)
    int q_mod = 0;                   // > 0: sample b reads the Q rows of sample b % q_mod (cross-attention of the shared-draw prefix: the
                                     // queries of a draw are the same under every prompt, so they are projected once per draw)
    int B, heads, Tq, Tk, D;
    float scale;
};
hipError_t launch_attention(const AttnParams& p, hipStream_t s);

// ---- K6: GroupNorm statistics + apply(+SiLU); K7: LayerNorm ----------------------------------
// x = concat(X[...,C1], X2[...,C-C1]) NHWC.  Two kernels: per-(sample, pixel chunk, group) fp64 partial sums (one read of x),
// then the apply kernel, whose prologue combines them into the per-channel affine y = x * a + b (+ SiLU) — second read, one write.
hipError_t launch_gn_stats(const f16* X, const f16* X2, int N, int HW, int C, int C1, int G,
                           double* partial /* [N][chunks][G][2] */, hipStream_t s);
int gn_stats_chunks(int HW);
hipError_t launch_gn_apply(const f16* X, const f16* X2, int N, int HW, int C, int C1, int G, float eps, const float* gamma,
                           const float* beta, const double* partial, int silu, f16* Y, hipStream_t s, int chunks = 0 /* 0: gn_stats_chunks(HW) */);
// The same statistics as per-(64-row block, channel pair) fp32 sums that a producing GEMM's epilogue can emit (r05):
//   blocks [N * HW / 64][C] fp32, entry (b, 2 k + {0, 1}) = (sum, sum of squares) of channels 2 k, 2 k + 1 over rows 64 b .. 64 b + 63:
//   rows 16 j + e of the block: acc = v_dot2_f32_f16(x_pair, (1, 1) | x_pair, acc) for j = 0..3 from 0, then the fixed tree
//   e ^ 1, e ^ 2, 7 - e, 15 - e over the 16 lanes e.  launch_gn_blocks computes rows [row0, N * HW) from the tensor in memory;
//   igemm_gn_rows(p) says how many leading rows a launch_igemm(p) with p.gn_blocks set has already written — bit-identical, so
//   which of the two produced a block never shows.  launch_gn_blocks_final: fp64 sums in a fixed order -> partial [N][1][G][2]
//   (launch_gn_apply / launch_gn_fold with chunks = 1).  HW % 64 == 0, C / G even.
template <int CTRL>
__device__ __forceinline__ float gn_dpp(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, false));
}
// the fixed tree over the 16 lanes of a DPP row (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror): every lane
// ends with the same bits (fp32 addition commutes)
__device__ __forceinline__ float gn_row16_sum(float x) {
    x += gn_dpp<0xB1>(x);
    x += gn_dpp<0x4E>(x);
    x += gn_dpp<0x141>(x);
    x += gn_dpp<0x140>(x);
    return x;
}
typedef _Float16 gn_half2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gn_pair_acc(float& s, float& q, unsigned w) {
    const gn_half2 v = __builtin_bit_cast(gn_half2, w);
    s = __builtin_amdgcn_fdot2(v, gn_half2{(_Float16)1.0f, (_Float16)1.0f}, s, false);
    q = __builtin_amdgcn_fdot2(v, v, q, false);
}
// GroupNorm over cat([x (C1 channels), skip (C2 channels)]) whose skip half was already summed for an earlier GroupNorm of the skip alone
// (r05, option gn_skip: the up path's norm1 against the down path's): px [N][chunks][G1][2] = the partial sums of x taken with G1 = C1 / cpg
// groups (cpg = (C1 + C2) / G), pskip [Ns][chunks][G][2] = the skip's own partial sums (its G groups of C2 / G channels: m = cpg / (C2 / G)
// consecutive ones make a group of the concatenation); sample n reads skip row n % Ns (the stacked copies of the shared-draw prefix).
// out [N][1][G][2]: chunks ascending, then the m skip groups ascending — a fixed order, so a sample's bits do not depend on its batch.
hipError_t launch_gn_merge_skip(const double* px, const double* pskip, int N, int Ns, int chunks, int G, int G1, int m, double* out, hipStream_t s);
hipError_t launch_gn_blocks(const f16* X, int rows, int C, int row0, float* blocks, hipStream_t s);
hipError_t launch_gn_blocks_final(const float* blocks, int N, int HW, int C, int G, double* partial, hipStream_t s);
int igemm_gn_rows(const IGemmParams& p);
bool igemm_gn_layer(const IGemmParams& p);      // false: no batch size ever gets block sums for this layer from an epilogue
// GroupNorm (no activation) folded into the following 1x1 convolution W [Cout][C], bias [Cout]: from the same partial sums,
// Wn [N][Cout][C] = fp16(W[o][c] * a[n][c]) and tn [N][Cout] = bias[o] + sum_c W[o][c] * b[n][c]  (fp32), y = x * a + b being the
// normalisation's per-(sample, channel) affine.  The convolution then runs on the RAW x with per-sample weights: the
// normalised tensor is never written or read.
hipError_t launch_gn_fold(const double* partial, int N, int HW, int C, int G, float eps, const float* gamma, const float* beta,
                          const f16* W, const f16* bias, int Cout, f16* Wn, float* tn, hipStream_t s);
hipError_t launch_layernorm(const f16* X, int rows, int C, const float* gamma, const float* beta,
                            float eps, f16* Y, hipStream_t s);
// per-row (mean, rstd) of X [rows][C] -> stats [rows][2] fp32 (the statistics half of LayerNorm)
hipError_t launch_ln_stats(const f16* X, int rows, int C, float eps, float* stats, hipStream_t s);

// ---- K8/K9/K10/K11 and glue -------------------------------------------------------------------
// temb0[b][320] = fp16(sinusoid table[t[b]])
hipError_t launch_time_gather(const f16* table, const int64_t* t, int B, int dim, f16* out, hipStream_t s);
hipError_t launch_silu(const f16* in, f16* out, long long n, hipStream_t s);
// add_noise fused with the im2col of conv_in (3x3, 4 ch):
// out [B*H*W][64] fp16, k = c*9 + ky*3 + kx for k < 36, zero after; conv_in itself is then one igemm.
// latent_f32 = 1: x / eps / the two coefficient tables are fp32 and the sum is rounded to fp16 once (the
// reference's autocast flow); 0: all fp16, table cast to fp16 first (fp16 scheduler).
// if sqrt_acp == nullptr the (fp16 or fp32) sample is used as is (plain U-Net forward / DIFT).
hipError_t launch_im2col_in(const void* x, const int32_t* x_index, const void* eps, const int64_t* t,
                            const void* sqrt_acp, const void* sqrt_1macp, int latent_f32, int B, int H, int W, f16* out,
                            hipStream_t s);
// conv_out 3x3 (C0 -> 4) on the normalised activations, fused eps-MSE (wavefront shuffle reduce).
// loss [B,4,H,W] fp32 = (float(fp16(conv)) - float(eps))^2 ; if eps == nullptr writes pred fp16 NCHW.
// sample b reads eps row (b % eps_rows) and writes output row (b / out_group) * out_stride + out_off + b % out_group
// (identity when eps_rows = out_group = B, out_stride = out_off = 0).
hipError_t launch_conv_out(const f16* Xn /* NHWC [B,H,W,C0] */, const f16* w /* [4][9*C0] k=(tap,c) */,
                           const f16* bias, const void* eps /* fp32 if eps_f32 else fp16 */, int eps_f32, int B, int H, int W, int C0,
                           float* loss, f16* pred, int eps_rows, int out_group, int out_stride, int out_off,
                           hipStream_t s);
// the same layer with the input rows staged once in LDS and walked by all nine taps on the matrix cores (conv_out.hip, r05; option
// conv_out_rows): launch_conv_out takes it where conv_out_rows_strip(H, W, C0) > 0 (C0 = 320, W <= ~160)
int conv_out_rows_strip(int H, int W, int C0);
hipError_t launch_conv_out_rows(const f16* Xn, const f16* w, const f16* bias, const void* eps, int eps_f32, int B, int H, int W, int C0,
                                float* loss, f16* pred, int eps_rows, int out_group, int out_stride, int out_off, hipStream_t s);
hipError_t launch_nhwc_to_nchw(const f16* X, int N, int HW, int C, f16* Y, hipStream_t s);
// mean over groups of `ens` consecutive samples, NHWC fp16 -> NCHW fp32
hipError_t launch_ensemble_mean(const f16* X, int groups, int ens, int HW, int C, float* Y, hipStream_t s);
// typicality reductions of the loss grids of n_images images: map [n_images][HW], scalar [n_images] (optional).
// cond_major = 0: loss [n_images][n_draws][n_cond][4][HW] (the reference's grid); 1: [n_cond][n_images][n_draws][4][HW]
// (dm_score_conds' row order)
hipError_t launch_typicality(const void* loss, int is_f16, int n_images, int n_draws, int n_cond, int HW, int cond_major,
                             float* map, float* scalar, hipStream_t s);

// image-space reduction of the latent typicality map: bilinear (align_corners=False) to (H, W), then
// AvgPool2d((kx, ky), stride 1); tmp [H][W-ky+1], out [H-kx+1][W-ky+1] fp32
hipError_t launch_typicality_image(const float* map, int h, int w, int H, int W, int kx, int ky, float* tmp,
                                   float* out, hipStream_t s);
// consumers' normalisations of an fp32 map (cluster.py:32-47, utils.py:14-20,130): mode 1 signed -> [0,1], 2 / max|.|,
// 3 positive only, 4 split (out2 = the negative part); mm = 2 floats of scratch (min, max)
hipError_t launch_map_normalize(const float* map, long long n, int mode, float* mm, float* out, float* out2, hipStream_t s);

// ---- VAE encoder (vae.hip) ---------------------------------------------------------------------
// RGB image NCHW fp16 -> im2col rows of encoder.conv_in: out [B*H*W][64], k = c*9 + ky*3 + kx (k < 27), zero after
hipError_t launch_im2col_rgb(const f16* x, int B, int H, int W, f16* out, hipStream_t s);
// single-head attention, head_dim 512: Q/K/V [B][T][ld], O [B][T][ldo]
hipError_t launch_attention512(const f16* Q, const f16* K, const f16* V, f16* O, int B, int T, int ld, int ldo,
                               float scale, hipStream_t s);
// quant_conv (1x1, 8->8) on the first 8 channels of Hm [B*HW][ldh] + `draws` posterior samples per image * scaling
// (NCHW outputs: latents [B*draws,4,HW], moments [B,8,HW])
hipError_t launch_posterior(const f16* Hm, int ldh, const f16* qw, const f16* qb, const f16* noise, int B, int draws,
                            int HW, float scaling, f16* latent16, float* latent32, float* moments, hipStream_t s);
// DIFT patch descriptors (cluster.py:291-299): out[p][c] = mean of feat[c][r0:r1, c0:c1], L2-normalised over c;
// feat [C][h][w] fp32, boxes [P][4] = (r0, r1, c0, c1) in feature cells
hipError_t launch_patch_embed(const float* feat, int C, int h, int w, const int32_t* boxes, int P, float* out, hipStream_t s);

// ---- CLIP text tower (clip.hip) ------------------------------------------------------------------
hipError_t launch_clip_embed(const int32_t* ids, const f16* tok, const f16* pos, int rows, int T, int C, int vocab, f16* out,
                             hipStream_t s);
// causal attention over qkv [n*T][3*heads*64] (q pre-scaled) -> out [n*T][heads*64]; T <= 80
hipError_t launch_clip_attention(const f16* qkv, int n, int T, int heads, f16* out, hipStream_t s);
hipError_t launch_quick_gelu(f16* x, long long n, hipStream_t s);
hipError_t launch_f16_to_f32(const f16* x, float* y, long long n, hipStream_t s);

}  // namespace dm
