// igemm.hip — K1/K2/K3: implicit-GEMM on gfx950 MFMA for every matmul-shaped op of the SDv1.5
// U-Net the reference calls at diffmining/typicality/compute.py:100 / dift.py:191:
// 3x3 conv (stride 1, stride 2, nearest-upsampled input), 1x1 conv and Linear, with fused
// bias / time-embedding / residual / GEGLU epilogues.
//
// Formulation: Y[m][co] = sum_k X~[m][k] * Wp[co][k], m = output pixel (NHWC row), k = (tap, cin).
// The WEIGHT tile is the MFMA "A" operand (rows = output channels) and the ACTIVATION tile the "B"
// operand (cols = pixels), so a lane holds 4 consecutive output channels of one pixel.
//
// Tile: 128 pixels x 320 channels (8 waves) or 128 x 160 (4 waves; when Cout % 320 != 0), k step 64
// (one tap, 64 input channels); each wave owns 64 px x 80 ch = 5x4 fragments of
// v_mfma_f32_16x16x32_f16 (80 accumulator VGPRs).  160 divides every channel count of the network.
//   * operand tiles go L2/HBM -> LDS with global_load_lds_dwordx4 (no VGPR staging, no ds_write);
//     the LDS image of one instruction is lane-linear (8 rows x 128 B), so the XOR swizzle that makes
//     the ds_read_b128 conflict-free is applied on the per-lane SOURCE address and again on the read;
//   * 3x3 halo zero padding / rows beyond M = lanes pointed at a 128-byte zero page;
//   * two LDS stages, one barrier per k step; the LDS-DMA pieces of the next k tile are issued one at
//     a time between groups of four MFMAs (an LDS-DMA costs ~60-180 issue cycles - in a burst after
//     the barrier the matrix pipe idles behind it);
//   * epilogue staged through LDS and written as whole rows with 16-byte stores.
// Measured alternatives (r01, same shapes): register staging 0.87x, burst DMA
// 0.93x, 256x160 tiles (2 or 3 LDS stages, counted vmcnt, register-double-buffered fragments,
// horizontal tap reuse for 3x3) 0.80-0.95x, direct fragment stores 0.85x on the wide short-K linears.
#include "igemm_tile.h"

namespace dm {

hipError_t launch_igemm64(const IGemmParams& p, hipStream_t s);      // igemm64.hip (64-channel waves)
hipError_t launch_igemm_splitk(const IGemmParams& p, hipStream_t s);  // igemm_splitk.hip
hipError_t launch_igemm_tile_ln(const IGemmParams& p, hipStream_t s);  // igemm_ln.hip
hipError_t launch_igemm_tile_ln_half(const IGemmParams& p, hipStream_t s);
hipError_t launch_igemm_pers(const IGemmParams& p, hipStream_t s);     // igemm_pers.hip (256 x 320 tile, persistent)
hipError_t launch_igemm_pers_ln(const IGemmParams& p, hipStream_t s);  // igemm_pers_ln.hip
hipError_t launch_igemm_pers_partial(const IGemmParams& p, hipStream_t s);   // igemm_pers_part.hip (split-K units, fp32 partials)
hipError_t launch_splitk_reduce(const IGemmParams& p, hipStream_t s);         // igemm_splitk.hip
hipError_t launch_igemm_pers_ws(const IGemmParams& p, hipStream_t s);         // igemm_pers_ws.hip: per-sample weights (GroupNorm fold)
hipError_t launch_igemm_tile_ws(const IGemmParams& p, hipStream_t s);         // igemm_ws.hip
hipError_t launch_igemm_pers_sc(const IGemmParams& p, hipStream_t s);         // igemm_pers_sc.hip: conv_shortcut folded into conv2
hipError_t launch_igemm_tile_sc(const IGemmParams& p, hipStream_t s);         // igemm_sc.hip
hipError_t launch_igemm_pers_tr(const IGemmParams& p, hipStream_t s);         // igemm_pers_tr.hip: 3x3 convolutions with horizontal tap reuse
bool igemm_pers_tr_ok(const IGemmParams& p);
hipError_t launch_igemm_tile_ko(const IGemmParams& p, hipStream_t s);         // igemm_ko.hip: the same k order on the 128-row tile

// Shape -> tile choice, measured per shape on one box with both arms interleaved (tools/ab_igemm.py, r02): the
// persistent 256 x 320 tile (igemm_pers_tile.h: 13.8 instead of 21.9 LDS-DMA bytes per kMAC, no per-tile prologue, stores
// draining under the next tile) wins or ties on every U-Net shape that gives it >= 2 tiles per CU — -12 % on the
// 320-channel q/k/v projections, -5..8 % on the 320-channel 3x3 convolutions, -7 % on ff.net.2 at 16x16, -3 % on the
// 1280-channel projections — as long as its tiles fill whole rounds over the CUs; head_rows() below deals with the rest.
// The choice depends on the batch size through the tile count; the two kernels are bit-identical
// (tests/test_gpu_ops.py::test_persistent_tile_is_bit_identical), so a sample's result does not.
// full rounds of 256 x 320 tiles a launch must have before the persistent kernel takes them: two (the stream across
// tiles is what hides the per-tile prologue), one where a tile has >= 40 k steps (3x3 convolutions from 320 channels up,
// ff.net.2 at 16x16: -6..-13 % at 256 tiles, batch 64; the short-k projections lose up to 20 % there)
static int min_rounds(const IGemmParams& p) {
    const int nk = ((p.mode == IG_DENSE) ? 1 : 9) * (p.Cin / BK);
    return nk >= 40 ? 1 : 2;
}
static bool use_big(const IGemmParams& p) {
    const int force = option(OPT_IGEMM_BIG);                 // -1: per shape; 0 / 1: A/B switch for every eligible shape
    if (p.Cout % 320 != 0 || !igemm_pers_ok(p)) return false;
    if (force >= 0) return force != 0;
    const long long tiles = (long long)((p.M + 255) / 256) * (p.Cout / 320);
    return tiles >= (long long)min_rounds(p) * device_cu_count();
}

// Layers at <= 8x8 spatial positions per sample (M = 10 240 rows at the bench batch: 320 tiles for 256 CUs):
// cut k into up to four parts so the launch has ~5 blocks per CU (DM_IGEMM_SPLITK=0 disables).  The decision
// depends on the layer (spatial size, k extent, Cout) and never on the batch size, so a sample's result
// does not depend on how many samples share the call.  Returns 1 when the shape runs unsplit.
int igemm_splitk_parts(const IGemmParams& p, int spatial) {
    const int on = option(OPT_IGEMM_SPLITK);
    if (!on || spatial > 64 || p.epi != EPI_PLAIN || p.Cout % 320 != 0 || p.Cin % BK != 0 || p.C1 % BK != 0) return 1;
    const int nk = ((p.mode == IG_DENSE) ? 1 : 9) * (p.Cin / BK);
    if (nk < 40) return 1;
    // 3x3 convolutions: three parts of three taps (the persistent split-K kernel walks whole taps); igemm_splitk = 2 keeps
    // the r01 rule (four parts on the 128-row tile) for A/B
    if (on != 2 && p.mode != IG_DENSE && (nk / 3) >= 10) return 3;
    for (int k = 4; k >= 2; --k)
        if (nk % k == 0 && nk / k >= 10) return k;
    return 1;
}

// the split-K launches the persistent 256 x 320 tile takes: 3x3 convolution, three tap-aligned parts, plain epilogue, and
// fewer (weighted) rounds than the 128-row tile
static bool splitk_on_pers(const IGemmParams& p) {
    if (option(OPT_IGEMM_SPLITK) == 2 || p.ksplit != 3 || p.mode == IG_DENSE || p.epi != EPI_PLAIN || p.ln_s) return false;
    if (p.Cout % 320 != 0 || p.Cin % BK != 0 || p.C1 % BK != 0) return false;
    IGemmParams q = p; q.temb = nullptr; q.res = nullptr;           // those are applied by the reduction kernel
    if (!igemm_pers_ok(q)) return false;
    // rounds over the CUs x relative k-step cost (a 128-row step is measured at 0.73 of a 256-row one): the 128-row
    // kernel keeps the launches whose rows do not fill its own first round (a 256-row tile would be half empty)
    const int n_cu = device_cu_count();
    const long long up = (long long)((p.M + 255) / 256) * (p.Cout / 320) * p.ksplit;
    const long long us = (long long)((p.M + 127) / 128) * (p.Cout / 320) * p.ksplit;
    return (double)((up + n_cu - 1) / n_cu) < 0.73 * (double)((us + n_cu - 1) / n_cu);
}

// Rows [r0, r1) of a launch as a launch of its own.  Dense rows are independent; for the convolution modes both cuts
// must lie on sample boundaries (multiples of OH*OW), so every pointer moves by whole samples.
static IGemmParams row_range(const IGemmParams& p, int r0, int r1) {
    IGemmParams q = p;
    const int C2 = p.Cin - p.C1;
    q.M = r1 - r0;
    q.Y = p.Y + (size_t)r0 * p.ldy;
    if (p.res) q.res = p.res + (size_t)r0 * p.ldres;
    if (p.ln_stats) q.ln_stats = p.ln_stats + (size_t)r0 * 2;
    if (p.mode == IG_DENSE) {
        q.X = p.X + (size_t)r0 * p.C1;
        if (p.X2) q.X2 = p.X2 + (size_t)r0 * C2;
        if (p.X3) q.X3 = p.X3 + (size_t)r0 * p.C3;
        if (p.X4) q.X4 = p.X4 + (size_t)r0 * (p.Csc - p.C3);
        q.W = q.OW = q.M;
    } else {
        const size_t n0 = (size_t)r0 / ((size_t)p.OH * p.OW), src = (size_t)p.H * p.W;
        q.X = p.X + n0 * src * p.C1;
        if (p.X2) q.X2 = p.X2 + n0 * src * C2;
        if (p.X3) q.X3 = p.X3 + n0 * src * p.C3;
        if (p.X4) q.X4 = p.X4 + n0 * src * (p.Csc - p.C3);
        if (p.temb) q.temb = p.temb + n0 * p.temb_ld;
    }
    return q;
}

// Tile quantisation: a launch whose 256 x 320 tiles do not fill a whole number of rounds over the CUs (640 tiles on
// 256 CUs = 2.5 rounds: every 16x16-level layer at the bench batch) keeps the full rounds on the persistent kernel and
// hands the remaining rows to the 128-row tile (half the rows per tile = twice the blocks for the last, partial
// round).  Both kernels give the same bits, so the cut only moves time.  Returns the first row of the tail (a multiple of
// 256 and, for the convolutions, of OH*OW), 0 = everything on the 128-row tile, M = everything on the persistent tile.
// Cost model in units of one round of 256 x 320 tiles; a round of 128-row tiles is measured at 0.59-0.62 of that on
// the long-k shapes and ~0.68 on the short ones (per-tile prologue and LDS-staged epilogue).  Measured at the bench
// batch (tools/ab_igemm.py igemm_tail 0 1 @16, profiles/r02_ab_head_tail.txt): -6..-9 % on every 1280-channel
// convolution / projection / shortcut at 16x16, +-0 where the rounds are whole.
static int head_rows(const IGemmParams& p) {
    const int force = option(OPT_IGEMM_BIG);
    if (p.Cout % 320 != 0 || !igemm_pers_ok(p)) return 0;
    if (force >= 0) return force ? p.M : 0;
    if (!use_big(p) && !option(OPT_IGEMM_TAIL)) return 0;
    if (!option(OPT_IGEMM_TAIL)) return p.M;
    const int n_cu = device_cu_count(), tc = p.Cout / 320;
    const long long rt = (p.M + 255) / 256, tiles = rt * tc;
    const long long rounds_full = tiles / n_cu;
    const int nk = ((p.mode == IG_DENSE) ? 1 : 9) * (p.Cin / BK);
    const double rho = nk >= 32 ? 0.62 : 0.68;
    const long long small_tiles = (long long)((p.M + 127) / 128) * tc;
    const double cost_big = (double)((tiles + n_cu - 1) / n_cu);
    const double cost_small = rho * (double)((small_tiles + n_cu - 1) / n_cu);
    double best = use_big(p) ? cost_big : cost_small;
    int best_rows = use_big(p) ? p.M : 0;
    if (rounds_full >= min_rounds(p) && tiles % n_cu != 0) {
        long long unit = 256;                                   // rows per cut step: whole tiles and whole samples
        if (p.mode != IG_DENSE) {
            const long long ohw = (long long)p.OH * p.OW;
            if (256 % ohw == 0) unit = 256; else if (ohw % 256 == 0) unit = ohw; else unit = 0;
        } else if (p.temb) unit = 0;
        if (unit) {
            long long head = (rounds_full * n_cu / tc) * 256 / unit * unit;        // rows
            if (head > 0 && head < p.M) {
                const IGemmParams h = row_range(p, 0, (int)head);
                const long long head_tiles = head / 256 * tc, tail_tiles = (long long)((p.M - head + 127) / 128) * tc;
                const double cost = (double)((head_tiles + n_cu - 1) / n_cu) + rho * (double)((tail_tiles + n_cu - 1) / n_cu);
                if (igemm_pers_ok(h) && cost < best - 0.05) { best = cost; best_rows = (int)head; }
            }
        }
    }
    return best_rows;
}

// which tile geometry launch_igemm picks for a plain (no split-K) shape: 0 = 128-row, 1 = 256 x 320 (for all or for the
// leading full rounds of the rows)
int igemm_tile_choice(const IGemmParams& p) {
    if (p.Cout % 160 != 0 || p.mode == IG_CONV3_S2P0) return 0;
    return head_rows(p) > 0 ? 1 : 0;
}

int igemm_head_rows(const IGemmParams& p) {
    if (p.Cout % 160 != 0 || p.mode == IG_CONV3_S2P0) return 0;
    return head_rows(p);
}

// The (dy, slab, dx) k order + horizontal tap reuse (igemm_pers_tr.hip / igemm_ko.hip).  Option tap_reuse: 1 = where it pays at
// the bench batch (tools/ab_igemm.py tap_reuse 0 2, profiles/r03_ab_tap_reuse.txt): every eligible layer of 64-pixel-wide images
// (-6..-9 % per launch with a time embedding — the unrolled k loop —, -5 % with a residual) and the time-embedding layers
// (ResnetBlock2D.conv1) of 32-pixel-wide ones (-4..-5 %); 2 = every eligible layer (32 wide with a residual: +-0; 16 wide: +-0 /
// +3 %: the out-of-order walk of their larger weights costs what the activation reuse saves); 0 = off.  A property of the
// layer (geometry, epilogue), never of the batch.
static bool tap_reuse_layer(const IGemmParams& p) {
    const int on = option(OPT_TAP_REUSE);
    if (on == 0 || !igemm_ko_layer(p)) return false;
    if (p.mode == IG_CONV3_UP) return on == 2 && p.OW == 64;           // Upsample2D.conv onto 64 pixels: built, +3 % (1.33 PFLOP/s without it: its taps share source pixels), A/B only
    return on == 2 || p.OW >= 64 || (p.OW == 32 && (p.temb != nullptr || p.has_temb));       // (128 wide: the first level of a 1024-pixel image, the X-ray configuration)
}

static hipError_t launch_small(const IGemmParams& p, hipStream_t s) {
    if (p.X3) return launch_igemm_tile_sc(p, s);
    if (tap_reuse_layer(p)) return launch_igemm_tile_ko(p, s);
    if (p.ln_s) return launch_igemm_tile_ln(p, s);
    return (p.Cout % 320 == 0) ? launch_t<4, 5>(p, s) : launch_t<2, 5>(p, s);
}

// Leading rows of launch_igemm(p) whose GroupNorm block sums (p.gn_blocks) the producing kernel's epilogue writes: the rows the
// persistent kernels take of a time-embedding launch (ResnetBlock2D.conv1).  The caller computes the remaining blocks from the
// output (launch_gn_blocks) — the same bits either way, so the cut (a function of the batch size) never shows in a result.
// Can ANY batch size get block sums for this layer from an epilogue?  A property of the layer (geometry, epilogue, options), never of
// M: the caller keeps the r04 statistics pass for the layers that never can (the 128- / 32- / 16-pixel tap-reuse kernels, split-K
// layers, channel counts the persistent tile does not take) — one launch instead of two there.
bool igemm_gn_layer(const IGemmParams& p) {
    if (!(p.temb || p.has_temb) || p.epi != EPI_PLAIN || p.w_sample_stride || p.ln_s || p.X3 || (p.OH * p.OW) % 64 != 0) return false;
    if (p.Cout % 320 != 0 || p.mode == IG_CONV3_S2P0 || p.Cin % BK != 0 || p.C1 % BK != 0) return false;
    if (igemm_splitk_parts(p, p.OH * p.OW) > 1) return false;
    if (tap_reuse_layer(p) && p.OW != 64) return false;
    return true;
}

int igemm_gn_rows(const IGemmParams& p) {
    if (!igemm_gn_layer(p)) return 0;
    if (!p.gn_blocks || !p.temb || p.epi != EPI_PLAIN || p.M % 64 != 0 || (p.ksplit > 1 && p.partial) || p.w_sample_stride || p.ln_s || p.X3) return 0;
    if (p.Cout % 160 != 0 || p.mode == IG_CONV3_S2P0 || p.Cin % BK != 0 || p.C1 % BK != 0 || p.M <= 0) return 0;
    if (option(OPT_IGEMM_EXP) == 1 && p.epi == EPI_GEGLU) return 0;
    const int head = head_rows(p);
    if (head <= 0) return 0;
    const IGemmParams h = head < p.M ? row_range(p, 0, head) : p;
    if (tap_reuse_layer(p) && (!igemm_pers_tr_ok(h) || p.OW != 64)) return 0;      // the tap-reuse kernel emits them for 64-pixel-wide images only (igemm_pers_tr.hip)
    return head < p.M ? head : p.M;
}

hipError_t launch_igemm(const IGemmParams& p_in, hipStream_t s) {
    IGemmParams p = p_in;
    if (p.gn_blocks && igemm_gn_rows(p) == 0) p.gn_blocks = nullptr;
    if (p.ksplit > 1 && p.partial) {
        if (!splitk_on_pers(p)) return launch_igemm_splitk(p, s);
        const hipError_t rc = launch_igemm_pers_partial(p, s);
        return rc != hipSuccess ? rc : launch_splitk_reduce(p, s);
    }
    if (p.w_sample_stride) {
        // GroupNorm folded into a 1x1 convolution: per-sample weights.  The persistent tile when a sample is whole 256-row tiles
        // and the launch has >= 2 rounds of them (or igemm_big forces it), else the 128-row tile; no head / tail cut (both give
        // the same bits, so which one runs is a matter of time only)
        if (p.Cout % 160 != 0 || p.mode != IG_DENSE || p.X3) return hipErrorInvalidValue;      // shapes the WS kernels do not cover
        const bool pers = p.Cout % 320 == 0 && p.rows_per_sample % 256 == 0 && igemm_pers_ok(p) &&      // (32-bit activation offsets)
                          (option(OPT_IGEMM_BIG) >= 0 ? option(OPT_IGEMM_BIG) != 0
                                                      : (long long)(p.M / 256) * (p.Cout / 320) >= 2LL * device_cu_count());
        return pers ? launch_igemm_pers_ws(p, s) : launch_igemm_tile_ws(p, s);
    }
    if (p.ln_s) {
        if (!p.ln_t || p.Cout % 160 != 0 || p.mode != IG_DENSE || p.C1 != p.Cin) return hipErrorInvalidValue;
    } else {
        if (p.Cout % 160 != 0 || p.mode == IG_CONV3_S2P0) {
            if (p.X3) return hipErrorInvalidValue;       // the folded shortcut exists on the 160-channel-wave tiles only: never drop it silently
            return launch_igemm64(p, s);                 // VAE channel counts
        }
        if (p.Cin % BK != 0 || p.C1 % BK != 0 || p.M <= 0) return hipErrorInvalidValue;
    }
    if (option(OPT_IGEMM_EXP) == 1 && p.epi == EPI_GEGLU) return p.ln_s ? launch_igemm_tile_ln_half(p, s) : launch_t<2, 5>(p, s);
    const int head = head_rows(p);
    if (head <= 0) return launch_small(p, s);
    const IGemmParams h = head < p.M ? row_range(p, 0, head) : p;
    if (tap_reuse_layer(p) && !igemm_pers_tr_ok(h)) return launch_small(p, s);     // (32-bit offsets: the whole launch on the 128-row tile)
    const hipError_t rc = p.X3 ? launch_igemm_pers_sc(h, s) : p.ln_s ? launch_igemm_pers_ln(h, s) : tap_reuse_layer(h) ? launch_igemm_pers_tr(h, s) : launch_igemm_pers(h, s);
    if (rc != hipSuccess || head >= p.M) return rc;
    return launch_small(row_range(p, head, p.M), s);
}

}  // namespace dm
