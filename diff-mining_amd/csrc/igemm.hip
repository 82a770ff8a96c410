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

hipError_t launch_igemm_big(const IGemmParams& p, hipStream_t s);     // igemm_big.hip (256 x 320 tile)
hipError_t launch_igemm64(const IGemmParams& p, hipStream_t s);      // igemm64.hip (64-channel waves)
hipError_t launch_igemm_splitk(const IGemmParams& p, hipStream_t s);  // igemm_splitk.hip
hipError_t launch_igemm_tile_ln(const IGemmParams& p, hipStream_t s);  // igemm_ln.hip
hipError_t launch_igemm_big_ln(const IGemmParams& p, hipStream_t s);   // igemm_big_ln.hip
hipError_t launch_igemm_pers(const IGemmParams& p, hipStream_t s);     // igemm_pers.hip (256 x 320 tile, persistent)
hipError_t launch_igemm_pers_ln(const IGemmParams& p, hipStream_t s);  // igemm_pers_ln.hip

// Shape -> tile choice (measured on MI355X at the bench batch, tools/bench_ops.py): the 256x320 tile
// pays on the k >= 640 linears, on the >= 640-channel / concat 3x3 convs and on the wide GEGLU
// projections (+8..25 %) when the launch still has >= 2 tiles per CU; 128x320 wins elsewhere.
static bool use_big(const IGemmParams& p) {
    const int force = option(OPT_IGEMM_BIG);                 // -1: per shape
    if (p.Cout % 320 != 0) return false;
    if (force >= 0) return force != 0;                       // DM_IGEMM_BIG=0/1: A/B switch for every eligible shape
    const long long tiles = (long long)((p.M + 255) / 256) * (p.Cout / 320);
    if (tiles < 1024) return false;
    if (p.mode == IG_DENSE) return p.Cin >= 640 || p.Cout >= 2560;
    return p.Cin >= 640;
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
    for (int k = 4; k >= 2; --k)
        if (nk % k == 0 && nk / k >= 10) return k;
    return 1;
}

// Persistent form of the 256 x 320 tile (bit-identical results).  Measured per shape on one box (tools/ab_igemm.py,
// r02): -12 / -7 / -1 % on the GEGLU projections (K = 320 / 640 / 1280), -8 % on the LayerNorm-folded q/k/v projections,
// -4..5 % on the concat convolutions, -1..0 % on the time-embedding convolutions; +1 % on the residual convolutions,
// +4..5 % on the residual linears (ff.net.2, to_out), +3 % on the nearest-upsample convolutions — those stay on the
// one-tile-per-block kernel.  igemm_persist = 0: never, = 2: wherever the kernel can run (A/B).
static bool use_pers(const IGemmParams& p) {
    const int on = option(OPT_IGEMM_PERSIST);
    if (on == 0 || !igemm_pers_ok(p)) return false;
    if (on == 2) return true;
    return !p.res && p.mode != IG_CONV3_UP;
}

// which tile geometry launch_igemm picks for a plain (no LN fold, no split-K) shape: 0 = 128-row, 1 = 256 x 320
int igemm_tile_choice(const IGemmParams& p) {
    if (p.Cout % 160 != 0 || p.mode == IG_CONV3_S2P0) return 0;
    return use_big(p) ? 1 : 0;
}

hipError_t launch_igemm(const IGemmParams& p, hipStream_t s) {
    if (p.ksplit > 1 && p.partial) return launch_igemm_splitk(p, s);
    if (p.ln_stats) {
        if (!p.ln_s || !p.ln_t || p.Cout % 160 != 0) return hipErrorInvalidValue;
        if (use_big(p)) return use_pers(p) ? launch_igemm_pers_ln(p, s) : launch_igemm_big_ln(p, s);
        return launch_igemm_tile_ln(p, s);
    }
    if (p.Cout % 160 != 0 || p.mode == IG_CONV3_S2P0) return launch_igemm64(p, s);        // VAE channel counts
    if (p.Cout % 160 != 0 || p.Cin % BK != 0 || p.C1 % BK != 0 || p.M <= 0) return hipErrorInvalidValue;
    if (use_big(p)) return use_pers(p) ? launch_igemm_pers(p, s) : launch_igemm_big(p, s);
    return (p.Cout % 320 == 0) ? launch_t<4, 5>(p, s) : launch_t<2, 5>(p, s);
}

}  // namespace dm
