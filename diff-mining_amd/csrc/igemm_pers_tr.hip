// igemm_pers_tr.hip — the persistent 256 px x 320 ch implicit-GEMM tile for 3x3 stride-1 convolutions with HORIZONTAL TAP REUSE:
// the k steps run (dy, 64-channel slab, dx) and the three dx steps of a (dy, slab) read ONE activation stage, laid out in LDS as the
// tile's image rows with a one-pixel halo left and right ((W + 2) LDS rows per image row: 264 / 272 / 288 rows for W = 64 / 32 /
// 16), the B-fragment read of tap dx being the same read shifted by dx rows.  Per three k steps the LDS-DMA moves 33-36 KiB of
// activations + 3 x 40 KiB of weights instead of 3 x 72 KiB (6.2-6.5 instead of 9 LDS-DMA instructions per wave and step):
// tools/probes/probe_feed.hip prices a k step at 1.58-1.65 us instead of 1.87-1.89 (profiles/r03_probe_feed_tap_reuse.txt); the kernel
// gets a third to a half of that: -7..-9 % per launch on the 64-pixel-wide time-embedding layers, -6 % with a residual, -4..-5 % on the
// 32-pixel-wide time-embedding layers, nothing at 16 pixels (profiles/r03_ab_tap_reuse.txt; the rule is tap_reuse_layer(), igemm.hip).
// Same arithmetic, k order and epilogue as igemm_tile.h's KO variant (igemm_ko.hip): bit-identical tiles, so which of the two
// kernels computes a row never shows in a result.  Everything else (tile hand-out, continuous k stream across tiles, epilogue
// straight from the accumulators, counted vmcnt at the tile top) is igemm_pers_tile.h's; reached from `unet(...)`,
// diffmining/typicality/compute.py:100 (ResnetBlock2D.conv1 / conv2 of the 64x64, 32x32 and 16x16 levels).
#include "igemm_pers_tile.h"

namespace dm {

namespace {

typedef volatile __attribute__((address_space(3))) int* lds_word_t;    // a volatile LDS word as ds_read / ds_write (a generic volatile pointer compiles to flat_*)

// EXTRA: PX_NONE / PX_TEMB / PX_RES as in igemm_pers_tile.h; WIMG: image width = output width (64, 32 or 16)
// UNROLL: the k loop walks (dy, slab) pairs with its three dx steps unrolled (compile-time dx: no branch and no run-time piece index in
// the MFMA stream; 3-5 % faster) — for the instantiations the register allocator handles without spilling accumulators inside the
// loop, which launch_tr_w lists and tests/test_abi.py checks against the compiled ISA; the others run the run-time-dx loop.
// UP: the convolution runs on the nearest-2x up-sampled image (Upsample2D): the stage rows are up-sampled coordinates, a row's source
// pixel is (ih >> 1, iw >> 1) of the H x W input.
template <int EXTRA, int WIMG, bool UNROLL = false, bool UP = false>
__global__ __launch_bounds__(512, 2)
void igemm_pers_tr_kernel(IGemmParams p, int ntiles, int cset) {
    constexpr int EPI = EPI_PLAIN;
    constexpr bool LN = false, WS = false;
    constexpr int WC = 2, CH = 2, NW = 8;
    constexpr int TP = 256, TC = 320;
    constexpr int WBYTES = TC * 128;
    constexpr int PITCH = WIMG + 2;                       // LDS rows per image row
    constexpr int IROWS = TP / WIMG;                      // image rows of a tile
    constexpr int ROWS = IROWS * PITCH;                   // 264 / 272 / 288
    constexpr int XP = (ROWS + 7) / 8;                    // 1 KiB pieces of an activation stage: 33 / 34 / 36
    constexpr int XBYTES = XP * 1024;
    constexpr int XI = (XP + NW - 1) / NW;                // pieces per wave (the last one only in the first XP - 8 (XI - 1) waves): 5
    constexpr int WI = TC / 8 / NW;                       // 5 weight pieces per wave and k step
    constexpr int AUX_BYTES = 2048;                       // bias + time-embedding row (shadows the 8 KiB slot of igemm_pers_tile.h)
    // GroupNorm block sums of the output from the epilogue (IGemmParams::gn_blocks): the time-embedding launches (ResnetBlock2D.conv1)
    // of 64-pixel-wide images — the 128-, 32- and 16-pixel instantiations spill with the 20 extra accumulators (tools/kernel_spills.py)
    constexpr bool GNBK = (EXTRA == PX_TEMB) && WIMG == 64;
    constexpr int NSTORE = GNBK ? 34 : 24;      // + 10 block-sum stores
    static_assert(XI == 5 && 2 * WBYTES + 2 * XBYTES + 2 * AUX_BYTES <= 160 * 1024, "LDS budget");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const xs0 = smem + 2 * WBYTES;
    char* const aux0 = xs0 + 2 * XBYTES;

    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wc = wid % WC;
    const int wp = wid / WC;

    // ---- this block's tiles: XCD x (= block % 8) owns a contiguous range; its blocks stride through it ----------
    const int tiles_c = p.Cout / TC;
    int tile, tend, tdyn;
    const int xcd = blockIdx.x & 7;
    {
        const int nblk = gridDim.x;
        const int loc = blockIdx.x >> 3;
        const int q = ntiles >> 3, r = ntiles & 7;
        const int tstart = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        tend = tstart + q + (xcd < r ? 1 : 0);
        tdyn = tstart + ((nblk - xcd + 7) >> 3);
        tile = tstart + loc;
    }
    int* const ctr = p.tile_ctr ? p.tile_ctr : &g_tile_ctr[cset][0];
    auto finish = [&]() __attribute__((always_inline)) {
        if (threadIdx.x == 0) {
            if (atomicAdd(&ctr[CTR_DONE], 1) == (int)gridDim.x - 1) {
#pragma unroll
                for (int x = 0; x < 8; ++x) ctr[x * 32] = 0;
                ctr[CTR_DONE] = 0;
                __threadfence();
            }
        }
    };
    if (tile >= tend) { finish(); return; }

    const int C1 = p.C1;
    const int C2 = p.Cin - C1;
    const int cpt = p.Cin / BK;
    const int ntrip = 3 * cpt;                                   // (dy, slab) pairs of a tile; k steps = 3 * ntrip
    const int Ktot = 9 * p.Cin;
    const int OHW = p.OH * p.OW;                                 // = H * W, a multiple of 256: a tile lies inside one sample
    constexpr bool temb_lds = (EXTRA == PX_TEMB);

    // ---- load-side state -------------------------------------------------------------------------------------------
    int lp0 = 0, lc0 = 0;             // tile whose operands are being fetched
    unsigned woff = 0;
    const unsigned wstride = (unsigned)(NW * 8) * (unsigned)Ktot;
    const f16* xbase = p.X;
    int ld_dy = 0, ld_cc = 0;         // (dy, slab) of the activation stage to fetch next

    auto set_wtile = [&](int tl) __attribute__((always_inline)) {       // weight rows of tile tl, k position 0
        const int pt = tl / tiles_c;
        lc0 = (tl - pt * tiles_c) * TC;
        const int ln = hw_lane();
        const int lrow = ln >> 3, lchunk = ((ln & 7) ^ lrow) * 8;
        woff = (unsigned)((size_t)(lc0 + wid * 8 + lrow) * Ktot + lchunk);
    };
    auto set_xtile = [&](int tl) __attribute__((always_inline)) {
        lp0 = (tl / tiles_c) * TP;
        ld_dy = 0; ld_cc = 0;
    };
    // source pixel of this lane's LDS row of activation piece i for the line offset x_dy of the tile at lp0 (-1: halo / outside the
    // image / beyond ROWS) — derived where the piece is issued (~12 VALU, 5 pieces per three k steps) instead of held: every
    // register kept across the k loop is one too many (a 5-register array of offsets filled between the stages spills)
    int x_dy = 0, x_cs = 0, x_cho = 0;
    auto src_pixel = [&](int i, int lrow) __attribute__((always_inline)) -> int {
        const int n = lp0 / OHW;
        const int oh0 = (lp0 - n * OHW) / WIMG;
        const int r = (wid + i * NW) * 8 + lrow;                    // LDS row of the stage
        const int ir = r / PITCH;
        const int iw = r - ir * PITCH - 1;
        const int ih = oh0 + ir + x_dy - 1;
        if constexpr (UP) {
            const bool oku = r < ROWS && iw >= 0 && iw < WIMG && ih >= 0 && ih < p.OH;
            return oku ? (n * p.H + (ih >> 1)) * p.W + (iw >> 1) : -1;
        }
        const bool ok = r < ROWS && iw >= 0 && iw < WIMG && ih >= 0 && ih < p.H;
        return ok ? (n * p.H + ih) * WIMG + iw : -1;
    };
    auto prepare_x = [&]() __attribute__((always_inline)) {             // the next activation stage: (ld_dy, ld_cc), then advance
        x_dy = ld_dy;
        const int ch = ld_cc * BK;
        if (ch < C1) { xbase = p.X; x_cs = C1; x_cho = ch; } else { xbase = p.X2; x_cs = C2; x_cho = ch - C1; }
        if (++ld_cc == cpt) { ld_cc = 0; ++ld_dy; }
    };
    auto load_w = [&](int buf, int idx, int adv) __attribute__((always_inline)) {       // weight piece idx of the next k step
        lptr_t dst = (lptr_t)(smem + buf * WBYTES + (wid + idx * NW) * 1024);
        __builtin_amdgcn_global_load_lds((gptr_t)(p.Wp + (size_t)(woff + (unsigned)idx * wstride)), dst, 16, 0, 0);
        if (idx == WI - 1) woff += (unsigned)adv;
    };
    // element offset of this lane's 16 bytes of activation piece i (-1: zero page), and the piece's LDS-DMA.  A k step derives the
    // offsets of its (at most two) pieces BEFORE its fragment reads, when 36 registers are still free, and holds two integers
    // through the first MFMA groups; derived at the point of issue the ~8 temporaries of the row -> pixel arithmetic spill
    auto x_offset = [&](int i, int ln) __attribute__((always_inline)) -> int {
        const int lrow = ln >> 3, lchunk = ((ln & 7) ^ lrow) * 8;
        const int pix = src_pixel(i, lrow);
        return pix >= 0 ? pix * x_cs + x_cho + lchunk : -1;
    };
    auto load_x = [&](int buf, int i, int off, int ln) __attribute__((always_inline)) {
        if (i == XI - 1 && wid >= XP - NW * (XI - 1)) return;                            // uniform per wave (i: run-time, uniform)
        lptr_t dst = (lptr_t)(xs0 + buf * XBYTES + (wid + i * NW) * 1024);
        const f16* a = (off >= 0) ? (xbase + (size_t)(unsigned)off)
                                  : reinterpret_cast<const f16*>(p.zero_page) + ((ln & 7) ^ (ln >> 3)) * 8;
        __builtin_amdgcn_global_load_lds((gptr_t)a, dst, 16, 0, 0);
    };
    auto load_aux = [&](int slot, int lc0, int lp0) __attribute__((always_inline)) {      // vectors of the tile at (lp0, lc0)
        char* ax = aux0 + slot * AUX_BYTES;
        const int lane = hw_lane();
        const char* zp = reinterpret_cast<const char*>(p.zero_page) + lane * 16;
        if (wid == 0) {
            const char* s = (p.bias && lane < TC / 8) ? reinterpret_cast<const char*>(p.bias + lc0) + lane * 16 : zp;
            __builtin_amdgcn_global_load_lds((gptr_t)s, (lptr_t)(ax + AUX_BIAS), 16, 0, 0);
        } else if (wid == 1) {
            if (temb_lds) {
                const int n = lp0 / OHW;
                const char* s = (lane < TC / 8) ? reinterpret_cast<const char*>(p.temb + (size_t)n * p.temb_ld + lc0) + lane * 16 : zp;
                __builtin_amdgcn_global_load_lds((gptr_t)s, (lptr_t)(ax + AUX_TEMB), 16, 0, 0);
            }
        }
    };

    floatx4 acc[CH][5][4];

    // one k step: weights of stage wcur, activation stage xcur read shifted by dx rows; the LDS-DMA of the next k step's weights
    // (adv = distance to the step after that) and pieces 2 dx, 2 dx + 1 (dx = 2: piece 4 only) of the
    // next activation stage interleaved.  dx is a run-time value on purpose: unrolled by three the k loop's register allocation spills accumulators.
    constexpr bool SNAKE = DM_MFMA_SNAKE && !(EXTRA == PX_TEMB && WIMG == 64 && UNROLL);
    auto step = [&](int wcur, int xcur, auto dxv, int adv) __attribute__((always_inline)) {
        const int dx = dxv;                  // an int (run-time loop) or an integral_constant (unrolled loop: everything below folds)
        const char* wt = smem + wcur * WBYTES;
        const char* xt = xs0 + xcur * XBYTES;
        const int ln = hw_lane();
        const int l15 = ln & 15, lg = ln >> 4;
        const int a_row_off = (wc * 80 * CH + l15) * 128;
        const int koff0 = ((lg ^ (l15 & 7)) << 4), koff1 = (((4 + lg) ^ (l15 & 7)) << 4);
        // B rows: pixel (wp * 64 + 16 j + l15) of the tile sits in LDS row rb + cj + l15, cj = (16 j / W) (W + 2) + 16 j % W
        const int rb = ((wp * 64) / WIMG) * PITCH + (wp * 64) % WIMG + dx + l15;
        int boff[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cj = ((16 * j) / WIMG) * PITCH + (16 * j) % WIMG;
            if (j > 0 && (cj & 7) == 0) { boff[j] = boff[0] + cj * 128; continue; }         // same key as j = 0: an immediate offset
            const int row = rb + cj;
            boff[j] = row * 128 + ((lg ^ (row & 7)) << 4);
        }
        half8 b0[4], b1[4], a[5];
#pragma unroll
        for (int j = 0; j < 4; ++j) b0[j] = *reinterpret_cast<const half8*>(xt + boff[j]);
#pragma unroll
        for (int i = 0; i < 5; ++i) a[i] = *reinterpret_cast<const half8*>(wt + a_row_off + i * 2048 + koff0);
        int grp = 0;
#pragma unroll
        for (int q = 0; q < 2 * CH; ++q) {
            const int sidx = q / CH, h = q % CH;
            if (q == CH - 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int cj = ((16 * j) / WIMG) * PITCH + (16 * j) % WIMG;
                    const int o1 = (j > 0 && (cj & 7) == 0) ? (boff[0] ^ 64) + cj * 128 : (boff[j] ^ 64);
                    b1[j] = *reinterpret_cast<const half8*>(xt + o1);
                }
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    // snake order over the pixel fragments (igemm_pers_tile.h): one operand changes per MFMA.  Starting reversed (even groups)
                    // is the parity at which no instantiation but one spills; that one — the GroupNorm-block-emitting PX_TEMB kernel of
                    // 64-pixel-wide images, 255 registers — spills 4-12 registers under every reordering and keeps the ascending order
                    const int j = (SNAKE && !(i & 1)) ? 3 - jj : jj;
                    acc[h][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], sidx ? b1[j] : b0[j], acc[h][i][j], 0, 0, 0);
                }
                if (q + 1 < 2 * CH) {
                    const int nq = q + 1, ns = nq / CH, nh = nq % CH;
                    a[i] = *reinterpret_cast<const half8*>(wt + a_row_off + nh * (80 * 128) + i * 2048 + (ns ? koff1 : koff0));
                }
                // two LDS-DMA pieces per MFMA group, weights first: all out within the first quarter of the step
                if (grp == 0) { load_w(wcur ^ 1, 0, adv); load_w(wcur ^ 1, 1, adv); }
                if (grp == 1) { load_w(wcur ^ 1, 2, adv); load_w(wcur ^ 1, 3, adv); }
                // (re-issuing an already fetched piece instead of the two uniform branches below — no branch in the MFMA stream —
                //  measured 2-4 % slower: the LDS-DMA instruction costs more than the branch)
                if (grp == 2) { load_w(wcur ^ 1, 4, adv); load_x(xcur ^ 1, 2 * dx, x_offset(2 * dx, ln), ln); }
                if (grp == 3) { if (dx < 2) load_x(xcur ^ 1, 2 * dx + 1, x_offset(2 * dx + 1, ln), ln); }
                ++grp;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    auto epilogue = [&](int p0, int c0out, int slot) __attribute__((always_inline)) {
        const char* ax = aux0 + slot * AUX_BYTES;
        constexpr int OCH = (EPI == EPI_GEGLU) ? 40 : 80;          // output channels of one (wave, h) sub-tile
        constexpr bool RES = (EXTRA == PX_RES);
        constexpr bool GNB = GNBK;
        const int c0o = (EPI == EPI_GEGLU) ? c0out / 2 : c0out;
        // the lane id is re-read from the hardware inside every epilogue: anything derived from the kernel-wide `lane`
        // is hoisted out of the tile loop by the compiler and then lives (or spills) across the k loop, which runs
        // at 256 registers; a spill reload here would also sit behind the LDS-DMA just issued (vmcnt is in-order)
        const int eln = hw_lane();
        const int e15 = eln & 15, eg = eln >> 4;
        f16* const sink = reinterpret_cast<f16*>(g_store_sink) + (wid * 64 + eln) * 8;
        // residual of unit u = (h, j) in the layout of the stores: 16 channels of block eg (two half8) + block 4's quarter
        // Residual variant, two phases: (A) every unit is converted / transposed into packed registers while ALL residual
        // loads are issued (unit u+1's right after unit u's conversion freed its 20 accumulator registers), (B) add + store.
        // Every load is older than every store (vmcnt retires in order: a load queued behind stores would wait for them),
        // and the wait for the first residual — which also sits behind the LDS-DMA of the next tile's first k step —
        // overlaps the conversion work instead of preceding it.
        constexpr int NU = 4 * CH;
        half8 rlo[RES ? NU : 1], rhi[RES ? NU : 1], plo[RES ? NU : 1], phi[RES ? NU : 1];
        half4 r4[RES ? NU : 1], p4[RES ? NU : 1];
        f16* ypu[RES ? NU : 1];
        auto load_res = [&](int u) __attribute__((always_inline)) {
            const int h = u >> 2, j = u & 3;
            int m = p0 + wp * 64 + 16 * j + e15;
            m = m < p.M ? m : p.M - 1;
            const f16* rp = p.res + (size_t)m * p.ldres + c0o + wc * (OCH * CH) + h * OCH;
            rlo[RES ? u : 0] = *reinterpret_cast<const half8*>(rp + 16 * eg);
            rhi[RES ? u : 0] = *reinterpret_cast<const half8*>(rp + 16 * eg + 8);
            r4[RES ? u : 0] = *reinterpret_cast<const half4*>(rp + 64 + 4 * eg);
        };
        if (RES) load_res(0);
#pragma unroll
        for (int h = 0; h < CH; ++h) {
            // GroupNorm block sums of the (wave, h) sub-tile's 64 rows x 80 channels (time-embedding launches = conv1, norm2's input):
            // ten channel pairs per lane, accumulated over the four row blocks j, then the fixed 16-lane tree (dm_kernels.h)
            float gs[GNB ? 10 : 1], gq[GNB ? 10 : 1];
            if constexpr (GNB) {
#pragma unroll
                for (int k = 0; k < 10; ++k) { gs[k] = 0.f; gq[k] = 0.f; }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int u = h * 4 + j;
                const int pr = wp * 64 + 16 * j + e15;
                const int m = p0 + pr;
                // folded LayerNorm: y = rstd (acc - mean s) + t, evaluated as fma(rstd, acc, fma(-rstd mean, s, t)) — two fused
                // operations per value instead of mul / sub / mul / add (r03: 4 of ~37 VALU instructions per GEGLU quad), and one
                // rounding less; the 128-row tile (igemm_tile.h) uses the same form
                float nrm = 0.f, rs = 1.f;
                if (LN && !WS) {
                    const float2 st = *reinterpret_cast<const float2*>(ax + AUX_STATS + pr * 8);
                    rs = st.y; nrm = -(st.y * st.x);
                }
                constexpr int NWD = (EPI == EPI_GEGLU) ? 1 : 2;
                unsigned R[4][NWD], R4[NWD];
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    const int ct = wc * 80 * CH + h * 80 + 16 * i + 4 * eg;          // tile-local GEMM channel
                    float v[4];
                    if (LN) {
                        const floatx4 t4 = *reinterpret_cast<const floatx4*>(ax + AUX_LNT + ct * 4);
                        const floatx4 s4 = *reinterpret_cast<const floatx4*>(ax + AUX_LNS + ct * 4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(rs, acc[h][i][j][r], __builtin_fmaf(nrm, s4[r], t4[r]));
                    } else {
                        const half4 bv = *reinterpret_cast<const half4*>(ax + AUX_BIAS + ct * 2);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = acc[h][i][j][r] + (float)bv[r];
                    }
                    unsigned w0, w1 = 0;
                    if (EPI == EPI_GEGLU) {
                        const f16 h0 = (f16)v[0], h1 = (f16)v[1], g0 = (f16)v[2], g1 = (f16)v[3];
                        const f16 q0 = (f16)gelu_erf((float)g0), q1 = (f16)gelu_erf((float)g1);
                        const half2_ o = half2_{(f16)((float)h0 * (float)q0), (f16)((float)h1 * (float)q1)};
                        w0 = __builtin_bit_cast(unsigned, o);
                    } else {
                        half4 o = half4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                        if (temb_lds) {
                            const half4 tv = *reinterpret_cast<const half4*>(ax + AUX_TEMB + ct * 2);
#pragma unroll
                            for (int r = 0; r < 4; ++r) o[r] = (f16)((float)o[r] + (float)tv[r]);
                        }
                        const uintx2 uu = __builtin_bit_cast(uintx2, o);
                        w0 = uu[0]; w1 = uu[1];
                    }
                    if (i < 4) { R[i][0] = w0; if (NWD == 2) R[i][NWD - 1] = w1; }
                    else { R4[0] = w0; if (NWD == 2) R4[NWD - 1] = w1; }
                }
                transpose4<NWD>(R);            // lane group eg now holds quarters 0..3 of channel block eg
                f16* yp = (m < p.M) ? p.Y + (size_t)m * p.ldy + c0o + wc * (OCH * CH) + h * OCH : nullptr;
                if (EPI == EPI_GEGLU) {
                    const uintx4 o8 = uintx4{R[0][0], R[1][0], R[2][0], R[3][0]};           // 8 output channels of block eg
                    *reinterpret_cast<uintx4*>(yp ? yp + 8 * eg : sink) = o8;
                    *reinterpret_cast<unsigned*>(yp ? yp + 32 + 2 * eg : sink) = R4[0];
                } else {
                    half8 lo = __builtin_bit_cast(half8, uintx4{R[0][0], R[0][NWD - 1], R[1][0], R[1][NWD - 1]});
                    half8 hi = __builtin_bit_cast(half8, uintx4{R[2][0], R[2][NWD - 1], R[3][0], R[3][NWD - 1]});
                    half4 o4 = __builtin_bit_cast(half4, uintx2{R4[0], R4[NWD - 1]});
                    if (RES) {
                        plo[RES ? u : 0] = lo; phi[RES ? u : 0] = hi; p4[RES ? u : 0] = o4; ypu[RES ? u : 0] = yp;
                        if (u + 1 < NU) load_res(u + 1);
                    } else {
                        *reinterpret_cast<half8*>(yp ? yp + 16 * eg : sink) = lo;
                        *reinterpret_cast<half8*>(yp ? yp + 16 * eg + 8 : sink) = hi;
                        *reinterpret_cast<half4*>(yp ? yp + 64 + 4 * eg : sink) = o4;
                        if constexpr (GNB) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                gn_pair_acc(gs[GNB ? 2 * i : 0], gq[GNB ? 2 * i : 0], R[i][0]);
                                gn_pair_acc(gs[GNB ? 2 * i + 1 : 0], gq[GNB ? 2 * i + 1 : 0], R[i][NWD - 1]);
                            }
                            gn_pair_acc(gs[GNB ? 8 : 0], gq[GNB ? 8 : 0], R4[0]);
                            gn_pair_acc(gs[GNB ? 9 : 0], gq[GNB ? 9 : 0], R4[NWD - 1]);
                        }
                    }
                }
            }
            if constexpr (GNB) {
#pragma unroll
                for (int k = 0; k < 10; ++k) { gs[k] = gn_row16_sum(gs[k]); gq[k] = gn_row16_sum(gq[k]); }
                const int r0 = p0 + wp * 64;
                // every lane issues the five stores (the vmcnt count at the next tile's top is per instruction): lanes other than e15 = 0,
                // blocks beyond M and launches without a block buffer write their slot of the sink page
                float* const fsink = reinterpret_cast<float*>(sink);
                float* const gb = (p.gn_blocks && e15 == 0 && r0 < p.M) ? p.gn_blocks + (size_t)(r0 >> 6) * p.Cout + c0o + wc * (OCH * CH) + h * OCH : nullptr;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<floatx4*>(gb ? gb + 16 * eg + 4 * i : fsink) = floatx4{gs[GNB ? 2 * i : 0], gq[GNB ? 2 * i : 0], gs[GNB ? 2 * i + 1 : 0], gq[GNB ? 2 * i + 1 : 0]};
                *reinterpret_cast<floatx4*>(gb ? gb + 64 + 4 * eg : fsink) = floatx4{gs[GNB ? 8 : 0], gq[GNB ? 8 : 0], gs[GNB ? 9 : 0], gq[GNB ? 9 : 0]};
            }
        }
        if (RES) {
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                half8 lo = plo[RES ? u : 0], hi = phi[RES ? u : 0];
                half4 o4 = p4[RES ? u : 0];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    lo[r] = (f16)((float)lo[r] + (float)rlo[RES ? u : 0][r]);
                    hi[r] = (f16)((float)hi[r] + (float)rhi[RES ? u : 0][r]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) o4[r] = (f16)((float)o4[r] + (float)r4[RES ? u : 0][r]);
                f16* yp = ypu[RES ? u : 0];
                *reinterpret_cast<half8*>(yp ? yp + 16 * eg : sink) = lo;
                *reinterpret_cast<half8*>(yp ? yp + 16 * eg + 8 : sink) = hi;
                *reinterpret_cast<half4*>(yp ? yp + 64 + 4 * eg : sink) = o4;
            }
        }
    };

    // ---- prologue: the first tile's k step 0 weights -> W stage 0, its first activation stage -> X stage 0, vectors -> slot 0 ----
    set_wtile(tile);
    set_xtile(tile);
    load_aux(0, lc0, lp0);
    prepare_x();
    {
        const int ln = hw_lane();
#pragma unroll
        for (int i = 0; i < WI; ++i) load_w(0, i, p.Cin);          // step 0 = (dy 0, slab 0, dx 0); the next one is dx 1: + Cin
#pragma unroll
        for (int i = 0; i < XI; ++i) load_x(0, i, x_offset(i, ln), ln);
    }
    prepare_x();                      // sources of the second activation stage

    int g = 0;                        // global k step of the stream: weight stage = g & 1
    int tg = 0;                       // global activation stage of the stream: stage = tg & 1
    int slot = 0;
    bool first = true;
    while (true) {
        const int pt = tile / tiles_c;
        const int p0 = pt * TP, c0out = (tile - pt * tiles_c) * TC;
#pragma unroll
        for (int h = 0; h < CH; ++h)
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[h][i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

        // the tile's k step 0 weights, its first activation stage and its vectors were requested BEFORE the previous epilogue's
        // stores, so "at most NSTORE outstanding" proves they have landed (igemm_pers_tile.h)
        if (first) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        else if (NSTORE == 34) asm volatile("s_waitcnt vmcnt(34)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(24)\n\ts_barrier" ::: "memory");
        first = false;
        int ticket = 0;
        // thread 0; the block-sum instantiations re-derive it (threadIdx.x kept alive across their k loop is the register that spills)
        auto thread0 = [&]() __attribute__((always_inline)) -> bool {
            if constexpr (GNBK) return wid == 0 && hw_lane() == 0;
            else return threadIdx.x == 0;
        };
        if (thread0()) ticket = atomicAdd(&ctr[xcd * 32], 1);
        int next = 0;
        bool has_next = false;
        if constexpr (UNROLL) {
            int cc = 0;                               // slab of the (dy, slab) pair
            for (int tr = 0; tr < ntrip; ++tr) {
                const bool last = tr == ntrip - 1;
                // weights: the step after the next one is the next tap (+ Cin) — or, from dx = 2, the first tap of the next slab of
                // the line (+ 64 - 2 Cin) / of the next line (+ 64)
                step(g & 1, tg & 1, std::integral_constant<int, 0>{}, p.Cin);
                ++g;
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                if (tr == 0 && thread0()) *(lds_word_t)(aux0 + slot * AUX_BYTES + 1020) = ticket;
                step(g & 1, tg & 1, std::integral_constant<int, 1>{}, (cc == cpt - 1) ? BK : BK - 2 * p.Cin);
                ++g;
                if (last) {        // the weights of the NEXT tile's first k step are requested by this tile's last step
                    next = tdyn + *(lds_word_t)(aux0 + slot * AUX_BYTES + 1020);
                    next = __builtin_amdgcn_readfirstlane(next);
                    has_next = next < tend;
                    set_wtile(has_next ? next : tile);
                }
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                step(g & 1, tg & 1, std::integral_constant<int, 2>{}, p.Cin);
                ++g;
                ++tg;
                if (tr == ntrip - 2) {               // (the ticket was published after this tile's first k step; ntrip >= 15)
                    int nx = tdyn + *(lds_word_t)(aux0 + slot * AUX_BYTES + 1020);
                    nx = __builtin_amdgcn_readfirstlane(nx);
                    const bool hn = nx < tend;
                    set_xtile(hn ? nx : tile);
                    if (hn) load_aux(slot ^ 1, (nx - (nx / tiles_c) * tiles_c) * TC, lp0);
                }
                prepare_x();
                if (++cc == cpt) cc = 0;
                if (!last) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            }
        } else {
        const int nk = 3 * ntrip;
            int dx = 0, tr = 0, cc = 0;                  // horizontal tap, (dy, slab) index and slab of the k step
            for (int kt = 0; kt < nk; ++kt) {
                // weights: the step after the next one is the next tap (+ Cin) — or, from dx = 2, the first tap of the next slab of
                // the line (+ 64 - 2 Cin) / of the next line (+ 64)
                const int adv = (dx == 1) ? ((cc == cpt - 1) ? BK : BK - 2 * p.Cin) : p.Cin;
                step(g & 1, tg & 1, dx, adv);
                ++g;
                if (dx == 1 && tr == ntrip - 1) {        // the weights of the NEXT tile's first k step are requested by this tile's last step
                    next = tdyn + *(lds_word_t)(aux0 + slot * AUX_BYTES + 1020);
                    next = __builtin_amdgcn_readfirstlane(next);
                    has_next = next < tend;
                    set_wtile(has_next ? next : tile);
                }
                if (dx == 2) {
                    ++tg;
                    // sources of the activation stage after the one just requested: from the second-to-last stage on they belong to the
                    // next tile (without one the block re-requests its own first stage: valid addresses, unused)
                    if (tr == ntrip - 2) {               // (the ticket was published after this tile's first k step; ntrip >= 15)
                        int nx = tdyn + *(lds_word_t)(aux0 + slot * AUX_BYTES + 1020);
                        nx = __builtin_amdgcn_readfirstlane(nx);
                        const bool hn = nx < tend;
                        set_xtile(hn ? nx : tile);
                        if (hn) load_aux(slot ^ 1, (nx - (nx / tiles_c) * tiles_c) * TC, lp0);
                    }
                    prepare_x();
                    dx = 0; ++tr;
                    if (++cc == cpt) cc = 0;
                } else ++dx;
                if (kt < nk - 1) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                if (kt == 0 && thread0()) *(lds_word_t)(aux0 + slot * AUX_BYTES + 1020) = ticket;
            }
        }
        epilogue(p0, c0out, slot);
        if (!has_next) break;
        tile = next;
        slot ^= 1;
    }
    finish();
}

}  // namespace

// which instantiations run the unrolled k loop: the ones hipcc 7.2 allocates without spills (tools/kernel_regs.sh; asserted by
// tests/test_abi.py::test_tap_reuse_kernels_do_not_spill) — measured per launch against the run-time-dx loop in
// profiles/r03_ab_tap_reuse.txt
#ifdef DM_TR_ALL_UNROLL
template <int EXTRA, int WIMG> struct TrUnroll { static constexpr bool value = true; };
#else
template <int EXTRA, int WIMG> struct TrUnroll { static constexpr bool value = false; };
#endif
#if !defined(DM_TR_NO_UNROLL) && !defined(DM_TR_ALL_UNROLL)
template <> struct TrUnroll<PX_NONE, 64> { static constexpr bool value = true; };
template <> struct TrUnroll<PX_TEMB, 64> { static constexpr bool value = true; };
template <> struct TrUnroll<PX_NONE, 128> { static constexpr bool value = true; };
template <> struct TrUnroll<PX_TEMB, 128> { static constexpr bool value = true; };
template <> struct TrUnroll<PX_TEMB, 32> { static constexpr bool value = true; };
template <> struct TrUnroll<PX_TEMB, 16> { static constexpr bool value = true; };
#endif

#ifndef DM_TR_UP_UNROLL
#define DM_TR_UP_UNROLL false
#endif
template <bool U>
static hipError_t launch_tr_up64(const IGemmParams& p, hipStream_t s) {      // Upsample2D.conv onto a 64-pixel-wide image: bias only
    constexpr size_t lds = 2 * (size_t)320 * 128 + 2 * (size_t)33 * 1024 + 2 * 2048;
    const int ntiles = (p.M / 256) * (p.Cout / 320);
    const int n_cu = device_cu_count();
    static std::atomic<uint64_t> attr_seen{0};
    if (first_use_on_device(attr_seen))
        (void)hipFuncSetAttribute((const void*)igemm_pers_tr_kernel<PX_NONE, 64, U, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    static std::atomic<unsigned> launch_no{0};
    const int cset = (int)(launch_no.fetch_add(1) % CSETS);
    launch_timed((igemm_pers_tr_kernel<PX_NONE, 64, U, true>), dim3(ntiles < n_cu ? ntiles : n_cu), dim3(512), lds, s, with_zero_page(p), ntiles, cset);
    return hipGetLastError();
}

template <int EXTRA, int WIMG>
static void launch_tr_k(const IGemmParams& p, int ntiles, int cset, dim3 g, size_t lds, hipStream_t s, bool set_attr) {
    constexpr bool U = TrUnroll<EXTRA, WIMG>::value;
    if (set_attr) (void)hipFuncSetAttribute((const void*)igemm_pers_tr_kernel<EXTRA, WIMG, U>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    else launch_timed((igemm_pers_tr_kernel<EXTRA, WIMG, U>), g, dim3(512), lds, s, with_zero_page(p), ntiles, cset);
}

template <int WIMG>
static hipError_t launch_tr_w(const IGemmParams& p, hipStream_t s) {
    constexpr int TP = 256, TC = 320;
    constexpr int XP = ((TP / WIMG) * (WIMG + 2) + 7) / 8;
    constexpr size_t lds = 2 * (size_t)TC * 128 + 2 * (size_t)XP * 1024 + 2 * 2048;
    const int ntiles = (p.M / TP) * (p.Cout / TC);
    const int n_cu = device_cu_count();
    const int grid = ntiles < n_cu ? ntiles : n_cu;
    const dim3 g(grid);
    static std::atomic<uint64_t> attr_seen{0};
    if (first_use_on_device(attr_seen)) {
        launch_tr_k<PX_NONE, WIMG>(p, 0, 0, g, lds, s, true);
        launch_tr_k<PX_TEMB, WIMG>(p, 0, 0, g, lds, s, true);
        launch_tr_k<PX_RES, WIMG>(p, 0, 0, g, lds, s, true);
    }
    static std::atomic<unsigned> launch_no{0};
    const int cset = (int)(launch_no.fetch_add(1) % CSETS);
    if (p.temb) launch_tr_k<PX_TEMB, WIMG>(p, ntiles, cset, g, lds, s, false);
    else if (p.res) launch_tr_k<PX_RES, WIMG>(p, ntiles, cset, g, lds, s, false);
    else launch_tr_k<PX_NONE, WIMG>(p, ntiles, cset, g, lds, s, false);
    return hipGetLastError();
}

// shapes the tap-reuse tile takes (the caller has already established igemm_ko_layer(p))
bool igemm_pers_tr_ok(const IGemmParams& p) {
    if (p.Cout % 320 != 0 || p.M % 256 != 0 || p.Cin < 320 || (p.temb && p.res)) return false;
    const long long cmax = p.C1 > p.Cin - p.C1 ? p.C1 : p.Cin - p.C1;
    return (long long)p.M * cmax < (1ll << 31) && 9ll * p.Cin * p.Cout < (1ll << 31);      // 32-bit offsets
}

hipError_t launch_igemm_pers_tr(const IGemmParams& p, hipStream_t s) {
    DM_REQUIRE_ZERO_PAGE();
    if (!igemm_ko_layer(p) || !igemm_pers_tr_ok(p)) return hipErrorInvalidValue;
    if (p.mode == IG_CONV3_UP) return p.OW == 64 ? launch_tr_up64<DM_TR_UP_UNROLL>(p, s) : hipErrorInvalidValue;
    return p.OW == 128 ? launch_tr_w<128>(p, s) : p.OW == 64 ? launch_tr_w<64>(p, s) : p.OW == 32 ? launch_tr_w<32>(p, s) : launch_tr_w<16>(p, s);
}

}  // namespace dm
