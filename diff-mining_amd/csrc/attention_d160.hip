// attention_d160.hip — head_dim-160 self-attention (the 1280-channel level: 256 tokens per sample at 16x16, 64 at 8x8), r06.
// Reached from `unet(...)`, diffmining/typicality/compute.py:100 (BasicTransformerBlock.attn1 of down_blocks[2], mid_block, up_blocks[1]).
//
// The generic kernel (attention.hip) runs these layers at 297 TFLOP/s: at head_dim 160 it needs 369 registers and 82 KB of LDS, i.e.
// ONE wave per SIMD, and four waves per block re-fetch K / V for every 128 queries.  Here a block is EIGHT waves = all 256 queries of a
// (sample, head) at 16x16 — K and V of the pair cross the L2 -> LDS path once —, a wave owns 32 queries and stays under 256 registers
// (two waves per SIMD) by streaming what the generic kernel holds: K fragments per 32-wide k step (16 registers), V^T fragments per
// 32-key half (40), one score tile.  head_dim 160 is five k steps of the 16x16x32 MFMA exactly and ten 16-row blocks of O^T: no padding
// anywhere, 80 MFMAs per 64-key tile and wave against ~110 VALU — matrix-bound, unlike head_dim 40.
//   * S^T = K Q^T so P is the PV B operand as it lies (attention.hip); V^T by ds_read_b64_tr_b16;
//   * rows are 320 bytes = 80 banks: the 16 rows of a ds_read_b128 would sit four deep, the 8 rows of a transpose read two deep.  The
//     LDS image is permuted through the LDS-DMA source addresses (the destination is lane-linear): K's four 16-byte chunks of a k step are
//     XORed with g(row) = (4 - (row >> 2)) & 3, V's 32-byte blocks with (row >> 2) & 1; every fragment read is conflict-free;
//   * arithmetic of the generic kernel's head_dim-160 path, order included: raw q.k in fp32, P = exp2(fma(s, scale log2 e, -m)), the
//     denominator as fp32 adds of the unrounded P, lazy rescale at 2^8 — so the two kernels agree to the last bit of the softmax and
//     differ only by the order of the PV k slots (none: the same permuted order) => bit-identical (asserted by the test).
#include "dm_kernels.h"

namespace dm {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int D = 160;
constexpr int KT = 64;                // keys per tile
constexpr int NW = 8;                 // waves per block
constexpr int QW = 32;                // queries per wave (two 16-query fragments)
constexpr int QF = 2;
constexpr int KS = 5, EF = 10;        // k steps of QK^T, 16-row blocks of O^T
constexpr float RESCALE_THR = 8.0f;   // log2 units
constexpr int RS = 2 * D;             // 320-byte LDS rows
constexpr int TILE = KT * RS;         // 20480
constexpr int KOFF = 0, VOFF = TILE;
constexpr int STAGE = 2 * TILE;       // 40960
constexpr int NPIECE = 2 * TILE / 1024 / NW;      // 5 LDS-DMA pieces per wave and tile

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int OFF>
__device__ __forceinline__ void tr_read(u32x2& out, unsigned base) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(out) : "v"(base), "n"(OFF) : "memory");
}

// the twenty transpose reads of a 32-key half SS: blocks e = 2 E2 (base `ve`) and 2 E2 + 1 (base `vo`), 4-key groups 2 SS, 2 SS + 1
template <int SS, int E2>
__device__ __forceinline__ void read_pair(u32x2 (&vraw)[EF][2], unsigned ve, unsigned vo) {
    tr_read<(2 * E2) * 32 + (2 * SS) * 16 * RS>(vraw[2 * E2][0], ve);
    tr_read<(2 * E2) * 32 + (2 * SS + 1) * 16 * RS>(vraw[2 * E2][1], ve);
    tr_read<(2 * E2 + 1) * 32 + (2 * SS) * 16 * RS>(vraw[2 * E2 + 1][0], vo);
    tr_read<(2 * E2 + 1) * 32 + (2 * SS + 1) * 16 * RS>(vraw[2 * E2 + 1][1], vo);
}
template <int SS>
__device__ __forceinline__ void read_half(u32x2 (&vraw)[EF][2], unsigned ve, unsigned vo) {
    read_pair<SS, 0>(vraw, ve, vo); read_pair<SS, 1>(vraw, ve, vo); read_pair<SS, 2>(vraw, ve, vo);
    read_pair<SS, 3>(vraw, ve, vo); read_pair<SS, 4>(vraw, ve, vo);
}

// Persistent: a block walks units (sample, head, 256-query block) u = blockIdx.x, + gridDim.x, ...  The scan over batch sizes
// (tools/attn_d160_scan.py) showed the one-unit-per-block form latency-bound — 29 us per unit for 5 us of MFMAs: Q in, four K / V tiles, O out,
// each behind the other, one block per CU — so the stream is kept continuous across units: the first K / V tile of the NEXT unit is
// requested at the top of the current unit's last tile (its stage is free), the next unit's Q rows right after the last tile's score
// MFMAs (the Q registers are dead from there on), and the O stores of a unit drain under the next unit's first tile.
__global__ __launch_bounds__(64 * NW, 2)
void attn_d160_kernel(AttnParams p, int nunits) {
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15;
    const int lg = lane >> 4;
    const int nqb = (p.Tq + NW * QW - 1) / (NW * QW);

    // ---- LDS-DMA: 20 K + 20 V pieces of 1 KiB per tile, five per wave (j = wid + 8 i; j < 20 is a K piece); piece jj covers the
    //      16-byte chunks idx = jj * 64 + lane -> (LDS row = idx / 20, LDS chunk = idx % 20); the source chunk undoes the permutation ----
    const f16* gsrc[NPIECE];
    auto set_kv = [&](int unit) __attribute__((always_inline)) {
        const int bh = unit / nqb;
        const int h = bh % p.heads, b = bh / p.heads;
        int kvb = p.kv_slot ? p.kv_slot[b] : (p.slot_div > 0 ? b / p.slot_div : b);
        if (p.n_slots > 0) kvb = kvb < 0 ? 0 : (kvb < p.n_slots ? kvb : p.n_slots - 1);
        const f16* Kb = p.K + (size_t)kvb * p.bsk + h * D;
        const f16* Vb = p.V + (size_t)kvb * p.bsv + h * D;
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) {
            const int j = wid + NW * i;
            const bool isv = j >= 20;
            const int idx = (isv ? j - 20 : j) * 64 + lane;
            const int row = idx / 20, pc = idx - row * 20;
            const int r2 = (row >> 2) & 3;
            const int lc = isv ? ((((pc >> 1) ^ (r2 & 1)) << 1) | (pc & 1)) : ((pc & ~3) | ((pc & 3) ^ ((4 - r2) & 3)));
            gsrc[i] = (isv ? Vb + (size_t)row * p.ldv : Kb + (size_t)row * p.ldk) + lc * 8;
        }
    };
    auto issue = [&](int stage) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) {
            const int j = wid + NW * i;
            char* dst = smem + stage * STAGE + ((j >= 20) ? VOFF + (j - 20) * 1024 : KOFF + j * 1024);
            __builtin_amdgcn_global_load_lds((gptr_t)gsrc[i], (lptr_t)dst, 16, 0, 0);
            gsrc[i] += (size_t)KT * ((j >= 20) ? p.ldv : p.ldk);
        }
    };
    // ---- Q fragments (B operand: query l15 + 16 jq, head_dim 32 s + 8 lg + 0..7), unscaled like the generic kernel's ----
    half8 qf[QF][KS];
    auto load_q = [&](int unit) __attribute__((always_inline)) {
        const int qblk = unit % nqb, bh = unit / nqb;
        const int h = bh % p.heads, b = bh / p.heads;
        const f16* Qb = p.Q + (size_t)b * p.bsq + h * D;
#pragma unroll
        for (int jq = 0; jq < QF; ++jq) {
            int q = qblk * (NW * QW) + wid * QW + 16 * jq + l15;
            q = q < p.Tq ? q : p.Tq - 1;
#pragma unroll
            for (int s = 0; s < KS; ++s) qf[jq][s] = *reinterpret_cast<const half8*>(Qb + (size_t)q * p.ldq + 32 * s + 8 * lg);
        }
    };
    // Unit order: the eight heads of a sample run on ONE XCD at the same time.  A 7680-byte row of the fused q/k/v matrix holds the 320-byte
    // slices of all heads, so neighbouring heads share 128-byte lines; with the heads scattered over the XCDs (block id mod 8) every such
    // line is fetched into two L2s.  XCD x owns the samples b = x, x + 8, ...; its blocks (slot = block id / 8) stride through the XCD's
    // list of (sample, head, query block) entries.
    // (Only where the samples divide evenly over the eight XCDs and fill the launch: otherwise — a single image's 20 samples — the plain
    // order u = block, block + grid, ... keeps the blocks balanced, which matters more than the shared lines.)
    const int upg = p.heads * nqb;                                  // units per sample
    const bool by_xcd = (p.B & 7) == 0 && nunits >= (int)gridDim.x && ((int)gridDim.x & 7) == 0;
    const int xcd = by_xcd ? (int)(blockIdx.x & 7) : 0, slot = by_xcd ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int nslot = by_xcd ? (int)gridDim.x >> 3 : (int)gridDim.x;           // blocks that share this block's list
    const int nent = by_xcd ? (p.B >> 3) * upg : nunits;                        // entries of the list
    auto entry_unit = [&](int e) __attribute__((always_inline)) {
        if (!by_xcd) return e;
        const int gi = e / upg;
        return (xcd + 8 * gi) * upg + (e - gi * upg);
    };
    int ent = slot;
    if (ent >= nent) return;
    int unit = entry_unit(ent);
    set_kv(unit);
    issue(0);
    load_q(unit);
    const float sc = p.scale * 1.44269504088896340736f;   // scores live in the log2 domain

    // K fragment (A operand): key 16 f + l15, head_dim 32 s + 8 lg + 0..7 = logical chunk 4 s + lg -> LDS chunk 4 s + (lg ^ g(row))
    const char* kbase = smem + l15 * RS + ((lg ^ ((4 - (l15 >> 2)) & 3)) << 4);
    // V^T fragment through transpose reads: lane group lg supplies rows 16 (2 ss + hh) + 4 lg + (l15 >> 2), 8 bytes (l15 & 3) of the
    // 32-byte block e ^ (lg & 1): two bases so that the block index stays an immediate (even e: + 32 (lg & 1), odd e: - 32 (lg & 1))
    const unsigned vb0 = (unsigned)(size_t)(smem + (4 * lg + (l15 >> 2)) * RS + 8 * (l15 & 3));
    const unsigned vbase_even = vb0 + 32u * (unsigned)(lg & 1), vbase_odd = vb0 - 32u * (unsigned)(lg & 1);

    const int ntiles = p.Tk / KT;
    int gt = 0;                        // tiles of the stream so far: stage = gt & 1
    while (true) {
        const bool has_next = ent + nslot < nent;
        const int next = has_next ? entry_unit(ent + nslot) : unit;
        floatx4 oacc[EF][QF];
#pragma unroll
        for (int e = 0; e < EF; ++e)
#pragma unroll
            for (int jq = 0; jq < QF; ++jq) oacc[e][jq] = floatx4{0, 0, 0, 0};
        float m_run[QF] = {-INFINITY, -INFINITY}, l_run[QF] = {0.f, 0.f};

        for (int t = 0; t < ntiles; ++t, ++gt) {
            const int cur = gt & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                 // the stream's tile gt landed for every wave; the other stage is free
            if (t + 1 < ntiles) issue(cur ^ 1);
            else if (has_next) { set_kv(next); issue(cur ^ 1); }          // the next unit's first tile
            const char* kt = kbase + cur * STAGE + KOFF;
            // ---------------- S^T = K Q^T: 5 k steps x 4 key blocks x 2 query blocks ----------------
            floatx4 sacc[4][QF];
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int jq = 0; jq < QF; ++jq) sacc[f][jq] = floatx4{0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const half8 kf = *reinterpret_cast<const half8*>(kt + f * 16 * RS + s * 64);
#pragma unroll
                    for (int jq = 0; jq < QF; ++jq) sacc[f][jq] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[jq][s], sacc[f][jq], 0, 0, 0);
                }
            if (t + 1 == ntiles && has_next) load_q(next);     // the Q registers are dead: the next unit's rows fly under the softmax, PV and stores
            // ---------------- online softmax (the generic kernel's non-folded path, same order) ----------------
            half8 pb[QF][2];
#pragma unroll
            for (int jq = 0; jq < QF; ++jq) {
                float mx = sacc[0][jq][0];
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = __builtin_fmaxf(mx, sacc[f][jq][r]);
                // lane-partial maximum (a query's 64 keys are spread over the four lane groups): some row of the wave grew past the
                // threshold iff some lane's partial did, so the ballot decides exactly as the generic kernel's row maximum does, and the two
                // cross-lane exchanges (a dependent ~200-cycle chain per tile) are paid only when a rescale happens — same bits
                float mxs = mx * sc;
                if (__builtin_amdgcn_ballot_w64(mxs > m_run[jq] + RESCALE_THR) != 0ull) {
                    mx = __builtin_fmaxf(mx, __shfl_xor(mx, 16));
                    mx = __builtin_fmaxf(mx, __shfl_xor(mx, 32));
                    mxs = mx * sc;
                    const float m_new = __builtin_fmaxf(m_run[jq], mxs);
                    const float alpha = __builtin_amdgcn_exp2f(m_run[jq] - m_new);
                    m_run[jq] = m_new;
                    l_run[jq] *= alpha;
#pragma unroll
                    for (int e = 0; e < EF; ++e) oacc[e][jq] *= alpha;
                }
                const float m_use = m_run[jq];
                float ps = 0.f;
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[f][jq][r], sc, -m_use));
                        ps += pv;
                        pb[jq][f >> 1][(f & 1) * 4 + r] = (f16)pv;
                    }
                l_run[jq] += ps;
            }
            // ---------------- O^T += V^T P^T: per 32-key half, all twenty transpose reads, then the twenty MFMAs ----------------
            const unsigned ve = vbase_even + (unsigned)(cur * STAGE + VOFF), vo = vbase_odd + (unsigned)(cur * STAGE + VOFF);
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                u32x2 vraw[EF][2];
                if (ss == 0) read_half<0>(vraw, ve, vo); else read_half<1>(vraw, ve, vo);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < EF; ++e) {
                    half8 va;
                    __builtin_memcpy(&va, &vraw[e][0], 8);
                    __builtin_memcpy(reinterpret_cast<char*>(&va) + 8, &vraw[e][1], 8);
#pragma unroll
                    for (int jq = 0; jq < QF; ++jq) oacc[e][jq] = __builtin_amdgcn_mfma_f32_16x16x32_f16(va, pb[jq][ss], oacc[e][jq], 0, 0, 0);
                }
            }
        }

        {
            const int qblk = unit % nqb, bh = unit / nqb;
            const int h = bh % p.heads, b = bh / p.heads;
            f16* Ob = p.O + (size_t)b * p.bso + h * D;
#pragma unroll
            for (int jq = 0; jq < QF; ++jq) {
                float l = l_run[jq];
                l += __shfl_xor(l, 16);
                l += __shfl_xor(l, 32);
                const float inv = 1.0f / l;
                const int q = qblk * (NW * QW) + wid * QW + 16 * jq + l15;
                if (q >= p.Tq) continue;
#pragma unroll
                for (int e = 0; e < EF; ++e) {
                    const int d = 16 * e + 4 * lg;
                    const half4 o = half4{(f16)(oacc[e][jq][0] * inv), (f16)(oacc[e][jq][1] * inv),
                                          (f16)(oacc[e][jq][2] * inv), (f16)(oacc[e][jq][3] * inv)};
                    *reinterpret_cast<half4*>(Ob + (size_t)q * p.ldo + d) = o;
                }
            }
        }
        if (!has_next) break;
        unit = next;
        ent += nslot;
    }
}

// ---- the 77-key cross-attention at head_dim 160 (attn2 of the 1280-channel level) -----------------------------------------------------
// One pass like attention_cross.hip (head_dim 40 / 80): all keys of the prompt fit five 16-key score blocks, so the softmax is exact in
// one step — scores, row max, exp2(sc (s - max)), P V — with keys >= Tk masked before the max and V rows >= Tk zero.  A block = eight
// waves = 256 queries of a (sample, head); K (80 rows) and V (96 rows: three 32-key PV steps) reach LDS once per block by LDS-DMA in the
// permuted image of the self-attention kernel above (rows beyond Tk come from a zero page).  The generic kernel ran this layer as two
// online-softmax tiles on one wave per SIMD: 149 TFLOP/s.  Denominator by fp32 adds of the unrounded P (no ones row: 160 = 10 x 16).
constexpr int XK_ROWS = 80, XV_ROWS = 96, XNKB = 5;
constexpr int XK_BYTES = XK_ROWS * RS, XV_BYTES = XV_ROWS * RS;                 // 25600 + 30720
constexpr int XPIECES = (XK_BYTES + XV_BYTES) / 1024;                          // 55 pieces of 1 KiB, seven slots per wave
__device__ __attribute__((aligned(256))) unsigned char g_attn160_zero[256];

template <int SS>
__device__ __forceinline__ void read_half_x(u32x2 (&vraw)[EF][2], unsigned ve, unsigned vo) {      // 32-key step SS of the 96 V rows
    read_pair<SS, 0>(vraw, ve, vo); read_pair<SS, 1>(vraw, ve, vo); read_pair<SS, 2>(vraw, ve, vo);
    read_pair<SS, 3>(vraw, ve, vo); read_pair<SS, 4>(vraw, ve, vo);
}

__global__ __launch_bounds__(64 * NW, 2)
void attn_d160_cross_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15;
    const int lg = lane >> 4;
    const int nqb = (p.Tq + NW * QW - 1) / (NW * QW);
    const int qblk = blockIdx.x % nqb;
    const int bh = blockIdx.x / nqb;
    const int h = bh % p.heads;
    const int b = bh / p.heads;
    const int q0 = qblk * (NW * QW) + wid * QW;
    int kvb = p.kv_slot ? p.kv_slot[b] : (p.slot_div > 0 ? b / p.slot_div : b);
    if (p.n_slots > 0) kvb = kvb < 0 ? 0 : (kvb < p.n_slots ? kvb : p.n_slots - 1);

    const f16* Qb = p.Q + (size_t)(p.q_mod > 0 ? b % p.q_mod : b) * p.bsq + h * D;
    const f16* Kb = p.K + (size_t)kvb * p.bsk + h * D;
    const f16* Vb = p.V + (size_t)kvb * p.bsv + h * D;
    f16* Ob = p.O + (size_t)b * p.bso + h * D;

    // K / V of the (prompt, head): piece j covers chunks idx = jj * 64 + lane of K (j < 25) or V -> (row, LDS chunk); rows >= Tk: zeros
#pragma unroll
    for (int i = 0; i < (XPIECES + NW - 1) / NW; ++i) {
        const int j = wid + NW * i;
        if (j < XPIECES) {
            const bool isv = j >= XK_BYTES / 1024;
            const int idx = (isv ? j - XK_BYTES / 1024 : j) * 64 + lane;
            const int row = idx / 20, pc = idx - row * 20;
            const int r2 = (row >> 2) & 3;
            const int lc = isv ? ((((pc >> 1) ^ (r2 & 1)) << 1) | (pc & 1)) : ((pc & ~3) | ((pc & 3) ^ ((4 - r2) & 3)));
            const f16* src = (row < p.Tk) ? (isv ? Vb + (size_t)row * p.ldv : Kb + (size_t)row * p.ldk) + lc * 8
                                          : reinterpret_cast<const f16*>(g_attn160_zero) + (lane & 15) * 8;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + j * 1024), 16, 0, 0);
        }
    }
    half8 qf[QF][KS];
#pragma unroll
    for (int jq = 0; jq < QF; ++jq) {
        int q = q0 + 16 * jq + l15;
        q = q < p.Tq ? q : p.Tq - 1;
#pragma unroll
        for (int s = 0; s < KS; ++s) qf[jq][s] = *reinterpret_cast<const half8*>(Qb + (size_t)q * p.ldq + 32 * s + 8 * lg);
    }
    const float sc = p.scale * 1.44269504088896340736f;
    const char* kt = smem + l15 * RS + ((lg ^ ((4 - (l15 >> 2)) & 3)) << 4);
    const unsigned vb0 = (unsigned)(size_t)(smem + XK_BYTES + (4 * lg + (l15 >> 2)) * RS + 8 * (l15 & 3));
    const unsigned ve = vb0 + 32u * (unsigned)(lg & 1), vo = vb0 - 32u * (unsigned)(lg & 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    floatx4 sacc[XNKB][QF];
#pragma unroll
    for (int f = 0; f < XNKB; ++f)
#pragma unroll
        for (int jq = 0; jq < QF; ++jq) sacc[f][jq] = floatx4{0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int f = 0; f < XNKB; ++f) {
            const half8 kf = *reinterpret_cast<const half8*>(kt + f * 16 * RS + s * 64);
#pragma unroll
            for (int jq = 0; jq < QF; ++jq) sacc[f][jq] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[jq][s], sacc[f][jq], 0, 0, 0);
        }
    half8 pb[QF][3];
    float l_sum[QF];
#pragma unroll
    for (int jq = 0; jq < QF; ++jq) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (16 * (XNKB - 1) + 4 * lg + r >= p.Tk) sacc[XNKB - 1][jq][r] = -1e30f;
        float mx = sacc[0][jq][0];
#pragma unroll
        for (int f = 0; f < XNKB; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = __builtin_fmaxf(mx, sacc[f][jq][r]);
        mx = __builtin_fmaxf(mx, __shfl_xor(mx, 16));
        mx = __builtin_fmaxf(mx, __shfl_xor(mx, 32));
        const float nm = -mx * sc;
        float ps = 0.f;
#pragma unroll
        for (int f = 0; f < XNKB; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[f][jq][r], sc, nm));
                ps += pv;
                pb[jq][f >> 1][(f & 1) * 4 + r] = (f16)pv;
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) pb[jq][2][4 + r] = (f16)0.0f;             // keys 80..95 do not exist
        l_sum[jq] = ps;
    }
    floatx4 oacc[EF][QF];
#pragma unroll
    for (int e = 0; e < EF; ++e)
#pragma unroll
        for (int jq = 0; jq < QF; ++jq) oacc[e][jq] = floatx4{0, 0, 0, 0};
#pragma unroll
    for (int ss = 0; ss < 3; ++ss) {
        u32x2 vraw[EF][2];
        if (ss == 0) read_half_x<0>(vraw, ve, vo); else if (ss == 1) read_half_x<1>(vraw, ve, vo); else read_half_x<2>(vraw, ve, vo);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < EF; ++e) {
            half8 va;
            __builtin_memcpy(&va, &vraw[e][0], 8);
            __builtin_memcpy(reinterpret_cast<char*>(&va) + 8, &vraw[e][1], 8);
#pragma unroll
            for (int jq = 0; jq < QF; ++jq) oacc[e][jq] = __builtin_amdgcn_mfma_f32_16x16x32_f16(va, pb[jq][ss], oacc[e][jq], 0, 0, 0);
        }
    }
#pragma unroll
    for (int jq = 0; jq < QF; ++jq) {
        float l = l_sum[jq];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const float inv = 1.0f / l;
        const int q = q0 + 16 * jq + l15;
        if (q >= p.Tq) continue;
#pragma unroll
        for (int e = 0; e < EF; ++e) {
            const int d = 16 * e + 4 * lg;
            const half4 o = half4{(f16)(oacc[e][jq][0] * inv), (f16)(oacc[e][jq][1] * inv),
                                  (f16)(oacc[e][jq][2] * inv), (f16)(oacc[e][jq][3] * inv)};
            *reinterpret_cast<half4*>(Ob + (size_t)q * p.ldo + d) = o;
        }
    }
}

}  // namespace

bool attention_d160_cross_supports(const AttnParams& p) {
    return p.D == 160 && p.Tk > 64 && p.Tk <= XK_ROWS && p.Tq >= 1;
}

hipError_t launch_attention_d160_cross(const AttnParams& p, hipStream_t s) {
    if (!attention_d160_cross_supports(p)) return hipErrorInvalidValue;
    constexpr int QBLK = NW * QW;
    dim3 grid(((p.Tq + QBLK - 1) / QBLK) * p.heads * p.B), block(64 * NW);
    const size_t lds = (size_t)XK_BYTES + XV_BYTES;
    launch_timed(attn_d160_cross_kernel, grid, block, lds, s, p);
    return hipGetLastError();
}

bool attention_d160_supports(const AttnParams& p) {
    return p.D == 160 && p.Tk >= 64 && (p.Tk % 64) == 0 && p.q_mod == 0 && p.Tq >= 1;
}

hipError_t launch_attention_d160(const AttnParams& p, hipStream_t s) {
    if (!attention_d160_supports(p)) return hipErrorInvalidValue;
    constexpr int QBLK = NW * QW;
    const int nunits = ((p.Tq + QBLK - 1) / QBLK) * p.heads * p.B;
    const int n_cu = device_cu_count();                       // one block of eight waves per CU (210 registers, 80 KB of LDS)
    dim3 grid(nunits < n_cu ? nunits : n_cu), block(64 * NW);
    const size_t lds = 2 * (size_t)STAGE;
    static std::atomic<uint64_t> attr_seen{0};      // hipFuncSetAttribute is per DEVICE, not per process
    if (first_use_on_device(attr_seen))
        (void)hipFuncSetAttribute((const void*)attn_d160_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    launch_timed(attn_d160_kernel, grid, block, lds, s, p, nunits);
    return hipGetLastError();
}

}  // namespace dm
