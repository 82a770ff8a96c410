// attention_qk32.hip — head_dim-40 self-attention with the scores on v_mfma_f32_32x32x16_f16 (r06).
// Reached from `unet(...)`, diffmining/typicality/compute.py:100 (BasicTransformerBlock.attn1 at the 64x64 level: diffusers'
// Attention with AttnProcessor2_0, softmax(q k^T / sqrt(40)) v per head).
//
// attention_pipe.hip computes S^T = K Q'^T of a 64-key x 32-query wave tile with sixteen 16x16x32 MFMAs whose k = 64 pads
// head_dim 40 (+ the two running-max columns) by 1.6x.  On the 32x32x16 form the same tile is SIX instructions at k = 48
// (192 matrix cycles instead of 256, ten MFMA issues fewer).  r04 priced that and did not build it because the 32x32 C layout
// hands a lane 16 keys of ONE query while the PV MFMA (16x16x32, O^T = V^T P^T) wants 8 keys of a query per lane with the four
// 16-lane groups holding the same 16 queries.  What that analysis missed is v_permlane16_swap:
//
//   S^T block mb (32 keys) in lane l:  query l & 31,  keys 32 mb + (r & 3) + 8 (r >> 2) + 4 h,  r = 0..15,  h = l >> 5
//   packed fp16 pairs pk[mb][p], p = r / 2:   p = 0..3 -> keys {0,1} {2,3} {8,9} {10,11} (+ 4 h),   p = 4..7 -> the same + 16
//   swap16(pk[mb][i], pk[mb][4 + i]), i = 0..3  (rows 1 / 3 of the first <-> rows 0 / 2 of the second):
//       pk[mb][0..3] = PV B operand of queries  0..15, pk[mb][4..7] = PV B operand of queries 16..31, k step mb, with k slot
//       (lane group g, j) = key 32 mb + {0, 16, 4, 20}[g] + 8 (j >> 2) + (j & 3)    — for BOTH query blocks
// i.e. eight VALU instructions per tile put P where PV wants it, and P still never touches LDS.  The V^T operand follows the
// same k-slot map through its transpose reads; V rows sit in LDS with key bits 2 and 4 exchanged (the LDS-DMA source address is
// per lane, so the permutation is free) so that the eight 4-key row blocks of one ds_read_b64_tr_b16 stay on distinct banks, and
// K rows carry their two 16-byte chunks of a k step exchanged when key bit 3 is set (the 32 rows a ds_read_b128 touches would
// otherwise sit two deep on the banks: 96-byte rows repeat mod 8).
//
// Everything else is attention_pipe.hip's: 96-byte LDS rows with a constant chunk ({1,1,0..} for K: the running max is
// subtracted inside the MFMA through two padded k columns; {1,0..} for V: the denominator is row 40 of O^T), the three-stage
// K/V ring with counted vmcnt, one barrier per tile, the lazy rescale (threshold 2^8) with one ballot per tile, scores
// ping-ponged between two register tiles, exp2 / pack of tile t under the score MFMAs of tile t+1.  NOT bit-identical to
// attention_pipe.hip: a score is now one 48-long fp32 chain instead of a 64-long one in another order.
#include "dm_kernels.h"

namespace dm {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int D = 40;
constexpr int KT = 64;                // keys per tile
constexpr int NT = 256;               // threads per block
constexpr int QW = 32;                // queries per wave
constexpr float RESCALE_THR = 8.0f;   // log2 units
constexpr int RS = 96;                // LDS row stride: 5 real chunks + 1 constant chunk
constexpr int TILE = KT * RS;         // 6144
constexpr int KOFF = 0, VOFF = TILE + 32;
constexpr int STAGE = 2 * (TILE + 32);           // 12352 (attention_pipe.hip's stage: the launcher shares its LDS size)
constexpr int NSTG = 3;               // K/V ring depth: K is fetched three, V two tiles ahead of their use
constexpr int EF = 3;                 // 16-row blocks of O^T (40 rows + the ones row)

__device__ __attribute__((aligned(16))) const unsigned short g_kconst32[8] = {0x3C00, 0x3C00, 0, 0, 0, 0, 0, 0};
__device__ __attribute__((aligned(16))) const unsigned short g_vconst32[8] = {0x3C00, 0, 0, 0, 0, 0, 0, 0};
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#define PIN(x) asm volatile("" : "+v"(x))

__device__ __forceinline__ float vmax2(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ void swap16(unsigned& a, unsigned& b) {      // a.row1 <-> b.row0, a.row3 <-> b.row2 (rows of 16 lanes)
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0]; b = r[1];
}

template <int OFF>
__device__ __forceinline__ void tr_read(u32x2& out, unsigned base) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(out) : "v"(base), "n"(OFF) : "memory");
}

__global__ __launch_bounds__(NT, 3)
void attn_qk32_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15;
    const int lg = lane >> 4;
    const int l31 = lane & 31;
    const int hh5 = lane >> 5;            // which 8 of the 16 k values of a 32x32x16 operand / which 4-key half of a C row group
    // XCD-aware block order: one XCD walks consecutive (sample, head) pairs, so all query blocks of a pair share that XCD's L2
    const int nqb = (p.Tq + 4 * QW - 1) / (4 * QW);
    int v;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, loc = bid >> 3;
        v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int qblk = v % nqb;
    const int bh = v / nqb;
    const int h = bh % p.heads;
    const int b = bh / p.heads;
    const int q0 = qblk * (4 * QW) + wid * QW;
    int kvb = p.kv_slot ? p.kv_slot[b] : (p.slot_div > 0 ? b / p.slot_div : b);
    if (p.n_slots > 0) kvb = kvb < 0 ? 0 : (kvb < p.n_slots ? kvb : p.n_slots - 1);

    const f16* Qb = p.Q + (size_t)b * p.bsq + h * D;
    const f16* Kb = p.K + (size_t)kvb * p.bsk + h * D;
    const f16* Vb = p.V + (size_t)kvb * p.bsv + h * D;
    f16* Ob = p.O + (size_t)b * p.bso + h * D;

    // ---- Q' = fp16(sc * q) as the B operand of the 32x32x16 form: lane = query l31, k step s holds head_dim 16 s + 8 hh5 + 0..7;
    //      head_dim 40 / 41 (k step 2, hh5 = 1, elements 0 / 1) carry -m_hi / -m_lo against the ones of K's constant chunk ----
    const float sc = p.scale * 1.44269504088896340736f;
    half8 qf[3];
    {
        int q = q0 + l31;
        q = q < p.Tq ? q : p.Tq - 1;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int d = 16 * s + 8 * hh5;
            if (d < D) qf[s] = *reinterpret_cast<const half8*>(Qb + (size_t)q * p.ldq + d);
            else qf[s] = half8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 8; ++k) qf[s][k] = (f16)((float)qf[s][k] * sc);
        }
    }

    // ---- LDS-DMA: 6 K + 6 V pieces of 1 KiB per tile, 3 per wave (j = wid + 4 i; j < 6 is a K piece); piece jj covers the
    //      16-byte chunks idx = jj*64 + lane -> (LDS row = idx / 6, LDS chunk = idx % 6).  K: row = key, and the chunks of a pair
    //      (2 s, 2 s + 1) are exchanged where key bit 3 is set; V: row = key with bits 2 and 4 exchanged, chunks in place.
    //      The logical chunk 5 is the constant one, fetched from a global constant --------------------------------------------
    const f16* gsrc[3];
    int ginc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int j = wid + 4 * i;
        const bool isv = j >= 6;
        const int jj = isv ? j - 6 : j;
        const int idx = jj * 64 + lane;
        const int row = idx / 6, pch = idx - row * 6;
        const int key = isv ? ((row & ~20) | ((row & 4) << 2) | ((row & 16) >> 2)) : row;
        const int ch = isv ? pch : (pch ^ ((row >> 3) & 1));
        const int ld = isv ? p.ldv : p.ldk;
        if (ch < 5) { gsrc[i] = (isv ? Vb : Kb) + (size_t)key * ld + ch * 8; ginc[i] = KT * ld; }
        else { gsrc[i] = reinterpret_cast<const f16*>(isv ? g_vconst32 : g_kconst32); ginc[i] = 0; }
    }
    auto piece_is_v = [&](int i) __attribute__((always_inline)) { return wid + 4 * i >= 6; };
    auto piece = [&](int i, int kst, int vst) __attribute__((always_inline)) {
        const int j = wid + 4 * i;
        char* dst = smem + ((j >= 6) ? vst * STAGE + VOFF + (j - 6) * 1024 : kst * STAGE + KOFF + j * 1024);
        __builtin_amdgcn_global_load_lds((gptr_t)gsrc[i], (lptr_t)dst, 16, 0, 0);
        gsrc[i] += ginc[i];
    };

    // K fragment (A operand, 32x32x16): lane = key l31 (+ 32 mb), k = 16 s + 8 hh5 + 0..7 -> LDS chunk 2 s + (hh5 ^ key bit 3)
    const char* kbase = smem + l31 * RS + 16 * (hh5 ^ ((lane >> 3) & 1));
    // V^T fragment (A operand, 16x16x32) through transpose reads: lane group g supplies the four LDS rows {0, 4, 16, 20}[g] + 0..3
    // (= keys {0, 16, 4, 20}[g] + 0..3 under the row permutation) of a (k step ss, half hh): + 32 ss + 8 hh rows
    const unsigned vbase = (unsigned)(size_t)(smem + (4 * (lg & 1) + 16 * (lg >> 1) + (l15 >> 2)) * RS + 8 * (l15 & 3));

    floatx4 oacc[EF][2];
#pragma unroll
    for (int e = 0; e < EF; ++e)
#pragma unroll
        for (int jq = 0; jq < 2; ++jq) oacc[e][jq] = floatx4{0, 0, 0, 0};
    float m_run = 0.f;                 // running max (log2 units) of query l31

    floatx16 SA[2], SB[2];             // raw score tiles sc*(q.k) - m_run of key blocks mb = 0 / 1, ping-ponged
    unsigned pk[2][8];                 // P as packed fp16 pairs; after the swaps pk[mb][4 jq .. 4 jq + 3] = PV B operand (jq, k step mb)

    // advance the running max (rare): rescale O, refresh the -m columns of Q', and fix the already computed score tile X up in place
    auto rescale = [&](floatx16 (&X)[2], const float mxl, bool first) __attribute__((always_inline)) {
        float mown = mxl;
        PIN(mown);
        const float mx = __builtin_fmaxf(mown, __shfl_xor(mown, 32));      // the other half wave holds the other keys of this query
        const float delta = first ? mx : __builtin_fmaxf(mx, 0.f);
        const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(-delta);
        m_run += delta;
        // O^T columns are queries l15 + 16 jq: this lane's own query is block (lg & 1), the other block's factor sits 16 lanes away
        const float alpha_x = __shfl_xor(alpha, 16);
        const float a0 = (lg & 1) ? alpha_x : alpha, a1 = (lg & 1) ? alpha : alpha_x;
#pragma unroll
        for (int e = 0; e < EF; ++e) { oacc[e][0] *= a0; oacc[e][1] *= a1; }
        if (hh5 == 1) {
            const f16 mh = (f16)m_run;
            const f16 ml = (f16)(m_run - (float)mh);
            qf[2][0] = -mh; qf[2][1] = -ml;
        }
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) X[mb][r] -= delta;
    };
    // exp slice i (0..15): scores 2 p, 2 p + 1 of key block mb = i / 8 (p = i % 8) -> one packed fp16 pair
    auto exp_slice = [&](const floatx16 (&X)[2], int i) __attribute__((always_inline)) {
        const int mb = i >> 3, pp = i & 7;
        const half2v hv = half2v{(f16)__builtin_amdgcn_exp2f(X[mb][2 * pp]), (f16)__builtin_amdgcn_exp2f(X[mb][2 * pp + 1])};
        unsigned u;
        __builtin_memcpy(&u, &hv, 4);
        PIN(u);
        pk[mb][pp] = u;
    };

    const int ntiles = p.Tk / KT;       // even, >= 4 (dispatch condition)

    // ---- prologue: K(0) -> S(0), first running max -------------------------------------------------
#pragma unroll
    for (int i = 0; i < 3; ++i) if (!piece_is_v(i)) piece(i, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 3; ++i) piece(i, 1, 0);              // K(1) -> stage 1, V(0) -> stage 0
#pragma unroll
    for (int i = 0; i < 3; ++i) piece(i, 2, 1);              // K(2) -> stage 2, V(1) -> stage 1
    {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) SA[mb][r] = 0.f;
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const half8 kf = *reinterpret_cast<const half8*>(kbase + KOFF + 32 * s + mb * 32 * RS);
                SA[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], SA[mb], 0, 0, 0);
            }
        float m = SA[0][0];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) m = __builtin_fmaxf(m, SA[mb][r]);
        rescale(SA, m, true);
    }

    // One iteration t: scores of tile t in X, tile t+1 into Y.  Ring of NSTG = 3 stages, s0 = t % 3:
    //   K(t+1) sits in stage (t+1)%3, V(t) in stage s0; DMA: K(t+3) -> stage s0, V(t+2) -> stage (t+2)%3.
    int s0 = 0;
    auto iteration = [&](floatx16 (&X)[2], floatx16 (&Y)[2], const bool next, const bool dma_k, const bool dma_v,
                         const bool wait3) __attribute__((always_inline)) {
        constexpr int KB = KOFF, VB = VOFF;
        const int s1 = (s0 == NSTG - 1) ? 0 : s0 + 1;
        const int s2 = (s1 == NSTG - 1) ? 0 : s1 + 1;
        const char* kcur = kbase + s1 * STAGE;                // K(t+1)
        const unsigned vcur = vbase + (unsigned)(s0 * STAGE); // V(t)
        if (wait3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");               // raw: __syncthreads() would drain the counted wait (vmcnt(0) fence)
        // ---------------- phase A: S(t+1) MFMAs || exp of S(t) || DMA issue || V^T reads || P swaps of key block 0 ----------------
        half8 kf[3][2];
        if (next) {
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) kf[s][mb] = *reinterpret_cast<const half8*>(kcur + KB + 32 * s + mb * 32 * RS);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) Y[mb][r] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) exp_slice(X, i);         // cover the latency of the K fragment reads
        __builtin_amdgcn_sched_barrier(0);
        u32x2 vraw[2][EF][2];
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            const int s = m >> 1, mb = m & 1;
            if (next) {
                Y[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[s][mb], qf[s], Y[mb], 0, 0, 0);
                PIN(Y[mb]);
            }
            exp_slice(X, 4 + 2 * m);
            exp_slice(X, 5 + 2 * m);
            // the wave's three LDS-DMA pieces, spread out
            if (m == 0 || m == 2 || m == 4) {
                const int i = m >> 1;
                if (piece_is_v(i) ? dma_v : dma_k) piece(i, s0, s2);
            }
            // key block 0 is exponentiated after slice 7 (m = 1): its four swaps ride behind the next two MFMAs
            if (m == 2) { swap16(pk[0][0], pk[0][4]); swap16(pk[0][1], pk[0][5]); }
            if (m == 3) { swap16(pk[0][2], pk[0][6]); swap16(pk[0][3], pk[0][7]); }
            // V(t)^T fragments: 12 transpose reads behind the last four MFMAs; offset = 32 e + (32 ss + 8 hh) RS
            if (m == 2) { tr_read<VB + 0 + 0 * 8 * RS>(vraw[0][0][0], vcur); tr_read<VB + 0 + 1 * 8 * RS>(vraw[0][0][1], vcur); tr_read<VB + 32 + 0 * 8 * RS>(vraw[0][1][0], vcur); }
            if (m == 3) { tr_read<VB + 32 + 1 * 8 * RS>(vraw[0][1][1], vcur); tr_read<VB + 64 + 0 * 8 * RS>(vraw[0][2][0], vcur); tr_read<VB + 64 + 1 * 8 * RS>(vraw[0][2][1], vcur); }
            if (m == 4) { tr_read<VB + 0 + 4 * 8 * RS>(vraw[1][0][0], vcur); tr_read<VB + 0 + 5 * 8 * RS>(vraw[1][0][1], vcur); tr_read<VB + 32 + 4 * 8 * RS>(vraw[1][1][0], vcur); }
            if (m == 5) { tr_read<VB + 32 + 5 * 8 * RS>(vraw[1][1][1], vcur); tr_read<VB + 64 + 4 * 8 * RS>(vraw[1][2][0], vcur); tr_read<VB + 64 + 5 * 8 * RS>(vraw[1][2][1], vcur); }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---------------- phase B: PV(t) MFMAs || P swaps of key block 1 || lane-partial max of S(t+1) ----------------
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        float mx = 0.f;
#pragma unroll
        for (int m = 0; m < 12; ++m) {
            // snake over the two query blocks (DM_MFMA_SNAKE, igemm_pers_tile.h): (e0,q0) (e0,q1) (e1,q1) (e1,q0) ... — one operand changes per MFMA
            const int ss = m / 6, e = (m % 6) >> 1, jq = (m & 1) ^ (DM_MFMA_SNAKE ? (e & 1) : 0);
            half8 va, pbv;
            __builtin_memcpy(&va, &vraw[ss][e][0], 8);
            __builtin_memcpy(reinterpret_cast<char*>(&va) + 8, &vraw[ss][e][1], 8);
            __builtin_memcpy(&pbv, &pk[ss][4 * jq], 16);
            oacc[e][jq] = __builtin_amdgcn_mfma_f32_16x16x32_f16(va, pbv, oacc[e][jq], 0, 0, 0);
            PIN(oacc[e][jq]);
            if (m < 4) swap16(pk[1][m], pk[1][4 + m]);       // key block 1 is first read by MFMA m = 6
            if (next && m < 8) {                             // 16 max3 over the 32 scores of this lane's query, two per MFMA
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int o = m * 2 + k;                 // 0..15: scores 2 (o % 8), 2 (o % 8) + 1 of key block o / 8
                    const float a0 = Y[o >> 3][2 * (o & 7)], a1 = Y[o >> 3][2 * (o & 7) + 1];
                    mx = (o == 0) ? vmax2(a0, a1) : vmax3(mx, a0, a1);
                }
                PIN(mx);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (next) {
            if (__builtin_amdgcn_ballot_w64(mx > RESCALE_THR) != 0ull) rescale(Y, mx, false);
        }
        s0 = s1;
    };

    // iterations 0 .. nt-4 issue a full set of pieces; nt-3 only V(nt-1); nt-2, nt-1 nothing
    for (int t = 0; t < ntiles - 4; t += 2) {
        iteration(SA, SB, true, true, true, true);
        iteration(SB, SA, true, true, true, true);
    }
    iteration(SA, SB, true, true, true, true);        // t = nt-4
    iteration(SB, SA, true, false, true, true);       // t = nt-3
    iteration(SA, SB, true, false, false, false);     // t = nt-2
    iteration(SB, SA, false, false, false, false);    // t = nt-1

#pragma unroll
    for (int jq = 0; jq < 2; ++jq) {
        // row d = 40 of O^T (the ones row of V^T) is the softmax denominator: fragment 2, lane group 2, register 0
        const float l = __shfl(oacc[2][jq][0], (2 << 4) | l15);
        const float inv = 1.0f / l;
        const int q = q0 + 16 * jq + l15;
        if (q >= p.Tq) continue;
#pragma unroll
        for (int e = 0; e < EF; ++e) {
            const int d = 16 * e + 4 * lg;
            if (d < D) {
                const half4 o = half4{(f16)(oacc[e][jq][0] * inv), (f16)(oacc[e][jq][1] * inv),
                                      (f16)(oacc[e][jq][2] * inv), (f16)(oacc[e][jq][3] * inv)};
                *reinterpret_cast<half4*>(Ob + (size_t)q * p.ldo + d) = o;
            }
        }
    }
}

}  // namespace

bool attention_qk32_supports(const AttnParams& p) {
    return p.D == 40 && p.Tk >= 256 && (p.Tk % 128) == 0 && p.q_mod == 0;
}

hipError_t launch_attention_qk32(const AttnParams& p, hipStream_t s) {
    if (!attention_qk32_supports(p)) return hipErrorInvalidValue;
    constexpr int QBLK = 4 * QW;
    dim3 grid(((p.Tq + QBLK - 1) / QBLK) * p.heads * p.B), block(NT);
    const size_t lds = NSTG * (size_t)STAGE;
    launch_timed(attn_qk32_kernel, grid, block, lds, s, p);
    return hipGetLastError();
}

}  // namespace dm
