// attention_pp.hip — head_dim-40 self-attention with the two waves of a SIMD held in ANTI-PHASE by construction (r05).
// Reached from `unet(...)`, diffmining/typicality/compute.py:100 (BasicTransformerBlock.attn1 at the 64x64 level; the
// 128x128 level of a 1024-pixel X-ray image, applications/xray/compute.py:88).
//
// attention_pipe.hip pipelines QK^T(t+1) / softmax(t) / PV(t) inside ONE wave and runs three such waves per SIMD; measured
//  its key tile costs the SUM of its matrix time and its VALU time: free-running identical waves fall
// into phase and their MFMA blocks meet MFMA blocks.  Here a workgroup is EIGHT waves = two sets of four (one wave of each
// set per SIMD) that execute the same program half an iteration apart, held there by one s_barrier per phase:
//
//   set A:  M(u-1) | V(u)   | M(u)   | V(u+1) | ...          M(u): S(u+1) = K(u+1) Q'^T (16 MFMAs, operands in registers)
//   set B:         | M(u-1) | V(u)   | M(u)   | ...                O   += V(u)^T P(u)   (12 MFMAs)            -- matrix pipe only
//                                                            V(u): lane-partial max of S(u), lazy rescale, exp2, fp16 pack -> P(u),
//                                                                  the fragment reads of K(u+1) / V(u) from LDS, the wave's LDS-DMA
//                                                                  pieces of K(u+3) / V(u+2)                   -- VALU, LDS, VMEM only
// so whenever one wave of a SIMD issues MFMAs its partner issues everything else.  Same mathematics and operand tricks as
// attention_pipe.hip (S^T = K Q^T so P is the PV B operand as it lies, V^T by ds_read_b64_tr_b16, the running max folded into
// two padded k columns against a constant LDS chunk, the denominator in a ones row of V^T, lazy rescale) -- only ONE score
// tile is live (M(u) writes S(u+1) after V(u) consumed S(u)), so the kernel needs ~170 registers at two waves per SIMD.
// K/V ring of NSETS + 1 stages shared by all waves (256 / 384 queries per workgroup: half / a third of the LDS-DMA bytes per
// query of the four-wave kernel).
// NSETS = 3 (twelve waves, three per SIMD): a single wave issues VALU work at half the rate the pipe accepts it from several
// (v_exp_f32 9.3 cycles alone, 3.4 per instruction from three waves: profiles/r02_probe_valu.txt), so with two sets the V phase of
// ONE wave (32 exp2 + 16 cvt + 18 max + 20 LDS reads + its DMA pieces) is longer than the 28 MFMAs it should hide behind.  With
// three sets the V work is cut in two phases V1 | V2 and at any time one wave of a SIMD is in M, one in V1, one in V2.  Hazards, with I_j the interval between barriers j and j+1, A: V(u) = I_{2u-1}, B: V(u) = I_{2u}:
//   K(u+1), V(u) are read in I_{2u-1} (A) and I_{2u} (B);  their pieces were issued in the V(u-2) phases and are waited for
//   (counted vmcnt) at the end of the V(u-1) phases, i.e. before barriers 2u-2 (A) and 2u-1 (B);  K(u+3) / V(u+2) overwrite the
//   buffers of K(u) / V(u-1), last read by B in I_{2u-2}, and are issued no earlier than I_{2u-1}.
#include "dm_kernels.h"

namespace dm {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int D = 40;
constexpr int KT = 64;                // keys per tile
constexpr int QF = 2;                 // 16-query fragments per wave
constexpr float RESCALE_THR = 8.0f;   // log2 units
constexpr int RS = 96;                // LDS row stride: 5 real chunks + 1 constant chunk
constexpr int TILE = KT * RS;         // 6144
constexpr int KOFF = 0, VOFF = TILE + 32;        // 32 zero bytes behind each tile (K reads overrun a row by 32 B)
constexpr int STAGE = 2 * (TILE + 32);           // 12352
constexpr int KS = 2, EF = 3;

__device__ __attribute__((aligned(16))) const unsigned short g_kconst_pp[8] = {0x3C00, 0x3C00, 0, 0, 0, 0, 0, 0};
__device__ __attribute__((aligned(16))) const unsigned short g_vconst_pp[8] = {0x3C00, 0, 0, 0, 0, 0, 0, 0};
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#define PIN(x) asm volatile("" : "+v"(x))
#define PHASE_BARRIER() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

__device__ __forceinline__ float vmax2(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

template <int OFF>
__device__ __forceinline__ void tr_read(u32x2& out, unsigned base) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(out) : "v"(base), "n"(OFF) : "memory");
}

// PRIO: 0 = no priorities; 1 = s_setprio 1 for the duration of every M phase; 2 = static s_setprio k for set k (the younger sets)
__device__ long long g_attnpp_dbg[3 * 8];

// NSETS: 2 or 3 wave sets of four waves; TIMING: 1 = s_memtime phase timers of one wave per set (debug variant);
// ABL: timing-only ablations (results are garbage): 1 = no MFMAs, 2 = no exp2 / pack, 4 = no fragment reads, 8 = no LDS-DMA, 16 = no max
template <int NSETS, int PRIO, int TIMING, int ABL = 0>
__global__ __launch_bounds__(256 * NSETS, NSETS)
void attn_pp_kernel(AttnParams p) {
    constexpr int NW = 4 * NSETS;         // waves per block
    constexpr int QBLK = NW * 16 * QF;    // queries per block
    constexpr int NSTG = NSETS + 1;       // K/V ring depth
    constexpr int NPW = (12 + NW - 1) / NW;   // LDS-DMA pieces per wave and iteration (some waves one fewer)
    long long dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = TIMING ? (long long)__builtin_readcyclecounter() : 0;
#define TICK(i) do { if (TIMING) { const long long _n = (long long)__builtin_readcyclecounter(); dbg[i] += _n - tlast; tlast = _n; } } while (0)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int set = wid >> 2;
    const bool setB = set == 1;   (void)setB;
    const int l15 = lane & 15;
    const int lg = lane >> 4;
    // XCD-aware block order: one XCD walks consecutive (sample, head) pairs, so all query blocks of a pair share that XCD's L2
    const int nqb = (p.Tq + QBLK - 1) / QBLK;
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
    const int q0 = qblk * QBLK + wid * (16 * QF);
    int kvb = p.kv_slot ? p.kv_slot[b] : (p.slot_div > 0 ? b / p.slot_div : b);
    if (p.n_slots > 0) kvb = kvb < 0 ? 0 : (kvb < p.n_slots ? kvb : p.n_slots - 1);

    const f16* Qb = p.Q + (size_t)b * p.bsq + h * D;
    const f16* Kb = p.K + (size_t)kvb * p.bsk + h * D;
    const f16* Vb = p.V + (size_t)kvb * p.bsv + h * D;
    f16* Ob = p.O + (size_t)b * p.bso + h * D;

    if (tid < 16 * NSTG) {   // the 32-byte zero pads behind the tiles
        const int w = tid & 7, which = tid >> 3;
        *reinterpret_cast<unsigned*>(smem + (which >> 1) * STAGE + ((which & 1) ? VOFF : KOFF) + TILE + w * 4) = 0u;
    }

    // ---- Q' = fp16(sc * q); k columns 40 / 41 carry -m_hi / -m_lo ---------------------------------
    const float sc = p.scale * 1.44269504088896340736f;
    half8 qf[QF][KS];
#pragma unroll
    for (int jq = 0; jq < QF; ++jq) {
        int q = q0 + 16 * jq + l15;
        q = q < p.Tq ? q : p.Tq - 1;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int d = 32 * s + 8 * lg;
            if (d < D) qf[jq][s] = *reinterpret_cast<const half8*>(Qb + (size_t)q * p.ldq + d);
            else qf[jq][s] = half8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 8; ++k) qf[jq][s][k] = (f16)((float)qf[jq][s][k] * sc);
        }
    }

    // ---- LDS-DMA: 6 K + 6 V pieces of 1 KiB per tile pair over eight waves: wave w issues piece w (K pieces 0..5, V pieces 0, 1 for
    //      w = 6, 7) and, set A only, piece 8 + w (V pieces 2..5).  Piece jj covers the 16-byte chunks idx = jj*64 + lane ->
    //      (key = idx / 6, ch = idx % 6); ch 5 is the constant chunk, fetched from a global constant ----------------------------
    const f16* gsrc[NPW];
    int ginc[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int j = wid + NW * i;
        const bool isv = j >= 6;
        const int jj = isv ? j - 6 : j;
        const int idx = (jj % 6) * 64 + lane;
        const int key = idx / 6, ch = idx - key * 6;
        const int ld = isv ? p.ldv : p.ldk;
        if (ch < 5) { gsrc[i] = (isv ? Vb : Kb) + (size_t)key * ld + ch * 8; ginc[i] = KT * ld; }
        else { gsrc[i] = reinterpret_cast<const f16*>(isv ? g_vconst_pp : g_kconst_pp); ginc[i] = 0; }
    }
    bool abl_loop = false;
    // piece i of this wave: a K piece (-> K stage kst) or a V piece (-> V stage vst)
    auto piece = [&](int i, int kst, int vst, bool do_k, bool do_v) __attribute__((always_inline)) {
        const int j = wid + NW * i;
        if (j >= 12 || ((ABL & 8) && abl_loop)) return;
        const bool isv = j >= 6;
        if (isv ? !do_v : !do_k) return;
        char* dst = smem + (isv ? vst * STAGE + VOFF + (j - 6) * 1024 : kst * STAGE + KOFF + j * 1024);
        __builtin_amdgcn_global_load_lds((gptr_t)gsrc[i], (lptr_t)dst, 16, 0, 0);
        gsrc[i] += ginc[i];
    };

    const char* kbase = smem + l15 * RS + 16 * lg;                                               // K fragment reads
    const unsigned vbase = (unsigned)(size_t)(smem + (4 * lg + (l15 >> 2)) * RS + 8 * (l15 & 3));   // V^T transpose reads

    floatx4 oacc[EF][QF];
#pragma unroll
    for (int e = 0; e < EF; ++e)
#pragma unroll
        for (int jq = 0; jq < QF; ++jq) oacc[e][jq] = floatx4{0, 0, 0, 0};
    float m_run[QF] = {0.f, 0.f};

    floatx4 S[4][QF];                  // raw score tile sc*(q.k) - m_run
    unsigned pbu[QF][2][4];            // P as packed fp16 pairs = PV B operand
    half8 kf[KS][4];                   // K(u+1) fragments
    u32x2 vraw[2][EF][2];              // V(u)^T fragments
    float mx[QF] = {0.f, 0.f};

    auto rescale = [&](const float (&mxl)[QF], bool first) __attribute__((always_inline)) {
#pragma unroll
        for (int jq = 0; jq < QF; ++jq) {
            float mown = mxl[jq];
            PIN(mown);
            float m = __builtin_fmaxf(mown, __shfl_xor(mown, 16));
            m = __builtin_fmaxf(m, __shfl_xor(m, 32));
            const float delta = first ? m : __builtin_fmaxf(m, 0.f);
            const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(-delta);
            m_run[jq] += delta;
#pragma unroll
            for (int e = 0; e < EF; ++e) oacc[e][jq] *= alpha;
            if (lg == 1) {
                const f16 mh = (f16)m_run[jq];
                const f16 ml = (f16)(m_run[jq] - (float)mh);
                qf[jq][1][0] = -mh; qf[jq][1][1] = -ml;
            }
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) S[f][jq][r] -= delta;
        }
    };
    auto exp_slice = [&](int i) __attribute__((always_inline)) {
        if (ABL & 2) return;
        const int jq = i >> 3, f = (i >> 1) & 3, rp = (i & 1) * 2;
        const half2v hh = half2v{(f16)__builtin_amdgcn_exp2f(S[f][jq][rp]), (f16)__builtin_amdgcn_exp2f(S[f][jq][rp + 1])};
        unsigned u;
        __builtin_memcpy(&u, &hh, 4);
        PIN(u);
        pbu[jq][f >> 1][(f & 1) * 2 + (rp >> 1)] = u;
    };
    auto lane_max = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int jq = 0; jq < QF; ++jq) {
            float m = vmax2(S[0][jq][0], S[0][jq][1]);
            m = vmax3(m, S[0][jq][2], S[0][jq][3]);
#pragma unroll
            for (int f = 1; f < 4; ++f) { m = vmax3(m, S[f][jq][0], S[f][jq][1]); m = vmax3(m, S[f][jq][2], S[f][jq][3]); }
            mx[jq] = m;
        }
    };
    auto read_k = [&](int st) __attribute__((always_inline)) {
        const char* kcur = kbase + st * STAGE + KOFF;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int f = 0; f < 4; ++f) kf[s][f] = *reinterpret_cast<const half8*>(kcur + 64 * s + f * 16 * RS);
    };
    auto read_v = [&](int st) __attribute__((always_inline)) {
        constexpr int VB = VOFF;
        const unsigned vcur = vbase + (unsigned)(st * STAGE);
        tr_read<VB + 0 + 0 * 1536>(vraw[0][0][0], vcur); tr_read<VB + 0 + 1 * 1536>(vraw[0][0][1], vcur);
        tr_read<VB + 32 + 0 * 1536>(vraw[0][1][0], vcur); tr_read<VB + 32 + 1 * 1536>(vraw[0][1][1], vcur);
        tr_read<VB + 64 + 0 * 1536>(vraw[0][2][0], vcur); tr_read<VB + 64 + 1 * 1536>(vraw[0][2][1], vcur);
        tr_read<VB + 0 + 2 * 1536>(vraw[1][0][0], vcur); tr_read<VB + 0 + 3 * 1536>(vraw[1][0][1], vcur);
        tr_read<VB + 32 + 2 * 1536>(vraw[1][1][0], vcur); tr_read<VB + 32 + 3 * 1536>(vraw[1][1][1], vcur);
        tr_read<VB + 64 + 2 * 1536>(vraw[1][2][0], vcur); tr_read<VB + 64 + 3 * 1536>(vraw[1][2][1], vcur);
    };

    const int ntiles = p.Tk / KT;       // >= 4 (dispatch condition)

    // ---- prologue: K(0..2), V(0..1) in flight; K(0) fragments into registers ------------------------------------------------
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int i = 0; i < NPW; ++i) piece(i, r, r, true, r < 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    read_k(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (PRIO == 2 && set == 1) __builtin_amdgcn_s_setprio(1);
    if (PRIO == 2 && set == 2) __builtin_amdgcn_s_setprio(2);
    PHASE_BARRIER();
    for (int i = 0; i < set; ++i) PHASE_BARRIER();          // set k runs k phases behind set 0
    TICK(7);

    // M phase: S(u+1) = K(u+1) Q'^T (if `next`), O += V(u)^T P(u) (if `pv`)
    auto mphase = [&](const bool next, const bool pv) __attribute__((always_inline)) {
        if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
        if (next && !(ABL & 1)) {
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const int s = m >> 3, f = (m >> 1) & 3, jq = (m & 1) ^ (DM_MFMA_SNAKE ? (f & 1) : 0);      // snake: one operand changes per MFMA (igemm_pers_tile.h)
                S[f][jq] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[s][f], qf[jq][s], s == 0 ? floatx4{0, 0, 0, 0} : S[f][jq], 0, 0, 0);
            }
        }
        if (pv && !(ABL & 1)) {
#pragma unroll
            for (int m = 0; m < 12; ++m) {
                const int ss = m / 6, e = (m % 6) >> 1, jq = (m & 1) ^ (DM_MFMA_SNAKE ? (e & 1) : 0);
                half8 va, pbv;
                __builtin_memcpy(&va, &vraw[ss][e][0], 8);
                __builtin_memcpy(reinterpret_cast<char*>(&va) + 8, &vraw[ss][e][1], 8);
                __builtin_memcpy(&pbv, &pbu[jq][ss][0], 16);
                oacc[e][jq] = __builtin_amdgcn_mfma_f32_16x16x32_f16(va, pbv, oacc[e][jq], 0, 0, 0);
            }
        }
        if (ABL & 2) {               // keep the MFMAs alive when nothing consumes their results
#pragma unroll
            for (int f = 0; f < 4; ++f) { PIN(S[f][0]); PIN(S[f][1]); }
#pragma unroll
            for (int e = 0; e < EF; ++e) { PIN(oacc[e][0]); PIN(oacc[e][1]); }
        }
        if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
        TICK(0);
        PHASE_BARRIER();
        TICK(1);
    };
    // V phase(s) of tile u (scores in S): lane-partial max, lazy rescale, exp2 + pack -> P(u); the fragment reads of K(u+1) / V(u);
    // the wave's LDS-DMA pieces of K(u+3) / V(u+2).  NSETS = 3: two phases, V1 = max + rescale + DMA + first half of the exps,
    // V2 = fragment reads + second half
    int s0 = 0;                         // u % NSTG
    auto vphase = [&](const bool first, const bool next, const bool dma_k, const bool dma_v) __attribute__((always_inline)) {
        const int s1 = (s0 == NSTG - 1) ? 0 : s0 + 1;
        const int s2 = (s1 == NSTG - 1) ? 0 : s1 + 1;
        const int s3 = (s2 == NSTG - 1) ? 0 : s2 + 1;
        if (NSETS == 2 && !(ABL & 4)) { if (next) read_k(s1); read_v(s0); }
        if (!(ABL & 16)) lane_max();
        piece(0, s3, s2, dma_k, dma_v);
        if (first) rescale(mx, true);
        else if (__builtin_amdgcn_ballot_w64(vmax2(mx[0], mx[1]) > RESCALE_THR) != 0ull) rescale(mx, false);
#pragma unroll
        for (int i = 0; i < 8; ++i) exp_slice(i);
        if (NPW > 1) piece(1, s3, s2, dma_k, dma_v);
        // the pieces of the previous iteration have landed (the ones just issued may stay in flight)
        if (dma_k && dma_v) {
            if (NSETS == 2) { if (wid >= 4) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (NSETS == 3) {
            TICK(2);
            PHASE_BARRIER();
            TICK(3);
            if (!(ABL & 4)) { if (next) read_k(s1); read_v(s0); }
        }
#pragma unroll
        for (int i = 8; i < 16; ++i) exp_slice(i);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        TICK(4);
        PHASE_BARRIER();
        TICK(5);
        s0 = s1;
    };

    abl_loop = true;
    mphase(true, false);                               // M(-1): S(0)
    vphase(true, true, true, true);                    // V(0)
    mphase(true, true);                                // M(0)
    for (int u = 1; u < ntiles - 3; ++u) {
        vphase(false, true, true, true);
        mphase(true, true);
    }
    vphase(false, true, false, true); mphase(true, true);      // u = nt-3: only V(nt-1) left to fetch
    vphase(false, true, false, false); mphase(true, true);     // u = nt-2
    vphase(false, false, false, false); mphase(false, true);   // u = nt-1
    for (int i = set; i < NSETS - 1; ++i) PHASE_BARRIER();
    if (PRIO == 2 && set > 0) __builtin_amdgcn_s_setprio(0);
    if (TIMING) {
        TICK(6);
        if (qblk == 1 && h == 1 && b == 1 && lane == 0 && (wid & 3) == 0)
            for (int i = 0; i < 8; ++i) g_attnpp_dbg[set * 8 + i] = dbg[i];
    }
#undef TICK

#pragma unroll
    for (int jq = 0; jq < QF; ++jq) {
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

bool attention_pp_supports(const AttnParams& p) {
    return p.D == 40 && p.Tk >= 256 && (p.Tk % 64) == 0 && p.Tq >= 256;
}

template <int NSETS, int PRIO, int TIMING, int ABL = 0>
static hipError_t launch_pp(const AttnParams& p, hipStream_t s) {
    constexpr int QBLK = 4 * NSETS * 16 * QF;
    dim3 grid(((p.Tq + QBLK - 1) / QBLK) * p.heads * p.B), block(256 * NSETS);
    const size_t lds = (NSETS + 1) * (size_t)STAGE;
    launch_timed((attn_pp_kernel<NSETS, PRIO, TIMING, ABL>), grid, block, lds, s, p);
    return hipGetLastError();
}

// variant (= option attn_pipe): 10 = three sets; 12 = three sets with static priorities (the one the dispatcher uses from 8192 keys).
// The product library holds these two only (r06: the two-set kernel measured -21 % and is gone; its record is DESIGN.md 4f and
// profiles/r05_ab_attn_antiphase_variants.txt).  A build with -DDM_ATTN_PP_ABLATE (tools/ab_attn_pp.py's debug library) adds 4 = two
// sets, 14 / 15 = phase timers and 21... = the timing-only ablations of 12, whose RESULTS ARE GARBAGE by construction.
hipError_t launch_attention_pp(const AttnParams& p, int variant, hipStream_t s) {
    if (!attention_pp_supports(p)) return hipErrorInvalidValue;
    switch (variant) {
        case 10: return launch_pp<3, 0, 0>(p, s);
        case 12: return launch_pp<3, 2, 0>(p, s);
#ifdef DM_ATTN_PP_ABLATE
        case 4: return launch_pp<2, 0, 0>(p, s);
        case 14: return launch_pp<2, 0, 1>(p, s);
        case 15: return launch_pp<3, 0, 1>(p, s);
        case 21: return launch_pp<3, 2, 0, 1>(p, s);       // ablations (timing only)
        case 22: return launch_pp<3, 2, 0, 2>(p, s);
        case 24: return launch_pp<3, 2, 0, 4>(p, s);
        case 28: return launch_pp<3, 2, 0, 8>(p, s);
        case 36: return launch_pp<3, 2, 0, 16>(p, s);
        case 50: return launch_pp<3, 2, 0, 30>(p, s);      // MFMAs only
        case 46: return launch_pp<3, 2, 0, 26>(p, s);      // MFMAs + fragment reads
        case 33: return launch_pp<3, 2, 0, 13>(p, s);      // exp2 + max only
        case 48: return launch_pp<3, 2, 0, 28>(p, s);      // MFMAs + exp2 / pack only
        case 40: return launch_pp<3, 2, 0, 20>(p, s);      // no fragment reads, no max
        case 32: return launch_pp<3, 2, 0, 12>(p, s);      // no fragment reads, no DMA
        case 51: return launch_pp<3, 2, 0, 31>(p, s);      // barriers only
#endif
        default: return hipErrorInvalidValue;
    }
}

extern "C" int dm_debug_attn_pp_timing(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attnpp_dbg), sizeof(long long) * 24) == hipSuccess ? 0 : 1;
}

}  // namespace dm
