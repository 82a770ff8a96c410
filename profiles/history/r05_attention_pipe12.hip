// attention_pipe12.hip — attention_pipe.hip with TWELVE waves per workgroup (r05): the same per-wave schedule, but the K/V ring is
// shared by 384 queries, so a wave issues ONE LDS-DMA piece per key tile instead of three (the SIMD's single issue port is the
// bound: DESIGN.md section 4f).  Software-pipelined flash attention for the head_dim-40 self-attention layers
// (4096 tokens at 64x64 latents: 60 % of the U-Net's attention time) reached from `unet(...)`,
// diffmining/typicality/compute.py:100.  Same mathematics and operand tricks as attention.hip
// (S^T = K Q^T so P is directly the PV B operand, V^T by ds_read_b64_tr_b16, running max folded into
// two padded k columns, denominator in a ones row of V^T, lazy rescale); what changes is the schedule.
//
// Measured on attention.hip (s_memtime phase timers): a key tile costs a wave ~2600 cycles of which
// ~1100 are VALU issue (175 VALU instructions + 32 exp2), i.e. the kernel is VALU-issue bound with two
// waves per SIMD, and the MFMA phases do not overlap the softmax of the same wave.  Here:
//   * LDS rows are 96 bytes: 40 real halfs + one constant 16-byte chunk ({1,1,0..} for K, {1,0,0..}
//     for V) that the LDS-DMA fetches from a global constant, so the padded operands need no per-lane
//     address selects — every LDS read is lane base + immediate (no address VALU in the loop);
//   * the loop is unrolled by two with the score registers ping-ponged (no register copies) and
//     specialised for full tiles (Tk % 128 == 0; other shapes use attention.hip);
//   * inside one wave, in program order:  phase A  S(t+1) = K(t+1) Q^T MFMAs interleaved with the exp2 /
//     convert of S(t), the LDS-DMA pieces of tiles t+2 / t+1 and the V^T transpose reads;  phase B
//     O += V(t)^T P(t) MFMAs interleaved with the lane-partial max of S(t+1).  K tiles are fetched one
//     iteration ahead of V tiles (two K + two V buffers, one barrier per iteration);
//   * the per-row maximum is only reduced across lanes when a rescale actually happens (one ballot per
//     iteration decides), and a rescale fixes the already computed S(t+1) up in place.
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
constexpr int NWV = 12;                // waves per block
constexpr int NT = 64 * NWV;          // threads per block
constexpr int QF = 2;                 // 16-query fragments per wave
constexpr float RESCALE_THR = 8.0f;   // log2 units
constexpr int RS = 96;                // LDS row stride: 5 real chunks + 1 constant chunk
constexpr int TILE = KT * RS;         // 6144
constexpr int KOFF = 0, VOFF = TILE + 32;        // 32 zero bytes behind each tile (K reads overrun a row by 32 B)
constexpr int STAGE = 2 * (TILE + 32);           // 12352
constexpr int NSTG = 3;               // K/V ring depth: K is fetched three, V two tiles ahead of their use
constexpr int KS = 2, EF = 3;

__device__ __attribute__((aligned(16))) const unsigned short g_kconst12[8] = {0x3C00, 0x3C00, 0, 0, 0, 0, 0, 0};
__device__ __attribute__((aligned(16))) const unsigned short g_vconst12[8] = {0x3C00, 0, 0, 0, 0, 0, 0, 0};
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#ifdef DM_ATTN_TIMING
__device__ long long g_attnp12_dbg[16];
__device__ unsigned long long g_attnp12_span[2] = {~0ull, 0ull};
#define TICK(i) do { const long long _n = (long long)__builtin_readcyclecounter(); dbg[i] += _n - tlast; tlast = _n; } while (0)
#else
#define TICK(i) do {} while (0)
#endif

#define PIN(x) asm volatile("" : "+v"(x))

// max through inline asm: __builtin_fmaxf (llvm.maxnum) first canonicalises operands that come out of MFMAs
// or asm (v_max x, x), 2-3 extra VALU per row block and tile; scores are never NaN here
__device__ __forceinline__ float vmax2(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

template <int OFF>
__device__ __forceinline__ void tr_read(u32x2& out, unsigned base) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(out) : "v"(base), "n"(OFF) : "memory");
}

#ifndef DM_ATTN_OCC
#define DM_ATTN_OCC 3
#endif
__global__ __launch_bounds__(NT, DM_ATTN_OCC)
void attn_pipe12_kernel(AttnParams p) {
#ifdef DM_ATTN_TIMING
    long long dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = (long long)__builtin_readcyclecounter();
    if (threadIdx.x == 0) atomicMin(&g_attnp12_span[0], (unsigned long long)tlast);
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15;
    const int lg = lane >> 4;
    // XCD-aware block order: one XCD walks consecutive (sample, head) pairs, so all query blocks of a
    // pair (which stream the same K/V) share that XCD's L2.
    const int nqb = (p.Tq + NWV * 16 * QF - 1) / (NWV * 16 * QF);
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
    const int q0 = qblk * (NWV * 16 * QF) + wid * (16 * QF);
    int kvb = p.kv_slot ? p.kv_slot[b] : (p.slot_div > 0 ? b / p.slot_div : b);
    if (p.n_slots > 0) kvb = kvb < 0 ? 0 : (kvb < p.n_slots ? kvb : p.n_slots - 1);      // memory safety: never beyond the registered prompts

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

    // ---- LDS-DMA: 6 K + 6 V pieces of 1 KiB per tile, 3 per wave (j = wid + 4 i; j < 6 is a K piece);
    //      piece jj covers the 16-byte chunks idx = jj*64 + lane -> (key = idx / 6, ch = idx % 6);
    //      ch 5 is the constant chunk, fetched from a global constant -------------------------------
    const f16* gsrc[1];
    int ginc[1];
#pragma unroll
    for (int i = 0; i < 1; ++i) {
        const int j = wid;
        const bool isv = j >= 6;
        const int jj = isv ? j - 6 : j;
        const int idx = jj * 64 + lane;
        const int key = idx / 6, ch = idx - key * 6;
        const int ld = isv ? p.ldv : p.ldk;
        if (ch < 5) { gsrc[i] = (isv ? Vb : Kb) + (size_t)key * ld + ch * 8; ginc[i] = KT * ld; }
        else { gsrc[i] = reinterpret_cast<const f16*>(isv ? g_vconst12 : g_kconst12); ginc[i] = 0; }
    }
    auto piece_is_v = [&](int i) __attribute__((always_inline)) { return wid >= 6; };
    auto piece = [&](int i, int kst, int vst) __attribute__((always_inline)) {
        const int j = wid;
        char* dst = smem + ((j >= 6) ? vst * STAGE + VOFF + (j - 6) * 1024 : kst * STAGE + KOFF + j * 1024);
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

    floatx4 SA[4][QF], SB[4][QF];      // raw score tiles sc*(q.k) - m_run, ping-ponged
    unsigned pbu[QF][2][4];            // P as packed fp16 pairs = PV B operand

    // advance the running max (rare): rescale O, refresh the -m columns of Q', and fix the already
    // computed score tile X (computed against the old max) up in place
    auto rescale = [&](floatx4 (&X)[4][QF], const float (&mxl)[QF], bool first) __attribute__((always_inline)) {
#pragma unroll
        for (int jq = 0; jq < QF; ++jq) {
            float mown = mxl[jq];
            PIN(mown);          // keeps the cross-lane reduction inside the rare branch (it was being speculated into the loop)
            float mx = __builtin_fmaxf(mown, __shfl_xor(mown, 16));
            mx = __builtin_fmaxf(mx, __shfl_xor(mx, 32));
            const float delta = first ? mx : __builtin_fmaxf(mx, 0.f);
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
                for (int r = 0; r < 4; ++r) X[f][jq][r] -= delta;
        }
    };
    // exp slice i (0..15): two scores of row block jq = i / 8 -> one packed fp16 pair of the PV B operand
    auto exp_slice = [&](const floatx4 (&X)[4][QF], int i) __attribute__((always_inline)) {
        const int jq = i >> 3, f = (i >> 1) & 3, rp = (i & 1) * 2;
        const half2v hh = half2v{(f16)__builtin_amdgcn_exp2f(X[f][jq][rp]), (f16)__builtin_amdgcn_exp2f(X[f][jq][rp + 1])};
        unsigned u;
        __builtin_memcpy(&u, &hh, 4);
        PIN(u);
        pbu[jq][f >> 1][(f & 1) * 2 + (rp >> 1)] = u;
    };

    const int ntiles = p.Tk / KT;       // even, >= 4 (dispatch condition)

    // ---- prologue: K(0) -> S(0), first running max -------------------------------------------------
    if (!piece_is_v(0)) piece(0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    piece(0, 1, 0);              // K(1) -> stage 1, V(0) -> stage 0
    piece(0, 2, 1);              // K(2) -> stage 2, V(1) -> stage 1
    {
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int jq = 0; jq < QF; ++jq) SA[f][jq] = floatx4{0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const half8 kf = *reinterpret_cast<const half8*>(kbase + KOFF + 64 * s + f * 16 * RS);
#pragma unroll
                for (int jq = 0; jq < QF; ++jq) SA[f][jq] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[jq][s], SA[f][jq], 0, 0, 0);
            }
        float mx[QF];
#pragma unroll
        for (int jq = 0; jq < QF; ++jq) {
            float m = SA[0][jq][0];
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) m = __builtin_fmaxf(m, SA[f][jq][r]);
            mx[jq] = m;
        }
        rescale(SA, mx, true);
    }
    TICK(7);

    // One iteration t: scores of tile t in X, tile t+1 into Y.  Ring of NSTG = 3 stages, s0 = t % 3:
    //   K(t+1) sits in stage (t+1)%3, V(t) in stage s0; DMA: K(t+3) -> stage s0, V(t+2) -> stage (t+2)%3.
    //   `wait3`: the three pieces this wave issued in the previous iteration may stay in flight.
    int s0 = 0;
    auto iteration = [&](floatx4 (&X)[4][QF], floatx4 (&Y)[4][QF], const bool next, const bool dma_k, const bool dma_v,
                         const bool wait3) __attribute__((always_inline)) {
        constexpr int KB = KOFF, VB = VOFF;
        const int s1 = (s0 == NSTG - 1) ? 0 : s0 + 1;
        const int s2 = (s1 == NSTG - 1) ? 0 : s1 + 1;
        const char* kcur = kbase + s1 * STAGE;                // K(t+1)
        const unsigned vcur = vbase + (unsigned)(s0 * STAGE); // V(t)
        if (wait3) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TICK(0);
        // raw barrier: __syncthreads() carries a fence that the compiler lowers to vmcnt(0), which would undo
        // the counted wait above (all LDS reads of the previous iteration were already waited for)
        asm volatile("s_barrier" ::: "memory");
        TICK(1);
        // ---------------- phase A: S(t+1) MFMAs || exp of S(t) || DMA issue || V^T reads ----------------
        half8 kf[KS][4];
        if (next) {
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int f = 0; f < 4; ++f) kf[s][f] = *reinterpret_cast<const half8*>(kcur + KB + 64 * s + f * 16 * RS);
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int jq = 0; jq < QF; ++jq) Y[f][jq] = floatx4{0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) exp_slice(X, i);         // cover the latency of the K fragment reads
        __builtin_amdgcn_sched_barrier(0);
        u32x2 vraw[2][EF][2];
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const int s = m >> 3, f = (m >> 1) & 3, jq = m & 1;
            if (next) {
                Y[f][jq] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[s][f], qf[jq][s], Y[f][jq], 0, 0, 0);
                PIN(Y[f][jq]);
            }
            if (m < 12) exp_slice(X, 4 + m);
            // the wave's three LDS-DMA pieces, spread out (a piece blocks the issuing wave ~170 cycles)
            if (m == 5) {
                if (piece_is_v(0) ? dma_v : dma_k) piece(0, s0, s2);
            }
            // V(t)^T fragments: 12 transpose reads behind the last four MFMAs; offset = 32 e + (2 ss + hh) 16 RS
            if (m == 12) { tr_read<VB + 0 + 0 * 1536>(vraw[0][0][0], vcur); tr_read<VB + 0 + 1 * 1536>(vraw[0][0][1], vcur); tr_read<VB + 32 + 0 * 1536>(vraw[0][1][0], vcur); }
            if (m == 13) { tr_read<VB + 32 + 1 * 1536>(vraw[0][1][1], vcur); tr_read<VB + 64 + 0 * 1536>(vraw[0][2][0], vcur); tr_read<VB + 64 + 1 * 1536>(vraw[0][2][1], vcur); }
            if (m == 14) { tr_read<VB + 0 + 2 * 1536>(vraw[1][0][0], vcur); tr_read<VB + 0 + 3 * 1536>(vraw[1][0][1], vcur); tr_read<VB + 32 + 2 * 1536>(vraw[1][1][0], vcur); }
            if (m == 15) { tr_read<VB + 32 + 3 * 1536>(vraw[1][1][1], vcur); tr_read<VB + 64 + 2 * 1536>(vraw[1][2][0], vcur); tr_read<VB + 64 + 3 * 1536>(vraw[1][2][1], vcur); }
            __builtin_amdgcn_sched_barrier(0);
        }
        TICK(3);
        // ---------------- phase B: PV(t) MFMAs || lane-partial max of S(t+1) ----------------
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        TICK(4);
        float mx[QF] = {0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 12; ++m) {
            const int ss = m / 6, e = (m % 6) >> 1, jq = m & 1;
            half8 va, pbv;
            __builtin_memcpy(&va, &vraw[ss][e][0], 8);
            __builtin_memcpy(reinterpret_cast<char*>(&va) + 8, &vraw[ss][e][1], 8);
            __builtin_memcpy(&pbv, &pbu[jq][ss][0], 16);
            oacc[e][jq] = __builtin_amdgcn_mfma_f32_16x16x32_f16(va, pbv, oacc[e][jq], 0, 0, 0);
            PIN(oacc[e][jq]);
            if (next && m < 8) {                             // 8 max3 per row block, two per MFMA
                const int j2 = m >> 2;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int o = (m & 3) * 2 + k;           // 0..7: scores 2o, 2o+1 of row block j2
                    const float a0 = Y[o >> 1][j2][(o & 1) * 2], a1 = Y[o >> 1][j2][(o & 1) * 2 + 1];
                    mx[j2] = (o == 0) ? vmax2(a0, a1) : vmax3(mx[j2], a0, a1);
                }
                PIN(mx[j2]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        TICK(5);
        if (next) {
            if (__builtin_amdgcn_ballot_w64(vmax2(mx[0], mx[1]) > RESCALE_THR) != 0ull) rescale(Y, mx, false);
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

#ifdef DM_ATTN_TIMING
    TICK(6);
    if (threadIdx.x == 0) atomicMax(&g_attnp12_span[1], (unsigned long long)tlast);
    if (qblk == 3 && h == 1 && b == 2 && (tid & 63) == 0 && wid < 2)
        for (int i = 0; i < 8; ++i) g_attnp12_dbg[wid * 8 + i] = dbg[i];
#endif
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

bool attention_pipe12_supports(const AttnParams& p) {
    return p.D == 40 && p.Tk >= 256 && (p.Tk % 128) == 0 && p.Tq >= 256;
}

hipError_t launch_attention_pipe12(const AttnParams& p, hipStream_t s) {
    if (!attention_pipe12_supports(p)) return hipErrorInvalidValue;
    constexpr int QBLK = NWV * 16 * QF;
    dim3 grid(((p.Tq + QBLK - 1) / QBLK) * p.heads * p.B), block(NT);
    const size_t lds = NSTG * (size_t)STAGE;
    launch_timed(attn_pipe12_kernel, grid, block, lds, s, p);
    return hipGetLastError();
}

#ifdef DM_ATTN_TIMING
extern "C" int dm_debug_attn12_timing(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attnp12_dbg), sizeof(long long) * 16) == hipSuccess ? 0 : 1;
}
extern "C" int dm_debug_attn12_occupancy() {
    int nb = -1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)attn_pipe12_kernel, NT, NSTG * (size_t)STAGE) != hipSuccess) return -1;
    return nb;
}
extern "C" int dm_debug_attn12_span(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attnp12_span), 16) != hipSuccess) return 1;
    if (reset) { unsigned long long z[2] = {~0ull, 0ull}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_attnp12_span), z, 16) != hipSuccess) return 1; }
    return 0;
}
#endif

}  // namespace dm
