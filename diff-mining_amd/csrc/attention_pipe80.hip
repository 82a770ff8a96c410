// attention_pipe80.hip — the software-pipelined flash attention of attention_pipe.hip for head_dim 80 (the 1024-token
// self-attention of the 32x32 level, 4096 tokens at 1024 px) reached from `unet(...)`,
// diffmining/typicality/compute.py:100.  Same mathematics and operand tricks (S^T = K Q^T so P is directly the PV B
// operand, V^T by ds_read_b64_tr_b16, running max folded into two padded k columns, denominator in a ones row of V^T,
// lazy rescale) and the same schedule, with the head_dim-dependent geometry as template constants, and one change of
// layout: the LDS rows are bare (160 bytes at D = 80, no constant chunk).  The running max enters the score MFMAs as the
// initial value of their accumulators (fp32 -m instead of the fp16 hi / lo pair in two padded k columns) and the row block
// of V^T that holds the ones row (rows 80..95: ones, then fifteen zero rows) is a register constant.  That removes 9 % of
// the LDS-DMA bytes, four of the 24 transpose reads per tile and — the part that pays — the bank conflicts of the V^T
// transpose reads: eight rows of 176 bytes overlap in the 64 banks, eight rows of 160 bytes do not (PMC: 12 % -> 4 % LDS
// conflict cycles over the attention family).  Same-box A/B against the constant-chunk layout (`attn_pipe` = 3):
// 0.61 -> 0.56 ms at 160 x 8 x 1024, 0.98 -> 0.92 ms at 20 x 8 x 4096.
//   * k steps of the score MFMA: ceil(D / 32) = 3 (k 80..95 multiply zero columns of Q');
//   * O^T row blocks: 5 from LDS + the constant one; row D is the softmax denominator;
//   * 10 K + 10 V LDS-DMA pieces per 64-key tile, five per wave;
//   * registers (two blocks of four waves per CU, 243 VGPRs): the V^T fragments of the second 32-key half are read
//     during the PV MFMAs of the first half, the K fragments of the third k step after the first step's MFMAs.
// A 64-key tile costs a wave 48 MFMAs (768 matrix cycles) against ~75 VALU + 32 exp2 (~690 issue cycles).  Measured
// (tools/ab_attn.py attn_pipe 2 1, same box) against attn_kernel<80>: 0.77 -> 0.56 ms per launch at 160 x 8 heads x 1024
// tokens (577 -> 790 TFLOP/s), 1.32 -> 0.92 ms at 20 x 8 x 4096.  Phase timers (tools/attn_timing.py 80 1024): an iteration
// is ~2500 cycles per wave (phase A 1500, phase B 600), and the first wait of a block (K(0) + Q from a busy memory
// system) ~9700 cycles = a fifth of a 16-tile block's life, hidden only by the one other block resident on the CU.
#include "dm_kernels.h"

#include <type_traits>
#include <utility>

namespace dm {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int KT = 64;                // keys per tile
constexpr int NT = 256;               // threads per block
constexpr int QF = 2;                 // 16-query fragments per wave
constexpr float RESCALE_THR = 8.0f;   // log2 units
constexpr int NSTG = 3;               // K/V ring depth: K is fetched three, V two tiles ahead of their use

// CC = 1: rows carry a constant 16-byte chunk behind the data ({1,1,0..} for K: the k columns that take -m_hi / -m_lo from
// Q'; {1,0..} for V: the ones row of V^T), as attention_pipe.hip does.  CC = 0: bare rows; the running max enters as the
// initial value of the score accumulators and the ones row block of V^T is a register constant (D % 16 == 0).
template <int D, int CC>
struct Geo {
    static constexpr int CH = D / 8;                       // real 16-byte chunks per row
    static constexpr int RS = (CH + CC) * 16;              // LDS row stride (bytes)
    static constexpr int KS = (D + 8 * CC + 31) / 32;      // k steps of S^T = K Q^T
    static constexpr int EF = (D + 1 + 15) / 16;           // 16-row blocks of O^T
    static constexpr int EFV = CC ? EF : D / 16;           // ... of which read from LDS
    static constexpr int S_M = D / 32, LG_M = (D % 32) / 8;   // where k = D, D + 1 live in the Q' fragments
    static constexpr int E_L = D / 16, LG_L = (D % 16) / 4;   // where row D of O^T lives in the accumulators
    static constexpr int TILE = KT * RS;
    static constexpr int PAD = 32;                         // zero bytes behind each tile (fragment reads overrun a row)
    static constexpr int KOFF = 0, VOFF = TILE + PAD;
    static constexpr int STAGE = 2 * (TILE + PAD);
    static constexpr int NP = KT * (CH + CC) / 64;         // 1 KiB LDS-DMA pieces per operand tile
    static constexpr int NPW = (2 * NP + 3) / 4;           // piece slots per wave
    static constexpr int SCRATCH = NSTG * STAGE;           // 1 KiB target of the dummy slots
    static constexpr int LDS = NSTG * STAGE + 1024;
    static_assert(D % 8 == 0 && (KT * (CH + CC)) % 64 == 0 && (CC || D % 16 == 0), "tile must be whole pieces");
    static_assert(16 * 3 + 64 * (KS - 1) + 16 <= RS + PAD && 32 * (EFV - 1) + 8 * 3 + 8 <= RS + PAD, "fragment overrun must stay inside the pad");
};

__device__ __attribute__((aligned(16))) const unsigned short g_kconst80[8] = {0x3C00, 0x3C00, 0, 0, 0, 0, 0, 0};
__device__ __attribute__((aligned(16))) const unsigned short g_vconst80[8] = {0x3C00, 0, 0, 0, 0, 0, 0, 0};
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#define PIN(x) asm volatile("" : "+v"(x))

#ifdef DM_ATTN_TIMING
__device__ long long g_attnp80_dbg[16];
__device__ unsigned long long g_attnp80_span[2] = {~0ull, 0ull};
#define TICK(i) do { const long long _n = (long long)__builtin_readcyclecounter(); dbg[i] += _n - tlast; tlast = _n; } while (0)
#else
#define TICK(i) do {} while (0)
#endif

__device__ __forceinline__ float vmax2(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

template <int OFF>
__device__ __forceinline__ void tr_read(u32x2& out, unsigned base) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(out) : "v"(base), "n"(OFF) : "memory");
}

// compile-time loop: f(std::integral_constant<int, i>) for i in [0, N)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <int D, int CC>
__global__ __launch_bounds__(NT, 2)
void attn_pipe80_kernel(AttnParams p) {
    using G = Geo<D, CC>;
    constexpr int RS = G::RS, KS = G::KS, EF = G::EF, EFV = G::EFV, STAGE = G::STAGE, KOFF = G::KOFF, VOFF = G::VOFF, NPW = G::NPW;
#ifdef DM_ATTN_TIMING
    long long dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = (long long)__builtin_readcyclecounter();
    if (threadIdx.x == 0) atomicMin(&g_attnp80_span[0], (unsigned long long)tlast);
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15;
    const int lg = lane >> 4;
    // XCD-aware block order: one XCD walks consecutive (sample, head) pairs, so all query blocks of a
    // pair (which stream the same K/V) share that XCD's L2.
    const int nqb = (p.Tq + 64 * QF - 1) / (64 * QF);
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
    const int q0 = qblk * (64 * QF) + wid * (16 * QF);
    int kvb = p.kv_slot ? p.kv_slot[b] : (p.slot_div > 0 ? b / p.slot_div : b);
    if (p.n_slots > 0) kvb = kvb < 0 ? 0 : (kvb < p.n_slots ? kvb : p.n_slots - 1);      // memory safety: never beyond the registered prompts

    const f16* Qb = p.Q + (size_t)b * p.bsq + h * D;
    const f16* Kb = p.K + (size_t)kvb * p.bsk + h * D;
    const f16* Vb = p.V + (size_t)kvb * p.bsv + h * D;
    f16* Ob = p.O + (size_t)b * p.bso + h * D;

    if (tid < 16 * NSTG) {   // the 32-byte zero pads behind the tiles
        const int w = tid & 7, which = tid >> 3;
        *reinterpret_cast<unsigned*>(smem + (which >> 1) * STAGE + ((which & 1) ? VOFF : KOFF) + G::TILE + w * 4) = 0u;
    }

    // ---- LDS-DMA: NP K + NP V pieces of 1 KiB per tile, slot i of wave w is piece j = w + 4 i (j < NP: K; j < 2 NP: V;
    //      beyond: a dummy); piece jj covers the 16-byte chunks idx = jj*64 + lane -> (key = idx / (CH+1), ch = idx % (CH+1));
    //      ch == CH is the constant chunk, fetched from a global constant ------------------------------
    const f16* gsrc[NPW];
    int ginc[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int j = wid + 4 * i;
        const bool isv = j >= G::NP;
        const int jj = isv ? j - G::NP : j;
        const int idx = jj * 64 + lane;
        const int key = idx / (G::CH + CC), ch = idx - key * (G::CH + CC);
        const int ld = isv ? p.ldv : p.ldk;
        if (j < 2 * G::NP && ch < G::CH) { gsrc[i] = (isv ? Vb : Kb) + (size_t)key * ld + ch * 8; ginc[i] = KT * ld; }
        else { gsrc[i] = reinterpret_cast<const f16*>(isv ? g_vconst80 : g_kconst80); ginc[i] = 0; }
    }
    auto piece_is_v = [&](int i) __attribute__((always_inline)) { return wid + 4 * i >= G::NP; };   // dummies count as V
    auto piece = [&](int i, int kst, int vst) __attribute__((always_inline)) {
        const int j = wid + 4 * i;
        char* dst = smem + ((j >= 2 * G::NP) ? G::SCRATCH
                            : (j >= G::NP) ? vst * STAGE + VOFF + (j - G::NP) * 1024 : kst * STAGE + KOFF + j * 1024);
        __builtin_amdgcn_global_load_lds((gptr_t)gsrc[i], (lptr_t)dst, 16, 0, 0);
        gsrc[i] += ginc[i];
    };

    const char* kbase = smem + l15 * RS + 16 * lg;                                               // K fragment reads
    const unsigned vbase = (unsigned)(size_t)(smem + (4 * lg + (l15 >> 2)) * RS + 8 * (l15 & 3));   // V^T transpose reads

    const float sc = p.scale * 1.44269504088896340736f;
    half8 qf[QF][KS];                  // Q' = fp16(sc * q), loaded in the prologue
    half8 ones_a;                      // A operand of the constant V^T row block (CC = 0): lane row 0 = ones
    {
        const f16 o = (l15 == 0) ? (f16)1.0f : (f16)0.0f;
        ones_a = half8{o, o, o, o, o, o, o, o};
    }
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
            PIN(mown);
            float mx = __builtin_fmaxf(mown, __shfl_xor(mown, 16));
            mx = __builtin_fmaxf(mx, __shfl_xor(mx, 32));
            const float delta = first ? mx : __builtin_fmaxf(mx, 0.f);
            const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(-delta);
            m_run[jq] += delta;
#pragma unroll
            for (int e = 0; e < EF; ++e) oacc[e][jq] *= alpha;
            if constexpr (CC) {
                if (lg == G::LG_M) {
                    const f16 mh = (f16)m_run[jq];
                    const f16 ml = (f16)(m_run[jq] - (float)mh);
                    qf[jq][G::S_M][0] = -mh; qf[jq][G::S_M][1] = -ml;
                }
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

    // ---- prologue: K(0), the Q rows, then (K(1), V(0)) and (K(2), V(1)) all go out before the first wait; vmcnt retires in
    //      order, so "at most two batches outstanding" means K(0) (and Q) have landed ---------------------
#pragma unroll
    for (int i = 0; i < NPW; ++i) if (!piece_is_v(i)) piece(i, 0, 0);
    // ---- Q' = fp16(sc * q); k columns D / D + 1 carry -m_hi / -m_lo.  The loads go out between the K(0) pieces and the
    //      next two DMA batches, so one memory round trip covers all of them --------------------------------
#pragma unroll
    for (int jq = 0; jq < QF; ++jq) {
        int q = q0 + 16 * jq + l15;
        q = q < p.Tq ? q : p.Tq - 1;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int d = 32 * s + 8 * lg;
            if (d < D) qf[jq][s] = *reinterpret_cast<const half8*>(Qb + (size_t)q * p.ldq + d);
            else qf[jq][s] = half8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
#pragma unroll
    for (int i = 0; i < NPW; ++i) piece(i, 1, 0);              // K(1) -> stage 1, V(0) -> stage 0
#pragma unroll
    for (int i = 0; i < NPW; ++i) piece(i, 2, 1);              // K(2) -> stage 2, V(1) -> stage 1
#pragma unroll
    for (int jq = 0; jq < QF; ++jq)
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int k = 0; k < 8; ++k) qf[jq][s][k] = (f16)((float)qf[jq][s][k] * sc);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * NPW) : "memory");
    asm volatile("s_barrier" ::: "memory");                    // raw: __syncthreads() would wait for vmcnt(0)
    TICK(6);
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
    //   `waitn`: the NPW pieces this wave issued in the previous iteration may stay in flight.
    int s0 = 0;
    auto iteration = [&](floatx4 (&X)[4][QF], floatx4 (&Y)[4][QF], const bool next, const bool dma_k, const bool dma_v,
                         const bool waitn) __attribute__((always_inline)) {
        const int s1 = (s0 == NSTG - 1) ? 0 : s0 + 1;
        const int s2 = (s1 == NSTG - 1) ? 0 : s1 + 1;
        const char* kcur = kbase + s1 * STAGE;                // K(t+1)
        const unsigned vcur = vbase + (unsigned)(s0 * STAGE); // V(t)
        if (waitn) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TICK(0);
        // raw barrier: __syncthreads() carries a fence that the compiler lowers to vmcnt(0), which would undo
        // the counted wait above (all LDS reads of the previous iteration were already waited for)
        asm volatile("s_barrier" ::: "memory");
        TICK(1);
        // ---------------- phase A: S(t+1) MFMAs || exp of S(t) || DMA issue || V^T reads (first key half) ----------------
        half8 kf[KS][4];
        if (next) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int f = 0; f < 4; ++f) kf[s][f] = *reinterpret_cast<const half8*>(kcur + KOFF + 64 * s + f * 16 * RS);
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int jq = 0; jq < QF; ++jq) {
                    const float ini = CC ? 0.f : -m_run[jq];          // CC = 0: scores start from -m (fp32, not hi / lo halves)
                    Y[f][jq] = floatx4{ini, ini, ini, ini};
                }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) exp_slice(X, i);         // cover the latency of the K fragment reads
        __builtin_amdgcn_sched_barrier(0);
        u32x2 vraw[2][EF][2];
        constexpr int NMA = KS * 4 * QF;                      // 24
        static_for<NMA>([&](auto M) __attribute__((always_inline)) {
            constexpr int m = decltype(M)::value;
            constexpr int s = m >> 3, f = (m >> 1) & 3, jq = (m & 1) ^ (DM_MFMA_SNAKE ? (f & 1) : 0);       // snake: one operand changes per MFMA (igemm_pers_tile.h)
            if (next) {
                if constexpr (KS > 2 && m == 8) {             // k steps >= 2: fragments fetched once step 0 has issued
#pragma unroll
                    for (int s2k = 2; s2k < KS; ++s2k)
#pragma unroll
                        for (int ff = 0; ff < 4; ++ff) kf[s2k][ff] = *reinterpret_cast<const half8*>(kcur + KOFF + 64 * s2k + ff * 16 * RS);
                }
                Y[f][jq] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[s][f], qf[jq][s], Y[f][jq], 0, 0, 0);
                PIN(Y[f][jq]);
            }
            if constexpr ((m & 1) == 0 && m / 2 < 12) exp_slice(X, 4 + m / 2);
            // the wave's LDS-DMA pieces, spread out (a piece blocks the issuing wave ~170 cycles)
            // (moving half of them among the PV MFMAs of phase B: +-1 %, measured)
            if constexpr ((m & 3) == 1 && (m >> 2) < NPW) {
                constexpr int i = m >> 2;
                if (piece_is_v(i) ? dma_v : dma_k) piece(i, s0, s2);
            }
            // V(t)^T fragments of keys 0..31: 2 EF transpose reads behind the last four MFMAs;
            // offset = 32 e + (2 ss + hh) 16 RS
            if constexpr (m >= NMA - 4) {
                constexpr int per = (2 * EFV + 3) / 4;
                static_for<per>([&](auto R) __attribute__((always_inline)) {
                    constexpr int r = (m - (NMA - 4)) * per + decltype(R)::value;
                    if constexpr (r < 2 * EFV) {
                        constexpr int e = r >> 1, hh = r & 1;
                        tr_read<VOFF + 32 * e + hh * 16 * RS>(vraw[0][e][hh], vcur);
                    }
                });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        TICK(3);
        // ---------------- phase B: PV(t) MFMAs || lane-partial max of S(t+1) || V^T reads (second key half) ----------------
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        TICK(4);
        static_for<2 * EFV>([&](auto R) __attribute__((always_inline)) {
            constexpr int r = decltype(R)::value, e = r >> 1, hh = r & 1;
            tr_read<VOFF + 32 * e + (2 + hh) * 16 * RS>(vraw[1][e][hh], vcur);
        });
        __builtin_amdgcn_sched_barrier(0);
        float mx[QF] = {0.f, 0.f};
        constexpr int NMB = 2 * EF * QF;                      // 24
        static_for<NMB>([&](auto M) __attribute__((always_inline)) {
            constexpr int m = decltype(M)::value;
            constexpr int ss = m / (EF * QF), e = (m % (EF * QF)) >> 1, jq = (m & 1) ^ (DM_MFMA_SNAKE ? (e & 1) : 0);
            if constexpr (m == EF * QF) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            half8 va, pbv;
            if constexpr (e < EFV) {
                __builtin_memcpy(&va, &vraw[ss][e][0], 8);
                __builtin_memcpy(reinterpret_cast<char*>(&va) + 8, &vraw[ss][e][1], 8);
            } else {
                va = ones_a;                                  // rows D .. D + 15 of V^T: the ones row and fifteen zero rows
            }
            __builtin_memcpy(&pbv, &pbu[jq][ss][0], 16);
            oacc[e][jq] = __builtin_amdgcn_mfma_f32_16x16x32_f16(va, pbv, oacc[e][jq], 0, 0, 0);
            PIN(oacc[e][jq]);
            if constexpr (m < 8) {                            // 8 max3 per row block, two per MFMA
                if (next) {
                    constexpr int j2 = m >> 2;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int o = (m & 3) * 2 + k;        // 0..7: scores 2o, 2o+1 of row block j2
                        const float a0 = Y[o >> 1][j2][(o & 1) * 2], a1 = Y[o >> 1][j2][(o & 1) * 2 + 1];
                        mx[j2] = (o == 0) ? vmax2(a0, a1) : vmax3(mx[j2], a0, a1);
                    }
                    PIN(mx[j2]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        TICK(5);
        if (next) {
            if (__builtin_amdgcn_ballot_w64(vmax2(mx[0], mx[1]) > RESCALE_THR) != 0ull) rescale(Y, mx, false);
        }
        TICK(2);
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
    if (threadIdx.x == 0) atomicMax(&g_attnp80_span[1], (unsigned long long)tlast);
    if (qblk == 3 && h == 1 && b == 2 && (tid & 63) == 0 && wid < 2)
        for (int i = 0; i < 8; ++i) g_attnp80_dbg[wid * 8 + i] = dbg[i];
#endif
#pragma unroll
    for (int jq = 0; jq < QF; ++jq) {
        // row D of O^T (the ones row of V^T) is the softmax denominator
        const float l = __shfl(oacc[G::E_L][jq][0], (G::LG_L << 4) | l15);
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

#ifndef DM_ATTN80_CC
#define DM_ATTN80_CC 0
#endif
bool attention_pipe80_supports(const AttnParams& p) {
    return p.D == 80 && p.Tk >= 256 && (p.Tk % 128) == 0;
}

hipError_t launch_attention_pipe80(const AttnParams& p, hipStream_t s) {
    if (!attention_pipe80_supports(p)) return hipErrorInvalidValue;
    constexpr int QBLK = 64 * QF;
    dim3 grid(((p.Tq + QBLK - 1) / QBLK) * p.heads * p.B), block(NT);
    static std::atomic<uint64_t> attr_seen{0};      // hipFuncSetAttribute is per DEVICE, not per process
    if (first_use_on_device(attr_seen)) {
        (void)hipFuncSetAttribute((const void*)attn_pipe80_kernel<80, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Geo<80, 0>::LDS);
        (void)hipFuncSetAttribute((const void*)attn_pipe80_kernel<80, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Geo<80, 1>::LDS);
    }
    constexpr size_t lds0 = Geo<80, 0>::LDS, lds1 = Geo<80, 1>::LDS;
    if (option(OPT_ATTN_PIPE) == 3) launch_timed((attn_pipe80_kernel<80, 1>), grid, block, lds1, s, p);   // A/B: rows with constant chunks
    else launch_timed((attn_pipe80_kernel<80, 0>), grid, block, lds0, s, p);
    return hipGetLastError();
}

#ifdef DM_ATTN_TIMING
extern "C" int dm_debug_attn80_timing(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attnp80_dbg), sizeof(long long) * 16) == hipSuccess ? 0 : 1;
}
extern "C" int dm_debug_attn80_span(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attnp80_span), 16) != hipSuccess) return 1;
    if (reset) { unsigned long long z[2] = {~0ull, 0ull}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_attnp80_span), z, 16) != hipSuccess) return 1; }
    return 0;
}
#endif

}  // namespace dm
