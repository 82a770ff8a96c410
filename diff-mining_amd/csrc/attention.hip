// attention.hip — K4/K5: flash-style attention on gfx950 MFMA for the SDv1.5 U-Net's
// `Attention` layers (self: 4096/1024/256/64 tokens; cross: 77 text tokens), head_dim 40/80/160,
// as executed under `unet(...)` at diffmining/typicality/compute.py:100.  No score matrix is
// materialised; softmax is online, in fp32, on the fp32 MFMA accumulators; P is rounded to fp16 for
// the PV product (what flash/xformers kernels do, compute.py:71-72).
//
// Work split: one block = 128 queries of one (sample, head); 4 waves x 32 queries; KV tiles of 64 keys.
//   * K and V tiles go L2/HBM -> LDS with global_load_lds (row-major [key][D], double buffered, one
//     barrier per tile);
//   * the score tile is computed TRANSPOSED, S^T = K · Q^T (K fragment = MFMA A operand), so a lane
//     holds 4 keys x 1 query per fragment.  Those registers are exactly a valid B operand of
//     O^T = V^T · P^T under a permuted k order (the MFMA k index is a dummy: A and B only have to
//     agree), so P never moves between lanes and never touches LDS;
//   * the V^T fragments (k-contiguous) come from the LDS transpose read ds_read_b64_tr_b16: each
//     16-lane group reads a 4-key x 16-d block and lane i receives 4 consecutive keys of column i;
//   * head_dim padding (40 -> 64 for the QK^T k extent, 40 -> 48 PV rows) is done by pointing the
//     out-of-range lanes at a zeroed LDS slot — never in HBM;
//   * lazy rescale: the running max is only advanced (and O, l rescaled) when some row of the wave
//     grew by more than 2^8, so the steady state has no O-wide multiply;
//   * the softmax is VALU bound at head_dim 40 (4-cycle VALU, one exp per score), so two of its
//     terms are moved into the MFMAs' idle padding: the running max rides in two extra k columns of
//     the padded QK^T operand (the MFMA returns sc*q.k - m), and the denominator accumulates in a
//     padded, all-ones row of V^T.  Steady state per score: exp2 + max + convert (D = 40: +11 %).
// Measured alternatives (r01): register staging + transposing ds_write_b16 0.55x, 128-key staged
// tiles 0.95x, 64 queries/wave (occupancy 1) 0.8x, forcing 4 waves/SIMD (spills) 0.45x.
#include "dm_kernels.h"
#include <cstdlib>

namespace dm {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int KT = 64;                // keys per tile
constexpr int NT = 256;               // threads per block
constexpr float RESCALE_THR = 8.0f;   // log2 units

__device__ __attribute__((aligned(256))) unsigned char g_attn_zero[256];
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int D, int QF>
__global__ __launch_bounds__(NT, (D > 80 ? 1 : 2))
void attn_kernel(AttnParams p) {
    constexpr int DP = ((D + 31) / 32) * 32, DV = ((D + 15) / 16) * 16;
    constexpr int KS = DP / 32, EF = DV / 16, NCH = D / 8;
    constexpr int RS = D * 2;                      // LDS row stride (bytes), rows contiguous
    constexpr int TBYTES = KT * RS;               // one K (or V) tile
    constexpr int ZREL = 2 * TBYTES;               // zeroed 64-byte slot at the end of each stage
    constexpr int STAGE = 2 * TBYTES + 64;
    // FOLD : the padded part of the QK^T k extent carries two extra columns, K' = [k, 1, 1] and
    //        Q' = [sc*q, -m_hi, -m_lo], so the MFMA itself delivers sc*(q.k) - m_run (fp32 accumulate;
    //        m split in two fp16 halves) and the steady-state softmax is exp2 + max + convert only.
    // LONES: one padded PV row of V^T is all ones, so the softmax denominator accumulates in the PV
    //        accumulator (row d = D) instead of with VALU adds.
    constexpr bool FOLD = (DP - D >= 8);            // D = 40, 80
    constexpr bool LONES = (DV - D >= 4);           // D = 40
    constexpr int SX = (D / 8) / 4, LGX = (D / 8) % 4;          // fragment (s, lane group) of chunk d = D..D+7
    constexpr int EX = D / 16, VLX = (D % 16) / 4;              // PV fragment / lane sub-group of row d = D
    constexpr int KONES = ZREL + 16, VONES = ZREL + 32;        // constant slots inside the 64-byte area
    constexpr int NI = 2 * NCH * (KT / 64);       // glds instructions per KV tile (K then V)
    constexpr int MI = (NI + 3) / 4;               // per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15;
    const int lg = lane >> 4;
    const int h = blockIdx.y;
    const int b = blockIdx.z;
    const int q0 = blockIdx.x * (64 * QF) + wid * (16 * QF);
    int kvb = p.kv_slot ? p.kv_slot[b] : (p.slot_div > 0 ? b / p.slot_div : b);
    if (p.n_slots > 0) kvb = kvb < 0 ? 0 : (kvb < p.n_slots ? kvb : p.n_slots - 1);      // memory safety: never beyond the registered prompts

    const f16* Qb = p.Q + (size_t)(p.q_mod > 0 ? b % p.q_mod : b) * p.bsq + h * D;
    const f16* Kb = p.K + (size_t)kvb * p.bsk + h * D;
    const f16* Vb = p.V + (size_t)kvb * p.bsv + h * D;
    f16* Ob = p.O + (size_t)b * p.bso + h * D;

    if (tid < 32) {          // per stage: 16 B zeros | 16 B {1,1,0..} (K ones columns) | 8 B {1,0,0,0} (V ones row) | zeros
        const int w = tid & 15;
        unsigned v = 0u;
        if (w == 4) v = 0x3C003C00u;                 // halfs {1, 1}
        if (w == 8) v = 0x00003C00u;                 // halfs {1, 0}
        *reinterpret_cast<unsigned*>(smem + (tid >> 4) * STAGE + ZREL + w * 4) = v;
    }

    // ---- Q fragments ----------------------------------------------------------------------------
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
        }
    }
    const float sc = p.scale * 1.44269504088896340736f;   // scores live in the log2 domain
    if (FOLD) {                                              // Q' = fp16(sc * q)
#pragma unroll
        for (int jq = 0; jq < QF; ++jq)
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int k = 0; k < 8; ++k) qf[jq][s][k] = (f16)((float)qf[jq][s][k] * sc);
    }

    // ---- glds bookkeeping: instruction j = wid + 4*i covers elements idx = (j % NCH ... ) ----------
    // K tile = instructions 0..NI/2-1, V tile = NI/2..NI-1; instruction jj of a tile writes LDS bytes
    // [jj*1024, +1024) = 16-byte chunks idx = jj*64 + lane -> (key = idx / NCH, ch = idx % NCH)
    const f16* gsrc[MI];
    int gkey[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int j = wid + 4 * i;
        const int jj = (j < NI / 2) ? j : j - NI / 2;
        const int idx = jj * 64 + lane;
        const int key = idx / NCH, ch = idx - key * NCH;
        gkey[i] = key;
        gsrc[i] = ((j < NI / 2) ? Kb : Vb) + (size_t)key * ((j < NI / 2) ? p.ldk : p.ldv) + ch * 8;
    }
    const f16* zero = reinterpret_cast<const f16*>(g_attn_zero);
    auto issue = [&](int buf, int k0) __attribute__((always_inline)) {      // L2/HBM -> LDS directly
        char* base = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int j = wid + 4 * i;
            if (j < NI) {
                const bool ok = (k0 + gkey[i] < p.Tk);
                const f16* a = ok ? gsrc[i] : zero;
                __builtin_amdgcn_global_load_lds((gptr_t)a, (lptr_t)(base + j * 1024), 16, 0, 0);
                gsrc[i] += (size_t)KT * ((j < NI / 2) ? p.ldk : p.ldv);
            }
        }
    };
    // ---- LDS read offsets (tile-relative; lanes beyond head_dim read the zero slot) ---------------
    int koff[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s)
        koff[s] = (32 * s + 8 * lg < D) ? (l15 * RS + 64 * s + 16 * lg) : ((FOLD && s == SX && lg == LGX) ? KONES : ZREL);
    int voff[EF];
#pragma unroll
    for (int e = 0; e < EF; ++e)
        voff[e] = (16 * e + 4 * (l15 & 3) < D) ? (TBYTES + (4 * lg + (l15 >> 2)) * RS + 32 * e + 8 * (l15 & 3))
                                                : ((LONES && e == EX && (l15 & 3) == VLX) ? VONES : ZREL);
    // a lane either reads real rows (offset advances with the fragment) or the zero slot (it does not)
    int kstep[KS], vstep[EF];
#pragma unroll
    for (int s = 0; s < KS; ++s) kstep[s] = (koff[s] >= ZREL) ? 0 : 16 * RS;
#pragma unroll
    for (int e = 0; e < EF; ++e) vstep[e] = (voff[e] >= ZREL) ? 0 : 16 * RS;

    floatx4 oacc[EF][QF];
#pragma unroll
    for (int e = 0; e < EF; ++e)
#pragma unroll
        for (int jq = 0; jq < QF; ++jq) oacc[e][jq] = floatx4{0, 0, 0, 0};
    float m_run[QF], l_run[QF];
#pragma unroll
    for (int jq = 0; jq < QF; ++jq) { m_run[jq] = FOLD ? 0.f : -INFINITY; l_run[jq] = 0.f; }

    auto compute = [&](int buf, int k0, bool tail, bool first) __attribute__((always_inline)) {
        const char* kt = smem + buf * STAGE;
        floatx4 sacc[4][QF];
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int jq = 0; jq < QF; ++jq) sacc[f][jq] = floatx4{0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const half8 kf = *reinterpret_cast<const half8*>(kt + koff[s] + f * kstep[s]);
#pragma unroll
                for (int jq = 0; jq < QF; ++jq)
                    sacc[f][jq] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[jq][s], sacc[f][jq], 0, 0, 0);
            }
        }
        half8 pb[QF][2];
#pragma unroll
        for (int jq = 0; jq < QF; ++jq) {
            if (tail) {
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (k0 + 16 * f + 4 * lg + r >= p.Tk) sacc[f][jq][r] = -INFINITY;
            }
            float mx = sacc[0][jq][0];
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = __builtin_fmaxf(mx, sacc[f][jq][r]);
            mx = __builtin_fmaxf(mx, __shfl_xor(mx, 16));
            mx = __builtin_fmaxf(mx, __shfl_xor(mx, 32));
            // lazy rescale: keep the stale running max while no row of the wave grew by more than
            // 2^RESCALE_THR; P is then bounded by 2^RESCALE_THR (exact in fp32, fp16 keeps 11 bits).
            if constexpr (FOLD) {
                // sacc already holds sc*(q.k) - m_run (first tile: m_run = 0)
                float ps = 0.f;
                if (first || __builtin_amdgcn_ballot_w64(mx > RESCALE_THR) != 0ull) {
                    const float delta = first ? mx : __builtin_fmaxf(mx, 0.f);    // the running max only grows
                    const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(-delta);
                    m_run[jq] += delta;
                    if (!LONES) l_run[jq] *= alpha;
#pragma unroll
                    for (int e = 0; e < EF; ++e) oacc[e][jq] *= alpha;
                    if (lg == LGX) {                                          // refresh the -m columns of Q'
                        const f16 mh = (f16)m_run[jq];
                        const f16 ml = (f16)(m_run[jq] - (float)mh);
                        qf[jq][SX][0] = -mh; qf[jq][SX][1] = -ml;
                    }
#pragma unroll
                    for (int f = 0; f < 4; ++f)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float pv = __builtin_amdgcn_exp2f(sacc[f][jq][r] - delta);
                            if (!LONES) ps += pv;
                            pb[jq][f >> 1][(f & 1) * 4 + r] = (f16)pv;
                        }
                } else {                                                      // steady state: exp2 and convert only
#pragma unroll
                    for (int f = 0; f < 4; ++f)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float pv = __builtin_amdgcn_exp2f(sacc[f][jq][r]);
                            if (!LONES) ps += pv;
                            pb[jq][f >> 1][(f & 1) * 4 + r] = (f16)pv;
                        }
                }
                if (!LONES) l_run[jq] += ps;
            } else {
                const float mxs = mx * sc;
                if (__builtin_amdgcn_ballot_w64(mxs > m_run[jq] + RESCALE_THR) != 0ull) {
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
        }
        // PV: the transpose reads are issued through inline asm — the builtin form makes hipcc drain
        // vmcnt(0) (the in-flight LDS-DMA of the NEXT tile) before every LDS transpose read, which
        // serialises the prefetch.  All 4*EF reads are issued, then one lgkmcnt(0), then the MFMAs.
        // (for large head_dim the reads are batched per 32-key half to bound the register footprint)
        constexpr int NB = (EF <= 5) ? 1 : 2;          // batches
#pragma unroll
        for (int bt = 0; bt < NB; ++bt) {
            u32x2 vraw[2 / NB][EF][2];
#pragma unroll
            for (int s2 = 0; s2 < 2 / NB; ++s2)
#pragma unroll
                for (int e = 0; e < EF; ++e)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const int ss = bt * (2 / NB) + s2;
                        const unsigned a = (unsigned)(size_t)(kt + voff[e] + (2 * ss + hh) * vstep[e]);
                        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(vraw[s2][e][hh]) : "v"(a) : "memory");
                    }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s2 = 0; s2 < 2 / NB; ++s2)
#pragma unroll
                for (int e = 0; e < EF; ++e) {
                    const int ss = bt * (2 / NB) + s2;
                    half8 va;
                    __builtin_memcpy(&va, &vraw[s2][e][0], 8);
                    __builtin_memcpy(reinterpret_cast<char*>(&va) + 8, &vraw[s2][e][1], 8);
#pragma unroll
                    for (int jq = 0; jq < QF; ++jq)
                        oacc[e][jq] = __builtin_amdgcn_mfma_f32_16x16x32_f16(va, pb[jq][ss], oacc[e][jq], 0, 0, 0);
                }
        }
    };

    const int ntiles = (p.Tk + KT - 1) / KT;
    auto compute_tile = [&](int cur, int k0) __attribute__((always_inline)) {
        if (k0 + 64 > p.Tk) compute(cur, k0, true, k0 == 0); else compute(cur, k0, false, k0 == 0);
    };
    issue(0, 0);
    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                 // tile t landed for every wave; the other buffer is free
        if (t + 1 < ntiles) issue(cur ^ 1, (t + 1) * KT);
        compute_tile(cur, t * KT);
    }

#pragma unroll
    for (int jq = 0; jq < QF; ++jq) {
        float l;
        if constexpr (LONES) {
            l = __shfl(oacc[EX][jq][0], (VLX << 4) | l15);      // row d = D of O^T lives in lane group VLX, r = 0
        } else {
            l = l_run[jq];
            l += __shfl_xor(l, 16);
            l += __shfl_xor(l, 32);
        }
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

template <int D, int QF>
hipError_t launch_t(const AttnParams& p, hipStream_t s) {
    constexpr int QBLK = 64 * QF;
    dim3 grid((p.Tq + QBLK - 1) / QBLK, p.heads, p.B), block(NT);
    const size_t lds = 2 * (2 * (size_t)KT * D * 2 + 64);
    static std::atomic<uint64_t> attr_seen{0};      // hipFuncSetAttribute is per DEVICE, not per process
    if (first_use_on_device(attr_seen)) {
        (void)hipFuncSetAttribute((const void*)attn_kernel<D, QF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    launch_timed((attn_kernel<D, QF>), grid, block, lds, s, p);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_attention_pipe(const AttnParams& p, hipStream_t s);    // attention_pipe.hip
bool attention_pipe_supports(const AttnParams& p);
hipError_t launch_attention_pipe80(const AttnParams& p, hipStream_t s);  // attention_pipe80.hip
bool attention_pipe80_supports(const AttnParams& p);
hipError_t launch_attention_cross(const AttnParams& p, hipStream_t s);   // attention_cross.hip
bool attention_cross_supports(const AttnParams& p);
hipError_t launch_attention_d160(const AttnParams& p, hipStream_t s);    // attention_d160.hip (r06: head_dim 160, eight waves per block)
bool attention_d160_supports(const AttnParams& p);
hipError_t launch_attention_d160_cross(const AttnParams& p, hipStream_t s);   // the 77-key cross-attention at head_dim 160 (one pass)
bool attention_d160_cross_supports(const AttnParams& p);
hipError_t launch_attention_qk32(const AttnParams& p, hipStream_t s);    // attention_qk32.hip (r06: scores on 32x32x16 MFMAs)
bool attention_qk32_supports(const AttnParams& p);
hipError_t launch_attention_pp(const AttnParams& p, int variant, hipStream_t s);   // attention_pp.hip
bool attention_pp_supports(const AttnParams& p);

hipError_t launch_attention(const AttnParams& p, hipStream_t s) {
    if (p.Tq <= 0 || p.Tk <= 0 || p.B <= 0) return hipErrorInvalidValue;
    // long head_dim-40 / 80 self-attention: software-pipelined variants (DM_ATTN_PIPE=0 disables)
    if (option(OPT_ATTN_CROSS) && attention_cross_supports(p)) return launch_attention_cross(p, s);   // 77-key cross-attention
    if (option(OPT_ATTN_CROSS) && attention_d160_cross_supports(p)) return launch_attention_d160_cross(p, s);      // ... at head_dim 160 (r06)
    const int pipe = option(OPT_ATTN_PIPE);
    if (p.q_mod > 0) {                       // only the generic and the 77-key kernels index Q modulo (cross-attention; handled above / below)
        switch (p.D) { case 40: return launch_t<40, 2>(p, s); case 80: return launch_t<80, 2>(p, s); case 160: return launch_t<160, 2>(p, s); default: return hipErrorInvalidValue; }
    }
    // the anti-phase kernel everywhere it applies (A/B): 10 / 12 = three wave sets without / with static priorities; 9 = the r04 dispatch.
    // set_option() admits no other value (a stray value used to select timing-only ablation instantiations)
    if ((pipe == 10 || pipe == 12) && attention_pp_supports(p)) return launch_attention_pp(p, pipe, s);
    if (pipe == 5 && attention_qk32_supports(p)) return launch_attention_qk32(p, s);      // r06: QK^T on 32x32x16 wherever it applies (A/B)
#ifdef DM_ATTN_PP_ABLATE
    if (pipe >= 4 && pipe != 9 && attention_pp_supports(p)) return launch_attention_pp(p, pipe, s);        // debug library only
#endif
    // default: the three-set anti-phase kernel where it measured faster (>= 8192 keys: the 128x128 level of a 1024-pixel image, -4...6 %
    // per launch; bit-identical to attn_pipe_kernel) and its 384-query blocks waste < 2 % of their rows; attn_pipe = 2 / 3 keep the r04 kernels
    if (pipe == 1 && p.Tk >= 8192 && attention_pp_supports(p) && (long long)((p.Tq + 383) / 384) * 384 * 50 <= 51LL * p.Tq)
        return launch_attention_pp(p, 12, s);
    // r06: head_dim 40 on the kernel whose scores run on 32x32x16 MFMAs (attention_qk32.hip: -1.3 % per launch, -0.16 ms per step ABBA,
    // the same distance to fp32 SDPA; profiles/r06_ab_attn_qk32.txt); attn_pipe = 9 / 2 keep attn_pipe_kernel (A/B)
    if (pipe == 1 && attention_qk32_supports(p)) return launch_attention_qk32(p, s);
    if (pipe && attention_pipe_supports(p)) return launch_attention_pipe(p, s);
    if ((pipe == 1 || pipe >= 3) && attention_pipe80_supports(p)) return launch_attention_pipe80(p, s);     // attn_pipe = 2: head_dim 40 only (A/B)
    // head_dim 160 (r06): eight waves sharing one (sample, head)'s K / V, <= 256 registers; bit-identical to the generic kernel.  attn_pipe = 9 / 2
    // keep the generic kernel (A/B)
    if ((pipe == 1 || pipe == 5) && attention_d160_supports(p)) return launch_attention_d160(p, s);
    switch (p.D) {
        case 40: return launch_t<40, 2>(p, s);
        case 80: return launch_t<80, 2>(p, s);
        case 160: return launch_t<160, 2>(p, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace dm
