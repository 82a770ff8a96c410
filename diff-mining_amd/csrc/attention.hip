// attention.hip — K4/K5: flash-style attention on gfx950 MFMA for the SDv1.5 U-Net's
// `Attention` layers (self: 4096/1024/256/64 tokens; cross: 77 text tokens), head_dim 40/80/160,
// as executed under `unet(...)` at diffmining/typicality/compute.py:100.  No score matrix is
// materialised; softmax is online, in fp32, on the fp32 MFMA accumulators; P is rounded to fp16 for
// the PV product (what flash/xformers kernels do, compute.py:71-72).
//
// Work split: one block = 128 queries of one (sample, head); 4 waves x 32 queries.  KV tiles of 64
// keys are staged global -> registers -> LDS (double buffered, one barrier per tile): K row-major
// (padded rows), V transposed ([d][key]) so both MFMA operands are k-contiguous ds_read_b128/b64.
//
// Trick: the score tile is computed TRANSPOSED, S^T = K · Q^T (K fragment = MFMA A operand), so a
// lane holds 4 keys x 1 query per fragment.  Those registers are exactly a valid B operand of the
// PV product O^T = V^T · P^T under a permuted k order (the MFMA k index is a dummy: A and B only
// have to agree), so P never moves between lanes and never touches LDS.  head_dim is zero padded
// in LDS only (40 -> 64 for QK^T k, 40 -> 48 for the PV output rows), never in HBM.
#include "dm_kernels.h"
#include <cstdlib>

namespace dm {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int QB = 128;    // queries per block
constexpr int KT = 64;     // keys per tile
#ifndef DM_ATTN40_OCC
#define DM_ATTN40_OCC 2
#endif
constexpr float RESCALE_THR = 8.0f;   // log2 units (attention v2 lazy rescale)
constexpr int NT = 256;

template <int D>
struct Cfg {
    static constexpr int DP = ((D + 31) / 32) * 32;     // k extent of QK^T (zero padded)
    static constexpr int DV = ((D + 15) / 16) * 16;     // PV output rows (zero padded)
    static constexpr int KS = DP / 32;
    static constexpr int EF = DV / 16;
    static constexpr int NCH = D / 8;                    // 16-byte chunks per row in HBM
    static constexpr int KSTR = DP * 2 + 16;             // K tile row stride (bytes), padded
    static constexpr int VSTR = KT * 2 + 16;             // V^T tile row stride (bytes), padded
    static constexpr int KBYTES = KT * KSTR;
    static constexpr int VBYTES = DV * VSTR;
    static constexpr int STAGE = KBYTES + VBYTES;
    static constexpr int LD_IT = (KT * NCH + NT - 1) / NT;
};

template <int D>
__global__ __launch_bounds__(NT)
void attn_kernel(AttnParams p) {
    using C = Cfg<D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int l15 = lane & 15;
    const int lg = lane >> 4;
    const int h = blockIdx.y;
    const int b = blockIdx.z;
    const int q0 = blockIdx.x * QB + wid * 32;
    const int kvb = p.kv_slot ? p.kv_slot[b] : b;

    const f16* Qb = p.Q + (size_t)b * p.bsq + h * D;
    const f16* Kb = p.K + (size_t)kvb * p.bsk + h * D;
    const f16* Vb = p.V + (size_t)kvb * p.bsv + h * D;
    f16* Ob = p.O + (size_t)b * p.bso + h * D;

    // ---- zero the LDS padding once (columns D..DP of K rows, rows D..DV of V^T) -----------------
    for (int i = tid * 16; i < 2 * C::STAGE; i += NT * 16)
        *reinterpret_cast<u32x4*>(smem + i) = u32x4{0u, 0u, 0u, 0u};

    // ---- Q fragments (B operand): lane holds Q[q = 16 jq + l15][d = 32 s + 8 lg .. +8] ---------
    half8 qf[2][C::KS];
#pragma unroll
    for (int jq = 0; jq < 2; ++jq) {
        int q = q0 + 16 * jq + l15;
        q = q < p.Tq ? q : p.Tq - 1;
#pragma unroll
        for (int s = 0; s < C::KS; ++s) {
            const int d = 32 * s + 8 * lg;
            if (d < D)
                qf[jq][s] = *reinterpret_cast<const half8*>(Qb + (size_t)q * p.ldq + d);
            else
                qf[jq][s] = half8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }

    floatx4 oacc[C::EF][2];
#pragma unroll
    for (int e = 0; e < C::EF; ++e) { oacc[e][0] = floatx4{0, 0, 0, 0}; oacc[e][1] = floatx4{0, 0, 0, 0}; }
    float m_run[2] = {-INFINITY, -INFINITY};
    float l_run[2] = {0.f, 0.f};
    const float sc = p.scale * 1.44269504088896340736f;   // scores live in the log2 domain

    // ---- KV tile staging ------------------------------------------------------------------------
    u32x4 kr[C::LD_IT], vr[C::LD_IT];
    auto load_regs = [&](int k0) {
#pragma unroll
        for (int it = 0; it < C::LD_IT; ++it) {
            const int idx = tid + it * NT;
            const int key = idx / C::NCH, ch = idx - key * C::NCH;
            if (idx < KT * C::NCH && k0 + key < p.Tk) {
                kr[it] = *reinterpret_cast<const u32x4*>(Kb + (size_t)(k0 + key) * p.ldk + ch * 8);
                vr[it] = *reinterpret_cast<const u32x4*>(Vb + (size_t)(k0 + key) * p.ldv + ch * 8);
            } else {
                kr[it] = u32x4{0u, 0u, 0u, 0u};
                vr[it] = u32x4{0u, 0u, 0u, 0u};
            }
        }
    };
    auto store_lds = [&](int buf) {
        char* kt = smem + buf * C::STAGE;
        char* vt = kt + C::KBYTES;
#pragma unroll
        for (int it = 0; it < C::LD_IT; ++it) {
            const int idx = tid + it * NT;
            const int key = idx / C::NCH, ch = idx - key * C::NCH;
            if (idx < KT * C::NCH) {
                *reinterpret_cast<u32x4*>(kt + key * C::KSTR + ch * 16) = kr[it];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj)
                    *reinterpret_cast<unsigned short*>(vt + (ch * 8 + jj) * C::VSTR + key * 2) =
                        (unsigned short)((vr[it][jj >> 1] >> ((jj & 1) * 16)) & 0xFFFFu);
            }
        }
    };

    auto compute = [&](int buf, int k0) {
        const char* kt = smem + buf * C::STAGE;
        const char* vt = kt + C::KBYTES;
        // S^T = K Q^T : sacc[f][jq][r] = score(key 16f + 4lg + r, query 16jq + l15)
        floatx4 sacc[4][2];
#pragma unroll
        for (int f = 0; f < 4; ++f) { sacc[f][0] = floatx4{0, 0, 0, 0}; sacc[f][1] = floatx4{0, 0, 0, 0}; }
#pragma unroll
        for (int s = 0; s < C::KS; ++s) {
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const half8 kf = *reinterpret_cast<const half8*>(kt + (16 * f + l15) * C::KSTR + (32 * s + 8 * lg) * 2);
                sacc[f][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[0][s], sacc[f][0], 0, 0, 0);
                sacc[f][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[1][s], sacc[f][1], 0, 0, 0);
            }
        }
        const bool tail = (k0 + KT > p.Tk);
        half8 pb[2][2];
#pragma unroll
        for (int jq = 0; jq < 2; ++jq) {
            float mx = -INFINITY;
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float sv = sacc[f][jq][r] * sc;
                    if (tail && (k0 + 16 * f + 4 * lg + r >= p.Tk)) sv = -INFINITY;
                    sacc[f][jq][r] = sv;
                    mx = fmaxf(mx, sv);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run[jq], mx);
            const float alpha = exp2f(m_run[jq] - m_new);
            m_run[jq] = m_new;
            float ps = 0.f;
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = exp2f(sacc[f][jq][r] - m_new);
                    ps += pv;
                    pb[jq][f >> 1][(f & 1) * 4 + r] = (f16)pv;
                }
            l_run[jq] = l_run[jq] * alpha + ps;
#pragma unroll
            for (int e = 0; e < C::EF; ++e) oacc[e][jq] *= alpha;
        }
        // O^T += V^T P^T with k slot (lg, j) <-> key 32 s2 + 16 (j>>2) + 4 lg + (j&3)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
            for (int e = 0; e < C::EF; ++e) {
                const char* vrow = vt + (16 * e + l15) * C::VSTR + (32 * s2 + 4 * lg) * 2;
                const half4 v0 = *reinterpret_cast<const half4*>(vrow);
                const half4 v1 = *reinterpret_cast<const half4*>(vrow + 32);
                const half8 va = half8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                oacc[e][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(va, pb[0][s2], oacc[e][0], 0, 0, 0);
                oacc[e][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(va, pb[1][s2], oacc[e][1], 0, 0, 0);
            }
        }
    };

    const int ntiles = (p.Tk + KT - 1) / KT;
    load_regs(0);
    __syncthreads();            // padding zero-fill complete before the first tile write
    store_lds(0);
    __syncthreads();
    for (int t = 0; t < ntiles - 1; ++t) {
        const int cur = t & 1;
        load_regs((t + 1) * KT);
        compute(cur, t * KT);
        store_lds(cur ^ 1);
        __syncthreads();
    }
    compute((ntiles - 1) & 1, (ntiles - 1) * KT);

    // ---- finalize: O[q][16e + 4lg + r] = oacc / l ------------------------------------------------
#pragma unroll
    for (int jq = 0; jq < 2; ++jq) {
        float l = l_run[jq];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const float inv = 1.0f / l;
        const int q = q0 + 16 * jq + l15;
        if (q >= p.Tq) continue;
#pragma unroll
        for (int e = 0; e < C::EF; ++e) {
            const int d = 16 * e + 4 * lg;
            if (d < D) {
                const half4 o = half4{(f16)(oacc[e][jq][0] * inv), (f16)(oacc[e][jq][1] * inv),
                                      (f16)(oacc[e][jq][2] * inv), (f16)(oacc[e][jq][3] * inv)};
                *reinterpret_cast<half4*>(Ob + (size_t)q * p.ldo + d) = o;
            }
        }
    }
}

template <int D>
hipError_t launch_t(const AttnParams& p, hipStream_t s) {
    using C = Cfg<D>;
    dim3 grid((p.Tq + QB - 1) / QB, p.heads, p.B), block(NT);
    const size_t lds = 2 * C::STAGE;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)attn_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL(attn_kernel<D>, grid, block, lds, s, p);
    return hipGetLastError();
}


// ---------------------------------------------------------------------------------------------------
// v2: K and V tiles go HBM/L2 -> LDS directly with global_load_lds (row-major [key][D], no VGPR
// staging, no transposing ds_write); the PV A operand (V^T, k-contiguous) is produced by the LDS
// transpose read ds_read_b64_tr_b16 (each 16-lane group reads a 4-key x 16-d block and receives it
// transposed: lane i gets 4 consecutive keys of column d = i).  head_dim padding (40->64 for the QK^T
// k extent, 40->48 for PV rows) is realised by pointing the out-of-range lanes at a zeroed LDS slot.
// QF = 16-query fragments per wave (queries per block = 64*QF).
// ---------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(256))) unsigned char g_attn_zero[256];
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

template <int D, int QF, int KTL, bool GLDS>
__global__ __launch_bounds__(NT, (D > 80 ? 1 : (D == 40 ? DM_ATTN40_OCC : 2)))
void attn2_kernel(AttnParams p) {
    constexpr int DP = ((D + 31) / 32) * 32, DV = ((D + 15) / 16) * 16;
    constexpr int KS = DP / 32, EF = DV / 16, NCH = D / 8;
    constexpr int RS = D * 2;                      // LDS row stride (bytes), rows contiguous
    constexpr int TBYTES = KTL * RS;               // one K (or V) tile
    constexpr int ZREL = 2 * TBYTES;               // zeroed 64-byte slot at the end of each stage
    constexpr int STAGE = 2 * TBYTES + 64;
    constexpr int NI = 2 * NCH * (KTL / 64);       // glds instructions per KV tile (K then V)
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
    const int kvb = p.kv_slot ? p.kv_slot[b] : b;

    const f16* Qb = p.Q + (size_t)b * p.bsq + h * D;
    const f16* Kb = p.K + (size_t)kvb * p.bsk + h * D;
    const f16* Vb = p.V + (size_t)kvb * p.bsv + h * D;
    f16* Ob = p.O + (size_t)b * p.bso + h * D;

    if (tid < 32) *reinterpret_cast<unsigned*>(smem + (tid >> 4) * STAGE + ZREL + (tid & 15) * 4) = 0u;

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
    u32x4 stg[MI];                                 // register staging (GLDS == false)
    auto issue = [&](int buf, int k0) __attribute__((always_inline)) {            // GLDS: HBM/L2 -> LDS directly; else -> registers
        char* base = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int j = wid + 4 * i;
            if (j < NI) {
                const bool ok = (k0 + gkey[i] < p.Tk);
                if (GLDS) {
                    const f16* a = ok ? gsrc[i] : zero;
                    __builtin_amdgcn_global_load_lds((gptr_t)a, (lptr_t)(base + j * 1024), 16, 0, 0);
                } else {
                    stg[i] = ok ? *reinterpret_cast<const u32x4*>(gsrc[i]) : u32x4{0u, 0u, 0u, 0u};
                }
                gsrc[i] += (size_t)KTL * ((j < NI / 2) ? p.ldk : p.ldv);
            }
        }
    };
    auto commit = [&](int buf) __attribute__((always_inline)) {                   // registers -> LDS (same lane-linear image as glds)
        char* base = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int j = wid + 4 * i;
            if (j < NI) *reinterpret_cast<u32x4*>(base + j * 1024 + lane * 16) = stg[i];
        }
    };

    // ---- LDS read offsets (tile-relative; lanes beyond head_dim read the zero slot) ---------------
    int koff[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) koff[s] = (32 * s + 8 * lg < D) ? (l15 * RS + 64 * s + 16 * lg) : ZREL;
    int voff[EF];
#pragma unroll
    for (int e = 0; e < EF; ++e)
        voff[e] = (16 * e + 4 * (l15 & 3) < D) ? (TBYTES + (4 * lg + (l15 >> 2)) * RS + 32 * e + 8 * (l15 & 3)) : ZREL;
    // a lane either reads real rows (offset advances with the fragment) or the zero slot (it does not)
    // NOTE: sub-tile 1 adds 64*RS to every address, so the zero slot is 64 bytes at ZREL and another
    // 64 bytes at ZREL + 64*RS is needed; instead lanes that read zeros subtract the sub offset again.
    int kstep[KS], vstep[EF];
#pragma unroll
    for (int s = 0; s < KS; ++s) kstep[s] = (koff[s] == ZREL) ? 0 : 16 * RS;
#pragma unroll
    for (int e = 0; e < EF; ++e) vstep[e] = (voff[e] == ZREL) ? 0 : 16 * RS;

    floatx4 oacc[EF][QF];
#pragma unroll
    for (int e = 0; e < EF; ++e)
#pragma unroll
        for (int jq = 0; jq < QF; ++jq) oacc[e][jq] = floatx4{0, 0, 0, 0};
    float m_run[QF], l_run[QF];
#pragma unroll
    for (int jq = 0; jq < QF; ++jq) { m_run[jq] = -INFINITY; l_run[jq] = 0.f; }
    const float sc = p.scale * 1.44269504088896340736f;

    auto compute = [&](int buf, int sub, int k0, bool tail) __attribute__((always_inline)) {
        const char* kt = smem + buf * STAGE + sub * (64 * RS);
        floatx4 sacc[4][QF];
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int jq = 0; jq < QF; ++jq) sacc[f][jq] = floatx4{0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const half8 kf = *reinterpret_cast<const half8*>(kt + koff[s] + f * kstep[s] - (kstep[s] ? 0 : sub * (64 * RS)));
#pragma unroll
                for (int jq = 0; jq < QF; ++jq)
                    sacc[f][jq] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[jq][s], sacc[f][jq], 0, 0, 0);
            }
        }
        half8 pb[QF][2];
#ifdef DM_EXP_NOSOFTMAX
#pragma unroll
        for (int jq = 0; jq < QF; ++jq)
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) pb[jq][f >> 1][(f & 1) * 4 + r] = (f16)sacc[f][jq][r];
#else
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
#ifdef DM_EXP_NOEXP
                    const float pv = __builtin_fmaf(sacc[f][jq][r], sc, -m_use) * 1e-3f;
#else
                    const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[f][jq][r], sc, -m_use));
#endif
                    ps += pv;
                    pb[jq][f >> 1][(f & 1) * 4 + r] = (f16)pv;
                }
            l_run[jq] += ps;
        }
#endif
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
                        const unsigned a = (unsigned)(size_t)(kt + voff[e] + (2 * ss + hh) * vstep[e] - (vstep[e] ? 0 : sub * (64 * RS)));
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
#ifdef DM_EXP_NOPV
                    asm volatile("" :: "v"(va));
#else
#pragma unroll
                    for (int jq = 0; jq < QF; ++jq)
                        oacc[e][jq] = __builtin_amdgcn_mfma_f32_16x16x32_f16(va, pb[jq][ss], oacc[e][jq], 0, 0, 0);
#endif
                }
        }
#ifdef DM_EXP_NOPV
#pragma unroll
        for (int jq = 0; jq < QF; ++jq) { asm volatile("" :: "v"(pb[jq][0]), "v"(pb[jq][1])); }
#endif
    };

    const int ntiles = (p.Tk + KTL - 1) / KTL;
    auto compute_tile = [&](int cur, int k0) __attribute__((always_inline)) {
        if (k0 + 64 > p.Tk) compute(cur, 0, k0, true); else compute(cur, 0, k0, false);
        if (KTL == 128 && k0 + 64 < p.Tk) {
            if (k0 + 128 > p.Tk) compute(cur, 1, k0 + 64, true); else compute(cur, 1, k0 + 64, false);
        }
    };
    if (GLDS) {
        issue(0, 0);
        for (int t = 0; t < ntiles; ++t) {
            const int cur = t & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t + 1 < ntiles) issue(cur ^ 1, (t + 1) * KTL);
            compute_tile(cur, t * KTL);
        }
    } else {
        issue(0, 0);
        __syncthreads();            // zero slots written
        commit(0);
        __syncthreads();
        for (int t = 0; t < ntiles; ++t) {
            const int cur = t & 1;
            if (t + 1 < ntiles) issue(cur ^ 1, (t + 1) * KTL);     // global loads in flight under compute
            compute_tile(cur, t * KTL);
            if (t + 1 < ntiles) commit(cur ^ 1);
            __syncthreads();
        }
    }

#pragma unroll
    for (int jq = 0; jq < QF; ++jq) {
        float l = l_run[jq];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
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

template <int D, int QF, int KTL, bool GLDS>
hipError_t launch2_t(const AttnParams& p, hipStream_t s) {
    constexpr int QBLK = 64 * QF;
    dim3 grid((p.Tq + QBLK - 1) / QBLK, p.heads, p.B), block(NT);
    const size_t lds = 2 * (2 * (size_t)KTL * D * 2 + 64);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)attn2_kernel<D, QF, KTL, GLDS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((attn2_kernel<D, QF, KTL, GLDS>), grid, block, lds, s, p);
    return hipGetLastError();
}

static int attn_variant() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("DM_ATTN"); v = e ? atoi(e) : 1; }
    return v;
}

}  // namespace

hipError_t launch_attention(const AttnParams& p, hipStream_t s) {
    if (p.Tq <= 0 || p.Tk <= 0 || p.B <= 0) return hipErrorInvalidValue;
    const int var = attn_variant();
    if (var >= 1) {
        switch (p.D * 10 + var) {
            case 401: return launch2_t<40, 2, 64, true>(p, s);
            case 403: return launch2_t<40, 2, 64, false>(p, s);
            case 404: return launch2_t<40, 2, 128, false>(p, s);
            case 405: return launch2_t<40, 2, 128, true>(p, s);
            default: break;
        }
        switch (p.D) {
            case 40: return launch2_t<40, 2, 64, true>(p, s);
            case 80: return launch2_t<80, 2, 64, true>(p, s);
            case 160: return launch2_t<160, 2, 64, true>(p, s);
            default: return hipErrorInvalidValue;
        }
    }
    switch (p.D) {
        case 40: return launch_t<40>(p, s);
        case 80: return launch_t<80>(p, s);
        case 160: return launch_t<160>(p, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace dm
