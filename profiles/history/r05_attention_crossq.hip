// attention_crossq.hip — LayerNorm2 -> attn2.to_q -> the 77-key cross-attention of a transformer block in ONE kernel (r05,
// head_dim 40 = the 320-channel level), reached from `unet(...)`, diffmining/typicality/compute.py:100
// (diffusers BasicTransformerBlock: norm2 -> attn2).
//
// The chain ran as two launches that are both bound by moving the token matrix: the LayerNorm-folded GEMM writes q [tokens x C]
// (419 MB at the 64x64 level of the bench batch), attention_cross.hip reads it back.  Here a block owns a (sample, head) pair (or a
// slice of its query blocks) as in attention_cross.hip and keeps, next to K and V of its (prompt, head), the head's 40 rows of the
// folded weight W' = Wq diag(gamma) resident in LDS; per 32 queries a wave
//   * reads its token rows x [32 x C] straight into MFMA B operands (lane (query c, g): 32 contiguous bytes per pair of k steps),
//   * forms q^T [d x query] = W'_h x^T on the matrix cores (A = weight rows from LDS; a ones row behind the 40 real rows gives
//     sum(x) per token for free; sum(x^2) by v_dot2), applies the folded LayerNorm exactly as the GEMM's epilogue does
//     (q = fma(rstd, acc, fma(-rstd mean, s_d, t_d)), s / t = pack_ln_fold's vectors) and rounds to fp16 — the reference's rounding
//     point (to_q's output under autocast),
//   * and uses the accumulators AS THEY LIE as the B operand of S^T = K Q^T: a lane holds q[d = 16 e + 4 g + r] of its query, the
//     MFMA wants k = 32 s + 8 g + 0..7 — k is a contraction index, so K's columns are stored in LDS in the permuted order
//     (slot (s, g, p): p < 4 -> d = 16 (2 s) + 4 g + p, p >= 4 -> d = 16 + 4 g + (p - 4) for s = 0; zero beyond head_dim) and q never
//     leaves its lane.  Softmax, P V and the staged 16-byte output stores are attention_cross.hip's.
// The q tensor is never written or read; the token matrix is read once per head from the XCD's L2 (the eight heads of a token
// range are co-scheduled on one XCD, as in attention_cross.hip).  Numerically equivalent to the two-launch chain, not bit-identical
// (the row statistics are summed in another order); a sample's bits do not depend on its batch.
#include "dm_kernels.h"

#include <type_traits>
#include <utility>

namespace dm {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2x __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int NT = 256;               // threads per block (4 waves x 32 queries)
constexpr int QF = 2;                 // 16-query fragments per wave
constexpr int D = 40, C = 320;        // head_dim, channels (8 heads)
constexpr int KROWS = 80, VROWS = 96, NKB = KROWS / 16;
constexpr int KSTEPS = C / 32;        // k steps of the projection
constexpr int WROWS = 48;             // weight rows held: 40 real, row 40 = ones (sum of the token's channels), 41..47 zero
constexpr int WRS = C * 2 + 32;       // 672 B: stride / 4 = 8 * odd -> conflict-free ds_read_b128 over 16 rows
constexpr int KRS = 160;              // permuted K rows: 64 k slots (128 B) + 32 B; stride / 4 = 40 = 8 * 5
constexpr int VRS = 96;               // V rows: 40 real halfs + the constant chunk {1, 0, ..} (ones row of V^T)
constexpr int EF = 3;
constexpr int WOFF = 0, WBYTES = WROWS * WRS;
constexpr int KOFF = WBYTES, KBYTES = KROWS * KRS;
constexpr int VOFF = KOFF + KBYTES, VBYTES = VROWS * VRS + 32;
constexpr int OST_ROW = D * 2 + 16, OST_WAVE = 32 * OST_ROW;
constexpr int OOFF = VOFF + VBYTES;
constexpr int LDS_BYTES = OOFF + 4 * OST_WAVE;
static_assert((WRS / 4) % 16 == 8 && (KRS / 4) % 16 == 8 && (VRS / 4) % 16 == 8, "row strides must spread eight rows over the 64 banks");

template <int OFF>
__device__ __forceinline__ void tr_read(u32x2& out, unsigned base) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(out) : "v"(base), "n"(OFF) : "memory");
}
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

__global__ __launch_bounds__(NT, 2)
void attn_crossq_kernel(AttnParams p, CrossQParams fq, int nsplit) {
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15;
    const int lg = lane >> 4;
    // block order as in attention_cross.hip: the heads of one (sample, query slice) share an XCD at the same time
    const int xx = blockIdx.x & 7, tt = blockIdx.x >> 3;
    const int h = tt % p.heads;
    const int unit = (tt / p.heads) * 8 + xx;                // (sample, slice)
    if (unit >= p.B * nsplit) return;
    const int b = unit / nsplit, part = unit - b * nsplit;
    int kvb = p.kv_slot ? p.kv_slot[b] : (p.slot_div > 0 ? b / p.slot_div : b);
    if (p.n_slots > 0) kvb = kvb < 0 ? 0 : (kvb < p.n_slots ? kvb : p.n_slots - 1);
    const int nqb = (p.Tq + 64 * QF - 1) / (64 * QF);
    const int qb0 = (int)((long long)part * nqb / nsplit), qb1 = (int)((long long)(part + 1) * nqb / nsplit);

    const f16* Xb = fq.X + (size_t)(p.q_mod > 0 ? b % p.q_mod : b) * fq.bsx;
    const f16* Kb = p.K + (size_t)kvb * p.bsk + h * D;
    const f16* Vb = p.V + (size_t)kvb * p.bsv + h * D;
    f16* Ob = p.O + (size_t)b * p.bso + h * D;

    // ---- token rows of a query block: B operands of the projection (16 bytes per lane and k step) ----
    auto load_x = [&](half8 (&x)[QF][KSTEPS], int qb) __attribute__((always_inline)) {
#pragma unroll
        for (int jq = 0; jq < QF; ++jq) {
            int qi = qb * (64 * QF) + wid * (16 * QF) + 16 * jq + l15;
            qi = qi < p.Tq ? qi : p.Tq - 1;
            // k steps 2 j and 2 j + 1 take channels 64 j + 16 lg + (0..7 | 8..15): a lane reads 32 contiguous bytes, four lanes one whole
            // 128-byte line of the token row (k is a contraction index: the weight fragments below are read in the same order)
            const f16* row = Xb + (size_t)qi * fq.ldx + 16 * lg;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) x[jq][ks] = *reinterpret_cast<const half8*>(row + 64 * (ks >> 1) + 8 * (ks & 1));
        }
    };
    half8 xa[QF][KSTEPS];
    if (qb0 < qb1) load_x(xa, qb0);

    // ---- resident operands: W'_h (48 rows), K (permuted columns, 80 rows), V (96 rows + constant chunk) ----
    {
        constexpr int WC = WRS / 16;                         // 42 chunks per weight row (40 real)
        const f16* Wh = fq.Wq + (size_t)(h * D) * C;
        for (int c = tid; c < WROWS * WC; c += NT) {
            const int row = c / WC, ch = c - row * WC;
            u32x4 val = u32x4{0u, 0u, 0u, 0u};
            if (ch < C / 8) {
                if (row < D) val = *reinterpret_cast<const u32x4*>(Wh + (size_t)row * C + ch * 8);
                else if (row == D) val = u32x4{0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u};
            }
            *reinterpret_cast<u32x4*>(smem + WOFF + row * WRS + ch * 16) = val;
        }
        constexpr int KC = KRS / 16;                         // 10 chunks per K row: 8 k-slot chunks + 2 of padding
        for (int c = tid; c < KROWS * KC; c += NT) {
            const int row = c / KC, ch = c - row * KC;
            u32x4 val = u32x4{0u, 0u, 0u, 0u};
            if (row < p.Tk && ch < 8) {
                const int s = ch >> 2, g = ch & 3;           // k slots 8 ch .. 8 ch + 7 = (s, g, p = 0..7)
                const f16* kr = Kb + (size_t)row * p.ldk;
                if (s == 0) {
                    const u32x2 lo = *reinterpret_cast<const u32x2*>(kr + 4 * g), hi = *reinterpret_cast<const u32x2*>(kr + 16 + 4 * g);
                    val = u32x4{lo[0], lo[1], hi[0], hi[1]};
                } else if (g < 2) {
                    const u32x2 lo = *reinterpret_cast<const u32x2*>(kr + 32 + 4 * g);
                    val = u32x4{lo[0], lo[1], 0u, 0u};
                }
            }
            *reinterpret_cast<u32x4*>(smem + KOFF + row * KRS + ch * 16) = val;
        }
        constexpr int VC = VRS / 16, NV = VBYTES / 16;
        for (int c = tid; c < NV; c += NT) {
            const int row = c / VC, ch = c - row * VC;
            u32x4 val = u32x4{0u, 0u, 0u, 0u};
            if (row < p.Tk && ch < D / 8) val = *reinterpret_cast<const u32x4*>(Vb + (size_t)row * p.ldv + ch * 8);
            else if (ch == D / 8 && row < VROWS) val = u32x4{0x00003C00u, 0u, 0u, 0u};
            *reinterpret_cast<u32x4*>(smem + VOFF + c * 16) = val;
        }
    }
    // the folded LayerNorm's per-channel vectors of this lane's channels d = 16 e + 4 g + r
    float sd[EF][4], td[EF][4];
#pragma unroll
    for (int e = 0; e < EF; ++e)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int d = 16 * e + 4 * lg + r;
            sd[e][r] = d < D ? fq.ln_s[h * D + d] : 0.f;
            td[e][r] = d < D ? fq.ln_t[h * D + d] : 0.f;
        }
    __syncthreads();

    const char* wbase = smem + WOFF + l15 * WRS + 32 * lg;                                              // weight fragment reads (channel order of load_x)
    const char* kbase = smem + KOFF + l15 * KRS + 16 * lg;                                              // K fragment reads
    const unsigned vbase = (unsigned)(size_t)(smem + VOFF + (4 * lg + (l15 >> 2)) * VRS + 8 * (l15 & 3));   // V^T transpose reads
    const float sc = p.scale * 1.44269504088896340736f;
    const float inv_c = 1.0f / (float)C;

    for (int qb = qb0; qb < qb1; ++qb) {
        // ---- q^T = W'_h x^T, sum(x) in row 40, sum(x^2) beside it ----
        floatx4 qacc[EF][QF];
        float ssq[QF];
#pragma unroll
        for (int jq = 0; jq < QF; ++jq) {
            ssq[jq] = 0.f;
#pragma unroll
            for (int e = 0; e < EF; ++e) qacc[e][jq] = floatx4{0, 0, 0, 0};
        }
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            half8 wf[EF];
#pragma unroll
            for (int e = 0; e < EF; ++e) wf[e] = *reinterpret_cast<const half8*>(wbase + e * 16 * WRS + 128 * (ks >> 1) + 16 * (ks & 1));
#pragma unroll
            for (int jq = 0; jq < QF; ++jq) {
#pragma unroll
                for (int e = 0; e < EF; ++e) qacc[e][jq] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[e], xa[jq][ks], qacc[e][jq], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < 8; k += 2) {
                    const half2x v = half2x{xa[jq][ks][k], xa[jq][ks][k + 1]};
                    ssq[jq] = __builtin_amdgcn_fdot2(v, v, ssq[jq], false);
                }
            }
        }
        half8 qa[QF][2];
#pragma unroll
        for (int jq = 0; jq < QF; ++jq) {
            const float sum = __shfl(qacc[2][jq][0], (2 << 4) | l15);            // row 40 of q^T: fragment 2, lane group 2, register 0
            float ss = ssq[jq];
            ss += __shfl_xor(ss, 16);
            ss += __shfl_xor(ss, 32);
            const float mean = sum * inv_c;
            float var = ss * inv_c - mean * mean;
            if (__builtin_amdgcn_ballot_w64(mean * mean > LN_REDO_RATIO2 * var) != 0ull) {
                // |mean| >> std: the one-pass variance cancels; retake it as sum((x - mean)^2) from the operands still in registers
                // (rare; the same rule as the GEMM's in-kernel statistics, dm_kernels.h LN_REDO_RATIO2)
                float s2 = 0.f;
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
                    for (int k = 0; k < 8; ++k) { const float dlt = (float)xa[jq][ks][k] - mean; s2 = __builtin_fmaf(dlt, dlt, s2); }
                s2 += __shfl_xor(s2, 16);
                s2 += __shfl_xor(s2, 32);
                var = s2 * inv_c;
            }
            var = var > 0.f ? var : 0.f;
            const float rstd = 1.0f / __builtin_sqrtf(var + fq.ln_eps);
            const float nrm = -rstd * mean;
            f16 qv[EF][4];
#pragma unroll
            for (int e = 0; e < EF; ++e)
#pragma unroll
                for (int r = 0; r < 4; ++r) qv[e][r] = (f16)__builtin_fmaf(rstd, qacc[e][jq][r], __builtin_fmaf(nrm, sd[e][r], td[e][r]));
            const f16 z = (f16)0.0f;
            qa[jq][0] = half8{qv[0][0], qv[0][1], qv[0][2], qv[0][3], qv[1][0], qv[1][1], qv[1][2], qv[1][3]};
            if (lg < 2) qa[jq][1] = half8{qv[2][0], qv[2][1], qv[2][2], qv[2][3], z, z, z, z};
            else qa[jq][1] = half8{z, z, z, z, z, z, z, z};
        }
        // the token rows of the next query block: in flight during this block's attention
        if (qb + 1 < qb1) load_x(xa, qb + 1);

        // ---- S^T = K Q^T over the permuted k slots: five 16-key blocks ----
        floatx4 sacc[NKB][QF];
#pragma unroll
        for (int f = 0; f < NKB; ++f)
#pragma unroll
            for (int jq = 0; jq < QF; ++jq) sacc[f][jq] = floatx4{0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int f = 0; f < NKB; ++f) {
                const half8 kf = *reinterpret_cast<const half8*>(kbase + 64 * s + f * 16 * KRS);
#pragma unroll
                for (int jq = 0; jq < QF; ++jq) sacc[f][jq] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qa[jq][s], sacc[f][jq], 0, 0, 0);
            }
        // ---- softmax over the Tk real keys (a lane holds keys 16 f + 4 lg + r of query l15) ----
        half8 pb[QF][3];
#pragma unroll
        for (int jq = 0; jq < QF; ++jq) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (16 * (NKB - 1) + 4 * lg + r >= p.Tk) sacc[NKB - 1][jq][r] = -1e30f;
            float mx = sacc[0][jq][0];
#pragma unroll
            for (int f = 0; f < NKB; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = __builtin_fmaxf(mx, sacc[f][jq][r]);
            mx = __builtin_fmaxf(mx, __shfl_xor(mx, 16));
            mx = __builtin_fmaxf(mx, __shfl_xor(mx, 32));
            const float nm = -mx * sc;
#pragma unroll
            for (int f = 0; f < NKB; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    pb[jq][f >> 1][(f & 1) * 4 + r] = (f16)__builtin_amdgcn_exp2f(__builtin_fmaf(sacc[f][jq][r], sc, nm));
#pragma unroll
            for (int r = 0; r < 4; ++r) pb[jq][2][4 + r] = (f16)0.0f;             // keys 80..95 do not exist
        }
        // ---- O^T = V^T P: three 32-key steps ----
        floatx4 oacc[EF][QF];
#pragma unroll
        for (int e = 0; e < EF; ++e)
#pragma unroll
            for (int jq = 0; jq < QF; ++jq) oacc[e][jq] = floatx4{0, 0, 0, 0};
        static_for<3>([&](auto SS) __attribute__((always_inline)) {
            constexpr int ss = decltype(SS)::value;
            u32x2 vraw[EF][2];
            static_for<2 * EF>([&](auto R) __attribute__((always_inline)) {
                constexpr int r = decltype(R)::value, e = r >> 1, hh = r & 1;
                tr_read<32 * e + (2 * ss + hh) * 16 * VRS>(vraw[e][hh], vbase);
            });
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
        });
        // ---- O = O^T / l through a wave-private LDS tile: 16-byte stores, D * 2 contiguous bytes per query ----
        char* ost = smem + OOFF + wid * OST_WAVE;
#pragma unroll
        for (int jq = 0; jq < QF; ++jq) {
            const float l = __shfl(oacc[2][jq][0], (2 << 4) | l15);               // row 40 of O^T: the softmax denominator
            const float inv = 1.0f / l;
#pragma unroll
            for (int e = 0; e < EF; ++e) {
                const int d = 16 * e + 4 * lg;
                if (d < D) {
                    const half4 o = half4{(f16)(oacc[e][jq][0] * inv), (f16)(oacc[e][jq][1] * inv),
                                          (f16)(oacc[e][jq][2] * inv), (f16)(oacc[e][jq][3] * inv)};
                    *reinterpret_cast<half4*>(ost + (16 * jq + l15) * OST_ROW + d * 2) = o;
                }
            }
        }
        {
            constexpr int CH = D / 8;
            const int qw = qb * (64 * QF) + wid * (16 * QF);
#pragma unroll
            for (int i = 0; i < (32 * CH + 63) / 64; ++i) {
                const int c = lane + 64 * i;
                const int row = c / CH, ch = c - row * CH;
                if (c < 32 * CH && qw + row < p.Tq) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(ost + row * OST_ROW + ch * 16);
                    *reinterpret_cast<u32x4*>(Ob + (size_t)(qw + row) * p.ldo + ch * 8) = v;
                }
            }
        }
    }
}

}  // namespace

bool attention_crossq_supports(const AttnParams& p, const CrossQParams& f) {
    return p.D == D && p.heads * p.D == C && p.Tk > 64 && p.Tk <= KROWS && p.Tq >= 256 && f.X && f.Wq && f.ln_s && f.ln_t && f.ldx >= C &&
           (p.ldk % 4) == 0 && (f.ldx % 8) == 0;
}

hipError_t launch_attention_crossq(const AttnParams& p, const CrossQParams& f, hipStream_t s) {
    if (!attention_crossq_supports(p, f)) return hipErrorInvalidValue;
    const int nqb = (p.Tq + 64 * QF - 1) / (64 * QF);
    const long long pairs = (long long)p.B * p.heads;
    // enough blocks for ~6 rounds over the resident slots (2 blocks of 4 waves per CU), at least 4 query blocks each (the resident
    // weights are 31 KB per block: longer slices than attention_cross.hip's)
    const long long want = 6LL * 2 * device_cu_count();
    int nsplit = (int)((want + pairs - 1) / pairs);
    if (nsplit > nqb / 4) nsplit = nqb / 4;
    if (nsplit < 1) nsplit = 1;
    const long long units = (long long)p.B * nsplit;
    dim3 grid((unsigned)(((units + 7) / 8) * 8 * p.heads)), block(NT);
    static std::atomic<uint64_t> attr_seen{0};
    if (first_use_on_device(attr_seen))
        (void)hipFuncSetAttribute((const void*)attn_crossq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    launch_timed(attn_crossq_kernel, grid, block, (size_t)LDS_BYTES, s, p, f, nsplit);
    return hipGetLastError();
}

}  // namespace dm
