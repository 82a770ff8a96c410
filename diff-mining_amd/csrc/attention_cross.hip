// attention_cross.hip — the 77-key cross-attention of the transformer blocks (`attn2`: queries = image tokens, keys /
// values = the cached projections of the prompt's 77 CLIP tokens) for head_dim 40 and 80, reached from `unet(...)`,
// diffmining/typicality/compute.py:100.
//
// The generic kernel (attention.hip) treats it like any other attention: one block per 128 queries fetches K and V
// (12-25 KB through the LDS-DMA), runs two key tiles of its online softmax and leaves — a 5 us dependent chain per block
// that moves more K/V bytes than Q bytes and runs 2.3x off the read-Q-once / write-O-once HBM time.  Here:
//   * one block owns a (sample, head) pair — or a slice of its query blocks — and keeps K and V (80 key rows, 77 real)
//     resident in LDS for all of them: K/V traffic drops by the number of query blocks per pair (32 at 64x64);
//   * all 77 keys fit one pass (five 16-key score blocks), so the softmax is exact in one step: scores, row max,
//     exp2(sc*(s - max)), P·V — no running max, no rescale; keys 77..79 are masked before the max;
//   * no barrier after the prologue: K/V are read-only, so the four waves drift apart and one wave's exp2 overlaps the
//     others' MFMAs (tools/probes/probe_overlap.hip);
//   * the Q rows of the next query block are requested before the current block is computed (the loop is bound by
//     getting Q in and O out; a wave keeps 2.5-5 KB of loads in flight);
//   * same operand tricks as the other attention kernels: S^T = K Q^T so P is directly the PV B operand, V^T by
//     ds_read_b64_tr_b16, softmax denominator in a ones row of V^T (head_dim 40: a constant 16-byte chunk behind each
//     96-byte LDS row; head_dim 80: a register-constant row block, bare 160-byte rows — both strides keep the
//     transpose reads free of bank conflicts).
#include "dm_kernels.h"

#include <type_traits>
#include <utility>

namespace dm {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int NT = 256;               // threads per block (4 waves x 32 queries)
constexpr int QF = 2;                 // 16-query fragments per wave
constexpr int KROWS = 80;             // key rows held (five 16-key score blocks)
constexpr int VROWS = 96;             // value rows addressed by the three 32-key PV steps (rows >= Tk are zero)
constexpr int NKB = KROWS / 16;       // score blocks

template <int D>
struct XGeo {
    static constexpr int CH = D / 8;                        // real 16-byte chunks per row
    static constexpr int CC = (D % 16 != 0) ? 1 : 0;        // constant chunk behind the data (ones row inside a mixed V^T block)
    static constexpr int RS = (CH + CC) * 16;               // LDS row stride: 96 / 160 bytes (RS/4 mod 64 = 8 * odd: conflict-free V^T reads)
    static constexpr int KS = (D + 31) / 32;                // k steps of S^T = K Q^T
    static constexpr int EF = (D + 1 + 15) / 16;            // 16-row blocks of O^T
    static constexpr int EFV = CC ? EF : D / 16;            // ... of which read from LDS
    static constexpr int E_L = D / 16, LG_L = (D % 16) / 4;  // where row D of O^T lives in the accumulators
    static constexpr int PAD = 32;
    static constexpr int KOFF = 0, KBYTES = KROWS * RS + PAD;
    static constexpr int VOFF = KBYTES, VBYTES = VROWS * RS + PAD;
    static constexpr int LDS_KV = KBYTES + VBYTES;
    static constexpr int OST_ROW = D * 2 + 16;              // output staging row (bytes): 96 / 176
    static constexpr int OST_WAVE = 32 * OST_ROW;
    static constexpr int LDS = LDS_KV + 4 * OST_WAVE;
    static_assert((RS / 4) % 16 == 8, "row stride must spread eight rows over the 64 banks");
    static_assert(16 * 3 + 64 * (KS - 1) + 16 <= RS + PAD && 32 * (EFV - 1) + 8 * 3 + 8 <= RS + PAD, "fragment overrun must stay inside the pad");
};

template <int OFF>
__device__ __forceinline__ void tr_read(u32x2& out, unsigned base) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(out) : "v"(base), "n"(OFF) : "memory");
}
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <int D>
__global__ __launch_bounds__(NT, (D > 40 ? 2 : 3))
void attn_cross_kernel(AttnParams p, int nsplit) {
    using G = XGeo<D>;
    constexpr int RS = G::RS, KS = G::KS, EF = G::EF, EFV = G::EFV, KOFF = G::KOFF, VOFF = G::VOFF, CH = G::CH, CC = G::CC;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15;
    const int lg = lane >> 4;
    // Block order: the heads of one (sample, query slice) get ids that are equal mod 8 and adjacent in dispatch order, i.e.
    // the same XCD at the same time — a 640-byte token row holds the 80-byte slices of all eight heads, so the 128-byte
    // lines the heads share are fetched once into ONE L2 instead of once per XCD that hosts one of the heads.
    const int x = blockIdx.x & 7, t = blockIdx.x >> 3;
    const int h = t % p.heads;
    const int unit = (t / p.heads) * 8 + x;                  // (sample, slice)
    if (unit >= p.B * nsplit) return;
    const int b = unit / nsplit, part = unit - b * nsplit;
    int kvb = p.kv_slot ? p.kv_slot[b] : (p.slot_div > 0 ? b / p.slot_div : b);
    if (p.n_slots > 0) kvb = kvb < 0 ? 0 : (kvb < p.n_slots ? kvb : p.n_slots - 1);      // memory safety: never beyond the registered prompts
    const int nqb = (p.Tq + 64 * QF - 1) / (64 * QF);
    const int qb0 = (int)((long long)part * nqb / nsplit), qb1 = (int)((long long)(part + 1) * nqb / nsplit);

    const f16* Qb = p.Q + (size_t)(p.q_mod > 0 ? b % p.q_mod : b) * p.bsq + h * D;
    const f16* Kb = p.K + (size_t)kvb * p.bsk + h * D;
    const f16* Vb = p.V + (size_t)kvb * p.bsv + h * D;
    f16* Ob = p.O + (size_t)b * p.bso + h * D;

    // ---- Q rows of the first query block (their latency runs under the K/V fill) -------------------
    auto load_q = [&](half8 (&q)[QF][KS], int qb) __attribute__((always_inline)) {
#pragma unroll
        for (int jq = 0; jq < QF; ++jq) {
            int qi = qb * (64 * QF) + wid * (16 * QF) + 16 * jq + l15;
            qi = qi < p.Tq ? qi : p.Tq - 1;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int d = 32 * s + 8 * lg;
                if (d < D) q[jq][s] = *reinterpret_cast<const half8*>(Qb + (size_t)qi * p.ldq + d);
                else q[jq][s] = half8{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
    };
    half8 qa[QF][KS], qn[QF][KS];
    if (qb0 < qb1) load_q(qa, qb0);

    // ---- K and V of this (prompt, head) into LDS, once: rows >= Tk and the padding are zero, the constant chunk of a V
    //      row (head_dim 40) is {1, 0, ..} = the ones row of V^T ---------------------------------------
    {
        constexpr int RC = RS / 16;
        constexpr int NK = G::KBYTES / 16, NV = G::VBYTES / 16;
        for (int c = tid; c < NK + NV; c += NT) {
            const bool isv = c >= NK;
            const int cc = isv ? c - NK : c;
            const int row = cc / RC, ch = cc - row * RC;
            u32x4 val = u32x4{0u, 0u, 0u, 0u};
            if (row < p.Tk && ch < CH) val = *reinterpret_cast<const u32x4*>((isv ? Vb + (size_t)row * p.ldv : Kb + (size_t)row * p.ldk) + ch * 8);
            else if (CC && isv && ch == CH && row < VROWS) val = u32x4{0x00003C00u, 0u, 0u, 0u};
            *reinterpret_cast<u32x4*>(smem + (isv ? VOFF : KOFF) + cc * 16) = val;
        }
    }
    __syncthreads();

    const char* kbase = smem + KOFF + l15 * RS + 16 * lg;                                                // K fragment reads
    const unsigned vbase = (unsigned)(size_t)(smem + VOFF + (4 * lg + (l15 >> 2)) * RS + 8 * (l15 & 3));   // V^T transpose reads
    half8 ones_a;                       // A operand of the constant V^T row block (head_dim 80): lane row 0 = ones
    {
        const f16 o = (l15 == 0) ? (f16)1.0f : (f16)0.0f;
        ones_a = half8{o, o, o, o, o, o, o, o};
    }
    const float sc = p.scale * 1.44269504088896340736f;      // scores go to the log2 domain inside the exp2 argument

    for (int qb = qb0; qb < qb1; ++qb) {
        if (qb + 1 < qb1) load_q(qn, qb + 1);                // next block's rows: in flight during this block's math
        // ---- S^T = K Q^T: five 16-key blocks ----
        floatx4 sacc[NKB][QF];
#pragma unroll
        for (int f = 0; f < NKB; ++f)
#pragma unroll
            for (int jq = 0; jq < QF; ++jq) sacc[f][jq] = floatx4{0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int f = 0; f < NKB; ++f) {
                const half8 kf = *reinterpret_cast<const half8*>(kbase + 64 * s + f * 16 * RS);
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
            // (builtin max, not inline asm: these reads follow the score MFMAs directly, and only for instructions the
            //  compiler sees does it insert the wait states an MFMA result needs before a VALU read)
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
            u32x2 vraw[EFV][2];
            static_for<2 * EFV>([&](auto R) __attribute__((always_inline)) {
                constexpr int r = decltype(R)::value, e = r >> 1, hh = r & 1;
                tr_read<32 * e + (2 * ss + hh) * 16 * RS>(vraw[e][hh], vbase);
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < EF; ++e) {
                half8 va;
                if (e < EFV) {
                    __builtin_memcpy(&va, &vraw[e < EFV ? e : 0][0], 8);
                    __builtin_memcpy(reinterpret_cast<char*>(&va) + 8, &vraw[e < EFV ? e : 0][1], 8);
                } else {
                    va = ones_a;
                }
#pragma unroll
                for (int jq = 0; jq < QF; ++jq) oacc[e][jq] = __builtin_amdgcn_mfma_f32_16x16x32_f16(va, pb[jq][ss], oacc[e][jq], 0, 0, 0);
            }
        });
        // ---- O = O^T / l, staged through a wave-private LDS tile [32 queries][D] so that the global stores are 16 bytes per
        //      lane and D * 2 contiguous bytes per query (direct fragment stores are 8 bytes per lane, 32 per query) ----
        char* ost = smem + G::LDS_KV + wid * G::OST_WAVE;
#pragma unroll
        for (int jq = 0; jq < QF; ++jq) {
            const float l = __shfl(oacc[G::E_L][jq][0], (G::LG_L << 4) | l15);      // row D of O^T: the softmax denominator
            const float inv = 1.0f / l;
#pragma unroll
            for (int e = 0; e < EF; ++e) {
                const int d = 16 * e + 4 * lg;
                if (d < D) {
                    const half4 o = half4{(f16)(oacc[e][jq][0] * inv), (f16)(oacc[e][jq][1] * inv),
                                          (f16)(oacc[e][jq][2] * inv), (f16)(oacc[e][jq][3] * inv)};
                    *reinterpret_cast<half4*>(ost + (16 * jq + l15) * G::OST_ROW + d * 2) = o;
                }
            }
        }
        {
            const int qw = qb * (64 * QF) + wid * (16 * QF);
#pragma unroll
            for (int i = 0; i < (32 * CH + 63) / 64; ++i) {
                const int c = lane + 64 * i;
                const int row = c / CH, ch = c - row * CH;
                if (c < 32 * CH && qw + row < p.Tq) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(ost + row * G::OST_ROW + ch * 16);
                    *reinterpret_cast<u32x4*>(Ob + (size_t)(qw + row) * p.ldo + ch * 8) = v;
                }
            }
        }
#pragma unroll
        for (int jq = 0; jq < QF; ++jq)
#pragma unroll
            for (int s = 0; s < KS; ++s) qa[jq][s] = qn[jq][s];
    }
}

template <int D>
hipError_t launch_cross_t(const AttnParams& p, hipStream_t s) {
    const int nqb = (p.Tq + 64 * QF - 1) / (64 * QF);
    const long long pairs = (long long)p.B * p.heads;
    // enough blocks for ~6 rounds over the resident slots (2-3 blocks of 4 waves per CU), at least 4 query blocks each
    const long long want = 6LL * 3 * device_cu_count();
    int nsplit = (int)((want + pairs - 1) / pairs);
    if (nsplit > nqb / 4) nsplit = nqb / 4;
    if (nsplit < 1) nsplit = 1;
    const long long units = (long long)p.B * nsplit;
    dim3 grid((unsigned)(((units + 7) / 8) * 8 * p.heads)), block(NT);
    constexpr size_t lds = XGeo<D>::LDS;
    launch_timed(attn_cross_kernel<D>, grid, block, lds, s, p, nsplit);
    return hipGetLastError();
}

}  // namespace

bool attention_cross_supports(const AttnParams& p) {
    return (p.D == 40 || p.D == 80) && p.Tk > 64 && p.Tk <= KROWS && p.Tq >= 256;
}

hipError_t launch_attention_cross(const AttnParams& p, hipStream_t s) {
    if (!attention_cross_supports(p)) return hipErrorInvalidValue;
    return p.D == 40 ? launch_cross_t<40>(p, s) : launch_cross_t<80>(p, s);
}

}  // namespace dm
