// Not built: the software-pipelined head_dim-40 fp32 attention tried in r05 and measured slower than attn32_kernel
// (profiles/r05_ab_attn32p.txt: 100.7 vs 104.5 TF/s on the DIFT fp32 workload, same box, alternating).  Kept as a record; it
// was a drop-in inside diff-mining_amd/csrc/f32_ops.hip (namespace dm32, after attn32_kernel).

static bool getenv_attn32p() { static const int v = [] { const char* e = getenv("DM_ATTN32P"); return e ? atoi(e) : 1; }(); return v != 0; }

// ---- head_dim 40, software-pipelined (r05) ---------------------------------------------------------------------------------------
// attn32_kernel runs QK^T(t) -> softmax(t) -> P V(t) in sequence: while a wave does the ~400 VALU instructions of a tile's softmax it
// issues no MFMA, and the blocks of a CU run the same program in step (profiles/r04_final_dift_f32_pmc.json: matrix pipe 71 % busy).  An
// fp32 16x16x4 MFMA occupies the pipe for 32 cycles and the issue port for a fraction of that, so ONE wave can carry the softmax of
// tile t in the shadow of the QK^T MFMAs of tile t + 1: two score tiles (S_cur / S_nxt), three LDS stages (K(t+1) is read one iteration
// before V(t+1)), one barrier per tile; same arithmetic in the same order as attn32_kernel (bit-identical).
template <int D>
__global__ __launch_bounds__(256, 2) void attn32p_kernel(AttnParams p) {
    constexpr int KT = 64, QT = 2, LDK = D + 2, LD = D + 4, DT = (D + 15) / 16, KS = D / 4, NKT = KT / 16;
    constexpr int STG = KT * LDK + KT * LD + 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;
    float* Vs = smem + KT * LDK;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, c = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * (64 * QT) + wid * (16 * QT);
    int kb = b;
    if (p.kv_slot) { kb = p.kv_slot[b]; kb = kb < 0 ? 0 : (kb >= p.n_slots ? p.n_slots - 1 : kb); }
    const float* Qb = p.Q + (long long)b * p.bsq + h * D;
    const float* Kb = p.K + (long long)kb * p.bsk + h * D;
    const float* Vb = p.V + (long long)kb * p.bsv + h * D;
    const float qscale = p.scale * 1.44269504088896340736f;
    typedef float v2f __attribute__((ext_vector_type(2)));

    float qf[QT][KS];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        int qi = q0 + qt * 16 + c; qi = qi < p.Tq ? qi : p.Tq - 1;
#pragma unroll
        for (int s = 0; s < KS; ++s) qf[qt][s] = Qb[(long long)qi * p.ldq + 4 * s + g] * qscale;
    }
    v4f o[DT][QT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) o[dt][qt] = v4f{0.f, 0.f, 0.f, 0.f};
    float mrun[QT], lrun[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) { mrun[qt] = -INFINITY; lrun[qt] = 0.f; }

    constexpr int NLD = (KT * (D / 4) + 255) / 256;
    v4f kreg[NLD], vreg[NLD];
    auto gload = [&](int k0) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + 256 * j;
            const int r = i / (D / 4), c4 = i - r * (D / 4), key = k0 + r;
            kreg[j] = v4f{0.f, 0.f, 0.f, 0.f}; vreg[j] = v4f{0.f, 0.f, 0.f, 0.f};
            if (i < KT * (D / 4) && key < p.Tk) {
                kreg[j] = *reinterpret_cast<const v4f*>(Kb + (long long)key * p.ldk + 4 * c4);
                vreg[j] = *reinterpret_cast<const v4f*>(Vb + (long long)key * p.ldv + 4 * c4);
            }
        }
    };
    auto lstore = [&](int stage) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + 256 * j;
            if (i < KT * (D / 4)) {
                const int r = i / (D / 4), c4 = i - r * (D / 4);
                float* kd = Ks + stage * STG + r * LDK + 4 * c4;
                *reinterpret_cast<v2f*>(kd) = v2f{kreg[j][0], kreg[j][1]};
                *reinterpret_cast<v2f*>(kd + 2) = v2f{kreg[j][2], kreg[j][3]};
                *reinterpret_cast<v4f*>(Vs + stage * STG + r * LD + 4 * c4) = vreg[j];
            }
        }
    };
    auto qk = [&](v4f (&S)[NKT][QT], int stage) {
        const float* Kc = Ks + stage * STG;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) S[kt][qt] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                const float a = Kc[(kt * 16 + c) * LDK + 4 * s + g];
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) S[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, qf[qt][s], S[kt][qt], 0, 0, 0);
            }
    };
    auto softmax = [&](v4f (&S)[NKT][QT], int k0) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = k0 + kt * 16 + 4 * g + r;
                    const float v = key < p.Tk ? S[kt][qt][r] : -INFINITY;
                    S[kt][qt][r] = v;
                    mx = fmaxf(mx, v);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mnew = fmaxf(mrun[qt], mx);
            const float alpha = fast_exp2(mrun[qt] - mnew);
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = fast_exp2(S[kt][qt][r] - mnew);
                    S[kt][qt][r] = pv;
                    sum += pv;
                }
            lrun[qt] = lrun[qt] * alpha + sum;
            mrun[qt] = mnew;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) o[dt][qt] *= alpha;
        }
    };
    auto pv = [&](const v4f (&S)[NKT][QT], int stage) {
        const float* Vc = Vs + stage * STG;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const float a = Vc[(kt * 16 + 4 * g + r) * LD + dt * 16 + c];
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) o[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, S[kt][qt][r], o[dt][qt], 0, 0, 0);
                }
    };

    const int nt = (p.Tk + KT - 1) / KT;
    v4f SA[NKT][QT], SB[NKT][QT];
    gload(0); lstore(0);
    if (nt > 1) gload(KT);
    __syncthreads();
    if (nt > 1) { lstore(1); if (nt > 2) gload(2 * KT); }
    qk(SA, 0);
    __syncthreads();
    // iteration t: scores of tile t in X; K(t+1) / V(t) are in stages (t+1) % 3 / t % 3; the registers hold tile t + 2
    int st = 0;                                   // t % 3
    auto iteration = [&](v4f (&X)[NKT][QT], v4f (&Y)[NKT][QT], int t, const bool next) {
        const int s1 = st == 2 ? 0 : st + 1, s2 = s1 == 2 ? 0 : s1 + 1;
        if (t + 2 < nt) { lstore(s2); if (t + 3 < nt) gload((t + 3) * KT); }
        __builtin_amdgcn_sched_barrier(0);
        if (next) qk(Y, s1);                      // matrix work of the next tile: independent of the softmax below
        softmax(X, t * KT);
        if (next) {
            // one scheduling region: ask for the softmax's VALU instructions in the shadow of the 80 QK^T MFMAs (a 16x16x4 fp32 MFMA holds
            // the pipe for 32 cycles; hipcc otherwise emits the MFMAs as one block and the softmax behind it)
#pragma unroll
            for (int i = 0; i < KS * NKT * QT; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
                if ((i & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // DS read (one K value feeds two MFMAs)
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);      // VALU
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        pv(X, st);
        __syncthreads();
        st = s1;
    };
    int t = 0;
    for (; t + 2 < nt; t += 2) { iteration(SA, SB, t, true); iteration(SB, SA, t + 1, true); }
    if (t + 1 < nt) { iteration(SA, SB, t, true); iteration(SB, SA, t + 1, false); }
    else iteration(SA, SB, t, false);

#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float l = lrun[qt];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const float inv = 1.0f / l;
        const int qi = q0 + qt * 16 + c;
        if (qi >= p.Tq) continue;
        float* orow = p.O + (long long)b * p.bso + (long long)qi * p.ldo + h * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d = dt * 16 + 4 * g;
            if (d < D) *reinterpret_cast<v4f*>(orow + d) = o[dt][qt] * inv;
        }
    }
}

template <int D>
hipError_t launch_attn_p(const AttnParams& p, hipStream_t s) {
    const size_t lds = (size_t)(64 * (D + 2) + 64 * (D + 4) + 16) * 3 * sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn32p_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((attn32p_kernel<D>), dim3((p.Tq + 127) / 128, p.heads, p.B), dim3(256), lds, s, p);
    return hipGetLastError();
}

