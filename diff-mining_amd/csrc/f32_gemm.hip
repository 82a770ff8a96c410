// f32_gemm.hip — fp32 implicit GEMM on the fp32 matrix cores (v_mfma_f32_16x16x4_f32), gfx950.
//
// The arithmetic of the reference's DIFT featuriser: `SDFeaturizer` builds its pipeline without torch_dtype and calls the U-Net
// without autocast (diffmining/typicality/dift.py:197-199, 191), i.e. every convolution / linear is an fp32 GEMM.  Same operator
// as igemm_tile.h (3x3 convolution stride 1 / 2 / after nearest up-sampling, 1x1 convolution, linear; two concatenated sources;
// bias + per-sample time embedding + residual in the epilogue) on fp32 tensors.
//
// Roofline: the fp32 MFMA rate is 256 FLOP/clk/CU (157 TFLOP/s) — 1/16 of the fp16 rate — so at this block tile (128 pixels x 160
// channels x 16 k per step: 80 MFMAs of 32 cycles per wave and step against 9 ds_read_b128 and 4.5 16-byte global loads per
// thread) the kernel is bound by the matrix pipe alone; nothing of the fp16 tile's LDS-DMA machinery is needed.
//   block = 4 waves (2 x 2), wave tile 64 pixels x 80 channels = 4 x 5 MFMA tiles, accumulators 80 VGPRs
//   MFMA operands: A = weights [channel i][k], B = activations [k][pixel j] -> D[i][j]: a lane holds 4 consecutive channels of one
//   pixel, so the epilogue stores 16 bytes per lane into the NHWC row
//   LDS: two stages of (160 + 128) rows x 16 floats (36 KiB), filled from registers (global loads of step t+1 in flight under the
//   MFMAs of step t), one barrier per step; a fragment read is lane (c, g) -> row c, floats 4g..4g+3: the 64 lanes cover 1 KiB
//   contiguously (conflict-free), register jj of the read pairs with k = 4g + jj in BOTH operands
#include "f32_kernels.h"

namespace dm32 {
namespace {

constexpr int BM = 128, BN = 160, BK = 16, NT = 256;
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(NT, 2) void gemm32_kernel(GemmParams p, int tiles_m, int tiles_n, int per_xcd) {
    __shared__ __attribute__((aligned(16))) float Ws[2][BN][BK];
    __shared__ __attribute__((aligned(16))) float Xs[2][BM][BK];
    // consecutive block ids land on different XCDs (8 L2s): give each XCD a contiguous range of tiles (channel tiles fastest: the
    // blocks of one XCD share their activation rows in its L2)
    const int tile = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (tile >= tiles_m * tiles_n) return;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid & 1, wn = wid >> 1;
    const int c = lane & 15, g = lane >> 4;
    const int K = (p.mode == 0 ? 1 : 9) * p.Cin;
    const int OHW = p.OH * p.OW;

    // the two activation rows this thread loads (rows tid/4 and tid/4 + 64 of the tile, floats 4 (tid & 3) .. + 3 of a k step)
    int xn[2], xoy[2], xox[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + (tid >> 2) + 64 * i;
        if (m < p.M) {
            if (p.mode == 0) { xn[i] = 0; xoy[i] = 0; xox[i] = m; }
            else { const int n = m / OHW, rem = m - n * OHW; xn[i] = n; xoy[i] = rem / p.OW; xox[i] = rem - xoy[i] * p.OW; }
        } else { xn[i] = -1; xoy[i] = 0; xox[i] = 0; }
    }
    const float sh = (float)p.H / (float)p.OH, sw = (float)p.W / (float)p.OW;

    auto gload = [&](int kt, v4f (&xr)[2], v4f (&wr)[3]) {
        const int k0 = kt * BK;
        int tap = 0, c0 = k0;
        if (p.mode != 0) { tap = k0 / p.Cin; c0 = k0 - tap * p.Cin; }
        const bool second = c0 >= p.C1;
        const float* base = second ? p.X2 : p.X;
        const int cs = second ? p.Cin - p.C1 : p.C1;
        const int cc = (second ? c0 - p.C1 : c0) + 4 * (tid & 3);
        const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            long long off = -1;
            if (xn[i] >= 0) {
                if (p.mode == 0) off = xox[i];
                else if (p.mode == 3) {              // F.interpolate(mode="nearest") to (OH, OW), then the 3x3 convolution (pad 1)
                    const int uh = xoy[i] + dy - 1, uw = xox[i] + dx - 1;
                    if (uh >= 0 && uh < p.OH && uw >= 0 && uw < p.OW) {
                        int ih = (int)floorf((float)uh * sh); ih = ih < p.H - 1 ? ih : p.H - 1;
                        int iw = (int)floorf((float)uw * sw); iw = iw < p.W - 1 ? iw : p.W - 1;
                        off = ((long long)xn[i] * p.H + ih) * p.W + iw;
                    }
                } else {
                    const int st = (p.mode == 1) ? 1 : 2, pd = (p.mode == 4) ? 0 : 1;
                    const int ih = xoy[i] * st + dy - pd, iw = xox[i] * st + dx - pd;
                    if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) off = ((long long)xn[i] * p.H + ih) * p.W + iw;
                }
            }
            xr[i] = (off >= 0) ? *reinterpret_cast<const v4f*>(base + off * cs + cc) : v4f{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int idx = tid + NT * i;
            v4f w = {0.f, 0.f, 0.f, 0.f};
            if (idx < BN * 4) {
                const int ch = n0 + (idx >> 2);
                if (ch < p.Cout) w = *reinterpret_cast<const v4f*>(p.Wp + (size_t)ch * K + k0 + 4 * (idx & 3));
            }
            wr[i] = w;
        }
    };
    auto lstore = [&](int buf, const v4f (&xr)[2], const v4f (&wr)[3]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<v4f*>(&Xs[buf][(tid >> 2) + 64 * i][4 * (tid & 3)]) = xr[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int idx = tid + NT * i;
            if (idx < BN * 4) *reinterpret_cast<v4f*>(&Ws[buf][idx >> 2][4 * (idx & 3)]) = wr[i];
        }
    };

    v4f acc[5][4];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK;
    v4f xr[2], wr[3];
    gload(0, xr, wr);
    lstore(0, xr, wr);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1, xr, wr);
        v4f a[5], b[4];
#pragma unroll
        for (int ct = 0; ct < 5; ++ct) a[ct] = *reinterpret_cast<const v4f*>(&Ws[buf][wn * 80 + ct * 16 + c][4 * g]);
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) b[pt] = *reinterpret_cast<const v4f*>(&Xs[buf][wm * 64 + pt * 16 + c][4 * g]);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int ct = 0; ct < 5; ++ct)
#pragma unroll
                for (int pt = 0; pt < 4; ++pt)
                    acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ct][jj], b[pt][jj], acc[ct][pt], 0, 0, 0);
        if (kt + 1 < nk) lstore(buf ^ 1, xr, wr);
        __syncthreads();
    }

    // epilogue: lane (c, g) holds channels ch .. ch + 3 of pixel m for every (ct, pt)
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
        const int m = m0 + wm * 64 + pt * 16 + c;
        if (m >= p.M) continue;
        const float* tb = p.temb ? p.temb + (size_t)(m / OHW) * p.temb_ld : nullptr;
#pragma unroll
        for (int ct = 0; ct < 5; ++ct) {
            const int ch = n0 + wn * 80 + ct * 16 + 4 * g;
            if (ch >= p.Cout) continue;
            v4f v = acc[ct][pt];
            if (p.bias) v += *reinterpret_cast<const v4f*>(p.bias + ch);
            if (tb) v += *reinterpret_cast<const v4f*>(tb + ch);
            if (p.res) v += *reinterpret_cast<const v4f*>(p.res + (size_t)m * p.ldres + ch);
            *reinterpret_cast<v4f*>(p.Y + (size_t)m * p.ldy + ch) = v;
        }
    }
}

}  // namespace

hipError_t launch_gemm(const GemmParams& p, hipStream_t s) {
    const int taps = p.mode == 0 ? 1 : 9;
    if (p.M <= 0 || p.Cout <= 0 || p.Cin % BK != 0 || p.C1 % BK != 0 || p.Cout % 4 != 0 || (p.C1 < p.Cin && !p.X2) ||
        (long long)taps * p.Cin > (1LL << 30) || p.ldy % 4 != 0 || (p.res && p.ldres % 4 != 0) || (p.temb && p.temb_ld % 4 != 0))
        return hipErrorInvalidValue;
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.Cout + BN - 1) / BN;
    const long long tiles = (long long)tiles_m * tiles_n;
    const int per_xcd = (int)((tiles + 7) / 8);
    hipLaunchKernelGGL(gemm32_kernel, dim3((unsigned)(per_xcd * 8)), dim3(NT), 0, s, p, tiles_m, tiles_n, per_xcd);
    return hipGetLastError();
}

}  // namespace dm32
