// r06: what do the k step's 28 fragment reads per wave cost beside its 80 MFMAs — issue slots or joules?
// probe_mix.hip (random operands): MFMAs alone 1532-1573 ns per step, + the 28 ds_read_b128 (no LDS-DMA) 1769-1825.  Here the same loop with the
// reads' LANE addresses (a) as in the kernel (64 lanes x 16 B from 64 different rows, conflict-free swizzle: every bank busy), (b) collapsed onto ONE
// 16-byte address per read (an LDS broadcast: the same ds_read_b128 instruction, the same issue slot and latency, one bank access instead of 64), and
// (c) issued but not consumed (the MFMAs keep their register operands).  (a) - (b) = what moving the bytes costs, (b) - none = what issuing costs.
//   hipcc --offload-arch=gfx950 -O3 -w tools/probes/probe_reads.hip -o tools/probes/bin/probe_reads && tools/probes/bin/probe_reads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int TP = 256, TC = 320;

// RD: 0 none, 1 distinct lane addresses (the kernel's), 2 broadcast;  USE: the MFMAs take the fragments just read (else: register operands)
template <int RD, int USE>
__global__ __launch_bounds__(512, 2) void reads_kernel(int nk, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int roff = RD == 2 ? ((wid & 3) * 64) * 128 : ((wid & 3) * 64 + l15) * 128 + ((lg ^ (l15 & 7)) << 4);
    for (int i = threadIdx.x; i < (TP + TC) * 64; i += 512) {
        unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        reinterpret_cast<_Float16*>(smem)[i] = (_Float16)(((float)(h & 0xFFFF) + (float)(h >> 16)) / 32768.0f - 2.0f);
    }
    __syncthreads();
    f4 acc[5][4];
    for (int i = 0; i < 5; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    h8 a[5], b[4];
    for (int i = 0; i < 9; ++i)
        for (int k = 0; k < 8; ++k) {
            unsigned h = (threadIdx.x * 9u + (unsigned)i) * 2654435761u + blockIdx.x * 40503u + (unsigned)k * 2246822519u;
            h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            const float v = ((float)(h & 0xFFFF) + (float)(h >> 16)) / 32768.0f - 2.0f;
            if (i < 5) a[i][k] = (_Float16)v; else b[i - 5][k] = (_Float16)v;
        }
    // the reads land in registers of their own, one quarter (five MFMA groups) ahead of the point where they are "consumed" (an empty asm that forces
    // the wait there, as the kernel's MFMAs do): no latency is exposed, and the MFMAs keep their random register operands in every arm, so the
    // arms differ in the reads alone.  USE = 1: the fragments DO replace the operands (arm 1 only: the kernel's real data flow)
    h8 da[5], db[4];
    for (int i = 0; i < 5; ++i) da[i] = a[i];
    for (int j = 0; j < 4; ++j) db[j] = b[j];
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (RD && (q == 1 || q == 3)) {           // the four pixel fragments of the NEXT 32-deep half step
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (USE) b[j] = db[j]; else asm volatile("" :: "v"(db[j]));
                    db[j] = *reinterpret_cast<const h8*>(smem + 40960 + roff + j * 2048 + (q == 1 ? 64 : 0));
                }
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                if (RD) { if (USE) a[i] = da[i]; else asm volatile("" :: "v"(da[i])); }      // the fragment requested a quarter ago
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int j = ((q * 5 + i) & 1) ? 3 - jj : jj;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
                }
                if (RD) da[i] = *reinterpret_cast<const h8*>(smem + roff + i * 2048 + ((q + 1) & 1) * 10240 + (((q + 1) >> 1) & 1) * 64);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float sum = 0.f;
    for (int i = 0; i < 5; ++i) for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][3];
    if (sum == 1.2345e-30f) sink[0] = sum;
}

template <int RD, int USE>
double run(int n_cu, int nk, float* sink) {
    const size_t lds = (TP + TC) * 128;
    (void)hipFuncSetAttribute((const void*)reads_kernel<RD, USE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((reads_kernel<RD, USE>), dim3(n_cu), dim3(512), lds, 0, nk / 8 + 1, sink);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((reads_kernel<RD, USE>), dim3(n_cu), dim3(512), lds, 0, nk, sink);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return ms * 1e6 / nk;                               // ns per 64-deep step (80 MFMAs + 28 reads per wave)
}

int main(int argc, char** argv) {
    const int nk = argc > 1 ? atoi(argv[1]) : 100000, rounds = argc > 2 ? atoi(argv[2]) : 4;
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    float* sink;
    (void)hipMalloc(&sink, 64);
    const char* names[5] = {"MFMAs alone", "+ 28 reads (kernel lane addresses) that REPLACE the operands", "+ 28 broadcast reads that replace the operands (MFMA data: all lanes equal)",
                            "+ 28 reads, kernel lane addresses, operands untouched", "+ 28 broadcast reads, operands untouched"};
    double sum[5] = {0, 0, 0, 0, 0};
    printf("# ns per step of 80 MFMAs (+ 28 ds_read_b128) per wave, eight waves per CU, random fp16 data, no LDS-DMA, no barrier; arms in mirrored order\n");
    for (int r = 0; r < rounds; ++r) {
        double v[5];
        auto one = [&](int k) {
            switch (k) {
                case 0: v[0] = run<0, 0>(p.multiProcessorCount, nk, sink); break;
                case 1: v[1] = run<1, 1>(p.multiProcessorCount, nk, sink); break;
                case 2: v[2] = run<2, 1>(p.multiProcessorCount, nk, sink); break;
                case 3: v[3] = run<1, 0>(p.multiProcessorCount, nk, sink); break;
                default: v[4] = run<2, 0>(p.multiProcessorCount, nk, sink); break;
            }
        };
        if (r & 1) for (int k = 4; k >= 0; --k) one(k); else for (int k = 0; k < 5; ++k) one(k);
        printf("round %d:", r);
        for (int k = 0; k < 5; ++k) { printf("  %.0f", v[k]); sum[k] += v[k]; }
        printf("\n");
    }
    for (int k = 0; k < 5; ++k) printf("%-62s %7.1f ns  (%+.1f %%)\n", names[k], sum[k] / rounds, (sum[k] / sum[0] - 1.0) * 100.0);
    return 0;
}
