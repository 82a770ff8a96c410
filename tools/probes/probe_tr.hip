// Probe: semantics of ds_read_b64_tr_b16 on gfx950.  LDS holds u16 value = element index.
// Each lane reads 8 bytes at byte offset addr[lane]; prints the 4 u16 each lane received.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned short u16;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const int* addr, unsigned* out) {
    __shared__ __attribute__((aligned(16))) u16 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (u16)i;
    __syncthreads();
    unsigned a = (unsigned)(size_t)(&lds[0]) + addr[threadIdx.x];
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    out[threadIdx.x * 2] = v[0]; out[threadIdx.x * 2 + 1] = v[1];
}
int main() {
    int* da; unsigned* dout;
    hipMalloc(&da, 64 * 4); hipMalloc(&dout, 128 * 4);
    for (int test = 0; test < 2; ++test) {
        std::vector<int> addr(64);
        // test 0: lane l -> byte l*8 (contiguous 512 B).  test 1: row-major [key][64 d] image:
        // lane l (group g=l>>4, i=l&15): key = 4g + i/4, d = 4*(i%4): byte = key*128 + d*2
        for (int l = 0; l < 64; ++l) {
            int g = l >> 4, i = l & 15;
            addr[l] = test == 0 ? l * 8 : ((4 * g + i / 4) * 128 + 4 * (i % 4) * 2);
        }
        hipMemcpy(da, addr.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, dout);
        std::vector<unsigned> out(128);
        hipMemcpy(out.data(), dout, 512, hipMemcpyDeviceToHost);
        printf("test %d\n", test);
        for (int l = 0; l < 64; ++l)
            printf("lane %2d addr %4d -> %4u %4u %4u %4u\n", l, addr[l], out[2 * l] & 0xffff, out[2 * l] >> 16,
                   out[2 * l + 1] & 0xffff, out[2 * l + 1] >> 16);
    }
    return 0;
}
