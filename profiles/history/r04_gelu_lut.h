// gelu_lut.h — the erf-GELU of the GEGLU epilogues from a 1 KiB LDS table (r04).
//
//   gelu(x) = x Phi(x) = max(x, 0) - |x| T(|x|),   T(a) = 0.5 erfc(a / sqrt 2) = Phi(-a)
//   T(a) ~ ((c3 f + c2) f + c1) f + c0  on interval i = floor(t), f = t - i, t = min(a * 64/6, 63)   (gelu_table.h, generated)
//
// One 16-byte LDS read (the LDS pipe is idle in the epilogue) + 9 plain VALU operations per value; the r01-r03 form
// (Abramowitz-Stegun 7.1.26) was 12 plain operations + v_rcp_f32 + v_exp_f32, both quarter rate: ~20 issue slots, and the GEGLU
// epilogue is 40 % of a 320 -> 2560 tile (profiles/r03_probe_pingpong.txt: 6.2 us of 15.4).  The tail T is formed directly, so
// negative x has no 1 - 1 cancellation; entry 63 is zero and t is clamped to 63, so |x| >= 5.906 gives exactly x or -0.
// Accuracy over ALL 63 488 finite fp16 inputs (the gate is rounded to fp16 before the GELU): max |error| 2.4e-7; the fp16-rounded
// result differs from the correctly rounded value for 4 inputs (A&S: 257) — tools/gen_gelu_table.py,
// tests/test_gpu_ops.py::test_geglu_gelu_over_every_fp16_gate.
// The persistent 256 x 320 tile, the 128-row tiles and the split variants all call this function on the same table: bit-identical.
#pragma once
#include "gelu_table.h"

namespace dm {

typedef float gelu_f4 __attribute__((ext_vector_type(4)));

// tab: the table in LDS (1 KiB, 16-byte aligned)
__device__ __forceinline__ float gelu_lut(float x, const char* tab) {
    const float ax = __builtin_fabsf(x);
    const float t = __builtin_fminf(ax * GELU_TAB_SCALE, (float)(GELU_TAB_N - 1));
    const int i = (int)t;                                        // t >= 0: truncation = floor
    const float f = __builtin_amdgcn_fractf(t);                  // t - floor(t), exact
    const gelu_f4 c = *reinterpret_cast<const gelu_f4*>(tab + i * 16);
    // Horner as opaque scalar v_fma_f32: the SLP vectoriser otherwise pairs the two GELUs of a GEGLU quad into v_pk_fma_f32 and
    // pays 8 v_mov per pair to shuffle their coefficients (different table entries) into register pairs — more than it saves
    float T;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(T) : "v"(c[3]), "v"(f), "v"(c[2]));
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(T) : "v"(T), "v"(f), "v"(c[1]));
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(T) : "v"(T), "v"(f), "v"(c[0]));
    return __builtin_fmaf(-ax, T, __builtin_fmaxf(x, 0.f));
}

// NB values at once: all table reads are issued before the first polynomial, so a wave pays the LDS latency once per batch instead
// of once per value (the compiler does not batch the single-value form on its own: each read sits directly in front of its use).
// Same arithmetic per value as gelu_lut(): bit-identical.
template <int NB>
__device__ __forceinline__ void gelu_lut_batch(const float (&x)[NB], float (&y)[NB], const char* tab) {
    float f[NB];
    gelu_f4 c[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const float t = __builtin_fminf(__builtin_fabsf(x[k]) * GELU_TAB_SCALE, (float)(GELU_TAB_N - 1));
        f[k] = __builtin_amdgcn_fractf(t);
        c[k] = *reinterpret_cast<const gelu_f4*>(tab + (int)t * 16);
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        float T;
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(T) : "v"(c[k][3]), "v"(f[k]), "v"(c[k][2]));
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(T) : "v"(T), "v"(f[k]), "v"(c[k][1]));
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(T) : "v"(T), "v"(f[k]), "v"(c[k][0]));
        y[k] = __builtin_fmaf(-__builtin_fabsf(x[k]), T, __builtin_fmaxf(x[k], 0.f));
    }
}

}  // namespace dm
