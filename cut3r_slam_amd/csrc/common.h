// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels.  Wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

typedef _Float16 h16;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CUT3R_OK 0
#define CUT3R_ERR_ARG 1
#define CUT3R_ERR_LAUNCH 2

#define DEVINL __device__ __forceinline__

DEVINL float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
DEVINL float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
DEVINL int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// RoPE rotation of one element: a*co + t in fp32, THEN rounded to the token type.  The empty asm keeps the fp32 value
// live so the compiler cannot merge the FMA and the fp16 conversion into v_fma_mixlo_f16 (one rounding instead of two):
// the stand-alone kernel and the GEMM-fused epilogue must round identically, and the reference (kernels.cu:50-53) computes
// in float and converts on the store as well.
DEVINL float rope_rot(float a, float co, float t) {
    float r = fmaf(a, co, t);
    asm volatile("" : "+v"(r));
    return r;
}

// exact GELU (nn.GELU default, erf form) -- src/croco/models/blocks.py:75 act_layer=nn.GELU
DEVINL float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// The same function for fp16 OUTPUTS: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, three orders below the fp16
// rounding of the result), written on |x| so the negative branch has no 1 + erf cancellation.  ~14 VALU instructions
// instead of libm's ~35 with branches: in a GEMM epilogue this is the difference between VALU-bound and free.
DEVINL float gelu_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(t, p, 1.421413741f);
    p = fmaf(t, p, -0.284496736f);
    p = fmaf(t, p, 0.254829592f);
    const float h = 0.5f * (p * t) * __builtin_amdgcn_exp2f(-z * z * 1.44269504088896340736f);   // 0.5 * (1 - erf(z))
    return x >= 0.0f ? x * (1.0f - h) : x * h;
}

static inline int cut3r_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? CUT3R_OK : CUT3R_ERR_LAUNCH;
}
