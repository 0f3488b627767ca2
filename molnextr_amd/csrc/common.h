// molnextr_amd/csrc/common.h — shared device/host helpers for libmolnextr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) float f32x8;

#define MNX_WAVE 64

// 16-bit GEMM operand type traits: the MFMA that consumes it.
template <typename T> struct H16;
template <> struct H16<bf16_t> {
    typedef bf16x8 v8;
    typedef bf16x4 v4;
    static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct H16<f16_t> {
    typedef f16x8 v8;
    typedef f16x4 v4;
    static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};

// fp32 "parity mode" (compute_dtype FP32): the same kernels instantiated on float operands. The 16x16x32 contraction is
// eight exact-fp32 v_mfma_f32_16x16x4_f32 steps: lane (fr, fg) holds k = 8 fg + i (i = 0..7) of its row on BOTH operands,
// step i feeds element i as k-slot fg, so the four lane groups cover k = i, 8+i, 16+i, 24+i — every k exactly once.
template <> struct H16<float> {
    typedef f32x8 v8;
    typedef f32x4 v4;
    static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) {
#pragma unroll
        for (int i = 0; i < 8; ++i) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[i], c, 0, 0, 0);
        return c;
    }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// exact-erf GELU, as nn.GELU / F.gelu default (reference transformers.py:201, components.py:356, onmt 'gelu')
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// GELU for 16-bit outputs (encoder MLP): erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, far below the bf16/fp16
// rounding of the stored result) — ~12 VALU ops instead of libm erff's ~40.
__device__ __forceinline__ float gelu_fast(float x) {
    // x * Phi(x), Phi(x) = 1 - u (x >= 0) or u (x < 0), u = 0.5 * p(t) * exp(-x^2/2), t = 1 / (1 + 0.3275911 |x| / sqrt2)
    const float a = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.23164190f, a, 1.0f));
    float p = fmaf(t, 0.5307027145f, -0.7265760135f);       // A&S 7.1.26 coefficients, pre-multiplied by 0.5
    p = fmaf(t, p, 0.7107068705f);
    p = fmaf(t, p, -0.142248368f);
    p = fmaf(t, p, 0.127414796f);
    const float u = p * t * __builtin_amdgcn_exp2f(a * a * -0.72134752044f);   // exp(-a^2/2) = 2^(-a^2/2 * log2 e)
    return x * (x >= 0.f ? 1.0f - u : u);
}

// bijective XCD-aware remap of a linear workgroup id (guide T1): consecutive ids land on the same XCD/L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int NX = 8;
    if (nwg < NX * 2) return bid;
    int q = nwg / NX, r = nwg % NX;
    int xcd = bid % NX, k = bid / NX;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}
