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

// Split-operand modes (kernels.h MNX_DT_BF16X3 / MNX_DT_F16X3): v = hi + lo up to 2^-22 |v| (fp16; the lo plane may be
// subnormal, the MFMA does not flush 16-bit inputs) or 2^-17 |v| (bf16).
// hi and lo MUST be derived from the SAME fp32 value. Two things break that silently (both seen in round 3, on ~1e-5 of
// the elements, each worth a whole 16-bit ulp): (1) deriving hi in one place and lo in another lets the compiler evaluate
// v twice; (2) hipcc fuses `(T)(a * b)` / `(T)fma(a, b, c)` into v_fma_mixlo_f16, which rounds the EXACT product once to
// 16 bits, while `v - (float)hi` uses the fp32-rounded v: on a near-tie hi comes from one neighbour and lo from the other.
// The empty asm makes v an opaque fp32 register value, so hi = RN16(v) and lo = RN16(v - hi) see the same v.
template <typename T>
__device__ __forceinline__ void split16x4(f32x4 v, typename H16<T>::v4& hi, typename H16<T>::v4& lo) {
    float a = v[0], b = v[1], c = v[2], d = v[3];
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    hi = (typename H16<T>::v4){(T)a, (T)b, (T)c, (T)d};
    lo = (typename H16<T>::v4){(T)__fsub_rn(a, (float)hi[0]), (T)__fsub_rn(b, (float)hi[1]),
                               (T)__fsub_rn(c, (float)hi[2]), (T)__fsub_rn(d, (float)hi[3])};
}
// fp16: the same two roundings in three instructions per PAIR of values instead of seven (round 6; the window attention is
// bound by VALU issue and splits 44 values per query row and item). hi = v_cvt_pk_f16_f32 (RNE, the conversion the compiler
// picks for the casts above); lo = v_fma_mix{lo,hi}_f16(hi as f16 operand, -1.0, v): fma(hi, -1, v) = v - hi EXACTLY (hi is v
// rounded to 11 bits: the difference has at most 13 significant bits), so the instruction's single rounding to fp16 is the
// rounding of `(T)__fsub_rn(v, (float)hi)` — bit-identical results (tests: the encoder's output digests, tools/attn_lab).
// The compiler does not form this by itself: it canonicalises fma(x, -1, y) to a subtraction and converts hi back first.
// Used by the unmasked window-attention kernel only (split16x4_mix; the masked instantiation spills with it): the GEMM epilogues
// keep the compiler's form — their hand-counted vmcnt waits are pinned on the code shape the compiler gives them
// (tests/test_device_math.py), and they are not VALU-bound.
// One asm block per four values, the two pairs interleaved: a partial (op_sel) VGPR write followed directly by a VALU read of
// that register costs a wait state on gfx950 (the compiler puts an s_nop between dependent single-instruction blocks). lo is
// written INTO the registers of v[0] / v[2] (low half by mixlo, high half by mixhi): six registers in all; they are early-clobber
// ("+&v"): they are written before v[1] / v[3] are read, and the compiler gives operands it knows to be equal (the zeros of the
// absent key block) ONE register otherwise. The leading s_nop: a VALU read of a transcendental's result (the probabilities come
// from v_exp_f32, the context scale from v_rcp_f32) needs one wait state on gfx940+, and the compiler's hazard pass does not
// look for that in front of inline asm (it does put the wait state between this block's partial writes and their consumer).
// The packed fp32 forms of the neighbouring arithmetic (v_pk_fma_f32 for the scores, v_pk_add_f32 for the exponent arguments)
// were measured in the same pass and are NOT used: compiler-generated, 72 instructions fewer, and a few thousand to 10^5 of
// 7.5e7 output words WRONG, different ones each run (profiles/r06_attn_lab_diet.txt). The cause was not isolated: the pattern
// on its own — v_pk_add_f32 straight into v_exp_f32, nine-wave workgroups, with and without an MFMA stream — forwards
// correctly (tools/probes/pk_f32_forward.hip: 0 of 9.7e9 results differ), and the kernel without them is word-identical to
// the per-item kernel at every stage.
template <typename T>
__device__ __forceinline__ void split16x4_mix(f32x4 v, typename H16<T>::v4& hi, typename H16<T>::v4& lo) { split16x4<T>(v, hi, lo); }
template <>
__device__ __forceinline__ void split16x4_mix<f16_t>(f32x4 v, f16x4& hi, f16x4& lo) {
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    unsigned h0, h1;
    float a = v[0], b = v[1], c = v[2], d = v[3];
    asm("s_nop 0\n\t"
        "v_cvt_pk_f16_f32 %0, %2, %4\n\t"
        "v_cvt_pk_f16_f32 %1, %3, %5\n\t"
        "v_fma_mixlo_f16 %2, %0, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %3, %1, -1.0, %3 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %2, %0, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %3, %1, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(h0), "=&v"(h1), "+&v"(a), "+&v"(c)
        : "v"(b), "v"(d));
    hi = __builtin_bit_cast(f16x4, (u32x2_t){h0, h1});
    lo = __builtin_bit_cast(f16x4, (u32x2_t){__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, c)});
}

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

// GELU for 16-bit outputs (encoder MLP) on the packed fp32 pipe. x * Phi(x) with
// Phi(x) = 0.5 + xc * Q(u), xc = clamp(x, -5, 5), u = 2 xc^2 / 25 - 1, Q = degree-12 Chebyshev fit of (Phi(x) - 0.5) / x
// converted to monomials in u (sum |coef| = 0.4, so fp32 Horner is well conditioned). |error| <= 2.3e-6 absolute over the
// reals (checked against scipy erf in float32, tests/test_device_math.py), far below the bf16/fp16 rounding of the stored result;
// the outer factor is max(x, -5) so that the tail x < -5 stays at -5 Phi(-5) = -1.4e-6 instead of growing with |x|.
// 9 full-rate VALU operations per value instead of the 12 + two quarter-rate transcendentals (rcp, exp2) of an
// exp-based erf: the GELU epilogue of fc1 was VALU-bound. The fp32 parity mode uses erff.
// Four values per call: the two packed chains are independent, so back-to-back dependent packed operations (which cost a
// wait state each) never meet.
__device__ __forceinline__ f32x4 gelu_fast4(f32x4 x) {
    f32x4 xc;
    xc[0] = __builtin_amdgcn_fmed3f(x[0], -5.0f, 5.0f);
    xc[1] = __builtin_amdgcn_fmed3f(x[1], -5.0f, 5.0f);
    xc[2] = __builtin_amdgcn_fmed3f(x[2], -5.0f, 5.0f);
    xc[3] = __builtin_amdgcn_fmed3f(x[3], -5.0f, 5.0f);
    const f32x4 u = xc * xc * 0.08f - 1.0f;
    f32x4 q = u * 7.353763795e-04f + -1.676730928e-03f;
    q = q * u + 1.374596148e-03f;
    q = q * u + -2.526916796e-03f;
    q = q * u + 6.766527425e-03f;
    q = q * u + -1.130712498e-02f;
    q = q * u + 1.623608917e-02f;
    q = q * u + -2.321312763e-02f;
    q = q * u + 3.147675842e-02f;
    q = q * u + -4.045128077e-02f;
    q = q * u + 5.151792988e-02f;
    q = q * u + -7.029590756e-02f;
    q = q * u + 1.413638145e-01f;
    f32x4 xo;
    xo[0] = fmaxf(x[0], -5.0f); xo[1] = fmaxf(x[1], -5.0f); xo[2] = fmaxf(x[2], -5.0f); xo[3] = fmaxf(x[3], -5.0f);
    return xo * (xc * q + 0.5f);
}
__device__ __forceinline__ float gelu_fast(float x) { return gelu_fast4((f32x4){x, x, x, x})[0]; }

// GELU of the split-operand modes (fp32-class results). libm's erff is two divergent branches per value (a polynomial below
// |x| = 1, an exp-based form above): ~40 VALU + a transcendental + 5 SALU per value, 6 us of a 14 us fc1 epilogue in
// gemm256x3_kernel. gelu_poly16 is the construction of gelu_fast4 carried to fp32 accuracy: x * Phi(x),
// Phi = 0.5 + xc * Q(u), xc = clamp(x, +-5.5), u = 2 xc^2 / 5.5^2 - 1, Q the degree-16 Chebyshev fit of (Phi(x) - 0.5) / x in
// monomials of u, every step one FMA (packed: two values per instruction). Against the exact function its error is
// <= 1.3e-7 max(1, |x|) (rms 2e-8) — the formula the reference evaluates, 0.5 x (1 + erf(x / sqrt 2)) in fp32 with a
// correctly rounded erf, has 1.1e-7 max(1, |x|) (rms 1.2e-8): tests/test_device_math.py checks both numbers. x < -5.5 returns
// -5.5 Phi(-5.5) = -1.0e-7 (exact: -> 0). A LAB option, off by default (MNX_GELU_POLY = 0): the product's split-mode epilogues
// (gemm256.hip, gemm.hip) all call gelu_split4 = libm erff — measured, the shorter polynomial epilogue makes the power-limited
// kernel slower (DESIGN.md "time is energy") —, one function for both kernels so that an image's features stay independent
// of which kernel a row lands in.
#ifndef MNX_GELU_POLY
#define MNX_GELU_POLY 0
#endif
__device__ __forceinline__ f32x4 gelu_poly16(f32x4 x) {
    f32x4 xc;
    xc[0] = __builtin_amdgcn_fmed3f(x[0], -5.5f, 5.5f);
    xc[1] = __builtin_amdgcn_fmed3f(x[1], -5.5f, 5.5f);
    xc[2] = __builtin_amdgcn_fmed3f(x[2], -5.5f, 5.5f);
    xc[3] = __builtin_amdgcn_fmed3f(x[3], -5.5f, 5.5f);
    const f32x4 u = xc * xc * 6.611570248e-02f - 1.0f;
    f32x4 q = u * 1.1807999527e-04f + -2.8567631769e-04f;
    q = q * u + 1.8391666569e-04f;
    q = q * u + -3.5675461462e-04f;
    q = q * u + 1.4074264731e-03f;
    q = q * u + -2.6587023769e-03f;
    q = q * u + 4.0741296491e-03f;
    q = q * u + -6.6506783549e-03f;
    q = q * u + 1.0468637002e-02f;
    q = q * u + -1.5062524020e-02f;
    q = q * u + 2.0307316921e-02f;
    q = q * u + -2.6036151485e-02f;
    q = q * u + 3.2076909454e-02f;
    q = q * u + -3.8793793902e-02f;
    q = q * u + 4.7737334640e-02f;
    q = q * u + -6.4172314669e-02f;
    q = q * u + 1.2855193299e-01f;
    f32x4 xo;
    xo[0] = fmaxf(x[0], -5.5f); xo[1] = fmaxf(x[1], -5.5f); xo[2] = fmaxf(x[2], -5.5f); xo[3] = fmaxf(x[3], -5.5f);
    return xo * (xc * q + 0.5f);
}
__device__ __forceinline__ f32x4 gelu_split4(f32x4 v) {
#if MNX_GELU_POLY
    return gelu_poly16(v);
#else
    return (f32x4){gelu_erf(v[0]), gelu_erf(v[1]), gelu_erf(v[2]), gelu_erf(v[3])};
#endif
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

// one 16-byte-per-lane LDS-DMA: LDS[lds + 16 lane] <- sbase[voff] (scalar base, 32-bit lane offset: no 64-bit VGPR address;
// the builtin form made hipcc keep a zero-extended 64-bit copy of every lane offset and spill them — scratch traffic
// would also break the vmcnt bookkeeping). M0 (the LDS address of an LDS-DMA) is a reserved register that the compiler
// neither tracks nor preserves around inline asm: a kernel that uses these helpers issues EVERY LDS-DMA through them
// and uses nothing else that reads M0 (tests/test_device_math.py checks the generated ISA for both).
__device__ __forceinline__ void dma16(unsigned voff, const char* sbase, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds) : "memory");
}
__device__ __forceinline__ void dma4(unsigned voff, const char* sbase, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(sbase), "s"(lds) : "memory");
}
typedef __attribute__((address_space(3))) char lds_char_t;
