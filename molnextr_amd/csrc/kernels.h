// kernels.h — internal launcher prototypes of libmolnextr_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mnx {

// F32: parity mode, every encoder operand in fp32 on the exact-fp32 MFMA (1/16 rate).
// BF16X3 / F16X3: split-operand modes. Every GEMM / attention operand v is carried as two 16-bit planes
// hi = T(v), lo = T(v - hi), and every product a.b is evaluated as ah.bh + ah.bl + al.bh on the 16-bit MFMA with fp32
// accumulation (the dropped al.bl term is 2^-22 relative for fp16, 2^-16 for bf16): fp32-class results at a third of the
// 16-bit MFMA rate instead of a sixteenth.
enum { MNX_DT_BF16 = 0, MNX_DT_F16 = 1, MNX_DT_F32 = 2, MNX_DT_BF16X3 = 3, MNX_DT_F16X3 = 4 };
inline bool dt_split(int dtype) { return dtype == MNX_DT_BF16X3 || dtype == MNX_DT_F16X3; }
inline int dt_base(int dtype) { return dtype == MNX_DT_BF16X3 ? MNX_DT_BF16 : dtype == MNX_DT_F16X3 ? MNX_DT_F16 : dtype; }
// bytes per operand element in HBM (split modes: two 2-byte planes)
inline size_t dt_size(int dtype) { return (dtype == MNX_DT_F32 || dt_split(dtype)) ? 4 : 2; }
enum { EPI_BIAS_16 = 0, EPI_GELU_16 = 1, EPI_RESID_F32 = 2, EPI_BIAS_F32 = 3 };

// Split-operand GEMM arguments (dtype BF16X3 / F16X3). *_lo = ELEMENT offset of an operand's lo plane from its hi plane
// (the pointer passed as A / W / C); C = epi(oscale * acc + bias): the weights of the fp16 split are stored scaled by a
// power of two per matrix (their lo plane would otherwise be subnormal), oscale undoes it exactly.
// terms: 3 = ah.wh + ah.wl + al.wh; 2 = ah.wh + ah.wl (the ACTIVATION's lo plane is neither read nor multiplied — a_lo is
// ignored —, the weight keeps both: the weight's rounding is systematic, the activation's is noise; compute_dtype FP16X3M,
// DESIGN.md section 4.3); 1 = ah.wh only (error-budget aid: the layer runs as the plain 16-bit mode would).
// c_planes (16-bit-output epilogues): 2 = hi and lo output planes; 1 = the hi plane only (c_lo ignored) — for a consumer
// that runs on two terms. The hi plane is the same bits either way.
struct SplitArgs {
    size_t a_lo = 0, w_lo = 0, c_lo = 0;
    float oscale = 1.f;
    int terms = 3;
    int c_planes = 2;
};

// ---- gemm.hip -------------------------------------------------------------------------------
// C[M,N] = epi(A[M,K] . W[N,K]^T + bias). A, W 16-bit (dtype). resid/C fp32 for EPI_RESID_F32 (may alias).
// dtype BF16X3 / F16X3: A, W (and C for the 16-bit-output epilogues) are hi planes, `sp` is required.
hipError_t launch_gemm16(int dtype, int epi, const void* A, const void* W, void* C, const float* bias,
                         const float* resid, int M, int N, int K, hipStream_t s, const SplitArgs* sp = nullptr);
// which kernel(s) launch_gemm16 runs a layer on: "x3" (gemm256x3), "x3+128" (whole rounds on gemm256x3, remaining rows on
// the 128x128 kernel), "g256", "gres", "128" — shape-only; used by tools/gemm_lab to label its table
const char* gemm16_route(int dtype, int epi, int M, int N, int K, int terms, bool has_bias);
// the 128x128-tile kernel of gemm.hip without the dispatch to gemm256.hip (tools/gemm_lab compares the two)
hipError_t launch_gemm16_tile128(int dtype, int epi, const void* A, const void* W, void* C, const float* bias,
                                 const float* resid, int M, int N, int K, hipStream_t s, const SplitArgs* sp = nullptr);

// ---- gemm256.hip: persistent 256x256-tile form for the 16-bit-output epilogues (bias, bias + GELU); bias required ----
bool gemm256_supports(int dtype, int epi, int M, int N, int K);   // shape / epilogue fit AND the tile count fills 256 CUs
hipError_t launch_gemm256(int dtype, int epi, const void* A, const void* W, void* C, const float* bias, int M, int N,
                          int K, hipStream_t s, const SplitArgs* sp = nullptr);
// gemm256x3_kernel (gemm256.hip): split dtypes with three (or, fp16 only, two) terms, any epilogue, M and N multiples of 256. Runs EVERY
// tile it is given on one workgroup per CU in rounds of 256: launch_gemm16 hands it whole rounds and the 128x128 kernel
// the remaining rows.
bool gemm256x3_supports(int dtype, int epi, int M, int N, int K);
// workgroups of a persistent launch = CUs the encoder stream may occupy (default 256, the whole chip)
void set_persistent_cus(int n);
int persistent_cus();
// measurement aids (gemm256.hip): shader clock of the gemm256x3_kernel launches since the last reset; sustained rate of a
// register-only fp16 MFMA loop on random operands (every CU, `iters` x 8 instructions per wave)
hipError_t x3_clock_read(double* mhz, bool reset);
hipError_t mfma_probe(int iters, hipStream_t s, double* tflops, double* mhz);
hipError_t launch_gemm256x3(int dtype, int epi, const void* A, const void* W, void* C, const float* bias,
                            const float* resid, int M, int N, int K, hipStream_t s, const SplitArgs* sp);

// ---- gemm_res.hip: persistent 256x128-tile form for the fp32-output epilogues (bias + fp32 residual, bias -> fp32) ----
bool gemm_res_supports(int dtype, int epi, int M, int N, int K);   // shape / epilogue fit
bool gemm_res_preferred(int dtype, int epi, int M, int N, int K);  // ... and measured faster than the 128x128 kernel
hipError_t launch_gemm_res(int dtype, int epi, const void* A, const void* W, float* C, const float* bias,
                           const float* resid, int M, int N, int K, hipStream_t s, const SplitArgs* sp = nullptr);

// ---- encoder.hip ----------------------------------------------------------------------------
// images [B,3,S,S] fp32 NCHW -> x [B,(S/4)^2,C] fp32 (conv 4x4/4 + bias + LayerNorm, eps 1e-5)
hipError_t launch_patch_embed(const float* img, const float* w_t /*[48][C]*/, const float* bias, const float* gamma,
                              const float* beta, float* x, int B, int S, int C, hipStream_t s);
// y16[M,C] = LayerNorm(x[M,C]) (eps) as 16-bit; optionally also fp32 copy y32. Split dtypes: y16 = hi plane, the lo plane
// is written y_lo ELEMENTS behind it. nonfinite_flag (device int, may be null): set to 1 when a row's result is not finite.
// Split dtypes with planes == 1: only the hi plane is written (the consumer runs on two terms), y_lo is ignored.
hipError_t launch_layernorm16(int dtype, const float* x, const float* gamma, const float* beta, void* y16, float* y32,
                              int M, int C, float eps, hipStream_t s, size_t y_lo = 0, int* nonfinite_flag = nullptr,
                              int planes = 2);
// patch-merging gather + LayerNorm(4C): x [B,H,W,C] fp32 -> y16 [B,(H/2)(W/2),4C]
hipError_t launch_merge_ln16(int dtype, const float* x, const float* gamma, const float* beta, void* y16, int B, int H,
                             int W, int C, float eps, hipStream_t s, size_t y_lo = 0, int planes = 2);
// window attention: qkv16 [B*H*W, 3C] -> out16 [B*H*W, C] (original token order); table [529, heads] fp32.
// Split dtypes: qkv_lo / out_lo = element offsets of the lo planes; terms as SplitArgs::terms.
hipError_t launch_window_attn(int dtype, const void* qkv16, const float* rel_table, void* out16, int B, int H, int W,
                              int C, int heads, int shift, hipStream_t s, size_t qkv_lo = 0, size_t out_lo = 0,
                              int terms = 3);
// y16 = T(x) (n elements). Split dtypes: hi plane = T(scale * x), lo plane (y_lo elements behind) = T(scale * x - hi).
hipError_t launch_cast16(int dtype, const float* x, void* y16, size_t n, hipStream_t s, size_t y_lo = 0,
                         float scale = 1.f);

// ---- preprocess.hip -------------------------------------------------------------------------
// HWC uint8 RGB page -> [3,S,S] fp32 (CropWhite(pad) + bilinear resize + gray + ImageNet normalise); bbox: 4 ints scratch
hipError_t launch_preprocess(const uint8_t* rgb, int H, int W, int pad, int square, int S, int* bbox, int* crop_out,
                             float* out, hipStream_t s);

// ---- decoder.hip ----------------------------------------------------------------------------
struct DecWeights;   // device pointers, see engine.cpp
struct DecState;
hipError_t launch_sgemm_tn(const float* A, const float* W, const float* bias, float* C, int M, int N, int K,
                           hipStream_t s, int perm_S = 0);

}  // namespace mnx
