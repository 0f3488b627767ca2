// kernels.h — internal launcher prototypes of libmolnextr_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mnx {

enum { MNX_DT_BF16 = 0, MNX_DT_F16 = 1, MNX_DT_F32 = 2 };   // F32: parity mode, every encoder operand in fp32
inline size_t dt_size(int dtype) { return dtype == MNX_DT_F32 ? 4 : 2; }
enum { EPI_BIAS_16 = 0, EPI_GELU_16 = 1, EPI_RESID_F32 = 2, EPI_BIAS_F32 = 3 };

// ---- gemm.hip -------------------------------------------------------------------------------
// C[M,N] = epi(A[M,K] . W[N,K]^T + bias). A, W 16-bit (dtype). resid/C fp32 for EPI_RESID_F32 (may alias).
hipError_t launch_gemm16(int dtype, int epi, const void* A, const void* W, void* C, const float* bias,
                         const float* resid, int M, int N, int K, hipStream_t s);
// the 128x128-tile kernel of gemm.hip without the dispatch to gemm256.hip (tools/gemm_lab compares the two)
hipError_t launch_gemm16_tile128(int dtype, int epi, const void* A, const void* W, void* C, const float* bias,
                                 const float* resid, int M, int N, int K, hipStream_t s);

// ---- gemm256.hip: persistent 256x256-tile form for the 16-bit-output epilogues (bias, bias + GELU); bias required ----
bool gemm256_supports(int dtype, int epi, int M, int N, int K);   // shape / epilogue fit AND the tile count fills 256 CUs
hipError_t launch_gemm256(int dtype, int epi, const void* A, const void* W, void* C, const float* bias, int M, int N,
                          int K, hipStream_t s);

// ---- encoder.hip ----------------------------------------------------------------------------
// images [B,3,S,S] fp32 NCHW -> x [B,(S/4)^2,C] fp32 (conv 4x4/4 + bias + LayerNorm, eps 1e-5)
hipError_t launch_patch_embed(const float* img, const float* w_t /*[48][C]*/, const float* bias, const float* gamma,
                              const float* beta, float* x, int B, int S, int C, hipStream_t s);
// y16[M,C] = LayerNorm(x[M,C]) (eps) as 16-bit; optionally also fp32 copy y32
hipError_t launch_layernorm16(int dtype, const float* x, const float* gamma, const float* beta, void* y16, float* y32,
                              int M, int C, float eps, hipStream_t s);
// patch-merging gather + LayerNorm(4C): x [B,H,W,C] fp32 -> y16 [B,(H/2)(W/2),4C]
hipError_t launch_merge_ln16(int dtype, const float* x, const float* gamma, const float* beta, void* y16, int B, int H,
                             int W, int C, float eps, hipStream_t s);
// window attention: qkv16 [B*H*W, 3C] -> out16 [B*H*W, C] (original token order); table [529, heads] fp32
hipError_t launch_window_attn(int dtype, const void* qkv16, const float* rel_table, void* out16, int B, int H, int W,
                              int C, int heads, int shift, hipStream_t s);
hipError_t launch_cast16(int dtype, const float* x, void* y16, size_t n, hipStream_t s);

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
