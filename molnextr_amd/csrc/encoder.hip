// encoder.hip — non-GEMM kernels of the Swin encoder.
//
// Activations are token-major [B, L, C] (= channels-last). The residual stream is fp32; the operands of the
// MFMA GEMMs are produced here as 16-bit (bf16 by default).
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace mnx {

// =============================================================================================
// K1  patch embedding: Conv2d(3, C, k=4, s=4) + bias -> LayerNorm(C)     (reference transformers.py:405-419)
//     One workgroup = one patch row of one image, 8 threads per patch position, NP patches per thread.
// =============================================================================================
// The [48][C] weight image (24 KiB at C = 128) and the row's 12 pixel lines are staged in LDS once per workgroup. Thread
// (p = tid >> 3, part = tid & 7) owns CPT = C/8 consecutive channels (<= 16) of the NP patches p, p + 32, ... of a chunk of
// 32 NP patches: every weight quad read from LDS feeds NP patches. What the first form of this kernel (one patch per thread,
// C a run-time argument; 0.18 of the HBM rate its 3.3 GB per 512 images need, profiles/r05_bench_steps20.json.log) lost, in the
// order it was found in the ISA:
//   * `if (j < cpt)` with a run-time cpt made every weight quad a branch of its own: ds_read, s_waitcnt lgkmcnt(0), a few
//     FMAs, branch — each LDS round trip exposed. CPT is a template argument now: straight-line code, reads issued ahead.
//   * the staging loops waited for every global load before its LDS store (one load in flight per thread, ten round trips per
//     workgroup). All of a thread's quads are requested first, at clamped addresses (a predicated load is a branch).
//   * one LDS float per FMA (now 20 quads per 192 FMAs), and at C = 128 the eight channel segments of a patch sit 64 B apart:
//     parts p and p + 4 on the same banks in every ds_read_b128 lane group (MI355X_MICROARCH.md, LDS) — rows are C + 4 floats
//     and the upper half of a row is stored 4 floats later.
// Per output the arithmetic is what it was: bias, then the 48 taps in (channel, ky, kx) order, the LayerNorm sums over the
// thread's channels and then over the 8 lanes of the patch. The variance and the affine step are explicit fmaf (sq = fma(d, d,
// sq); fma(d * rstd, gamma, beta)): that is what the first form compiled to, and left to -ffp-contract the compiler vectorised
// the three-patch form into separate multiplies and adds — the encoder output changed in its last bits (tools/features_hash.py
// compares the two libraries: identical now).
constexpr int PE_NP = 3;            // 96 patches per chunk: one chunk per row at 384 x 384
template <int CPT>
__global__ __launch_bounds__(256) void patch_embed_kernel(const float* __restrict__ img, const float* __restrict__ w_t,
                                                          const float* __restrict__ bias,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ x,
                                                          int S, int G, int rpw) {
    constexpr int C = 8 * CPT;
    constexpr int PW = 128 * PE_NP;             // pixels per staged line
    constexpr int WS = C + 4;                   // weight row stride in LDS
    constexpr bool SWZ = CPT == 16;             // parts 4..7 (channels 64..127) stored 4 floats later
    constexpr int NWQ = (12 * C + 255) / 256;   // weight quads per thread
    constexpr int NPQ = (12 * (PW / 4) + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* wt = sm;             // [48][WS]
    float* pix = sm + 48 * WS;  // [3][4][PW]
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int p = tid >> 3, part = tid & 7;
    const int c0 = part * CPT;
    f32x4 wq[NWQ];
#pragma unroll
    for (int k = 0; k < NWQ; ++k) wq[k] = ((const f32x4*)w_t)[min(tid + 256 * k, 12 * C - 1)];
    // a workgroup takes `rpw` consecutive patch rows (round 6: two at G % 2 == 0 — the 24 KB of weights are staged once per
    // workgroup, a quarter of what a patch row moves into the CU) in chunks of 96 patches; per output nothing changes
    const int nchunk = (G + 32 * PE_NP - 1) / (32 * PE_NP);
    for (int ch = 0; ch < rpw * nchunk; ++ch) {
        const int py = blockIdx.x * rpw + ch / nchunk, px0 = (ch % nchunk) * 32 * PE_NP;
        f32x4 pq[NPQ];
#pragma unroll
        for (int k = 0; k < NPQ; ++k) {
            const int i = min(tid + 256 * k, 12 * (PW / 4) - 1);
            const int line = i / (PW / 4), xq = i % (PW / 4);       // line = ci * 4 + ky
            const int gx = px0 * 4 + xq * 4;                        // S % 4 == 0: a quad is inside the image or outside it
            pq[k] = *(const f32x4*)(img + ((size_t)(b * 3 + (line >> 2)) * S + (py * 4 + (line & 3))) * S + min(gx, S - 4));
        }
        f32x4 bq[CPT / 4];
#pragma unroll
        for (int j = 0; j < CPT / 4; ++j) bq[j] = *(const f32x4*)(bias + c0 + 4 * j);
        __builtin_amdgcn_sched_barrier(0);      // keep the requests together, ahead of the first wait
        if (ch > 0) __syncthreads();            // the previous chunk's pixels have been consumed
        else {
#pragma unroll
            for (int k = 0; k < NWQ; ++k) {
                const int i = tid + 256 * k;
                const int row = i / (C / 4), c = (i % (C / 4)) * 4;
                if (i < 12 * C) *(f32x4*)(wt + row * WS + c + (SWZ ? (c >> 6) * 4 : 0)) = wq[k];
            }
        }
#pragma unroll
        for (int k = 0; k < NPQ; ++k) {
            const int i = tid + 256 * k;
            const int line = i / (PW / 4), xq = i % (PW / 4);
            f32x4 v = pq[k];
            if (px0 * 4 + xq * 4 >= S) v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (i < 12 * (PW / 4)) *(f32x4*)(pix + line * PW + xq * 4) = v;
        }
        __syncthreads();
        float acc[PE_NP][CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j)
#pragma unroll
            for (int q = 0; q < PE_NP; ++q) acc[q][j] = bq[j >> 2][j & 3];
        // tap loop, software-pipelined by hand: the weight quads of tap t + 1 (and, at kx = 3, the next line's pixels) are
        // requested before tap t's FMAs, the scheduling barrier keeps a tap's requests ahead of the previous tap's FMAs. (244
        // registers at C = 128: two workgroups per CU, where the LDS would allow three — capped at 168 registers the compiler
        // spills 70; two are enough to cover one workgroup's staging with the other's taps.)
        const float* wbase = wt + c0 + (SWZ ? (part >> 2) * 4 : 0);
        const float* pbase = pix + p * 4;
        f32x4 wc[CPT / 4], pv[PE_NP];
#pragma unroll
        for (int j = 0; j < CPT / 4; ++j) wc[j] = *(const f32x4*)(wbase + 4 * j);
#pragma unroll
        for (int q = 0; q < PE_NP; ++q) pv[q] = *(const f32x4*)(pbase + q * 128);
#pragma unroll 1
        for (int line = 0; line < 12; ++line) {
            f32x4 pn[PE_NP];
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) {
                f32x4 wn[CPT / 4];
                const float* wr = wbase + min(line * 4 + kx + 1, 47) * WS;
#pragma unroll
                for (int j = 0; j < CPT / 4; ++j) wn[j] = *(const f32x4*)(wr + 4 * j);
                if (kx == 3) {
                    const float* pr = pbase + min(line + 1, 11) * PW;
#pragma unroll
                    for (int q = 0; q < PE_NP; ++q) pn[q] = *(const f32x4*)(pr + q * 128);
                }
#pragma unroll
                for (int j = 0; j < CPT; j += 4) {
                    const f32x4 w4 = wc[j >> 2];
#pragma unroll
                    for (int q = 0; q < PE_NP; ++q) {
                        const float v = pv[q][kx];
                        acc[q][j] = fmaf(v, w4[0], acc[q][j]); acc[q][j + 1] = fmaf(v, w4[1], acc[q][j + 1]);
                        acc[q][j + 2] = fmaf(v, w4[2], acc[q][j + 2]); acc[q][j + 3] = fmaf(v, w4[3], acc[q][j + 3]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < CPT / 4; ++j) wc[j] = wn[j];
            }
#pragma unroll
            for (int q = 0; q < PE_NP; ++q) pv[q] = pn[q];
        }
        __builtin_amdgcn_sched_barrier(0);      // gamma / beta requested here, not above the tap loop (32 registers)
        f32x4 g4[CPT / 4], b4[CPT / 4];
#pragma unroll
        for (int j = 0; j < CPT / 4; ++j) { g4[j] = *(const f32x4*)(gamma + c0 + 4 * j); b4[j] = *(const f32x4*)(beta + c0 + 4 * j); }
#pragma unroll
        for (int q = 0; q < PE_NP; ++q) {
            const int px = px0 + q * 32 + p;
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < CPT; ++j) s += acc[q][j];
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
            const float mean = s / (float)C;
            float sq = 0.f;
#pragma unroll
            for (int j = 0; j < CPT; ++j) { acc[q][j] -= mean; sq = fmaf(acc[q][j], acc[q][j], sq); }
            sq += __shfl_xor(sq, 1, 64); sq += __shfl_xor(sq, 2, 64); sq += __shfl_xor(sq, 4, 64);
            const float rstd = rsqrtf(sq / (float)C + 1e-5f);
            if (px < G) {
                float* o = x + ((size_t)(b * G + py) * G + px) * C + c0;
#pragma unroll
                for (int j = 0; j < CPT; j += 4) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(acc[q][j + e] * rstd, g4[j >> 2][e], b4[j >> 2][e]);
                    *(f32x4*)(o + j) = v;
                }
            }
        }
    }
}

hipError_t launch_patch_embed(const float* img, const float* w_t, const float* bias, const float* gamma,
                              const float* beta, float* x, int B, int S, int C, hipStream_t s) {
    if (C > 128 || (C & 31) || (S & 3) || S < 4) return hipErrorInvalidValue;
    const int G = S / 4;
    const int rpw = (G % 2 == 0) ? 2 : 1;
    dim3 grid(G / rpw, B), block(256);
    size_t smem = (size_t)(48 * (C + 4) + 3 * 4 * 128 * PE_NP) * sizeof(float);
    switch (C >> 3) {
        case 4: hipLaunchKernelGGL(patch_embed_kernel<4>, grid, block, smem, s, img, w_t, bias, gamma, beta, x, S, G, rpw); break;
        case 8: hipLaunchKernelGGL(patch_embed_kernel<8>, grid, block, smem, s, img, w_t, bias, gamma, beta, x, S, G, rpw); break;
        case 12: hipLaunchKernelGGL(patch_embed_kernel<12>, grid, block, smem, s, img, w_t, bias, gamma, beta, x, S, G, rpw); break;
        default: hipLaunchKernelGGL(patch_embed_kernel<16>, grid, block, smem, s, img, w_t, bias, gamma, beta, x, S, G, rpw); break;
    }
    return hipGetLastError();
}

// =============================================================================================
// K2  LayerNorm over the channel dim, fp32 in -> 16-bit out (GEMM operand) [+ fp32 out].  One wave per row.
//     MERGE=true fuses the PatchMerging 2x2 gather (reference transformers.py:325-333): logical row
//     (b, y2, x2) is the concat of x[b, 2y2+dy, 2x2+dx, :] for (dy,dx) = (0,0),(1,0),(0,1),(1,1).
// =============================================================================================
//     SPLIT (dtypes BF16X3 / F16X3): the 16-bit output is two planes, hi and lo = v - hi (y_lo elements behind); y_lo == 0
//     inside the kernel means "hi plane only" (launch_layernorm16 planes == 1: the consumer runs on two terms).
//     LPR = lanes per row: 64, or 32 for C <= 128 (stage 1: a 64-lane wave would keep half its lanes idle on the largest
//     M of the encoder — two rows per wave instead; the butterfly sums are bit-identical, the upper half only added zeros).
template <int LPR>
__device__ __forceinline__ float row_sum(float v) {     // butterfly over the LPR lanes that share a row
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <typename T, bool MERGE, int NV, bool SPLIT, int LPR = 64>
__global__ __launch_bounds__(256) void layernorm16_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, T* __restrict__ y16,
                                                          float* __restrict__ y32, int M, int C, float eps, int H,
                                                          int W, int Cin, size_t y_lo, int* __restrict__ flag) {
    typedef typename H16<T>::v4 v4;
    constexpr int RPW = 64 / LPR;                      // rows per wave
    const int lane = threadIdx.x & (LPR - 1);
    const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + ((threadIdx.x & 63) / LPR);
    if (row >= M) return;
    const float* src[4];
    if (MERGE) {
        const int W2 = W >> 1, H2 = H >> 1;
        const int x2 = row % W2, y2 = (row / W2) % H2, b = row / (W2 * H2);
#pragma unroll
        for (int p = 0; p < 4; ++p)
            src[p] = x + ((size_t)(b * H + 2 * y2 + (p & 1)) * W + 2 * x2 + (p >> 1)) * Cin;
    } else {
        src[0] = x + (size_t)row * C;
    }
    f32x4 v[NV];  // C <= 256 * NV
    float sum = 0.f;
    // All NV loads of the row are issued before anything is waited for: written as "if (e < C) load", every load was a branch
    // of its own and the compiler waited for each (NV serial memory round trips per row). The address is clamped into the row
    // instead (a lane beyond C re-reads the row's last 16 bytes) and the value zeroed by a select.
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int e = lane * 4 + j * 256;
        const int ec = e < C ? e : C - 4;
        if (MERGE) {
            const int p = ec / Cin;
            v[j] = *(const f32x4*)(src[p] + (ec - p * Cin));
        } else {
            v[j] = *(const f32x4*)(src[0] + ec);
        }
    }
    // gamma / beta likewise (NV <= 4: 8 NV registers): loaded per chunk inside the store loop below, each chunk's wait for
    // them also sat out the previous chunk's store acknowledgements (loads and stores retire through one in-order counter)
    constexpr bool GB_EARLY = NV <= 4;
    f32x4 gv[GB_EARLY ? NV : 1], bv[GB_EARLY ? NV : 1];
    if (GB_EARLY) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int e = lane * 4 + j * 256;
            const int ec = e < C ? e : C - 4;
            gv[j] = *(const f32x4*)(gamma + ec);
            bv[j] = *(const f32x4*)(beta + ec);
        }
    }
    __builtin_amdgcn_sched_barrier(0);      // the scheduler would otherwise sink each load next to its first use again
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int e = lane * 4 + j * 256;
        if (!(e < C)) v[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        sum += v[j][0] + v[j][1] + v[j][2] + v[j][3];
    }
    const float mean = row_sum<LPR>(sum) / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int e = lane * 4 + j * 256;
        if (e < C) {
            v[j] -= mean;
            sq += v[j][0] * v[j][0] + v[j][1] * v[j][1] + v[j][2] * v[j][2] + v[j][3] * v[j][3];
        }
    }
    const float rstd = rsqrtf(row_sum<LPR>(sq) / (float)C + eps);
    // one non-finite input poisons mean and variance of the whole row: rstd is NaN (or 0 for an infinite variance)
    if (flag && lane == 0 && !(rstd > 0.f && rstd < 3.0e38f)) atomicOr(flag, 1);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int e = lane * 4 + j * 256;
        if (e < C) {
            const f32x4 g = GB_EARLY ? gv[GB_EARLY ? j : 0] : *(const f32x4*)(gamma + e);
            const f32x4 bt = GB_EARLY ? bv[GB_EARLY ? j : 0] : *(const f32x4*)(beta + e);
            const f32x4 o = v[j] * rstd * g + bt;
            if (y16) {
                if (SPLIT) {
                    v4 hi, lo;
                    split16x4<T>(o, hi, lo);
                    *(v4*)(y16 + (size_t)row * C + e) = hi;
                    if (y_lo) *(v4*)(y16 + y_lo + (size_t)row * C + e) = lo;
                } else {
                    v4 o4 = {(T)o[0], (T)o[1], (T)o[2], (T)o[3]};
                    *(v4*)(y16 + (size_t)row * C + e) = o4;
                }
            }
            if (y32) *(f32x4*)(y32 + (size_t)row * C + e) = o;
        }
    }
}

template <typename T, bool MERGE, bool SPLIT>
static void ln_dispatch(dim3 grid, hipStream_t s, const float* x, const float* gamma, const float* beta, T* y16,
                        float* y32, int M, int C, float eps, int H, int W, int Cin, size_t y_lo, int* flag) {
    dim3 block(256);
    if (C <= 128 && !MERGE)     // two rows per wave
        hipLaunchKernelGGL((layernorm16_kernel<T, MERGE, 1, SPLIT, 32>), dim3((grid.x + 1) / 2), block, 0, s, x, gamma, beta, y16, y32, M, C, eps, H, W, Cin, y_lo, flag);
    else if (C <= 256)
        hipLaunchKernelGGL((layernorm16_kernel<T, MERGE, 1, SPLIT>), grid, block, 0, s, x, gamma, beta, y16, y32, M, C, eps, H, W, Cin, y_lo, flag);
    else if (C <= 512)
        hipLaunchKernelGGL((layernorm16_kernel<T, MERGE, 2, SPLIT>), grid, block, 0, s, x, gamma, beta, y16, y32, M, C, eps, H, W, Cin, y_lo, flag);
    else if (C <= 1024)
        hipLaunchKernelGGL((layernorm16_kernel<T, MERGE, 4, SPLIT>), grid, block, 0, s, x, gamma, beta, y16, y32, M, C, eps, H, W, Cin, y_lo, flag);
    else
        hipLaunchKernelGGL((layernorm16_kernel<T, MERGE, 8, SPLIT>), grid, block, 0, s, x, gamma, beta, y16, y32, M, C, eps, H, W, Cin, y_lo, flag);
}

template <bool MERGE>
static hipError_t ln_by_dtype(int dtype, dim3 grid, hipStream_t s, const float* x, const float* gamma, const float* beta,
                              void* y16, float* y32, int M, int C, float eps, int H, int W, int Cin, size_t y_lo,
                              int* flag, int planes) {
    if (planes != 1 && planes != 2) return hipErrorInvalidValue;
    if (dt_split(dtype) && y16 && planes == 2 && y_lo == 0) return hipErrorInvalidValue;
    if (planes == 1) y_lo = 0;
    switch (dtype) {
        case MNX_DT_F16: ln_dispatch<f16_t, MERGE, false>(grid, s, x, gamma, beta, (f16_t*)y16, y32, M, C, eps, H, W, Cin, 0, flag); break;
        case MNX_DT_F32: ln_dispatch<float, MERGE, false>(grid, s, x, gamma, beta, (float*)y16, y32, M, C, eps, H, W, Cin, 0, flag); break;
        case MNX_DT_BF16: ln_dispatch<bf16_t, MERGE, false>(grid, s, x, gamma, beta, (bf16_t*)y16, y32, M, C, eps, H, W, Cin, 0, flag); break;
        case MNX_DT_F16X3: ln_dispatch<f16_t, MERGE, true>(grid, s, x, gamma, beta, (f16_t*)y16, y32, M, C, eps, H, W, Cin, y_lo, flag); break;
        case MNX_DT_BF16X3: ln_dispatch<bf16_t, MERGE, true>(grid, s, x, gamma, beta, (bf16_t*)y16, y32, M, C, eps, H, W, Cin, y_lo, flag); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_layernorm16(int dtype, const float* x, const float* gamma, const float* beta, void* y16, float* y32,
                              int M, int C, float eps, hipStream_t s, size_t y_lo, int* nonfinite_flag, int planes) {
    if (C > 2048 || (C & 3)) return hipErrorInvalidValue;
    return ln_by_dtype<false>(dtype, dim3((M + 3) / 4), s, x, gamma, beta, y16, y32, M, C, eps, 0, 0, 0, y_lo, nonfinite_flag, planes);
}

hipError_t launch_merge_ln16(int dtype, const float* x, const float* gamma, const float* beta, void* y16, int B, int H,
                             int W, int C, float eps, hipStream_t s, size_t y_lo, int planes) {
    if (4 * C > 2048 || (C & 3) || (H & 1) || (W & 1)) return hipErrorInvalidValue;
    const int M = B * (H / 2) * (W / 2);
    return ln_by_dtype<true>(dtype, dim3((M + 3) / 4), s, x, gamma, beta, y16, nullptr, M, 4 * C, eps, H, W, C, y_lo, nullptr, planes);
}

template <typename T>
__global__ void cast16_kernel(const float* __restrict__ x, T* __restrict__ y, size_t n4) {
    typedef typename H16<T>::v4 v4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 v = *(const f32x4*)(x + i * 4);
        v4 o = {(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
        *(v4*)(y + i * 4) = o;
    }
}
// split modes: hi = T(scale * x), lo = T(scale * x - hi); scale is a power of two (exact)
template <typename T>
__global__ void cast16_split_kernel(const float* __restrict__ x, T* __restrict__ y, size_t n4, size_t y_lo, float scale) {
    typedef typename H16<T>::v4 v4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 v = *(const f32x4*)(x + i * 4) * scale;
        v4 hi, lo;
        split16x4<T>(v, hi, lo);
        *(v4*)(y + i * 4) = hi;
        *(v4*)(y + y_lo + i * 4) = lo;
    }
}

hipError_t launch_cast16(int dtype, const float* x, void* y16, size_t n, hipStream_t s, size_t y_lo, float scale) {
    if (n & 3) return hipErrorInvalidValue;
    const size_t n4 = n / 4;
    dim3 grid((unsigned)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048)), block(256);
    if (dt_split(dtype)) {
        if (y_lo < n) return hipErrorInvalidValue;
        if (dtype == MNX_DT_F16X3)
            hipLaunchKernelGGL((cast16_split_kernel<f16_t>), grid, block, 0, s, x, (f16_t*)y16, n4, y_lo, scale);
        else
            hipLaunchKernelGGL((cast16_split_kernel<bf16_t>), grid, block, 0, s, x, (bf16_t*)y16, n4, y_lo, scale);
    } else if (dtype == MNX_DT_F16)
        hipLaunchKernelGGL((cast16_kernel<f16_t>), grid, block, 0, s, x, (f16_t*)y16, n4);
    else if (dtype == MNX_DT_F32)
        hipLaunchKernelGGL((cast16_kernel<float>), grid, block, 0, s, x, (float*)y16, n4);
    else
        hipLaunchKernelGGL((cast16_kernel<bf16_t>), grid, block, 0, s, x, (bf16_t*)y16, n4);
    return hipGetLastError();
}

// =============================================================================================
// K4  (shifted-)window attention, window 12x12 = 144 tokens, head_dim 32
//     (reference transformers.py:68-97 partition/reverse, :147-178 attention, :220-243 shift mask, :260-282 roll)
//
//     One workgroup (9 waves, one 16-query tile each) per (image, window, head). The cyclic shift, the window partition and their
//     inverses are index arithmetic on the token row — nothing is materialised: the kernel reads q,k,v of
//     the 144 window tokens straight from the [B*L, 3C] qkv buffer and writes the head's 32 output channels
//     back at the tokens' ORIGINAL rows. Scores live in MFMA accumulators only.
//
//     S^T = K.Q^T is computed key-major ("swapped") so every lane owns ONE query column: softmax statistics
//     are 36 in-register values + two cross-lane steps, and the exponentiated probabilities are already in
//     the B-operand layout of the second MFMA  O^T = V^T.P^T  (V is transposed once through LDS). The k-slot
//     order of that MFMA is permuted identically on both operands so no data movement is needed.
// =============================================================================================
constexpr int WS = 12, WN = 144, HD = 32;
#ifndef MNX_ATTN_STAMP          // tools/attn_lab defines it to record per-workgroup cycle counters; nothing in the product
#define MNX_ATTN_STAMP(i)
#endif
constexpr int KS_STRIDE = 40;    // elements per K row in LDS (80 B: conflict-free ds_read_b128)
constexpr int VT_STRIDE = 168;   // elements per V^T row in LDS (336 B: conflict-free ds_read_b64), keys 144..167 zero

template <typename T>
__global__ __launch_bounds__(576) void window_attn_kernel(const T* __restrict__ qkv,
                                                                  const float* __restrict__ table, T* __restrict__ out,
                                                                  int H, int W, int C, int heads, int shift) {
    constexpr int QT = 1, NTHR = 576;   // 9 waves, one 16-query tile each
    typedef typename H16<T>::v8 v8;
    typedef typename H16<T>::v4 v4;
    __shared__ __attribute__((aligned(16))) T Ks[WN * KS_STRIDE];
    __shared__ __attribute__((aligned(16))) T Vt[HD * VT_STRIDE];
    __shared__ float tab[(2 * WS - 1) * (2 * WS - 1)];
    __shared__ int rowof[WN];
    __shared__ __attribute__((aligned(16))) int kinfo[WN];   // per key: (ky*23 + kx) | region id << 16

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nWw = W / WS, nWh = H / WS;
    // XCD-aware: the heads of one window are consecutive ids, and consecutive ids share an XCD (= an L2), so the two
    // heads that share every 128-byte line of q / k / v / out hit the same L2 instead of fetching the line twice
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int head = bid % heads; bid /= heads;
    const int wx = bid % nWw; bid /= nWw;
    const int wy = bid % nWh;
    const int b = bid / nWh;

    const bool last_y = shift > 0 && wy == nWh - 1, last_x = shift > 0 && wx == nWw - 1;
    for (int t = tid; t < WN; t += NTHR) {
        const int ty = t / WS, tx = t % WS;
        int ys = wy * WS + ty, xs = wx * WS + tx;
        int yo = ys + shift; if (yo >= H) yo -= H;
        int xo = xs + shift; if (xo >= W) xo -= W;
        rowof[t] = (b * H + yo) * W + xo;
        const int reg = (last_y ? (ty < WS - shift ? 1 : 2) : 0) * 3 + (last_x ? (tx < WS - shift ? 1 : 2) : 0);
        kinfo[t] = (ty * (2 * WS - 1) + tx) | (reg << 16);
    }
    for (int i = tid; i < 529; i += NTHR) tab[i] = table[i * heads + head];
    for (int i = tid; i < HD * (VT_STRIDE - WN); i += NTHR)
        Vt[(i / (VT_STRIDE - WN)) * VT_STRIDE + WN + i % (VT_STRIDE - WN)] = (T)0.f;
    __syncthreads();

    const size_t ld = (size_t)3 * C;
    for (int i = tid; i < WN * 4; i += NTHR) {       // 576 16-byte chunks each for K and V
        const int key = i >> 2, g8 = i & 3;
        const T* base = qkv + (size_t)rowof[key] * ld + head * HD + g8 * 8;
        const v8 kv = *(const v8*)(base + C);
        const v8 vv = *(const v8*)(base + 2 * C);
        *(v8*)(Ks + key * KS_STRIDE + g8 * 8) = kv;
#pragma unroll
        for (int j = 0; j < 8; ++j) Vt[(g8 * 8 + j) * VT_STRIDE + key] = vv[j];
    }
    const int fr = lane & 15, fg = lane >> 4;
    v8 qf[QT];
    int qrow[QT];
#pragma unroll
    for (int q = 0; q < QT; ++q) {
        qrow[q] = rowof[(wave * QT + q) * 16 + fr];
        qf[q] = *(const v8*)(qkv + (size_t)qrow[q] * ld + head * HD + fg * 8);
    }
    __syncthreads();

    // ---- S^T[key][query] -------------------------------------------------------------------
    f32x4 acc[QT][9];
#pragma unroll
    for (int kt = 0; kt < 9; ++kt) {
        const v8 kf = *(const v8*)(Ks + (kt * 16 + fr) * KS_STRIDE + fg * 8);
#pragma unroll
        for (int q = 0; q < QT; ++q)
            acc[q][kt] = H16<T>::mfma(kf, qf[q], (f32x4){0.f, 0.f, 0.f, 0.f});
    }

    // ---- scale + relative-position bias + shift mask, softmax over keys ----------------------
    const float scale = 0.17677669529663687f;  // 32^-0.5 (reference scales q before QK^T; same product)
    float inv_sum[QT];
#pragma unroll
    for (int q = 0; q < QT; ++q) {
        const int qinfo = kinfo[(wave * QT + q) * 16 + fr];
        // bias index = (qy-ky+11)*23 + (qx-kx+11) = (qy*23+qx + 11*24) - (ky*23+kx)
        const int qa = (qinfo & 0xffff) + (WS - 1) * (2 * WS);
        const int rq = qinfo >> 16;
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < 9; ++kt) {
            const int4 ki4 = *(const int4*)(kinfo + kt * 16 + fg * 4);
            const int kis[4] = {ki4.x, ki4.y, ki4.z, ki4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = acc[q][kt][r] * scale + tab[qa - (kis[r] & 0xffff)];
                if ((kis[r] >> 16) != rq) s += -100.0f;
                acc[q][kt][r] = s;
                mx = fmaxf(mx, s);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 9; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __expf(acc[q][kt][r] - mx);
                acc[q][kt][r] = p;
                sum += p;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        inv_sum[q] = 1.0f / sum;
    }

    // ---- O^T[d][query] = V^T . P^T over 5 blocks of 32 key slots (last block half zero) -------
    f32x4 oacc[QT][2];
#pragma unroll
    for (int q = 0; q < QT; ++q) oacc[q][0] = oacc[q][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 5; ++m) {
        v8 pf[QT];
#pragma unroll
        for (int q = 0; q < QT; ++q) {
            const f32x4 lo = acc[q][2 * m];
            const f32x4 hi = (m < 4) ? acc[q][m < 4 ? 2 * m + 1 : 8] : (f32x4){0.f, 0.f, 0.f, 0.f};
            pf[q] = (v8){(T)lo[0], (T)lo[1], (T)lo[2], (T)lo[3], (T)hi[0], (T)hi[1], (T)hi[2], (T)hi[3]};
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const T* vrow = Vt + (dt * 16 + fr) * VT_STRIDE + m * 32 + fg * 4;
            const v4 a = *(const v4*)vrow, c = *(const v4*)(vrow + 16);
            const v8 vf = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
#pragma unroll
            for (int q = 0; q < QT; ++q) oacc[q][dt] = H16<T>::mfma(vf, pf[q], oacc[q][dt]);
        }
    }
#pragma unroll
    for (int q = 0; q < QT; ++q)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const f32x4 o = oacc[q][dt] * inv_sum[q];
            v4 o4 = {(T)o[0], (T)o[1], (T)o[2], (T)o[3]};
            *(v4*)(out + (size_t)qrow[q] * C + head * HD + dt * 16 + fg * 4) = o4;
        }
}

// Split-operand form (dtypes BF16X3 / F16X3): q, k, v arrive as hi / lo planes, S = kh.qh + kh.ql + kl.qh and
// O = vh.ph + vh.pl + vl.ph on the same MFMAs (small terms first), the context leaves as hi / lo planes. The
// un-normalised probabilities are scaled by 2^10 before they are split (exp(s - max) <= 1 would put most lo parts into the
// fp16 subnormal range); the scale cancels in O / sum. Same structure and index arithmetic as window_attn_kernel.
// Softmax of the split-operand window attention in the log2 domain (round 5): the score is formed as acc * (scale * log2 e) +
// bias * log2 e (the bias column is multiplied once when it is staged in LDS), the shift mask is -100 log2 e, and a probability is
// v_exp_f32(score - (max - 10)) — the 2^10 that keeps the lo plane of P out of the fp16 subnormals folded into the exponent —
// instead of v_exp_f32((score - max) * log2 e) * 1024: two VALU operations per score fewer (72 of ~625 per item and lane). Both
// split kernels use the same form (their outputs stay bit-identical to each other).
#ifndef MNX_ATTN_EXP2
#define MNX_ATTN_EXP2 1
#endif
constexpr float ATTN_LOG2E = 1.4426950408889634f;

template <typename T>
__global__ __launch_bounds__(576) void window_attn_split_kernel(const T* __restrict__ qkv, size_t qkv_lo,
                                                                const float* __restrict__ table, T* __restrict__ out,
                                                                size_t out_lo, int H, int W, int C, int heads, int shift,
                                                                int terms) {
    constexpr int NTHR = 576;
    typedef typename H16<T>::v8 v8;
    typedef typename H16<T>::v4 v4;
    __shared__ __attribute__((aligned(16))) T Ks[2][WN * KS_STRIDE];
    __shared__ __attribute__((aligned(16))) T Vt[2][HD * VT_STRIDE];
    __shared__ float tab[(2 * WS - 1) * (2 * WS - 1)];
    __shared__ int rowof[WN];
    __shared__ __attribute__((aligned(16))) int kinfo[WN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nWw = W / WS, nWh = H / WS;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int head = bid % heads; bid /= heads;
    const int wx = bid % nWw; bid /= nWw;
    const int wy = bid % nWh;
    const int b = bid / nWh;

    MNX_ATTN_STAMP(0);
    const bool last_y = shift > 0 && wy == nWh - 1, last_x = shift > 0 && wx == nWw - 1;
    for (int t = tid; t < WN; t += NTHR) {
        const int ty = t / WS, tx = t % WS;
        int ys = wy * WS + ty, xs = wx * WS + tx;
        int yo = ys + shift; if (yo >= H) yo -= H;
        int xo = xs + shift; if (xo >= W) xo -= W;
        rowof[t] = (b * H + yo) * W + xo;
        const int reg = (last_y ? (ty < WS - shift ? 1 : 2) : 0) * 3 + (last_x ? (tx < WS - shift ? 1 : 2) : 0);
        kinfo[t] = (ty * (2 * WS - 1) + tx) | (reg << 16);
    }
    for (int i = tid; i < 529; i += NTHR) tab[i] = table[i * heads + head] * (MNX_ATTN_EXP2 ? ATTN_LOG2E : 1.0f);
    for (int i = tid; i < 2 * HD * (VT_STRIDE - WN); i += NTHR) {
        const int pl = i / (HD * (VT_STRIDE - WN)), j = i % (HD * (VT_STRIDE - WN));
        Vt[pl][(j / (VT_STRIDE - WN)) * VT_STRIDE + WN + j % (VT_STRIDE - WN)] = (T)0.f;
    }
    __syncthreads();
    MNX_ATTN_STAMP(1);

    const size_t ld = (size_t)3 * C;
    for (int i = tid; i < 2 * WN * 4; i += NTHR) {       // 576 16-byte chunks each for K and V, two planes
        const int pl = i / (WN * 4), key = (i >> 2) % WN, g8 = i & 3;
        const T* base = qkv + (pl ? qkv_lo : (size_t)0) + (size_t)rowof[key] * ld + head * HD + g8 * 8;
        const v8 kv = *(const v8*)(base + C);
        const v8 vv = *(const v8*)(base + 2 * C);
        *(v8*)(Ks[pl] + key * KS_STRIDE + g8 * 8) = kv;
#pragma unroll
        for (int j = 0; j < 8; ++j) Vt[pl][(g8 * 8 + j) * VT_STRIDE + key] = vv[j];
    }
    const int fr = lane & 15, fg = lane >> 4;
    const int qrow = rowof[wave * 16 + fr];
    const v8 qh = *(const v8*)(qkv + (size_t)qrow * ld + head * HD + fg * 8);
    const v8 ql = *(const v8*)(qkv + qkv_lo + (size_t)qrow * ld + head * HD + fg * 8);
    MNX_ATTN_STAMP(2);
    __syncthreads();
    MNX_ATTN_STAMP(3);

    const bool x3 = terms == 3;
    f32x4 acc[9];
#pragma unroll
    for (int kt = 0; kt < 9; ++kt) {
        const v8 kh = *(const v8*)(Ks[0] + (kt * 16 + fr) * KS_STRIDE + fg * 8);
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        if (x3) {
            const v8 kl = *(const v8*)(Ks[1] + (kt * 16 + fr) * KS_STRIDE + fg * 8);
            a = H16<T>::mfma(kl, qh, a);
            a = H16<T>::mfma(kh, ql, a);
        }
        acc[kt] = H16<T>::mfma(kh, qh, a);
    }

    const float scale = 0.17677669529663687f * (MNX_ATTN_EXP2 ? ATTN_LOG2E : 1.0f);
    const float maskv = -100.0f * (MNX_ATTN_EXP2 ? ATTN_LOG2E : 1.0f);
    MNX_ATTN_STAMP(4);
    const int qinfo = kinfo[wave * 16 + fr];
    const int qa = (qinfo & 0xffff) + (WS - 1) * (2 * WS);
    const int rq = qinfo >> 16;
    float mx = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < 9; ++kt) {
        const int4 ki4 = *(const int4*)(kinfo + kt * 16 + fg * 4);
        const int kis[4] = {ki4.x, ki4.y, ki4.z, ki4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float sc = acc[kt][r] * scale + tab[qa - (kis[r] & 0xffff)];
            if ((kis[r] >> 16) != rq) sc += maskv;
            acc[kt][r] = sc;
            mx = fmaxf(mx, sc);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mxm = mx - 10.0f;
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 9; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float pv = MNX_ATTN_EXP2 ? __builtin_amdgcn_exp2f(acc[kt][r] - mxm) : __expf(acc[kt][r] - mx) * 1024.0f;
            acc[kt][r] = pv;
            sum += pv;
        }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv_sum = 1.0f / sum;
    MNX_ATTN_STAMP(5);

    f32x4 oacc[2];
    oacc[0] = oacc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 5; ++m) {
        const f32x4 p0 = acc[2 * m];
        const f32x4 p1 = (m < 4) ? acc[m < 4 ? 2 * m + 1 : 8] : (f32x4){0.f, 0.f, 0.f, 0.f};
        v4 h0, l0, h1, l1;
        split16x4<T>(p0, h0, l0);
        split16x4<T>(p1, h1, l1);
        const v8 ph = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
        const v8 pl = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const T* vrow = Vt[0] + (dt * 16 + fr) * VT_STRIDE + m * 32 + fg * 4;
            const v4 a = *(const v4*)vrow, c = *(const v4*)(vrow + 16);
            const v8 vh = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
            if (x3) {
                const T* vrl = Vt[1] + (dt * 16 + fr) * VT_STRIDE + m * 32 + fg * 4;
                const v4 al = *(const v4*)vrl, cl = *(const v4*)(vrl + 16);
                const v8 vl = {al[0], al[1], al[2], al[3], cl[0], cl[1], cl[2], cl[3]};
                oacc[dt] = H16<T>::mfma(vl, ph, oacc[dt]);
                oacc[dt] = H16<T>::mfma(vh, pl, oacc[dt]);
            }
            oacc[dt] = H16<T>::mfma(vh, ph, oacc[dt]);
        }
    }
    MNX_ATTN_STAMP(6);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        v4 hi, lo;
        split16x4<T>(oacc[dt] * inv_sum, hi, lo);
        const size_t o = (size_t)qrow * C + head * HD + dt * 16 + fg * 4;
        *(v4*)(out + o) = hi;
        *(v4*)(out + out_lo + o) = lo;
    }
    MNX_ATTN_STAMP(7);
}

// ---------------------------------------------------------------------------------------------
// K4p  persistent, double-buffered form of the split-operand window attention.
// tools/attn_lab (per-workgroup clock stamps) showed where window_attn_split_kernel's time goes: a workgroup lives 13 us, 6 of
// them in three SERIAL memory round trips (bias table -> K / V rows -> q rows) with ~1.5 workgroups resident per CU to
// hide them (nine-wave workgroups do not pack three to a CU). Here a workgroup stays on its CU and walks a contiguous
// range of (window, head) items; while item i is multiplied, item i+1's K and V rows (both planes, 36 KB) arrive by
// LDS-DMA in the other half of a double buffer, its q rows and bias column in registers:
//   * K / V rows sit ROW-major in LDS (64 bytes per key, no register staging, no transposing writes); the V^T operand
//     of O^T = V^T.P^T comes from ds_read_b64_tr_b16 (gfx950's transposing LDS read: the 16 lanes of a group hand in
//     four rows x four 8-byte pieces and receive columns; tools/probes/ds_read_tr_b16.hip prints the mapping). The
//     16-byte pieces of V row r are stored at piece ^ ((r >> 2) & 3) — the DMA lane simply fetches a different global
//     piece — which leaves the transposing read at its 2-way (= optimal, 512 bytes over 64 banks) conflict level.
//   * one wait (vmcnt(0)) and one barrier per item: everything outstanding at the top of item i was issued a whole
//     item earlier (DMA and q of item i; the context STORES of item i-2, which are kept in registers one item longer
//     for exactly this reason).
//   * the shift-mask / relative-position key table has four variants (last row / last column of windows): built once.
// Arithmetic — term order, k-slot assignment, scaling — is window_attn_split_kernel's: results are bit-identical.
// ---------------------------------------------------------------------------------------------
constexpr int WA_ARR = WN * HD * 2;                 // bytes of one of K hi, K lo, V hi, V lo (144 rows x 64 B)
constexpr int WA_BUF = 4 * WA_ARR;                  // one item's K / V
constexpr int WA_TABN = 576;                        // floats per bias column slot (529 used)
constexpr int WA_LDS = 2 * WA_BUF + 2 * WA_TABN * 4 + 4 * WN * 4;   // 80640 B: two workgroups per CU
typedef short s16x4_t __attribute__((__vector_size__(8)));

template <typename T>
__device__ __forceinline__ typename H16<T>::v8 lds_tr8(const lds_char_t* p, int off0, int off1) {
    typedef __attribute__((address_space(3))) s16x4_t lv;
    const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lv*)(p + off0));
    const s16x4_t c = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lv*)(p + off1));
    typedef typename H16<T>::v4 v4;
    const v4 x = __builtin_bit_cast(v4, a), y = __builtin_bit_cast(v4, c);
    return (typename H16<T>::v8){x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
}

// MASKED / cls: the windows of a layer come in two kinds — those whose scores need the shift mask (last row / last column
// of windows of a shifted layer: cls 2, MASKED) and those that do not (cls 0: every window of an unshifted layer, cls 1:
// the inner windows of a shifted one). They are separate launches of separate instantiations, so that the unmasked ones
// (most of stages 1-2, all unshifted layers) do not execute the region compare (4 of ~20 VALU operations per score).
template <typename T, bool MASKED>
__global__ __launch_bounds__(576, 6) void window_attn_pipe_kernel(const T* __restrict__ qkv, size_t qkv_lo,
                                                               const float* __restrict__ table, T* __restrict__ out,
                                                               size_t out_lo, int H, int W, int C, int heads, int shift,
                                                               int cls, int n_items) {
    typedef typename H16<T>::v8 v8;
    typedef typename H16<T>::v4 v4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tab = (float*)(smem + 2 * WA_BUF);               // [2][WA_TABN]
    int* kinfo4 = (int*)(smem + 2 * WA_BUF + 2 * WA_TABN * 4);   // [4][WN]: (ky*23 + kx) | region id << 16
    const lds_char_t* lds = (const lds_char_t*)smem;
    const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(lds_char_t*)smem;

    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int nWw = W / WS, nWh = H / WS;
    for (int i = threadIdx.x; i < 4 * WN; i += 576) {
        const int var = i / WN, t = i % WN, ty = t / WS, tx = t % WS;
        const int reg = ((var & 2) ? (ty < WS - shift ? 1 : 2) : 0) * 3 + ((var & 1) ? (tx < WS - shift ? 1 : 2) : 0);
        int e = (ty * (2 * WS - 1) + tx) | (reg << 16);
        if (!MASKED && var == 1) e = (ty * W + tx) * 6 * C;      // the unmasked kernel's offset tables, see below
        if (!MASKED && var == 2) e = (ty * W + tx) * 2 * C;
        kinfo4[i] = e;
    }
    // Unmasked kernel (round 6): its windows never wrap around the image (cls 0: no shift; cls 1: the inner windows of a shifted
    // layer), so the row of window token t is a per-window base (scalar arithmetic) + ty * W + tx, and every byte offset a lane
    // needs is "scalar base + table[t]": two 144-entry tables — (ty W + tx) 6C, the token's offset in a qkv plane, and
    // (ty W + tx) 2C, in a context plane — in the slots of mask variants 1 and 2, which this instantiation never reads. One LDS
    // read replaces ~30 VALU operations (three of them quarter-rate integer multiplies) per offset, three offsets per item. The
    // tables are complete at the first barrier: the fetches in front of the loop take the arithmetic path.
    // (written by the loop above: every entry of kinfo4 has ONE writer)
    const int* rel_qkv = kinfo4 + WN;
    const int* rel_ctx = kinfo4 + 2 * WN;
    // Register diet: the loop body needs <= 80 VGPRs for two workgroups per CU (a 9-wave workgroup puts 3 waves on
    // SIMD 0, two of them 6: 512 / 6). Everything derived from the lane id (fragment coordinates, window coordinates of
    // the rows a lane fetches, LDS addresses) is therefore RE-derived at each use from an opaque copy of threadIdx.x —
    // a handful of VALU operations per item — instead of living in registers across the whole item (the compiler hoists
    // such values out of the loop and then spills them; a scratch reload inside the loop drains the fetch queue).
    auto thread_id = [&]() {
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));
        return t;
    };
    const char* plane0 = (const char*)qkv;
    const char* plane1 = (const char*)(qkv + qkv_lo);

    const int it0 = (int)((long long)blockIdx.x * n_items / gridDim.x);
    const int it1 = (int)((long long)(blockIdx.x + 1) * n_items / gridDim.x);
    if (it0 >= it1) return;

    // the item being fetched (suffix _n): window, head (uniform), then its K / V by DMA and its q rows / bias column
    // into registers; the item whose context waits to be stored (suffix _p)
    struct Item { int head, b, wy, wx; };            // uniform (scalar registers)
    auto locate = [=](int it) {
        Item x;
        x.head = it % heads;
        int wi = it / heads;
        if (cls == 0) {                              // all windows, row-major
            x.wx = wi % nWw; wi /= nWw;
            x.wy = wi % nWh;
            x.b = wi / nWh;
        } else if (cls == 1) {                       // inner windows: wy < nWh - 1, wx < nWw - 1
            x.wx = wi % (nWw - 1); wi /= nWw - 1;
            x.wy = wi % (nWh - 1);
            x.b = wi / (nWh - 1);
        } else {                                     // border: the last row of windows (nWw of them), then the last column
            const int j = wi % (nWh + nWw - 1);
            x.b = wi / (nWh + nWw - 1);
            x.wy = j < nWw ? nWh - 1 : j - nWw;
            x.wx = j < nWw ? j : nWw - 1;
        }
        return x;
    };
    auto token_row = [=](const Item& w, int key) {  // row of window token `key` in the [B*H*W, .] buffers (shift folded in)
        const int ty = (key * 171) >> 11, tx = key - ty * WS;     // key / 12, key % 12 for key < 144
        int y = w.wy * WS + ty + shift; if (y >= H) y -= H;
        int x = w.wx * WS + tx + shift; if (x >= W) x -= W;
        return (unsigned)((w.b * H + y) * W + x);
    };
    // byte offset of (token `key`, head of the item) in a qkv plane / in a context plane (< 2^32)
    auto qkv_off = [=](const Item& w, int key, bool tab) {
        if (!MASKED && tab)
            return (unsigned)(((w.b * H + w.wy * WS + shift) * W + w.wx * WS + shift) * 6 * C + w.head * HD * 2) + (unsigned)rel_qkv[key];
        return (token_row(w, key) * 3u * (unsigned)C + (unsigned)(w.head * HD)) * 2u;
    };
    auto ctx_off = [=](const Item& w, int key, bool tab) {
        if (!MASKED && tab)
            return (unsigned)(((w.b * H + w.wy * WS + shift) * W + w.wx * WS + shift) * 2 * C + w.head * HD * 2) + (unsigned)rel_ctx[key];
        return (token_row(w, key) * (unsigned)C + (unsigned)(w.head * HD)) * 2u;
    };
    // DMA: wave w moves key rows 16w..16w+15, lane l row 16w + l/4, LDS piece l%4 (K: global piece l%4; V: global
    // piece (l%4) ^ ((l >> 4) & 3), see above)
    auto fetch_kv = [&](const Item& w, int buf, bool tab) {
        const int lane = thread_id() & 63;
        const unsigned kb = qkv_off(w, wave * 16 + (lane >> 2), tab);
        const unsigned offK = kb + (unsigned)C * 2u + (unsigned)(lane & 3) * 16u;
        const unsigned offV = kb + (unsigned)C * 4u + (unsigned)((lane & 3) ^ ((lane >> 4) & 3)) * 16u;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * WA_BUF + wave * 1024));
        dma16(offK, plane0, dst);
        dma16(offK, plane1, dst + WA_ARR);
        dma16(offV, plane0, dst + 2 * WA_ARR);
        dma16(offV, plane1, dst + 3 * WA_ARR);
    };
    // (all global accesses of the loop are "uniform base + 32-bit lane offset": no 64-bit address registers)
    const char* outp0 = (const char*)out;
    const char* outp1 = (const char*)(out + out_lo);
    v8 qh_n, ql_n;
    float tab_n;
    int var_n;
    unsigned ooff_n;                                // bytes into a context plane; < 2^32 (M*C*2 <= 5.3e8)
    auto fetch_q = [&](const Item& w, bool tab) {   // MFMA role: query 16w + fr, channel group fg
        const int tid = thread_id(), fr = tid & 15, fg = (tid >> 4) & 3;
        const unsigned qb = qkv_off(w, wave * 16 + fr, tab) + (unsigned)(fg * 16);
        qh_n = *(const v8*)(plane0 + qb);
        ql_n = *(const v8*)(plane1 + qb);
        tab_n = tid < 529 ? *(const float*)((const char*)table + (unsigned)(tid * heads + w.head) * 4u) : 0.f;
        var_n = MASKED ? (w.wy == nWh - 1 ? 2 : 0) + (w.wx == nWw - 1 ? 1 : 0) : 0;
        ooff_n = ctx_off(w, wave * 16 + fr, tab) + (unsigned)(fg * 8);
    };

    v4 ohi_p[2], olo_p[2];
    unsigned ooff_p = 0;
    bool have_p = false;
    int cur = 0;
    Item nx = locate(it0);
    fetch_kv(nx, 0, false);
    fetch_q(nx, false);
#pragma clang loop unroll(disable)                   // (also keeps the compiler from peeling the first item off: one copy of the body)
    for (int it = it0; it < it1; ++it) {
        // item `it`: its K / V are in LDS, its q rows / bias column in registers. The prefetched registers pass THROUGH the
        // wait so that the compiler takes them as complete here and adds no vmcnt wait of its own later in the item
        // (it cannot see the LDS-DMA in the queue: any wait it adds drains the next item's fetch as well)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(qh_n), "+v"(ql_n), "+v"(tab_n) : : "memory");
        const v8 qh = qh_n, ql = ql_n;
        const unsigned ooff = ooff_n;
        const int* kinfo = kinfo4 + var_n * WN;
        float* tabc = tab + cur * WA_TABN;
        const int tid = thread_id(), fr = tid & 15, fg = (tid >> 4) & 3;
        if (tid < 529) tabc[tid] = tab_n * (MNX_ATTN_EXP2 ? ATTN_LOG2E : 1.0f);
        __syncthreads();                                      // every wave's DMA landed; buffer cur^1 is free (item it-1 is done)
        const bool more = it + 1 < it1;
        if (more) {
            nx = locate(it + 1);
            fetch_kv(nx, cur ^ 1, true);
        }
        if (have_p) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                *(v4*)(outp0 + ooff_p + dt * 32) = ohi_p[dt];
                *(v4*)(outp1 + ooff_p + dt * 32) = olo_p[dt];
            }
        }
        const char* kv = smem + cur * WA_BUF;

        // ---- S^T[key][query], small terms first -------------------------------------------------
        f32x4 acc[9];
#pragma unroll
        for (int kt = 0; kt < 9; ++kt) {
            const v8 kh = *(const v8*)(kv + (kt * 16 + fr) * 64 + fg * 16);
            const v8 kl = *(const v8*)(kv + WA_ARR + (kt * 16 + fr) * 64 + fg * 16);
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            a = H16<T>::mfma(kl, qh, a);
            a = H16<T>::mfma(kh, ql, a);
            acc[kt] = H16<T>::mfma(kh, qh, a);
        }

        // ---- scale + relative-position bias + shift mask, softmax over keys ----------------------
        // bias index = (qy-ky+11)*23 + (qx-kx+11) = (qy*23+qx + 11*24) - (ky*23+kx). A lane's four keys of a tile
        // (16 kt + 4 fg + r) are consecutive tokens of one window row (12 % 4 == 0), so their four bias values are
        // CONSECUTIVE table entries, descending: one index per tile, two paired LDS reads instead of four gathers.
        const float scale = 0.17677669529663687f * (MNX_ATTN_EXP2 ? ATTN_LOG2E : 1.0f);
        const float maskv = -100.0f * (MNX_ATTN_EXP2 ? ATTN_LOG2E : 1.0f);
        const int qinfo = kinfo[wave * 16 + fr];
        const int qa = (qinfo & 0xffff) + (WS - 1) * (2 * WS) - 3;
        const int rq = qinfo >> 16;
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < 9; ++kt) {
            int kis[4];
            if (MASKED) {
                const int4 ki4 = *(const int4*)(kinfo + kt * 16 + fg * 4);
                kis[0] = ki4.x; kis[1] = ki4.y; kis[2] = ki4.z; kis[3] = ki4.w;
            } else {
                kis[0] = kinfo[kt * 16 + fg * 4];        // region ids are all 0: the entry IS ky*23 + kx
            }
            const float* tp = tabc + (qa - (kis[0] & 0xffff));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float sc = acc[kt][r] * scale + tp[3 - r];
                if (MASKED && (kis[r] >> 16) != rq) sc += maskv;
                acc[kt][r] = sc;
                mx = fmaxf(mx, sc);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mxm = mx - 10.0f;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 9; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = MNX_ATTN_EXP2 ? __builtin_amdgcn_exp2f(acc[kt][r] - mxm) : __expf(acc[kt][r] - mx) * 1024.0f;
                acc[kt][r] = pv;
                sum += pv;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv_sum = 1.0f / sum;

        // ---- O^T[d][query] = V^T . P^T over 5 blocks of 32 key slots -------------------------------
        // k-slot j of lane group fg <-> key 32m + 4fg + j (j < 4) and 32m + 16 + 4fg + (j - 4): the transposing read
        // of rows 32m (+16) + 4fg .. +3 delivers exactly these for channel d = 16dt + fr. Block 4 has no second half
        // (keys 144..159): its probabilities are zero, its V slots re-read the first half (finite values).
        f32x4 oacc[2];
        oacc[0] = oacc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int vrow = (4 * fg + (fr >> 2)) * 64 + (fr & 1) * 8;
        const int piece0 = (((fr >> 1) & 1) ^ fg) * 16;
        const lds_char_t* vb = lds + cur * WA_BUF + 2 * WA_ARR + vrow;
#pragma unroll
        for (int m = 0; m < 5; ++m) {
            const f32x4 p0 = acc[2 * m];
            const f32x4 p1 = (m < 4) ? acc[m < 4 ? 2 * m + 1 : 8] : (f32x4){0.f, 0.f, 0.f, 0.f};
            v4 h0, l0, h1, l1;
            if constexpr (MASKED) {                  // (the masked kernel has no register to spare for the asm form: it spills)
                split16x4<T>(p0, h0, l0);
                split16x4<T>(p1, h1, l1);
            } else {
                split16x4_mix<T>(p0, h0, l0);
                split16x4_mix<T>(p1, h1, l1);
            }
            const v8 ph = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
            const v8 pl = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
            const int r0 = m * 2048, r1 = m < 4 ? m * 2048 + 1024 : m * 2048;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const lds_char_t* vp = vb + (dt ? (piece0 ^ 32) : piece0);
                const v8 vl = lds_tr8<T>(vp + WA_ARR, r0, r1);
                const v8 vh = lds_tr8<T>(vp, r0, r1);
                oacc[dt] = H16<T>::mfma(vl, ph, oacc[dt]);
                oacc[dt] = H16<T>::mfma(vh, pl, oacc[dt]);
                oacc[dt] = H16<T>::mfma(vh, ph, oacc[dt]);
            }
            // the next item's q rows / bias column: issued here, not with the DMA, because only now (the first 64 keys'
            // probabilities are consumed) are there registers for them; the rest of the item covers the latency
            if (m == 1 && more) fetch_q(nx, true);
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            if constexpr (MASKED) split16x4<T>(oacc[dt] * inv_sum, ohi_p[dt], olo_p[dt]);
            else split16x4_mix<T>(oacc[dt] * inv_sum, ohi_p[dt], olo_p[dt]);
        }
        ooff_p = ooff;
        have_p = true;
        cur ^= 1;
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        *(v4*)(outp0 + ooff_p + dt * 32) = ohi_p[dt];
        *(v4*)(outp1 + ooff_p + dt * 32) = olo_p[dt];
    }
}

// > 64 KB of dynamic LDS needs the per-function opt-in, once per device (bit d of `done`: device d has it)
template <auto Kern>      // keyed on the kernel value: one flag per instantiation (see gemm256.hip)
static hipError_t attn_lds_opt_in() {
    static unsigned long long done = 0;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && (__atomic_load_n(&done, __ATOMIC_ACQUIRE) >> dev & 1ull)) return hipSuccess;
    e = hipFuncSetAttribute((const void*)Kern, hipFuncAttributeMaxDynamicSharedMemorySize, WA_LDS);
    if (e == hipSuccess && dev >= 0 && dev < 64) __atomic_fetch_or(&done, 1ull << dev, __ATOMIC_RELEASE);
    return e;
}

hipError_t launch_window_attn(int dtype, const void* qkv16, const float* rel_table, void* out16, int B, int H, int W,
                              int C, int heads, int shift, hipStream_t s, size_t qkv_lo, size_t out_lo, int terms) {
    if (C != heads * HD || H % WS || W % WS) return hipErrorInvalidValue;
    dim3 grid(B * (H / WS) * (W / WS) * heads);
    if (dt_split(dtype)) {
        if (qkv_lo == 0 || out_lo == 0 || (terms != 1 && terms != 3)) return hipErrorInvalidValue;
        if (terms == 3) {
            // persistent form: two 576-thread workgroups per CU (80.6 KB of LDS each), contiguous item ranges; one launch
            // for the windows without shift mask, one for the border windows of a shifted layer
            const int nWh = H / WS, nWw = W / WS;
            const int n_plain = shift > 0 ? B * (nWh - 1) * (nWw - 1) * heads : (int)grid.x;
            const int n_border = shift > 0 ? B * (nWh + nWw - 1) * heads : 0;
#define MNX_ATTN_PIPE(TT, MASKED, CLS, NITEMS)                                                                          \
    do {                                                                                                                \
        hipError_t e_ = attn_lds_opt_in<window_attn_pipe_kernel<TT, MASKED>>();                                           \
        if (e_ != hipSuccess) return e_;                                                                                \
        hipLaunchKernelGGL((window_attn_pipe_kernel<TT, MASKED>), dim3((NITEMS) < 2 * persistent_cus() ? (NITEMS) : 2 * persistent_cus()), dim3(576), WA_LDS, s,  \
                           (const TT*)qkv16, qkv_lo, rel_table, (TT*)out16, out_lo, H, W, C, heads, shift, CLS, NITEMS); \
    } while (0)
            if (dtype == MNX_DT_F16X3) {
                if (n_plain > 0) MNX_ATTN_PIPE(f16_t, false, shift > 0 ? 1 : 0, n_plain);
                if (n_border > 0) MNX_ATTN_PIPE(f16_t, true, 2, n_border);
            } else {
                if (n_plain > 0) MNX_ATTN_PIPE(bf16_t, false, shift > 0 ? 1 : 0, n_plain);
                if (n_border > 0) MNX_ATTN_PIPE(bf16_t, true, 2, n_border);
            }
#undef MNX_ATTN_PIPE
            return hipGetLastError();
        }
        if (dtype == MNX_DT_F16X3)
            hipLaunchKernelGGL((window_attn_split_kernel<f16_t>), grid, dim3(576), 0, s, (const f16_t*)qkv16, qkv_lo, rel_table,
                               (f16_t*)out16, out_lo, H, W, C, heads, shift, terms);
        else
            hipLaunchKernelGGL((window_attn_split_kernel<bf16_t>), grid, dim3(576), 0, s, (const bf16_t*)qkv16, qkv_lo, rel_table,
                               (bf16_t*)out16, out_lo, H, W, C, heads, shift, terms);
        return hipGetLastError();
    }
#define MNX_ATTN(TT)                                                                                                  \
    hipLaunchKernelGGL((window_attn_kernel<TT>), grid, dim3(576), 0, s, (const TT*)qkv16, rel_table, (TT*)out16, H, W, \
                       C, heads, shift)
    if (dtype == MNX_DT_F16) MNX_ATTN(f16_t);
    else if (dtype == MNX_DT_F32) MNX_ATTN(float);
    else MNX_ATTN(bf16_t);
#undef MNX_ATTN
    return hipGetLastError();
}

}  // namespace mnx
