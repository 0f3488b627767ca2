// decoder.hip — autoregressive decoder step, greedy pick, cross-KV/enc_transform SGEMM, bond head.
//
// Everything here is fp32 (the decoder is ~1 % of the path's FLOPs but decides every token, so it keeps the
// reference's precision): SURVEY K6-K12.
//
// One decode step = 8 small kernels per layer + 1 "head" kernel, all reading the step index from device memory
// so the whole step is a fixed hipGraph that the host replays (engine.hip). Batch rows are fixed slots; a
// finished row keeps its slot (no compaction on device) and the reference's row renumbering is emulated only
// where it is observable: the positional-encoding row (see embed prologue).
#include <stdlib.h>

#include "common.h"
#include "kernels.h"
#include "dec_types.h"
#include "kvq.h"

namespace mnx {

// =============================================================================================
// fp32 SGEMM  C[M,N] = A[M,K] . W[N,K]^T + bias     (enc_transform: reference components.py:206-216;
// cross-attention K/V projection, once per image: onmt MultiHeadedAttention 'context' cache, driven by
// reference models/decoder.py:269-276; bond-head first Linear split in two halves: components.py:355-357)
// 64x64 tile, 16-deep K slices, 4x4 outputs per thread.
// =============================================================================================
// perm_S > 0: C is stored as [M / perm_S][N / 256][8][perm_S][32] instead of [M][N] — the cross-attention K/V layout
// [image][layer][K|V][head][position][32]: the 144 key rows that one (sequence, head) workgroup of a decode step
// reads are one contiguous 18 KB stream (with [position][layer*512] they were 128-byte lines 12 KB apart: a new
// DRAM page per key; measured +3 % end to end for [position][256]).
__global__ __launch_bounds__(256) void sgemm_tn_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                       const float* __restrict__ bias, float* __restrict__ C, int M,
                                                       int N, int K, int perm_S) {
    __shared__ float As[16][68], Ws[16][68];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int lr = tid >> 2, lk = (tid & 3) * 4;
    const float* ap = A + (size_t)min(m0 + lr, M - 1) * K + lk;
    const float* wp = W + (size_t)min(n0 + lr, N - 1) * K + lk;
    float acc[4][4] = {};
    f32x4 a = *(const f32x4*)ap, w = *(const f32x4*)wp;
    for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            As[lk + j][lr] = a[j];
            Ws[lk + j][lr] = w[j];
        }
        __syncthreads();
        const int kn = k0 + 16 < K ? k0 + 16 : k0;       // the next slice travels under this slice's 256 FMAs
        a = *(const f32x4*)(ap + kn);
        w = *(const f32x4*)(wp + kn);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const f32x4 av = *(const f32x4*)&As[k][ty * 4], wv = *(const f32x4*)&Ws[k][tx * 4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], wv[j], acc[i][j]);
        }
        __syncthreads();
    }
    const int n = n0 + tx * 4;
    if (n >= N) return;
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (bias) b4 = *(const f32x4*)(bias + n);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty * 4 + i;
        if (m < M) {
            const size_t off = perm_S > 0 ? ((((size_t)(m / perm_S) * (N >> 8) + (n >> 8)) * 8 + ((n & 255) >> 5)) * perm_S +
                                             (m % perm_S)) * 32 + (n & 31)
                                          : (size_t)m * N + n;
            *(f32x4*)(C + off) = (f32x4){acc[i][0], acc[i][1], acc[i][2], acc[i][3]} + b4;
        }
    }
}

hipError_t launch_sgemm_tn(const float* A, const float* W, const float* bias, float* C, int M, int N, int K,
                           hipStream_t s, int perm_S) {
    if ((K & 15) || (N & 3) || (perm_S > 0 && ((N & 255) || M % perm_S))) return hipErrorInvalidValue;
    dim3 grid((N + 63) / 64, (M + 63) / 64), block(256);
    hipLaunchKernelGGL(sgemm_tn_kernel, grid, block, 0, s, A, W, bias, C, M, N, K, perm_S);
    return hipGetLastError();
}

// The projected memory K / V of an admission (sgemm_tn_kernel's perm_S output: fp32 rows of 32 channels, [image][layer][K|V][head]
// [position][32]) -> the quantised blocks the decode ticks read (kvq.h), once per image. 8 lanes per row, 16 bytes each.
__global__ __launch_bounds__(256) void kvq_pack_kernel(const float* __restrict__ src, char* __restrict__ dst, long long n_rows,
                                                       int S, int Sq) {
    const long long r = (long long)blockIdx.x * 32 + (threadIdx.x >> 3);
    const int part = threadIdx.x & 7;
    if (r >= n_rows) return;            // (whole 8-lane groups leave together)
    const f32x4 v = *(const f32x4*)(src + r * 32 + part * 4);
    float amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
    int q[4];
    float scale = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) kvq_quant(v[u], amax, q[u], scale);
    const long long blk_i = r / S;
    const int key = (int)(r - blk_i * S);
    char* blk = dst + (size_t)blk_i * kvq_block_bytes(Sq);
    kvq_store4(blk, Sq, key, part * 4, q);
    if (part == 0) kvq_store_scale(blk, Sq, key, scale);
}

hipError_t kvq_pack_enqueue(const float* src, char* dst, int n_blocks, int S, int Sq, hipStream_t s) {
    const long long n_rows = (long long)n_blocks * S;
    hipLaunchKernelGGL(kvq_pack_kernel, dim3((unsigned)((n_rows + 31) / 32)), dim3(256), 0, s, src, dst, n_rows, S, Sq);
    return hipGetLastError();
}

// =============================================================================================
// Decode tick. Rows are SLOTS (dec_types.h): up to 256 sequences resident at once, each at its own position t.
// One tick advances every alive slot by one token:
//   dec_begin_kernel                  PE ranks of every slot, alive counters
//   per layer (8 kernels)             dec_linear<LN1|embed, qkv>, dec_attn(self), dec_linear<wo,+res>,
//                                     dec_linear<LN2, q>, dec_attn(cross), dec_linear<wo2,+res>,
//                                     dec_linear<LN, w1, GELU>, dec_linear<w2,+res>
//   dec_head_kernel                   final LN, logits, log-softmax, grammar mask, argmax, bookkeeping
// All kernels read positions / alive flags from device memory, so one hipGraph per slot count replays forever.
// =============================================================================================

// Skinny linear: out[r, n] = epi( pro(in)[r, :] . W[n, :] + b[n] ) for the 32 slots of row tile blockIdx.y.
//   One workgroup = 8 output columns x 32 rows; the [32, 256] input slab is staged in LDS (layer-normed in
//   registers on the way when PRO says so — every workgroup recomputes the tiny LN rather than paying a kernel
//   boundary). Thread (r = tid & 31, c = tid >> 5).
// PRO: 0 plain input | 1 LayerNorm(gamma, beta, eps 1e-6) | 2 token embedding + row-PE, then LayerNorm
// EPI: 0 q/k/v scatter (q scaled, k/v appended to the self cache at position t[slot])
//      1 x[r, n] += result (residual, in place)   2 q-scale store   3 GELU store
//      4 split-K partial: workgroup z = blockIdx.z multiplies k in [256 z, 256 z + 256) only and stores its partial sums
//        (bias in slice 0) to out[z][r, n]; nothing is added here. Used for w_2 (K = 1024), the one linear that was twice as
//        slow as the others in a small tick (16 workgroups x 256 KB): the K / 256 slices are summed, in a fixed order
//        (((p0 + p1) + p2) + p3) + x, by the kernel that reads the residual stream next (PRO 1 of the following layer, which
//        writes the summed stream to the OTHER stream buffer — its 24 column blocks all read the old one —, or the head).
constexpr int TN = 32, XS = 260;

struct LinArgs {
    const float* in;      // [slots, K]      (PRO 2: unused)
    const float* W;       // [N, K]
    const float* bias;    // [N]
    const float* gamma;   // LN weight / bias (PRO 1, 2)
    const float* beta;
    float* out;           // EPI 1: x [slots, N] in place; EPI 2/3: [slots, N]; EPI 0: q buffer [slots, 256]
    char* kcache;         // EPI 0: this layer's self K cache: (slot, head) blocks of Tq rows (kvq.h)
    char* vcache;
    int Tq;               // rows per block of the self cache = kvq_rows(T)
    float* x_write;       // PRO 2: residual stream x [slots, 256] written by column-block 0
    const float* emb;     // PRO 2: [V, 256]
    const float* pe;      // PRO 2: [pe_len, 256]
    const float* part;    // PRO 1: K-slice partial sums [W2_SLICES][slots, 256] of the previous layer's w_2, to be added to `in`
                          //        (then x_write receives the summed stream from column-block 0); EPI 4: unused (out = the partials)
    const DecState* st;
    int N, K, T, heads;
    int part_stride;      // floats between two K-slices of `part` (slots * 256)
    int n_part;           // number of K slices (K / 256 of the linear that wrote them)
    int row_base;         // first row of the tick branch this launch belongs to (dec_enqueue_tick_rows)
};

template <int PRO, int EPI>
__global__ __launch_bounds__(256) void dec_linear_kernel(LinArgs a) {
    // fp32 MFMA (v_mfma_f32_16x16x4_f32: exact fp32 FMA chain) on a 32-row x 32-column tile; the 4 waves split K.
    // Latency is what this kernel costs (5-10 us at any row count, a tick is 50 of them): every load that does not
    // depend on the tick's state (weights, the input slab, the row view) is issued before anything is waited for, there
    // is no early exit to hide loads behind (an idle tile computes on zeros and stores nothing), and for K > 256 the
    // next K chunk is in flight while the current one is multiplied.
    __shared__ __attribute__((aligned(16))) float xs[32 * XS];   // input slab; reused as the cross-wave reduction buffer
    __shared__ __attribute__((aligned(16))) float ws[TN * XS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * TN;
    const int row0 = a.row_base + blockIdx.y * ROW_TILE;   // rows are positions in the tick's compact active list
    // LayerNorm / embedding / staging thread mapping: 8 threads per row, each owns 8 float4 (channels part*4 + 32*j)
    const int lrow = tid >> 3, part = tid & 7;
    f32x4 xv[8], wv[8];
    const int kbeg = EPI == 4 ? (int)blockIdx.z * 256 : 0;             // EPI 4: this workgroup's K slice
    const int kend = EPI == 4 ? 256 : a.K;                               // (relative to kbeg)
    const float* wsrc = a.W + (size_t)(n0 + lrow) * a.K + part * 4 + kbeg;
#pragma unroll
    for (int j = 0; j < 8; ++j) wv[j] = *(const f32x4*)(wsrc + 32 * j);
    const int4 rv = a.st->rowv[row0 + lrow];         // {slot, t, prev_tok, rank}; dummy beyond n_active
    const float* src = a.in + (size_t)(row0 + lrow) * a.K + part * 4 + kbeg;
    if (PRO != 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[j] = *(const f32x4*)(src + 32 * j);
    }
    const int n_act = a.st->n_active;
    const f32x4 bias4 = *(const f32x4*)(a.bias + n0 + part * 4);
    f32x4 res4 = {0.f, 0.f, 0.f, 0.f};               // EPI 1: the residual elements this thread updates in place
    if (EPI == 1) res4 = *(const f32x4*)(a.out + (size_t)(row0 + lrow) * a.N + n0 + part * 4);
    if (PRO == 2) {
        // x0 = E[tok] * sqrt(256) + pe[rank]   (reference components.py:290, embedding.py:52-59)
        const float* e = a.emb + (size_t)rv.z * 256 + part * 4;
        const float* p = a.pe + (size_t)rv.w * 256 + part * 4;
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[j] = *(const f32x4*)(e + 32 * j) * 16.0f + *(const f32x4*)(p + 32 * j);
    }
    if (PRO == 1 && a.part) {    // the previous layer's w_2, summed here (fixed order), residual stream included
        const float* pp = a.part + (size_t)(row0 + lrow) * 256 + part * 4;
        const size_t ps = (size_t)a.part_stride;
        f32x4 sum[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) sum[j] = *(const f32x4*)(pp + 32 * j);
        for (int z = 1; z < a.n_part; ++z) {
#pragma unroll
            for (int j = 0; j < 8; ++j) sum[j] += *(const f32x4*)(pp + z * ps + 32 * j);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[j] = sum[j] + xv[j];
    }
    const bool live = row0 + lrow < n_act;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) acc[i][0] = acc[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fg = lane >> 4;
    for (int k0 = 0; k0 < kend; k0 += 256) {
        // ---- xv / wv hold the input slab [32, 256] and the weight tile [32, 256] of this K chunk ----
        if (!live) {
#pragma unroll
            for (int j = 0; j < 8; ++j) xv[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if ((PRO == 2 || (PRO == 1 && a.part)) && blockIdx.x == 0 && live) {
#pragma unroll
            for (int j = 0; j < 8; ++j) *(f32x4*)(a.x_write + (size_t)(row0 + lrow) * 256 + part * 4 + 32 * j) = xv[j];
        }
        if (PRO != 0) {  // LayerNorm in registers: 8 lanes per row, two-pass (K == 256)
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += xv[j][0] + xv[j][1] + xv[j][2] + xv[j][3];
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
            const float mean = s * (1.0f / 256.0f);
            float sq = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                xv[j] -= mean;
                sq += xv[j][0] * xv[j][0] + xv[j][1] * xv[j][1] + xv[j][2] * xv[j][2] + xv[j][3] * xv[j][3];
            }
            sq += __shfl_xor(sq, 1, 64); sq += __shfl_xor(sq, 2, 64); sq += __shfl_xor(sq, 4, 64);
            const float rstd = rsqrtf(sq * (1.0f / 256.0f) + 1e-6f);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                xv[j] = xv[j] * rstd * *(const f32x4*)(a.gamma + part * 4 + 32 * j) +
                        *(const f32x4*)(a.beta + part * 4 + 32 * j);
        }
        if (k0 > 0) __syncthreads();                 // previous chunk's fragment reads are done
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            *(f32x4*)(xs + lrow * XS + part * 4 + 32 * j) = xv[j];
            *(f32x4*)(ws + lrow * XS + part * 4 + 32 * j) = wv[j];
        }
        if (k0 + 256 < kend) {                       // next K chunk: in flight during this chunk's MFMAs
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                wv[j] = *(const f32x4*)(wsrc + k0 + 256 + 32 * j);
                xv[j] = *(const f32x4*)(src + k0 + 256 + 32 * j);
            }
        }
        __syncthreads();
        // wave w owns k in [64w, 64w+64): per 16-k chunk one ds_read_b128 per operand tile feeds 4 MFMA k-steps
        // (k-slot (step j, lane group g) <-> k = kb + 4g + j on BOTH operands, so the contraction is unchanged)
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            const int kb = wave * 64 + kc * 16 + fg * 4;
            const f32x4 a0 = *(const f32x4*)(xs + (fr) * XS + kb), a1 = *(const f32x4*)(xs + (16 + fr) * XS + kb);
            const f32x4 b0 = *(const f32x4*)(ws + (fr) * XS + kb), b1 = *(const f32x4*)(ws + (16 + fr) * XS + kb);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], b1[j], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], b0[j], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
            }
        }
    }
    __syncthreads();
    // cross-wave reduction through LDS: red[wave][row][col], D layout: lane holds rows fg*4+r, column fr
    float* red = xs;                                  // 4 * 32 * 33 floats = 16.9 KB <= 33 KB
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                red[(wave * 32 + mt * 16 + fg * 4 + r) * 33 + nt * 16 + fr] = acc[mt][nt][r];
    __syncthreads();
    const int slot = rv.x;                            // persistent state (caches) is per slot,
    const int row = row0 + lrow;                      // activations of this tick are per active-list row
    if (!live) return;
    const int nc = part * 4;                          // 4 consecutive output columns per thread
    f32x4 v;
#pragma unroll
    for (int u = 0; u < 4; ++u)
        v[u] = (red[(0 * 32 + lrow) * 33 + nc + u] + red[(1 * 32 + lrow) * 33 + nc + u]) +
               (red[(2 * 32 + lrow) * 33 + nc + u] + red[(3 * 32 + lrow) * 33 + nc + u]);
    const int n = n0 + nc;
    if (EPI != 4 || blockIdx.z == 0) v += bias4;
    if (EPI == 4) {
        *(f32x4*)(a.out + (size_t)blockIdx.z * a.part_stride + (size_t)row * a.N + n) = v;
    } else if (EPI == 0) {
        const int part_ = n >> 8, ch = n & 255, hd = ch >> 5, d = ch & 31;
        if (part_ == 0) {
            *(f32x4*)(a.out + (size_t)row * 256 + ch) = v * 0.17677669529663687f;   // q / sqrt(32) before QK^T (onmt MHA)
        } else {
            // this tile's 32 columns are the 32 channels of ONE head's key (or value) row: the 8 lanes of the row (part 0..7,
            // consecutive lanes) agree on its max and append it to the slot's cache as 24-bit block fixed point (kvq.h)
            float amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
            amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
            amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
            amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
            int qv[4];
            float scale = 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) kvq_quant(v[u], amax, qv[u], scale);
            char* blk = (part_ == 1 ? a.kcache : a.vcache) + ((size_t)slot * a.heads + hd) * kvq_block_bytes(a.Tq);
            kvq_store4(blk, a.Tq, rv.y, d, qv);
            if (part == 0) kvq_store_scale(blk, a.Tq, rv.y, scale);
        }
    } else if (EPI == 1) {
        *(f32x4*)(a.out + (size_t)row * a.N + n) = res4 + v;
    } else if (EPI == 2) {
        *(f32x4*)(a.out + (size_t)row * a.N + n) = v * 0.17677669529663687f;
    } else {
        v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]);
        *(f32x4*)(a.out + (size_t)row * a.N + n) = v;
    }
}

// =============================================================================================
// Single-query attention, 4 waves per (sequence, head): softmax(q.K^T) . V (fp32 throughout, as onmt:
// scores.float(), no mask for a single query position). Self: keys 0..t[slot] of the slot's cache (32 floats
// apart). Cross: the 144 projected memory rows of the slot's memory block.
// (A per-sequence form that fuses attention + final_linear + LayerNorm + next query into one workgroup per row —
//  3 launches per layer instead of 6 — was built and measured in round 2: slower at every row count, because one
//  workgroup then streams 256-512 KB of fp32 weights through one CU)
// =============================================================================================
struct AttnArgs {
    const float* q;      // [slots, 256] pre-scaled
    const char* K;       // base of this layer's keys: blocks of nkb rows (kvq.h), block (owner, head) at owner * row_stride +
    const char* V;       // head * head_stride BYTES; owner = the slot (self), the memory block (cross), an ancestor's slot (ANC)
    float* ctx;          // [slots, 256]
    const DecState* st;
    long long row_stride, head_stride;   // bytes
    int nkb, fixed_keys, heads, cross;
    const int* anc;      // ANC: [slots, anc_stride] slot that holds key tau of the hypothesis (beam search)
    int anc_stride;
    int row_base;        // first row of the tick branch
};

template <bool ANC>
__global__ __launch_bounds__(256) void dec_attn_kernel(AttnArgs a) {
    // 4 waves per (slot, head): the keys are split over all 256 lanes for the scores (one 128-byte key row per
    // lane, all loads in flight at once) and over the 4 waves for P.V; partial results meet in LDS.
    __shared__ float ps[512];
    __shared__ float red[8];
    __shared__ __attribute__((aligned(16))) float po[4][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = a.row_base + blockIdx.x / a.heads, hd = blockIdx.x % a.heads;
    // no n_active here: an idle row carries the dummy row view (slot 0 at position 0, memory block 0), computes a
    // throw-away context row and touches no per-slot state — one dependent round trip less before the key loads
    const int4 rv = a.st->rowv[row];
    const int slot = rv.x;
    const int nkeys = a.cross ? a.fixed_keys : rv.y + 1;
    const long long rowb = a.cross ? (long long)a.st->row_mem[row] : (long long)slot;
    const char* Kb = a.K + rowb * a.row_stride + hd * a.head_stride;
    const char* Vb = a.V + rowb * a.row_stride + hd * a.head_stride;
    // P.V: wave w takes keys w*8 + kg + 32*i (kg = lane>>3), channel quad dq = lane&7: 32 keys per block-load. The first
    // VPRE block-loads (160 keys: all of the memory's 144, most self-attention rows) are requested HERE, before the keys: V does
    // not depend on the scores, and after the softmax the loop below paid two more dependent round trips (its unrolled body, then
    // its remainder) in a kernel that is five round trips long at the row counts where it is latency-bound (192-256 rows: one
    // round of workgroups).
    constexpr int VPRE = 5;
    const int kg = lane >> 3, dq = lane & 7;
    const int nk32 = (nkeys + 31) & ~31;
    auto vblock = [&](int kk) {      // the block that holds value row kk (ANC: the hypothesis' own past lives in its ancestors' slots)
        return ANC ? a.V + (long long)a.anc[(size_t)slot * a.anc_stride + kk] * a.row_stride + hd * a.head_stride : Vb;
    };
    KvqV vpre[VPRE];
#pragma unroll
    for (int i = 0; i < VPRE; ++i) {
        const int key = wave * 8 + kg + 32 * i;
        const int kk = key < nkeys ? key : nkeys - 1;
        kvq_fetch_v(vpre[i], vblock(kk), a.nkb, kk, dq);
    }
    f32x4 q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = *(const f32x4*)(a.q + (size_t)row * 256 + hd * 32 + i * 4);
    // both key rows of the thread (keys tid and tid + 256) are requested before either is multiplied (round 5 fetched the second
    // after the first one's dot product: one more dependent round trip for rows past 256 keys); clamped addresses, no branch
    KvqK kr[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int key = tid + j * 256;
        const int kk = key < nkeys ? key : nkeys - 1;
        const char* kb = ANC ? a.K + (long long)a.anc[(size_t)slot * a.anc_stride + kk] * a.row_stride + hd * a.head_stride : Kb;
        if (j == 0 || nkeys > 256) kvq_fetch_k(kr[j], kb, a.nkb, kk);      // (uniform per workgroup)
    }
    float sc[2];
    float mx = -3.0e38f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int key = tid + j * 256;
        float s = -3.0e38f;
        if (key < nkeys) s = kvq_dot32(q, kr[j]);       // four fmaf chains on the row's integers x its power-of-two scale
        sc[j] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int key = tid + j * 256;
        const float p = key < nkeys ? expf(sc[j] - mx) : 0.f;
        ps[key] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    sum = (red[4] + red[5]) + (red[6] + red[7]);
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < VPRE; ++i)
        if (32 * i < nk32) {        // uniform: key = 32 i + (< 32)
            const float p = ps[wave * 8 + kg + 32 * i] * vpre[i].sc;     // 0 for key >= nkeys; the row's scale folded in (exact)
            const f32x4 v = kvq_decode_v(vpre[i]);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaf(v[e], p, o[e]);
        }
#pragma unroll 4
    for (int key = wave * 8 + kg + 32 * VPRE; key < nk32; key += 32) {
        const int kk = key < nkeys ? key : nkeys - 1;
        KvqV vr;
        kvq_fetch_v(vr, vblock(kk), a.nkb, kk, dq);
        const float p = ps[key] * vr.sc;        // 0 for key >= nkeys
        const f32x4 v = kvq_decode_v(vr);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaf(v[e], p, o[e]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        o[i] += __shfl_xor(o[i], 8, 64);
        o[i] += __shfl_xor(o[i], 16, 64);
        o[i] += __shfl_xor(o[i], 32, 64);
    }
    if (lane < 8) *(f32x4*)(&po[wave][dq * 4]) = o;
    __syncthreads();
    if (tid < 8) {
        const f32x4 r = (*(const f32x4*)&po[0][tid * 4] + *(const f32x4*)&po[1][tid * 4]) +
                        (*(const f32x4*)&po[2][tid * 4] + *(const f32x4*)&po[3][tid * 4]);
        *(f32x4*)(a.ctx + (size_t)row * 256 + hd * 32 + tid * 4) = r * (1.0f / sum);
    }
}

// =============================================================================================
// Head of the tick: final LayerNorm -> hidden state (kept for the bond head) -> output_layer -> log_softmax
// -> grammar mask -> EOS ban at position 0 -> argmax -> per-slot bookkeeping.
//   reference models/decoder.py:470, components.py:296-306, tokenization.py:383-392,
//   decode_strategy.py:50-56, greedy_search.py:139-161
// =============================================================================================
struct HeadArgs {
    const float* x;        // [slots, 256]
    const float* part;     // K-slice partial sums of the last layer's w_2 (EPI 4 of dec_linear_kernel), added to x here
    int part_stride, n_part;
    const float* tree_bias;  // fused tick (dec_fused.hip): `part` holds 16 partials of the last w_2, summed pairwise by index
                             // (the order every consumer of that tick uses), then + this bias [256], then + x; null otherwise
    const float* gamma;
    const float* beta;
    const float* wout_t;   // [256, VP]  (output_layer.weight transposed, padded)
    const float* bout;     // [V]
    DecState* st;
    int* tokens;           // [slots, T]
    float* token_logp;     // [slots, T]
    float* hidden;         // [slots, T, 256]
    float* logits_trace;   // [T, trace_rows, V] or null (slots 0..trace_rows-1)
    float* blp;            // BEAM: [slots, BEAM_LP_STRIDE] masked log-probs out (the pick kernel chooses)
    const int* forced;     // [trace_rows, T] or null. Teacher forcing (test aid, mnx_decode_forced): slot s < trace_rows
                           // advances with forced[s][t] instead of its own argmax; tokens[] still records the argmax,
                           // token_logp[] the masked log-prob of the FORCED id
    int V, VP, T, x0, y0, eos, trace_rows;
    int row_base;          // first row of the tick branch
    int xcd;               // dec_head4_kernel: workgroup x takes the row whose 4-row group g has g % 8 == x % 8 (FusedArgs::xcd)
};

template <bool BEAM>
__global__ __launch_bounds__(256) void dec_head_kernel(HeadArgs a) {
    __shared__ float hv[256];
    __shared__ float red[8];
    __shared__ int redi[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = a.row_base + blockIdx.x;
    const int4 rv = a.st->rowv[row];
    const int n_act = a.st->n_active;
    f32x4 xrow = *(const f32x4*)(a.x + (size_t)row * 256 + lane * 4);
    if (a.tree_bias) {
        const float* pp = a.part + (size_t)row * 256 + lane * 4;
        const size_t ps = (size_t)a.part_stride;
        f32x4 p[16];
#pragma unroll
        for (int z = 0; z < 16; ++z) p[z] = *(const f32x4*)(pp + z * ps);
#pragma unroll
        for (int w = 1; w < 16; w *= 2)
#pragma unroll
            for (int i = 0; i < 16; i += 2 * w) p[i] += p[i + w];
        xrow = xrow + (p[0] + *(const f32x4*)(a.tree_bias + lane * 4));
    } else if (a.part) {
        const float* pp = a.part + (size_t)row * 256 + lane * 4;
        const size_t ps = (size_t)a.part_stride;
        f32x4 sum = *(const f32x4*)pp;
        for (int z = 1; z < a.n_part; ++z) sum += *(const f32x4*)(pp + z * ps);
        xrow = sum + xrow;
    }
    if (row >= n_act) return;
    const int slot = rv.x, t = rv.y;
    if (wave == 0) {
        f32x4 v = xrow;
        const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.0f / 256.0f);
        v -= mean;
        const float var = wave_sum(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]) * (1.0f / 256.0f);
        const f32x4 o = v * rsqrtf(var + 1e-6f) * *(const f32x4*)(a.gamma + lane * 4) + *(const f32x4*)(a.beta + lane * 4);
        *(f32x4*)(hv + lane * 4) = o;
        *(f32x4*)(a.hidden + ((size_t)slot * a.T + t) * 256 + lane * 4) = o;
    }
    __syncthreads();
    const bool valid = tid < a.V;
    float logit = -3.0e38f;
    if (valid) {
        float s = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 8
        for (int k = 0; k < 256; k += 4) {     // 32 independent coalesced loads in flight per unrolled body
            s = fmaf(hv[k], a.wout_t[k * a.VP + tid], s);
            s1 = fmaf(hv[k + 1], a.wout_t[(k + 1) * a.VP + tid], s1);
            s2 = fmaf(hv[k + 2], a.wout_t[(k + 2) * a.VP + tid], s2);
            s3 = fmaf(hv[k + 3], a.wout_t[(k + 3) * a.VP + tid], s3);
        }
        logit = (s + s1) + (s2 + s3) + a.bout[tid];
        if (a.logits_trace && slot < a.trace_rows) a.logits_trace[((size_t)t * a.trace_rows + slot) * a.V + tid] = logit;
    }
    // log_softmax
    float m = wave_max(logit);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float e = valid ? expf(logit - m) : 0.f;
    e = wave_sum(e);
    if (lane == 0) red[4 + wave] = e;
    __syncthreads();
    const float lse = m + logf(red[4] + red[5] + red[6] + red[7]);
    float lp = logit - lse;
    const int prev = rv.z;
    if (prev >= a.x0 && prev < a.y0) { if (tid < a.y0) lp = -10000.0f; }     // after an x-bin: only y-bins
    else if (prev >= a.y0)           { if (tid >= a.x0) lp = -10000.0f; }    // after a y-bin: no coordinate bins
    if (t == 0 && tid == a.eos) lp = -1e20f;                                  // min_length = 1
    if (BEAM) {
        if (valid) a.blp[(size_t)slot * BEAM_LP_STRIDE + tid] = lp;
        return;
    }
    if (!valid) lp = -3.0e38f;
    // argmax, lowest index wins ties (topk(1))
    float bv = lp;
    int bi = tid;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    __syncthreads();
    if (lane == 0) { red[wave] = bv; redi[wave] = bi; }
    __syncthreads();
    const int ftok = (a.forced && slot < a.trace_rows) ? a.forced[(size_t)slot * a.T + t] : -1;
    if (ftok >= 0 && tid == ftok) a.token_logp[(size_t)slot * a.T + t] = lp;
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (red[w] > bv || (red[w] == bv && redi[w] < bi)) { bv = red[w]; bi = redi[w]; }
        a.tokens[(size_t)slot * a.T + t] = bi;
        if (ftok < 0) a.token_logp[(size_t)slot * a.T + t] = bv;
        const int adv = ftok >= 0 ? ftok : bi;
        a.st->prev_tok[slot] = adv;
        a.st->len[slot] = t + 1;
        a.st->t[slot] = t + 1;
        if ((a.st->stop_on_eos[slot] && adv == a.eos) || t + 1 >= a.st->max_len[slot]) a.st->alive[slot] = 0;
    }
}

// The head of the fused tick (dec_fused.hip): the same chain for one row on 1024 threads — the output layer's 232 x 256
// weights (237 KB, what this kernel's time is made of) are requested by four times as many lanes, thread (kq, column)
// multiplying k in [64 kq, 64 kq + 64) (four interleaved fmaf chains, as dec_head_kernel's), the quarters summed
// (q0 + q1) + (q2 + q3). Greedy only; everything after the logits is dec_head_kernel's code on the first 256 threads.
__global__ __launch_bounds__(1024) void dec_head4_kernel(HeadArgs a) {
    __shared__ float hv[256];
    __shared__ float lq[4][256];
    __shared__ float red[8];
    __shared__ int redi[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = tid & 255, kq = tid >> 8;
    const int groups = gridDim.x >> 2;
    const int row = a.row_base + (a.xcd ? 4 * ((int)blockIdx.x % groups) + (int)blockIdx.x / groups : (int)blockIdx.x);
    const int4 rv = a.st->rowv[row];
    const int n_act = a.st->n_active;
    const bool valid = col < a.V;
    // requests in the order of need: the row's stream + partial planes (wave 0), then — a dummy row (capacity > alive rows)
    // leaves here, before it asks for 237 KB of weights — the output layer's weights (16 per batch, 4 batches; they are not
    // needed before the LayerNorm below)
    f32x4 p[16];
    f32x4 xv = {0.f, 0.f, 0.f, 0.f};
    if (wave == 0) {
        const float* pp = a.part + (size_t)row * 256 + lane * 4;
        const size_t ps = (size_t)a.part_stride;
#pragma unroll
        for (int z = 0; z < 16; ++z) p[z] = *(const f32x4*)(pp + z * ps);
        xv = *(const f32x4*)(a.x + (size_t)row * 256 + lane * 4);
    }
    if (row >= n_act) return;
    float wk[16];
    const float* wp = a.wout_t + (size_t)(64 * kq) * a.VP + (valid ? col : 0);
#pragma unroll
    for (int k = 0; k < 16; ++k) wk[k] = wp[(size_t)k * a.VP];
    if (wave == 0) {
        f32x4 v = xv;
#pragma unroll
        for (int w = 1; w < 16; w *= 2)
#pragma unroll
            for (int i = 0; i < 16; i += 2 * w) p[i] += p[i + w];
        v = v + (p[0] + *(const f32x4*)(a.tree_bias + lane * 4));
        const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.0f / 256.0f);
        v -= mean;
        const float var = wave_sum(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]) * (1.0f / 256.0f);
        const f32x4 o = v * rsqrtf(var + 1e-6f) * *(const f32x4*)(a.gamma + lane * 4) + *(const f32x4*)(a.beta + lane * 4);
        *(f32x4*)(hv + lane * 4) = o;
        *(f32x4*)(a.hidden + ((size_t)rv.x * a.T + rv.y) * 256 + lane * 4) = o;
    }
    __syncthreads();
    {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        const float* hq = hv + 64 * kq;
#pragma unroll
        for (int kb = 0; kb < 64; kb += 16) {
            float wn[16];
            if (kb + 16 < 64) {
#pragma unroll
                for (int k = 0; k < 16; ++k) wn[k] = wp[(size_t)(kb + 16 + k) * a.VP];
            }
#pragma unroll
            for (int k = 0; k < 16; k += 4) {
                s0 = fmaf(hq[kb + k], wk[k], s0);
                s1 = fmaf(hq[kb + k + 1], wk[k + 1], s1);
                s2 = fmaf(hq[kb + k + 2], wk[k + 2], s2);
                s3 = fmaf(hq[kb + k + 3], wk[k + 3], s3);
            }
            if (kb + 16 < 64) {
#pragma unroll
                for (int k = 0; k < 16; ++k) wk[k] = wn[k];
            }
        }
        lq[kq][col] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    const int slot = rv.x, t = rv.y;
    float logit = -3.0e38f;
    if (tid < 256 && valid) {
        logit = ((lq[0][col] + lq[1][col]) + (lq[2][col] + lq[3][col])) + a.bout[col];
        if (a.logits_trace && slot < a.trace_rows) a.logits_trace[((size_t)t * a.trace_rows + slot) * a.V + col] = logit;
    }
    // log_softmax (the first four waves hold the 232 logits, the others contribute neutral elements)
    float m = wave_max(logit);
    if (lane == 0 && wave < 4) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float e = (tid < 256 && valid) ? expf(logit - m) : 0.f;
    e = wave_sum(e);
    if (lane == 0 && wave < 4) red[4 + wave] = e;
    __syncthreads();
    const float lse = m + logf(red[4] + red[5] + red[6] + red[7]);
    float lp = logit - lse;
    const int prev = rv.z;
    if (prev >= a.x0 && prev < a.y0) { if (col < a.y0) lp = -10000.0f; }     // after an x-bin: only y-bins
    else if (prev >= a.y0)           { if (col >= a.x0) lp = -10000.0f; }    // after a y-bin: no coordinate bins
    if (t == 0 && col == a.eos) lp = -1e20f;                                  // min_length = 1
    if (!(tid < 256 && valid)) lp = -3.0e38f;
    // argmax, lowest index wins ties (topk(1)); threads beyond the first 256 carry neutral elements
    float bv = lp;
    int bi = tid;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0 && wave < 4) { red[wave] = bv; redi[wave] = bi; }
    __syncthreads();
    const int ftok = (a.forced && slot < a.trace_rows) ? a.forced[(size_t)slot * a.T + t] : -1;
    if (ftok >= 0 && tid == ftok) a.token_logp[(size_t)slot * a.T + t] = lp;
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (red[w] > bv || (red[w] == bv && redi[w] < bi)) { bv = red[w]; bi = redi[w]; }
        a.tokens[(size_t)slot * a.T + t] = bi;
        if (ftok < 0) a.token_logp[(size_t)slot * a.T + t] = bv;
        const int adv = ftok >= 0 ? ftok : bi;
        a.st->prev_tok[slot] = adv;
        a.st->len[slot] = t + 1;
        a.st->t[slot] = t + 1;
        if ((a.st->stop_on_eos[slot] && adv == a.eos) || t + 1 >= a.st->max_len[slot]) a.st->alive[slot] = 0;
    }
}

// Opens a tick: PE rank of every slot (rank among the alive slots of its chunk, by row index) and the alive
// counters the host polls. One workgroup; the kernel boundary is the all-rows barrier.
__global__ __launch_bounds__(BEGIN_THREADS) void dec_begin_kernel(DecState* st, int slots) {
    __shared__ unsigned int s_mask[MAX_CHUNKS];     // bit r = row r of the chunk is alive (rows per chunk <= 32)
    __shared__ int s_wave[BEGIN_THREADS / 64];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < MAX_CHUNKS) s_mask[tid] = 0u;
    if (tid == 0) s_base = 0;
    __syncthreads();
    // pass 1: per-chunk alive bitmaps + compact list of alive slots in slot order
    for (int s0 = 0; s0 < slots; s0 += BEGIN_THREADS) {
        const int s = s0 + tid;
        const int al = s < slots ? st->alive[s] : 0;
        const unsigned long long bal = __ballot(al != 0);
        if (lane == 0) s_wave[wave] = __popcll(bal);
        if (al) atomicOr(&s_mask[st->chunk[s] & (MAX_CHUNKS - 1)], 1u << (st->rowc[s] & 31));
        __syncthreads();
        int base = s_base, total = 0;
        for (int w = 0; w < BEGIN_THREADS / 64; ++w) {
            const int cw = s_wave[w];
            if (w < wave) base += cw;
            total += cw;
        }
        if (al) st->active[base + __popcll(bal & ((1ull << lane) - 1ull))] = s;
        __syncthreads();
        if (tid == 0) s_base += total;
        __syncthreads();
    }
    // pass 2: PE rank of every alive slot = alive chunk-mates with a smaller row index
    for (int s = tid; s < slots; s += BEGIN_THREADS)
        if (st->alive[s]) {
            const int c = st->chunk[s] & (MAX_CHUNKS - 1), rc = st->rowc[s] & 31;
            st->rank[s] = __popc(s_mask[c] & ((1u << rc) - 1u));
        }
    if (tid < MAX_CHUNKS) st->chunk_alive[tid] = __popc(s_mask[tid]);
    if (tid == 0) { st->n_active = s_base; st->tick = st->tick + 1; }
    // row view (dec_types.h): active[] and rank[] above were written by other threads of this workgroup
    __syncthreads();
    const int n = s_base;
    for (int r = tid; r < slots; r += BEGIN_THREADS) {
        int4 v = {0, 0, 0, 0};
        int mb = 0;
        if (r < n) {
            const int s = st->active[r];
            v = (int4){s, st->t[s], st->prev_tok[s], st->rank[s]};
            mb = st->mem_blk[s];
        }
        st->rowv[r] = v;
        st->row_mem[r] = mb;
    }
}

__global__ __launch_bounds__(BEGIN_THREADS) void dec_reset_kernel(DecState* st) {
    const int tid = threadIdx.x;
    if (tid == 0) { st->tick = 0; st->n_active = 0; }
    if (tid < MAX_CHUNKS) st->chunk_alive[tid] = 0;
    for (int i = tid; i < MAX_SLOTS; i += BEGIN_THREADS) {
        st->alive[i] = 0; st->t[i] = 0; st->len[i] = 0; st->chunk[i] = -1; st->rank[i] = 0;
    }
}

// Admit n rows of one reference batch into the given slots (any free slots).
__global__ void dec_admit_kernel(DecState* st, const int* slots, const int* rowc, int n, int chunk_tag, int mem_blk0,
                                 int max_len, int stop_on_eos, int sos) {
    const int i = threadIdx.x;
    if (i < n) {
        const int s = slots[i];
        st->alive[s] = 1; st->t[s] = 0; st->prev_tok[s] = sos; st->len[s] = 0;
        st->chunk[s] = chunk_tag; st->rowc[s] = rowc ? rowc[i] : i; st->rank[s] = 0;
        st->mem_blk[s] = mem_blk0 + i; st->max_len[s] = max_len; st->stop_on_eos[s] = stop_on_eos;
    }
    if (i == 0) { atomicAdd(&st->n_active, n); atomicAdd(&st->chunk_alive[chunk_tag], n); }
}

// ---- host-side enqueue helpers (engine.hip captures the tick into a hipGraph) -----------------
template <int PRO, int EPI>
static void lin(hipStream_t s, const LinArgs& a, int slots) {
    hipLaunchKernelGGL((dec_linear_kernel<PRO, EPI>), dim3(a.N / TN, slots / ROW_TILE, EPI == 4 ? a.K / 256 : 1), dim3(256), 0, s, a);
}

hipError_t dec_enqueue_status(const DecBuffers& b, int slots, hipStream_t s) {
    hipLaunchKernelGGL(dec_begin_kernel, dim3(1), dim3(BEGIN_THREADS), 0, s, b.st, slots);
    return hipGetLastError();
}

hipError_t dec_enqueue_reset(const DecBuffers& b, hipStream_t s) {
    hipLaunchKernelGGL(dec_reset_kernel, dim3(1), dim3(BEGIN_THREADS), 0, s, b.st);
    return hipGetLastError();
}

hipError_t dec_enqueue_admit(const DecBuffers& b, const int* slots_dev, const int* rowc_dev, int n, int chunk_tag,
                             int mem_blk0, int max_len, int stop_on_eos, hipStream_t s) {
    hipLaunchKernelGGL(dec_admit_kernel, dim3(1), dim3(64), 0, s, b.st, slots_dev, rowc_dev, n, chunk_tag, mem_blk0,
                       max_len, stop_on_eos, 1);
    return hipGetLastError();
}

__global__ void beam_begin_kernel(DecState* st, int B, int K, int ref_batch);
__global__ void beam_pick_kernel(DecState* st, BeamBuffers bm, const float* hidden, int* etok, int T, int V, int eos);

hipError_t dec_enqueue_tick(const DecWeights& w, const DecBuffers& b, int slots_scan, int rows, float* logits_trace,
                            int trace_rows, hipStream_t s, const BeamBuffers* beam, const int* forced, int fused_tile) {
    // slots_scan: state slots the begin kernel scans; rows: capacity of the compact active list this tick is
    // launched for (a multiple of 32, >= the number of alive slots — the host guarantees it)
    if (beam) hipLaunchKernelGGL(beam_begin_kernel, dim3(1), dim3(256), 0, s, b.st, beam->B, beam->K, beam->ref_batch);
    else hipLaunchKernelGGL(dec_begin_kernel, dim3(1), dim3(BEGIN_THREADS), 0, s, b.st, slots_scan);
    return dec_enqueue_tick_rows(w, b, 0, rows, logits_trace, trace_rows, s, beam, forced, fused_tile);
}

// The layers + head of a tick for rows [row_base, row_base + rows) of the compact active list (the begin kernel has run).
// Rows are independent through the whole stack, so a tick may be enqueued as several BRANCHES of rows on different streams
// (engine.hip captures them as parallel branches of the tick graph): every activation buffer is indexed by row.
hipError_t dec_enqueue_tick_rows(const DecWeights& w, const DecBuffers& b, int row_base, int rows, float* logits_trace,
                                 int trace_rows, hipStream_t s, const BeamBuffers* beam, const int* forced, int fused_tile) {
    const int D = 256, H = w.heads, T = b.T;
    const int slots = rows;
    const bool split_w2 = beam == nullptr;
    const bool fused = beam == nullptr && fused_tile > 0;
    const float *fx = nullptr, *fp = nullptr;
    if (fused) {    // three launches per layer (dec_fused.hip); the head sums the last w_2's 16 partials
        hipError_t e = dec_enqueue_fused_layers(w, b, row_base, rows, fused_tile, s, &fx, &fp);
        if (e != hipSuccess) return e;
    }
    for (int l = 0; l < (fused ? 0 : w.layers); ++l) {
        const DecLayerW& L = w.L[l];
        const size_t self_blk = kvq_block_bytes(b.Tq), mem_blk = kvq_block_bytes(b.Sq);
        char* kc = b.self_k + (size_t)l * b.slots * H * self_blk;
        char* vc = b.self_v + (size_t)l * b.slots * H * self_blk;
        LinArgs a = {};
        a.st = b.st; a.T = T; a.Tq = b.Tq; a.heads = H; a.row_base = row_base;
        // LN1 (+ embedding at layer 0) -> q, k, v
        // the residual stream alternates between two buffers: layer l > 0 sums (stream of layer l-1) + (its w_2 slices)
        // while it normalises them, and column-block 0 writes the sum to the other buffer for the rest of layer l
        // (beam search keeps w_2 unsplit and the stream in one buffer: its oracle comparison runs hundreds of steps through
        //  1-ulp score ties, which only the summation order it was validated with reproduces — tests/test_gpu_parity.py)
        float* xin = (l == 0 || !split_w2) ? b.x : (((l - 1) & 1) ? b.x2 : b.x);
        float* xl = (split_w2 && (l & 1)) ? b.x2 : b.x;
        a.in = xin; a.W = L.wqkv; a.bias = L.bqkv; a.gamma = L.ln1_g; a.beta = L.ln1_b; a.out = b.q;
        a.kcache = kc; a.vcache = vc; a.x_write = xl; a.emb = w.emb; a.pe = w.pe; a.N = 3 * D; a.K = D;
        a.part = (l == 0 || !split_w2) ? nullptr : b.part; a.part_stride = b.slots * D; a.n_part = w.dff / 256;
        if (l == 0) lin<2, 0>(s, a, slots); else lin<1, 0>(s, a, slots);
        a.part = nullptr;
        AttnArgs at = {};
        at.q = b.q; at.K = kc; at.V = vc; at.ctx = b.ctx; at.st = b.st; at.heads = H; at.cross = 0; at.row_base = row_base;
        at.row_stride = (long long)(H * self_blk); at.head_stride = (long long)self_blk; at.nkb = b.Tq; at.fixed_keys = 0;
        if (beam) {
            at.anc = beam->anc; at.anc_stride = beam->anc_stride;
            hipLaunchKernelGGL(dec_attn_kernel<true>, dim3(slots * H), dim3(256), 0, s, at);
        } else {
            hipLaunchKernelGGL(dec_attn_kernel<false>, dim3(slots * H), dim3(256), 0, s, at);
        }
        // self final_linear + residual
        a.in = b.ctx; a.W = L.wo; a.bias = L.bo; a.out = xl; a.N = D; a.K = D;
        lin<0, 1>(s, a, slots);
        // LN2 -> context query
        a.in = xl; a.W = L.wq2; a.bias = L.bq2; a.gamma = L.ln2_g; a.beta = L.ln2_b; a.out = b.q;
        lin<1, 2>(s, a, slots);
        at.anc = nullptr;
        at.K = b.mem_kv + (size_t)l * 2 * H * mem_blk;    // memory K/V: [memory block][layer][K|V][head] blocks of Sq rows
        at.V = at.K + (size_t)H * mem_blk;
        at.row_stride = (long long)((size_t)w.layers * 2 * H * mem_blk); at.head_stride = (long long)mem_blk; at.nkb = b.Sq;
        at.fixed_keys = b.S; at.cross = 1;
        hipLaunchKernelGGL(dec_attn_kernel<false>, dim3(slots * H), dim3(256), 0, s, at);
        // context final_linear + residual
        a.in = b.ctx; a.W = L.wo2; a.bias = L.bo2; a.out = xl; a.N = D; a.K = D;
        lin<0, 1>(s, a, slots);
        // feed-forward: LN -> w_1 -> GELU -> w_2 (four K slices; summed with the stream by the next reader)
        a.in = xl; a.W = L.w1; a.bias = L.b1; a.gamma = L.lnf_g; a.beta = L.lnf_b; a.out = b.h; a.N = w.dff; a.K = D;
        lin<1, 3>(s, a, slots);
        a.in = b.h; a.W = L.w2; a.bias = L.b2; a.N = D; a.K = w.dff;
        if (split_w2) { a.out = b.part; lin<0, 4>(s, a, slots); }
        else { a.out = xl; lin<0, 1>(s, a, slots); }
    }
    HeadArgs h = {};
    h.x = (split_w2 && ((w.layers - 1) & 1)) ? b.x2 : b.x; h.part = split_w2 ? b.part : nullptr; h.part_stride = b.slots * D;
    h.n_part = w.dff / 256;
    if (fused) { h.x = fx; h.part = fp; h.part_stride = b.fpart_rows * D; h.tree_bias = w.L[w.layers - 1].b2; h.xcd = (((fused_tile / 1000) & 1) && fused_tile < 2000 && rows % 32 == 0) ? 1 : 0; } h.gamma = w.lnF_g; h.beta = w.lnF_b; h.wout_t = w.wout_t; h.bout = w.bout; h.st = b.st;
    h.tokens = b.tokens; h.token_logp = b.logp; h.hidden = b.hidden; h.logits_trace = logits_trace;
    h.V = w.vocab; h.VP = w.vpad; h.T = T; h.x0 = w.sym_offset; h.y0 = w.sym_offset + w.bins;
    h.eos = 2; h.trace_rows = trace_rows; h.forced = forced; h.row_base = row_base;
    if (beam) {
        h.blp = beam->blp;
        hipLaunchKernelGGL(dec_head_kernel<true>, dim3(slots), dim3(256), 0, s, h);
        hipLaunchKernelGGL(beam_pick_kernel, dim3(beam->B), dim3(256), 0, s, b.st, *beam, b.hidden, b.tokens, T, w.vocab, 2);
    } else if (fused) {
        hipLaunchKernelGGL(dec_head4_kernel, dim3(slots), dim3(1024), 0, s, h);
    } else {
        hipLaunchKernelGGL(dec_head_kernel<false>, dim3(slots), dim3(256), 0, s, h);
    }
    return hipGetLastError();
}

// =============================================================================================
// On-device atom positions: the 'indices' that CharTokenizer.sequence_to_smiles derives from a decoded id
// sequence (reference tokenization.py:464-515): for every atom token group followed by "x y <next>", the
// position of <next>. One thread per sequence (a <= 480-step scan).
// =============================================================================================
__global__ __launch_bounds__(64) void atom_scan_kernel(const int* __restrict__ lens, const int* __restrict__ tokens,
                                                       const TokenClasses* __restrict__ tc, const int* __restrict__ slots,
                                                       int n_rows, int T, int kmax, int* __restrict__ atom_idx,
                                                       int* __restrict__ n_atoms) {
    // one workgroup per sequence: the ids are staged in LDS by all lanes, then lane 0 runs the sequential scan
    __shared__ int seq[512];
    __shared__ unsigned char fl[256];
    const int row = blockIdx.x;
    const int slot = slots ? slots[row] : row;
    const int n = min(lens[slot], 512);
    for (int i = threadIdx.x; i < n; i += 64) seq[i] = tokens[(size_t)slot * T + i];
    for (int i = threadIdx.x; i < 256; i += 64) fl[i] = tc->flags[i];
    __syncthreads();
    if (threadIdx.x != 0) return;
    const int x0 = tc->x0, y0 = tc->y0, lb = tc->lbracket, rb = tc->rbracket;
    const int iC = tc->id_C, il = tc->id_l, iB = tc->id_B, ir = tc->id_r;
    int i = 0, k = 0;
    while (i < n) {
        const int t = seq[i];
        if (t == 2 || t == 0) break;                                  // <eos> / <pad>
        if (t >= x0) { ++i; continue; }                               // coordinate bins
        if (!(fl[t] & 2)) { ++i; continue; }                          // not an atom token
        int j;
        if (t == lb) {
            j = i + 1;
            while (j < n && seq[j] < x0 && (fl[seq[j]] & 1)) {
                ++j;
                if (seq[j - 1] == rb) break;
            }
        } else if (i + 1 < n && ((t == iC && seq[i + 1] == il) || (t == iB && seq[i + 1] == ir))) {
            j = i + 2;
        } else {
            j = i + 1;
        }
        if (j + 2 < n && seq[j] >= x0 && seq[j] < y0 && seq[j + 1] >= y0) {
            if (k < kmax) atom_idx[(size_t)row * kmax + k] = j + 2;
            ++k;
            i = j + 2;
        } else {
            i = j;
        }
    }
    n_atoms[row] = k < kmax ? k : kmax;
}

hipError_t atoms_enqueue(const DecBuffers& b, const TokenClasses* tc_dev, const int* slots_dev, int n, int kmax,
                         int* atom_idx, int* n_atoms, hipStream_t s) {
    hipLaunchKernelGGL(atom_scan_kernel, dim3(n), dim3(64), 0, s, b.st->len, b.tokens, tc_dev, slots_dev, n, b.T, kmax,
                       atom_idx, n_atoms);
    return hipGetLastError();
}

hipError_t atoms_enqueue_raw(const TokenClasses* tc_dev, const int* tokens, const int* lens, int n, int T, int kmax,
                             int* atom_idx, int* n_atoms, hipStream_t s) {
    hipLaunchKernelGGL(atom_scan_kernel, dim3(n), dim3(64), 0, s, lens, tokens, tc_dev, (const int*)nullptr, n, T, kmax,
                       atom_idx, n_atoms);
    return hipGetLastError();
}

// =============================================================================================
// Bond head (reference components.py:365-400,478-484)
//   logits[i,j] = W2 . GELU(W1a.h_i + W1b.h_j + b1) + b2 ; the two halves of the first Linear are applied
//   once per atom (SGEMM above) instead of once per pair.
// =============================================================================================
__global__ void edge_gather_kernel(const float* __restrict__ hidden, const int* __restrict__ slot_map,
                                   const int* __restrict__ atom_idx, const int* __restrict__ n_atoms,
                                   float* __restrict__ g, int kmax, int max_len) {
    const int b = blockIdx.y, i = blockIdx.x, lane = threadIdx.x;
    int idx = i < n_atoms[b] ? atom_idx[b * kmax + i] : 0;
    idx = min(max(idx, 0), max_len - 1);
    const size_t row = slot_map ? (size_t)slot_map[b] : (size_t)b;
    *(f32x4*)(g + ((size_t)b * kmax + i) * 256 + lane * 4) =
        *(const f32x4*)(hidden + (row * max_len + idx) * 256 + lane * 4);
}

// UV [B*kmax, 512]: cols 0..255 = W1a.h (+0), cols 256..511 = W1b.h + b1.  One workgroup per (b, i); lane = j.
__global__ __launch_bounds__(256) void edge_pair_kernel(const float* __restrict__ UV, const float* __restrict__ w2,
                                                        const float* __restrict__ b2, const int* __restrict__ n_atoms,
                                                        float* __restrict__ prob, int kmax) {
    __shared__ float us[256];
    __shared__ float w2s[7 * 256];
    const int b = blockIdx.y, i = blockIdx.x, tid = threadIdx.x;
    const int k = n_atoms[b];
    if (i >= k) return;
    us[tid] = UV[((size_t)b * kmax + i) * 512 + tid];
    for (int q = tid; q < 7 * 256; q += 256) w2s[q] = w2[q];
    __syncthreads();
    for (int j = tid; j < k; j += 256) {
        const float* vp = UV + ((size_t)b * kmax + j) * 512 + 256;
        float o[7];
#pragma unroll
        for (int c = 0; c < 7; ++c) o[c] = b2[c];
        for (int ch = 0; ch < 256; ch += 4) {
            const f32x4 v = *(const f32x4*)(vp + ch);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float z = gelu_erf(us[ch + u] + v[u]);
#pragma unroll
                for (int c = 0; c < 7; ++c) o[c] = fmaf(z, w2s[c * 256 + ch + u], o[c]);
            }
        }
        float m = o[0];
#pragma unroll
        for (int c = 1; c < 7; ++c) m = fmaxf(m, o[c]);
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < 7; ++c) { o[c] = expf(o[c] - m); sum += o[c]; }
        float* pp = prob + (((size_t)b * kmax + i) * kmax + j) * 8;
#pragma unroll
        for (int c = 0; c < 7; ++c) pp[c] = o[c] / sum;
    }
}

// get_edge_prediction (components.py:383-400): averages are taken in float64 on float32 probabilities, exactly
// as the reference does on Python float lists; argmax = first maximum.
__global__ void edge_sym_kernel(const float* __restrict__ prob, const int* __restrict__ n_atoms,
                                unsigned char* __restrict__ edges, double* __restrict__ scores, int kmax) {
    const int b = blockIdx.z, i = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = n_atoms[b];
    if (i >= k || j >= k) return;
    const float* pij = prob + (((size_t)b * kmax + i) * kmax + j) * 8;
    const float* pji = prob + (((size_t)b * kmax + j) * kmax + i) * 8;
    double e[7];
    if (i == j) {
#pragma unroll
        for (int c = 0; c < 7; ++c) e[c] = (double)pij[c];
    } else {
#pragma unroll
        for (int c = 0; c < 5; ++c) e[c] = i < j ? ((double)pij[c] + (double)pji[c]) / 2 : ((double)pji[c] + (double)pij[c]) / 2;
        if (i < j) {
            e[5] = ((double)pij[5] + (double)pji[6]) / 2;
            e[6] = ((double)pij[6] + (double)pji[5]) / 2;
        } else {  // lower triangle mirrors the upper one with 5 <-> 6 swapped
            e[5] = ((double)pji[6] + (double)pij[5]) / 2;
            e[6] = ((double)pji[5] + (double)pij[6]) / 2;
        }
    }
    int best = 0;
    double bv = e[0];
#pragma unroll
    for (int c = 1; c < 7; ++c)
        if (e[c] > bv) { bv = e[c]; best = c; }
    edges[((size_t)b * kmax + i) * kmax + j] = (unsigned char)best;
    if (scores) scores[((size_t)b * kmax + i) * kmax + j] = bv;
}

hipError_t edges_enqueue(const DecWeights& w, const DecBuffers& bf, const float* hidden, const int* slot_map,
                         const int* atom_idx, const int* n_atoms, int B, int kmax, int max_len,
                         unsigned char* edges, double* scores, hipStream_t s) {
    hipLaunchKernelGGL(edge_gather_kernel, dim3(kmax, B), dim3(64), 0, s, hidden, slot_map, atom_idx, n_atoms,
                       bf.edge_g, kmax, max_len);
    hipError_t e = launch_sgemm_tn(bf.edge_g, w.edge_w1cat, w.edge_b1cat, bf.edge_uv, B * kmax, 512, 256, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(edge_pair_kernel, dim3(kmax, B), dim3(256), 0, s, bf.edge_uv, w.edge_w2, w.edge_b2, n_atoms,
                       bf.edge_prob, kmax);
    hipLaunchKernelGGL(edge_sym_kernel, dim3((kmax + 63) / 64, kmax, B), dim3(64), 0, s, bf.edge_prob, n_atoms, edges,
                       scores, kmax);
    return hipGetLastError();
}

}  // namespace mnx

namespace mnx {

// ---- measurement aid (mnx_probe_decode_attn): `rows` sequences resident at position t, then the self- and the cross-
// attention kernel of layer 0 launched `iters` times each between HIP events (isolated: algorithmic HBM bytes are known exactly)
__global__ void dec_probe_state_kernel(DecState* st, int rows, int t, int max_len) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows) {
        st->alive[i] = 1; st->t[i] = t; st->prev_tok[i] = 5; st->len[i] = t; st->chunk[i] = (i >> 5) & (MAX_CHUNKS - 1);
        st->rowc[i] = i & 31; st->rank[i] = i & 31; st->mem_blk[i] = i; st->max_len[i] = max_len; st->stop_on_eos[i] = 0;
    }
}

hipError_t dec_probe_attn(const DecWeights& w, const DecBuffers& b, int rows, int t, int iters, hipEvent_t* ev,
                          hipStream_t s) {
    const int D = 256, H = w.heads, T = b.T;
    const int cap = (rows + ROW_TILE - 1) / ROW_TILE * ROW_TILE;
    hipLaunchKernelGGL(dec_reset_kernel, dim3(1), dim3(BEGIN_THREADS), 0, s, b.st);
    hipLaunchKernelGGL(dec_probe_state_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, b.st, rows, t, T);
    hipLaunchKernelGGL(dec_begin_kernel, dim3(1), dim3(BEGIN_THREADS), 0, s, b.st, cap);
    // Launch i reads the K/V of LAYER i % layers, as the six attention launches of a real tick do: the bytes touched
    // over one cycle (6 x 102 MB self at 768 rows / position 64, 6 x 226 MB cross) exceed the 256 MB Infinity Cache, so
    // the per-launch time is an HBM figure — re-launching one layer (round 2) measured the cache instead.
    const size_t self_blk = kvq_block_bytes(b.Tq), mem_blk = kvq_block_bytes(b.Sq);
    const size_t self_layer = (size_t)b.slots * H * self_blk;
    AttnArgs at = {};
    at.q = b.q; at.ctx = b.ctx; at.st = b.st; at.heads = H; at.cross = 0;
    at.row_stride = (long long)(H * self_blk); at.head_stride = (long long)self_blk; at.nkb = b.Tq;
    auto self_launch = [&](int i) {
        const int l = i % w.layers;
        at.K = b.self_k + (size_t)l * self_layer; at.V = b.self_v + (size_t)l * self_layer;
        hipLaunchKernelGGL(dec_attn_kernel<false>, dim3(cap * H), dim3(256), 0, s, at);
    };
    for (int i = 0; i < w.layers; ++i) self_launch(i);                                     // warm-up: one full cycle
    hipError_t e = hipEventRecord(ev[0], s);
    for (int i = 0; i < iters; ++i) self_launch(i);
    if (e == hipSuccess) e = hipEventRecord(ev[1], s);
    at.row_stride = (long long)((size_t)w.layers * 2 * H * mem_blk); at.head_stride = (long long)mem_blk; at.nkb = b.Sq;
    at.fixed_keys = b.S; at.cross = 1;
    auto cross_launch = [&](int i) {
        const int l = i % w.layers;
        at.K = b.mem_kv + (size_t)l * 2 * H * mem_blk; at.V = at.K + (size_t)H * mem_blk;
        hipLaunchKernelGGL(dec_attn_kernel<false>, dim3(cap * H), dim3(256), 0, s, at);
    };
    for (int i = 0; i < w.layers; ++i) cross_launch(i);                                    // warm-up
    if (e == hipSuccess) e = hipEventRecord(ev[2], s);
    for (int i = 0; i < iters; ++i) cross_launch(i);
    if (e == hipSuccess) e = hipEventRecord(ev[3], s);
    hipLaunchKernelGGL(dec_reset_kernel, dim3(1), dim3(BEGIN_THREADS), 0, s, b.st);
    return e != hipSuccess ? e : hipGetLastError();
}

}  // namespace mnx

namespace mnx {

// Gather per-slot results into the caller's per-image layout: out[row] <- slot buffers of slots[row].
__global__ void rows_gather_kernel(const DecState* st, const int* __restrict__ slots, const int* __restrict__ tokens,
                                   const float* __restrict__ logp, const float* __restrict__ hidden, int T, int out_len,
                                   int* __restrict__ o_tokens, int* __restrict__ o_len, float* __restrict__ o_logp,
                                   float* __restrict__ o_hidden) {
    const int row = blockIdx.x, tid = threadIdx.x;
    const int slot = slots ? slots[row] : row;
    const int n = min(st->len[slot], out_len);
    if (tid == 0) o_len[row] = n;
    for (int i = tid; i < out_len; i += blockDim.x) {
        o_tokens[(size_t)row * out_len + i] = i < n ? tokens[(size_t)slot * T + i] : 0;
        if (o_logp) o_logp[(size_t)row * out_len + i] = i < n ? logp[(size_t)slot * T + i] : 0.f;
    }
    if (o_hidden) {
        const f32x4* src = (const f32x4*)(hidden + (size_t)slot * T * 256);
        f32x4* dst = (f32x4*)(o_hidden + (size_t)row * out_len * 256);
        for (int i = tid; i < n * 64; i += blockDim.x) dst[i] = src[i];
    }
}

hipError_t gather_enqueue(const DecBuffers& b, const int* slots_dev, int n_rows, int out_len, int* o_tokens, int* o_len,
                          float* o_logp, float* o_hidden, hipStream_t s) {
    hipLaunchKernelGGL(rows_gather_kernel, dim3(n_rows), dim3(256), 0, s, b.st, slots_dev, b.tokens, b.logp, b.hidden,
                       b.T, out_len, o_tokens, o_len, o_logp, o_hidden);
    return hipGetLastError();
}

// admit variant used by mnx_decode_greedy: slot i = row i, chunk ids given per row on the device
__global__ void dec_admit_rows_kernel(DecState* st, const int* chunk_ids, int n, int max_len, int stop_on_eos, int sos) {
    const int i = threadIdx.x;
    if (i < n) {
        st->alive[i] = 1; st->t[i] = 0; st->prev_tok[i] = sos; st->len[i] = 0;
        st->chunk[i] = chunk_ids ? chunk_ids[i] : 0; st->rowc[i] = i; st->rank[i] = 0;
        st->mem_blk[i] = i; st->max_len[i] = max_len; st->stop_on_eos[i] = stop_on_eos;
    }
    if (i == 0) st->n_active = n;
}

hipError_t dec_enqueue_admit_rows(const DecBuffers& b, const int* chunk_ids_dev, int n, int max_len, int stop_on_eos,
                                  hipStream_t s) {
    hipLaunchKernelGGL(dec_admit_rows_kernel, dim3(1), dim3(64), 0, s, b.st, chunk_ids_dev, n, max_len, stop_on_eos, 1);
    return hipGetLastError();
}

}  // namespace mnx

namespace mnx {

// =============================================================================================
// Beam search. The strategy follows the reference's BeamSearch.advance / update_finished
// (decoding/beam_search.py:84-190; pinned through oracle/beam.py's BeamStrategy): scores are cumulative log-probs
// divided by (emitted tokens + 2) — the reference counts <sos> and the new token —, a flat top-K over K x V per
// image (ties: lowest flat index), finished hypotheses keep their row with a cumulative log-prob of -1e10, an image
// leaves the batch when its top beam has finished at some step and >= n_best hypotheses are stored; the n_best
// kept are the best by score (stable). The decode loop around it is ours (the reference's cannot run):
// back-pointers are followed through an ancestry table instead of re-ordering the K/V caches: entry tau of a
// hypothesis names the slot whose cache row tau / hidden row tau it inherits, and whose id row tau-1 it emitted.
// The positional-encoding row is the row in the current (alive images x K) batch, as for greedy.
// =============================================================================================
__global__ void beam_init_kernel(DecState* st, BeamBuffers bm, int max_len, int sos) {
    const int n = bm.B * bm.K;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        st->alive[i] = 1; st->t[i] = 0; st->prev_tok[i] = sos; st->len[i] = 0;
        st->chunk[i] = 0; st->rowc[i] = i; st->rank[i] = i; st->mem_blk[i] = i / bm.K;
        st->max_len[i] = max_len; st->stop_on_eos[i] = 1;
        bm.bs->cum[i] = (i % bm.K == 0) ? 0.0f : -__builtin_inff();     // beam_search.py:43-45
        bm.anc[(size_t)i * bm.anc_stride] = i;
    }
    for (int i = threadIdx.x; i < MAX_BEAM_IMGS; i += blockDim.x) { bm.bs->top_fin[i] = 0; bm.bs->n_hyps[i] = 0; bm.bs->pool_n[i] = 0; }
    if (threadIdx.x == 0) st->n_active = n;
}

// rows of the step: alive images in index order x K. PE row = position in the (alive images x K) batch of the image's OWN
// reference batch (images [g ref_batch, (g + 1) ref_batch)): several reference batches share a step, none sees the others.
__global__ void beam_begin_kernel(DecState* st, int B, int K, int ref_batch) {
    __shared__ int s_alive[MAX_BEAM_IMGS];
    __shared__ int s_before[MAX_BEAM_IMGS];       // alive images with a smaller index (all batches): the row order
    const int tid = threadIdx.x;
    for (int i = tid; i < MAX_BEAM_IMGS; i += blockDim.x) s_alive[i] = i < B ? st->alive[i * K] : 0;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < B; ++i) { s_before[i] = run; run += s_alive[i]; }
    }
    __syncthreads();
    const int total = B > 0 ? s_before[B - 1] + s_alive[B - 1] : 0;
    for (int e = tid; e < B * K; e += blockDim.x) {
        const int img = e / K, j = e % K;
        if (s_alive[img]) {
            const int first = img / ref_batch * ref_batch;             // first image of this image's reference batch
            const int r_in_batch = s_before[img] - s_before[first];
            st->rank[e] = r_in_batch * K + j;
            st->active[s_before[img] * K + j] = e;
        }
    }
    if (tid == 0) { st->n_active = total * K; st->chunk_alive[0] = total * K; st->tick = st->tick + 1; }
    __syncthreads();
    for (int r = tid; r < ((B * K + ROW_TILE - 1) & ~(ROW_TILE - 1)); r += blockDim.x) {     // row view (dec_types.h)
        int4 v = {0, 0, 0, 0};
        int mb = 0;
        if (r < total * K) {
            const int s = st->active[r];
            v = (int4){s, st->t[s], st->prev_tok[s], st->rank[s]};
            mb = st->mem_blk[s];
        }
        st->rowv[r] = v;
        st->row_mem[r] = mb;
    }
}

__global__ __launch_bounds__(256) void beam_pick_kernel(DecState* st, BeamBuffers bm, const float* hidden, int* etok,
                                                        int T, int V, int eos) {
    __shared__ int lanc[MAX_BEAM][BEAM_ANC_MAX];
    __shared__ float r_val[4];
    __shared__ int r_idx[4];
    __shared__ float sel_score[MAX_BEAM];
    __shared__ int sel_idx[MAX_BEAM];
    __shared__ int task_src[MAX_BEAM], task_dst[MAX_BEAM];
    __shared__ int n_tasks;
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = bm.K, s0 = img * K;
    if (!st->alive[s0]) return;
    const int t = st->t[s0];                       // every hypothesis of the image is at the same step
    const int max_len = st->max_len[s0];
    const float len_f = (float)(t + 2);            // beam_search.py:99 `step + 1` with step = len(alive_seq)
    const float NEG_INF = -__builtin_inff();
    constexpr int PER = MAX_BEAM * BEAM_LP_STRIDE / 256;
    float val[PER];
    int idx[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int flat = tid + 256 * i;
        val[i] = NEG_INF; idx[i] = 0x7fffffff;
        if (flat < K * V) {
            const int j = flat / V, v = flat - j * V;
            val[i] = (bm.blp[(size_t)(s0 + j) * BEAM_LP_STRIDE + v] + bm.bs->cum[s0 + j]) / len_f;   // :97-100
            idx[i] = flat;
        }
    }
    for (int r = 0; r < K; ++r) {                  // top-K by K arg-max rounds (:69-82), ties -> lowest flat index
        float bv = NEG_INF;
        int bi = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < PER; ++i)
            if (val[i] > bv || (val[i] == bv && idx[i] < bi)) { bv = val[i]; bi = idx[i]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { r_val[wave] = bv; r_idx[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w)
                if (r_val[w] > bv || (r_val[w] == bv && r_idx[w] < bi)) { bv = r_val[w]; bi = r_idx[w]; }
            sel_score[r] = bv; sel_idx[r] = bi;
        }
        __syncthreads();
        const int won = sel_idx[r];
#pragma unroll
        for (int i = 0; i < PER; ++i)
            if (idx[i] == won) { val[i] = NEG_INF; idx[i] = 0x7fffffff; }
    }
    // ancestry of the K parents -> LDS, then every new hypothesis inherits its parent's
    const int na = t + 1;
    for (int e = tid; e < K * na; e += 256) {
        const int j = e / na, tau = e - j * na;
        lanc[j][tau] = bm.anc[(size_t)(s0 + j) * bm.anc_stride + tau];
    }
    __syncthreads();
    for (int e = tid; e < K * na; e += 256) {
        const int i = e / na, tau = e - i * na;
        bm.anc[(size_t)(s0 + i) * bm.anc_stride + tau] = lanc[sel_idx[i] / V][tau];
    }
    if (tid < K) {
        const int slot = s0 + tid, tok = sel_idx[tid] % V;
        const bool fin = tok == eos || t + 1 >= max_len;           // decode_strategy.py:54-56
        bm.anc[(size_t)slot * bm.anc_stride + t + 1] = slot;
        etok[(size_t)slot * T + t] = tok;
        st->prev_tok[slot] = tok; st->t[slot] = t + 1; st->len[slot] = t + 1;
        bm.bs->cum[slot] = fin ? -1e10f : sel_score[tid] * len_f;  // :105, :134
    }
    if (tid == 0) {                                                // update_finished (:131-165)
        BeamState* bs = bm.bs;
        int top = bs->top_fin[img], nh = bs->n_hyps[img], pn = bs->pool_n[img], nt = 0;
        for (int i = 0; i < K; ++i) {
            const int tok = sel_idx[i] % V;
            if (!(tok == eos || t + 1 >= max_len)) continue;
            if (i == 0) top = 1;
            ++nh;
            const float sc = sel_score[i];
            int pos = 0;
            while (pos < pn && bs->pscore[img][bs->order[img][pos]] >= sc) ++pos;    // stable: after equal scores
            if (pos >= bm.n_best) continue;
            int storage;
            if (pn == bm.n_best) storage = bs->order[img][pn - 1];                   // the worst kept one drops out
            else storage = pn++;
            for (int q = (pn < bm.n_best ? pn : bm.n_best) - 1; q > pos; --q) bs->order[img][q] = bs->order[img][q - 1];
            bs->order[img][pos] = storage;
            bs->pscore[img][storage] = sc;
            bs->plen[img][storage] = t + 1;
            task_src[nt] = i; task_dst[nt] = storage; ++nt;
        }
        bs->top_fin[img] = top; bs->n_hyps[img] = nh; bs->pool_n[img] = pn;
        n_tasks = nt;
        if (top && nh >= bm.n_best)                                                   // :152-153: the image is done
            for (int j = 0; j < K; ++j) st->alive[s0 + j] = 0;
    }
    __syncthreads();
    for (int q = 0; q < n_tasks; ++q) {            // materialise the kept hypotheses (ids + decoder outputs)
        const int i = task_src[q], parent = sel_idx[i] / V, tok = sel_idx[i] % V;
        const size_t dst = ((size_t)img * bm.pool_stride + task_dst[q]) * T;
        for (int tau = tid; tau <= t; tau += 256)
            bm.ptok[dst + tau] = tau == t ? tok : etok[(size_t)lanc[parent][tau + 1] * T + tau];
        if (bm.phid)
            for (int e = tid; e < (t + 1) * 64; e += 256) {
                const int tau = e >> 6, c = e & 63;
                ((f32x4*)bm.phid)[(dst + tau) * 64 + c] =
                    ((const f32x4*)hidden)[((size_t)lanc[parent][tau] * T + tau) * 64 + c];
            }
        __syncthreads();
    }
}

__global__ void beam_gather_kernel(BeamBuffers bm, int T, int out_len, int* __restrict__ o_tokens,
                                   int* __restrict__ o_len, float* __restrict__ o_scores, float* __restrict__ o_hidden) {
    const int r = blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
    const size_t row = (size_t)img * bm.n_best + r;
    const bool have = r < bm.bs->pool_n[img];
    const int st = have ? bm.bs->order[img][r] : 0;
    const int n = have ? min(bm.bs->plen[img][st], out_len) : 0;
    if (tid == 0) { o_len[row] = n; o_scores[row] = have ? bm.bs->pscore[img][st] : -__builtin_inff(); }
    const size_t src = ((size_t)img * bm.pool_stride + st) * T;
    for (int i = tid; i < out_len; i += blockDim.x) o_tokens[row * out_len + i] = i < n ? bm.ptok[src + i] : 0;
    if (o_hidden && bm.phid)
        for (int i = tid; i < n * 64; i += blockDim.x)
            ((f32x4*)o_hidden)[row * out_len * 64 + i] = ((const f32x4*)bm.phid)[src * 64 + i];
}

hipError_t beam_enqueue_init(const DecBuffers& b, const BeamBuffers& bm, int max_len, hipStream_t s) {
    hipLaunchKernelGGL(beam_init_kernel, dim3(1), dim3(256), 0, s, b.st, bm, max_len, 1);
    return hipGetLastError();
}

hipError_t beam_enqueue_gather(const DecBuffers& b, const BeamBuffers& bm, int out_len, int* o_tokens, int* o_len,
                               float* o_scores, float* o_hidden, hipStream_t s) {
    hipLaunchKernelGGL(beam_gather_kernel, dim3(bm.n_best, bm.B), dim3(256), 0, s, bm, b.T, out_len, o_tokens, o_len,
                       o_scores, o_hidden);
    return hipGetLastError();
}

}  // namespace mnx
