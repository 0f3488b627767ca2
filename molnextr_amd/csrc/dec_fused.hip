// dec_fused.hip — the greedy decode tick as THREE kernels per decoder layer (round 4).
//
// Reference loop being replaced: components.py:284-320 (decode loop), models/decoder.py:254-279 (one layer:
// LN -> self-attention -> +res -> LN -> context attention -> +res -> feed-forward), onmt MultiHeadedAttention /
// PositionwiseFeedForward for the arithmetic of the three blocks.
//
// Round 3's tick was 8 launches per layer (50 per tick) and every one of them cost ~4.5-6 us at 64 rows whatever it
// computed: a tick is a chain of dependent launches, not a bandwidth problem. Rows are independent through the whole
// stack and so is everything per attention head up to the output projection, so the layer is cut where a cut costs no
// cross-workgroup exchange inside a launch:
//
//   dec_fa_kernel  grid (head, row tile)      x = stream + sum(partials of the previous FFN) [embedding at layer 0];
//                                             LN1 -> q_h, k_h, v_h (96 columns of wqkv) -> self-attention of head h over
//                                             the slot's cache (+ the new key, which never leaves LDS) -> ctx_h . Wo[:, h]^T
//                                             = this head's 32-k PARTIAL of the output projection  -> partial[h]
//   dec_fb_kernel  grid (head, row tile)      x = stream + sum(8 head partials) + bo; LN2 -> q_h (32 columns) -> cross-
//                                             attention of head h over the image's 144 projected memory rows -> partial[h]
//   dec_fc_kernel  grid (ff slice, row tile)  x = stream + sum(8 head partials) + bo2; LN -> 64 hidden units of w_1 ->
//                                             GELU -> their 64-k partial of w_2 -> partial[slice]   (16 slices)
//
// The consumer of a stage sums the producer's partials in ONE fixed tree order (pairwise by index) and adds the bias and
// the residual stream; the workgroup with blockIdx.x == 0 writes the summed stream for the stage after it (two stream
// buffers and two partial buffers alternate, so no workgroup overwrites what another one of the same launch still reads).
// 18 + begin + head = 20 launches per tick instead of 50.
//
// Arithmetic is defined per ELEMENT, not per thread mapping: every dot product is a fixed set of fmaf / MFMA chains
// combined in a fixed order, every softmax denominator is 16 strided chains + a butterfly, every P.V is 8 strided chains +
// a butterfly, whatever the row tile R (4, 8 or 16 rows per workgroup) — a row's results do not depend on the tile size
// the host picked for the tick's capacity, nor on which other rows share its tile.
//
// All matrix work is v_mfma_f32_16x16x4_f32 (exact fp32 chains); with R < 16 the spare rows of the 16-row tile repeat
// rows 0..R-1 and are dropped.
#include "common.h"
#include "kernels.h"
#include "dec_types.h"

namespace mnx {

constexpr int FXS = 264;    // LDS row stride (floats) of 256-wide rows; strides = 8 (mod 32) make the 16-lane groups of a
constexpr int FHS = 40;     // ds_read_b128 fragment read hit 64 distinct banks (32-wide rows)
constexpr int FFS = 72;     // (64-wide rows)
constexpr int PS_SELF = 512, PS_CROSS = 160;     // score row length (floats): T <= 511 keys, 144 memory rows
constexpr float QSCALE = 0.17677669529663687f;   // 1 / sqrt(32): onmt scales the query before QK^T
constexpr int FF_SLICE = 64;                      // hidden units per dec_fc workgroup

struct FusedArgs {
    const DecState* st;
    const float* xin;        // residual stream read by this stage [rows, 256] (unused by the embedding form)
    float* xout;             // summed stream, written by the blockIdx.x == 0 workgroups
    const float* part_in;    // [NP][part_stride] partials of the previous stage
    const float* bias_in;    // [256] bias of the linear whose partials those are
    float* part_out;         // [8 | 16][part_stride]
    int part_stride;         // floats between two partial planes (slots * 256)
    const float *gamma, *beta;
    // dec_fa
    const float *wqkv, *bqkv, *wo;
    float *kcache, *vcache;  // this layer's self K / V cache [slots, heads, T, 32]
    const float *emb, *pe;
    // dec_fb
    const float *wq2, *bq2, *wo2;
    const float* memk;       // this layer's memory keys: block b, head h at memk + b * mem_stride + h * S * 32; values S * 256 behind
    long long mem_stride;
    int S;
    // dec_fc
    const float *w1, *b1, *w2;
    int dff;
    int T, heads;
};

__device__ __forceinline__ f32x4 ldg4(const float* p) { return *(const f32x4*)p; }

template <int NP>
__device__ __forceinline__ f32x4 tree_sum(f32x4 (&p)[NP]) {     // ((p0 + p1) + (p2 + p3)) + ...
#pragma unroll
    for (int w = 1; w < NP; w *= 2)
#pragma unroll
        for (int i = 0; i < NP; i += 2 * w) p[i] += p[i + w];
    return p[0];
}

// canonical 32-long dot product: four interleaved fmaf chains (element e of the quad j goes to chain e), (s0+s1)+(s2+s3)
__device__ __forceinline__ float dot32(const f32x4 (&q)[8], const f32x4 (&k)[8]) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        s0 = fmaf(q[i][0], k[i][0], s0); s1 = fmaf(q[i][1], k[i][1], s1);
        s2 = fmaf(q[i][2], k[i][2], s2); s3 = fmaf(q[i][3], k[i][3], s3);
    }
    return (s0 + s1) + (s2 + s3);
}

// ---- stage prologue: x = stream (+ tree(partials) + bias) or embedding; stream out; LayerNorm(eps 1e-6) -> xs[r][FXS] ----
// One wave per row at a time, lane c owns columns 4c..4c+3.
template <int R, int NP, bool EMB>
__device__ __forceinline__ void fused_prologue(const FusedArgs& a, int row0, int n_act, bool writer, float* xs) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4 g = ldg4(a.gamma + lane * 4), be = ldg4(a.beta + lane * 4);
    f32x4 bi = {0.f, 0.f, 0.f, 0.f};
    if (NP > 0) bi = ldg4(a.bias_in + lane * 4);
    constexpr int RW = R / 4;                 // rows per wave
    constexpr int UB = RW < 2 ? RW : 2;       // rows whose loads are in flight together
#pragma unroll 1
    for (int i0 = 0; i0 < RW; i0 += UB) {
        f32x4 v[UB];
        f32x4 p[UB][NP > 0 ? NP : 1];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int row = row0 + wave + 4 * (i0 + u);
            if (EMB) {
                // x0 = E[tok] * sqrt(256) + pe[rank]   (reference components.py:290, embedding.py:52-59)
                const int4 rv = a.st->rowv[row];
                v[u] = ldg4(a.emb + (size_t)rv.z * 256 + lane * 4) * 16.0f + ldg4(a.pe + (size_t)rv.w * 256 + lane * 4);
            } else {
                v[u] = ldg4(a.xin + (size_t)row * 256 + lane * 4);
#pragma unroll
                for (int z = 0; z < NP; ++z) p[u][z] = ldg4(a.part_in + (size_t)z * a.part_stride + (size_t)row * 256 + lane * 4);
            }
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int r = wave + 4 * (i0 + u), row = row0 + r;
            f32x4 x = v[u];
            if (!EMB && NP > 0) x = x + (tree_sum<(NP > 0 ? NP : 1)>(p[u]) + bi);
            if (writer && row < n_act) *(f32x4*)(a.xout + (size_t)row * 256 + lane * 4) = x;
            const float mean = wave_sum((x[0] + x[1]) + (x[2] + x[3])) * (1.0f / 256.0f);
            x -= mean;
            const float var = wave_sum((x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3])) * (1.0f / 256.0f);
            const f32x4 o = x * rsqrtf(var + 1e-6f) * g + be;
            *(f32x4*)(xs + r * FXS + lane * 4) = o;
        }
    }
}

// ---- weight rows [32 rows x 256 k] x NB blocks -> registers -> LDS ws[32 b + lrow][FXS] ----
// thread (lrow = tid >> 3, part = tid & 7): eight lanes cover one 128-byte segment of a row per load instruction
template <int NB>
__device__ __forceinline__ void wload256(f32x4 (&wv)[NB][8], const float* W, const int (&rowbase)[NB]) {
    const int lrow = threadIdx.x >> 3, part = threadIdx.x & 7;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const float* src = W + (size_t)(rowbase[b] + lrow) * 256 + part * 4;
#pragma unroll
        for (int j = 0; j < 8; ++j) wv[b][j] = ldg4(src + 32 * j);
    }
}
template <int NB>
__device__ __forceinline__ void wstore256(const f32x4 (&wv)[NB][8], float* ws) {
    const int lrow = threadIdx.x >> 3, part = threadIdx.x & 7;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int j = 0; j < 8; ++j) *(f32x4*)(ws + (32 * b + lrow) * FXS + part * 4 + 32 * j) = wv[b][j];
}

// ---- out[r][c] = xs[r][:] . ws[c][:] over K = 256: wave w multiplies k in [64 w, 64 w + 64) as 16 MFMA k-steps
// (k-slot g of step j of chunk kc <-> k = 64 w + 16 kc + 4 g + j on both operands); the four waves' chains meet in `red`.
template <int NT, int R>
__device__ __forceinline__ void mfma_k256(const float* xs, const float* ws, f32x4 (&acc)[NT]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fg = lane >> 4;
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
        const int kb = wave * 64 + kc * 16 + fg * 4;
        const f32x4 av = *(const f32x4*)(xs + (fr & (R - 1)) * FXS + kb);
        f32x4 bv[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) bv[n] = *(const f32x4*)(ws + (16 * n + fr) * FXS + kb);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[n][j], acc[n], 0, 0, 0);
    }
}
// D layout: lane (fr, fg) holds rows 4 fg + i, column fr of every n-tile
template <int NT, int R>
__device__ __forceinline__ void red_store(const f32x4 (&acc)[NT], float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fg = lane >> 4;
    constexpr int RS = NT * 16 + 4;
    if (fg * 4 < R) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) red[(wave * R + fg * 4 + i) * RS + n * 16 + fr] = acc[n][i];
    }
}
template <int NT, int R>
__device__ __forceinline__ float red_get(const float* red, int r, int c) {
    constexpr int RS = NT * 16 + 4;
    return (red[(0 * R + r) * RS + c] + red[(1 * R + r) * RS + c]) + (red[(2 * R + r) * RS + c] + red[(3 * R + r) * RS + c]);
}

// ---- [256 n x KW k] weight slice (k columns k0..k0+KW of a [256, ldw] matrix) -> registers -> LDS [n][STR] ----
template <int KW>
struct SliceRegs { f32x4 v[KW / 4]; };      // KW = 32: 8 quads per thread, KW = 64: 16
template <int KW>
__device__ __forceinline__ void sload(SliceRegs<KW>& s, const float* W, int ldw, int k0) {
    constexpr int LPR = KW / 4;              // lanes per row
    constexpr int RPI = 256 / LPR;           // rows per load instruction of the workgroup
    const int c4 = threadIdx.x % LPR, n0 = threadIdx.x / LPR;
#pragma unroll
    for (int j = 0; j < 256 / RPI; ++j) s.v[j] = ldg4(W + (size_t)(n0 + RPI * j) * ldw + k0 + c4 * 4);
}
template <int KW, int STR>
__device__ __forceinline__ void sstore(const SliceRegs<KW>& s, float* dst) {
    constexpr int LPR = KW / 4, RPI = 256 / LPR;
    const int c4 = threadIdx.x % LPR, n0 = threadIdx.x / LPR;
#pragma unroll
    for (int j = 0; j < 256 / RPI; ++j) *(f32x4*)(dst + (n0 + RPI * j) * STR + c4 * 4) = s.v[j];
}

// ---- partial[row][n] = sum_{k < KW} in[r][k] * wsl[n][k]: wave w owns columns [64 w, 64 w + 64), one chain of KW / 4 steps ----
template <int KW, int STR, int R>
__device__ __forceinline__ void mfma_slice_store(const float* in /*[R][STR]*/, const float* wsl /*[256][STR]*/, float* out,
                                                 int row0, int n_act) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fg = lane >> 4;
    f32x4 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < KW / 16; ++kc) {
        const int kb = kc * 16 + fg * 4;
        const f32x4 av = *(const f32x4*)(in + (fr & (R - 1)) * STR + kb);
        f32x4 bv[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) bv[n] = *(const f32x4*)(wsl + (64 * wave + 16 * n + fr) * STR + kb);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[n][j], acc[n], 0, 0, 0);
    }
    if (fg * 4 < R) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = row0 + fg * 4 + i;
            if (row < n_act) {
#pragma unroll
                for (int n = 0; n < 4; ++n) out[(size_t)row * 256 + 64 * wave + 16 * n + fr] = acc[n][i];
            }
        }
    }
}

// ---- single-query attention of R rows x one head. L = 256 / R lanes per row.
//   scores: lane li takes keys li, li + L, ...            (dot32, one 128-byte key row per lane)
//   denominator: 16 chains (keys = c mod 16, ascending) + butterfly 8, 4, 2, 1
//   P.V: 8 chains (keys = g mod 8, ascending; the new key of self-attention is the last element of its chain)
//        + butterfly over g bit 0, 1, 2;   ctx = o * (1 / sum)
// Kb / Vb: the row's cached keys / values [ncache][32]; knew / vnew (self): this step's key / value in LDS.
template <int R, bool CROSS, int KP, int VP>
struct AttnPre {
    static constexpr int L = 256 / R, NG = L / 8, M = 8 / NG;
    f32x4 k[KP][8];
    f32x4 v[M][VP];
};

template <int R, bool CROSS, int KP, int VP>
__device__ __forceinline__ void attn_prefetch(AttnPre<R, CROSS, KP, VP>& pre, const float* Kb, const float* Vb, int ncache) {
    constexpr int L = 256 / R, NG = L / 8, M = 8 / NG;
    const int li = threadIdx.x % L, kgl = li >> 3, dq = li & 7;
    const int last = ncache > 0 ? ncache - 1 : 0;       // clamped: always a valid cache row, unused beyond ncache
#pragma unroll
    for (int p = 0; p < KP; ++p) {
        const int key = min(p * L + li, last);
#pragma unroll
        for (int i = 0; i < 8; ++i) pre.k[p][i] = ldg4(Kb + (size_t)key * 32 + i * 4);
    }
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
        for (int i = 0; i < VP; ++i) {
            const int key = min(kgl + NG * m + 8 * i, last);
            pre.v[m][i] = ldg4(Vb + (size_t)key * 32 + dq * 4);
        }
}

template <int R, bool CROSS, int KP, int VP, int PS>
__device__ __forceinline__ void attn_rows(const AttnPre<R, CROSS, KP, VP>& pre, const float* Kb, const float* Vb, int ncache,
                                          const float* qs, const float* ks, const float* vs, float* ps_all, float* cs) {
    constexpr int L = 256 / R, NG = L / 8, M = 8 / NG;
    const int rl = threadIdx.x / L, li = threadIdx.x % L, kgl = li >> 3, dq = li & 7;
    float* ps = ps_all + rl * PS;
    const int nkeys = CROSS ? ncache : ncache + 1;
    f32x4 q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = *(const f32x4*)(qs + rl * 32 + i * 4);
    float mx = -3.0e38f;
#pragma unroll
    for (int p = 0; p < KP; ++p) {
        const int key = p * L + li;
        if (key < ncache) {
            const float s = dot32(q, pre.k[p]);
            ps[key] = s;
            mx = fmaxf(mx, s);
        }
    }
#pragma unroll 2
    for (int key = KP * L + li; key < ncache; key += L) {
        f32x4 kv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) kv[i] = ldg4(Kb + (size_t)key * 32 + i * 4);
        const float s = dot32(q, kv);
        ps[key] = s;
        mx = fmaxf(mx, s);
    }
    if (!CROSS && li == 0) {                 // the key of this step: never left the workgroup
        f32x4 kv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) kv[i] = *(const f32x4*)(ks + rl * 32 + i * 4);
        const float s = dot32(q, kv);
        ps[ncache] = s;
        mx = fmaxf(mx, s);
    }
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    for (int key = li; key < ncache; key += L) ps[key] = expf(ps[key] - mx);
    if (!CROSS && li == 0) ps[ncache] = expf(ps[ncache] - mx);
    __syncthreads();
    float sum = 0.f;
    for (int key = li & 15; key < nkeys; key += 16) sum += ps[key];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    f32x4 o[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const int vg = kgl + NG * m;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < VP; ++i) {
            const int key = vg + 8 * i;
            if (key < ncache) {
                const float pk = ps[key];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = fmaf(pre.v[m][i][e], pk, acc[e]);
            }
        }
#pragma unroll 4
        for (int key = vg + 8 * VP; key < ncache; key += 8) {
            const f32x4 vv = ldg4(Vb + (size_t)key * 32 + dq * 4);
            const float pk = ps[key];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = fmaf(vv[e], pk, acc[e]);
        }
        if (!CROSS && vg == (ncache & 7)) {
            const f32x4 vv = *(const f32x4*)(vs + rl * 32 + dq * 4);
            const float pk = ps[ncache];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = fmaf(vv[e], pk, acc[e]);
        }
        o[m] = acc;
    }
    // butterfly over the chain index g = kgl + NG m: bit by bit from the lowest (lane bits first, then registers)
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
        for (int sh = 8; sh < L; sh <<= 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) o[m][e] += __shfl_xor(o[m][e], sh, 64);
#pragma unroll
    for (int w = 1; w < M; w *= 2)
#pragma unroll
        for (int m = 0; m < M; m += 2 * w) o[m] += o[m + w];
    if (kgl == 0) *(f32x4*)(cs + rl * FHS + dq * 4) = o[0] * (1.0f / sum);
}

// LDS plan (floats) of the three kernels
template <int R> struct FaLds {
    static constexpr int xs = 0, small = xs + R * FXS;            // small: qs, ks, vs [R][32], cs [R][FHS]
    static constexpr int qs = small, ks = qs + R * 32, vs = ks + R * 32, cs = vs + R * 32;
    static constexpr int ws = cs + R * FHS;                       // [96][FXS]; after the qkv MFMAs: red | wos | ps
    static constexpr int red = ws, wos = red + 4 * R * 100, ps = wos + 256 * FHS;
    static constexpr int end_a = ws + 96 * FXS, end_b = ps + R * PS_SELF;
    static constexpr int total = end_a > end_b ? end_a : end_b;
};
template <int R> struct FbLds {
    static constexpr int xs = 0, qs = xs + R * FXS, cs = qs + R * 32;
    static constexpr int wos = cs + R * FHS;                      // [256][FHS]
    static constexpr int ws = wos + 256 * FHS;                    // [32][FXS]; after the q MFMAs: red | ps
    static constexpr int red = ws, ps = red + 4 * R * 36;
    static constexpr int end_a = ws + 32 * FXS, end_b = ps + R * PS_CROSS;
    static constexpr int total = end_a > end_b ? end_a : end_b;
};
template <int R> struct FcLds {
    static constexpr int xs = 0, hs = xs + R * FXS;               // hs [R][FFS]
    static constexpr int ws = hs + R * FFS;                       // [64][FXS]; after the w_1 MFMAs: red | w2s [256][FFS]
    static constexpr int red = ws, w2s = red + 4 * R * 68;
    static constexpr int end_a = ws + 64 * FXS, end_b = w2s + 256 * FFS;
    static constexpr int total = end_a > end_b ? end_a : end_b;
};

// =============================================================================================
// dec_fa: LN1 (+ embedding | + previous FFN partials) -> q, k, v of one head -> self-attention -> Wo partial
//   models/decoder.py:254-262 (input norm, self_attn, drop + residual), onmt MultiHeadedAttention
// =============================================================================================
template <int R, bool EMB>
__global__ __launch_bounds__(256) void dec_fa_kernel(FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef FaLds<R> Ld;
    constexpr int L = 256 / R, KP = 2, VP = (R == 4 ? 8 : R == 8 ? 4 : 2);
    const int tid = threadIdx.x;
    const int h = blockIdx.x, row0 = blockIdx.y * R;
    const int4 rv = a.st->rowv[row0 + tid / L];          // {slot, t, prev_tok, rank} of the row this thread attends for
    const int n_act = a.st->n_active;
    f32x4 wv[3][8];
    const int rb[3] = {32 * h, 256 + 32 * h, 512 + 32 * h};
    wload256<3>(wv, a.wqkv, rb);
    fused_prologue<R, EMB ? 0 : 16, EMB>(a, row0, n_act, h == 0, smem + Ld::xs);
    wstore256<3>(wv, smem + Ld::ws);
    SliceRegs<32> wov;
    sload<32>(wov, a.wo, 256, 32 * h);
    const float* Kb = a.kcache + ((size_t)rv.x * a.heads + h) * a.T * 32;
    const float* Vb = a.vcache + ((size_t)rv.x * a.heads + h) * a.T * 32;
    AttnPre<R, false, KP, VP> pre;
    attn_prefetch<R, false, KP, VP>(pre, Kb, Vb, rv.y);
    __syncthreads();
    f32x4 acc[6];
    mfma_k256<6, R>(smem + Ld::xs, smem + Ld::ws, acc);
    __syncthreads();                                     // every wave is done with ws: it becomes red | wos | ps
    red_store<6, R>(acc, smem + Ld::red);
    sstore<32, FHS>(wov, smem + Ld::wos);
    __syncthreads();
    for (int idx = tid; idx < R * 96; idx += 256) {
        const int r = idx / 96, c = idx - r * 96, part = c >> 5, d = c & 31;
        const float v = red_get<6, R>(smem + Ld::red, r, c) + a.bqkv[part * 256 + 32 * h + d];
        if (part == 0) {
            smem[Ld::qs + r * 32 + d] = v * QSCALE;
        } else {
            smem[(part == 1 ? Ld::ks : Ld::vs) + r * 32 + d] = v;
            const int row = row0 + r;
            if (row < n_act) {                           // append to the slot's cache (for the ticks after this one)
                const int4 rr = a.st->rowv[row];
                float* cache = part == 1 ? a.kcache : a.vcache;
                cache[(((size_t)rr.x * a.heads + h) * a.T + rr.y) * 32 + d] = v;
            }
        }
    }
    __syncthreads();
    attn_rows<R, false, KP, VP, PS_SELF>(pre, Kb, Vb, rv.y, smem + Ld::qs, smem + Ld::ks, smem + Ld::vs, smem + Ld::ps,
                                         smem + Ld::cs);
    __syncthreads();
    mfma_slice_store<32, FHS, R>(smem + Ld::cs, smem + Ld::wos, a.part_out + (size_t)h * a.part_stride, row0, n_act);
}

// =============================================================================================
// dec_fb: stream + self-attention partials; LN2 -> q of one head -> cross-attention over the memory -> Wo2 partial
//   models/decoder.py:264-276 (query norm, context_attn, drop + residual)
// =============================================================================================
template <int R>
__global__ __launch_bounds__(256) void dec_fb_kernel(FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef FbLds<R> Ld;
    constexpr int L = 256 / R, KP = (R == 4 ? 3 : 2), VP = (R == 4 ? 8 : R == 8 ? 4 : 2);
    const int tid = threadIdx.x;
    const int h = blockIdx.x, row0 = blockIdx.y * R;
    const int mb = a.st->row_mem[row0 + tid / L];
    const int n_act = a.st->n_active;
    f32x4 wv[1][8];
    const int rb[1] = {32 * h};
    wload256<1>(wv, a.wq2, rb);
    SliceRegs<32> wov;
    sload<32>(wov, a.wo2, 256, 32 * h);
    const float* Kb = a.memk + (size_t)mb * a.mem_stride + (size_t)h * a.S * 32;
    const float* Vb = Kb + (size_t)a.S * 256;
    AttnPre<R, true, KP, VP> pre;
    attn_prefetch<R, true, KP, VP>(pre, Kb, Vb, a.S);
    fused_prologue<R, 8, false>(a, row0, n_act, h == 0, smem + Ld::xs);
    wstore256<1>(wv, smem + Ld::ws);
    sstore<32, FHS>(wov, smem + Ld::wos);
    __syncthreads();
    f32x4 acc[2];
    mfma_k256<2, R>(smem + Ld::xs, smem + Ld::ws, acc);
    __syncthreads();
    red_store<2, R>(acc, smem + Ld::red);
    __syncthreads();
    for (int idx = tid; idx < R * 32; idx += 256) {
        const int r = idx >> 5, d = idx & 31;
        smem[Ld::qs + r * 32 + d] = (red_get<2, R>(smem + Ld::red, r, d) + a.bq2[32 * h + d]) * QSCALE;
    }
    __syncthreads();
    attn_rows<R, true, KP, VP, PS_CROSS>(pre, Kb, Vb, a.S, smem + Ld::qs, nullptr, nullptr, smem + Ld::ps, smem + Ld::cs);
    __syncthreads();
    mfma_slice_store<32, FHS, R>(smem + Ld::cs, smem + Ld::wos, a.part_out + (size_t)h * a.part_stride, row0, n_act);
}

// =============================================================================================
// dec_fc: stream + cross-attention partials; LN -> 64 hidden units of w_1 -> GELU -> their partial of w_2
//   models/decoder.py:278 (feed_forward), onmt PositionwiseFeedForward (layer_norm, w_1, gelu, w_2, + x)
// =============================================================================================
template <int R>
__global__ __launch_bounds__(256) void dec_fc_kernel(FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef FcLds<R> Ld;
    const int tid = threadIdx.x;
    const int sl = blockIdx.x, row0 = blockIdx.y * R;
    const int n_act = a.st->n_active;
    f32x4 wv[2][8];
    const int rb[2] = {FF_SLICE * sl, FF_SLICE * sl + 32};
    wload256<2>(wv, a.w1, rb);
    fused_prologue<R, 8, false>(a, row0, n_act, sl == 0, smem + Ld::xs);
    wstore256<2>(wv, smem + Ld::ws);
    SliceRegs<64> w2v;
    sload<64>(w2v, a.w2, a.dff, FF_SLICE * sl);
    __syncthreads();
    f32x4 acc[4];
    mfma_k256<4, R>(smem + Ld::xs, smem + Ld::ws, acc);
    __syncthreads();
    red_store<4, R>(acc, smem + Ld::red);
    sstore<64, FFS>(w2v, smem + Ld::w2s);
    __syncthreads();
    for (int idx = tid; idx < R * 64; idx += 256) {
        const int r = idx >> 6, c = idx & 63;
        smem[Ld::hs + r * FFS + c] = gelu_erf(red_get<4, R>(smem + Ld::red, r, c) + a.b1[FF_SLICE * sl + c]);
    }
    __syncthreads();
    mfma_slice_store<64, FFS, R>(smem + Ld::hs, smem + Ld::w2s, a.part_out + (size_t)sl * a.part_stride, row0, n_act);
}

// ---- host side -------------------------------------------------------------------------------
template <typename K>
static hipError_t opt_in(K kern, int bytes) {
    return hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

template <int R>
static hipError_t fused_init_r() {
    hipError_t e = opt_in(dec_fa_kernel<R, true>, FaLds<R>::total * 4);
    if (e == hipSuccess) e = opt_in(dec_fa_kernel<R, false>, FaLds<R>::total * 4);
    if (e == hipSuccess) e = opt_in(dec_fb_kernel<R>, FbLds<R>::total * 4);
    if (e == hipSuccess) e = opt_in(dec_fc_kernel<R>, FcLds<R>::total * 4);
    return e;
}

// once per device (engine creation; never inside a stream capture): every instantiation opts in to its LDS size
hipError_t dec_fused_init() {
    hipError_t e = fused_init_r<4>();
    if (e == hipSuccess) e = fused_init_r<8>();
    if (e == hipSuccess) e = fused_init_r<16>();
    return e;
}

template <int R>
static void fused_layers(const DecWeights& w, const DecBuffers& b, int rows, hipStream_t s) {
    const int D = 256, H = w.heads, T = b.T;
    const dim3 blk(256);
    int stage = 0;      // stage k reads stream k & 1 and partial buffer (k - 1) & 1, writes stream / partials (k + 1) & 1 / k & 1
    float* xb[2] = {b.x, b.x2};
    float* pb[2] = {b.fpart, b.fpart + (size_t)16 * b.slots * D};
    FusedArgs a = {};
    a.st = b.st; a.part_stride = b.slots * D; a.T = T; a.heads = H; a.emb = w.emb; a.pe = w.pe; a.S = b.S; a.dff = w.dff;
    a.mem_stride = (long long)b.S * w.layers * 2 * D;
    for (int l = 0; l < w.layers; ++l) {
        const DecLayerW& Lw = w.L[l];
        // ---- self-attention block
        a.xin = xb[stage & 1]; a.xout = xb[(stage + 1) & 1]; a.part_in = pb[(stage + 1) & 1]; a.part_out = pb[stage & 1];
        a.bias_in = l > 0 ? w.L[l - 1].b2 : nullptr;
        a.gamma = Lw.ln1_g; a.beta = Lw.ln1_b; a.wqkv = Lw.wqkv; a.bqkv = Lw.bqkv; a.wo = Lw.wo;
        a.kcache = b.self_k + (size_t)l * b.slots * H * T * 32;
        a.vcache = b.self_v + (size_t)l * b.slots * H * T * 32;
        if (l == 0) hipLaunchKernelGGL((dec_fa_kernel<R, true>), dim3(H, rows / R), blk, FaLds<R>::total * 4, s, a);
        else hipLaunchKernelGGL((dec_fa_kernel<R, false>), dim3(H, rows / R), blk, FaLds<R>::total * 4, s, a);
        ++stage;
        // ---- context-attention block
        a.xin = xb[stage & 1]; a.xout = xb[(stage + 1) & 1]; a.part_in = pb[(stage + 1) & 1]; a.part_out = pb[stage & 1];
        a.bias_in = Lw.bo; a.gamma = Lw.ln2_g; a.beta = Lw.ln2_b; a.wq2 = Lw.wq2; a.bq2 = Lw.bq2; a.wo2 = Lw.wo2;
        a.memk = b.mem_kv + (size_t)l * 2 * b.S * D;
        hipLaunchKernelGGL((dec_fb_kernel<R>), dim3(H, rows / R), blk, FbLds<R>::total * 4, s, a);
        ++stage;
        // ---- feed-forward block
        a.xin = xb[stage & 1]; a.xout = xb[(stage + 1) & 1]; a.part_in = pb[(stage + 1) & 1]; a.part_out = pb[stage & 1];
        a.bias_in = Lw.bo2; a.gamma = Lw.lnf_g; a.beta = Lw.lnf_b; a.w1 = Lw.w1; a.b1 = Lw.b1; a.w2 = Lw.w2;
        hipLaunchKernelGGL((dec_fc_kernel<R>), dim3(w.dff / FF_SLICE, rows / R), blk, FcLds<R>::total * 4, s, a);
        ++stage;
    }
}

// The 3 x layers kernels of a greedy tick for `rows` rows of capacity (a multiple of 32). Returns the stream buffer and
// the partial buffer the head has to sum (16 partials of the last w_2 + its bias).
hipError_t dec_enqueue_fused_layers(const DecWeights& w, const DecBuffers& b, int rows, int row_tile, hipStream_t s,
                                    const float** x_final, const float** part_final) {
    if (w.dff != 16 * FF_SLICE || w.heads != 8 || b.T + 1 > PS_SELF || b.S > PS_CROSS || (rows % 16) || !b.fpart)
        return hipErrorInvalidValue;
    if (row_tile == 4) fused_layers<4>(w, b, rows, s);
    else if (row_tile == 8) fused_layers<8>(w, b, rows, s);
    else if (row_tile == 16) fused_layers<16>(w, b, rows, s);
    else return hipErrorInvalidValue;
    const int stages = 3 * w.layers;
    *x_final = (stages & 1) ? b.x2 : b.x;
    *part_final = b.fpart + (size_t)((stages - 1) & 1) * 16 * b.slots * 256;
    return hipGetLastError();
}

}  // namespace mnx
