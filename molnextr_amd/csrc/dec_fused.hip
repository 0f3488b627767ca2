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
// Geometry. The two attention stages give every ROW 256 threads (R = 2 or 4 rows per workgroup, 512 / 1024 threads): the
// attention of a row is the old dec_attn_kernel's (key per lane, 32 value chains), so that every K / V row of the slot is
// requested at once; the linear parts of the stage use whatever waves they need. The feed-forward stage has 256 threads
// and RC = 4, 8 or 16 rows. What a stage costs at 64 rows is a chain of ~11 short phases (tools/fused_stamps.py prints
// them from the kernels' own 100 MHz stamps, lab build `make STAMPS=1`), so every load is requested as early as its
// address is known, in the order of need (a wave's loads return in order: what is asked for first is waited for first).
//
// Arithmetic is defined per ELEMENT, not per workgroup shape: every dot product is a fixed set of fmaf / MFMA chains
// combined in a fixed order (K = 256 linears: eight 32-long chains added pairwise by index; the per-head and per-slice
// partials: one chain), the softmax and P.V orders are those of dec_attn_kernel — a row's numbers do not depend on the row
// tiles the host picked for the tick's capacity, nor on which other rows share its tile (tests/test_gpu_parity.py checks
// bit-equality across tiles). The 8-launches-per-layer tick of decoder.hip adds the same products in another order: same
// tokens, log-probs within 1e-5.
//
// Matrix work is v_mfma_f32_4x4x1_16B_f32 (exact fp32; 4 rows x 64 columns per instruction, one k per instruction): with
// 2-4 row tiles a 16-row MFMA tile is 3/4 padding (measured: qkv phase 1.5 -> 0.5 us). Its B operand is "lane l = column
// n0 + l": the weights are kept TRANSPOSED [k][n] (DecLayerW::*_t), so one k of 64 columns is one coalesced 256-byte load
// straight into the operand register — the weights never touch LDS.
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "common.h"
#include "kernels.h"
#include "dec_types.h"
#include "kvq.h"

namespace mnx {

constexpr int FXS = 260;    // LDS row stride (floats) of the activation rows the MFMA's A operand is read from (256-wide,
constexpr int FHS = 36;     // 32-wide, 64-wide): = 4 (mod 32), the four rows a ds_read_b128 touches sit in different banks
constexpr int FFS = 68;
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
    // Weights are read in the TRANSPOSED layout [k][n] (DecLayerW::*_t): lane l of the 4x4x1 MFMA's B operand is column n0 + l,
    // so one k of 64 columns is one coalesced 256-byte load straight into the operand register — no LDS staging, no barrier
    // between "weights arrived" and "weights usable", and the kernels' LDS drops from 80-125 KB to 10-40 KB.
    // dec_fa
    const float *wqkv, *bqkv, *wo;     // wqkv_t [256][768], wo_t [256][256]
    char *kcache, *vcache;   // this layer's self K / V cache: (slot, head) blocks of Tq rows, 24-bit block fixed point (kvq.h)
    int Tq, Sq;              // rows per block of the self cache / of the memory K / V
    float* qbuf;             // mid form: the scaled queries [rows, 256] handed from dec_ma_kernel to dec_mb_kernel
    const float *emb, *pe;
    // dec_fb
    const float *wq2, *bq2, *wo2;      // wq2_t, wo2_t [256][256]
    const char* memk;        // this layer's memory keys: memory block b, head h at memk + b * mem_stride + h * kvq_block_bytes(Sq);
    long long mem_stride;    // the values heads * kvq_block_bytes(Sq) behind (bytes)
    int S;
    // dec_fc
    const float *w1, *b1, *w2;         // w1_t [256][dff], w2_t [dff][256]
    int dff;
    int T, heads;
    // lab aid (MNX_FUSED_STAMPS=<file>): [stage][block][phase] 100 MHz wall-clock stamps of the LAST tick, see tools/fused_stamps.py
    unsigned long long* stamps;
    int stage;
    int row_base;            // first row of the tick branch this launch belongs to
    // xcd != 0: the grid is (row tile slot, unit) instead of (unit, row tile) and slot x maps to the row tile whose 4-row group
    // g satisfies g % 8 == x % 8. Workgroup b runs on XCD b % 8 (observed, for speed only), so every stage then works on a
    // row's planes on the SAME XCD that wrote them: the consumer's plane loads hit that XCD's L2 instead of missing to memory.
    int xcd;
};

// row tile of grid slot x of nt slots, tiles of R rows: consecutive 4-row groups on consecutive XCDs (nt * R / 4 a multiple of 8)
template <int R>
__device__ __forceinline__ int xcd_tile(int x, int nt) {
    if (R >= 4) return x;
    constexpr int Q = 4 / R;                 // tiles per 4-row group
    const int groups = nt / Q;
    return Q * (x % groups) + x / groups;
}
constexpr int STAMP_BLOCKS = 512, STAMP_PHASES = 12;
#ifdef MNX_FUSED_STAMPS     // lab build only (make STAMPS=1): the stamps cost registers in kernels that have none to spare
#define FSTAMP(ph)                                                                                                        \
    do {                                                                                                                  \
        if (a.stamps && threadIdx.x == 0) {                                                                               \
            const int bid_ = blockIdx.y * gridDim.x + blockIdx.x;                                                         \
            if (bid_ < STAMP_BLOCKS) a.stamps[((size_t)a.stage * STAMP_BLOCKS + bid_) * STAMP_PHASES + (ph)] = __builtin_amdgcn_s_memrealtime(); \
        }                                                                                                                 \
    } while (0)
#else
#define FSTAMP(ph) do { } while (0)
#endif

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
// One wave per row at a time (row r on wave r mod NW), lane c owns columns 4c..4c+3. after_issue(): called once, after the
// first rows' loads have been requested.
template <int R, int NW, int NP, bool EMB, int UBMAX = 2, typename F>
__device__ __forceinline__ bool fused_prologue(const FusedArgs& a, int row0, int n_act, bool writer, float* xs, F&& after_issue) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int RW = (R + NW - 1) / NW;     // rows per wave
    constexpr int UB = RW < UBMAX ? RW : UBMAX;   // rows whose loads are in flight together
    if (wave >= R) {                          // (wave-uniform; no barrier inside)
        if (row0 >= n_act) return false;
        after_issue();
        return true;
    }
    const f32x4 g = ldg4(a.gamma + lane * 4), be = ldg4(a.beta + lane * 4);
    f32x4 bi = {0.f, 0.f, 0.f, 0.f};
    if (NP > 0) bi = ldg4(a.bias_in + lane * 4);
#pragma unroll 1
    for (int i0 = 0; i0 < RW; i0 += UB) {
        f32x4 v[UB];
        f32x4 p[UB][NP > 0 ? NP : 1];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int row = row0 + wave + NW * (i0 + u);
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
        if (i0 == 0) {                        // the kernel's other requests go BEHIND the stream's: first asked, first waited for
            if (row0 >= n_act) return false;  // a tile of dummy rows (capacity > alive rows): the whole workgroup leaves
            after_issue();
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int r = wave + NW * (i0 + u), row = row0 + r;
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
    return true;
}

// The same for the kernels with 256 threads per row: the four waves of a row share the partial planes (wave rw sums the
// index block [rw NP / 4, (rw + 1) NP / 4) pairwise = one subtree of the SAME tree), the subtrees meet in LDS (`psum`
// [R][4][256]) and the row's first wave finishes the sum, writes the stream and normalises. In two halves, so that a kernel
// can put its other requests (weights, K / V rows) BEHIND these loads: a wave's loads return in order, whatever is
// requested first is waited for first.
template <int NP>
struct ProRegs {
    f32x4 x, g, be, bi;
    f32x4 p[NP >= 4 ? NP / 4 : 1];
};
template <int R, int NP, bool EMB>
__device__ __forceinline__ void prologue_issue(const FusedArgs& a, int row0, ProRegs<NP>& pr) {
    const int lane = threadIdx.x & 63, rl = threadIdx.x >> 8, rw = (threadIdx.x >> 6) & 3;
    const int row = row0 + rl;
    pr.x = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (EMB) {
        if (rw == 0) {
            // x0 = E[tok] * sqrt(256) + pe[rank]   (reference components.py:290, embedding.py:52-59)
            const int4 rv = a.st->rowv[row];
            pr.x = ldg4(a.emb + (size_t)rv.z * 256 + lane * 4) * 16.0f + ldg4(a.pe + (size_t)rv.w * 256 + lane * 4);
        }
    } else {
        constexpr int Q = NP >= 4 ? NP / 4 : 1;
#pragma unroll
        for (int z = 0; z < Q; ++z) pr.p[z] = ldg4(a.part_in + (size_t)(rw * Q + z) * a.part_stride + (size_t)row * 256 + lane * 4);
        if (rw == 0) pr.x = ldg4(a.xin + (size_t)row * 256 + lane * 4);
    }
    pr.g = ldg4(a.gamma + lane * 4);
    pr.be = ldg4(a.beta + lane * 4);
    pr.bi = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (!EMB) pr.bi = ldg4(a.bias_in + lane * 4);
}
template <int R, int NP, bool EMB>
__device__ __forceinline__ void prologue_finish(const FusedArgs& a, int row0, int n_act, bool writer, float* xs, float* psum,
                                                ProRegs<NP>& pr) {
    const int lane = threadIdx.x & 63, rl = threadIdx.x >> 8, rw = (threadIdx.x >> 6) & 3;
    const int row = row0 + rl;
    if (!EMB) *(f32x4*)(psum + (rl * 4 + rw) * 256 + lane * 4) = tree_sum<(NP >= 4 ? NP / 4 : 1)>(pr.p);
    __syncthreads();
    if (rw != 0) return;
    f32x4 x = pr.x;
    if (!EMB) {
        const float* pp = psum + rl * 1024 + lane * 4;
        const f32x4 t = (*(const f32x4*)pp + *(const f32x4*)(pp + 256)) + (*(const f32x4*)(pp + 512) + *(const f32x4*)(pp + 768));
        x = x + (t + pr.bi);
    }
    if (writer && row < n_act) *(f32x4*)(a.xout + (size_t)row * 256 + lane * 4) = x;
    const float mean = wave_sum((x[0] + x[1]) + (x[2] + x[3])) * (1.0f / 256.0f);
    x -= mean;
    const float var = wave_sum((x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3])) * (1.0f / 256.0f);
    *(f32x4*)(xs + rl * FXS + lane * 4) = x * rsqrtf(var + 1e-6f) * pr.g + pr.be;
}

// ---- the 4-row matrix instruction: v_mfma_f32_4x4x1_16B_f32 = 16 independent 4x4 outer products. Lane l supplies
// A = x[row l & 3][k] (every block gets the same four rows) and B = W[n0 + l][k]; accumulator register i of lane l is
// out[row i][n0 + l]: one instruction multiplies 4 rows by 64 columns for ONE k, at a quarter of the 16x16x4 cost per row
// tile — the decode tick's row tiles are 2-4 rows, a 16-row MFMA tile would be 3/4 padding. A chain over k is a
// sequential exact-fp32 fmaf chain in program order: the same numbers whatever the row tile.
// The B operands of a chain are its lane's column of the transposed weights: NK dwords, requested (bload) long before the
// chain runs. x: LDS [R][XSTR] (+ k offset); rows beyond R repeat rows 0..R-1.
template <int NK>
__device__ __forceinline__ void bload(float (&b)[NK], const float* wt, int ldw) {
#pragma unroll
    for (int k = 0; k < NK; ++k) b[k] = wt[(size_t)k * ldw];
}
template <int R, int XSTR, int NK>
__device__ __forceinline__ void chain4(const float* x, const float (&b)[NK], f32x4 (&acc)[(R + 3) / 4]) {
    constexpr int NRG = (R + 3) / 4, NQ = NK / 4;
    const int lane = threadIdx.x & 63;
    const float* xr = x + ((lane & 3) & (R - 1)) * XSTR;
#pragma unroll
    for (int g = 0; g < NRG; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // the chain is NK dependent instructions: its A operands are read two quads ahead, so that no step waits for LDS
    f32x4 a4[3][NRG];
#pragma unroll
    for (int q = 0; q < 2 && q < NQ; ++q)
#pragma unroll
        for (int g = 0; g < NRG; ++g) a4[q][g] = *(const f32x4*)(xr + g * 4 * XSTR + 4 * q);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (q + 2 < NQ) {
#pragma unroll
            for (int g = 0; g < NRG; ++g) a4[(q + 2) % 3][g] = *(const f32x4*)(xr + g * 4 * XSTR + 4 * (q + 2));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int g = 0; g < NRG; ++g)
                acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[q % 3][g][e], b[4 * q + e], acc[g], 0, 0, 0);
    }
}

// ---- out[r][c] = xs[r][:] . W[c][:] over K = 256 as EIGHT chains (chain kc: k = 32 kc .. 32 kc + 31 ascending), combined
// pairwise by index ((c0 + c1) + (c2 + c3)) + ((c4 + c5) + (c6 + c7)) through `red`. A column block cb of 64 columns has 8
// units; a workgroup of NW waves gives every wave CPW = max(1, 8 NCB / NW) consecutive chains of one column block.
template <int NCOL, int NW>
struct LinUnit {
    static constexpr int NCB = (NCOL + 63) / 64;
    static constexpr int CPW = 8 * NCB > NW ? 8 * NCB / NW : 1;      // chains per wave
    static constexpr int WPB = 8 / CPW;                               // waves per column block
    bool active, valid;
    int kc0, n;            // first chain, local column (< NCOL when valid)
};
template <int NCOL, int NW>
__device__ __forceinline__ LinUnit<NCOL, NW> lin_unit() {
    typedef LinUnit<NCOL, NW> U;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    U u;
    u.active = wave < U::WPB * U::NCB;
    u.kc0 = (wave % U::WPB) * U::CPW;
    u.n = (wave / U::WPB) * 64 + lane;
    u.valid = u.active && u.n < NCOL;
    return u;
}
// the unit's chains: xs [R][FXS], bw = the lane's column of the transposed weights for k in [32 kc0, 32 (kc0 + CPW))
template <int NCOL, int NW, int R>
__device__ __forceinline__ void lin256_chains(const LinUnit<NCOL, NW>& u, const float* xs,
                                              const float (&bw)[32 * LinUnit<NCOL, NW>::CPW], float* red) {
    constexpr int CPW = LinUnit<NCOL, NW>::CPW, RS = NCOL + 4;
    if (!u.active) return;
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
        float b[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) b[k] = bw[32 * j + k];
        f32x4 acc[(R + 3) / 4];
        chain4<R, FXS, 32>(xs + 32 * (u.kc0 + j), b, acc);
        if (u.valid) {
#pragma unroll
            for (int i = 0; i < R; ++i) red[((u.kc0 + j) * R + i) * RS + u.n] = acc[i >> 2][i & 3];
        }
    }
}
template <int NCOL, int R>
__device__ __forceinline__ float red_get(const float* red, int r, int c) {
    constexpr int RS = NCOL + 4;
    const float* p = red + r * RS + c;
    return ((p[0 * R * RS] + p[1 * R * RS]) + (p[2 * R * RS] + p[3 * R * RS])) +
           ((p[4 * R * RS] + p[5 * R * RS]) + (p[6 * R * RS] + p[7 * R * RS]));
}

// ---- partial[row][n] = sum_{k < KW} in[r][k] * W[n][k0 + k], ONE chain (k ascending) per element: wave cb < 4 owns columns
// [64 cb, 64 cb + 64); its B operands (b, loaded by the caller from wt + k0 * 256 + 64 cb + lane) ----
template <int KW, int STR, int R>
__device__ __forceinline__ void slice_mfma_store(const float* in /*[R][STR]*/, const float (&b)[KW], float* out, int row0, int n_act) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= 4) return;
    f32x4 acc[(R + 3) / 4];
    chain4<R, STR, KW>(in, b, acc);
#pragma unroll
    for (int i = 0; i < R; ++i)
        if (row0 + i < n_act) out[(size_t)(row0 + i) * 256 + 64 * wave + lane] = acc[i >> 2][i & 3];
}

// ---- single-query attention of one head for R rows, 256 threads (4 waves) per row — the arithmetic of round 3's
// dec_attn_kernel, so that a row's numbers do not depend on R:
//   scores: thread rt of the row takes keys rt and rt + 256 (dot32, one 128-byte key row per lane)
//   denominator: per-thread p(rt) + p(rt + 256), wave butterfly, (w0 + w1) + (w2 + w3)
//   P.V: 32 chains (wave rw, lane group kg: keys 8 rw + kg + 32 i ascending), butterfly over kg, (w0 + w1) + (w2 + w3);
//        ctx = o * (1 / sum)
// Kb / Vb: the row's cached keys / values [ncache][32]. Self-attention: this step's key / value (position ncache) never left
// the workgroup: ks / vs in LDS. Everything that does not depend on this tick's activations is requested by attn_prefetch
// before the workgroup waits for anything: key rt, and the first VP value rows of the thread's chain.
template <int VP>
struct AttnPre {
    KvqK k;              // key row rt as fetched (decoded after the wait)
    KvqV v[VP];
};

template <int VP>
__device__ __forceinline__ void attn_prefetch_k(AttnPre<VP>& pre, const char* Kb, int nkb, int ncache) {
    const int rt = threadIdx.x & 255;
    const int key = min(rt, ncache > 0 ? ncache - 1 : 0);   // clamped: always a row of the slot's cache, unused beyond ncache
    kvq_fetch_k(pre.k, Kb, nkb, key);
}
template <int VP>
__device__ __forceinline__ void attn_prefetch_v(AttnPre<VP>& pre, const char* Vb, int nkb, int ncache) {
    const int rt = threadIdx.x & 255, rw = rt >> 6, kg = (rt & 63) >> 3, dq = rt & 7;
    const int last = ncache > 0 ? ncache - 1 : 0;
#pragma unroll
    for (int i = 0; i < VP; ++i) kvq_fetch_v(pre.v[i], Vb, nkb, min(rw * 8 + kg + 32 * i, last), dq);
}

struct AttnLds { float *qs, *ks, *vs, *ps, *cs, *redm, *reds, *po; };   // per-kernel LDS arrays, all [R][...]

// MODE 0: self-attention, this step's key / value (position ncache) in LDS (dec_fa_kernel); 1: cross-attention over ncache
// memory rows; 2: self-attention with every key / value in the cache, ncache = t + 1 (dec_mb_kernel: the row of this step
// was appended by the launch before). 0 and 2 evaluate the same chains on the same numbers.
template <int R, int MODE, int VP, int PS>
__device__ __forceinline__ void attn_rows(const AttnPre<VP>& pre, const char* Kb, const char* Vb, int nkb, int ncache, const AttnLds& m,
                                          const FusedArgs& a) {
    constexpr bool CROSS = MODE == 1, NEWLDS = MODE == 0;
    const int rl = threadIdx.x >> 8, rt = threadIdx.x & 255, lane = rt & 63, rw = rt >> 6, kg = lane >> 3, dq = lane & 7;
    float* ps = m.ps + rl * PS;
    const int nkeys = NEWLDS ? ncache + 1 : ncache;
    f32x4 q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = *(const f32x4*)(m.qs + rl * 32 + i * 4);
    constexpr int NJ = CROSS ? 1 : 2;        // keys per thread: the memory has <= PS_CROSS (< 256) rows
    // the thread's second key row (rt + 256; rows past 256 cached keys only) is requested before the first one is multiplied —
    // where there are registers for it (R = 4: 1024 threads, 128 registers each: fetched when it is needed)
    constexpr bool K2_EARLY = R < 4;
    KvqK k2;
    if (K2_EARLY && NJ == 2 && ncache > 256) kvq_fetch_k(k2, Kb, nkb, min(rt + 256, ncache - 1));       // (uniform per row)
    float sc[NJ];
    float mx = -3.0e38f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int key = rt + 256 * j;
        float s = -3.0e38f;
        if (key < ncache) {
            if (!K2_EARLY && j == 1) kvq_fetch_k(k2, Kb, nkb, key);
            s = kvq_dot32(q, j == 0 ? pre.k : k2);      // dot32 on the row's integers x its power-of-two scale (exact)
        } else if (NEWLDS && key == ncache) {
            f32x4 kv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) kv[i] = *(const f32x4*)(m.ks + rl * 32 + i * 4);   // this step's key, already rounded through kvq_quant
            s = dot32(q, kv);
        }
        sc[j] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    if (lane == 0) m.redm[rl * 4 + rw] = mx;
    __syncthreads();
    FSTAMP(7);
    mx = fmaxf(fmaxf(m.redm[rl * 4 + 0], m.redm[rl * 4 + 1]), fmaxf(m.redm[rl * 4 + 2], m.redm[rl * 4 + 3]));
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int key = rt + 256 * j;
        const float p = key < nkeys ? expf(sc[j] - mx) : 0.f;
        if (key < PS) ps[key] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    if (lane == 0) m.reds[rl * 4 + rw] = sum;
    __syncthreads();
    FSTAMP(8);
    sum = (m.reds[rl * 4 + 0] + m.reds[rl * 4 + 1]) + (m.reds[rl * 4 + 2] + m.reds[rl * 4 + 3]);
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < VP; ++i) {
        const int key = rw * 8 + kg + 32 * i;
        if (key < ncache) {
            const float pk = ps[key] * pre.v[i].sc;          // the row's scale folded into the probability (exact)
            const f32x4 vv = kvq_decode_v(pre.v[i]);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaf(vv[e], pk, o[e]);
        } else if (NEWLDS && key == ncache) {
            const f32x4 vv = *(const f32x4*)(m.vs + rl * 32 + dq * 4);
            const float pk = ps[key];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaf(vv[e], pk, o[e]);
        }
    }
#pragma unroll 8
    for (int key = rw * 8 + kg + 32 * VP; key < nkeys; key += 32) {
        f32x4 vv;
        float pk = ps[key];
        if (NEWLDS && key == ncache) {
            vv = *(const f32x4*)(m.vs + rl * 32 + dq * 4);
        } else {
            KvqV vr;
            kvq_fetch_v(vr, Vb, nkb, key, dq);
            vv = kvq_decode_v(vr);
            pk *= vr.sc;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaf(vv[e], pk, o[e]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        o[e] += __shfl_xor(o[e], 8, 64);
        o[e] += __shfl_xor(o[e], 16, 64);
        o[e] += __shfl_xor(o[e], 32, 64);
    }
    if (lane < 8) *(f32x4*)(m.po + (rl * 4 + rw) * 32 + dq * 4) = o;
    __syncthreads();
    FSTAMP(9);
    if (rt < 8) {
        const float* pp = m.po + rl * 128 + rt * 4;
        const f32x4 r = (*(const f32x4*)pp + *(const f32x4*)(pp + 32)) + (*(const f32x4*)(pp + 64) + *(const f32x4*)(pp + 96));
        *(f32x4*)(m.cs + rl * FHS + rt * 4) = r * (1.0f / sum);
    }
}

// LDS plan (floats) of the three kernels: activations only (the weights never touch LDS)
template <int R> struct FaLds {
    static constexpr int xs = 0;
    static constexpr int qs = xs + R * FXS, ks = qs + R * 32, vs = ks + R * 32, cs = vs + R * 32;   // [R][32] x3, [R][FHS]
    static constexpr int redm = cs + R * FHS, reds = redm + R * 4, po = reds + R * 4;              // [R][4] x2, [R][4][32]
    static constexpr int psum = po + R * 128;                     // [R][4][256] subtrees of the prologue
    static constexpr int red = psum + R * 1024;                   // [8][R][96 + 4]
    static constexpr int ps = red + 8 * R * 100;                  // [R][PS_SELF]
    static constexpr int total = ps + R * PS_SELF;                // 44.8 KB at R = 4
};
template <int R> struct FbLds {
    static constexpr int xs = 0, qs = xs + R * FXS, cs = qs + R * 32;
    static constexpr int redm = cs + R * FHS, reds = redm + R * 4, po = reds + R * 4;
    static constexpr int psum = po + R * 128;
    static constexpr int red = psum + R * 1024;                   // [8][R][32 + 4]
    static constexpr int ps = red + 8 * R * 36;                   // [R][PS_CROSS]
    static constexpr int total = ps + R * PS_CROSS;               // 30.3 KB at R = 4
};
template <int R> struct FcLds {
    static constexpr int xs = 0, hs = xs + R * FXS;               // hs [R][FFS]
    static constexpr int red = hs + R * FFS;                      // [8][R][64 + 4]
    static constexpr int total = red + 8 * R * 68;                // 13.7 KB at R = 4
};

// =============================================================================================
// dec_fa: LN1 (+ embedding | + previous FFN partials) -> q, k, v of one head -> self-attention -> Wo partial
//   models/decoder.py:254-262 (input norm, self_attn, drop + residual), onmt MultiHeadedAttention
// 256 threads per row (R rows, 256 R threads): the linear parts use the first 8 (then 4) waves, the attention 4 waves per row.
// =============================================================================================
template <int R, bool EMB>
__global__ __launch_bounds__(256 * R) void dec_fa_kernel(FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef FaLds<R> Ld;
    constexpr int VP = R >= 4 ? 4 : 8;                   // value rows requested ahead (1024 threads: 128 registers each)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = a.xcd ? blockIdx.y : blockIdx.x;
    const int row0 = a.row_base + (a.xcd ? xcd_tile<R>(blockIdx.x, gridDim.x) : (int)blockIdx.y) * R;
    FSTAMP(0);
    const int4 rv = a.st->rowv[row0 + (tid >> 8)];       // {slot, t, prev_tok, rank} of the row this thread attends for
    const int n_act = a.st->n_active;
    // requests in the order of need: stream + partials, the unit's 64 weights, this thread's key, then the value rows
    ProRegs<EMB ? 0 : 16> pr;
    prologue_issue<R, EMB ? 0 : 16, EMB>(a, row0, pr);
    if (row0 >= n_act) return;       // a tile of dummy rows (capacity > alive rows): nothing to do. Checked AFTER the stream /
                                     // plane requests are out (n_active takes a round trip; the weights are not needed before LN)
    typedef LinUnit<96, 4 * R> U;
    const U u = lin_unit<96, 4 * R>();
    float bw[32 * U::CPW];
    if (u.active) {                                      // column of wqkv_t: q | k | v of head h
        const int c = u.valid ? u.n : 0;
        bload<32 * U::CPW>(bw, a.wqkv + (size_t)(32 * u.kc0) * 768 + (c >> 5) * 256 + 32 * h + (c & 31), 768);
    }
    // (the four waves of a row share its slot: the block bases are wave-uniform = scalar registers)
    const size_t blk_u = ((size_t)__builtin_amdgcn_readfirstlane(rv.x) * a.heads + h) * kvq_block_bytes(a.Tq);
    const char* Kb = a.kcache + blk_u;
    const char* Vb = a.vcache + blk_u;
    AttnPre<VP> pre;
    attn_prefetch_k<VP>(pre, Kb, a.Tq, rv.y);
    prologue_finish<R, EMB ? 0 : 16, EMB>(a, row0, n_act, h == 0, smem + Ld::xs, smem + Ld::psum, pr);
    FSTAMP(1);
    attn_prefetch_v<VP>(pre, Vb, a.Tq, rv.y);
    FSTAMP(2);
    __syncthreads();                                     // xs complete
    FSTAMP(3);
    lin256_chains<96, 4 * R, R>(u, smem + Ld::xs, bw, smem + Ld::red);
    float bo[32];                                        // the head's slice of wo_t, columns 64 wave + lane (waves 0..3)
    if (R < 4 && wave < 4) bload<32>(bo, a.wo + (size_t)(32 * h) * 256 + 64 * wave + lane, 256);
    FSTAMP(4);
    __syncthreads();
    FSTAMP(5);
    for (int idx = tid; idx < R * 96; idx += 256 * R) {
        const int r = idx / 96, c = idx - r * 96, part = c >> 5, d = c & 31;
        const float v = red_get<96, R>(smem + Ld::red, r, c) + a.bqkv[part * 256 + 32 * h + d];
        // 32 consecutive threads hold the 32 channels of one (row, q | k | v): R * 96 is a multiple of 32 and so is the stride,
        // so the half-wave is never split by the loop bound
        if (part == 0) {
            smem[Ld::qs + r * 32 + d] = v * QSCALE;
        } else {
            // this step's key / value row as the cache will hold it (kvq.h): rounded here, used from LDS by this tick and
            // appended to the slot's cache for the ticks after this one — the same number in both places
            float amax = fabsf(v);
#pragma unroll
            for (int o_ = 16; o_ > 0; o_ >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o_, 64));
            int qi;
            float scale;
            kvq_quant(v, amax, qi, scale);
            smem[(part == 1 ? Ld::ks : Ld::vs) + r * 32 + d] = kvq_value(qi, scale);
            const int row = row0 + r;
            if (row < n_act) {
                const int4 rr = a.st->rowv[row];
                char* blk = (part == 1 ? a.kcache : a.vcache) + ((size_t)rr.x * a.heads + h) * kvq_block_bytes(a.Tq);
                kvq_store1(blk, a.Tq, rr.y, d, qi);
                if (d == 0) kvq_store_scale(blk, a.Tq, rr.y, scale);
            }
        }
    }
    __syncthreads();
    FSTAMP(6);
    const AttnLds m = {smem + Ld::qs, smem + Ld::ks, smem + Ld::vs, smem + Ld::ps, smem + Ld::cs, smem + Ld::redm,
                       smem + Ld::reds, smem + Ld::po};
    attn_rows<R, 0, VP, PS_SELF>(pre, Kb, Vb, a.Tq, rv.y, m, a);
    if (R >= 4 && wave < 4) bload<32>(bo, a.wo + (size_t)(32 * h) * 256 + 64 * wave + lane, 256);   // 128 registers per thread: not earlier
    __syncthreads();
    FSTAMP(10);
    slice_mfma_store<32, FHS, R>(smem + Ld::cs, bo, a.part_out + (size_t)h * a.part_stride, row0, n_act);
    FSTAMP(11);
}

// =============================================================================================
// Mid form (ticks of more than dec_fused_max rows, round 5): dec_fa cut in two where its row tiles want different sizes.
// dec_fa re-reads the head's 96 KB of wqkv and the 17 KB-per-row stream + partial planes for every 2-4 rows: 1.1 us per row and
// tick against the 0.4 of decoder.hip's 32-row linears — at 256+ rows that is more bytes than the K / V rows the
// attention is there for. The linear half wants MANY rows per workgroup, the attention half many (row, head) workgroups:
//   dec_ma_kernel  grid (head, 16-row tile), 512 threads: stream + partials -> LN1 -> q | k | v of the head; q (scaled) ->
//                  qbuf, k / v -> the slot's cache. The weights are read once per 16 rows.
//   dec_mb_kernel  grid (head, R-row tile), 256 R threads: q from qbuf, ALL keys / values from the cache (the row of this
//                  step included) -> attention -> Wo partial: dec_fa's second half.
// dec_fb / dec_fc follow unchanged (their weights are 2-4x lighter per row). Every chain, tree and softmax order is dec_fa's:
// a row's numbers are bit-identical in both forms — which form a tick takes (mnx_predict picks by capacity, and the capacity
// follows host timing) cannot be seen in the results.
// =============================================================================================
struct MaLds {
    static constexpr int R = 16;
    static constexpr int xs = 0, red = xs + R * FXS;              // red [8][R][96 + 4]
    static constexpr int total = red + 8 * R * 100;               // 67.8 KB
};
template <int R> struct MbLds {
    static constexpr int qs = 0, cs = qs + R * 32;
    static constexpr int redm = cs + R * FHS, reds = redm + R * 4, po = reds + R * 4;
    static constexpr int ps = po + R * 128;
    static constexpr int total = ps + R * PS_SELF;                // 11.3 KB at R = 4
};

template <bool EMB>
__global__ __launch_bounds__(512) void dec_ma_kernel(FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef MaLds Ld;
    constexpr int R = Ld::R;
    const int tid = threadIdx.x;
    const int h = blockIdx.x;
    const int row0 = a.row_base + (int)blockIdx.y * R;
    const int n_act = a.st->n_active;
    typedef LinUnit<96, 8> U;                             // 2 column blocks x 8 chains on 8 waves: 2 chains per wave
    const U u = lin_unit<96, 8>();
    static_assert(U::CPW == 2, "two chains per wave");
    // the lane's column of wqkv_t (q | k | v of head h) for its two chains: the first chain's 32 weights are requested behind
    // the first rows' stream / plane loads, the second chain's once the prologue has released its 64 plane registers (they
    // arrive while the first chain runs) — requested all at once the kernel spilled
    const int wc = u.valid ? u.n : 0;
    const float* wcol = a.wqkv + (size_t)(32 * u.kc0) * 768 + (wc >> 5) * 256 + 32 * h + (wc & 31);
    float bw0[32], bw1[32];
    const bool alive = fused_prologue<R, 8, EMB ? 0 : 16, EMB, 1>(a, row0, n_act, h == 0, smem + Ld::xs, [&]() {
        if (u.active) bload<32>(bw0, wcol, 768);
    });
    if (!alive) return;
    if (u.active) bload<32>(bw1, wcol + (size_t)32 * 768, 768);
    float bw[64];
#pragma unroll
    for (int k = 0; k < 32; ++k) { bw[k] = bw0[k]; bw[32 + k] = bw1[k]; }
    __syncthreads();                                      // xs complete
    lin256_chains<96, 8, R>(u, smem + Ld::xs, bw, smem + Ld::red);
    __syncthreads();
    for (int idx = tid; idx < R * 96; idx += 512) {
        const int r = idx / 96, c = idx - r * 96, part = c >> 5, d = c & 31;
        const float v = red_get<96, R>(smem + Ld::red, r, c) + a.bqkv[part * 256 + 32 * h + d];
        const int row = row0 + r;
        // (32 consecutive threads = the 32 channels of one (row, q | k | v); the shuffles below run before any lane leaves)
        float amax = fabsf(v);
#pragma unroll
        for (int o_ = 16; o_ > 0; o_ >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o_, 64));
        if (row >= n_act) continue;
        if (part == 0) {
            a.qbuf[(size_t)row * 256 + 32 * h + d] = v * QSCALE;
        } else {
            const int4 rr = a.st->rowv[row];
            int qi;
            float scale;
            kvq_quant(v, amax, qi, scale);
            char* blk = (part == 1 ? a.kcache : a.vcache) + ((size_t)rr.x * a.heads + h) * kvq_block_bytes(a.Tq);
            kvq_store1(blk, a.Tq, rr.y, d, qi);
            if (d == 0) kvq_store_scale(blk, a.Tq, rr.y, scale);
        }
    }
}

template <int R>
__global__ __launch_bounds__(256 * R) void dec_mb_kernel(FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef MbLds<R> Ld;
    constexpr int VP = R >= 4 ? 4 : 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, rt = tid & 255;
    const int h = blockIdx.x;
    const int row0 = a.row_base + (int)blockIdx.y * R;
    const int4 rv = a.st->rowv[row0 + (tid >> 8)];       // {slot, t, prev_tok, rank}; dummy rows: slot 0, t 0
    const int n_act = a.st->n_active;
    const int nk = rv.y + 1;                             // keys of the row: positions 0 .. t
    const size_t blk_u = ((size_t)__builtin_amdgcn_readfirstlane(rv.x) * a.heads + h) * kvq_block_bytes(a.Tq);
    const char* Kb = a.kcache + blk_u;
    const char* Vb = a.vcache + blk_u;
    AttnPre<VP> pre;
    attn_prefetch_k<VP>(pre, Kb, a.Tq, nk);
    f32x4 qv = {0.f, 0.f, 0.f, 0.f};
    if (rt < 8) qv = ldg4(a.qbuf + (size_t)(row0 + (tid >> 8)) * 256 + 32 * h + rt * 4);
    float bo[32];                                        // the head's slice of wo_t, columns 64 wave + lane (waves 0..3)
    if (R < 4 && wave < 4) bload<32>(bo, a.wo + (size_t)(32 * h) * 256 + 64 * wave + lane, 256);
    attn_prefetch_v<VP>(pre, Vb, a.Tq, nk);
    if (row0 >= n_act) return;                           // a tile of dummy rows (uniform per workgroup)
    if (rt < 8) *(f32x4*)(smem + Ld::qs + (tid >> 8) * 32 + rt * 4) = qv;
    __syncthreads();
    const AttnLds m = {smem + Ld::qs, nullptr, nullptr, smem + Ld::ps, smem + Ld::cs, smem + Ld::redm, smem + Ld::reds,
                       smem + Ld::po};
    attn_rows<R, 2, VP, PS_SELF>(pre, Kb, Vb, a.Tq, nk, m, a);
    if (R >= 4 && wave < 4) bload<32>(bo, a.wo + (size_t)(32 * h) * 256 + 64 * wave + lane, 256);   // 128 registers per thread: not earlier
    __syncthreads();
    slice_mfma_store<32, FHS, R>(smem + Ld::cs, bo, a.part_out + (size_t)h * a.part_stride, row0, n_act);
}

// =============================================================================================
// dec_fb: stream + self-attention partials; LN2 -> q of one head -> cross-attention over the memory -> Wo2 partial
//   models/decoder.py:264-276 (query norm, context_attn, drop + residual)
// =============================================================================================
template <int R>
__global__ __launch_bounds__(256 * R) void dec_fb_kernel(FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef FbLds<R> Ld;
    constexpr int VP = 5;                                // 144 memory rows = 4.5 x 32: every value row is prefetched
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = a.xcd ? blockIdx.y : blockIdx.x;
    const int row0 = a.row_base + (a.xcd ? xcd_tile<R>(blockIdx.x, gridDim.x) : (int)blockIdx.y) * R;
    FSTAMP(0);
    const int mb = __builtin_amdgcn_readfirstlane(a.st->row_mem[row0 + (tid >> 8)]);    // wave-uniform: scalar block bases
    const int n_act = a.st->n_active;
    const char* Kb = a.memk + (size_t)mb * a.mem_stride + (size_t)h * kvq_block_bytes(a.Sq);
    const char* Vb = Kb + (size_t)a.heads * kvq_block_bytes(a.Sq);
    // requests in the order of need: stream + partials, the unit's weights, then the memory rows (nothing of this tick is
    // needed to ask for them, but whatever is requested first is waited for first)
    ProRegs<8> pr;
    prologue_issue<R, 8, false>(a, row0, pr);
    if (row0 >= n_act) return;       // a tile of dummy rows
    typedef LinUnit<32, 4 * R> U;
    const U u = lin_unit<32, 4 * R>();
    float bw[32 * U::CPW];
    if (u.active) bload<32 * U::CPW>(bw, a.wq2 + (size_t)(32 * u.kc0) * 256 + 32 * h + (u.valid ? u.n : 0), 256);
    AttnPre<VP> pre;
    attn_prefetch_k<VP>(pre, Kb, a.Sq, a.S);
    if (R < 4) attn_prefetch_v<VP>(pre, Vb, a.Sq, a.S);
    prologue_finish<R, 8, false>(a, row0, n_act, h == 0, smem + Ld::xs, smem + Ld::psum, pr);
    FSTAMP(1);
    if (R >= 4) attn_prefetch_v<VP>(pre, Vb, a.Sq, a.S);      // 1024 threads = 128 registers each: the value rows wait for the prologue's registers
    FSTAMP(2);
    __syncthreads();
    FSTAMP(3);
    lin256_chains<32, 4 * R, R>(u, smem + Ld::xs, bw, smem + Ld::red);
    float bo[32];
    if (wave < 4) bload<32>(bo, a.wo2 + (size_t)(32 * h) * 256 + 64 * wave + lane, 256);
    FSTAMP(4);
    __syncthreads();
    FSTAMP(5);
    for (int idx = tid; idx < R * 32; idx += 256 * R) {
        const int r = idx >> 5, d = idx & 31;
        smem[Ld::qs + r * 32 + d] = (red_get<32, R>(smem + Ld::red, r, d) + a.bq2[32 * h + d]) * QSCALE;
    }
    __syncthreads();
    FSTAMP(6);
    const AttnLds m = {smem + Ld::qs, nullptr, nullptr, smem + Ld::ps, smem + Ld::cs, smem + Ld::redm, smem + Ld::reds,
                       smem + Ld::po};
    attn_rows<R, 1, VP, PS_CROSS>(pre, Kb, Vb, a.Sq, a.S, m, a);
    __syncthreads();
    FSTAMP(10);
    slice_mfma_store<32, FHS, R>(smem + Ld::cs, bo, a.part_out + (size_t)h * a.part_stride, row0, n_act);
    FSTAMP(11);
}

// =============================================================================================
// dec_fc: stream + cross-attention partials; LN -> 64 hidden units of w_1 -> GELU -> their partial of w_2
//   models/decoder.py:278 (feed_forward), onmt PositionwiseFeedForward (layer_norm, w_1, gelu, w_2, + x)
// =============================================================================================
template <int R>
__global__ __launch_bounds__(256) void dec_fc_kernel(FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef FcLds<R> Ld;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sl = a.xcd ? blockIdx.y : blockIdx.x;
    const int row0 = a.row_base + (a.xcd ? xcd_tile<R>(blockIdx.x, gridDim.x) : (int)blockIdx.y) * R;
    FSTAMP(0);
    const int n_act = a.st->n_active;
    float b1w[64], b2w[64];
    // wave w: chains 2 w, 2 w + 1 of the slice's 64 hidden units (column 64 sl + lane of w1_t); then columns 64 w + lane of w2_t
    const bool alive = fused_prologue<R, 4, 8, false>(a, row0, n_act, sl == 0, smem + Ld::xs, [&]() {
        bload<64>(b1w, a.w1 + (size_t)(64 * wave) * a.dff + FF_SLICE * sl + lane, a.dff);
        bload<64>(b2w, a.w2 + (size_t)(FF_SLICE * sl) * 256 + 64 * wave + lane, 256);
    });
    if (!alive) return;
    FSTAMP(1);
    FSTAMP(2);
    __syncthreads();
    FSTAMP(3);
    const LinUnit<64, 4> u = lin_unit<64, 4>();
    lin256_chains<64, 4, R>(u, smem + Ld::xs, b1w, smem + Ld::red);
    FSTAMP(4);
    __syncthreads();
    FSTAMP(5);
    for (int idx = tid; idx < R * 64; idx += 256) {
        const int r = idx >> 6, c = idx & 63;
        smem[Ld::hs + r * FFS + c] = gelu_erf(red_get<64, R>(smem + Ld::red, r, c) + a.b1[FF_SLICE * sl + c]);
    }
    __syncthreads();
    FSTAMP(10);
    slice_mfma_store<64, FFS, R>(smem + Ld::hs, b2w, a.part_out + (size_t)sl * a.part_stride, row0, n_act);
    FSTAMP(11);
}

// ---- host side -------------------------------------------------------------------------------
template <typename K>
static hipError_t opt_in(K kern, int bytes) {
    return hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

template <int R>
static hipError_t fused_init_ab() {
    hipError_t e = opt_in(dec_fa_kernel<R, true>, FaLds<R>::total * 4);
    if (e == hipSuccess) e = opt_in(dec_fa_kernel<R, false>, FaLds<R>::total * 4);
    if (e == hipSuccess) e = opt_in(dec_fb_kernel<R>, FbLds<R>::total * 4);
    return e;
}

static unsigned long long* g_stamps = nullptr;     // lab aid, see FusedArgs::stamps
static const size_t STAMP_WORDS = (size_t)3 * MAX_DEC_LAYERS * STAMP_BLOCKS * STAMP_PHASES;

// writes the stamps of the last fused tick to `path` (binary u64 [stages][STAMP_BLOCKS][STAMP_PHASES]); tools/fused_stamps.py
void dec_fused_dump_stamps(const char* path) {
    if (!g_stamps || !path) return;
    std::vector<unsigned long long> host(STAMP_WORDS);
    if (hipMemcpy(host.data(), g_stamps, STAMP_WORDS * 8, hipMemcpyDeviceToHost) != hipSuccess) return;
    if (FILE* f = fopen(path, "wb")) {
        fwrite(host.data(), 8, STAMP_WORDS, f);
        fclose(f);
    }
}

// once per device (engine creation; never inside a stream capture): every instantiation opts in to its LDS size
hipError_t dec_fused_init() {
    if (getenv("MNX_FUSED_STAMPS") && !g_stamps) {
        if (hipMalloc((void**)&g_stamps, STAMP_WORDS * 8) != hipSuccess) return hipErrorOutOfMemory;
        (void)hipMemset(g_stamps, 0, STAMP_WORDS * 8);
    }
    hipError_t e = fused_init_ab<2>();
    if (e == hipSuccess) e = fused_init_ab<4>();
    if (e == hipSuccess) e = opt_in(dec_ma_kernel<true>, MaLds::total * 4);
    if (e == hipSuccess) e = opt_in(dec_ma_kernel<false>, MaLds::total * 4);
    if (e == hipSuccess) e = opt_in(dec_mb_kernel<2>, MbLds<2>::total * 4);
    if (e == hipSuccess) e = opt_in(dec_mb_kernel<4>, MbLds<4>::total * 4);
    if (e == hipSuccess) e = opt_in(dec_fc_kernel<4>, FcLds<4>::total * 4);
    if (e == hipSuccess) e = opt_in(dec_fc_kernel<8>, FcLds<8>::total * 4);
    if (e == hipSuccess) e = opt_in(dec_fc_kernel<16>, FcLds<16>::total * 4);
    return e;
}

// R: rows per workgroup of the two attention stages (256 threads per row); RC: rows per workgroup of the feed-forward stage
template <int R, int RC>
static void fused_layers(const DecWeights& w, const DecBuffers& b, int row_base, int rows, int xcd, bool mid, hipStream_t s) {
    const int D = 256, H = w.heads, T = b.T;
    int stage = 0;      // stage k reads stream k & 1 and partial buffer (k - 1) & 1, writes stream / partials (k + 1) & 1 / k & 1
    float* xb[2] = {b.x, b.x2};
    float* pb[2] = {b.fpart, b.fpart + (size_t)16 * b.fpart_rows * D};
    FusedArgs a = {};
    a.st = b.st; a.part_stride = b.fpart_rows * D; a.T = T; a.heads = H; a.emb = w.emb; a.pe = w.pe; a.S = b.S; a.dff = w.dff;
    const size_t self_blk = kvq_block_bytes(b.Tq), mem_blk = kvq_block_bytes(b.Sq);
    a.Tq = b.Tq; a.Sq = b.Sq;
    a.mem_stride = (long long)((size_t)w.layers * 2 * H * mem_blk);
    a.stamps = g_stamps;
    a.row_base = row_base;
    a.xcd = (xcd && !mid && rows % 32 == 0) ? 1 : 0;
    a.qbuf = b.q;
    for (int l = 0; l < w.layers; ++l) {
        const DecLayerW& Lw = w.L[l];
        // ---- self-attention block
        a.stage = stage;
        a.xin = xb[stage & 1]; a.xout = xb[(stage + 1) & 1]; a.part_in = pb[(stage + 1) & 1]; a.part_out = pb[stage & 1];
        a.bias_in = l > 0 ? w.L[l - 1].b2 : nullptr;
        a.gamma = Lw.ln1_g; a.beta = Lw.ln1_b; a.wqkv = Lw.wqkv_t; a.bqkv = Lw.bqkv; a.wo = Lw.wo_t;
        a.kcache = b.self_k + (size_t)l * b.slots * H * self_blk;
        a.vcache = b.self_v + (size_t)l * b.slots * H * self_blk;
        const dim3 gab = a.xcd ? dim3(rows / R, H) : dim3(H, rows / R);
        if (mid) {      // the linear half on 16-row tiles, then the attention half on R-row tiles
            const dim3 gma(H, rows / MaLds::R);
            if (l == 0) hipLaunchKernelGGL((dec_ma_kernel<true>), gma, dim3(512), MaLds::total * 4, s, a);
            else hipLaunchKernelGGL((dec_ma_kernel<false>), gma, dim3(512), MaLds::total * 4, s, a);
            hipLaunchKernelGGL((dec_mb_kernel<R>), gab, dim3(256 * R), MbLds<R>::total * 4, s, a);
        } else if (l == 0) hipLaunchKernelGGL((dec_fa_kernel<R, true>), gab, dim3(256 * R), FaLds<R>::total * 4, s, a);
        else hipLaunchKernelGGL((dec_fa_kernel<R, false>), gab, dim3(256 * R), FaLds<R>::total * 4, s, a);
        ++stage;
        // ---- context-attention block
        a.stage = stage;
        a.xin = xb[stage & 1]; a.xout = xb[(stage + 1) & 1]; a.part_in = pb[(stage + 1) & 1]; a.part_out = pb[stage & 1];
        a.bias_in = Lw.bo; a.gamma = Lw.ln2_g; a.beta = Lw.ln2_b; a.wq2 = Lw.wq2_t; a.bq2 = Lw.bq2; a.wo2 = Lw.wo2_t;
        a.memk = b.mem_kv + (size_t)l * 2 * H * mem_blk;
        hipLaunchKernelGGL((dec_fb_kernel<R>), gab, dim3(256 * R), FbLds<R>::total * 4, s, a);
        ++stage;
        // ---- feed-forward block
        a.stage = stage;
        a.xin = xb[stage & 1]; a.xout = xb[(stage + 1) & 1]; a.part_in = pb[(stage + 1) & 1]; a.part_out = pb[stage & 1];
        a.bias_in = Lw.bo2; a.gamma = Lw.lnf_g; a.beta = Lw.lnf_b; a.w1 = Lw.w1_t; a.b1 = Lw.b1; a.w2 = Lw.w2_t;
        hipLaunchKernelGGL((dec_fc_kernel<RC>), a.xcd ? dim3(rows / RC, w.dff / FF_SLICE) : dim3(w.dff / FF_SLICE, rows / RC), dim3(256),
                           FcLds<RC>::total * 4, s, a);
        ++stage;
    }
}

// The 3 (mid form: 4) x layers kernels of a greedy tick for `rows` rows of capacity (a multiple of 16). row_tile is encoded as
// R * 100 + RC (R in {2, 4}: rows per attention workgroup; RC in {4, 8, 16}: rows per feed-forward workgroup), + 1000: row tiles
// pinned to XCDs, + 2000: the mid form.
// Returns the stream buffer and the partial buffer the head has to sum (16 partials of the last w_2 + its bias).
hipError_t dec_enqueue_fused_layers(const DecWeights& w, const DecBuffers& b, int row_base, int rows, int row_tile, hipStream_t s,
                                    const float** x_final, const float** part_final) {
    const bool mid = row_tile >= 2000;       // 2000 + 100 R + RC: the mid form (dec_ma + dec_mb instead of dec_fa)
    const int xcd = (row_tile / 1000) & 1;   // 1000 + 100 R + RC: XCD-local row tiles (FusedArgs::xcd)
    row_tile %= 1000;
    if (w.dff != 16 * FF_SLICE || w.heads != 8 || b.T + 1 > PS_SELF || b.S > PS_CROSS || (rows % 16) || !b.fpart ||
        row_base + rows > b.fpart_rows)
        return hipErrorInvalidValue;
    switch (row_tile) {
        case 204: fused_layers<2, 4>(w, b, row_base, rows, xcd, mid, s); break;
        case 208: fused_layers<2, 8>(w, b, row_base, rows, xcd, mid, s); break;
        case 404: fused_layers<4, 4>(w, b, row_base, rows, xcd, mid, s); break;
        case 408: fused_layers<4, 8>(w, b, row_base, rows, xcd, mid, s); break;
        case 416: fused_layers<4, 16>(w, b, row_base, rows, xcd, mid, s); break;
        default: return hipErrorInvalidValue;
    }
    const int stages = 3 * w.layers;
    *x_final = (stages & 1) ? b.x2 : b.x;
    *part_final = b.fpart + (size_t)((stages - 1) & 1) * 16 * b.fpart_rows * 256;
    return hipGetLastError();
}

}  // namespace mnx
