// gemm.hip — 16-bit MFMA GEMM for the Swin "pointwise" Linear layers.
//
//   C[M,N] = epilogue( A[M,K] . W[N,K]^T + bias[N] )            A, W: bf16 (or fp16), K contiguous in both
//
// replaces the nn.Linear calls of the reference encoder (MolNexTR/models/transformers.py:139,141,154,176,
// timm Mlp fc1/fc2 :218,290, PatchMerging.reduction :307,334) on token-major [B*L, C] activations.
//
// Structure (gfx950): 128x128 output tile per 256-thread workgroup, 4 waves as 2(M) x 2(N), each wave 64x64 =
// 4x4 MFMA 16x16x32 tiles with fp32 accumulators; BK=64; operands staged global -> VGPR -> LDS (16-byte chunks,
// XOR-swizzled so ds_read_b128 fragment reads are bank-conflict free), double-buffered with the next tile's
// global loads in flight under the MFMAs. The MFMA is issued "swapped" (A-operand = W rows, B-operand =
// activation rows) so each lane ends up holding 4 CONSECUTIVE output columns of one row: the epilogue stores
// 8-byte (16-bit out) or 16-byte (fp32 out) vectors and reads bias/residual as float4.
#include "common.h"
#include "kernels.h"

namespace mnx {

constexpr int BM = 128, BK = 64;

template <typename T>
struct Stage {
    typename H16<T>::v8 a[4], w[4];
};
// BN = 128: 128x128 tile (64 KiB LDS, 2 workgroups/CU). BN = 64: 128x64 tile (48 KiB, 3 workgroups/CU) for
// shapes whose 128x128 tile count does not fill the chip evenly (N <= 512 at stage 3/4).

// byte offset of 16-byte chunk c (0..7) of tile row r in a [128][64] 16-bit LDS tile
__device__ __forceinline__ int lds_off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

// Epilogue shared by both main-loop variants. The MFMA accumulators hold, per lane, 4 consecutive columns of 16 different
// rows — stored directly that is 8-byte pieces of 32-byte row segments, and an ablation showed those stores cost
// 25-55 % of the kernel on the wide outputs. So the tile is transposed through LDS (free after the K loop) and written
// as whole 128-byte lines, 16 bytes per lane; the residual (fp32, in place) is read in the same coalesced pattern.
template <typename T, int EPI, int BN>
__device__ __forceinline__ void gemm_epilogue(f32x4 (&acc)[BN / 32][4], char* smem, void* Cout,
                                              const float* __restrict__ bias, const float* resid, int M, int N, int m0,
                                              int n0, int wm, int wn, int fr, int fg) {
    typedef typename H16<T>::v4 v4;
    typedef typename H16<T>::v8 v8;
    constexpr int NT = BN / 32;
    constexpr bool OUT16 = (EPI == EPI_BIAS_16 || EPI == EPI_GELU_16);
    constexpr int ELT = OUT16 ? 2 : 4;
    constexpr int ROWB = BN * ELT + 16;                  // padded LDS row (bytes)
    constexpr int CH_ROW = BN * ELT / 16;                // 16-byte chunks per output row
    const int tid = threadIdx.x;
    // fp32 tiles of 128x128 do not fit the staging area at once: two passes of 64 rows (one per wave row wm)
    constexpr int PASSES = (!OUT16 && BN == 128) ? 2 : 1;
    constexpr int ROWS = BM / PASSES;
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
        if (PASSES == 1 || wm == pass) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int nl = wn * (BN / 2) + nt * 16 + fg * 4;
                f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
                if (bias && n0 + nl < N) b4 = *(const f32x4*)(bias + n0 + nl);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int ml = (PASSES == 1 ? wm * 64 : 0) + mt * 16 + fr;
                    f32x4 v = acc[nt][mt] + b4;
                    if (EPI == EPI_GELU_16) {
                        v = gelu_fast4(v);
                    }
                    if (OUT16) {
                        v4 o4 = {(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
                        *(v4*)(smem + ml * ROWB + nl * 2) = o4;
                    } else {
                        *(f32x4*)(smem + ml * ROWB + nl * 4) = v;
                    }
                }
            }
        }
        __syncthreads();
        constexpr int NIT = ROWS * CH_ROW / 256;
        // In-place residual (resid may alias Cout): written as "load, add, store" per iteration, the compiler has to keep every
        // load behind the previous iteration's store — the generated code was eight serial memory round trips per thread, each
        // wait also sitting out the previous store's acknowledgement. An element is read and written by the same thread in the
        // same iteration and the iterations touch different addresses, so ALL residual loads are issued first (8 x 16 bytes
        // per thread in flight), then the adds and stores: the same additions, one round trip.
        f32x4 res[EPI == EPI_RESID_F32 ? NIT : 1];
        if (EPI == EPI_RESID_F32) {
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const int id = tid + i * 256;
                const int rl = id / CH_ROW, ch = id % CH_ROW;
                const int m = m0 + pass * ROWS + rl;
                const int n = n0 + ch * (16 / ELT);
                // unconditional (clamped to the matrix: an out-of-range lane re-reads a valid element it never uses) — a
                // predicated load is a branch, and the compiler waits for each branch's load on its own
                const int mc = m < M ? m : M - 1, nc = n < N ? n : N - 4;
                res[i] = *(const f32x4*)(resid + (size_t)mc * N + nc);
            }
        }
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int id = tid + i * 256;
            const int rl = id / CH_ROW, ch = id % CH_ROW;
            const int m = m0 + pass * ROWS + rl;
            const int n = n0 + ch * (16 / ELT);
            if (m < M && n < N) {
                const size_t o = (size_t)m * N + n;
                if (OUT16) {
                    const v8 v = *(const v8*)(smem + rl * ROWB + ch * 16);
                    *(v8*)((T*)Cout + o) = v;
                } else {
                    f32x4 v = *(const f32x4*)(smem + rl * ROWB + ch * 16);
                    if (EPI == EPI_RESID_F32) v += res[i];
                    *(f32x4*)((float*)Cout + o) = v;
                }
            }
        }
        if (PASSES > 1) __syncthreads();
    }
}

// Epilogue of the split-operand modes (kernels.h SplitArgs): C = epi(oscale * acc + bias) with the exact-erf GELU (the
// point of these modes is fp32-class results), 16-bit outputs written as TWO planes (hi, then lo = v - hi, c_lo elements
// behind; SplitArgs::c_planes == 1: the hi plane only) through the same LDS transposition; fp32 outputs go through
// gemm_epilogue unchanged.
template <typename T, int EPI, int BN>
__device__ __forceinline__ void gemm_epilogue_split(f32x4 (&acc)[BN / 32][4], char* smem, void* Cout,
                                                    const float* __restrict__ bias, const float* resid, int M, int N,
                                                    int m0, int n0, int wm, int wn, int fr, int fg, const SplitArgs sp) {
    typedef typename H16<T>::v4 v4;
    typedef typename H16<T>::v8 v8;
    constexpr int NT = BN / 32;
    constexpr bool OUT16 = (EPI == EPI_BIAS_16 || EPI == EPI_GELU_16);
    if (!OUT16) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[nt][mt] *= sp.oscale;
        gemm_epilogue<T, EPI, BN>(acc, smem, Cout, bias, resid, M, N, m0, n0, wm, wn, fr, fg);
        return;
    }
    constexpr int ROWB = BN * 2 + 16;
    constexpr int CH_ROW = BN * 2 / 16;
    const int tid = threadIdx.x;
    // both planes are derived HERE, from one evaluation of every value (see split16 in common.h), and kept as 16-bit
    v4 hi[NT][4], lo[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int nl = wn * (BN / 2) + nt * 16 + fg * 4;
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (bias && n0 + nl < N) b4 = *(const f32x4*)(bias + n0 + nl);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            f32x4 v = acc[nt][mt] * sp.oscale + b4;
            if (EPI == EPI_GELU_16) v = gelu_split4(v);
            split16x4<T>(v, hi[nt][mt], lo[nt][mt]);
        }
    }
#pragma unroll
    for (int plane = 0; plane < 2; ++plane) {
        if (plane == 1 && sp.c_planes == 1) break;     // the consumer runs on two terms: hi plane only
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int nl = wn * (BN / 2) + nt * 16 + fg * 4;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int ml = wm * 64 + mt * 16 + fr;
                *(v4*)(smem + ml * ROWB + nl * 2) = plane == 0 ? hi[nt][mt] : lo[nt][mt];
            }
        }
        __syncthreads();
        T* Cp = (T*)Cout + (plane == 0 ? (size_t)0 : sp.c_lo);
#pragma unroll
        for (int i = 0; i < BM * CH_ROW / 256; ++i) {
            const int id = tid + i * 256;
            const int rl = id / CH_ROW, ch = id % CH_ROW;
            const int m = m0 + rl;
            const int n = n0 + ch * 8;
            if (m < M && n < N) *(v8*)(Cp + (size_t)m * N + n) = *(const v8*)(smem + rl * ROWB + ch * 16);
        }
        __syncthreads();
    }
}

template <typename T, int EPI, int BN, bool SPLIT>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const T* __restrict__ A, const T* __restrict__ W,
                                                      void* Cout, const float* __restrict__ bias,
                                                      const float* resid, int M, int N, int K, int tiles_n,
                                                      int n_tiles, const SplitArgs sp) {
    typedef typename H16<T>::v8 v8;
    constexpr int NT = BN / 32;          // 16-wide n-tiles per wave (waves are 2(M) x 2(N))
    constexpr int WLD = BN / 32;         // W chunks per thread per K-tile
    constexpr int STG = (BM + BN) * BK * 2;   // bytes per stage
    __shared__ __attribute__((aligned(16))) char smem[2 * STG];  // [buf][A 128x64 | W BNx64] 16-bit
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int tile = xcd_remap(blockIdx.x, n_tiles);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;

    // global->register staging coordinates: 4 chunks of A and 4 of W per thread per K-tile
    const int ld_c = tid & 7, ld_r = tid >> 3;  // chunk column, base row (0..31)
    const T* a_ptr[4];
    const T* w_ptr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int ra = min(m0 + ld_r + 32 * i, M - 1), rw = min(n0 + ld_r + 32 * (i < WLD ? i : 0), N - 1);
        a_ptr[i] = A + (size_t)ra * K + ld_c * 8;
        w_ptr[i] = W + (size_t)rw * K + ld_c * 8;
    }
    // split modes: K-tile j of the loop is term j % 3 of K-tile j / 3: (A hi, W lo), (A hi, W hi), (A lo, W hi) — the one
    // order in which every encoder GEMM kernel accumulates the three terms (see gemm_tn_glds_kernel); two terms: the first
    // two of them
    const int nterm = SPLIT ? sp.terms : 1;
    const int nk = ((K + BK - 1) / BK) * nterm;
    Stage<T> st;
    auto load_g = [&](int j) {
        const int kt = nterm > 1 ? j / nterm : j, term = nterm > 1 ? j - kt * nterm : 1;
        const int k0 = kt * BK;
        const size_t ka = (size_t)k0 + (term == 2 ? sp.a_lo : 0), kw = (size_t)k0 + (term == 0 ? sp.w_lo : 0);
        const bool ok = (k0 + ld_c * 8) < K;  // K % 8 == 0: a chunk is all-in or all-out
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v8 z = {};
            st.a[i] = ok ? *(const v8*)(a_ptr[i] + ka) : z;
            if (i < WLD) st.w[i] = ok ? *(const v8*)(w_ptr[i] + kw) : z;
        }
    };
    auto store_s = [&](int buf) {
        char* ab = smem + buf * STG;
        char* wb = ab + BM * BK * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r = ld_r + 32 * i;
            *(v8*)(ab + lds_off(r, ld_c)) = st.a[i];
            if (i < WLD) *(v8*)(wb + lds_off(r, ld_c)) = st.w[i];
        }
    };

    f32x4 acc[NT][4];  // [nt][mt]
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    load_g(0);
    store_s(0);
    __syncthreads();
    const int fr = lane & 15, fg = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_g(kt + 1);
        const char* ab = smem + (kt & 1) * STG;
        const char* wb = ab + BM * BK * 2;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            v8 af[4], wf[NT];
#pragma unroll
            for (int t = 0; t < 4; ++t) af[t] = *(const v8*)(ab + lds_off(wm * 64 + t * 16 + fr, ks * 4 + fg));
#pragma unroll
            for (int t = 0; t < NT; ++t) wf[t] = *(const v8*)(wb + lds_off(wn * (BN / 2) + t * 16 + fr, ks * 4 + fg));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = H16<T>::mfma(wf[nt], af[mt], acc[nt][mt]);
        }
        if (kt + 1 < nk) store_s((kt + 1) & 1);
        __syncthreads();
    }

    if (SPLIT) gemm_epilogue_split<T, EPI, BN>(acc, smem, Cout, bias, resid, M, N, m0, n0, wm, wn, fr, fg, sp);
    else gemm_epilogue<T, EPI, BN>(acc, smem, Cout, bias, resid, M, N, m0, n0, wm, wn, fr, fg);
}

// Main-loop variant with direct global->LDS DMA (global_load_lds_dwordx4): no VGPR staging, no ds_write pass.
// The LDS image written by a wave-instruction is lane-linear (base + lane*16 = 8 rows x 128 B), so the XOR swizzle
// is applied on the per-lane SOURCE address and again on the fragment reads (same involution). Requires K % 64 == 0.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

// Double-buffered K loop (64 / 48 KiB LDS, 2-3 workgroups per CU).
template <typename T, int EPI, int BN, bool SPLIT>
__global__ __launch_bounds__(256) void gemm_tn_glds_kernel(const T* __restrict__ A, const T* __restrict__ W, void* Cout,
                                                           const float* __restrict__ bias, const float* resid, int M,
                                                           int N, int K, int tiles_n, int n_tiles, const SplitArgs sp) {
    typedef typename H16<T>::v8 v8;
    constexpr int NT = BN / 32;
    constexpr int WLD = BN / 32;
    constexpr int ASZ = BM * BK * 2, WSZ = BN * BK * 2;        // one A tile (16 KiB), one W tile
    constexpr int EPI_LDS = 34816;    // largest epilogue staging area: 128 rows x (128 x 2 + 16) B = 128 x (64 x 4 + 16) B
    constexpr int RING = 2 * ASZ + 2 * WSZ;                    // two A buffers, then two W buffers
    __shared__ __attribute__((aligned(16))) char smem[RING > EPI_LDS ? RING : EPI_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int tile = xcd_remap(blockIdx.x, n_tiles);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    // wave w fills A rows [32w, 32w+32) with 4 DMA instructions of 8 rows each, W rows [BN/4*w, ..) with WLD
    const int r_in = lane >> 3, p = lane & 7;
    const T* a_src[4];
    const T* w_src[WLD];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + r_in;
        a_src[i] = A + (size_t)min(m0 + r, M - 1) * K + ((p ^ ((r >> 1) & 7)) << 3);
    }
#pragma unroll
    for (int i = 0; i < WLD; ++i) {
        const int r = (wave * WLD + i) * 8 + r_in;
        w_src[i] = W + (size_t)min(n0 + r, N - 1) * K + ((p ^ ((r >> 1) & 7)) << 3);
    }
    // One step of the K loop multiplies an A tile by a W tile out of LDS while the DMA of later tiles is in flight; one
    // __syncthreads per step drains that DMA and frees the buffers the step has read.
    // Plain modes: step kt uses A / W buffers kt & 1 and prefetches K-tile kt + 1 into the other pair.
    // Split modes with three terms: K-tile kt is three steps that SHARE their fills — 4 tile fills per 3 MFMA passes:
    //     step   product        prefetched during the step
    //     s1     A hi . W lo    W hi (kt)                -> the other W buffer
    //     s2     A hi . W hi    A lo (kt)                -> the other A buffer
    //     s3     A lo . W hi    A hi, W lo (kt + 1)      -> the buffers s1 / s2 released
    // Split modes with two terms (the activation's lo plane dropped): K-tile kt is two steps and 3 tile fills:
    //     s1     A hi . W lo    W hi (kt)                -> the other W buffer
    //     s2     A hi . W hi    A hi (kt + 1)            -> the other A buffer;  W lo (kt + 1) -> the buffer s1 released
    // (A hi alternates between the two A buffers by K-tile; W lo is always in W buffer 0, W hi in 1.)
    // The term order hi.lo, hi.hi, lo.hi is the one in which gemm256x3_kernel (gemm256.hip), whose phases are chained by
    // register reuse, can accumulate EVERY row; launch_gemm16 splits a layer's rows between the two kernels by batch size,
    // and with the same order of fp32 additions per output element in both, an image's features do not depend on the
    // batch it was encoded in (tests: ..._is_batch_invariant).
    const int nterm = SPLIT ? sp.terms : 1;
    auto issue_a = [&](size_t koff, int buf) {
        char* ab = smem + buf * ASZ;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((glb_void_t*)(a_src[i] + koff), (lds_void_t*)(ab + (wave * 4 + i) * 1024), 16, 0, 0);
    };
    auto issue_w = [&](size_t koff, int buf) {
        char* wb = smem + 2 * ASZ + buf * WSZ;
#pragma unroll
        for (int i = 0; i < WLD; ++i)
            __builtin_amdgcn_global_load_lds((glb_void_t*)(w_src[i] + koff), (lds_void_t*)(wb + (wave * WLD + i) * 1024), 16, 0, 0);
    };
    f32x4 acc[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nk = K / BK, nsteps = nk * nterm;
    const bool three = SPLIT && nterm == 3, two = SPLIT && nterm == 2;
    issue_a(0, 0);
    issue_w((three || two) ? sp.w_lo : (size_t)0, 0);
    __syncthreads();
    const int fr = lane & 15, fg = lane >> 4;
    int il = 0;                            // three-term schedule: W buffer holding the current K-tile's W lo (W hi: the other one; A hi is always in A buffer 0, A lo in 1)
    for (int j = 0; j < nsteps; ++j) {
        int ia, iw;
        if (three) {
            const int kt = j / 3, term = j - kt * 3;
            const size_t k0 = (size_t)kt * BK;
            if (term == 0) { ia = 0; iw = il; issue_w(k0, il ^ 1); }
            else if (term == 1) { ia = 0; iw = il ^ 1; issue_a(k0 + sp.a_lo, 1); }
            else {
                ia = 1; iw = il ^ 1;
                if (kt + 1 < nk) { issue_a(k0 + BK, 0); issue_w(k0 + BK + sp.w_lo, il); }
            }
        } else if (two) {
            const int kt = j >> 1;
            const size_t k0 = (size_t)kt * BK;
            ia = kt & 1;
            if ((j & 1) == 0) { iw = 0; issue_w(k0, 1); }
            else {
                iw = 1;
                if (kt + 1 < nk) { issue_a(k0 + BK, ia ^ 1); issue_w(k0 + BK + sp.w_lo, 0); }
            }
        } else {
            ia = iw = j & 1;
            if (j + 1 < nsteps) { issue_a((size_t)(j + 1) * BK, ia ^ 1); issue_w((size_t)(j + 1) * BK, iw ^ 1); }
        }
        const char* ab = smem + ia * ASZ;
        const char* wb = smem + 2 * ASZ + iw * WSZ;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            v8 af[4], wf[NT];
#pragma unroll
            for (int t = 0; t < 4; ++t) af[t] = *(const v8*)(ab + lds_off(wm * 64 + t * 16 + fr, ks * 4 + fg));
#pragma unroll
            for (int t = 0; t < NT; ++t) wf[t] = *(const v8*)(wb + lds_off(wn * (BN / 2) + t * 16 + fr, ks * 4 + fg));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = H16<T>::mfma(wf[nt], af[mt], acc[nt][mt]);
        }
        __syncthreads();   // drains the DMA issued in this step (vmcnt(0)) and frees the buffers the step has read
    }
    if (SPLIT) gemm_epilogue_split<T, EPI, BN>(acc, smem, Cout, bias, resid, M, N, m0, n0, wm, wn, fr, fg, sp);
    else gemm_epilogue<T, EPI, BN>(acc, smem, Cout, bias, resid, M, N, m0, n0, wm, wn, fr, fg);
}

// fp32 "parity mode" GEMM (compute_dtype FP32): C = epi(A.W^T + b) with fp32 operands on v_mfma_f32_16x16x4_f32 (exact
// fp32 FMA chains, 1/16 of the bf16 rate). 64x64 tile, 4 waves of 32x32, operands from global memory straight into the
// fragment layout (lane (fr, fg): float4 at [row fr][k + 4 fg]; the k-slot permutation is the same on both operands).
// Swapped like the 16-bit kernel (A-operand = W rows) so a lane owns 4 consecutive output columns of one row. Not tuned:
// it exists so that the whole path can be checked against the reference at fp32 accuracy (tokens exact from pixels).
template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                       float* Cout, const float* __restrict__ bias, const float* resid,
                                                       int M, int N, int K) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int m0 = blockIdx.y * 64 + (wave & 1) * 32, n0 = blockIdx.x * 64 + (wave >> 1) * 32;
    const float* ap[2];
    const float* wp[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        ap[t] = A + (size_t)min(m0 + t * 16 + fr, M - 1) * K + fg * 4;
        wp[t] = W + (size_t)min(n0 + t * 16 + fr, N - 1) * K + fg * 4;
    }
    f32x4 acc[2][2];   // [nt][mt]
#pragma unroll
    for (int i = 0; i < 2; ++i) acc[i][0] = acc[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 16) {           // K % 16 == 0
        f32x4 a[2], w[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            a[t] = *(const f32x4*)(ap[t] + k0);
            w[t] = *(const f32x4*)(wp[t] + k0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[nt][j], a[mt][j], acc[nt][mt], 0, 0, 0);
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n = n0 + nt * 16 + fg * 4;
        if (n >= N) continue;
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (bias) b4 = *(const f32x4*)(bias + n);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int m = m0 + mt * 16 + fr;
            if (m >= M) continue;
            f32x4 v = acc[nt][mt] + b4;
            if (EPI == EPI_GELU_16) { v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]); }
            const size_t o = (size_t)m * N + n;
            if (EPI == EPI_RESID_F32) v += *(const f32x4*)(resid + o);
            *(f32x4*)(Cout + o) = v;
        }
    }
}

static hipError_t launch_f32(int epi, const void* A, const void* W, void* C, const float* bias, const float* resid,
                             int M, int N, int K, hipStream_t s) {
    if ((K & 15) || (N & 3)) return hipErrorInvalidValue;
    dim3 grid((N + 63) / 64, (M + 63) / 64), block(256);
#define MNX_F32_CASE(E)                                                                                                   \
    case E:                                                                                                               \
        hipLaunchKernelGGL((gemm_f32_kernel<E>), grid, block, 0, s, (const float*)A, (const float*)W, (float*)C, bias,    \
                           resid, M, N, K);                                                                               \
        break;
    switch (epi) {
        MNX_F32_CASE(EPI_BIAS_16)
        MNX_F32_CASE(EPI_GELU_16)
        MNX_F32_CASE(EPI_RESID_F32)
        MNX_F32_CASE(EPI_BIAS_F32)
        default: return hipErrorInvalidValue;
    }
#undef MNX_F32_CASE
    return hipGetLastError();
}

template <typename T, int BN, bool SPLIT>
static hipError_t launch_bn(int epi, const void* A, const void* W, void* C, const float* bias, const float* resid,
                            int M, int N, int K, hipStream_t s, const SplitArgs& sp) {
    const int tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN;
    dim3 grid(tm * tn), block(256);
    const bool glds = (K % BK) == 0;
#define MNX_GEMM_CASE(E)                                                                                                  \
    case E:                                                                                                               \
        if (glds)                                                                                                         \
            hipLaunchKernelGGL((gemm_tn_glds_kernel<T, E, BN, SPLIT>), grid, block, 0, s, (const T*)A, (const T*)W, C,   \
                               bias, resid, M, N, K, tn, tm * tn, sp);                                                    \
        else                                                                                                              \
            hipLaunchKernelGGL((gemm_tn_kernel<T, E, BN, SPLIT>), grid, block, 0, s, (const T*)A, (const T*)W, C, bias,  \
                               resid, M, N, K, tn, tm * tn, sp);                                                          \
        break;
    switch (epi) {
        MNX_GEMM_CASE(EPI_BIAS_16)
        MNX_GEMM_CASE(EPI_GELU_16)
        MNX_GEMM_CASE(EPI_RESID_F32)
        MNX_GEMM_CASE(EPI_BIAS_F32)
        default: return hipErrorInvalidValue;
    }
#undef MNX_GEMM_CASE
    return hipGetLastError();
}

template <typename T, bool SPLIT>
static hipError_t launch_t(int epi, const void* A, const void* W, void* C, const float* bias, const float* resid,
                           int M, int N, int K, hipStream_t s, const SplitArgs& sp) {
    // tile choice: 128x128 unless its tile count leaves the 512 resident-workgroup slots (256 CUs x 2) badly
    // quantised; then 128x64 tiles (3 workgroups per CU)
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    const double waves128 = (double)t128 / 512.0;
    const bool small = t128 < 512 || (waves128 < 3.0 && (waves128 - (long)waves128) > 0.0 && (waves128 - (long)waves128) < 0.6);
#ifdef MNX_TILE128_BN      // tools/gemm_lab: force one tile width for every shape (64 or 128)
    if (MNX_TILE128_BN == 64 && N >= 64) return launch_bn<T, 64, SPLIT>(epi, A, W, C, bias, resid, M, N, K, s, sp);
    if (MNX_TILE128_BN == 128) return launch_bn<T, 128, SPLIT>(epi, A, W, C, bias, resid, M, N, K, s, sp);
#endif
    // N = 128 (stage-1 proj / fc2): 64-column tiles (+10 % in round 2); with the residual loads of round 5 in flight together the
    // K = 512 layer (fc2) is 6 % faster on 128-column tiles again, K = 128 (proj) the same on both (profiles/r05_gemm_lab_tile_width.txt)
    const bool narrow = N <= 128 && K < 512;
    if ((small || narrow) && N >= 64) return launch_bn<T, 64, SPLIT>(epi, A, W, C, bias, resid, M, N, K, s, sp);
    return launch_bn<T, 128, SPLIT>(epi, A, W, C, bias, resid, M, N, K, s, sp);
}

hipError_t launch_gemm16_tile128(int dtype, int epi, const void* A, const void* W, void* C, const float* bias,
                                 const float* resid, int M, int N, int K, hipStream_t s, const SplitArgs* sp) {
    if ((K & 7) || (N & 7) || M <= 0) return hipErrorInvalidValue;
    if (dtype == MNX_DT_F32) return launch_f32(epi, A, W, C, bias, resid, M, N, K, s);
    if (dt_split(dtype)) {
        if (!sp || sp->terms < 1 || sp->terms > 3 || (sp->c_planes != 1 && sp->c_planes != 2)) return hipErrorInvalidValue;
        return dtype == MNX_DT_F16X3 ? launch_t<f16_t, true>(epi, A, W, C, bias, resid, M, N, K, s, *sp)
                                     : launch_t<bf16_t, true>(epi, A, W, C, bias, resid, M, N, K, s, *sp);
    }
    const SplitArgs none;
    return dtype == MNX_DT_F16 ? launch_t<f16_t, false>(epi, A, W, C, bias, resid, M, N, K, s, none)
                               : launch_t<bf16_t, false>(epi, A, W, C, bias, resid, M, N, K, s, none);
}

// rows that launch_gemm16 gives to gemm256x3_kernel (the rest goes to the 128x128 kernel); 0 = none
static int x3_main_rows(int dtype, int epi, int M, int N, int K, int terms) {
    if (!dt_split(dtype) || (terms != 3 && !(terms == 2 && dtype == MNX_DT_F16X3)) || !gemm256x3_supports(dtype, epi, M, N, K)) return 0;
    const int tn = N / 256, tm = M / 256, tiles = tm * tn, cus = persistent_cus();
    int tm_main = tm;
    if (tiles % cus != 0 && (tiles % cus) * 10 < cus * 8) tm_main = (tiles / cus) * cus / tn;
    return tm_main * 256;
}

const char* gemm16_route(int dtype, int epi, int M, int N, int K, int terms, bool has_bias) {
    const int r = x3_main_rows(dtype, epi, M, N, K, terms);
    if (r == M) return "x3";
    if (r > 0) return "x3+128";
    if (has_bias && gemm256_supports(dtype, epi, M, N, K)) return "g256";
    if (gemm_res_preferred(dtype, epi, M, N, K)) return "gres";
    return "128";
}

hipError_t launch_gemm16(int dtype, int epi, const void* A, const void* W, void* C, const float* bias,
                         const float* resid, int M, int N, int K, hipStream_t s, const SplitArgs* sp) {
    // Split modes with three (two) terms: the six- (four-) phase 256x256 kernel (gemm256x3_kernel) takes the rows that fill whole
    // rounds of 256 tiles (one workgroup per CU walks its tiles in rounds; a last round that is at least 80 % full is
    // taken too), the 128x128 kernel (2-3 workgroups per CU) the remaining rows. Shape-only, like everything below.
    if (const int r0 = sp ? x3_main_rows(dtype, epi, M, N, K, sp->terms) : 0) {
        const size_t es = 2;
        hipError_t e = launch_gemm256x3(dtype, epi, A, W, C, bias, resid, r0, N, K, s, sp);
        if (e != hipSuccess || r0 == M) return e;
        const bool out16 = (epi == EPI_BIAS_16 || epi == EPI_GELU_16);
        return launch_gemm16_tile128(dtype, epi, (const char*)A + (size_t)r0 * K * es, W,
                                     (char*)C + (size_t)r0 * N * (out16 ? es : sizeof(float)), bias,
                                     resid ? resid + (size_t)r0 * N : nullptr, M - r0, N, K, s, sp);
    }
    // shape-only dispatch (never data-dependent; the one environment input is the persistent workgroup count of MNX_ENC_CUS,
    // which moves rows between gemm256x3_kernel and the 128x128 kernel — the two add the same numbers in the same order, so
    // the results do not change: test_persistent_encoder_grids_on_fewer_cus_...): the persistent 256x256 kernel for the 16-bit-output
    // layers whose tile count fills the chip, the persistent 256x128 kernel for the fp32-output layers likewise, the
    // 128x128 kernel for everything else
    if (bias && gemm256_supports(dtype, epi, M, N, K)) return launch_gemm256(dtype, epi, A, W, C, bias, M, N, K, s, sp);
    if (gemm_res_preferred(dtype, epi, M, N, K)) return launch_gemm_res(dtype, epi, A, W, (float*)C, bias, resid, M, N, K, s, sp);
    return launch_gemm16_tile128(dtype, epi, A, W, C, bias, resid, M, N, K, s, sp);
}

}  // namespace mnx
