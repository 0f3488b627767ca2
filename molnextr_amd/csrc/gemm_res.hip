// gemm_res.hip — persistent form of the encoder GEMM for its fp32-output layers: proj and fc2 (bias + fp32 residual, in
// place) and the patch-merging reduction (no bias): C = A.W^T [+ b] [+ C], A [M,K], W [N,K] 16-bit, K contiguous;
// M a multiple of 256, N of 128, K of 64. launch_gemm16 (gemm.hip) routes here when gemm_res_supports() says so.
// Replaces the nn.Linear calls of MolNexTR/models/transformers.py:176 (attn.proj), timm Mlp fc2 (:218, :290) and
// PatchMerging.reduction (:307, :334) on token-major activations.
//
// Why a third kernel (profiles/r02_gemm_shapes.md): the residual layers were half of the encoder's
// GEMM time on the 128x128 kernel (stage-3 fc2 636, proj 422 TFLOP/s), whose K loop keeps ONE 32 KiB K-tile in flight per
// workgroup (2 per CU) and whose epilogue (LDS transposition, residual read, store) runs strictly after the loop. They do
// not fit gemm256.hip: N = 128 .. 512 gives too few 256x256 tiles to fill 256 CUs evenly, and its 128 accumulator
// registers leave no room for a residual tile.
//
// Structure (gfx950, one 512-thread workgroup per CU, grid = min(tiles, 256), tile t -> workgroup t mod grid):
//   * 256x128 output tile, 8 waves as 4 (M) x 2 (N), each wave 64x64 = 4x4 MFMA 16x16x32 tiles (64 accumulator
//     registers), MFMA issued "swapped" (A-operand = W rows) so a lane owns 4 consecutive output columns of a row;
//   * BK = 64: a K-tile is 32 KiB of A + 16 KiB of W in LDS ([rows][64] 16-bit, the XOR swizzle of gemm.hip, conflict-free
//     ds_read_b128), filled by global_load_lds_dwordx4 (6 per wave: scalar base + 32-bit lane offset). THREE K-tile
//     buffers (144 KiB): two K-tiles are in flight while one is multiplied, retired with counted vmcnt;
//   * persistent: the K-tile stream runs through all tiles of the workgroup without draining, so the first K-tiles of the
//     next tile load while the finished tile is written;
//   * one workgroup barrier per K-tile (after the wait for its fill: every wave's pieces have landed, and every wave has
//     finished reading the buffer that the fill issued right after the barrier overwrites);
//   * epilogue straight from the accumulators: a lane's 4 consecutive columns are one 16-byte access, the 4 lane groups
//     of a row cover 64 contiguous bytes; the residual tile (64 registers) is fetched by inline-asm loads issued ONE K-TILE
//     BEFORE the tile's last one, so it lands under the last 64 MFMAs instead of stalling the in-order vmcnt queue at the
//     epilogue; the bias sits in LDS (copied once per workgroup). In place (C == residual) is safe: every element is
//     read and written by the same lane;
//   * split-operand modes (kernels.h SplitArgs): as gemm256.hip — K-tile j of a tile is term j % 3 of K-tile j / 3,
//     C = oscale * acc + bias + residual.
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace mnx {

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

constexpr int RM = 256, RN = 128, RK = 64;
constexpr int R_A = RM * RK * 2, R_B = RN * RK * 2;      // 32 KiB + 16 KiB per K-tile
constexpr int R_BUF = R_A + R_B;
constexpr int R_BIAS = 4096;                             // bias[N] copy, N <= 1024
constexpr int R_LDS = 3 * R_BUF + R_BIAS;                // 148 KiB
constexpr int R_FILL = 6;                                // DMA instructions per wave per K-tile
constexpr int R_EXTRA = 16;                              // residual loads, or stores, per wave per tile

__device__ __forceinline__ int lds_off_r(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

// s_waitcnt vmcnt(n), n = 6 + 16 k: the counted waits of the K-tile stream. Loads and stores of a wave retire in issue
// order; a count that is too small only stalls, one that is too large would read a slot before it has landed.
__device__ __forceinline__ void wait_vm_dyn(int n) {
    if (n >= R_FILL + 2 * R_EXTRA) asm volatile("s_waitcnt vmcnt(38)" ::: "memory");
    else if (n >= R_FILL + R_EXTRA) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
    else if (n >= R_FILL) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// one 16-byte residual load: dst <- base[voff + OFF] (scalar base, 32-bit lane offset, immediate offset), untracked by
// the compiler's waitcnt insertion
template <int OFF>
__device__ __forceinline__ void load_res(f32x4& dst, unsigned voff, const float* base) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=&v"(dst) : "v"(voff), "s"(base), "n"(OFF) : "memory");
}

template <typename T, int EPI, bool SPLIT>
__global__ __launch_bounds__(512) void gemm_res_kernel(const T* __restrict__ A, const T* __restrict__ W, float* C,
                                                        const float* __restrict__ bias, const float* resid, int M, int N,
                                                        int K, int tiles_n, int n_tiles, const SplitArgs sp) {
    static_assert(EPI == EPI_RESID_F32 || EPI == EPI_BIAS_F32, "fp32-output epilogues only");
    typedef typename H16<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;
    const int nterm = SPLIT ? sp.terms : 1;
    const int nk = K / RK, nkk = nk * nterm;
    const int my_first = blockIdx.x, stride = gridDim.x;
    const int my_tiles = (n_tiles - my_first + stride - 1) / stride;
    const int total_kt = my_tiles * nkk;
    const long a_lo_b = SPLIT ? (long)sp.a_lo * 2 : 0, w_lo_b = SPLIT ? (long)sp.w_lo * 2 : 0;

    // bias -> LDS once per workgroup (plain loads: nothing else is in flight yet)
    float* bias_s = (float*)(smem + 3 * R_BUF);
    for (int i = tid; i < N; i += 512) bias_s[i] = bias ? bias[i] : 0.f;

    // per-thread byte offsets of its DMA pieces inside a tile: wave w fills A rows [32w, 32w+32) and W rows [16w, 16w+16)
    const int r_in = lane >> 3, pc = lane & 7;
    unsigned offA[4], offB[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + r_in;
        offA[i] = (unsigned)((r * K + ((pc ^ ((r >> 1) & 7)) << 3)) * 2);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wave * 2 + i) * 8 + r_in;
        offB[i] = (unsigned)((r * K + ((pc ^ ((r >> 1) & 7)) << 3)) * 2);
    }
    struct Pos { const char* a; const char* w; int kt, seq, term; };
    auto tile_origin = [&](int seq, int& m0, int& n0) {
        const int tile = xcd_remap(my_first + seq * stride, n_tiles);
        m0 = (tile / tiles_n) * RM; n0 = (tile % tiles_n) * RN;
    };
    auto pos_at = [&](int seq) {
        int m0, n0; tile_origin(seq, m0, n0);
        Pos q; q.a = (const char*)(A + (size_t)m0 * K); q.w = (const char*)(W + (size_t)n0 * K); q.kt = 0; q.seq = seq;
        q.term = 0;
        return q;
    };
    auto advance = [&](Pos& q) {
        if (SPLIT && nterm == 3) {
            if (q.term == 0) { q.w += w_lo_b; q.term = 1; return; }                    // (A hi, W lo)
            if (q.term == 1) { q.w -= w_lo_b; q.a += a_lo_b; q.term = 2; return; }     // (A lo, W hi)
            q.a -= a_lo_b; q.term = 0;
        }
        if (++q.kt == nk) { if (q.seq + 1 < my_tiles) q = pos_at(q.seq + 1); else { q.kt = 0; ++q.seq; } }
        else { q.a += RK * 2; q.w += RK * 2; }
    };
    auto fill = [&](const Pos& q, int buf) {
        char* ab = smem + buf * R_BUF;
        char* wb = ab + R_A;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((glb_void_t*)(q.a + offA[i]), (lds_void_t*)(ab + (wave * 4 + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((glb_void_t*)(q.w + offB[i]), (lds_void_t*)(wb + (wave * 2 + i) * 1024), 16, 0, 0);
    };

    f32x4 acc[4][4];   // [nt][mt]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // residual tile of the output tile being accumulated (EPI_RESID_F32): written by inline-asm loads so that they are
    // retired by the counted waits, never by a compiler-inserted vmcnt(0). Nothing may read res[][] before the explicit
    // wait + register fence in the epilogue (guarded by the all-element GPU test of this kernel).
    f32x4 res[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) res[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int m0, n0;
    tile_origin(0, m0, n0);
    __syncthreads();                              // bias copy visible; also orders it before the first DMA
    Pos pf = pos_at(0);                           // fill stream position: K-tile g + 2 after the prologue
    fill(pf, 0); advance(pf);
    if (total_kt > 1) { fill(pf, 1); advance(pf); }

    int kt = 0, seq = 0;
    int e_prev2 = 0, e_prev1 = 0;                 // non-fill VMEM operations issued in iterations g-2 and g-1
    for (int g = 0; g < total_kt; ++g) {
        const bool last_kt = (kt == nkk - 1);
        const bool pre_last = (kt == nkk - 2);
        // ---- K-tile g has landed: fills(g) were issued in iteration g-2; younger = extras(g-2) + fills(g+1) + extras(g-1)
        if (g + 1 >= total_kt) wait_vm_dyn(0);
        else wait_vm_dyn(R_FILL + e_prev2 + e_prev1);
        __builtin_amdgcn_s_barrier();
        int e_now = 0;
        if (g + 2 < total_kt) { fill(pf, (g + 2) % 3); advance(pf); }
        if (EPI == EPI_RESID_F32 && pre_last) {
            // scalar tile base + one 32-bit lane offset per 16-row slab + immediate column offset: no 64-bit VGPR address
            // pairs are live across the 16 loads in flight (DESIGN.md section 6.3 documents a hazard with those)
            const float* rbase = resid + (size_t)m0 * N + n0;
            const unsigned lane_off = (unsigned)(((wr * 64 + fr) * N + wc * 64 + fg * 4) * 4);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const unsigned vo = lane_off + (unsigned)(mt * 16 * N * 4);
                load_res<0>(res[0][mt], vo, rbase);
                load_res<64>(res[1][mt], vo, rbase);
                load_res<128>(res[2][mt], vo, rbase);
                load_res<192>(res[3][mt], vo, rbase);
            }
            e_now += R_EXTRA;
        }
        const char* ab = smem + (g % 3) * R_BUF;
        const char* wb = ab + R_A;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            v8 af[4], wf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) af[t] = *(const v8*)(ab + lds_off_r(wr * 64 + t * 16 + fr, ks * 4 + fg));
#pragma unroll
            for (int t = 0; t < 4; ++t) wf[t] = *(const v8*)(wb + lds_off_r(wc * 64 + t * 16 + fr, ks * 4 + fg));
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = H16<T>::mfma(wf[nt], af[mt], acc[nt][mt]);
        }
        if (last_kt) {
            // ---- epilogue of tile `seq`. The residual loads were issued one K-tile ago; younger than them: the fills
            // issued in this iteration (if any)
            if (EPI == EPI_RESID_F32) {
                if (g + 2 < total_kt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) asm volatile("" : "+v"(res[nt][mt]));   // res is defined from here on
            }
            float* cp = C + (size_t)(m0 + wr * 64 + fr) * N + n0 + wc * 64 + fg * 4;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const f32x4 b4 = *(const f32x4*)(bias_s + n0 + wc * 64 + nt * 16 + fg * 4);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    f32x4 v = SPLIT ? acc[nt][mt] * sp.oscale + b4 : acc[nt][mt] + b4;
                    if (EPI == EPI_RESID_F32) v += res[nt][mt];
                    *(f32x4*)(cp + (size_t)mt * 16 * N + nt * 16) = v;
                    acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
            e_now += R_EXTRA;
            kt = 0; ++seq;
            if (seq < my_tiles) tile_origin(seq, m0, n0);
        } else {
            ++kt;
        }
        e_prev2 = e_prev1;
        e_prev1 = e_now;
    }
}

}  // namespace

bool gemm_res_supports(int dtype, int epi, int M, int N, int K) {
    if (dtype != MNX_DT_BF16 && dtype != MNX_DT_F16 && !dt_split(dtype)) return false;
    if (epi != EPI_RESID_F32 && epi != EPI_BIAS_F32) return false;
    if (M % RM || N % RN || K % RK || N > R_BIAS / 4) return false;
    if (!dt_split(dtype) && K < 2 * RK) return false;          // the residual prefetch needs >= 2 K-tiles per output tile
    return true;
}

// Where launch_gemm16 prefers this kernel over the 128x128 one (measured per shape, profiles/r03_gemm_shapes.md):
//   * plain 16-bit operands only — in the split modes the 128x128 kernel shares the fills of its three terms (4 tile
//     fills per 3 MFMA passes), which this kernel's combined A|W ring does not, and wins on every shape;
//   * >= 4 rounds of 256 tiles: one workgroup per CU walks tiles in rounds, and at 2.25 rounds (stage 4, the last
//     patch-merging) the 128x128 kernel's 512 resident workgroups quantise no worse and run a faster K loop.
bool gemm_res_preferred(int dtype, int epi, int M, int N, int K) {
    if (!gemm_res_supports(dtype, epi, M, N, K) || dt_split(dtype)) return false;
    return (M / RM) * (N / RN) >= 1024;
}

template <auto Kern>      // keyed on the kernel value: one flag per instantiation (see gemm256.hip)
static hipError_t lds_opt_in_res() {
    static unsigned long long done = 0;          // bit d: device d has the attribute
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && (__atomic_load_n(&done, __ATOMIC_ACQUIRE) >> dev & 1ull)) return hipSuccess;
    e = hipFuncSetAttribute((const void*)Kern, hipFuncAttributeMaxDynamicSharedMemorySize, R_LDS);
    if (e == hipSuccess && dev >= 0 && dev < 64) __atomic_fetch_or(&done, 1ull << dev, __ATOMIC_RELEASE);
    return e;
}

hipError_t launch_gemm_res(int dtype, int epi, const void* A, const void* W, float* C, const float* bias,
                           const float* resid, int M, int N, int K, hipStream_t s, const SplitArgs* sp) {
    if (M % RM || N % RN || K % RK || N > R_BIAS / 4) return hipErrorInvalidValue;
    const bool split = dt_split(dtype);
    if (split && (!sp || (sp->terms != 1 && sp->terms != 3))) return hipErrorInvalidValue;
    if (epi == EPI_RESID_F32 && !resid) return hipErrorInvalidValue;
    const SplitArgs spv = split ? *sp : SplitArgs();
    if ((K / RK) * (split ? spv.terms : 1) < 2) return hipErrorInvalidValue;
    const int tm = M / RM, tn = N / RN;
    const int grid = tm * tn < 256 ? tm * tn : 256;
#define MNX_GRES_CASE(TT, E, SP)                                                                                          \
    case E: {                                                                                                             \
        const hipError_t attr = lds_opt_in_res<gemm_res_kernel<TT, E, SP>>();                                               \
        if (attr != hipSuccess) return attr;                                                                              \
        hipLaunchKernelGGL((gemm_res_kernel<TT, E, SP>), dim3(grid), dim3(512), R_LDS, s, (const TT*)A, (const TT*)W, C,  \
                           bias, resid, M, N, K, tn, tm * tn, spv);                                                       \
        break;                                                                                                            \
    }
#define MNX_GRES_TYPE(TT, SP)                                                                                             \
    switch (epi) { MNX_GRES_CASE(TT, EPI_RESID_F32, SP) MNX_GRES_CASE(TT, EPI_BIAS_F32, SP) default: return hipErrorInvalidValue; }
    if (dtype == MNX_DT_F16) { MNX_GRES_TYPE(f16_t, false) }
    else if (dtype == MNX_DT_BF16) { MNX_GRES_TYPE(bf16_t, false) }
    else if (dtype == MNX_DT_F16X3) { MNX_GRES_TYPE(f16_t, true) }
    else if (dtype == MNX_DT_BF16X3) { MNX_GRES_TYPE(bf16_t, true) }
    else return hipErrorInvalidValue;
#undef MNX_GRES_TYPE
#undef MNX_GRES_CASE
    return hipGetLastError();
}

}  // namespace mnx
