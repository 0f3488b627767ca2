// tools/gemm_lab/lab.hip — standalone A/B bench of the encoder GEMM kernels on the Swin-B shapes (no Python, no engine).
//
//   make -C tools/gemm_lab && tools/gemm_lab/lab [images_per_group=64] [iters=20] [only-names,comma-separated|-] [bf16|fp16x3|fp16x2]
//
// Mode fp16x3 times the split-operand form of the same kernels (hi + lo planes, three MFMA terms per product; TFLOP/s are
// ALGORITHMIC, 2*M*N*K): timing only — its arithmetic is checked element by element in tests/test_gpu_parity.py. Mode fp16x2:
// the same on TWO terms (SplitArgs::terms = 2: the activation's lo plane is not read; the GELU epilogue writes one plane) —
// what compute_dtype FP16X3M runs in the layers of its table.
//
// For every (stage, layer) GEMM shape of an encoder group it checks sampled output rows of both kernels against a naive
// fp32 reference (same 16-bit inputs) and prints microseconds and TFLOP/s of
//   base = the 128x128 kernel of gemm.hip          (mnx::launch_gemm16_tile128)
//   disp = what the product dispatches              (mnx::launch_gemm16: gemm256.hip where gemm256_supports(), gemm_res.hip
//                                                    where gemm_res_supports())
// The table under profiles/ (r02_gemm_shapes.md) is this program's output.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

#include "common.h"
#include "kernels.h"

#define CK(x)                                                                                     \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) {                                                                   \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));   \
            exit(1);                                                                              \
        }                                                                                         \
    } while (0)

__global__ void fill_bf16(bf16_t* p, size_t n, unsigned seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)(i * 2654435761u) ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (bf16_t)(((int)(h & 0xffff) - 32768) * (scale / 32768.f));
    }
}
__global__ void fill_f16(f16_t* p, size_t n, unsigned seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)(i * 2654435761u) ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (f16_t)(((int)(h & 0xffff) - 32768) * (scale / 32768.f));
    }
}
__global__ void fill_f32(float* p, size_t n, unsigned seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)(i * 2654435761u) ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = ((int)(h & 0xffff) - 32768) * (scale / 32768.f);
    }
}
// background HBM traffic for the 'stream' mode: dst = src, 16 bytes per lane
__global__ void stream_copy(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// reference for sampled rows: out[s][n] = epi(sum_k A[row_s][k] * W[n][k] + bias[n]) (+ resid)
__global__ void ref_rows(const bf16_t* A, const bf16_t* W, const float* bias, const float* resid, const int* rows,
                         int S, int N, int K, int epi, float* out) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, s = blockIdx.y;
    if (n >= N) return;
    const int m = rows[s];
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc += (float)A[(size_t)m * K + k] * (float)W[(size_t)n * K + k];
    acc += bias[n];
    if (epi == 1) acc = 0.5f * acc * (1.0f + erff(acc * 0.70710678f));
    if (epi == 2) acc += resid[(size_t)m * N + n];
    out[(size_t)s * N + n] = acc;
}

// split modes: the dispatched kernels must reproduce the 128x128 kernel bit for bit (they add the same fp32 numbers in the same
// order: DESIGN.md 4.3) — every 4-byte word of both outputs is compared
__global__ void cmp_words(const unsigned* a, const unsigned* b, size_t n, unsigned long long* out) {
    unsigned long long bad = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) bad += a[i] != b[i];
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o, 64);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(out, bad);
}

struct Shape { const char* name; int epi, M, N, K; };

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64;
    const int iters = argc > 2 ? atoi(argv[2]) : 20;
    const char* only = (argc > 3 && strcmp(argv[3], "-")) ? argv[3] : nullptr;
    const bool two = argc > 4 && !strcmp(argv[4], "fp16x2");
    const bool split = two || (argc > 4 && !strcmp(argv[4], "fp16x3"));
    // argv[5] = "stream": while the dispatched kernel is timed, a second stream copies 1 GiB blocks (HBM read + write) with
    // argv[6] workgroups (default 1024): does the GEMM share a bottleneck with plain HBM traffic?
    const bool bg = argc > 5 && !strcmp(argv[5], "stream");
    const int bg_wgs = argc > 6 ? atoi(argv[6]) : 1024;
    hipStream_t st2 = nullptr;
    float4 *bsrc = nullptr, *bdst = nullptr;
    const size_t bn = (size_t)1 << 26;      // float4 elements = 1 GiB
    hipEvent_t b0, b1;
    if (bg) {
        CK(hipStreamCreate(&st2));
        CK(hipMalloc(&bsrc, bn * 16)); CK(hipMalloc(&bdst, bn * 16));
        CK(hipMemset(bsrc, 1, bn * 16));
        CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(stream_copy, dim3(bg_wgs), dim3(256), 0, st2, bsrc, bdst, bn);
        CK(hipEventRecord(b0, st2));
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(stream_copy, dim3(bg_wgs), dim3(256), 0, st2, bsrc, bdst, bn);
        CK(hipEventRecord(b1, st2));
        CK(hipEventSynchronize(b1));
        float ms;
        CK(hipEventElapsedTime(&ms, b0, b1));
        printf("background copy alone: %.2f TB/s (read + write), %d workgroups\n", 10 * 2.0 * bn * 16 / (ms * 1e-3) * 1e-12, bg_wgs);
    }
    const int dt = split ? mnx::MNX_DT_F16X3 : mnx::MNX_DT_BF16;
    // environment: MNX_LAB_ZERO=1 all operands zero (what does the clock do without toggling data?), MNX_LAB_CUS=n persistent
    // launches on n workgroups, MNX_LAB_NOBASE=1 the 128x128 kernel is run once (for the comparison) but not timed
    const bool zero = getenv("MNX_LAB_ZERO") && atoi(getenv("MNX_LAB_ZERO"));
    const bool nobase = getenv("MNX_LAB_NOBASE") && atoi(getenv("MNX_LAB_NOBASE"));
    if (getenv("MNX_LAB_CUS")) mnx::set_persistent_cus(atoi(getenv("MNX_LAB_CUS")));
    printf("persistent workgroups %d%s\n", mnx::persistent_cus(), zero ? ", ZERO operands" : "");
    std::vector<Shape> shapes;
    const int L[4] = {9216, 2304, 576, 144}, C[4] = {128, 256, 512, 1024};
    static char names[64][32];
    int ni = 0;
    for (int si : {0, 1, 2, 3}) {
        const int M = B * L[si], c = C[si];
        snprintf(names[ni], 32, "qkv s%d", si); shapes.push_back({names[ni++], 0, M, 3 * c, c});
        snprintf(names[ni], 32, "proj s%d", si); shapes.push_back({names[ni++], 2, M, c, c});
        snprintf(names[ni], 32, "fc1 s%d", si); shapes.push_back({names[ni++], 1, M, 4 * c, c});
        snprintf(names[ni], 32, "fc2 s%d", si); shapes.push_back({names[ni++], 2, M, c, 4 * c});
        if (si < 3) { snprintf(names[ni], 32, "merge s%d", si); shapes.push_back({names[ni++], 3, M / 4, 2 * c, 4 * c}); }
    }
    shapes.push_back({"sq8192", 0, 8192, 8192, 8192});
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("images per group %d, %d launches per timing, operands %s\n", B, iters, two ? "fp16x2 (split, 2 MFMA terms: activation lo plane dropped; algorithmic TFLOP/s)" : split ? "fp16x3 (split, 3 MFMA terms; algorithmic TFLOP/s)" : "bf16");
    printf("%-9s epi %8s %6s %6s | %10s %8s | %10s %8s %-6s| max |err| vs fp32 reference (base, disp)\n", "shape", "M", "N",
           "K", "base us", "TF", "disp us", "TF", "kernel");
    for (const Shape& sh : shapes) {
        if (only && !strstr(only, sh.name)) continue;
        const bool out16 = sh.epi < 2;
        const size_t nA = (size_t)sh.M * sh.K, nW = (size_t)sh.N * sh.K, nC = (size_t)sh.M * sh.N;
        bf16_t *A, *W;
        float *bias, *resid0;
        void* Cc;
        const int planes = split ? 2 : 1;
        CK(hipMalloc(&A, nA * 2 * planes)); CK(hipMalloc(&W, nW * 2 * planes)); CK(hipMalloc(&bias, sh.N * 4));
        CK(hipMalloc(&Cc, nC * (out16 ? 2 * planes : 4)));
        CK(hipMalloc(&resid0, sh.epi == 2 ? nC * 4 : 16));
        if (split) {    // hi planes of unit size, lo planes 2^-11 of that (what a rounding residual looks like)
            fill_f16<<<2048, 256, 0, st>>>((f16_t*)A, nA, 1u, 1.0f);
            fill_f16<<<2048, 256, 0, st>>>((f16_t*)A + nA, nA, 5u, 1.0f / 2048.f);
            fill_f16<<<2048, 256, 0, st>>>((f16_t*)W, nW, 2u, 1.0f / sqrtf((float)sh.K));
            fill_f16<<<2048, 256, 0, st>>>((f16_t*)W + nW, nW, 6u, 1.0f / sqrtf((float)sh.K) / 2048.f);
        } else {
            fill_bf16<<<2048, 256, 0, st>>>(A, nA, 1u, 1.0f);
            fill_bf16<<<2048, 256, 0, st>>>(W, nW, 2u, 1.0f / sqrtf((float)sh.K));
        }
        if (zero) { CK(hipMemsetAsync(A, 0, nA * 2 * planes, st)); CK(hipMemsetAsync(W, 0, nW * 2 * planes, st)); }
        fill_f32<<<64, 256, 0, st>>>(bias, sh.N, 3u, 0.5f);
        if (sh.epi == 2) fill_f32<<<2048, 256, 0, st>>>(resid0, nC, 4u, 1.0f);
        const int S = 24;
        std::vector<int> rows(S);
        for (int i = 0; i < S; ++i) rows[i] = (int)(((long long)i * 2654435761ll + 17) % sh.M);
        rows[0] = 0; rows[1] = sh.M - 1; rows[2] = sh.M / 2 + 255 < sh.M ? sh.M / 2 + 255 : sh.M - 1;
        int* drows; float* ref;
        CK(hipMalloc(&drows, S * 4)); CK(hipMalloc(&ref, (size_t)S * sh.N * 4));
        CK(hipMemcpyAsync(drows, rows.data(), S * 4, hipMemcpyHostToDevice, st));
        ref_rows<<<dim3((sh.N + 127) / 128, S), 128, 0, st>>>(A, W, bias, resid0, drows, S, sh.N, sh.K, sh.epi, ref);
        std::vector<float> href((size_t)S * sh.N);
        CK(hipMemcpyAsync(href.data(), ref, href.size() * 4, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        mnx::SplitArgs sp;
        sp.a_lo = nA; sp.w_lo = nW; sp.c_lo = out16 ? nC : 0; sp.oscale = 1.0f; sp.terms = two ? 2 : 3;
        if (two && sh.epi == 1) sp.c_planes = 1;          // fc1 -> fc2 on two terms: the hi plane only
        double mhz0 = 0.0;
        (void)mnx::x3_clock_read(&mhz0, true);
        auto launch = [&](int v) {
            const float* resid = sh.epi == 2 ? (const float*)Cc : nullptr;      // in-place residual, as the encoder runs it
            return v ? mnx::launch_gemm16(dt, sh.epi, A, W, Cc, bias, resid, sh.M, sh.N, sh.K, st, split ? &sp : nullptr)
                     : mnx::launch_gemm16_tile128(dt, sh.epi, A, W, Cc, bias, resid, sh.M, sh.N, sh.K, st, split ? &sp : nullptr);
        };
        double us[2] = {0, 0}, err[2] = {0, 0};
        const size_t cbytes = nC * (out16 ? 2 * (two && sh.epi == 1 ? 1 : planes) : 4);
        void* Cb = nullptr;                       // split modes: the 128x128 kernel's output, for the word-by-word comparison
        unsigned long long* dbad = nullptr;
        unsigned long long bad = 0;
        if (split) { CK(hipMalloc(&Cb, cbytes)); CK(hipMalloc(&dbad, 8)); CK(hipMemsetAsync(dbad, 0, 8, st)); }
        for (int v = 0; v < 2; ++v) {
            if (sh.epi == 2) CK(hipMemcpyAsync(Cc, resid0, nC * 4, hipMemcpyDeviceToDevice, st));
            CK(launch(v));
            if (split && v == 0) CK(hipMemcpyAsync(Cb, Cc, cbytes, hipMemcpyDeviceToDevice, st));
            if (split && v == 1) {
                cmp_words<<<2048, 256, 0, st>>>((const unsigned*)Cb, (const unsigned*)Cc, cbytes / 4, dbad);
                CK(hipMemcpyAsync(&bad, dbad, 8, hipMemcpyDeviceToHost, st));
                CK(hipStreamSynchronize(st));
            }
            if (nobase && v == 0) continue;
            std::vector<char> hrow((size_t)sh.N * 4);
            for (int i = 0; i < S && !split; ++i) {
                CK(hipMemcpyAsync(hrow.data(), (char*)Cc + (size_t)rows[i] * sh.N * (out16 ? 2 : 4), (size_t)sh.N * (out16 ? 2 : 4),
                                  hipMemcpyDeviceToHost, st));
                CK(hipStreamSynchronize(st));
                for (int n = 0; n < sh.N; ++n) {
                    float got;
                    if (out16) { unsigned u = ((unsigned)((unsigned short*)hrow.data())[n]) << 16; memcpy(&got, &u, 4); }
                    else got = ((float*)hrow.data())[n];
                    const double d = fabs((double)got - href[(size_t)i * sh.N + n]);
                    if (d > err[v]) err[v] = d;
                }
            }
            for (int i = 0; i < 3; ++i) CK(launch(v));     // warm (epi 2 keeps accumulating in place: values irrelevant here)
            int bg_n = 0;
            if (bg && v == 1) {
                CK(hipStreamSynchronize(st));
                bg_n = (int)(us[0] * iters / 450.0) + 4;          // ~0.45 ms per 1 GiB copy alone: enough to outlast the timed launches
                CK(hipEventRecord(b0, st2));
                for (int i = 0; i < bg_n; ++i) hipLaunchKernelGGL(stream_copy, dim3(bg_wgs), dim3(256), 0, st2, bsrc, bdst, bn);
                CK(hipEventRecord(b1, st2));
            }
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i) CK(launch(v));
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (bg && v == 1) {
                CK(hipEventSynchronize(b1));
                float bms;
                CK(hipEventElapsedTime(&bms, b0, b1));
                printf("   with background copy: %d x 1 GiB in %.2f ms = %.2f TB/s over its whole run (GEMM part %.2f ms)\n", bg_n, bms,
                       bg_n * 2.0 * bn * 16 / (bms * 1e-3) * 1e-12, ms);
            }
            us[v] = ms * 1000.0 / iters;
        }
        const double fl = 2.0 * sh.M * sh.N * sh.K;
        const double tol = out16 ? 2e-2 : 1e-3;
        printf("%-9s %3d %8d %6d %6d | %10.1f %8.1f | %10.1f %8.1f %-6s| %.2e %.2e%s", sh.name, sh.epi, sh.M, sh.N, sh.K, us[0],
               us[0] > 0 ? fl / us[0] / 1e6 : 0.0, us[1], fl / us[1] / 1e6,
               mnx::gemm16_route(dt, sh.epi, sh.M, sh.N, sh.K, two ? 2 : 3, true), err[0], err[1],
               (err[0] > tol || err[1] > tol) ? "  FAIL" : "");
        if (split) printf(" | words differing from the 128x128 kernel: %llu of %zu%s", bad, cbytes / 4, bad ? "  MISMATCH" : "");
        double mhz = 0.0;
        (void)mnx::x3_clock_read(&mhz, true);
        if (mhz > 0.0) printf(" | gemm256x3 shader clock %.0f MHz", mhz);
        printf("\n");
        if (split) { CK(hipFree(Cb)); CK(hipFree(dbad)); }
        CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(bias)); CK(hipFree(Cc)); CK(hipFree(resid0)); CK(hipFree(drows)); CK(hipFree(ref));
    }
    return 0;
}
