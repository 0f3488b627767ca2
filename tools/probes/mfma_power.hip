// tools/probes/mfma_power.hip — what the matrix pipes SUSTAIN on random operands, by instruction shape (round 5).
//   hipcc --offload-arch=gfx950 -O2 mfma_power.hip -o mfma_power && ./mfma_power
// tools/probes/mfma_rate.hip (round 3) multiplies nearly constant operands: 0.89 of the nominal peak. gemm256x3_kernel's
// clock stamps (tools/gemm_lab, round 5) say the chip runs real GEMM data at 1.65-1.8 GHz, zero operands at 2.39 GHz: the
// matrix pipes are POWER-limited and the power depends on the data. This probe asks the two questions a kernel author needs:
// (1) what rate do random fp16 operands sustain on all 256 CUs (register-only, no LDS / memory), and (2) is
// v_mfma_f32_32x32x16_f16 (half the operand register reads per flop, twice the accumulator traffic) cheaper per flop than
// v_mfma_f32_16x16x32_f16 — i.e. does it hold a higher clock? Operands rotate through four A and four B registers so that
// the inputs of consecutive instructions differ, as they do in a GEMM loop. Reports TFLOP/s, the shader clock
// (s_memtime / s_memrealtime) and cycles per instruction per SIMD.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ unsigned long long stamps[256 * 2];

__device__ __forceinline__ f16x8 rnd8(unsigned seed, float scale) {
    f16x8 v;
    for (int i = 0; i < 8; ++i) {
        unsigned h = (seed + i) * 2654435761u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        v[i] = (_Float16)(((int)(h & 0xffff) - 32768) * (scale / 32768.f));
    }
    return v;
}

// MODE 0: 16x16x32, 8 accumulators (32 registers); MODE 1: 32x32x16, 2 accumulators (32 registers). Same flop per iteration.
// bmask / bscale: the B operand as a "lo plane" — scaled down by bscale and with its low mantissa bits cleared (AND with bmask):
// does an operand with fewer significant bits cost less power?
template <int MODE>
__global__ __launch_bounds__(512) void loop(float* out, int iters, float scale, unsigned bmask = 0xffffu, float bscale = 1.f) {
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = rnd8((threadIdx.x * 8 + i) * 16 + blockIdx.x * 65536, scale);
        b[i] = rnd8((threadIdx.x * 8 + 4 + i) * 16 + blockIdx.x * 65536, scale * bscale);
        for (int j = 0; j < 8; ++j) {
            unsigned short u = __builtin_bit_cast(unsigned short, b[i][j]);
            u &= (unsigned short)bmask;
            b[i][j] = __builtin_bit_cast(_Float16, u);
        }
    }
    unsigned long long c0 = 0, r0 = 0;
    if (threadIdx.x == 0) { c0 = __builtin_readcyclecounter(); r0 = __builtin_amdgcn_s_memrealtime(); }
    float s = 0.f;
    if (MODE == 0) {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        f32x16 acc[2];
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[(i + (i >> 1)) & 3], acc[i & 1], 0, 0, 0);
        }
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 16; ++j) s += acc[i][j];
    }
    if (threadIdx.x == 0) {
        stamps[blockIdx.x * 2] = __builtin_readcyclecounter() - c0;
        stamps[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    if (s == 12345.678f) out[0] = s;      // never true: keeps the loop alive
}

int main() {
    float* out;
    if (hipMalloc(&out, 4) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 400000;       // ~25-40 ms per launch
    printf("%-12s %-8s %5s %6s | %8s %9s %8s | %s\n", "instruction", "operands", "CUs", "waves", "ms", "TFLOP/s", "of 2.5PF", "shader clock MHz (median), cycles per instruction per SIMD");
    for (int mode = 0; mode < 2; ++mode)
        for (float scale : {0.f, 1.f})
            for (int cus : {128, 256}) {
                const int waves = 8;
                auto launch = [&](int n) {
                    if (mode == 0) hipLaunchKernelGGL(loop<0>, dim3(cus), dim3(waves * 64), 0, 0, out, n, scale);
                    else hipLaunchKernelGGL(loop<1>, dim3(cus), dim3(waves * 64), 0, 0, out, n, scale);
                };
                launch(20000);
                (void)hipEventRecord(e0, 0);
                launch(iters);
                (void)hipEventRecord(e1, 0);
                (void)hipEventSynchronize(e1);
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, e0, e1);
                std::vector<unsigned long long> hs(512);
                (void)hipMemcpyFromSymbol(hs.data(), HIP_SYMBOL(stamps), sizeof(unsigned long long) * 512);
                std::vector<double> mhz;
                for (int w = 0; w < cus; ++w) if (hs[2 * w + 1]) mhz.push_back((double)hs[2 * w] / (double)hs[2 * w + 1] * 100.0);
                std::sort(mhz.begin(), mhz.end());
                const double clk = mhz.empty() ? 0.0 : mhz[mhz.size() / 2];
                const double flop = (double)cus * waves * iters * 8.0 * 16.0 * 16.0 * 32.0 * 2.0;
                const double tf = flop / (ms * 1e-3) * 1e-12;
                const double n_inst_per_simd = 2.0 * iters * (mode == 0 ? 8 : 4);    // two waves per SIMD
                printf("%-12s %-8s %5d %6d | %8.2f %9.1f %8.3f | %6.0f  %5.2f\n", mode == 0 ? "16x16x32" : "32x32x16", scale == 0.f ? "zero" : "random",
                       cus, waves, ms, tf, tf / 2500.0 * 256.0 / cus, clk, ms * 1e-3 * clk * 1e6 / n_inst_per_simd);
            }
    // the B operand as a lo plane: 2^-11 of A's size, mantissa cut to 10 / 6 / 4 / 2 / 0 stored bits (16x16x32, 256 CUs)
    printf("\nB operand as a lo plane (x 2^-11), low mantissa bits cleared; 16x16x32, 256 CUs x 8 waves\n");
    for (unsigned keep : {10u, 6u, 4u, 2u, 0u}) {
        const unsigned mask = 0xffffu & ~((1u << (10 - keep)) - 1u);
        hipLaunchKernelGGL(loop<0>, dim3(256), dim3(512), 0, 0, out, 20000, 1.f, mask, 1.f / 2048.f);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(loop<0>, dim3(256), dim3(512), 0, 0, out, iters, 1.f, mask, 1.f / 2048.f);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> hs(512);
        (void)hipMemcpyFromSymbol(hs.data(), HIP_SYMBOL(stamps), sizeof(unsigned long long) * 512);
        std::vector<double> mhz;
        for (int w = 0; w < 256; ++w) if (hs[2 * w + 1]) mhz.push_back((double)hs[2 * w] / (double)hs[2 * w + 1] * 100.0);
        std::sort(mhz.begin(), mhz.end());
        const double flop = 256.0 * 8 * iters * 8.0 * 16.0 * 16.0 * 32.0 * 2.0;
        printf("  stored mantissa bits %2u (mask 0x%04x): %8.2f ms  %8.1f TFLOP/s  clock %.0f MHz\n", keep, mask, ms, flop / (ms * 1e-3) * 1e-12,
               mhz.empty() ? 0.0 : mhz[mhz.size() / 2]);
    }
    // the same with A masked too would be the hi.hi term of bf16-like operands: A and B with 7 stored mantissa bits
    return 0;
}
