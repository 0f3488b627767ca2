// tools/probes/mfma_rate.hip — what the matrix pipes of MI355X sustain when every CU runs them, and when half do.
//   hipcc --offload-arch=gfx950 -O2 mfma_rate.hip -o mfma_rate && ./mfma_rate
// Register-only loop of independent v_mfma_f32_16x16x32_f16 (8 accumulators per wave, no memory traffic), one or two waves
// per SIMD, on 64 / 128 / 256 CUs for ~2 ms and for ~50 ms (power management reacts in milliseconds). The dense peak of
// MI355X_MICROARCH.md (2.5 PFLOP/s) is 256 CUs x 4 SIMDs x 1024 flop/clk x 2.4 GHz; this prints the fraction reached.
#include <hip/hip_runtime.h>

#include <cstdio>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[0] = s;      // never true: keeps the loop alive
}

int main() {
    float* out;
    if (hipMalloc(&out, 4) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int iters : {40000, 1000000})
        for (int waves : {4, 8})
            for (int cus : {64, 128, 256}) {
                hipLaunchKernelGGL(mfma_loop, dim3(cus), dim3(waves * 64), 0, 0, out, 1000);
                (void)hipEventRecord(e0, 0);
                hipLaunchKernelGGL(mfma_loop, dim3(cus), dim3(waves * 64), 0, 0, out, iters);
                (void)hipEventRecord(e1, 0);
                (void)hipEventSynchronize(e1);
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, e0, e1);
                const double flop = (double)cus * waves * iters * 8.0 * 16.0 * 16.0 * 32.0 * 2.0;
                const double pf = flop / (ms * 1e-3) * 1e-15;
                printf("%3d workgroups x %d waves, %7d iterations: %8.2f ms  %6.3f PFLOP/s  = %.3f of 2.5 x (CUs / 256)  per-CU clock-equivalent %.2f GHz\n",
                       cus, waves, iters, ms, pf, pf / (2.5 * cus / 256.0), pf * 1e15 / (cus * 4.0 * 1024.0) * 1e-9);
            }
    return 0;
}
