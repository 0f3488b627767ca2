// tools/attn_lab/lab.hip — where a window-attention workgroup spends its life (standalone, no Python, no engine).
//
//   make -C tools/attn_lab && tools/attn_lab/lab [images=224] [stage=3] [iters=10] [stamps=0]
//
// Checks the dispatched split-operand kernel (the persistent, double-buffered one) bit for bit against the
// one-(window, head)-per-workgroup kernel and times both. With stamps=1 it also shows where the per-item kernel's
// workgroups spend their life: encoder.hip is compiled with MNX_ATTN_STAMP defined, wave 0 of every workgroup records the constant-rate wall clock (100 MHz) at eight points. Prints the launch time, the HBM rate, and the mean
// cycles between stamps over all workgroups of the last launch:
//   0 start | 1 row table + bias table in LDS | 2 K / V / q loads issued | 3 K, V^T in LDS (loads landed, transposed)
//   4 S = K.Q^T issued | 5 softmax done | 6 O = V^T.P^T issued | 7 context stored
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ unsigned long long* g_stamps;
#define MNX_ATTN_STAMP(i)                                                                                  \
    do {                                                                                                   \
        if (threadIdx.x == 0 && g_stamps) g_stamps[(size_t)blockIdx.x * 8 + (i)] = wall_clock64(); \
    } while (0)
#include "encoder.hip"

#define CK(x)                                                                                     \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) {                                                                   \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));   \
            exit(1);                                                                              \
        }                                                                                         \
    } while (0)

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

int main(int argc, char** argv) {
    const int images = argc > 1 ? atoi(argv[1]) : 224, stage = argc > 2 ? atoi(argv[2]) : 3, iters = argc > 3 ? atoi(argv[3]) : 10;
    const bool stamp = argc > 4 && atoi(argv[4]);
    const int res = 96 >> (stage - 1), C = 128 << (stage - 1), heads = 4 << (stage - 1);
    const size_t M = (size_t)images * res * res, nq = M * 3 * C, no = M * C;
    f16_t *qkv, *out; float* table; unsigned long long* stamps;
    CK(hipMalloc(&qkv, 2 * nq * sizeof(f16_t)));
    CK(hipMalloc(&out, 2 * no * sizeof(f16_t)));
    CK(hipMalloc(&table, 529 * heads * sizeof(float)));
    const int wgs = images * (res / 12) * (res / 12) * heads;
    CK(hipMalloc(&stamps, (size_t)wgs * 8 * sizeof(unsigned long long)));
    hipLaunchKernelGGL(fill_f16, dim3(2048), dim3(256), 0, 0, qkv, nq, 11u, 1.0f);
    hipLaunchKernelGGL(fill_f16, dim3(2048), dim3(256), 0, 0, qkv + nq, nq, 12u, 1.0f / 2048.f);
    hipLaunchKernelGGL(fill_f32, dim3(64), dim3(256), 0, 0, table, (size_t)529 * heads, 13u, 0.5f);
    CK(hipDeviceSynchronize());
    const double bytes = (double)(2 * nq + 2 * no) * sizeof(f16_t);
    int resident = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, mnx::window_attn_split_kernel<f16_t>, 576, 0));
    printf("runtime occupancy: %d workgroups of 576 threads per CU\n", resident);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f16_t* out_ref;
    CK(hipMalloc(&out_ref, 2 * no * sizeof(f16_t)));
    const dim3 grid_old(wgs);
    auto run_old = [&](f16_t* o, int shift) {
        hipLaunchKernelGGL((mnx::window_attn_split_kernel<f16_t>), grid_old, dim3(576), 0, 0, (const f16_t*)qkv, nq, table, o, no,
                           res, res, C, heads, shift, 3);
        CK(hipGetLastError());
    };
    auto run_new = [&](f16_t* o, int shift) {
        CK(mnx::launch_window_attn(mnx::MNX_DT_F16X3, qkv, table, o, images, res, res, C, heads, shift, 0, nq, no, 3));
    };
    for (int shift = 0; shift <= 6; shift += 6) {
        unsigned long long* none = nullptr;
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &none, sizeof(none)));
        // the dispatched kernel against the one-(window, head)-per-workgroup kernel: same arithmetic, so bit-identical
        CK(hipMemset(out, 0xff, 2 * no * sizeof(f16_t)));
        CK(hipMemset(out_ref, 0xee, 2 * no * sizeof(f16_t)));
        run_old(out_ref, shift);
        run_new(out, shift);
        CK(hipDeviceSynchronize());
        {
            std::vector<unsigned short> a(2 * no), b(2 * no);
            CK(hipMemcpy(a.data(), out, a.size() * 2, hipMemcpyDeviceToHost));
            CK(hipMemcpy(b.data(), out_ref, b.size() * 2, hipMemcpyDeviceToHost));
            size_t diff = 0, first = (size_t)-1;
            for (size_t i = 0; i < a.size(); ++i)
                if (a[i] != b[i]) { if (!diff) first = i; ++diff; }
            printf("stage %d shift %d: dispatched vs per-item kernel: %zu of %zu 16-bit words differ%s\n", stage, shift, diff, a.size(),
                   diff ? " FAIL" : "");
            if (diff) printf("   first at %zu (plane %zu row %zu col %zu): %04x vs %04x\n", first, first / no, (first % no) / C, first % C,
                             a[first], b[first]);
        }
        for (int which = 0; which < 2; ++which) {
            for (int i = 0; i < 2; ++i) which ? run_new(out, shift) : run_old(out, shift);
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < iters; ++i) which ? run_new(out, shift) : run_old(out, shift);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / iters;
            printf("stage %d  %d images  shift %d  %-10s %8.1f us / launch  %.2f TB/s (%.3f of 8)\n", stage, images, shift,
                   which ? "dispatched" : "per-item", us, bytes / us * 1e-6, bytes / us * 1e-6 / 8.0);
        }
        if (!stamp) continue;
        float ms = 0.f;
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &stamps, sizeof(stamps)));
        CK(hipEventRecord(e0, 0));
        run_old(out, shift);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h((size_t)wgs * 8);
        CK(hipMemcpy(h.data(), stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long lo = ~0ull, hi = 0;
        double seg[7] = {0, 0, 0, 0, 0, 0, 0}, life = 0;
        for (int w = 0; w < wgs; ++w) {
            const unsigned long long* t = &h[(size_t)w * 8];
            if (t[0] < lo) lo = t[0];
            if (t[7] > hi) hi = t[7];
            for (int i = 0; i < 7; ++i) seg[i] += (double)(t[i + 1] - t[i]);
            life += (double)(t[7] - t[0]);
        }
        const double cyc_per_us = (double)(hi - lo) / (ms * 1e3);
        printf("  per-item kernel, stamped launch %.1f us, counter %.1f ticks/us; mean workgroup life %.2f us; %.1f workgroups in flight per CU\n",
               ms * 1e3, cyc_per_us, life / wgs / cyc_per_us, life / (double)(hi - lo) / 256.0);
        const char* names[7] = {"setup tables", "K / V loads + LDS writes", "q loads land, barrier", "S = K.Q^T", "bias, mask, softmax",
                                "split P, O = V^T.P^T", "split + store"};
        for (int i = 0; i < 7; ++i)
            printf("    %d->%d %-28s %6.2f us  %5.1f %%\n", i, i + 1, names[i], seg[i] / wgs / cyc_per_us, 100.0 * seg[i] / life);
    }
    return 0;
}
