// tools/probes/grid_barrier.hip — what a device-wide barrier among co-resident workgroups costs on MI355X.
//   hipcc --offload-arch=gfx950 -O2 grid_barrier.hip -o grid_barrier && ./grid_barrier
// The decode tick is 50 dependent launches of 5-10 us each (decode tick). A persistent tick kernel would
// replace every kernel boundary by a barrier among its workgroups; this prints the price of that barrier (central
// counter, agent-scope atomics, bounded spin) for G workgroups of 256 threads spread over the chip, and for G
// workgroups confined to ONE XCD (ids = 0 mod 8 of an 8G grid: workgroups are dealt round-robin to the 8 XCDs).
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ void barrier_loop(unsigned* counter, unsigned* fail, int iters, int stride, int members) {
    if (blockIdx.x % stride) return;
    for (int it = 0; it < iters; ++it) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(it + 1) * (unsigned)members;
            int spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want)
                if (++spins > (1 << 22)) { *fail = 1; break; }
        }
        __syncthreads();
    }
}
__global__ void empty_kernel(unsigned* p) { if (p == nullptr) __builtin_trap(); }

int main() {
    unsigned *counter, *fail;
    if (hipMalloc(&counter, 8) != hipSuccess) return 1;
    fail = counter + 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int one_xcd = 0; one_xcd < 2; ++one_xcd)
        for (int G : {8, 32, 64, 128, 256, 512}) {
            if (one_xcd && G > 64) continue;
            const int stride = one_xcd ? 8 : 1;
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipMemset(counter, 0, 8);
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(barrier_loop, dim3(G * stride), dim3(256), 0, 0, counter, fail, iters, stride, G);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms = 0.f;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            unsigned f = 0;
            hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
            printf("%s  %3d workgroups: %.2f us per barrier%s\n", one_xcd ? "one XCD  " : "whole chip", G, best * 1e3f / iters,
                   f ? "  (spin limit hit: not co-resident?)" : "");
        }
    // for scale: back-to-back dependent empty launches on one stream (not a graph)
    hipEventRecord(e0, 0);
    for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, 0, counter);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    printf("1000 dependent empty launches (stream, no graph): %.2f us each\n", ms);
    return 0;
}
