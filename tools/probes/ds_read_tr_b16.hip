// tools/probes/ds_read_tr_b16.hip — what gfx950's LDS transpose read returns, as a table (one wave, known LDS image).
//   hipcc --offload-arch=gfx950 -O2 ds_read_tr_b16.hip -o ds_read_tr_b16 && ./ds_read_tr_b16
// LDS holds the 16-bit value i at element i. Test A: lane l passes byte address 8*l (64 contiguous 4-element chunks).
// Test B: a row-major [key][32] tile (64-byte rows): lane (g = l>>4, L = l&15) passes &tile[4g + L/4][4*(L%4)], the
// addressing the window-attention kernel wants for O^T = V^T.P^T: it should receive tile[4g + j][L] in element j.
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ void probe(unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    const int l = threadIdx.x;
    for (int i = l; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)lds;   // LDS byte address of the array
    unsigned long long a, b;
    const unsigned addr_a = base + 8u * l;
    const int g = l >> 4, L = l & 15;
    const unsigned addr_b = base + (unsigned)((4 * g + L / 4) * 64 + (L % 4) * 8);
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(a) : "v"(addr_a) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(b) : "v"(addr_b) : "memory");
    for (int j = 0; j < 4; ++j) {
        out[l * 4 + j] = (unsigned short)(a >> (16 * j));
        out[256 + l * 4 + j] = (unsigned short)(b >> (16 * j));
    }
}

int main() {
    unsigned short* d;
    unsigned short h[512];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 1;
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    int ok_a = 0, ok_b = 0;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 4; ++j) {
            ok_a += h[l * 4 + j] == (l & 15) + j * 16 + (l >> 4) * 64;
            ok_b += h[256 + l * 4 + j] == (4 * (l >> 4) + j) * 32 + (l & 15);
        }
    printf("A (contiguous chunks): %d/256 match lds[(l&15) + 16 j + 64 (l>>4)]\n", ok_a);
    printf("B (row-major [key][32] tile): %d/256 match tile[4 (l>>4) + j][l&15]\n", ok_b);
    for (int l = 0; l < 64; l += 5)
        printf("  lane %2d  A: %4d %4d %4d %4d   B: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3],
               h[256 + l * 4], h[256 + l * 4 + 1], h[256 + l * 4 + 2], h[256 + l * 4 + 3]);
    return 0;
}
