// tools/probes/mfma_f16_subnormal.hip — does v_mfma_f32_16x16x32_f16 keep fp16 SUBNORMAL inputs on gfx950?
// The split-fp16 operand mode (MNX_DTYPE_FP16X3) stores lo = fp16(v - fp16(v)), which is subnormal for |v| < 0.25; if the
// matrix pipe flushed such inputs the mode would lose ~2^-14 relative accuracy. Prints the products.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_f16_subnormal.hip -o tools/probes/mfma_f16_subnormal && ./...
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void probe(const float* av, float* out) {
    // A row r (lane & 15) holds av[r] in k-slot 0 of lane group 0, zero elsewhere; B = 1.0 everywhere -> D[r][*] = av[r]
    const int lane = threadIdx.x, fr = lane & 15, fg = lane >> 4;
    f16x8 a = {}, b;
    for (int i = 0; i < 8; ++i) b[i] = (_Float16)1.0f;
    if (fg == 0) a[0] = (_Float16)av[fr];
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d, 0, 0, 0);
    // C/D: col = lane & 15, row = (lane >> 4) * 4 + i
    if ((lane & 15) == 0)
        for (int i = 0; i < 4; ++i) out[fg * 4 + i] = d[i];
}

int main() {
    float h[16], o[16];
    for (int i = 0; i < 16; ++i) h[i] = 6.103515625e-05f / (float)(1 << i);   // 2^-14 (min normal) down to 2^-29
    h[15] = 3.0e-7f;
    float *da, *dout;
    hipMalloc(&da, sizeof(h)); hipMalloc(&dout, sizeof(o));
    hipMemcpy(da, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, dout);
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    int kept = 0;
    for (int i = 0; i < 16; ++i) {
        const float want = (float)(_Float16)h[i];
        printf("in %.6e  fp16(in) %.6e  mfma %.6e  %s\n", h[i], want, o[i], o[i] == want ? "kept" : (o[i] == 0.f ? "FLUSHED" : "DIFFERENT"));
        kept += o[i] == want;
    }
    printf("subnormal fp16 inputs kept by the MFMA: %d / 16\n", kept);
    return kept == 16 ? 0 : 1;
}
