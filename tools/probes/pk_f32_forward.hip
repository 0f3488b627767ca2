// tools/probes/pk_f32_forward.hip — does a packed fp32 result forward correctly into the instructions right behind it on gfx950?
// Round 6 found that window_attn_pipe_kernel with compiler-generated v_pk_add_f32 (exponent arguments) / v_pk_fma_f32 (scores)
// produced a few thousand to 10^5 wrong output words of 7.5e7, different ones on every run (profiles/r06_attn_lab_diet.txt).
// This probe isolates the pattern of that ISA —  v_pk_add_f32 v[a:a+1], .. ; v_exp_f32 vX, v[a] ; v_exp_f32 vY, v[a+1]  — with
// 0 / 1 / 2 wait states between the packed instruction and its consumers, and counts results that differ from the scalar form
// (v_sub_f32 ; v_exp_f32), in workgroups of 9 waves, two per CU, with and without an MFMA stream in the same waves.
//   hipcc --offload-arch=gfx950 -O3 pk_f32_forward.hip -o pk_f32_forward && ./pk_f32_forward
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define PK_BODY(NOPSTR)                                                                                         \
    asm volatile("v_mov_b32 v100, %2\n\t"                                                                        \
                 "v_mov_b32 v101, %3\n\t"                                                                        \
                 "v_add_f32_e32 v102, 0xc1200000, %4\n\t"                                                        \
                 "v_pk_add_f32 v[100:101], v[100:101], v[102:103] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t" \
                 NOPSTR                                                                                          \
                 "v_exp_f32_e32 %0, v100\n\t"                                                                    \
                 "v_exp_f32_e32 %1, v101"                                                                        \
                 : "=&v"(e0), "=&v"(e1) : "v"(a), "v"(b), "v"(m) : "v100", "v101", "v102", "v103")

template <int NOPS, bool MFMA>
__global__ __launch_bounds__(576) void probe(const float* __restrict__ in, unsigned* __restrict__ bad, int iters) {
    const int t = blockIdx.x * 576 + threadIdx.x;
    float a = in[2 * t], b = in[2 * t + 1];
    unsigned nbad = 0;
    f32x4 macc = {0.f, 0.f, 0.f, 0.f};
    f16x8 ma, mb;
    for (int i = 0; i < 8; ++i) { ma[i] = (_Float16)(a * 0.01f + i); mb[i] = (_Float16)(b * 0.01f - i); }
    for (int it = 0; it < iters; ++it) {
        const float m = a * 0.5f + b * 0.25f;
        float e0, e1;
        if (MFMA) macc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ma, mb, macc, 0, 0, 0);
        if (NOPS == 0) PK_BODY("");
        else if (NOPS == 1) PK_BODY("s_nop 0\n\t");
        else PK_BODY("s_nop 1\n\t");
        const float mm = m - 10.0f;
        float da = a - mm, db = b - mm;
        asm volatile("" : "+v"(da), "+v"(db));
        const float s0 = __builtin_amdgcn_exp2f(da), s1 = __builtin_amdgcn_exp2f(db);
        nbad += (__float_as_uint(s0) != __float_as_uint(e0)) + (__float_as_uint(s1) != __float_as_uint(e1));
        a = a * 0.999f + 0.013f * (float)(it & 7);
        b = b * 1.001f - 0.017f * (float)(it & 3);
    }
    bad[t] = nbad + (macc[0] == 12345.678f ? 1u : 0u);
}

template <int NOPS, bool MFMA>
static void run(const float* d_in, unsigned* d_bad, int wgs, int iters, const char* what) {
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((probe<NOPS, MFMA>), dim3(wgs), dim3(576), 0, 0, d_in, d_bad, iters);
        std::vector<unsigned> h((size_t)wgs * 576);
        hipMemcpy(h.data(), d_bad, h.size() * 4, hipMemcpyDeviceToHost);
        unsigned long long tot = 0, thr = 0;
        for (unsigned v : h) { tot += v; thr += v != 0; }
        printf("%-34s run %d: %llu of %llu results differ from the scalar form (%llu threads)\n", what, rep, tot,
               (unsigned long long)h.size() * iters * 2, thr);
    }
}

int main() {
    const int wgs = 512 * 4, iters = 4096;
    std::vector<float> in((size_t)wgs * 576 * 2);
    unsigned s = 12345u;
    for (auto& v : in) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f * 8.0f - 4.0f; }
    float* d_in; unsigned* d_bad;
    hipMalloc(&d_in, in.size() * 4);
    hipMalloc(&d_bad, (size_t)wgs * 576 * 4);
    hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice);
    run<0, false>(d_in, d_bad, wgs, iters, "pk_add -> exp, 0 wait states");
    run<1, false>(d_in, d_bad, wgs, iters, "pk_add -> exp, 1 wait state");
    run<2, false>(d_in, d_bad, wgs, iters, "pk_add -> exp, 2 wait states");
    run<0, true>(d_in, d_bad, wgs, iters, "pk_add -> exp, 0 ws, + MFMA stream");
    run<1, true>(d_in, d_bad, wgs, iters, "pk_add -> exp, 1 ws, + MFMA stream");
    run<2, true>(d_in, d_bad, wgs, iters, "pk_add -> exp, 2 ws, + MFMA stream");
    return 0;
}
