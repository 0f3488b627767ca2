// preprocess.hip — the transform in front of the encoder, on device:
//   CropWhite(pad) [-> PadToSquare] -> Resize(S,S, bilinear) -> ToGray -> Normalize(ImageNet) -> CHW fp32
//   reference MolNexTR/dataset.py:158-185 (augment=False), MolNexTR/data_aug.py:98-143, MolNexTR/model.py:104.
// Integer / byte work, HBM-trivial (one pass over the page for the bounding box, then 4 taps per output pixel).
// It computes bit for bit what molnextr_amd/preprocess.py computes (the host restatement of albumentations 1.1.0 /
// OpenCV semantics; neither library exists in the build image, so both are "parity unpinned" against the reference).
#include "common.h"
#include "kernels.h"

namespace mnx {

// bbox = {min row, max row, min col, max col} of pixels that differ from white in any channel; max = -1 when blank
__global__ void prep_bbox_init_kernel(int* bbox, int H, int W) {
    if (threadIdx.x == 0) { bbox[0] = H; bbox[1] = -1; bbox[2] = W; bbox[3] = -1; }
}

__global__ __launch_bounds__(256) void prep_bbox_kernel(const uint8_t* __restrict__ rgb, int H, int W, int* bbox) {
    __shared__ int s_min[4], s_max[4];
    const int y = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint8_t* row = rgb + (size_t)y * W * 3;
    int mn = W, mx = -1;
    for (int x = tid; x < W; x += 256) {
        const bool ink = row[3 * x] != 255 || row[3 * x + 1] != 255 || row[3 * x + 2] != 255;
        if (ink) { mn = min(mn, x); mx = max(mx, x); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = min(mn, __shfl_xor(mn, o, 64));
        mx = max(mx, __shfl_xor(mx, o, 64));
    }
    if (lane == 0) { s_min[wave] = mn; s_max[wave] = mx; }
    __syncthreads();
    if (tid == 0) {
        mn = min(min(s_min[0], s_min[1]), min(s_min[2], s_min[3]));
        mx = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
        if (mx >= 0) {
            atomicMin(&bbox[0], y); atomicMax(&bbox[1], y);
            atomicMin(&bbox[2], mn); atomicMax(&bbox[3], mx);
        }
    }
}

// cv2.resize(INTER_LINEAR) tap for 8-bit images: source index, neighbour, and the two 11-bit weights
__device__ __forceinline__ void linear_tap(int d, int src, int dst, int& i0, int& i1, int& w0, int& w1) {
    const double scale = 1.0 / ((double)dst / (double)src);
    // separate multiply and subtract (no fused multiply-add): the host restatement rounds twice
    const float fx = (float)__dsub_rn(__dmul_rn((double)d + 0.5, scale), 0.5);
    int sx = (int)floorf(fx);
    float frac = fx - (float)sx;
    if (sx < 0) { frac = 0.f; sx = 0; }
    if (sx >= src - 1) { frac = 0.f; sx = src - 1; }
    w0 = (int)rintf((1.0f - frac) * 2048.0f);
    w1 = (int)rintf(frac * 2048.0f);
    i0 = sx;
    i1 = min(sx + 1, src - 1);
}

struct PrepArgs {
    const uint8_t* rgb;
    const int* bbox;
    int* crop_out;       // or null: {crop_top, crop_bottom, crop_left, crop_right} as CropWhite.update_params reports
    float* out;          // [3, S, S]
    int H, W, pad, S, square;
    float mean255[3], inv[3];
};

__global__ __launch_bounds__(256) void prep_resize_kernel(PrepArgs a) {
    const int dx = blockIdx.x * 16 + (threadIdx.x & 15), dy = blockIdx.y * 16 + (threadIdx.x >> 4);
    int top = 0, bottom = a.H, left = 0, right = a.W;
    if (a.bbox[1] >= 0) { top = a.bbox[0]; bottom = a.bbox[1] + 1; left = a.bbox[2]; right = a.bbox[3] + 1; }
    const int hc = bottom - top, wc = right - left;
    if (a.crop_out && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        a.crop_out[0] = top; a.crop_out[1] = a.H - bottom; a.crop_out[2] = left; a.crop_out[3] = a.W - right;
    }
    int Hp = hc + 2 * a.pad, Wp = wc + 2 * a.pad, pad_t = a.pad, pad_l = a.pad;
    if (a.square) {      // PadToSquare after CropWhite (reference data_aug.py:286-301): diff//2 first, the rest after
        const int diff = Hp > Wp ? Hp - Wp : Wp - Hp;
        if (Hp <= Wp) { pad_t += diff / 2; Hp = Wp; } else { pad_l += diff / 2; Wp = Hp; }
    }
    if (dx >= a.S || dy >= a.S) return;
    int y0, y1, wy0, wy1, x0, x1, wx0, wx1;
    linear_tap(dy, Hp, a.S, y0, y1, wy0, wy1);
    linear_tap(dx, Wp, a.S, x0, x1, wx0, wx1);
    auto px = [&](int y, int x, int c) -> int {      // the cropped page with its white border, never materialised
        y -= pad_t; x -= pad_l;
        if (y < 0 || y >= hc || x < 0 || x >= wc) return 255;
        return a.rgb[((size_t)(top + y) * a.W + left + x) * 3 + c];
    };
    int ch[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int r0 = px(y0, x0, c) * wx0 + px(y0, x1, c) * wx1;
        const int r1 = px(y1, x0, c) * wx0 + px(y1, x1, c) * wx1;
        const int v = (((wy0 * (r0 >> 4)) >> 16) + ((wy1 * (r1 >> 4)) >> 16) + 2) >> 2;
        ch[c] = min(max(v, 0), 255);
    }
    const int gray = (ch[0] * 4899 + ch[1] * 9617 + ch[2] * 1868 + 8192) >> 14;     // cv2 RGB2GRAY
    const float g = (float)(gray & 255);
#pragma unroll
    for (int c = 0; c < 3; ++c) a.out[((size_t)c * a.S + dy) * a.S + dx] = (g - a.mean255[c]) * a.inv[c];
}

hipError_t launch_preprocess(const uint8_t* rgb, int H, int W, int pad, int square, int S, int* bbox, int* crop_out,
                             float* out, hipStream_t s) {
    hipLaunchKernelGGL(prep_bbox_init_kernel, dim3(1), dim3(64), 0, s, bbox, H, W);
    hipLaunchKernelGGL(prep_bbox_kernel, dim3(H), dim3(256), 0, s, rgb, H, W, bbox);
    PrepArgs a;
    a.rgb = rgb; a.bbox = bbox; a.crop_out = crop_out; a.out = out; a.H = H; a.W = W; a.pad = pad; a.S = S; a.square = square;
    const float mean[3] = {0.485f, 0.456f, 0.406f}, sd[3] = {0.229f, 0.224f, 0.225f};   // IMAGENET_DEFAULT_MEAN / STD
    for (int c = 0; c < 3; ++c) {
        volatile float m = mean[c] * 255.0f, d = sd[c] * 255.0f;   // fp32 products, as numpy computes them
        a.mean255[c] = m;
        a.inv[c] = 1.0f / d;
    }
    hipLaunchKernelGGL(prep_resize_kernel, dim3((S + 15) / 16, (S + 15) / 16), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace mnx
