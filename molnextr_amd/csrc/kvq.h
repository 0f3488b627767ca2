// kvq.h — the decoder's K / V cache rows as 24-bit block fixed point (3 bytes per element + one scale per row).
//
// What is cached (reference models/decoder.py:269-276 + onmt MultiHeadedAttention layer_cache: self_keys / self_values grown
// by torch.cat every step, memory_keys / memory_values projected once per image) is read once per decoded token by every
// later step: at 640 rows of capacity the fp32 K / V stream was 55 % of a tick. A cached ROW — the 32 channels of one
// (sequence | memory block, layer, head, position) — is stored as
//
//     q[c] = rint(v[c] * 2^(23 - e))  in [-2^23, 2^23),   2^e > max_c |v[c]|        (one exponent per row)
//     hi[c] = q[c] >> 8   (int16)        lo[c] = q[c] & 255   (uint8)        scale = 2^(e - 23)   (float)
//
// i.e. absolute error <= 2^(e - 24) = 2^-24 of the row's largest element: closer to the fp32 value than an fp16 hi + 8-bit lo
// pair on the large elements that carry a dot product, coarser on elements far below the row maximum, which do not.
// CPU emulation of the whole teacher-forced decode with every cached row rounded this way (tools/study_split_terms.py --kv
// int24b, profiles/r06_kv_block_study_*.json): log-prob error 1.1e-5 / 4.5e-6 on the two checkpoints, 0 flips in 9720 steps
// (fp32 cache: 6e-6 — summation-order noise; fp16 + 8-bit lo: 6.7e-5 / 6.2e-6; 16-bit block: 1.4e-3).
//
// The decoded value q * scale is EXACT in fp32 (|q| < 2^24, scale a power of two), and s * dot(x, q) == dot(x, s * q) bit for
// bit for a power-of-two s: readers multiply the integers and apply the scale once per row (to the score, or folded into the
// probability). Every tick form therefore sees the same numbers; the forms that keep this step's key / value in LDS round it
// through kvq_quant first, so that it is the number later steps will read.
//
// Layout of a BLOCK = the nk rows of one (owner, head) (nk = T or S rounded up to a multiple of 4):
//     [nk][32] int16 hi   |   [nk][32] uint8 lo   |   [nk] float scale          = nk * KVQ_ROW bytes, 16-byte aligned pieces
// A lane that scores key j reads 64 + 32 + 4 bytes of row j (seven requests, all in flight together); a lane that accumulates
// four value channels reads 8 + 4 + 4 bytes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace mnx {

constexpr int KVQ_ROW = 100;          // bytes per cached row of 32 channels: 64 (hi) + 32 (lo) + 4 (scale)
constexpr int KVQ_HI = 64, KVQ_LO = 32;
__host__ __device__ inline int kvq_rows(int n) { return (n + 3) & ~3; }
__host__ __device__ inline size_t kvq_block_bytes(int nk) { return (size_t)nk * KVQ_ROW; }   // nk = kvq_rows(.)

struct KvqK { uint4 hi[4]; uint4 lo[2]; float sc; };      // one row as fetched (25 registers; the fp32 row was 32)
struct KvqV { uint2 hi; unsigned lo; float sc; };           // four channels of one row as fetched

// (offsets inside a block are unsigned 32-bit — a block is at most 512 x 100 bytes —: with a wave-uniform block base the loads
//  are "scalar base + 32-bit lane offset", one address register per request instead of two)
__device__ __forceinline__ void kvq_fetch_k(KvqK& r, const char* blk, int nk, int key) {
    const unsigned oh = (unsigned)key * KVQ_HI, ol = (unsigned)nk * KVQ_HI + (unsigned)key * KVQ_LO;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.hi[i] = *(const uint4*)(blk + (oh + 16u * i));
    r.lo[0] = *(const uint4*)(blk + ol);
    r.lo[1] = *(const uint4*)(blk + (ol + 16u));
    r.sc = *(const float*)(blk + ((unsigned)nk * (KVQ_HI + KVQ_LO) + (unsigned)key * 4u));
}
__device__ __forceinline__ void kvq_fetch_v(KvqV& r, const char* blk, int nk, int key, int dq) {
    r.hi = *(const uint2*)(blk + ((unsigned)key * KVQ_HI + (unsigned)dq * 8u));
    r.lo = *(const unsigned*)(blk + ((unsigned)nk * KVQ_HI + (unsigned)key * KVQ_LO + (unsigned)dq * 4u));
    r.sc = *(const float*)(blk + ((unsigned)nk * (KVQ_HI + KVQ_LO) + (unsigned)key * 4u));
}

// the integers of a row as floats (exact): channel c -> k[c >> 2][c & 3]; multiply a dot product with them by r.sc
__device__ __forceinline__ float kvq_int(unsigned hi16 /* low 16 bits */, unsigned lo8) {
    return fmaf((float)(short)hi16, 256.0f, (float)lo8);
}
__device__ __forceinline__ void kvq_decode_k(const KvqK& r, f32x4 (&k)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {                 // hi[i]: channels 8 i .. 8 i + 7 (two per dword); lo[i >> 1]: 16 channels (four per dword)
        const unsigned h[4] = {r.hi[i].x, r.hi[i].y, r.hi[i].z, r.hi[i].w};
        const unsigned l0 = (i & 1) ? r.lo[i >> 1].z : r.lo[i >> 1].x, l1 = (i & 1) ? r.lo[i >> 1].w : r.lo[i >> 1].y;
        k[2 * i] = (f32x4){kvq_int(h[0] & 0xffffu, l0 & 0xffu), kvq_int(h[0] >> 16, (l0 >> 8) & 0xffu),
                           kvq_int(h[1] & 0xffffu, (l0 >> 16) & 0xffu), kvq_int(h[1] >> 16, l0 >> 24)};
        k[2 * i + 1] = (f32x4){kvq_int(h[2] & 0xffffu, l1 & 0xffu), kvq_int(h[2] >> 16, (l1 >> 8) & 0xffu),
                               kvq_int(h[3] & 0xffffu, (l1 >> 16) & 0xffu), kvq_int(h[3] >> 16, l1 >> 24)};
    }
}
// scale * dot(q, row): four interleaved fmaf chains (element e of quad i goes to chain e, i ascending), (s0 + s1) + (s2 + s3) —
// the decoder's canonical 32-long dot product on the row's integers, decoded eight channels at a time (the raw row is 25
// registers, a decoded one 32: decoding everything first needs both)
__device__ __forceinline__ float kvq_dot32(const f32x4 (&q)[8], const KvqK& r) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned h[4] = {r.hi[i].x, r.hi[i].y, r.hi[i].z, r.hi[i].w};
        const unsigned l0 = (i & 1) ? r.lo[i >> 1].z : r.lo[i >> 1].x, l1 = (i & 1) ? r.lo[i >> 1].w : r.lo[i >> 1].y;
        s0 = fmaf(q[2 * i][0], kvq_int(h[0] & 0xffffu, l0 & 0xffu), s0);
        s1 = fmaf(q[2 * i][1], kvq_int(h[0] >> 16, (l0 >> 8) & 0xffu), s1);
        s2 = fmaf(q[2 * i][2], kvq_int(h[1] & 0xffffu, (l0 >> 16) & 0xffu), s2);
        s3 = fmaf(q[2 * i][3], kvq_int(h[1] >> 16, l0 >> 24), s3);
        s0 = fmaf(q[2 * i + 1][0], kvq_int(h[2] & 0xffffu, l1 & 0xffu), s0);
        s1 = fmaf(q[2 * i + 1][1], kvq_int(h[2] >> 16, (l1 >> 8) & 0xffu), s1);
        s2 = fmaf(q[2 * i + 1][2], kvq_int(h[3] & 0xffffu, (l1 >> 16) & 0xffu), s2);
        s3 = fmaf(q[2 * i + 1][3], kvq_int(h[3] >> 16, l1 >> 24), s3);
    }
    return ((s0 + s1) + (s2 + s3)) * r.sc;
}
__device__ __forceinline__ f32x4 kvq_decode_v(const KvqV& r) {
    return (f32x4){kvq_int(r.hi.x & 0xffffu, r.lo & 0xffu), kvq_int(r.hi.x >> 16, (r.lo >> 8) & 0xffu),
                   kvq_int(r.hi.y & 0xffffu, (r.lo >> 16) & 0xffu), kvq_int(r.hi.y >> 16, r.lo >> 24)};
}

// Quantisation of one element given its row's max |v| (every lane of the row passes the same amax): q and the row's scale.
// amax = f * 2^e with f in [0.5, 1): |v| <= amax < 2^e. e is clamped so that 2^(23 - e) is finite (a row of zeros / denormals
// stores zeros); a non-finite row stores saturated integers (the head's log-softmax then reports the NaN / Inf as before).
__device__ __forceinline__ void kvq_quant(float v, float amax, int& q, float& scale) {
    int e;
    (void)frexpf(amax, &e);
    e = e < -100 ? -100 : (e > 120 ? 120 : e);
    const float up = __builtin_bit_cast(float, (unsigned)(23 - e + 127) << 23);     // 2^(23 - e), exact
    scale = __builtin_bit_cast(float, (unsigned)(e - 23 + 127) << 23);              // 2^(e - 23)
    const float x = fminf(fmaxf(v * up, -8388608.0f), 8388607.0f);                   // NaN -> -2^23 (fmaxf drops it)
    q = __float2int_rn(x);
}
__device__ __forceinline__ float kvq_value(int q, float scale) { return (float)q * scale; }    // exact

// store element c (0..31) of row `key` (the caller stores the scale once per row with kvq_store_scale)
__device__ __forceinline__ void kvq_store1(char* blk, int nk, int key, int c, int q) {
    *(short*)(blk + (size_t)key * KVQ_HI + c * 2) = (short)(q >> 8);
    *(unsigned char*)(blk + (size_t)nk * KVQ_HI + (size_t)key * KVQ_LO + c) = (unsigned char)(q & 255);
}
// four consecutive channels c0 .. c0 + 3 (c0 % 4 == 0)
__device__ __forceinline__ void kvq_store4(char* blk, int nk, int key, int c0, const int (&q)[4]) {
    uint2 h;
    h.x = ((unsigned)(q[0] >> 8) & 0xffffu) | ((unsigned)(q[1] >> 8) << 16);
    h.y = ((unsigned)(q[2] >> 8) & 0xffffu) | ((unsigned)(q[3] >> 8) << 16);
    *(uint2*)(blk + (size_t)key * KVQ_HI + c0 * 2) = h;
    *(unsigned*)(blk + (size_t)nk * KVQ_HI + (size_t)key * KVQ_LO + c0) =
        (unsigned)(q[0] & 255) | ((unsigned)(q[1] & 255) << 8) | ((unsigned)(q[2] & 255) << 16) | ((unsigned)(q[3] & 255) << 24);
}
__device__ __forceinline__ void kvq_store_scale(char* blk, int nk, int key, float scale) {
    *(float*)(blk + (size_t)nk * (KVQ_HI + KVQ_LO) + (size_t)key * 4) = scale;
}

}  // namespace mnx
