// dec_types.h — decoder-side device structures shared by decoder.hip and engine.cpp (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mnx {

constexpr int MAX_ROWS = 32;       // rows per decode call (= the reference's natural batch unit)
constexpr int MAX_DEC_LAYERS = 8;

// Per-call decode state, lives in device memory and is advanced by the step graph itself.
struct DecState {
    int step;                 // index of the step being computed
    int ticket;               // arrival counter of dec_head_kernel workgroups
    int n_alive;
    int alive[MAX_ROWS];
    int prev_tok[MAX_ROWS];
    int len[MAX_ROWS];
    int chunk[MAX_ROWS];      // reference-batch id of each row (rows of one id share a PE numbering)
};

struct DecLayerW {
    const float *ln1_g, *ln1_b;
    const float *wqkv, *bqkv;      // [768,256] rows: query | keys | values   (self_attn.linear_*)
    const float *wo, *bo;          // self_attn.final_linear
    const float *ln2_g, *ln2_b;
    const float *wq2, *bq2;        // context_attn.linear_query
    const float *wo2, *bo2;        // context_attn.final_linear
    const float *lnf_g, *lnf_b;    // feed_forward.layer_norm
    const float *w1, *b1, *w2, *b2;
};

struct DecWeights {
    DecLayerW L[MAX_DEC_LAYERS];
    const float *emb, *pe, *lnF_g, *lnF_b, *wout_t, *bout;
    const float *w_enc, *b_enc;                // enc_trans_layer.0  [256,1024]
    const float *w_memkv, *b_memkv;            // [layers*512, 256]: per layer context keys | values
    const float *edge_w1cat, *edge_b1cat;      // [512,256]: W1[:, :256] | W1[:, 256:], bias 0 | b1
    const float *edge_w2, *edge_b2;            // [7,256], [7]
    int layers, heads, dff, vocab, vpad, sym_offset, bins, pe_len, enc_dim;
};

struct DecBuffers {
    DecState* st;
    float *x, *q, *ctx, *h;                    // [32,256] x3, [32,1024]
    float *self_k, *self_v;                    // [layers, max_batch, heads, T, 32]
    float *memory;                             // [max_batch*S, 256]
    float *mem_kv;                             // [max_batch*S, layers*512]
    float *edge_g, *edge_uv, *edge_prob;       // [B*kmax,256], [B*kmax,512], [B,kmax,kmax,8]
    int T, S, max_batch, kmax;
};

hipError_t dec_enqueue_init(const DecBuffers& b, const int* chunk_dev, int B, hipStream_t s);
hipError_t dec_enqueue_step(const DecWeights& w, const DecBuffers& b, int B, int max_len, int stop_on_eos,
                            int* tokens, float* token_logp, float* hidden, float* logits_trace, hipStream_t s);
hipError_t edges_enqueue(const DecWeights& w, const DecBuffers& bf, const float* hidden, const int* atom_idx,
                         const int* n_atoms, int B, int kmax, int max_len, unsigned char* edges, double* scores,
                         hipStream_t s);

}  // namespace mnx
