// dec_types.h — decoder-side device structures shared by decoder.hip and engine.hip (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mnx {

constexpr int MAX_SLOTS = 4096;    // capacity of the sequence-state arrays (cfg.dec_slots <= this; default 2048 in use)
constexpr int BEGIN_THREADS = 1024;
constexpr int ROW_TILE = 32;       // rows per workgroup of the skinny linears; also the max reference batch
constexpr int MAX_DEC_LAYERS = 8;
constexpr int MAX_CHUNKS = 128;    // reference batches in flight (slots of one batch may be scattered)

// Decode state of every slot; lives in device memory and is advanced by the tick graph itself.
// A "slot" is one sequence being decoded. Slots of one reference batch ("chunk") share a positional-encoding
// numbering: slot s gets pe[rank(s)], rank = number of alive slots of the same chunk with a smaller row index
// (the reference adds a sequence-first PE to a batch-first tensor and compacts finished rows out of the batch).
struct DecState {
    int tick;
    int n_active;                  // alive slots (polled by the host)
    int chunk_alive[MAX_CHUNKS];   // alive slots per chunk tag (polled by the host)
    int alive[MAX_SLOTS];
    int t[MAX_SLOTS];              // next position to decode (= tokens emitted so far)
    int prev_tok[MAX_SLOTS];
    int len[MAX_SLOTS];
    int chunk[MAX_SLOTS];          // chunk tag 0..MAX_CHUNKS-1
    int rowc[MAX_SLOTS];           // row index inside the reference batch
    int rank[MAX_SLOTS];           // PE row for the current tick
    int mem_blk[MAX_SLOTS];        // which 144-row block of mem_kv holds this slot's cross-attention K/V
    int max_len[MAX_SLOTS];
    int stop_on_eos[MAX_SLOTS];
    int active[MAX_SLOTS];         // compact list of the alive slots for the current tick (rows 0..n_active-1)
    // Row view of the tick, written by the begin kernel for EVERY row up to the scanned capacity: {slot, t, prev_tok,
    // PE rank} and the memory block. The tick's kernels read these with the row index alone, in parallel with
    // n_active, instead of chasing n_active -> active[row] -> t[slot] (each hop a memory round trip on the critical
    // path of a 5 us kernel). Rows >= n_active hold a harmless dummy (slot 0, position 0, block 0): kernels may read
    // and compute on them and only have to keep them from WRITING per-slot state.
    int4 rowv[MAX_SLOTS];
    int row_mem[MAX_SLOTS];
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
    // transposed copies [k][n] for the fused tick (dec_fused.hip reads a weight column per lane): [256][768], [256][256] x3,
    // [256][dff], [dff][256]
    const float *wqkv_t, *wo_t, *wq2_t, *wo2_t, *w1_t, *w2_t;
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
    float *x, *q, *ctx, *h;                    // [slots,256] x3, [slots,1024]
    float *x2, *part;                          // the other residual-stream buffer [slots,256]; w_2 K-slice partials [dff/256][slots,256]
    float *fpart;                              // fused tick (dec_fused.hip): two alternating partial buffers [2][16][fpart_rows,256]
    int fpart_rows;                            // rows a plane holds = the largest capacity that runs fused / mid (<= slots)
    char *self_k, *self_v;                     // [layers, slots, heads] blocks of Tq rows, 24-bit block fixed point (kvq.h)
    float *memory;                             // [32*S, 256]   scratch of one admission
    float *mem_kv32;                           // [32, layers, K|V, heads, S, 32] fp32 scratch of one admission (SGEMM output)
    char *mem_kv;                              // [mem_blocks, layers, K|V, heads] blocks of Sq rows (kvq.h)
    int Tq, Sq;                                // rows per block: T, S rounded up to a multiple of 4
    int* tokens;                               // [slots, T]
    float* logp;                               // [slots, T]
    float* hidden;                             // [slots, T, 256]
    float *edge_g, *edge_uv, *edge_prob;       // [32*kmax,256], [32*kmax,512], [32,kmax,kmax,8]
    int T, S, slots, mem_blocks, kmax;
};

// ---- beam search (a12): hypotheses of image i live in slots i*K .. i*K+K-1 ------------------------------
// Up to MAX_BEAM_IMGS images = several reference batches are searched in ONE step sequence (mnx_predict_beam): images are
// independent but for the positional-encoding row, which is numbered inside each image's own reference batch
// (BeamBuffers::ref_batch images).
constexpr int MAX_BEAM_IMGS = 256;
constexpr int MAX_BEAM = 8;
constexpr int BEAM_LP_STRIDE = 256;   // floats per row of the masked log-prob buffer (vocab <= 256)
constexpr int BEAM_ANC_MAX = 512;     // ancestry entries per hypothesis held in LDS (max_len + 1 <= 512)

struct BeamState {
    int top_fin[MAX_BEAM_IMGS];             // the image's top beam has finished at some step
    int n_hyps[MAX_BEAM_IMGS];              // finished hypotheses seen so far (all of them, not only the kept ones)
    int pool_n[MAX_BEAM_IMGS];              // kept hypotheses (<= n_best)
    int order[MAX_BEAM_IMGS][MAX_BEAM];     // rank -> storage index of the kept hypotheses (score descending, stable)
    float pscore[MAX_BEAM_IMGS][MAX_BEAM];  // by storage index
    int plen[MAX_BEAM_IMGS][MAX_BEAM];
    float cum[MAX_BEAM_IMGS * MAX_BEAM];    // cumulative log-prob of every live hypothesis (by slot)
};

struct BeamBuffers {
    BeamState* bs;
    float* blp;      // [B*K, BEAM_LP_STRIDE] masked log-probs of the current step
    int* anc;        // [B*K, anc_stride]: slot that holds step tau of the hypothesis (K/V, hidden at tau; id at tau-1)
    int* ptok;       // [B, pool_stride, T] ids of the kept hypotheses
    float* phid;     // [B, pool_stride, T, 256] decoder outputs of the kept hypotheses, or null
    int B, K, n_best, anc_stride;
    int ref_batch;   // images per reference batch: image i belongs to batch i / ref_batch (the PE numbering unit); B <= 32: B
    int pool_stride; // kept hypotheses stored per image (>= n_best)
};

// token classes for the on-device atom-position scan (CharTokenizer.sequence_to_smiles 'indices')
struct TokenClasses {
    unsigned char flags[256];      // bit0 is_symbol, bit1 is_atom
    int lbracket, rbracket, id_C, id_l, id_B, id_r, x0, y0, vocab;
};

// fp32 projected memory K / V of `n_blocks` (image, layer, K|V, head) blocks of S rows -> quantised blocks at dst (kvq.h)
hipError_t kvq_pack_enqueue(const float* src, char* dst, int n_blocks, int S, int Sq, hipStream_t s);
hipError_t dec_enqueue_admit(const DecBuffers& b, const int* slots_dev, const int* rowc_dev, int n, int chunk_tag,
                             int mem_blk0, int max_len, int stop_on_eos, hipStream_t s);
hipError_t dec_enqueue_reset(const DecBuffers& b, hipStream_t s);
hipError_t dec_enqueue_status(const DecBuffers& b, int slots, hipStream_t s);
// forced: [trace_rows, T] ids or null — teacher forcing of slots 0..trace_rows-1 (test aid, see HeadArgs)
// fused_tile: 0 = the 8-launches-per-layer tick of decoder.hip (always used by beam search); 100 R + RC = the three-launches-
// per-layer tick of dec_fused.hip with R (2, 4) rows per attention workgroup and RC (4, 8, 16) rows per feed-forward workgroup
hipError_t dec_enqueue_tick(const DecWeights& w, const DecBuffers& b, int slots_scan, int rows, float* logits_trace,
                            int trace_rows, hipStream_t s, const BeamBuffers* beam = nullptr, const int* forced = nullptr,
                            int fused_tile = 0);
// dec_fused.hip
hipError_t dec_fused_init();
void dec_fused_dump_stamps(const char* path);   // lab aid (MNX_FUSED_STAMPS)
hipError_t dec_enqueue_fused_layers(const DecWeights& w, const DecBuffers& b, int row_base, int rows, int row_tile, hipStream_t s,
                                    const float** x_final, const float** part_final);
// layers + head of a tick for rows [row_base, row_base + rows) only (after the begin kernel): one BRANCH of a tick
hipError_t dec_enqueue_tick_rows(const DecWeights& w, const DecBuffers& b, int row_base, int rows, float* logits_trace,
                                 int trace_rows, hipStream_t s, const BeamBuffers* beam, const int* forced, int fused_tile);
hipError_t beam_enqueue_init(const DecBuffers& b, const BeamBuffers& bm, int max_len, hipStream_t s);
hipError_t beam_enqueue_gather(const DecBuffers& b, const BeamBuffers& bm, int out_len, int* o_tokens, int* o_len,
                               float* o_scores, float* o_hidden, hipStream_t s);
hipError_t dec_probe_attn(const DecWeights& w, const DecBuffers& b, int rows, int t, int iters, hipEvent_t* ev,
                          hipStream_t s);
hipError_t dec_enqueue_admit_rows(const DecBuffers& b, const int* chunk_ids_dev, int n, int max_len, int stop_on_eos,
                                  hipStream_t s);
hipError_t gather_enqueue(const DecBuffers& b, const int* slots_dev, int n_rows, int out_len, int* o_tokens, int* o_len,
                          float* o_logp, float* o_hidden, hipStream_t s);
hipError_t atoms_enqueue(const DecBuffers& b, const TokenClasses* tc_dev, const int* slots_dev, int n, int kmax,
                         int* atom_idx, int* n_atoms, hipStream_t s);
hipError_t atoms_enqueue_raw(const TokenClasses* tc_dev, const int* tokens, const int* lens, int n, int T, int kmax,
                             int* atom_idx, int* n_atoms, hipStream_t s);
hipError_t edges_enqueue(const DecWeights& w, const DecBuffers& bf, const float* hidden, const int* slot_map,
                         const int* atom_idx, const int* n_atoms, int B, int kmax, int row_stride_T,
                         unsigned char* edges, double* scores, hipStream_t s);

}  // namespace mnx
