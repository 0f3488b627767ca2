/* molnextr_hip.h — C ABI of libmolnextr_hip.so, the MI355X (gfx950) engine behind the MolNexTR predict path.
 *
 * The reference (CYF2000127/MolNexTR) has no plugin / FFI interface; the seam this library sits behind is the
 * pair of Python calls in `molnextr.predict_images`
 *
 *     features, hiddens = self.encoder(images)                      MolNexTR/model.py:107   (main.py:279)
 *     batch_predictions = self.decoder.decode(features, hiddens)    MolNexTR/model.py:108   (main.py:280)
 *
 * Each entry point below names the reference code it replaces. All functions are `extern "C"`, take plain
 * pointers and sizes (no torch / C++ types), return 0 on success or a negative mnx_status, never throw, and
 * enqueue their GPU work on the caller's HIP stream. Device pointers are raw HBM addresses (e.g. a PyTorch-ROCm
 * tensor's data_ptr()). A handle is bound to one device and is not re-entrant: one in-flight call per handle —
 * except that ONE mnx_preprocess call (its only state is a private 16-byte scratch) may run on another thread and
 * stream beside any other entry point, so that the next images can be uploaded and transformed while mnx_predict
 * works; different handles (GPUs) may be driven from different threads or processes. The only process-wide state is the
 * message of the last failed mnx_create (read it with mnx_last_error(NULL) from the thread that called mnx_create).
 */
#ifndef MOLNEXTR_HIP_H
#define MOLNEXTR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MNX_ABI_VERSION 7

typedef struct mnx_engine mnx_engine;

typedef enum {
    MNX_OK = 0,
    MNX_ERR_INVALID_ARG = -1,   /* bad pointer / size / config                                   */
    MNX_ERR_WEIGHTS = -2,       /* a tensor of the weight contract is missing or has a wrong shape */
    MNX_ERR_HIP = -3,           /* a HIP runtime call failed (see mnx_last_error)                 */
    MNX_ERR_NO_DEVICE = -4,     /* no gfx950 device at the requested index                        */
    MNX_ERR_CAPACITY = -5,      /* batch / length / atom count exceeds what mnx_create reserved   */
    MNX_ERR_RANGE = -6          /* the encoder produced non-finite features (fp16 operand range)  */
} mnx_status;

/* Operand type of the encoder MFMA GEMMs / window attention (accumulation, residual stream, LayerNorm and softmax are
 * fp32 in every mode; the decoder and the bond head are fp32 always).
 *   BF16, FP16     one 16-bit plane per operand: the fastest modes; the operand rounding (2^-9 / 2^-12 relative) reaches
 *                  the logits as 6e-2 / 9e-3, so argmax decisions near a tie can differ from the reference;
 *   FP16X3, BF16X3 split operands: every GEMM / attention operand v is carried as two 16-bit planes hi = T(v),
 *                  lo = T(v - hi), every product evaluated as hi.hi + hi.lo + lo.hi on the 16-bit MFMA with fp32
 *                  accumulation, exact-erf GELU. FP16X3 (weights stored scaled by a power of two per matrix so that their
 *                  lo planes are normal numbers) reproduces the fp32 path to fp32 rounding level: tokens / atoms / bonds
 *                  equal the reference from pixels at a third of the 16-bit MFMA rate. This is the default of the Python
 *                  facade and of bench.py. Activations must stay inside the fp16 range (|v| < 65504): a non-finite
 *                  encoder output is reported as MNX_ERR_RANGE by mnx_predict / mnx_predict_beam and by
 *                  mnx_encoder_status; BF16X3 has the fp32 range and ~2^-16 relative product error;
 *   FP32           every encoder operand in fp32 on the exact-fp32 matrix instructions (1/16 of the bf16 rate): the
 *                  reference-arithmetic mode the split modes are checked against;
 *   FP16X3M        FP16X3 with a per-stage, per-op-class term count ("M" = mixed): the op classes of
 *                  MNX_FP16X3M_TWO_TERM_BY_STAGE evaluate
 *                  a.w as ah.wh + ah.wl — the ACTIVATION's lo plane is neither written by its producer nor read nor
 *                  multiplied; the weight keeps both planes (its rounding is systematic over every token, the
 *                  activation's is noise: dropping ah.wl instead costs 3x the error). Two thirds of the matrix work and half
 *                  the activation bytes in those layers (60 % of the encoder's GEMM time: qkv, fc1, fc2 of Swin stage 3).
 *                  Measured from pixels on both fixture checkpoints: every token / atom / bond still equal to the
 *                  reference's, 0 argmax flips in 12863 teacher-forced steps, log-probs within 1.8e-4, raw logits within
 *                  5.0e-4 (FP16X3: 2e-5 / 8e-5; north_star allows 1e-3); on 384 further images against the oracle the raw
 *                  logits reach 8.7e-4, on 384 more of a hostile checkpoint 1.2e-3 (FP16X3: 2.4e-4), still 0 flips in 220 000
 *                  steps — an OPT-IN throughput mode for callers who accept logits at north_star's edge: the
 *                  default stays FP16X3 (profiles/r06_two_term_tables_gpu.json, r06_extended_parity_*.json, DESIGN.md
 *                  section 4.3, tests/test_gpu_pixels.py).
 *                  Same weights, range and MNX_ERR_RANGE behaviour as FP16X3. mnx_set_op_terms changes the table. */
enum { MNX_DTYPE_BF16 = 0, MNX_DTYPE_FP16 = 1, MNX_DTYPE_FP32 = 2, MNX_DTYPE_BF16X3 = 3, MNX_DTYPE_FP16X3 = 4,
       MNX_DTYPE_FP16X3M = 5 };
/* op classes of the split modes (bits of mnx_set_split_terms / mnx_set_op_terms) */
enum { MNX_OP_QKV = 1, MNX_OP_ATTN = 2, MNX_OP_PROJ = 4, MNX_OP_FC1 = 8, MNX_OP_FC2 = 16, MNX_OP_MERGE = 32 };
/* FP16X3M's table: the two-term op classes of Swin-B's stages 1..4 (the patch-merging reduction BEHIND stage s counts as
 * stage s). tools/study_split_terms.py --two is the CPU emulation that picked it, tests/test_gpu_pixels.py the gate. */
#define MNX_FP16X3M_TWO_TERM_BY_STAGE { 0, 0, MNX_OP_QKV | MNX_OP_FC1 | MNX_OP_FC2, 0 }
#define MNX_FP16X3M_FIRST_BLOCK_BY_STAGE { 0, 0, 0, 0 }     /* first Swin block of the stage that runs the table (all: 0) */

/* Architecture + capacity. Defaults of the reference inference config are in the comments
 * (MolNexTR/models/transformers.py:547-551 swin_base; MolNexTR/model.py:50-81; MolNexTR/utils.py:25). */
typedef struct {
    int32_t img_size;        /* 384; every stage's grid a multiple of `window`, the last stage's grid <= 512 positions */
    int32_t patch;           /* 4   */
    int32_t embed_dim;       /* 128 */
    int32_t n_stages;        /* 4   */
    int32_t depths[4];       /* 2,2,18,2   */
    int32_t heads[4];        /* 4,8,16,32  (head_dim must be 32) */
    int32_t window;          /* 12 */
    int32_t dec_layers;      /* 6   */
    int32_t dec_dim;         /* 256 */
    int32_t dec_heads;       /* 8   */
    int32_t dec_ff;          /* 1024 */
    int32_t vocab;           /* 229 = 101 symbols + 64 x-bins + 64 y-bins */
    int32_t sym_offset;      /* 101: first x-bin id */
    int32_t coord_bins;      /* 64  */
    int32_t pe_len;          /* 5000 */
    int32_t max_len;         /* 480: decode capacity (FORMAT_INFO['chartok_coords']['max_len']) */
    int32_t max_batch;       /* images per mnx_encode call the workspace is sized for */
    int32_t max_atoms;       /* kmax of mnx_edges (<= max_len / 3) */
    int32_t compute_dtype;   /* MNX_DTYPE_FP16X3 (fast AND reference-exact; the host side's default); MNX_DTYPE_BF16 = fastest */
    int32_t dec_slots;       /* sequences resident in the decoder during mnx_predict: multiple of 32, <= 4096; 0 = 2048 */
} mnx_config;

/* One named fp32 tensor of the checkpoint, in HOST memory. Names are the reference state-dict keys
 * ("transformer.layers.2.blocks.7.attn.qkv.weight", "decoder.chartok_coords.output_layer.bias", ...).
 * Index buffers (relative_position_index) and pe.pe are validated/recomputed, pass them or not. */
typedef struct {
    const char* name;
    const float* data;
    int32_t ndim;
    int64_t shape[4];
} mnx_weight_desc;

int mnx_abi_version(void);

/* Replaces `molnextr._get_model` + `loading` (MolNexTR/model.py:17-28,83-95): copies and packs the weights
 * into HBM (engine-owned), validates EVERY tensor of the contract by name and shape (the reference loads with
 * strict=False and ignores mismatches), and reserves all workspace for max_batch images — no allocation
 * happens after create. On failure *out is NULL and the message is available via mnx_last_error(NULL). */
int mnx_create(const mnx_config* cfg, const mnx_weight_desc* weights, int32_t n_weights, int32_t device,
               mnx_engine** out);
void mnx_destroy(mnx_engine* h);

/* NUL-terminated description of the last error on this handle (or of the last failed mnx_create if h==NULL). */
const char* mnx_last_error(const mnx_engine* h);
size_t mnx_workspace_bytes(const mnx_engine* h);

/* Replaces `Encoder.forward` -> `Vision_Transformer.forward` (MolNexTR/components.py:162-174,
 * MolNexTR/models/transformers.py:504-515). images: device fp32 [B,3,S,S] NCHW, normalised.
 * features_out: device fp32 [B, (S/32)^2, 8*embed_dim]. Asynchronous on `stream`. */
int mnx_encode(mnx_engine* h, const float* images, int32_t B, float* features_out, void* stream);

/* Debug/test aid: copy the fp32 residual stream after execution item `item` of the next mnx_encode calls into
 * `dst` (device). Items: 0 = patch_embed, then every Swin block and every patch-merging in execution order.
 * item < 0 disables. */
int mnx_set_encoder_tap(mnx_engine* h, int32_t item, float* dst);

/* Test / measurement aid for the split modes (compute_dtype BF16X3 / FP16X3 / FP16X3M): choose per op class whether its
 * products are evaluated with the mode's own term count (bit set, the default: three, or two for the classes of
 * mnx_set_op_terms) or with the hi.hi term alone, i.e. as the plain 16-bit mode would. Bits: MNX_OP_* — 1 qkv Linear,
 * 2 window attention (QK^T and PV), 4 proj Linear, 8 fc1, 16 fc2, 32 patch-merging reduction.
 * Used by tests/test_gpu_pixels.py to measure which op classes the feature error comes from. No effect in other modes. */
int mnx_set_split_terms(mnx_engine* h, int32_t mask);

/* compute_dtype FP16X3 / FP16X3M only: the Linear op classes (MNX_OP_* bits, not MNX_OP_ATTN) of encoder stage `stage`
 * (0-based; -1 = every stage) that run on TWO terms (ah.wh + ah.wl) in the Swin blocks first_block .. last_block of the stage
 * (0-based, last_block may exceed the depth; the patch-merging reduction counts as the stage's last block). FP16X3 starts
 * with none, FP16X3M with MNX_FP16X3M_TWO_TERM_BY_STAGE from MNX_FP16X3M_FIRST_BLOCK_BY_STAGE on; the weights are the same in both, so one engine can be measured under several tables
 * (tests/test_gpu_pixels.py; tools/study_split_terms.py is the CPU emulation). A 16-bit activation whose only consumer runs on
 * two terms is written as one plane. Takes effect at the next mnx_encode / mnx_predict call; not to be changed while one is
 * in flight. */
int mnx_set_op_terms(mnx_engine* h, int32_t stage, int32_t two_term_mask, int32_t first_block, int32_t last_block);

/* Synchronises `stream` and reports (then clears) whether any mnx_encode since the last call produced a non-finite
 * feature row — the only way the fp16 operand modes can fail on a checkpoint whose activations exceed 65504. */
int mnx_encoder_status(mnx_engine* h, int32_t* nonfinite, void* stream);

/* Replaces `TransformerDecoderAR.decode(beam_size=1)` (MolNexTR/components.py:253-334) including enc_transform
 * (:206-216), Embeddings with the batch-row positional-encoding quirk (MolNexTR/models/embedding.py:52-59),
 * TransformerDecoder stepwise forward (MolNexTR/models/decoder.py:431-486), output layer + log_softmax +
 * CharTokenizer.get_output_mask grammar (MolNexTR/tokenization.py:383-392) and GreedySearch
 * (MolNexTR/decoding/greedy_search.py:96-191).
 *   features   device fp32 [B,144,1024]                          B <= 32 per call
 *   chunk_id   device int32 [B] or NULL: rows with equal ids emulate ONE reference batch — row r gets the
 *              positional encoding of its rank among the still-undecoded rows of its chunk, as the reference
 *              does when it compacts finished rows out of the batch. NULL = all rows are one batch.
 *   max_len    <= cfg.max_len; rows stop at EOS or at max_len tokens (reference max_length)
 *   stop_on_eos 1 = reference behaviour; 0 = fixed-length decode (bench/test aid)
 *   tokens     device int32 [B,max_len]   ids without SOS, including EOS; entries >= lengths[b] are undefined
 *   lengths    device int32 [B]
 *   token_logp device fp32 [B,max_len] or NULL   log-prob of each emitted token (post-mask)
 *   hidden     device fp32 [B,max_len,256] or NULL   post-final-LayerNorm decoder outputs (input of mnx_edges)
 *   logits_trace device fp32 [max_len,B,vocab] or NULL (test aid: raw output_layer logits of every step)
 * Synchronous with respect to its outputs: returns after the last step has completed on `stream`. */
int mnx_decode_greedy(mnx_engine* h, const float* features, int32_t B, const int32_t* chunk_id, int32_t max_len,
                      int32_t stop_on_eos, int32_t* tokens, int32_t* lengths, float* token_logp, float* hidden,
                      float* logits_trace, void* stream);

/* Test aid: mnx_decode_greedy with TEACHER FORCING. Every row advances with forced_ids[b][t] (device int32 [B,max_len],
 * an id sequence that ends with EOS or fills max_len — e.g. the reference's own output) instead of its own argmax, so
 * the history, the finish steps and therefore the positional-encoding rows of the whole batch are the reference's at
 * every step. Outputs: argmax_ids [B,max_len] = what this engine would have chosen at each step GIVEN the reference
 * history; forced_logp [B,max_len] = masked log-prob it assigns to the forced id; lengths = steps taken; logits_trace as
 * mnx_decode_greedy. tests/test_gpu_pixels.py uses it to measure the log-prob error of the 16-bit operand modes along
 * the reference trajectory, free of knock-on effects, and to count argmax flips per token. */
int mnx_decode_forced(mnx_engine* h, const float* features, int32_t B, const int32_t* chunk_id, int32_t max_len,
                      const int32_t* forced_ids, int32_t* argmax_ids, int32_t* lengths, float* forced_logp,
                      float* logits_trace, void* stream);

/* Beam search over one reference batch — the `beam_size > 1` branch of TransformerDecoderAR.decode
 * (MolNexTR/components.py:253-334 with decoding/beam_search.py). The reference's own branch cannot run;
 * the strategy follows BeamSearch.advance/update_finished (beam_search.py:84-190: average log-prob over emitted
 * tokens + 2, flat top-k over beam x vocab, -1e10 for finished beams, stop when the top beam finished and n_best
 * hypotheses exist), the loop re-orders per step and tracks decoder outputs for the bond head (ours; documented in
 * DESIGN.md).
 *   features   device fp32 [B,144,1024], B <= 32 (one reference batch; PE row = row in the alive-images x beam batch)
 *   beam       1..8,  n_best 1..beam
 *   tokens     device int32 [B,n_best,max_len]  hypotheses by descending score; ids without SOS, EOS included
 *   lengths    device int32 [B,n_best]          (0 where fewer than n_best hypotheses finished — cannot happen when
 *                                                n_best <= beam, kept for robustness)
 *   scores     device fp32 [B,n_best]           average log-prob as defined above
 *   hidden     device fp32 [B,n_best,max_len,256] or NULL   decoder outputs along each hypothesis
 * Synchronous with respect to its outputs. */
int mnx_decode_beam(mnx_engine* h, const float* features, int32_t B, int32_t beam, int32_t n_best, int32_t max_len,
                    int32_t* tokens, int32_t* lengths, float* scores, float* hidden, void* stream);

/* Replaces the 'edges' branch of `Decoder.decode` (MolNexTR/components.py:470-491): GraphPredictor.forward
 * (:365-380), softmax over the 7 bond classes and get_edge_prediction (:383-400) incl. its float64 averaging.
 *   hidden   device fp32 [B,max_len,256] as written by mnx_decode_greedy
 *   atom_idx device int32 [B,kmax]: decoder position of each atom (CharTokenizer.sequence_to_smiles 'indices')
 *   n_atoms  device int32 [B]
 *   edges    device uint8 [B,kmax,kmax] bond class 0..6 (rows/cols >= n_atoms[b] undefined)
 *   scores   device fp64 [B,kmax,kmax] or NULL
 * Asynchronous on `stream`. */
int mnx_edges(mnx_engine* h, const float* hidden, const int32_t* atom_idx, const int32_t* n_atoms, int32_t B,
              int32_t kmax, int32_t max_len, uint8_t* edges, double* scores, void* stream);

/* The transform in front of the encoder (MolNexTR/dataset.py:158-185 with augment=False; data_aug.py:98-143,286-301;
 * applied per image at model.py:104): CropWhite(pad) [-> PadToSquare] -> Resize(img_size, bilinear) -> ToGray ->
 * Normalize -> CHW.
 *   rgb      device uint8 [height,width,3] (RGB, as cv2.cvtColor(BGR2RGB) leaves it)
 *   pad      white border added around the ink bounding box (the reference uses 50)
 *   pad_to_square  1 = insert PadToSquare after CropWhite, as `get_transforms` does for test files 'real/acs.csv' and
 *            'real/UOB.csv' (dataset.py:163-164): the shorter side is padded white, diff//2 before, the rest after
 *   crop_out device int32 [4] or NULL: {crop_top, crop_bottom, crop_left, crop_right} exactly as
 *            CropWhite.update_params computes them (data_aug.py:106-136) — pinned by tests/golden/crop_pad.json
 *   out      device fp32 [3,img_size,img_size] — one image of mnx_encode's input
 * Asynchronous on `stream`. Bit-identical to molnextr_amd/preprocess.py. CropWhite / PadToSquare are pinned on the
 * reference's own classes (golden crop boxes, shapes, content hashes); the OpenCV resize / gray arithmetic is restated
 * from its documented behaviour and unpinned (OpenCV is not installable here), see DESIGN.md. */
int mnx_preprocess(mnx_engine* h, const uint8_t* rgb, int32_t height, int32_t width, int32_t pad,
                   int32_t pad_to_square, int32_t* crop_out, float* out, void* stream);

/* Token classes for the on-device atom-position scan used by mnx_predict (the 'indices' that
 * CharTokenizer.sequence_to_smiles derives, MolNexTR/tokenization.py:464-515). flags[id]: bit0 = is_symbol(id),
 * bit1 = is_atom(id) for id < n (= number of vocabulary symbols); the ids of '[' ']' 'C' 'l' 'B' 'r'. */
int mnx_set_token_classes(mnx_engine* h, const uint8_t* flags, int32_t n, int32_t lbracket, int32_t rbracket,
                          int32_t id_C, int32_t id_l, int32_t id_B, int32_t id_r);

/* The atom-position scan on its own (test aid and building block of mnx_predict): tokens device int32 [n,T],
 * lengths device int32 [n] -> atom_idx device int32 [n,kmax], n_atoms device int32 [n]. */
int mnx_atom_scan(mnx_engine* h, const int32_t* tokens, const int32_t* lengths, int32_t n, int32_t T, int32_t kmax,
                  int32_t* atom_idx, int32_t* n_atoms, void* stream);

/* The whole hot path for a list of images, with continuous batching: replaces the body of the chunk loop of
 * `molnextr.predict_images` (MolNexTR/model.py:102-109: encoder + decoder.decode for every chunk) up to, but not
 * including, the host-side detokenisation to symbols / coordinates.
 *   images    device fp32 [n_img,3,S,S]
 *   ref_batch images are decoded as consecutive reference batches of this many rows (<= 32): every batch is one
 *             positional-encoding numbering, exactly as if the reference had been called with this batch_size
 * Up to cfg.dec_slots sequences (dec_slots / 32 reference batches; 2048 by default) are resident on the GPU at once;
 * every decode tick advances all of them by one token, finished batches are retired (atom positions + bond head run on device) and the freed rows are
 * refilled with the next batch while the encoder of the following batch runs on a second stream.
 *   stop_on_eos 1 = reference behaviour; 0 = every sequence runs to max_len (bench aid: deterministic decode work)
 *   tokens    device int32 [n_img,max_len]; lengths device int32 [n_img]
 *   n_atoms   device int32 [n_img]; atom_idx device int32 [n_img,kmax]; edges device uint8 [n_img,kmax,kmax]
 * Synchronous with respect to its outputs. */
int mnx_predict(mnx_engine* h, const float* images, int32_t n_img, int32_t ref_batch, int32_t max_len,
                int32_t stop_on_eos, int32_t* tokens, int32_t* lengths, int32_t* n_atoms, int32_t* atom_idx,
                uint8_t* edges, int32_t kmax, void* stream);

/* mnx_predict with beam search (BASELINE config 5): the same inputs and outputs, every reference batch searched as
 * mnx_decode_beam does (n_best = 1: the best hypothesis; atom positions and the bond head run on ITS tokens and decoder
 * outputs) while the encoder of the following launch groups runs on the second stream. Up to MNX_BEAM_GROUPS (environment,
 * default 8; bounded by 256 images and by dec_slots rows) reference batches of an encoder launch group share ONE step
 * sequence — 8 x 32 x 5 = 1280 rows per step —: images are independent but for the positional-encoding row, which is numbered
 * inside each image's own reference batch, so the hypotheses are exactly those of batch-by-batch searches (2.3x the
 * throughput of one batch at a time). Replaces `decoder.decode(features, hiddens, beam_size=beam)` inside the chunk loop of
 * predict_images (MolNexTR/model.py:102-109, components.py:443) — a branch the reference itself cannot execute (see
 * mnx_decode_beam).
 *   scores    device fp32 [n_img]: average log-prob of the returned hypothesis
 * Synchronous with respect to its outputs. */
int mnx_predict_beam(mnx_engine* h, const float* images, int32_t n_img, int32_t ref_batch, int32_t beam, int32_t max_len,
                     int32_t* tokens, int32_t* lengths, float* scores, int32_t* n_atoms, int32_t* atom_idx,
                     uint8_t* edges, int32_t kmax, void* stream);

/* Kernel-level timing aid for bench.py: runs the 16-bit MFMA GEMM of the encoder on caller buffers.
 * C[M,N] = A[M,K] . W[N,K]^T + bias, A/W 16-bit device, epi: 0 bias->16-bit, 1 bias+GELU->16-bit,
 * 2 bias+residual(fp32, in place in C), 3 bias->fp32. epi | 0x100 (epi 2, 3; test aid) runs the persistent fp32-output
 * kernel (gemm_res.hip) whatever the shape dispatch would choose. */
int mnx_gemm16(mnx_engine* h, int32_t epi, const void* A, const void* W, void* C, const float* bias, int32_t M,
               int32_t N, int32_t K, void* stream);

/* The same for the split modes (the engine's compute_dtype must be BF16X3 / FP16X3): A, W (and C for epi 0 / 1) point at
 * hi planes, a_lo / w_lo / c_lo are the ELEMENT offsets of the lo planes, C = epi(oscale * (Ah.Wh + Ah.Wl + Al.Wh) + bias)
 * with the exact-erf GELU; terms = 3, 2 (Ah.Wh + Ah.Wl: a_lo is ignored; FP16X3 / FP16X3M engines) or 1 (Ah.Wh alone).
 * Test aids in `epi`: | 0x100 the persistent fp32-output kernel of gemm_res.hip, | 0x200 the 128x128 kernel whatever the
 * dispatch would choose, | 0x400 (epi 0 / 1) write the hi output plane only (c_lo ignored). */
int mnx_gemm16_split(mnx_engine* h, int32_t epi, const void* A, int64_t a_lo, const void* W, int64_t w_lo, float oscale,
                     void* C, int64_t c_lo, const float* bias, int32_t M, int32_t N, int32_t K, int32_t terms,
                     void* stream);

/* Measurement aid for bench.py: while enabled, mnx_encode brackets every kernel launch of the sampled calls with a
 * pair of HIP events recorded on the stream the kernel is launched on (also inside mnx_predict, i.e. live in a timed
 * region). `enable` = n > 0: every n-th mnx_encode call since the enable is sampled, at most 4 calls (the event pool
 * stays small and is reused; the launches of the other calls run un-bracketed). mnx_profile_read synchronises and
 * returns, for one kernel class, the totals accumulated since the last reset: summed event-to-event milliseconds,
 * algorithmic work and launch count.
 *   kind 0  MFMA GEMM             work = FLOP (2*M*N*K per launch): stages with C < 512 and the patch-merging reductions
 *   kind 4  MFMA GEMM             the same for the block Linears with C >= 512 (Swin-B stages 3 and 4: the MFMA-bound shapes)
 *   kind 1  LayerNorm             work = HBM bytes (fp32 in, operand-type out [+ fp32 out])
 *   kind 2  window attention      work = HBM bytes (qkv in, context out)
 *   kind 3  patch embedding       work = HBM bytes (image in, fp32 tokens out)
 *   kind < 0  reset the pool */
int mnx_profile_enable(mnx_engine* h, int32_t enable);
int mnx_profile_read(mnx_engine* h, int32_t kind, double* ms, double* work, int64_t* launches);

/* Measurement aid: the two attention launches of a decode layer (mnx::dec_attn_kernel over the self K/V cache, and over
 * the projected memory K/V) launched `iters` times each between HIP events with `rows` sequences resident at position t.
 * Launch i reads the K/V of layer i % dec_layers, exactly as the launches of a real tick do, so that one cycle touches
 * more bytes than the 256 MB Infinity Cache holds and the time per launch is an HBM figure. Isolated (nothing else
 * runs), so the algorithmic HBM bytes per launch are known exactly: rows*8*(t+1)*256 B (self K+V) and rows*8*144*256 B
 * (cross K+V). Overwrites the decoder state: not to be called while a decode call is in flight. */
int mnx_probe_decode_attn(mnx_engine* h, int32_t rows, int32_t t, int32_t iters, double* self_ms, double* cross_ms,
                          void* stream);

/* Measurement aids (ABI 6): the encoder GEMMs are POWER-limited on this chip — the shader clock under the split-operand GEMM
 * kernel is 1.65-1.9 GHz with real operands against 2.4 GHz nominal — so a rate is reported next to the clock it
 * was reached at and next to what the matrix pipes sustain on this device.
 * mnx_gemm_clock: shader clock (MHz; shader cycles / 100 MHz wall ticks of one workgroup per launch) averaged over the persistent
 *   GEMM launches (mnx::gemm256x3_kernel) since the last reset; 0.0 when none ran. Synchronises the device. The counters are
 *   per process (all engines of the device share them).
 * mnx_probe_mfma: runs a register-only loop of v_mfma_f32_16x16x32_f16 on random operands on every CU for about ms_target
 *   milliseconds (after a shorter settling launch) and returns its rate (TFLOP/s) and shader clock: the ceiling of any fp16
 *   MFMA kernel under this device's power budget — no reference-side counterpart (measurement only). */
int mnx_gemm_clock(mnx_engine* h, int32_t reset, double* mhz);
int mnx_probe_mfma(mnx_engine* h, int32_t ms_target, double* tflops, double* mhz, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MOLNEXTR_HIP_H */
