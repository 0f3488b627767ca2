/* examples/predict_c_abi.c — the C ABI used from plain C (no Python, no torch): what a non-Python host of the MolNexTR
 * predict path would write. Builds with:   hipcc -Iinclude examples/predict_c_abi.c -Lmolnextr_amd/lib -lmolnextr_hip
 * (or gcc + -lamdhip64). It only shows the call sequence; weights come from the caller's checkpoint reader. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "molnextr_hip.h"

/* hipMalloc / hipFree prototypes kept local so that the example compiles with a bare C compiler too */
extern int hipMalloc(void** p, size_t n);
extern int hipFree(void* p);
extern int hipMemcpy(void* dst, const void* src, size_t n, int kind);

int run(const mnx_weight_desc* weights, int n_weights, const float* host_images /* [n,3,384,384] */, int n_images) {
    mnx_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.img_size = 384; cfg.patch = 4; cfg.embed_dim = 128; cfg.n_stages = 4; cfg.window = 12;
    { const int d[4] = {2, 2, 18, 2}, h[4] = {4, 8, 16, 32}; memcpy(cfg.depths, d, sizeof d); memcpy(cfg.heads, h, sizeof h); }
    cfg.dec_layers = 6; cfg.dec_dim = 256; cfg.dec_heads = 8; cfg.dec_ff = 1024;
    cfg.vocab = 229; cfg.sym_offset = 101; cfg.coord_bins = 64; cfg.pe_len = 5000;
    cfg.max_len = 480; cfg.max_batch = 64; cfg.max_atoms = 160; cfg.compute_dtype = MNX_DTYPE_FP16X3; cfg.dec_slots = 3072;

    mnx_engine* eng = NULL;
    if (mnx_create(&cfg, weights, n_weights, /*device=*/0, &eng) != MNX_OK) {
        fprintf(stderr, "mnx_create: %s\n", mnx_last_error(NULL));
        return 1;
    }
    /* token classes of the vocabulary (CharTokenizer.is_symbol / is_atom) — see molnextr_amd/engine.py for the table */
    /* mnx_set_token_classes(eng, flags, 101, id_lbracket, id_rbracket, id_C, id_l, id_B, id_r); */

    const size_t img_elems = (size_t)3 * 384 * 384;
    float* images = NULL;
    int32_t *tokens = NULL, *lengths = NULL, *n_atoms = NULL, *atom_idx = NULL;
    uint8_t* edges = NULL;
    hipMalloc((void**)&images, n_images * img_elems * sizeof(float));
    hipMalloc((void**)&tokens, (size_t)n_images * 480 * 4);
    hipMalloc((void**)&lengths, (size_t)n_images * 4);
    hipMalloc((void**)&n_atoms, (size_t)n_images * 4);
    hipMalloc((void**)&atom_idx, (size_t)n_images * 160 * 4);
    hipMalloc((void**)&edges, (size_t)n_images * 160 * 160);
    hipMemcpy(images, host_images, n_images * img_elems * sizeof(float), 1 /* hipMemcpyHostToDevice */);

    /* reference batches of 16 images, as `predict_images(batch_size=16)` numbers them (MolNexTR/model.py:97) */
    int rc = mnx_predict(eng, images, n_images, /*ref_batch=*/16, /*max_len=*/480, /*stop_on_eos=*/1, tokens, lengths, n_atoms,
                         atom_idx,
                         edges, /*kmax=*/160, /*stream=*/NULL);
    if (rc != MNX_OK) fprintf(stderr, "mnx_predict: %s\n", mnx_last_error(eng));
    /* ... copy tokens / lengths / atom_idx / edges back and detokenise (tokenization.py:464-515) ... */

    hipFree(images); hipFree(tokens); hipFree(lengths); hipFree(n_atoms); hipFree(atom_idx); hipFree(edges);
    mnx_destroy(eng);
    return rc != MNX_OK;
}
