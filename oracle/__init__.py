"""oracle/ — CPU restatement of the MolNexTR predict hot path. TEST INFRASTRUCTURE ONLY.

This package is the checker, never the product. Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
it. `molnextr_amd/` must never import it: the product path fails loudly when
the HIP library is missing, it does not fall back to this code.

What it restates (all `file:line` relative to the reference repo
CYF2000127/MolNexTR, mounted read-only at /root/reference in the build
container only):

  swin.py     Encoder.forward -> Vision_Transformer (Swin-B)
              MolNexTR/components.py:162-174, MolNexTR/models/transformers.py:68-515
  decoder.py  enc_transform, Embeddings (+ the batch-row positional-encoding
              quirk), TransformerDecoder stepwise forward, output layer,
              log_softmax, grammar mask, GreedySearch with row compaction
              MolNexTR/components.py:206-334, MolNexTR/models/decoder.py:224-486,
              MolNexTR/models/embedding.py:30-61, MolNexTR/tokenization.py:383-392,
              MolNexTR/decoding/{decode_strategy,greedy_search}.py
  edges.py    GraphPredictor + softmax + get_edge_prediction
              MolNexTR/components.py:350-400,470-491

Arithmetic is fp32 on CPU (torch CPU ops are used as the BLAS), exact-erf GELU,
LayerNorm eps 1e-5 (Swin) / 1e-6 (decoder), exactly as the reference computes.

Pinning status (see DESIGN.md "Oracle"):
  * Every function here is checked against the reference's own files imported
    in the build container (tools/gen_golden.py) and against the fixtures that
    script committed under tests/golden/.
  * The arithmetic of OpenNMT-py 2.2.0 (MultiHeadedAttention,
    PositionwiseFeedForward, Elementwise) and timm 0.4.12 (Mlp) is NOT in the
    reference tree and not installable here; it is restated from the published
    algorithm and anchored on the reference's call sites and state-dict key
    names only: PARITY UNPINNED for those pieces.
  * The reference ships no tests; its single known-answer vector
    (examples/1.png in prediction.ipynb) needs molnextr_best.pth, which is not
    available offline.
"""
from .config import SwinConfig, DecoderConfig, SWIN_B_384, DECODER_DEFAULT  # noqa: F401
