"""CPU restatement of the autoregressive decoder + greedy search. Test infrastructure only.

Consumes the reference's decoder state-dict keys (`decoder.chartok_coords.*`).
"""
import math
from dataclasses import dataclass, field
from typing import List, Optional

import torch
import torch.nn.functional as F

from .config import DecoderConfig, DECODER_DEFAULT

P = "decoder.chartok_coords."
LN_EPS = 1e-6            # MolNexTR/models/decoder.py:75,216,293 and onmt PositionwiseFeedForward
MASK_FILL = -10000.0     # MolNexTR/components.py:303 (applied AFTER log_softmax)
EOS_BAN = -1e20          # MolNexTR/decoding/decode_strategy.py:52


def _ln(x, sd, prefix):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], LN_EPS)


def _lin(x, sd, prefix):
    return F.linear(x, sd[prefix + ".weight"], sd[prefix + ".bias"])


def sinusoid_pe(max_len, dim):
    """MolNexTR/models/embedding.py:30-35."""
    pe = torch.zeros(max_len, dim)
    pos = torch.arange(0, max_len).unsqueeze(1).float()
    div = torch.exp(torch.arange(0, dim, 2, dtype=torch.float) * -(math.log(10000.0) / dim))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def enc_transform(features, sd):
    """MolNexTR/components.py:206-216 (enc_pos_emb is off): Linear 1024->256 over [B,144,1024]."""
    B = features.shape[0]
    return _lin(features.reshape(B, -1, features.shape[-1]).float(), sd, P + "enc_trans_layer.0")


def grammar_mask(prev_tok, cfg: DecoderConfig = DECODER_DEFAULT):
    """CharTokenizer.get_output_mask, MolNexTR/tokenization.py:383-392 (+ is_x/is_y :153-159, sep_xy=True).
    prev_tok [n] int64 -> bool [n,V], True = forbidden."""
    n = prev_tok.shape[0]
    ids = torch.arange(cfg.vocab)[None, :].expand(n, -1)
    x0, y0 = cfg.sym_offset, cfg.sym_offset + cfg.bins
    is_x = ((prev_tok >= x0) & (prev_tok < y0))[:, None]
    is_y = (prev_tok >= y0)[:, None]
    return (is_x & (ids < y0)) | (is_y & (ids >= x0))       # after x: only y-bins; after y: no coord bins


def cross_kv(memory, sd, cfg: DecoderConfig = DECODER_DEFAULT):
    """Cross-attention K/V, projected once at step 0 (onmt MultiHeadedAttention 'context' cache,
    driven by MolNexTR/models/decoder.py:269-276,482-486). Returns [L][2] of [B,h,S,dh]."""
    B, S, _ = memory.shape
    h, dh = cfg.heads, cfg.d_model // cfg.heads
    out = []
    for l in range(cfg.layers):
        lp = f"{P}decoder.transformer_layers.{l}.context_attn"
        k = _lin(memory, sd, lp + ".linear_keys").reshape(B, S, h, dh).transpose(1, 2)
        v = _lin(memory, sd, lp + ".linear_values").reshape(B, S, h, dh).transpose(1, 2)
        out.append((k, v))
    return out


def _mha(q, k, v, sd, prefix, cfg):
    """q [n,256] (one query position), k/v [n,h,t,dh]. onmt MultiHeadedAttention: q/sqrt(dh) before QK^T,
    fp32 scores, softmax, P.V, final_linear. No mask at decode time (T=1; all-False source mask)."""
    n = q.shape[0]
    h, dh = cfg.heads, cfg.d_model // cfg.heads
    qh = q.reshape(n, h, 1, dh) / math.sqrt(dh)
    p = (qh @ k.transpose(2, 3)).float().softmax(-1)
    ctx = (p @ v).reshape(n, h * dh)
    return _lin(ctx, sd, prefix + ".final_linear")


@dataclass
class GreedyResult:
    tokens: List[List[int]]            # per row: ids without SOS, including EOS when emitted
    token_logp: List[List[float]]      # log-prob of each emitted token (post-mask)
    hidden: List[torch.Tensor]         # per row [T,256] post-final-LN decoder outputs
    scores: List[float]                # exp(mean(token_logp))  greedy_search.py:172
    finish_step: List[int]
    logits_trace: Optional[list] = None  # when trace=True: per step (alive_rows, logits[n,V])


@torch.no_grad()
def greedy_decode(features, sd, cfg: DecoderConfig = DECODER_DEFAULT, max_len: Optional[int] = None,
                  trace: bool = False, stop_on_eos: bool = True) -> GreedyResult:
    """TransformerDecoderAR.decode with beam_size=1 for ONE reference batch
    (MolNexTR/components.py:253-334 + decoding/greedy_search.py + models/decoder.py:431-486).

    Reproduces the reference's batch-row positional-encoding quirk: `dec_embedding(tgt)` is called without
    `step`, and the sequence-first PositionalEncoding receives a batch-first [n,1,D] tensor, so row r of the
    CURRENT (compacted) batch gets pe[r] at every step (components.py:290, embedding.py:52-59). Finished rows
    are removed from the batch (greedy_search.py:182-190), so a sequence's PE row changes mid-decode.

    stop_on_eos=False is a test/bench aid (fixed-length decode): EOS never finishes a row.
    """
    max_len = cfg.max_len if max_len is None else max_len
    memory = enc_transform(features, sd)                          # components.py:259
    B, S, D = memory.shape
    h, dh, L = cfg.heads, cfg.d_model // cfg.heads, cfg.layers
    mem_kv = cross_kv(memory, sd, cfg)
    emb_w = sd[P + "embeddings.make_embedding.emb_luts.0.weight"]
    pe = sd[P + "embeddings.make_embedding.pe.pe"].reshape(-1, D)
    self_k = torch.zeros(L, B, h, max_len, dh)
    self_v = torch.zeros(L, B, h, max_len, dh)

    alive = list(range(B))                                        # original row ids, in compacted-batch order
    prev = torch.full((B,), cfg.sos_id, dtype=torch.long)
    toks = [[] for _ in range(B)]
    logps = [[] for _ in range(B)]
    hid = torch.zeros(B, max_len, D)
    fin_step = [-1] * B
    logits_trace = [] if trace else None

    for step in range(max_len):                                   # components.py:284
        idx = torch.tensor(alive)
        n = len(alive)
        tok_in = prev[idx]
        x = emb_w[tok_in] * math.sqrt(D) + pe[:n]                 # embedding.py:52,59 — PE by compacted row
        for l in range(L):
            lp = f"{P}decoder.transformer_layers.{l}"
            xn = _ln(x, sd, lp + ".layer_norm_1")                 # decoder.py:260
            self_k[l, idx, :, step] = _lin(xn, sd, lp + ".self_attn.linear_keys").reshape(n, h, dh)
            self_v[l, idx, :, step] = _lin(xn, sd, lp + ".self_attn.linear_values").reshape(n, h, dh)
            q = _lin(xn, sd, lp + ".self_attn.linear_query")
            a = _mha(q, self_k[l, idx, :, :step + 1], self_v[l, idx, :, :step + 1], sd, lp + ".self_attn", cfg)
            query = a + x                                         # decoder.py:266
            qn = _ln(query, sd, lp + ".layer_norm_2")             # decoder.py:268
            q2 = _lin(qn, sd, lp + ".context_attn.linear_query")
            mid = _mha(q2, mem_kv[l][0][idx], mem_kv[l][1][idx], sd, lp + ".context_attn", cfg)
            y = mid + query                                       # decoder.py:277
            ff = lp + ".feed_forward"                             # onmt PositionwiseFeedForward (own LN + residual)
            x = _lin(F.gelu(_lin(_ln(y, sd, ff + ".layer_norm"), sd, ff + ".w_1")), sd, ff + ".w_2") + y
        out = _ln(x, sd, P + "decoder.layer_norm")                # decoder.py:470
        logits = _lin(out, sd, P + "output_layer")                # components.py:296
        if trace:
            logits_trace.append((list(alive), logits.clone()))
        lp_ = F.log_softmax(logits, dim=-1)                       # components.py:298
        lp_ = lp_.masked_fill(grammar_mask(tok_in, cfg), MASK_FILL)  # components.py:300-303
        if step + 1 <= 1:                                         # decode_strategy.py:50-52, min_length=1
            lp_[:, cfg.eos_id] = EOS_BAN
        best_lp, best = lp_.max(dim=-1)                           # greedy_search.py:75-79 (topk(1))
        hid[idx, step] = out
        finished = []
        for r, t, s in zip(alive, best.tolist(), best_lp.tolist()):
            toks[r].append(t)
            logps[r].append(s)
            prev[r] = t
            if (stop_on_eos and t == cfg.eos_id) or step + 1 == max_len:   # greedy_search.py:146, decode_strategy.py:54-56
                finished.append(r)
                fin_step[r] = step
        if finished:
            alive = [r for r in alive if r not in finished]       # greedy_search.py:182-190 (compaction keeps order)
            if not alive:
                break

    hidden = [hid[r, :len(toks[r])].clone() for r in range(B)]
    scores = [float(torch.tensor(lp_r).mean().exp()) for lp_r in logps]
    return GreedyResult(toks, logps, hidden, scores, fin_step, logits_trace)
