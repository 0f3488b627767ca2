#!/usr/bin/env python3
"""CPU emulation of the encoder's operand modes (test infrastructure; a study script, not collected by pytest).

Question: the split modes evaluate a.b as ah.bh + ah.bl + al.bh on the 16-bit MFMA. The two correction
terms are 2^-11 of the main term; could THEY run at the FP8 rate (block-scaled e4m3, v_mfma_scale_f32_16x16x128_f8f6f4:
twice the 16-bit rate for the same operand bytes: hi16 + hi8 + lo8 = 4 B), i.e. 2 instead of 3 units of matrix work?

The script runs the whole Swin-B encoder (the oracle's op decomposition) with every matrix product (qkv, q.k^T, p.v, proj,
fc1, fc2, patch-merging reductions) replaced by an emulation of the scheme — operands rounded exactly as the kernels round
them (hi = T(v), lo = T(v - hi), fp16 weights pre-scaled by a power of two per matrix, softmax probabilities x 2^10 before
the split), products of the rounded operands accumulated in fp32 — and then pushes the features through the oracle's
decoder TEACHER-FORCED along the reference ids of tests/golden/pixels_e2e.npz: feature max / rms error, max log-prob
error, argmax flips over all steps. The fp16x3 / bf16x3 / fp16 / bf16 rows reproduce what the GPU measures (DESIGN.md
section 6.1), which is what makes the fp8 rows believable.

  python tools/study_split_terms.py [--images 32] [--schemes fp16x3,bf16x3,...] [--ckpt 0|stress] [--ranges]
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from molnextr_amd import weights as W                                      # noqa: E402
from oracle import decoder as OD                                           # noqa: E402
from oracle import swin as OS                                              # noqa: E402
from oracle.config import SWIN_B_384, DECODER_DEFAULT                      # noqa: E402

E4M3_MAX = 448.0


def r16(x, dt):
    return x.to(dt).float()


def split(x, dt):
    hi = r16(x, dt)
    return hi, r16(x - hi, dt)


def mx8(x, block=32):
    """OCP MX e4m3: blocks of 32 along the LAST (contraction) axis share a power-of-two scale 2^(floor(log2 amax) - 8)."""
    shp = x.shape
    K = shp[-1]
    pad = (-K) % block
    xp = F.pad(x, (0, pad)) if pad else x
    b = xp.reshape(*xp.shape[:-1], -1, block)
    amax = b.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    s = torch.exp2(torch.floor(torch.log2(amax)) - 8.0)
    q = (b / s).clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).float() * s
    return q.reshape(xp.shape)[..., :K]


class Scheme:
    """mm(a, w): a [..., K] activations, w [N, K] (or [..., N, K]) 'weight-like' operand -> a @ w^T in fp32 after the scheme's
    operand rounding. w_is_weight: per-matrix power-of-two pre-scale (fp16 split only)."""

    def __init__(self, name, two=()):
        self.name = name
        self.amax = {}
        # op classes (tags "qkv", "proj", "fc1", "fc2", "merge", "qk", "pv"; optionally "tag.sN" = only in stage N, 0-based) that
        # drop the activation's lo plane: a.w = ah.wh + ah.wl (two MFMA terms; the weight keeps both planes). Everything else
        # follows the scheme's name. The GPU form of this is compute_dtype FP16X3M (mnx_set_op_terms).
        self.two = set(two)
        self.stage = 0

    def is_two(self, tag):
        return tag in self.two or f"{tag}.s{self.stage}" in self.two

    def note(self, tag, *ts):
        m = max(float(t.abs().max()) for t in ts)
        self.amax[tag] = max(self.amax.get(tag, 0.0), m)

    def mm(self, a, w, tag, w_is_weight=True):
        self.note(tag + ".a", a)
        self.note(tag + ".w", w)
        n = self.name
        if n == "fp32":
            return a @ w.transpose(-1, -2)
        if n in ("fp16", "bf16"):
            dt = torch.float16 if n == "fp16" else torch.bfloat16
            return r16(a, dt) @ r16(w, dt).transpose(-1, -2)
        dt = torch.bfloat16 if n.startswith("bf16") else torch.float16
        scale = 1.0
        if dt == torch.float16 and w_is_weight:
            amax = float(w.abs().max())
            if amax > 0:
                scale = 2.0 ** (14 - math.floor(math.log2(amax)))      # max |2^k w| in [2^14, 2^15)
        ws = w * scale
        ah, al = split(a, dt)
        wh, wl = split(ws, dt)
        main = ah @ wh.transpose(-1, -2)
        if n == "fp16x3+al8" and self.is_two(tag):      # the layers of `two`: third term al.wh on block-scaled e4m3 operands (FP8 rate)
            corr = ah @ wl.transpose(-1, -2) + mx8(al) @ mx8(wh).transpose(-1, -2)
        elif n == "fp16x3+al8":
            corr = ah @ wl.transpose(-1, -2) + al @ wh.transpose(-1, -2)
        elif n in ("fp16x3", "bf16x3") and self.is_two(tag):
            corr = ah @ wl.transpose(-1, -2)
        elif n in ("fp16x3", "bf16x3"):
            corr = ah @ wl.transpose(-1, -2) + al @ wh.transpose(-1, -2)
        elif n == "fp16+fp8x2":          # both correction terms on block-scaled e4m3 operands
            corr = mx8(ah) @ mx8(wl).transpose(-1, -2) + mx8(al) @ mx8(wh).transpose(-1, -2)
        elif n == "fp16+fp8lo":          # only the lo operands in e4m3 (not expressible on the MFMA: both operands share a format
            corr = ah @ mx8(wl).transpose(-1, -2) + mx8(al) @ wh.transpose(-1, -2)      # class; shows which side costs what)
        elif n == "fp16x2":              # one correction term dropped (the activation's lo plane): hi.hi + hi.lo_w
            corr = ah @ wl.transpose(-1, -2)
        else:
            raise ValueError(n)
        return (main + corr) / scale


def window_attention(xw, sd, p, heads, ws, mask, S):
    Bn, N, C = xw.shape
    d = C // heads
    qkv = S.mm(xw, sd[p + ".qkv.weight"], "qkv") + sd[p + ".qkv.bias"]
    qkv = qkv.reshape(Bn, N, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * d ** -0.5, qkv[1], qkv[2]
    attn = S.mm(q, k, "qk", w_is_weight=False)
    table = sd[p + ".relative_position_bias_table"]
    bias = table[OS.relative_position_index(ws).reshape(-1)].reshape(N, N, heads).permute(2, 0, 1)
    attn = attn + bias[None]
    if mask is not None:
        nW = mask.shape[0]
        attn = (attn.reshape(Bn // nW, nW, heads, N, N) + mask[None, :, None]).reshape(Bn, heads, N, N)
    # the kernels keep exp(s - max) un-normalised, scale it by 2^10 before the split and divide by the fp32 sum afterwards
    e = torch.exp(attn - attn.amax(-1, keepdim=True))
    den = e.sum(-1, keepdim=True)
    if S.name == "fp32":
        out = (e / den) @ v
    else:
        out = S.mm(e * 1024.0, v.transpose(-1, -2), "pv", w_is_weight=False) / (1024.0 * den)
    out = out.transpose(1, 2).reshape(Bn, N, C)
    return S.mm(out, sd[p + ".proj.weight"], "proj") + sd[p + ".proj.bias"]


def swin_block(x, H, Wd, sd, p, heads, ws, shift, S):
    B, L, C = x.shape
    xn = OS._ln(x, sd, p + ".norm1").reshape(B, H, Wd, C)
    mask = None
    if shift > 0:
        xn = torch.roll(xn, shifts=(-shift, -shift), dims=(1, 2))
        rid = OS.to_windows(OS.shift_region_ids(H, Wd, ws, shift).reshape(1, H, Wd, 1).float(), ws)[..., 0]
        diff = rid[:, None, :] - rid[:, :, None]
        mask = torch.where(diff != 0, torch.tensor(-100.0), torch.tensor(0.0))
    aw = window_attention(OS.to_windows(xn, ws), sd, p + ".attn", heads, ws, mask, S)
    a = OS.from_windows(aw, ws, B, H, Wd)
    if shift > 0:
        a = torch.roll(a, shifts=(shift, shift), dims=(1, 2))
    x = x + a.reshape(B, L, C)
    h = F.gelu(S.mm(OS._ln(x, sd, p + ".norm2"), sd[p + ".mlp.fc1.weight"], "fc1") + sd[p + ".mlp.fc1.bias"])
    return x + S.mm(h, sd[p + ".mlp.fc2.weight"], "fc2") + sd[p + ".mlp.fc2.bias"]


@torch.no_grad()
def encoder(img, sd, S, cfg=SWIN_B_384):
    x, H, Wd = OS.patch_embed(img.float(), sd, cfg)           # fp32 in every mode (patch_embed_kernel)
    for s, (depth, heads) in enumerate(zip(cfg.depths, cfg.heads)):
        S.stage = s
        for b in range(depth):
            shift = 0 if b % 2 == 0 else cfg.window // 2
            x = swin_block(x, H, Wd, sd, f"transformer.layers.{s}.blocks.{b}", heads, cfg.window, shift, S)
        S.note(f"stream.s{s}", x)
        if s < len(cfg.depths) - 1:
            p = f"transformer.layers.{s}.downsample"
            B, L, C = x.shape
            xx = x.reshape(B, H, Wd, C)
            xx = torch.cat([xx[:, 0::2, 0::2], xx[:, 1::2, 0::2], xx[:, 0::2, 1::2], xx[:, 1::2, 1::2]], -1)
            xx = xx.reshape(B, (H // 2) * (Wd // 2), 4 * C)
            x = S.mm(OS._ln(xx, sd, p + ".norm"), sd[p + ".reduction.weight"], "merge")
            H, Wd = H // 2, Wd // 2
    return OS._ln(x, sd, "transformer.norm")


def round_sig(x, bits, fp16_range=False):
    """x rounded to `bits` significant bits (round to nearest even on the fp32 encoding): what a K / V cache entry stored in
    fewer bytes keeps. bits = 8: bf16; 11: fp16 (fp16_range: per (layer, head) power-of-two scale into [2^14, 2^15), fp16
    subnormals below 2^-24 of that); 16: bf16 hi + 8-bit lo (3 bytes); 19: fp16 hi + 8-bit lo (3 bytes)."""
    if bits >= 24:
        return x
    xi = x.contiguous().view(torch.int32)
    drop = 24 - bits
    half = (1 << (drop - 1)) - 1
    r = (xi + half + ((xi >> drop) & 1)) & ~((1 << drop) - 1)
    y = r.view(torch.float32)
    if fp16_range:          # scaled fp16: magnitudes below 2^-24 x (power-of-two scale of the tensor) lose bits / flush
        amax = float(x.abs().max())
        if amax > 0:
            k = 15 - math.frexp(amax)[1]
            q = math.ldexp(1.0, -24 - k)         # the scaled format's subnormal quantum, in x's units
            small = x.abs() < math.ldexp(1.0, -14 - k)
            y = torch.where(small, torch.round(x / q) * q, y)
    return y


def round_block(x, bits):
    """Block fixed point along the LAST axis (the 32 channels of one key / value row of one head): the row shares a power-of-two
    scale 2^e >= max |x| of the row, every element is an integer of `bits` bits (two's complement, round to nearest) times
    2^(e - bits + 1). What a K / V cache row stored as int16 / int24 + one exponent byte keeps: absolute error <= 2^(e - bits)."""
    amax = x.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    e = torch.floor(torch.log2(amax)) + 1.0                    # 2^e > amax
    q = torch.exp2(e - (bits - 1))
    lim = 2.0 ** (bits - 1)
    return torch.clamp(torch.round(x / q), -lim, lim - 1) * q


@torch.no_grad()
def forced_decode(features, sd, ids, lens, cfg=DECODER_DEFAULT, kv=None):
    """oracle.decoder.greedy_decode's loop, teacher-forced along `ids` (rows stop at `lens`): per (row, step) the masked
    log-prob of the forced id and whether the argmax differs from it. kv: optional rounding applied to every self- and
    memory-K/V entry as it is stored (the K / V cache byte study; per (layer, head) tensors)."""
    P = OD.P
    memory = OD.enc_transform(features, sd)
    B, S_, D = memory.shape
    h, dh, L = cfg.heads, cfg.d_model // cfg.heads, cfg.layers
    mem_kv = OD.cross_kv(memory, sd, cfg)
    if kv is not None and not isinstance(kv, tuple):
        kv = (kv, kv)       # (rounding of keys, rounding of values)
    if kv is not None:      # [B, h, S, dh] per layer: rounded per head
        mem_kv = [tuple(torch.stack([kv[i](t[:, hh]) for hh in range(h)], 1) for i, t in enumerate(pair)) for pair in mem_kv]
    emb_w = sd[P + "embeddings.make_embedding.emb_luts.0.weight"]
    pe = sd[P + "embeddings.make_embedding.pe.pe"].reshape(-1, D)
    T = int(lens.max())
    self_k = torch.zeros(L, B, h, T, dh)
    self_v = torch.zeros(L, B, h, T, dh)
    alive = list(range(B))
    prev = torch.full((B,), cfg.sos_id, dtype=torch.long)
    logp = np.zeros((B, T), np.float32)
    flip = np.zeros((B, T), bool)
    for step in range(T):
        idx = torch.tensor(alive)
        n = len(alive)
        tok_in = prev[idx]
        x = emb_w[tok_in] * math.sqrt(D) + pe[:n]
        for l in range(L):
            lp = f"{P}decoder.transformer_layers.{l}"
            xn = OD._ln(x, sd, lp + ".layer_norm_1")
            k_new = OD._lin(xn, sd, lp + ".self_attn.linear_keys").reshape(n, h, dh)
            v_new = OD._lin(xn, sd, lp + ".self_attn.linear_values").reshape(n, h, dh)
            if kv is not None:
                k_new = torch.stack([kv[0](k_new[:, hh]) for hh in range(h)], 1)
                v_new = torch.stack([kv[1](v_new[:, hh]) for hh in range(h)], 1)
            self_k[l, idx, :, step] = k_new
            self_v[l, idx, :, step] = v_new
            q = OD._lin(xn, sd, lp + ".self_attn.linear_query")
            a = OD._mha(q, self_k[l, idx, :, :step + 1], self_v[l, idx, :, :step + 1], sd, lp + ".self_attn", cfg)
            query = a + x
            qn = OD._ln(query, sd, lp + ".layer_norm_2")
            q2 = OD._lin(qn, sd, lp + ".context_attn.linear_query")
            mid = OD._mha(q2, mem_kv[l][0][idx], mem_kv[l][1][idx], sd, lp + ".context_attn", cfg)
            y = mid + query
            ff = lp + ".feed_forward"
            x = OD._lin(F.gelu(OD._lin(OD._ln(y, sd, ff + ".layer_norm"), sd, ff + ".w_1")), sd, ff + ".w_2") + y
        out = OD._ln(x, sd, P + "decoder.layer_norm")
        lp_ = F.log_softmax(OD._lin(out, sd, P + "output_layer"), dim=-1)
        lp_ = lp_.masked_fill(OD.grammar_mask(tok_in, cfg), OD.MASK_FILL)
        if step == 0:
            lp_[:, cfg.eos_id] = OD.EOS_BAN
        best = lp_.argmax(-1)
        nxt = []
        for r_i, r in enumerate(alive):
            f = int(ids[r, step])
            logp[r, step] = float(lp_[r_i, f])
            flip[r, step] = int(best[r_i]) != f
            prev[r] = f
            if step + 1 < lens[r]:
                nxt.append(r)
        alive = nxt
        if not alive:
            break
    return logp, flip


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=32)
    ap.add_argument("--schemes", default="fp16x3,bf16x3,fp16+fp8x2,fp16+fp8lo,fp16x2,fp16,bf16")
    ap.add_argument("--ckpt", default="0", help="0: synthetic_checkpoint(0) against tests/golden/pixels_e2e; stress: "
                                                "synthetic_checkpoint(1, stress=True) against tests/golden/pixels_stress")
    ap.add_argument("--ranges", action="store_true", help="print max |operand| per op class (fp16 range check)")
    ap.add_argument("--kv", default=None,
                    help="K / V cache byte study instead of the operand schemes: comma list of bf16,fp16,bf16+8,fp16+8,int12b,int16b,int20b,int24b or <keys>/<values> pairs of them (int24b/int16b) — the decoder "
                         "runs teacher-forced on the fp32 features with every cached key / value rounded to that format")
    ap.add_argument("--two", action="append", default=[],
                    help="repeatable: comma list of op classes that run on TWO terms (ah.wh + ah.wl: the activation's lo plane is "
                         "dropped) inside the split schemes, e.g. --two fc1,fc2 --two fc1,fc2,qkv.s2 ; one table row per option "
                         "and split scheme of --schemes (tags qkv proj fc1 fc2 merge qk pv, '.sN' restricts to stage N)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    torch.set_num_threads(int(os.environ.get("STUDY_THREADS", "8")))
    if args.ckpt == "stress":
        ck = W.synthetic_checkpoint(1, stress=True)
        gold = dict(np.load(os.path.join(ROOT, "tests", "golden", "pixels_stress.npz")))
        case, first = "s16", 700
    else:
        ck = W.synthetic_checkpoint(0)
        gold = dict(np.load(os.path.join(ROOT, "tests", "golden", "pixels_e2e.npz")))
        case, first = "m32", 0
    N = min(args.images, gold[f"{case}_ids"].shape[0])
    img = W.synthetic_images(N, first_index=first)
    ids, lens, g_lp, margin = (gold[f"{case}_{k}"][:N] for k in ("ids", "lens", "token_logp", "margin"))
    T = int(lens.max())
    ids, g_lp, margin = ids[:, :T], g_lp[:, :T], margin[:, :T]
    # the reference batch is the PE numbering unit: forcing a prefix of the batch keeps rows' ranks only if the dropped
    # rows are behind them, which holds for a prefix
    t0 = time.time()
    ref = torch.cat([encoder(img[i:i + 4], ck["encoder"], Scheme("fp32")) for i in range(0, N, 4)])
    gs = gold["feat_strided"][:N]
    print(f"fp32 emulation vs golden strided features: max {np.abs(ref[:, ::9, ::16].numpy() - gs).max():.2e} "
          f"(rms {float(ref.pow(2).mean().sqrt()):.3f}) [{time.time() - t0:.0f} s]", flush=True)
    lp_ref, fl_ref = forced_decode(ref, ck["decoder"], ids, lens)
    steps = int(lens.sum())
    m = np.arange(ids.shape[1])[None, :] < lens[:, None]
    print(f"fp32 features through the forced decoder: log-prob vs golden max {np.abs(lp_ref - g_lp)[m].max():.2e}, flips "
          f"{int(fl_ref[m].sum())} / {steps}", flush=True)
    rows = []
    if args.kv:
        fmt = {"bf16": (8, False, 2), "fp16": (11, True, 2), "bf16+8": (16, False, 3), "fp16+8": (19, True, 3), "fp32": (24, False, 4)}
        for b_ in (12, 16, 20, 24):      # block fixed point: int<b> per element + one exponent byte per 32-channel row
            fmt[f"int{b_}b"] = (b_, None, b_ / 8 + 1 / 32)
        def rounding(fname):
            b, rg, _ = fmt[fname]
            return (lambda t: round_block(t, b)) if rg is None else (lambda t: round_sig(t, b, rg))
        for name in args.kv.split(","):
            t0 = time.time()
            if "/" in name:           # "<keys>/<values>": different formats for the two halves of the cache
                kn, vn = name.split("/")
                rnd = (rounding(kn), rounding(vn))
                bits, nbytes = min(fmt[kn][0], fmt[vn][0]), (fmt[kn][2] + fmt[vn][2]) / 2
            else:
                bits, _, nbytes = fmt[name]
                rnd = rounding(name)
            lp, fl = forced_decode(ref, ck["decoder"], ids, lens, kv=rnd)
            fm = [float(margin[b, t]) for b, t in zip(*np.nonzero(fl & m))]
            rec = {"kv_format": name, "bytes_per_element": nbytes, "significant_bits": bits, "images": N,
                   "logp_max_err": float(np.abs(lp - lp_ref)[m].max()), "logp_rms_err": float(np.sqrt((((lp - lp_ref)[m]) ** 2).mean())),
                   "flips": int((fl & m).sum()), "steps": steps, "flip_margins": [round(x, 6) for x in fm[:20]],
                   "seconds": round(time.time() - t0)}
            rows.append(rec)
            print(json.dumps(rec), flush=True)
        if args.out:
            with open(args.out, "w") as fo:
                json.dump({"checkpoint": args.ckpt, "case": case, "study": "K/V cache bytes", "rows": rows}, fo, indent=1)
        return
    runs = [(name, ()) for name in args.schemes.split(",") if name]
    runs += [(name, tuple(t for t in two.split(",") if t)) for two in args.two for name in args.schemes.split(",")
             if name in ("fp16x3", "bf16x3", "fp16x3+al8")]
    for name, two in runs:
        t0 = time.time()
        S = Scheme(name, two)
        f = torch.cat([encoder(img[i:i + 4], ck["encoder"], S) for i in range(0, N, 4)])
        d = (f - ref)
        lp, fl = forced_decode(f, ck["decoder"], ids, lens)
        fm = [float(margin[b, t]) for b, t in zip(*np.nonzero(fl & m))]
        rec = {"scheme": name, "two_terms": ",".join(two), "images": N, "feature_max_err": float(d.abs().max()), "feature_rms_err": float(d.pow(2).mean().sqrt()),
               "logp_max_err": float(np.abs(lp - lp_ref)[m].max()), "flips": int((fl & m).sum()), "steps": steps,
               "flip_margins": [round(x, 6) for x in fm[:20]], "seconds": round(time.time() - t0)}
        if args.ranges:
            rec["amax"] = {k: float(f"{v:.4g}") for k, v in sorted(S.amax.items())}
        rows.append(rec)
        print(json.dumps(rec), flush=True)
    if args.out:
        with open(args.out, "w") as fo:
            json.dump({"checkpoint": args.ckpt, "case": case, "rows": rows}, fo, indent=1)


if __name__ == "__main__":
    main()
