"""Weight contract of the MolNexTR predict path + deterministic synthetic checkpoints.

The engine consumes exactly the two state dicts the reference checkpoint holds,
`states['encoder']` and `states['decoder']` (reference MolNexTR/model.py:83-95, main.py:389-398),
with the key names the reference modules register (SURVEY.md §8 a-W):

  encoder  transformer.patch_embed.{proj,norm}.*, transformer.layers.{s}.blocks.{b}.{norm1,attn.qkv,attn.proj,
           attn.relative_position_bias_table,attn.relative_position_index,norm2,mlp.fc1,mlp.fc2}.*,
           transformer.layers.{s}.downsample.{norm,reduction}.*, transformer.norm.*
           (reference MolNexTR/models/transformers.py:123-141,210-218,307-308,402-403,477)
  decoder  decoder.chartok_coords.{enc_trans_layer.0, embeddings.make_embedding.{emb_luts.0,pe.pe},
           decoder.transformer_layers.{l}.{self_attn,context_attn}.{linear_keys,linear_values,linear_query,
           final_linear}, ...feed_forward.{w_1,w_2,layer_norm}, ...layer_norm_1, ...layer_norm_2,
           decoder.layer_norm, output_layer}.*, decoder.edges.mlp.{0,2}.*
           (reference MolNexTR/components.py:183-232,355-358, MolNexTR/models/decoder.py:61-75,213-216,293)

Unlike the reference loader (`strict=False`, silently ignoring mismatches — MolNexTR/model.py:17-28),
`validate_state` checks every expected key and shape and raises on any mismatch.

There is no network and no pretrained checkpoint in this environment, so `synthetic_checkpoint(seed)`
fills that exact key set from a counter-based integer hash (no torch RNG), which makes the weights
bit-identical on every machine — the GPU box regenerates them without the reference.
"""
from __future__ import annotations

import math
import re
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Tuple

import numpy as np
import torch


@dataclass(frozen=True)
class EncoderDims:
    """swin_base: MolNexTR/models/transformers.py:547-551."""
    img_size: int = 384
    patch: int = 4
    embed_dim: int = 128
    depths: Tuple[int, ...] = (2, 2, 18, 2)
    heads: Tuple[int, ...] = (4, 8, 16, 32)
    window: int = 12

    @property
    def num_features(self):
        return self.embed_dim * 2 ** (len(self.depths) - 1)


@dataclass(frozen=True)
class DecoderDims:
    """MolNexTR/model.py:60-76; vocab = 101 symbols + 64 x-bins + 64 y-bins (tokenization.py:172-178)."""
    layers: int = 6
    d_model: int = 256
    heads: int = 8
    d_ff: int = 1024
    vocab: int = 229
    enc_dim: int = 1024
    pe_len: int = 5000
    edge_classes: int = 7


SWIN_B = EncoderDims()
DEC = DecoderDims()


# ----------------------------------------------------------------------------------------------
# expected key -> shape tables
# ----------------------------------------------------------------------------------------------
def encoder_spec(e: EncoderDims = SWIN_B) -> "OrderedDict[str, tuple]":
    s: "OrderedDict[str, tuple]" = OrderedDict()
    C = e.embed_dim
    t = "transformer."
    s[t + "patch_embed.proj.weight"] = (C, 3, e.patch, e.patch)
    s[t + "patch_embed.proj.bias"] = (C,)
    s[t + "patch_embed.norm.weight"] = (C,)
    s[t + "patch_embed.norm.bias"] = (C,)
    n_tab = (2 * e.window - 1) ** 2
    N = e.window * e.window
    for si, (depth, heads) in enumerate(zip(e.depths, e.heads)):
        c = C * 2 ** si
        for b in range(depth):
            p = f"{t}layers.{si}.blocks.{b}."
            s[p + "norm1.weight"] = (c,)
            s[p + "norm1.bias"] = (c,)
            s[p + "attn.relative_position_bias_table"] = (n_tab, heads)
            s[p + "attn.relative_position_index"] = (N, N)
            s[p + "attn.qkv.weight"] = (3 * c, c)
            s[p + "attn.qkv.bias"] = (3 * c,)
            s[p + "attn.proj.weight"] = (c, c)
            s[p + "attn.proj.bias"] = (c,)
            s[p + "norm2.weight"] = (c,)
            s[p + "norm2.bias"] = (c,)
            s[p + "mlp.fc1.weight"] = (4 * c, c)
            s[p + "mlp.fc1.bias"] = (4 * c,)
            s[p + "mlp.fc2.weight"] = (c, 4 * c)
            s[p + "mlp.fc2.bias"] = (c,)
        if si < len(e.depths) - 1:
            p = f"{t}layers.{si}.downsample."
            s[p + "reduction.weight"] = (2 * c, 4 * c)
            s[p + "norm.weight"] = (4 * c,)
            s[p + "norm.bias"] = (4 * c,)
    s[t + "norm.weight"] = (e.num_features,)
    s[t + "norm.bias"] = (e.num_features,)
    return s


def decoder_spec(d: DecoderDims = DEC) -> "OrderedDict[str, tuple]":
    s: "OrderedDict[str, tuple]" = OrderedDict()
    p = "decoder.chartok_coords."
    D = d.d_model
    s[p + "enc_trans_layer.0.weight"] = (D, d.enc_dim)
    s[p + "enc_trans_layer.0.bias"] = (D,)
    s[p + "decoder.layer_norm.weight"] = (D,)
    s[p + "decoder.layer_norm.bias"] = (D,)
    for l in range(d.layers):
        q = f"{p}decoder.transformer_layers.{l}."
        for attn in ("self_attn", "context_attn"):
            for lin in ("linear_keys", "linear_values", "linear_query", "final_linear"):
                s[f"{q}{attn}.{lin}.weight"] = (D, D)
                s[f"{q}{attn}.{lin}.bias"] = (D,)
        s[q + "feed_forward.w_1.weight"] = (d.d_ff, D)
        s[q + "feed_forward.w_1.bias"] = (d.d_ff,)
        s[q + "feed_forward.w_2.weight"] = (D, d.d_ff)
        s[q + "feed_forward.w_2.bias"] = (D,)
        s[q + "feed_forward.layer_norm.weight"] = (D,)
        s[q + "feed_forward.layer_norm.bias"] = (D,)
        s[q + "layer_norm_1.weight"] = (D,)
        s[q + "layer_norm_1.bias"] = (D,)
        s[q + "layer_norm_2.weight"] = (D,)
        s[q + "layer_norm_2.bias"] = (D,)
    s[p + "output_layer.weight"] = (d.vocab, D)
    s[p + "output_layer.bias"] = (d.vocab,)
    s[p + "embeddings.make_embedding.emb_luts.0.weight"] = (d.vocab, D)
    s[p + "embeddings.make_embedding.pe.pe"] = (d.pe_len, 1, D)
    s["decoder.edges.mlp.0.weight"] = (D, 2 * D)
    s["decoder.edges.mlp.0.bias"] = (D,)
    s["decoder.edges.mlp.2.weight"] = (d.edge_classes, D)
    s["decoder.edges.mlp.2.bias"] = (d.edge_classes,)
    return s


def strip_module_prefix(state: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """The reference strips DDP's 'module.' prefix (MolNexTR/model.py:24-26)."""
    return {k.replace("module.", ""): v for k, v in state.items()}


def validate_state(state: Dict[str, torch.Tensor], spec: "OrderedDict[str, tuple]", what: str) -> None:
    """Strict check of a state dict against the contract. Raises ValueError listing every problem."""
    problems = []
    for k, shape in spec.items():
        if k not in state:
            problems.append(f"missing {k} {shape}")
        elif tuple(state[k].shape) != tuple(shape):
            problems.append(f"shape {k}: got {tuple(state[k].shape)}, want {tuple(shape)}")
    extra = [k for k in state if k not in spec]
    if extra:
        problems.append(f"unexpected keys: {extra[:8]}{' ...' if len(extra) > 8 else ''}")
    if problems:
        raise ValueError(f"{what} state dict does not match the MolNexTR weight contract:\n  " + "\n  ".join(problems))


# ----------------------------------------------------------------------------------------------
# deterministic synthetic checkpoint
# ----------------------------------------------------------------------------------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(text: str) -> int:
    h = 0xCBF29CE484222325
    for ch in text.encode("utf-8"):
        h = ((h ^ ch) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def hash_uniform(name: str, n: int, stream: int = 0) -> np.ndarray:
    """n float64 values in [0,1), a pure function of (name, stream, index). Integer arithmetic only."""
    key = np.uint64(_fnv1a64(f"{name}#{stream}"))
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + key
    bits = _splitmix64(ctr) >> np.uint64(11)            # 53 random bits
    return bits.astype(np.float64) * (1.0 / 9007199254740992.0)


def hash_normal(name: str, shape, std: float = 1.0, seed: int = 0) -> torch.Tensor:
    """Bounded near-normal (Irwin-Hall of 4 uniforms, |z| <= 3.46), exactly reproducible across machines."""
    n = int(np.prod(shape)) if len(shape) else 1
    acc = np.zeros(n, dtype=np.float64)
    for k in range(4):
        acc += hash_uniform(name, n, stream=seed * 4 + k)
    z = (acc - 2.0) * math.sqrt(3.0) * std
    return torch.from_numpy(z.astype(np.float32).reshape(shape))


def relative_position_index(window: int) -> torch.Tensor:
    """idx[i,j] = (dy + w-1)*(2w-1) + (dx + w-1) with (dy,dx) = coord(i)-coord(j)
    (reference MolNexTR/models/transformers.py:127-136)."""
    t = torch.arange(window * window)
    y, x = t // window, t % window
    return (y[:, None] - y[None, :] + window - 1) * (2 * window - 1) + (x[:, None] - x[None, :] + window - 1)


def sinusoid_table(n: int, dim: int) -> torch.Tensor:
    """pe[p,2i]=sin(p*w_i), pe[p,2i+1]=cos(p*w_i), w_i=exp(-2i*ln(1e4)/dim) (reference MolNexTR/models/embedding.py:30-35)."""
    pe = torch.zeros(n, dim)
    pos = torch.arange(0, n).unsqueeze(1).float()
    div = torch.exp(torch.arange(0, dim, 2, dtype=torch.float) * -(math.log(10000.0) / dim))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def _synth_tensor(key: str, shape, seed: int, window: int) -> torch.Tensor:
    leaf = key.rsplit(".", 1)[-1]
    if key.endswith("relative_position_index"):
        return relative_position_index(window)
    if key.endswith("pe.pe"):
        return sinusoid_table(shape[0], shape[2]).unsqueeze(1)
    if key.endswith("relative_position_bias_table"):
        return hash_normal(key, shape, 0.5, seed)
    is_norm = ".norm" in key or "layer_norm" in key or key.endswith("patch_embed.norm.weight") \
        or key.endswith("patch_embed.norm.bias")
    if is_norm and "reduction" not in key:
        if leaf == "weight":
            return 1.0 + hash_normal(key, shape, 0.1, seed)
        return hash_normal(key, shape, 0.05, seed)
    if leaf == "bias":
        return hash_normal(key, shape, 0.05, seed)
    # matrices: variance-preserving fan-in scaling so activations stay O(1) through 24+6 layers
    if key.endswith("patch_embed.proj.weight"):
        fan_in = shape[1] * shape[2] * shape[3]
        return hash_normal(key, shape, 1.0 / math.sqrt(fan_in), seed)
    if key.endswith("emb_luts.0.weight"):
        return hash_normal(key, shape, 1.0 / 16.0, seed)        # x sqrt(256) -> unit variance
    fan_in = shape[-1]
    gain = 1.0
    if key.endswith(("attn.proj.weight", "mlp.fc2.weight", "final_linear.weight", "w_2.weight")):
        gain = 0.5                                             # residual-branch outputs
    if key.endswith("output_layer.weight"):
        gain = 2.0                                             # logits with O(1) top-1 margins
    return hash_normal(key, shape, gain / math.sqrt(fan_in), seed)


def _shape_decode_dynamics(d: "OrderedDict[str, torch.Tensor]", dec: DecoderDims) -> None:
    """Give the random decoder molecule-like greedy dynamics (values only — same keys, shapes and arithmetic).

    A purely random decoder never emits EOS and never emits "symbol x y" triples, so every sequence would run
    to max_length with zero atoms: nothing like the reference's real workload, and it would leave row
    compaction (the batch-row positional-encoding quirk) and the bond head untested. Three edits fix that:
      1. embedding dims 0..5 carry a one-hot token class (SOS / letter / x-bin / y-bin / link char / other);
      2. output_layer reads those dims as a class-bigram preference: letter -> x-bin, (grammar: x -> y),
         y -> atom letter | link char | EOS, link -> atom letter;
      3. layer-0 self-attention head 0 attends to the SOS key with weight 50/(50+t), writing an "age"
         feature into residual dim 6 that the EOS logit reads with a negative weight, so EOS becomes
         likely after ~30..330 tokens, at a different step for every row/image.
    Measured on the CPU oracle (seed 0, 16 synthetic images): decoded lengths 33..330 tokens, mean 135; 32 atoms on
    average — the shape of a real molecule workload (the reference caps sequences at 480)."""
    if dec.vocab != 229 or dec.d_model < 64:
        return
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "vocab", "vocab_chars.json")) as f:
        stoi = json.load(f)
    P = "decoder.chartok_coords."
    V, m, p, eos_pref, K, half, age0 = dec.vocab, 0.25, 4.0, 2.8, 6.0, 50.0, 24.0
    letters = [i for s, i in stoi.items() if i >= 5 and s.isalpha()]
    upper = [stoi[c] for c in "CNOSFPBIH"]
    link = [stoi[c] for c in "()=#123"]
    xb, yb = list(range(101, 165)), list(range(165, 229))
    cls = torch.full((V,), 5, dtype=torch.long)
    cls[1] = 0
    cls[letters] = 1
    cls[xb] = 2
    cls[yb] = 3
    cls[link] = 4
    emb = d[P + "embeddings.make_embedding.emb_luts.0.weight"]
    emb[:, :8] = 0
    emb[torch.arange(V), cls] = m
    pref = torch.zeros(6, V)
    pref[0, upper] = 1.0
    pref[1, xb] = 1.0
    pref[3, upper] = 0.8
    pref[3, stoi["C"]] = 1.2
    pref[3, stoi["N"]] = 1.0
    pref[3, stoi["O"]] = 1.0
    pref[3, link] = 0.9
    pref[3, 2] = eos_pref
    pref[4, upper] = 1.0
    W = d[P + "output_layer.weight"]
    W[:, :8] = 0
    W[:, :6] = p * pref.t()
    W[2, 6] = -K
    L0 = P + "decoder.transformer_layers.0.self_attn."
    dh = dec.d_model // dec.heads
    for nm in ("linear_keys", "linear_values", "linear_query"):
        d[L0 + nm + ".weight"][0:dh, :] = 0
        d[L0 + nm + ".bias"][0:dh] = 0
    sos_feat = 16.0 * m / 1.2
    d[L0 + "linear_keys.weight"][0, 0] = 4.0 / sos_feat
    d[L0 + "linear_query.bias"][0] = math.log(half) * math.sqrt(dh) / 4.0
    d[L0 + "linear_values.weight"][0, 0] = age0 / sos_feat
    d[L0 + "final_linear.weight"][:, 0:dh] = 0
    d[L0 + "final_linear.weight"][6, 0] = 1.0


def synthetic_encoder_state(seed: int = 0, enc: EncoderDims = SWIN_B) -> "OrderedDict[str, torch.Tensor]":
    """Only the encoder half of `synthetic_checkpoint` (used with small test configurations)."""
    return OrderedDict((k, _synth_tensor(k, shp, seed, enc.window)) for k, shp in encoder_spec(enc).items())


def _stress_encoder(e: "OrderedDict[str, torch.Tensor]", seed: int) -> None:
    """Hostile-but-sane value statistics for the ENCODER (the part of the path that runs on reduced-precision operands):
    what a trained Swin-B can look like and `synthetic_checkpoint(0)` (all matrices std 1/sqrt(fan_in), LN gains ~1) never
    exercises. In place, keys and shapes untouched:
      * every LayerNorm gain log-uniform in [0.1, 8] per channel (the output norm's rescaled to unit rms); norm2 (the MLP's input norm) of every fifth block gets
        two channels x50 (gains up to 400: operands far from the unit scale the split-operand weight scaling assumes);
      * every Linear weight scaled by its own log-uniform factor: qkv in [0.1, 0.35] (with gains up to 8 in front of it the
        attention scores still have a standard deviation of a few units: saturated softmaxes would make ANY two fp32
        implementations disagree), all others in [0.25, 4] — per-matrix standard deviations from 0.008 to 0.35;
      * relative-position bias tables with std 3, clipped to +-8;
      * three channels of the patch-embedding norm bias at +-50: outlier channels in the residual stream of stage 1
        (LayerNorm statistics dominated by a few channels)."""
    for k in list(e.keys()):
        t = e[k]
        leaf = k.rsplit(".", 1)[-1]
        if k.endswith("relative_position_bias_table"):
            e[k] = (hash_normal(k + "/stress", tuple(t.shape), 3.0, seed)).clamp_(-8.0, 8.0)
        elif (".norm" in k or k.endswith(("norm.weight", "norm.bias"))) and leaf == "weight" and t.dim() == 1:
            u = hash_uniform(k + "/gain", t.numel())
            g = np.exp(np.log(0.1) + u * (np.log(8.0) - np.log(0.1))).astype(np.float32)
            m = re.search(r"layers\.(\d+)\.blocks\.(\d+)\.norm2\.weight$", k)
            if m and int(m.group(2)) % 5 == 1:
                idx = (hash_uniform(k + "/outlier", 2) * t.numel()).astype(np.int64)
                g[idx] *= 50.0
            if k == "transformer.norm.weight":      # the encoder's output norm: same spread of gains, unit-rms features
                g /= np.sqrt(np.mean(g * g))        # (the synthetic decoder's molecule-like dynamics assume that scale)
            e[k] = torch.from_numpy(g)
        elif k.endswith("patch_embed.norm.bias"):
            idx = (hash_uniform(k + "/outlier", 3) * t.numel()).astype(np.int64)
            t[idx[0]] = 50.0
            t[idx[1]] = -50.0
            t[idx[2]] = 50.0
        elif leaf == "weight" and t.dim() == 2:
            u = float(hash_uniform(k + "/scale", 1)[0])
            lo, hi = (0.1, 0.35) if k.endswith("attn.qkv.weight") else (0.25, 4.0)
            t.mul_(float(np.exp(np.log(lo) + u * (np.log(hi) - np.log(lo)))))


def synthetic_checkpoint(seed: int = 0, enc: EncoderDims = SWIN_B, dec: DecoderDims = DEC,
                         molecule_like: bool = True, stress: bool = False) -> dict:
    """A checkpoint dict with the reference's layout: {'encoder': sd, 'decoder': sd, 'args': {...}}
    (reference main.py:389-398). Deterministic in (seed, dims). stress: see _stress_encoder."""
    e = synthetic_encoder_state(seed, enc)
    if stress:
        _stress_encoder(e, seed)
    d = OrderedDict((k, _synth_tensor(k, shp, seed, enc.window)) for k, shp in decoder_spec(dec).items())
    if molecule_like:
        _shape_decode_dynamics(d, dec)
    args = {"formats": ["chartok_coords", "edges"], "input_size": enc.img_size, "coord_bins": 64, "sep_xy": True}
    return {"encoder": e, "decoder": d, "args": args, "synthetic_seed": seed}


def synthetic_images(batch: int, size: int = 384, first_index: int = 0) -> torch.Tensor:
    """Normalised [B,3,S,S] fp32 tensors with the statistics of the reference's pre-processed input
    (white page, dark strokes, grey replicated to 3 channels, ImageNet mean/std: reference
    MolNexTR/dataset.py:176-183). Image i is a pure function of (first_index+i)."""
    mean = np.array([0.485, 0.456, 0.406], dtype=np.float32)
    std = np.array([0.229, 0.224, 0.225], dtype=np.float32)
    out = np.empty((batch, 3, size, size), dtype=np.float32)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    for i in range(batch):
        u = hash_uniform(f"image{1000 + first_index + i}", 6 * 40 + 1)
        n_seg = 8 + int(u[0] * 33)
        g = np.full((size, size), 255.0, dtype=np.float32)
        for s in range(n_seg):
            x0, y0, x1, y1, w, shade = u[1 + 6 * s: 7 + 6 * s]
            x0, y0, x1, y1 = (v * (size - 1) for v in (x0, y0, x1, y1))
            dx, dy = x1 - x0, y1 - y0
            L2 = dx * dx + dy * dy + 1e-6
            t = np.clip(((xx - x0) * dx + (yy - y0) * dy) / L2, 0.0, 1.0)
            dist = np.hypot(xx - (x0 + t * dx), yy - (y0 + t * dy))
            g = np.where(dist <= 1.0 + w, np.float32(40.0 * shade), g)
        for c in range(3):
            out[i, c] = (g / 255.0 - mean[c]) / std[c]
    return torch.from_numpy(out)


PAGE_CASES = [(470, 923, "strokes"), (64, 64, "strokes"), (1, 1, "blank"), (1, 1, "ink"), (37, 911, "strokes"),
              (600, 800, "strokes"), (384, 384, "border"), (300, 17, "strokes"), (50, 70, "blank"), (40, 40, "corners"),
              (90, 120, "noise"), (33, 200, "row"), (200, 33, "col"), (120, 120, "offwhite"), (211, 97, "channel")]


def synthetic_page(case: int) -> np.ndarray:
    """Ragged HWC uint8 RGB test pages (white paper with ink), a pure function of the case index: inputs of the
    pre-processing fixtures (CropWhite / PadToSquare, reference MolNexTR/data_aug.py:98-150,286-301)."""
    h, w, kind = PAGE_CASES[case]
    img = np.full((h, w, 3), 255, np.uint8)
    u = hash_uniform(f"page{case}", 6 * 12 + h * w * 3 if kind == "noise" else 6 * 12)
    if kind == "strokes":
        for s in range(12):
            y, x, hh, ww, r, g = u[6 * s:6 * s + 6]
            y, x = int(y * h), int(x * w)
            hh, ww = 1 + int(hh * max(1, h // 3)), 1 + int(ww * max(1, w // 3))
            img[y:y + hh, x:x + ww] = (int(r * 256), int(g * 256), int((r + g) * 128) % 256)
    elif kind == "ink":
        img[0, 0] = 0
    elif kind == "border":          # ink on all four page borders
        img[0, 5:9] = 0; img[-1, 100] = 10; img[7, 0] = 20; img[200, -1] = 30
    elif kind == "corners":
        img[0, 0] = 0; img[-1, -1] = 254
    elif kind == "noise":
        img = (u[72:].reshape(h, w, 3) * 256).astype(np.uint8)
    elif kind == "row":
        img[h // 2, 3:w - 7] = 0
    elif kind == "col":
        img[5:h - 2, w // 2] = 0
    elif kind == "offwhite":        # a pixel that differs from white in ONE channel only counts as ink
        img[30, 40] = (255, 255, 254); img[90, 70, 0] = 254
    elif kind == "channel":
        img[20:25, 10:60, 1] = 0
    return img
