"""CPU restatement of the Swin encoder the reference executes. Test infrastructure only.

Functional: consumes the reference's encoder state-dict keys directly
(`transformer.patch_embed.*`, `transformer.layers.{s}.blocks.{b}.*`, ...).
"""
import math

import torch
import torch.nn.functional as F

from .config import SwinConfig, SWIN_B_384

LN_EPS = 1e-5  # nn.LayerNorm default, MolNexTR/models/transformers.py:201,427


def _ln(x, sd, prefix):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], LN_EPS)


def _lin(x, sd, prefix, bias=True):
    return F.linear(x, sd[prefix + ".weight"], sd[prefix + ".bias"] if bias else None)


def relative_position_index(ws):
    """MolNexTR/models/transformers.py:127-136: idx[i,j] = (dy+ws-1)*(2ws-1) + (dx+ws-1), d = coord(i)-coord(j)."""
    t = torch.arange(ws * ws)
    y, x = t // ws, t % ws
    dy = y[:, None] - y[None, :] + ws - 1
    dx = x[:, None] - x[None, :] + ws - 1
    return dy * (2 * ws - 1) + dx


def shift_region_ids(H, W, ws, shift):
    """Region id of every token of the SHIFTED map; MolNexTR/models/transformers.py:223-234.
    Slices [0,-ws), [-ws,-shift), [-shift,end) on both axes -> ids 0..8."""
    def axis(n):
        r = torch.zeros(n, dtype=torch.long)
        r[n - ws:n - shift] = 1
        r[n - shift:] = 2
        return r
    return axis(H)[:, None] * 3 + axis(W)[None, :]


def to_windows(x, ws):
    """[B,H,W,C] -> [B*nW, ws*ws, C]; MolNexTR/models/transformers.py:68-80 (window id row-major, token y*ws+x)."""
    B, H, W, C = x.shape
    x = x.reshape(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(B * (H // ws) * (W // ws), ws * ws, C)


def from_windows(w, ws, B, H, W):
    """Inverse of to_windows; MolNexTR/models/transformers.py:83-97."""
    C = w.shape[-1]
    x = w.reshape(B, H // ws, W // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(B, H, W, C)


def window_attention(xw, sd, p, heads, ws, mask):
    """MolNexTR/models/transformers.py:147-178. xw [B*nW,N,C]; mask [nW,N,N] or None."""
    Bn, N, C = xw.shape
    d = C // heads
    qkv = _lin(xw, sd, p + ".qkv").reshape(Bn, N, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * d ** -0.5, qkv[1], qkv[2]          # scale q BEFORE QK^T (:157)
    attn = q @ k.transpose(-2, -1)                        # [Bn,h,N,N]
    table = sd[p + ".relative_position_bias_table"]       # [(2ws-1)^2, h]
    idx = sd.get(p + ".relative_position_index")
    if idx is None:
        idx = relative_position_index(ws)
    bias = table[idx.reshape(-1)].reshape(N, N, heads).permute(2, 0, 1)
    attn = attn + bias[None]
    if mask is not None:
        nW = mask.shape[0]
        attn = (attn.reshape(Bn // nW, nW, heads, N, N) + mask[None, :, None]).reshape(Bn, heads, N, N)
    attn = attn.softmax(-1)
    out = (attn @ v).transpose(1, 2).reshape(Bn, N, C)
    return _lin(out, sd, p + ".proj")


def swin_block(x, H, W, sd, p, heads, ws, shift):
    """MolNexTR/models/transformers.py:245-292 (eval mode: drop_path/dropout are identity)."""
    B, L, C = x.shape
    assert L == H * W and H % ws == 0 and W % ws == 0, "oracle restates the no-padding case (384^2 and test sizes)"
    xn = _ln(x, sd, p + ".norm1").reshape(B, H, W, C)
    mask = None
    if shift > 0:
        xn = torch.roll(xn, shifts=(-shift, -shift), dims=(1, 2))                       # :261-262
        rid = to_windows(shift_region_ids(H, W, ws, shift).reshape(1, H, W, 1).float(), ws)[..., 0]  # [nW,N]
        diff = rid[:, None, :] - rid[:, :, None]
        mask = torch.where(diff != 0, torch.tensor(-100.0), torch.tensor(0.0))          # :238-239
    aw = window_attention(to_windows(xn, ws), sd, p + ".attn", heads, ws, mask)
    a = from_windows(aw, ws, B, H, W)
    if shift > 0:
        a = torch.roll(a, shifts=(shift, shift), dims=(1, 2))                           # :279-280
    x = x + a.reshape(B, L, C)                                                          # :289
    h = F.gelu(_lin(_ln(x, sd, p + ".norm2"), sd, p + ".mlp.fc1"))                      # timm Mlp: fc1->GELU(erf)->fc2
    return x + _lin(h, sd, p + ".mlp.fc2")                                              # :290


def patch_merging(x, H, W, sd, p):
    """MolNexTR/models/transformers.py:310-336: concat (0,0),(1,0),(0,1),(1,1) -> LN(4C) -> Linear(4C,2C,no bias)."""
    B, L, C = x.shape
    x = x.reshape(B, H, W, C)
    x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
    x = x.reshape(B, (H // 2) * (W // 2), 4 * C)
    return _lin(_ln(x, sd, p + ".norm"), sd, p + ".reduction", bias=False), H // 2, W // 2


def patch_embed(img, sd, cfg):
    """MolNexTR/models/transformers.py:405-419: conv k=s=patch -> [B, (H/p)*(W/p), C] -> LN."""
    x = F.conv2d(img, sd["transformer.patch_embed.proj.weight"], sd["transformer.patch_embed.proj.bias"],
                 stride=cfg.patch)
    B, C, H, W = x.shape
    x = x.flatten(2).transpose(1, 2)
    return _ln(x, sd, "transformer.patch_embed.norm"), H, W


@torch.no_grad()
def encoder_forward(img, sd, cfg: SwinConfig = SWIN_B_384, return_hiddens=False, tap=None):
    """Encoder.forward (MolNexTR/components.py:162-174) -> Vision_Transformer.forward (transformers.py:504-515).

    img [B,3,S,S] fp32 normalised NCHW. Returns features [B, (S/32)^2, num_features].
    `tap(name, tensor)` is an optional callback used by tests to capture intermediates.
    """
    x, H, W = patch_embed(img.float(), sd, cfg)
    if tap:
        tap("patch_embed", x)
    hiddens = []
    for s, (depth, heads) in enumerate(zip(cfg.depths, cfg.heads)):
        for b in range(depth):
            shift = 0 if b % 2 == 0 else cfg.window // 2          # transformers.py:363 (not clamped at the 12x12 stage)
            x = swin_block(x, H, W, sd, f"transformer.layers.{s}.blocks.{b}", heads, cfg.window, shift)
            if tap:
                tap(f"s{s}b{b}", x)
        hiddens.append(x)
        if s < len(cfg.depths) - 1:
            x, H, W = patch_merging(x, H, W, sd, f"transformer.layers.{s}.downsample")
            if tap:
                tap(f"merge{s}", x)
    x = _ln(x, sd, "transformer.norm")                              # transformers.py:512
    return (x, hiddens) if return_hiddens else x
