#!/usr/bin/env python3
"""sha256 lines of what the library computes for fixed synthetic inputs: the encoder's output for 64 images (two launch groups
of 32); greedy decode of one batch through the fused tick and (MNX_DEC_FUSED_MAX=0) through the eight-launch tick: tokens,
log-probs, hidden states; a 480-step decode without EOS (keys beyond 256 per row); beam search 5 x 8 (ancestry-addressed keys);
the whole predict path on 96 images. Run once per library build on the same box and compare the lines
(tools/gpu/r05_patch_embed.sh): a kernel rewritten for speed with its arithmetic left alone must reproduce every one of them."""
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from molnextr_amd import weights as W  # noqa: E402
from molnextr_amd.engine import Engine  # noqa: E402


def digest(*tensors):
    h = hashlib.sha256()
    for t in tensors:
        h.update(t.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()[:32]


def decoded(d):
    """tokens / log-probs / hidden of the decoded positions only (what lies beyond a row's length is not defined)"""
    n = d["lengths"].cpu().tolist()
    out = [d["lengths"]]
    for i, k in enumerate(n):
        out += [d["tokens"][i, :k], d["token_logp"][i, :k], d["hidden"][i, :k]]
    return out


ck = W.synthetic_checkpoint(0)
dtype = os.environ.get("DTYPE", "fp16x3")
eng = Engine(ck["encoder"], ck["decoder"], max_batch=32, dtype=dtype)
g = torch.Generator().manual_seed(5)
img = (torch.rand(96, 3, 384, 384, generator=g) * 2 - 1).cuda()
img[:4] = W.synthetic_images(4).cuda()
feats = [eng.encode(img[i:i + 32].contiguous()).clone() for i in range(0, 64, 32)]
torch.cuda.synchronize()
print("features sha256", digest(*feats))
print("greedy/fused sha256", digest(*decoded(eng.decode_greedy(feats[0]))))
p = eng.predict(img)
print("predict96 sha256", digest(p["tokens"], p["lengths"], p["n_atoms"], p["edges"]))
b = eng.decode_beam(feats[1][:8].contiguous(), beam=5, n_best=2, max_len=160)
print("beam5x8 sha256", digest(b["tokens"], b["lengths"], b["scores"]))
eng.close()
os.environ["MNX_DEC_FUSED_MAX"] = "0"
eng = Engine(ck["encoder"], ck["decoder"], max_batch=32, dtype=dtype)
print("greedy/eight-launch sha256", digest(*decoded(eng.decode_greedy(feats[0]))))
d = eng.decode_greedy(feats[1][:4].contiguous(), stop_on_eos=False, max_len=480)
print("greedy/480 sha256", digest(d["tokens"], d["token_logp"], d["hidden"]))
