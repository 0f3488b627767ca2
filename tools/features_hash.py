#!/usr/bin/env python3
"""sha256 of the encoder's output for 64 synthetic 384x384 images (two launch groups of 32) and for a 100x100-pixel-odd tiny
case through the same patch embedding: prints one line. Used by tools/gpu/r05_patch_embed.sh to show that a rewritten kernel is
bit-identical to the library it replaces (run once per library build, compare the lines)."""
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from molnextr_amd import weights as W  # noqa: E402
from molnextr_amd.engine import Engine  # noqa: E402

ck = W.synthetic_checkpoint(0)
eng = Engine(ck["encoder"], ck["decoder"], max_batch=32, dtype=os.environ.get("DTYPE", "fp16x3"))
g = torch.Generator().manual_seed(5)
img = (torch.rand(64, 3, 384, 384, generator=g) * 2 - 1).cuda()
img[:4] = W.synthetic_images(4).cuda()
h = hashlib.sha256()
for i in range(0, 64, 32):
    f = eng.encode(img[i:i + 32].contiguous())
    torch.cuda.synchronize()
    h.update(f.cpu().numpy().tobytes())
print("features sha256", h.hexdigest()[:32])
