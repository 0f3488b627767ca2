#!/usr/bin/env python3
"""Runs a few Swin-B encodes at B=32 (for rocprofv3 --pmc passes over the encoder kernels)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from molnextr_amd import weights as W  # noqa: E402
from molnextr_amd.engine import Engine  # noqa: E402

ck = W.synthetic_checkpoint(0)
eng = Engine(ck["encoder"], ck["decoder"], max_batch=32)
img = W.synthetic_images(4).cuda().repeat(8, 1, 1, 1).contiguous()
for _ in range(int(os.environ.get("ENCODES", "3"))):
    eng.encode(img)
torch.cuda.synchronize()
print("done")
