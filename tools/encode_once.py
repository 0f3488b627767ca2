#!/usr/bin/env python3
"""Runs a few Swin-B encodes at B=$BATCH (default 64, the bench's encoder launch group) for rocprofv3 --pmc passes."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from molnextr_amd import weights as W  # noqa: E402
from molnextr_amd.engine import Engine  # noqa: E402

ck = W.synthetic_checkpoint(0)
B = int(os.environ.get("BATCH", "64"))
eng = Engine(ck["encoder"], ck["decoder"], max_batch=B, dtype=os.environ.get("DTYPE", "fp16x3"))
img = W.synthetic_images(4).cuda().repeat(B // 4, 1, 1, 1).contiguous()
for _ in range(int(os.environ.get("ENCODES", "3"))):
    eng.encode(img)
torch.cuda.synchronize()
print("done")
