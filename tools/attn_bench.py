#!/usr/bin/env python3
"""Times the Swin window-attention kernel inside a full encode (rocprof-free): encode time with ablations."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from molnextr_amd import weights as W
from molnextr_amd.engine import Engine
ck = W.synthetic_checkpoint(0)
eng = Engine(ck["encoder"], ck["decoder"], max_batch=32)
img = W.synthetic_images(4).cuda().repeat(8, 1, 1, 1).contiguous()
for _ in range(3):
    eng.encode(img)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(10):
    eng.encode(img)
torch.cuda.synchronize()
print(f"ablate {os.environ.get('MNX_ATTN_ABLATE', '0')}: encode {(time.time() - t0) * 100:.3f} ms")
