#!/usr/bin/env python3
"""Micro-benchmark of the encoder MFMA GEMM (mnx_gemm16) on the Swin-B shapes at B=$BATCH (32). MI355X only."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from molnextr_amd import weights as W  # noqa: E402
from molnextr_amd.engine import Engine  # noqa: E402

TINY = W.EncoderDims(img_size=96, patch=4, embed_dim=32, depths=(2, 2), heads=(1, 2), window=12)
dec = W.DecoderDims(enc_dim=TINY.num_features)
ck = W.synthetic_checkpoint(0, enc=TINY, dec=dec)
eng = Engine(ck["encoder"], ck["decoder"], max_batch=2, enc=TINY, dec=dec)
dev = torch.device("cuda:0")
SC = int(os.environ.get("BATCH", "32")) // 32          # rows scale with the encoder launch group (BATCH=64: x2)
shapes = [("qkv s2", 0, 18432, 1536, 512), ("proj s2", 2, 18432, 512, 512), ("fc1 s2", 1, 18432, 2048, 512),
          ("fc2 s2", 2, 18432, 512, 2048), ("qkv s0", 0, 294912, 384, 128), ("fc1 s0", 1, 294912, 512, 128),
          ("fc2 s0", 2, 294912, 128, 512), ("qkv s1", 0, 73728, 768, 256), ("fc1 s1", 1, 73728, 1024, 256),
          ("fc2 s1", 2, 73728, 256, 1024), ("qkv s3", 0, 4608, 3072, 1024), ("fc1 s3", 1, 4608, 4096, 1024),
          ("fc2 s3", 2, 4608, 1024, 4096), ("big", 3, 8192, 8192, 8192)]
only = sys.argv[1:]
iters = int(os.environ.get("ITERS", "20"))
for name, epi, M, N, K in shapes:
    M = M * SC if name != "big" else M
    if only and name.split()[0] not in only and name not in only:
        continue
    A = torch.randn(M, K, device=dev).bfloat16()
    Wt = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    bias = torch.randn(N, device=dev)
    out = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi >= 2 else torch.bfloat16)
    for _ in range(3):
        eng.gemm16(epi, A, Wt, out, bias)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        eng.gemm16(epi, A, Wt, out, bias)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"{name:8s} epi{epi} M{M} N{N} K{K}: {us:8.1f} us  {2 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)
if os.environ.get("CALIB"):
    # known-byte-count calibration for FETCH_SIZE / WRITE_SIZE: a 512 MiB fp32 copy (read 512 MiB, write 512 MiB)
    x = torch.randn(128 * 1024 * 1024, device=dev)
    y = torch.empty_like(x)
    for _ in range(3):
        y.copy_(x)
    torch.cuda.synchronize()
    print("calibration copy done: 536870912 bytes read, 536870912 bytes written per launch")
