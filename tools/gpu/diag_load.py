"""Diagnostic: is one batch's decode bitwise reproducible while another stream keeps the GPU busy?"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from molnextr_amd import weights as W
from molnextr_amd.engine import Engine
ck = W.synthetic_checkpoint(0)
eng = Engine(ck["encoder"], ck["decoder"], device=0, max_batch=32, dec_slots=64)
dev = torch.device("cuda:0")
imgs = W.synthetic_images(32).to(dev)
f = eng.encode(imgs)
torch.cuda.synchronize()
ref = eng.decode_greedy(f, max_len=96, stop_on_eos=False, trace_logits=True)
side = torch.cuda.Stream()
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
mode = os.environ.get("LOAD", "gemm")
bad = 0
N = int(os.environ.get("ITERS", "12"))
for it in range(N):
    with torch.cuda.stream(side):
        if mode == "gemm":
            for _ in range(30):
                c = a @ b
        elif mode == "encode":
            for _ in range(3):
                eng.encode(imgs)
        elif mode == "elementwise":
            for _ in range(200):
                c = a * 1.0001
        elif mode == "softmax":
            for _ in range(60):
                c = torch.softmax(a.float(), dim=-1)
        elif mode == "layernorm":
            xx = a.float()
            for _ in range(60):
                c = torch.nn.functional.layer_norm(xx, (8192,))
        elif mode.startswith("g:"):   # our GEMM on one shape: g:epi:M:N:K
            _, epi, M_, N_, K_ = mode.split(":")
            epi, M_, N_, K_ = int(epi), int(M_), int(N_), int(K_)
            if "ga" not in globals():
                globals()["ga"] = torch.randn(M_, K_, device=dev).bfloat16()
                globals()["gw"] = (torch.randn(N_, K_, device=dev) / K_ ** 0.5).bfloat16()
                globals()["gb"] = torch.randn(N_, device=dev)
                globals()["go"] = torch.zeros(M_, N_, device=dev, dtype=torch.float32 if epi >= 2 else torch.bfloat16)
            for _ in range(40):
                eng.gemm16(epi, ga, gw, go, gb)
        elif mode == "gemm16":      # our own GEMM kernel only
            o = torch.zeros(8192, 8192, device=dev, dtype=torch.bfloat16)
            for _ in range(20):
                eng.gemm16(0, a, b, o, None)
    r = eng.decode_greedy(f, max_len=96, stop_on_eos=False, trace_logits=True)
    torch.cuda.synchronize()
    same_t = torch.equal(r["tokens"], ref["tokens"])
    same_h = torch.equal(r["hidden"], ref["hidden"])
    if not (same_t and same_h):
        bad += 1
        d = (r["hidden"] != ref["hidden"]).nonzero()
        first = d[:, 1].min().item() if d.numel() else -1
        rows = sorted(set(d[d[:, 1] == first][:, 0].tolist()))[:8] if d.numel() else []
        md = (r["hidden"][:, first] - ref["hidden"][:, first]).abs().max().item() if d.numel() else 0
        print(f"iter {it}: tokens equal {same_t}, hidden equal {same_h}; first differing step {first}, rows {rows}, max diff there {md:.3e}")
print(f"LOAD={mode}: non-reproducible iterations: {bad} of {N}")
