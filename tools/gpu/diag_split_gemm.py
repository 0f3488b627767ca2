"""GPU diagnostic: where does the split-operand GEMM differ from a float64 product? Prints, per epilogue, the largest
errors with the hi / lo planes at those elements. Usage: python tools/gpu/diag_split_gemm.py [M N K] [dtype]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from molnextr_amd import weights as W
from molnextr_amd.engine import Engine

M, N, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (1000, 384, 128)
dtype = sys.argv[4] if len(sys.argv) > 4 else "fp16x3"
TINY = W.EncoderDims(img_size=96, patch=4, embed_dim=32, depths=(2, 2), heads=(1, 2), window=12)
dec = W.DecoderDims(enc_dim=TINY.num_features)
ck = W.synthetic_checkpoint(0, enc=TINY, dec=dec)
e = Engine(ck["encoder"], ck["decoder"], max_batch=2, enc=TINY, dec=dec, dtype=dtype)
dev = torch.device("cuda:0")
td = torch.float16 if dtype == "fp16x3" else torch.bfloat16
g = torch.Generator().manual_seed(M + N + K)
A = torch.randn(M, K, generator=g).to(dev)
Wt = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
bias = torch.randn(N, generator=g).to(dev)
ref = A.double() @ Wt.double().t() + bias.double()
ws = 2.0 ** 12 if dtype == "fp16x3" else 1.0


def planes(x, s=1.0):
    hi = (x * s).to(td)
    return torch.stack([hi, (x * s - hi.float()).to(td)]).contiguous()


A2, W2 = planes(A), planes(Wt, ws)
for epi, want in ((0, ref), (1, torch.nn.functional.gelu(ref))):
    o2 = torch.zeros(2, M, N, device=dev, dtype=td)
    e.gemm16_split(epi, A2, W2, o2, bias, oscale=1.0 / ws)
    torch.cuda.synchronize()
    got = o2[0].double() + o2[1].double()
    err = (got - want).abs()
    print(f"epi {epi}: max err {err.max().item():.3e}, elements with err > 1e-4: {(err > 1e-4).sum().item()} of {err.numel()}, "
          f"lo plane nonzero: {(o2[1] != 0).float().mean().item():.3f}")
    idx = torch.topk(err.flatten(), 8).indices
    for i in idx.tolist():
        m, n = divmod(i, N)
        x = ref[m, n].item()
        print(f"   [{m},{n}] pre-activation {x:+.6f} want {want[m, n].item():+.8f} hi {o2[0, m, n].item():+.8f} lo {o2[1, m, n].item():+.3e} "
              f"err {err[m, n].item():.3e} gelu32(x) {torch.nn.functional.gelu(torch.tensor(x, dtype=torch.float32)).item():+.8f}")
    bad = (err > 1e-4)
    if bad.any():
        rows = bad.any(1).nonzero().flatten()
        cols = bad.any(0).nonzero().flatten()
        print("   bad rows:", rows[:20].tolist(), "... count", rows.numel(), " bad cols:", cols[:20].tolist(), "... count", cols.numel())
e.close()
