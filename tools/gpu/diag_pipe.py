"""Diagnostic: pipeline (mnx_predict) vs per-batch decode, repeated, with dirtied caches."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from molnextr_amd import weights as W
from molnextr_amd import engine as E_
if os.environ.get("MNX_LIB"):
    E_._LIB_PATH = os.path.join(ROOT, os.environ["MNX_LIB"])
    E_.ABI_VERSION = int(os.environ.get("MNX_LIB_ABI", "3"))
from molnextr_amd.engine import Engine
ck = W.synthetic_checkpoint(0)
eng = Engine(ck["encoder"], ck["decoder"], device=0, max_batch=32)
dev = torch.device("cuda:0")
imgs = W.synthetic_images(80).to(dev)
# dirty the caches: beam search over 256 slots, long greedy decodes on random features
f0 = W.hash_normal("beam_features_32_8", (32, 144, 1024), 0.5).to(dev)
eng.decode_beam(f0, beam=8, n_best=1, max_len=200)
eng.decode_greedy(W.hash_normal("decoder_greedy_features", (6, 144, 1024), 0.5).to(dev))
ref = []
for first in (0, 32, 64):
    f = eng.encode(imgs[first:first + 32].contiguous())
    r = eng.decode_greedy(f)
    ref.append((r["lengths"].cpu().numpy(), r["tokens"].cpu().numpy()))
N_IT = int(os.environ.get('ITERS', '8'))
n_bad = 0
for it in range(N_IT):
    a = eng.predict(imgs, ref_batch=32)
    la, ta = a["lengths"].cpu().numpy(), a["tokens"].cpu().numpy()
    bad = []
    for ci, first in enumerate((0, 32, 64)):
        rl, rt = ref[ci]
        for b in range(len(rl)):
            i = first + b
            if la[i] != rl[b] or not np.array_equal(ta[i, :la[i]], rt[b, :rl[b]]):
                n = min(la[i], rl[b])
                neq = np.nonzero(ta[i, :n] != rt[b, :n])[0]
                bad.append((i, int(la[i]), int(rl[b]), int(neq[0]) if neq.size else n))
    n_bad += bool(bad)
    if bad:
        print("iter", it, "mismatches (image, len_pipe, len_batch, first_diff):", bad)
    if it == 3:   # re-dirty between iterations
        eng.decode_beam(f0, beam=8, n_best=1, max_len=100)
print('iterations with a mismatch:', n_bad, 'of', N_IT)
# per-batch repeat after everything
for ci, first in enumerate((0, 32, 64)):
    f = eng.encode(imgs[first:first + 32].contiguous())
    r = eng.decode_greedy(f)
    print("batch", ci, "repeat equal:", np.array_equal(r["lengths"].cpu().numpy(), ref[ci][0]))
