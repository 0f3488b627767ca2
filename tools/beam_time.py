#!/usr/bin/env python3
"""Times one reference batch (B=32) through greedy and beam-5 decoding (BASELINE config 5), encoder features given."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from molnextr_amd import weights as W  # noqa: E402
from molnextr_amd.engine import Engine  # noqa: E402

ck = W.synthetic_checkpoint(0)
eng = Engine(ck["encoder"], ck["decoder"], max_batch=32)
feats = eng.encode(W.synthetic_images(32).cuda())
for name, fn in (("greedy", lambda: eng.decode_greedy(feats)),
                 ("beam5", lambda: eng.decode_beam(feats, beam=5, n_best=1)),
                 ("beam5 n_best5", lambda: eng.decode_beam(feats, beam=5, n_best=5))):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    steps = int(out["lengths"].max())
    print(f"{name}: {dt * 1e3:.1f} ms for 32 images ({32 / dt:.0f} molecules/s one batch at a time), "
          f"longest hypothesis {steps} tokens, {dt / steps * 1e6:.0f} us per step")
