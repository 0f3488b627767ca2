#!/usr/bin/env python3
"""Per-phase wall-clock view of the LAST fused decode tick (dec_fused.hip built-in stamps, 100 MHz s_memrealtime).

  MNX_FUSED_STAMPS=/tmp/st.bin python tools/fused_stamps.py run ROWS STEPS [tile] [tile_ff]   (on the GPU box)
      decodes ROWS synthetic images for exactly STEPS tokens (EOS ignored) through mnx_predict, so that the last tick
      has ROWS alive rows at position STEPS-1; the engine dumps the stamps when it is closed
  python tools/fused_stamps.py show /tmp/st.bin
      per stage (A = self-attention block, B = context attention, C = feed-forward; 3 per layer): first start -> last end,
      and the median over workgroups of every phase's duration
"""
import os
import sys

import numpy as np

BLOCKS, PHASES = 512, 12
NAMES = {"A": ["stream / partial / weight-column / key requests -> partials summed, LN (1)", "value-row requests (2)",
               "barrier: xs complete (3)", "qkv chains (weights arrive), Wo request (4)", "barrier: chains stored (5)",
               "combine q/k/v, cache append, barrier (6)", "scores (K arrives), max, barrier (7)", "exp, sum, barrier (8)",
               "P.V (V arrives), barrier (9)", "ctx, barrier (10)", "Wo partial chain + stores (11)"],
         "B": ["stream / partial / weight-column / memory-row requests -> partials summed, LN (1)", "value-row requests (2)",
               "barrier: xs complete (3)", "q chains (weights arrive), Wo2 request (4)", "barrier: chains stored (5)",
               "combine q, barrier (6)", "scores (K arrives), max, barrier (7)", "exp, sum, barrier (8)",
               "P.V (V arrives), barrier (9)", "ctx, barrier (10)", "Wo2 partial chain + stores (11)"],
         "C": ["stream / partial / W1 / W2 column requests -> summed, LN (1)", None, "barrier: xs complete (3)",
               "w_1 chains (weights arrive) (4)", "barrier: chains stored (5)", None, None, None, None, "GELU, barrier (10)",
               "w_2 partial chain + stores (11)"]}


def show(path):
    st = np.fromfile(path, dtype=np.uint64).reshape(-1, BLOCKS, PHASES).astype(np.int64)
    t0 = None
    for k in range(st.shape[0]):
        blk = st[k][st[k][:, 0] > 0]
        if len(blk) == 0:
            continue
        kind = "ABC"[k % 3]
        if t0 is None:
            t0 = blk[:, 0].min()
        start, end = blk[:, 0].min(), blk[:, 11].max()
        print(f"stage {k:2d} ({kind}, layer {k // 3}): {len(blk)} workgroups, first start {0.01 * (start - t0):8.2f} us, "
              f"last end {0.01 * (end - t0):8.2f} us, span {0.01 * (end - start):6.2f} us, start skew {0.01 * (blk[:, 0].max() - start):5.2f} us, "
              f"median workgroup {0.01 * np.median(blk[:, 11] - blk[:, 0]):6.2f} us")
        if k < 3 or k >= st.shape[0] - 3 or k in (6, 7, 8):
            prev = 0
            for ph in range(1, PHASES):
                if NAMES[kind][ph - 1] is None:
                    continue
                d = blk[:, ph] - blk[:, prev]
                print(f"      {NAMES[kind][ph - 1]:84s} median {0.01 * np.median(d):6.2f}  max {0.01 * d.max():6.2f} us")
                prev = ph


def run(rows, steps, tile, tile_ff):
    import torch
    from molnextr_amd import weights as W
    from molnextr_amd.engine import Engine
    os.environ["MNX_DEC_TILE"] = str(tile)
    os.environ["MNX_DEC_TILE_FF"] = str(tile_ff)
    os.environ["MNX_DEC_FUSED_MAX"] = "4096"
    ck = W.synthetic_checkpoint(0)
    eng = Engine(ck["encoder"], ck["decoder"], device=0, max_batch=32, dec_slots=max(64, (rows + 31) // 32 * 32))
    imgs = W.synthetic_images(rows).to("cuda:0")
    for _ in range(2):
        t = torch.cuda.Event(enable_timing=True)
        u = torch.cuda.Event(enable_timing=True)
        t.record()
        eng.predict(imgs, ref_batch=32, max_len=steps, stop_on_eos=False)
        u.record()
        torch.cuda.synchronize()
        print(f"predict({rows} images, {steps} forced steps): {t.elapsed_time(u):.1f} ms")
    eng.close()


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if sys.argv[1] == "show":
        show(sys.argv[2])
    else:
        run(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 4,
            int(sys.argv[5]) if len(sys.argv) > 5 else 4)
