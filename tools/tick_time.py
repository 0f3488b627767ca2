#!/usr/bin/env python3
"""Wall time of one greedy decode tick by row count and tick form (run on the GPU box).

    python tools/tick_time.py [rows,rows,...] [forms]        forms: any of  unfused,fused,mid  (default: all three)

For every row count R the engine decodes R synthetic images for exactly T1 and for T2 tokens (EOS ignored, one mnx_predict
job each: encode once, then R alive rows in every tick); (time(T2) - time(T1)) / (T2 - T1) is the tick at R rows and key
positions T1..T2 — encoder, admission and retirement cancel. Forms are selected through the engine's knobs:
unfused = MNX_DEC_TILE=0 (decoder.hip, 8 launches per layer), fused = MNX_DEC_FUSED_MAX=4096 (dec_fused.hip, 3 launches),
mid = MNX_DEC_FUSED_MAX=16 MNX_DEC_MID_MAX=4096 (dec_ma + dec_mb + dec_fb + dec_fc, 4 launches). Prints one table and, when
both were run, whether the fused and the mid form produced the same tokens (they are bit-identical by construction)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
FORMS = {"unfused": {"MNX_DEC_TILE": "0"},
         "fused": {"MNX_DEC_TILE": "-1", "MNX_DEC_FUSED_MAX": "4096", "MNX_DEC_MID_MAX": "0"},
         "mid": {"MNX_DEC_TILE": "-1", "MNX_DEC_FUSED_MAX": "16", "MNX_DEC_MID_MAX": "4096"}}


def main():
    from molnextr_amd import weights as W
    from molnextr_amd.engine import Engine
    rows_list = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "64,128,192,256,384,512,640,1024").split(",")]
    forms = (sys.argv[2] if len(sys.argv) > 2 else "unfused,fused,mid").split(",")
    T1, T2 = 100, 200
    ck = W.synthetic_checkpoint(0)
    dev = torch.device("cuda:0")
    imgs = W.synthetic_images(max(rows_list)).to(dev)
    res, toks = {}, {}
    for form in forms:
        old = {k: os.environ.get(k) for k in FORMS[form]}
        os.environ.update(FORMS[form])
        eng = Engine(ck["encoder"], ck["decoder"], device=0, max_batch=256, dec_slots=max(1024, (max(rows_list) + 31) // 32 * 32))
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
        try:
            for R in rows_list:
                x = imgs[:R].contiguous()
                t = {}
                for T in (T1, T2):
                    best = 1e9
                    for _ in range(3):
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        out = eng.predict(x, ref_batch=32, max_len=T, stop_on_eos=False)
                        torch.cuda.synchronize()
                        best = min(best, time.perf_counter() - t0)
                    t[T] = best
                res[(form, R)] = (t[T2] - t[T1]) / (T2 - T1) * 1e6
                toks[(form, R)] = out["tokens"].cpu()
        finally:
            eng.close()
    print(f"tick wall time, us (key positions {T1}..{T2}, every row alive); rows = rows of capacity")
    # fused and mid evaluate the same chains on the same numbers: their tokens must agree even over 200 free-running steps
    # past EOS; the eight-launch form adds in another order (1e-5 apart) and may leave them at a near-tie of such a run
    both = "fused" in forms and "mid" in forms
    print("rows  " + "".join(f"{f:>10s}" for f in forms) + ("   fused == mid tokens" if both else ""))
    for R in rows_list:
        same = both and torch.equal(toks[("fused", R)], toks[("mid", R)])
        print(f"{R:5d} " + "".join(f"{res[(f, R)]:10.1f}" for f in forms) + (f"   {same}" if both else ""))


if __name__ == "__main__":
    main()
