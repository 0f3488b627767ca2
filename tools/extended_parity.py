#!/usr/bin/env python3
"""From-pixels parity of the HIP path on MORE images than the committed fixtures hold: test infrastructure, GPU box only.

The fixtures (tests/golden/pixels_e2e.*, pixels_stress.*) are outputs of the reference's own classes on 38 + 16 images; the
gates of tests/test_gpu_pixels.py are statements about those 12 863 steps. This tool asks the same questions of other images of
the same generators, against the CPU ORACLE (oracle/, bit-equal to the reference on every fixture): per reference batch of 32
images the oracle encodes and decodes greedily (with its raw logits), the engine encodes the same pixels in the chosen operand
mode, decodes free-running (tokens must be the oracle's) and teacher-forced along the oracle's ids (log-prob of every emitted
token, argmax flips, raw logits of EVERY step). It writes one JSON record; nothing here is part of the product path.

    python tools/extended_parity.py [--ckpt 0|stress] [--batches 8] [--first 1000] [--dtype fp16x3m | --two TAGS] [--out file.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from molnextr_amd import weights as W  # noqa: E402
from molnextr_amd.engine import DEFAULT_DTYPE, Engine  # noqa: E402
from oracle.decoder import greedy_decode  # noqa: E402
from oracle.swin import encoder_forward  # noqa: E402


def check(eng, ck, first, batches, rows=32, label=None, verbose=True):
    """`batches` reference batches of `rows` synthetic images starting at image index `first`: the record described above."""
    dev = torch.device("cuda", eng.device)
    rec = {"first_image": first, "batches": [], "images": 0, "steps": 0, "flips": 0,
           "rows_exact_free_running": 0, "logit_max_err_all_steps": 0.0, "logit_max_err_steps0_3": 0.0, "logp_max_err": 0.0,
           "feature_max_err": 0.0, "flip_margins": []}
    if label:
        rec.update(label)
    fsq, fn = 0.0, 0
    t0 = time.time()
    for b in range(batches):
        img = W.synthetic_images(rows, first_index=first + rows * b)
        ref_f = torch.cat([encoder_forward(img[i:i + 8], ck["encoder"]) for i in range(0, rows, 8)])
        ref = greedy_decode(ref_f, ck["decoder"], trace=True)
        lens = np.array([len(t) for t in ref.tokens], np.int32)
        T = 480
        ids = np.zeros((rows, T), np.int32)
        for r, t in enumerate(ref.tokens):
            ids[r, :len(t)] = t
        feats = eng.encode(img.to(dev))
        d = (feats.cpu() - ref_f)
        ferr = float(d.abs().max())
        fsq += float((d.double() ** 2).sum()); fn += d.numel()
        free = eng.decode_greedy(feats, max_len=T)
        fl = free["lengths"].cpu().numpy()
        ft = free["tokens"].cpu().numpy()
        rows_exact = sum(int(fl[r] == lens[r] and ft[r, :lens[r]].tolist() == ref.tokens[r]) for r in range(rows))
        out = eng.decode_forced(feats, torch.from_numpy(ids), max_len=T, trace_logits=True)
        assert np.array_equal(out["lengths"].cpu().numpy(), lens)
        am = out["argmax"].cpu().numpy()
        lp = out["forced_logp"].cpu().numpy()
        lg = out["logits"].cpu().numpy()                      # [T, B, V]; row b of step t defined while t < lens[b]
        lerr_all, lerr_03, perr, flips, margins = 0.0, 0.0, 0.0, 0, []
        for step, (alive, ref_lg) in enumerate(ref.logits_trace):
            e = np.abs(lg[step, alive] - ref_lg.numpy()).max()
            lerr_all = max(lerr_all, float(e))
            if step < 4:
                lerr_03 = max(lerr_03, float(e))
        for r in range(rows):
            n = int(lens[r])
            perr = max(perr, float(np.abs(lp[r, :n] - np.array(ref.token_logp[r], np.float32)).max()))
            bad = np.nonzero(am[r, :n] != ids[r, :n])[0]
            flips += len(bad)
            for t in bad:                                      # the oracle's own top-1 / top-2 margin at a flipped step
                alive, ref_lg = ref.logits_trace[int(t)]
                top = torch.log_softmax(ref_lg[alive.index(r)], -1).topk(2).values
                margins.append(float(top[0] - top[1]))
        brec = {"first_image": first + rows * b, "steps": int(lens.sum()), "len_max": int(lens.max()), "feature_max_err": ferr,
                "logit_max_err_all_steps": lerr_all, "logit_max_err_steps0_3": lerr_03, "logp_max_err": perr, "flips": flips,
                "rows_exact_free_running": rows_exact}
        rec["batches"].append(brec)
        rec["images"] += rows
        rec["steps"] += brec["steps"]
        rec["flips"] += flips
        rec["rows_exact_free_running"] += rows_exact
        rec["flip_margins"] += [round(m, 6) for m in margins]
        for k in ("logit_max_err_all_steps", "logit_max_err_steps0_3", "logp_max_err", "feature_max_err"):
            rec[k] = max(rec[k], brec[k])
        if verbose:
            print(json.dumps(brec), f"[{time.time() - t0:.0f} s]", flush=True)
    rec["feature_rms_err"] = (fsq / max(fn, 1)) ** 0.5
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ckpt", default="0", choices=["0", "stress"])
    ap.add_argument("--batches", type=int, default=8)
    ap.add_argument("--first", type=int, default=1000)
    ap.add_argument("--dtype", default=DEFAULT_DTYPE)
    ap.add_argument("--two", default=None, help="fp16x3 engine with THIS two-term table (Engine.set_op_terms tags, e.g. qkv.s2,fc1.s2)")
    ap.add_argument("--threads", type=int, default=32)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    ck = W.synthetic_checkpoint(1, stress=True) if a.ckpt == "stress" else W.synthetic_checkpoint(0)
    eng = Engine(ck["encoder"], ck["decoder"], device=0, max_batch=32, dtype="fp16x3" if a.two is not None else a.dtype, dec_slots=64)
    if a.two is not None:
        eng.set_op_terms(tuple(t for t in a.two.split(",") if t))
    label = {"checkpoint": a.ckpt, "dtype": a.dtype if a.two is None else "fp16x3 + two-term table " + a.two}
    rec = check(eng, ck, a.first, a.batches, label=label)
    eng.close()
    print("EXTENDED_PARITY", json.dumps({k: v for k, v in rec.items() if k != "batches"}), flush=True)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(rec, f, indent=1)


if __name__ == "__main__":
    main()
