#!/usr/bin/env python3
"""GPU bring-up diagnostics: prints per-stage errors of the HIP path against golden fixtures / the oracle.
Run on the MI355X box:  python tools/gpu_diag.py [gemm] [tiny] [full] [dec] [edges] [time]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from molnextr_amd import weights as W  # noqa: E402
from molnextr_amd.engine import Engine  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
TINY = W.EncoderDims(img_size=96, patch=4, embed_dim=32, depths=(2, 2), heads=(1, 2), window=12)
what = set(sys.argv[1:]) or {"gemm", "tiny", "full", "dec", "edges", "time"}
dev = torch.device("cuda:0")


def report(name, got, want, tol=None):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    err = np.abs(got - want)
    rms = np.sqrt((want ** 2).mean()) + 1e-30
    print(f"  {name:34s} max|err| {err.max():.3e}  rms_err/rms {np.sqrt((err ** 2).mean()) / rms:.3e}  "
          f"nan {int(np.isnan(got).sum())}" + (f"  {'OK' if err.max() <= tol else 'FAIL'}" if tol else ""), flush=True)


def tiny_engine(dtype="bf16"):
    dec = W.DecoderDims(enc_dim=TINY.num_features)
    ck = W.synthetic_checkpoint(0, enc=TINY, dec=dec)
    return Engine(ck["encoder"], ck["decoder"], max_batch=2, enc=TINY, dec=dec, dtype=dtype), ck


if "gemm" in what:
    print("== gemm16 vs torch", flush=True)
    eng, _ = tiny_engine()
    g = torch.Generator(device="cpu").manual_seed(0)
    for (M, N, K) in [(128, 128, 64), (256, 384, 128), (300, 96, 32), (4608, 1024, 4096), (129, 132, 72)]:
        A = torch.randn(M, K, generator=g).to(dev).bfloat16()
        Wt = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).bfloat16()
        bias = torch.randn(N, generator=g).to(dev)
        ref = A.float() @ Wt.float().t() + bias
        out = torch.zeros(M, N, device=dev, dtype=torch.float32)
        eng.gemm16(3, A, Wt, out, bias)
        torch.cuda.synchronize()
        report(f"f32 out M{M} N{N} K{K}", out.cpu().numpy(), ref.cpu().numpy())
        o16 = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
        eng.gemm16(1, A, Wt, o16, bias)
        report(f"gelu bf16 out M{M} N{N} K{K}", o16.float().cpu().numpy(),
               torch.nn.functional.gelu(ref).cpu().numpy())
        res = torch.randn(M, N, generator=g).to(dev)
        r2 = res.clone()
        eng.gemm16(2, A, Wt, r2, bias)
        report(f"resid M{M} N{N} K{K}", r2.cpu().numpy(), (ref + res).cpu().numpy())
    eng.close()

if "tiny" in what:
    print("== tiny Swin (96px, C32, 2+2 blocks) every item vs reference golden", flush=True)
    gold = np.load(os.path.join(GOLD, "swin_tiny.npz"))
    img = W.hash_normal("swin_tiny_img", (2, 3, 96, 96), 1.0).to(dev)
    for dtype in ("bf16", "fp16"):
        eng, _ = tiny_engine(dtype)
        names = ["patch_embed", "s0b0", "s0b1", "merge0", "s1b0", "s1b1"]
        print(f" dtype {dtype}")
        for i, n in enumerate(names):
            dst = torch.zeros(gold[n].shape, device=dev)
            eng.set_tap(i, dst)
            f = eng.encode(img)
            torch.cuda.synchronize()
            report(n, dst.cpu().numpy(), gold[n])
        eng.set_tap(-1, None)
        report("features", eng.encode(img).cpu().numpy(), gold["features"])
        eng.close()

if what & {"full", "dec", "edges", "time"}:
    t0 = time.time()
    ck = W.synthetic_checkpoint(0)
    print(f"synthetic checkpoint built in {time.time() - t0:.1f}s", flush=True)
    t0 = time.time()
    eng = Engine(ck["encoder"], ck["decoder"], max_batch=32)
    print(f"engine created in {time.time() - t0:.1f}s, workspace {eng.workspace_bytes / 2**30:.2f} GiB", flush=True)

if "full" in what:
    print("== full Swin-B @384 vs reference golden slices", flush=True)
    gold = np.load(os.path.join(GOLD, "swin_full.npz"))
    img = W.synthetic_images(2).to(dev)
    f = eng.encode(img).cpu().numpy()
    report("features_head", f[:, :4, :], gold["features_head"])
    report("features_strided", f[:, ::9, ::16], gold["features_strided"])
    print("  abs-sum rel err", np.abs(np.abs(f).sum(axis=(1, 2)) / gold["features_abs_sum"] - 1))

if "dec" in what:
    print("== greedy decode vs reference golden (B=6, max_len 480)", flush=True)
    gold = np.load(os.path.join(GOLD, "decoder_greedy.npz"))
    feats = W.hash_normal("decoder_greedy_features", (6, 144, 1024), 0.5).to(dev)
    t0 = time.time()
    r = eng.decode_greedy(feats, trace_logits=True)
    torch.cuda.synchronize()
    print(f"  decode wall {time.time() - t0:.3f}s")
    lens = r["lengths"].cpu().numpy()
    print("  lens", lens.tolist(), "gold", gold["lens"].tolist())
    toks = r["tokens"].cpu().numpy()
    for b in range(6):
        n = int(gold["lens"][b])
        m = min(n, int(lens[b]))
        same = toks[b, :m] == gold["ids"][b, :m]
        first = int(np.argmin(same)) if not same.all() else -1
        print(f"  row {b}: len {lens[b]} vs {n}; first token mismatch at {first}")
    lg = r["logits"].cpu().numpy()
    for s in range(4):
        report(f"logits step {s}", lg[s], gold[f"logits_step{s}"])
    hid = r["hidden"].cpu().numpy()
    report("hidden[:, :8]", hid[:, :8], gold["hidden_head"])
    lp = r["token_logp"].cpu().numpy()
    for b in range(6):
        n = min(int(gold["lens"][b]), int(lens[b]))
        report(f"token_logp row {b}", lp[b, :n], gold["token_logp"][b, :n])
    gold = np.load(os.path.join(GOLD, "decoder_short.npz"))
    feats = W.hash_normal("decoder_short_features", (3, 144, 1024), 0.5).to(dev)
    r = eng.decode_greedy(feats, max_len=24)
    print("  short lens", r["lengths"].cpu().tolist(), "tokens equal",
          bool((r["tokens"].cpu().numpy() == gold["ids"]).all()))

if "edges" in what:
    print("== bond head vs reference golden", flush=True)
    gold = np.load(os.path.join(GOLD, "edges.npz"))
    for name, T in (("a", 40), ("b", 90), ("c", 12), ("d", 20)):
        hidden = torch.zeros(1, 480, 256)
        hidden[0, :T] = W.hash_normal(f"edges_hidden_{name}", (T, 256), 1.0)
        idx = gold[f"{name}_idx"]
        k = len(idx)
        ai = torch.zeros(1, 160, dtype=torch.int32)
        ai[0, :k] = torch.from_numpy(idx)
        e, s = eng.edges(hidden.to(dev), ai.to(dev), torch.tensor([k], dtype=torch.int32), want_scores=True)
        e = e.cpu().numpy()[0, :k, :k]
        s = s.cpu().numpy()[0, :k, :k]
        print(f"  case {name}: k={k} edges equal {bool((e == gold[f'{name}_edges']).all())} "
              f"mismatches {int((e != gold[f'{name}_edges']).sum())}  score max err {np.abs(s - gold[f'{name}_scores']).max():.2e}")

if "time" in what:
    print("== timing (B=32)", flush=True)
    img = W.synthetic_images(4).to(dev).repeat(8, 1, 1, 1).contiguous()
    for _ in range(2):
        f = eng.encode(img)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(5):
        f = eng.encode(img)
    torch.cuda.synchronize()
    print(f"  encode B=32: {(time.time() - t0) / 5 * 1e3:.2f} ms")
    for ml, stop in ((128, False), (480, True)):
        r = eng.decode_greedy(f, max_len=ml, stop_on_eos=stop)
        torch.cuda.synchronize()
        t0 = time.time()
        r = eng.decode_greedy(f, max_len=ml, stop_on_eos=stop)
        torch.cuda.synchronize()
        dt = time.time() - t0
        L = r["lengths"].cpu().numpy()
        print(f"  decode B=32 max_len={ml} stop={stop}: {dt * 1e3:.2f} ms  lens mean {L.mean():.1f} max {L.max()}  "
              f"({dt / max(L.max(), 1) * 1e6:.1f} us/step)")
print("diag done", flush=True)
