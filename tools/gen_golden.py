#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE itself.

Container-only: imports the reference's own hot-path files from /root/reference (read-only) through
tools/ref_import.py and runs them on deterministic inputs (hash-generated weights and tensors from
molnextr_amd.weights — no torch RNG, so the GPU box regenerates the same inputs without the reference).
Only inputs-by-recipe and expected OUTPUTS are stored; no reference source travels.

    python tools/gen_golden.py            # rewrites tests/golden/*.npz, *.json

Fixtures (all small):
  swin_tiny.npz        tiny Vision_Transformer (img 96, C 32, depths 2/2, heads 1/2, window 12): every block output
  swin_full.npz        swin_base @384 on 2 synthetic images under synthetic_checkpoint(0): feature slices + moments
  decoder_greedy.npz   TransformerDecoderAR.decode (greedy, max_length 480) on hash features, B=6: ids, log-probs,
                       finish steps, hidden slices, first-steps logits   (covers the batch-row PE quirk + compaction)
  decoder_short.npz    same with max_length 24 on B=3 (max-length finish path)
  embedding_pe.npz     Embeddings.forward on [B,1,1] ids: the row-indexed positional encoding
  edges.npz            GraphPredictor + softmax + get_edge_prediction on hash hidden states
  tokenizer.json       get_output_mask truth table (229 ids) + sequence_to_smiles cases
  predict_e2e.json     Decoder.decode end-to-end on B=4: smiles / symbols / coords / indices / edges
  predict_e2e_conf.json  same with compute_confidence=True on B=3: atom / edge / overall scores
  beam_strategy.npz    BeamSearch.advance/update_finished driven with scripted log-probs: back-pointers, ids, scores,
                       surviving images per step, final n-best predictions
  pixels_autocast_fp16.npz/.json  the same path under the reference's evaluation precision (fp16 autocast, main.py:277):
                 how far the reference moves ITSELF from its fp32 result (the yardstick for the one-plane operand modes)
  pixels_e2e.npz/.json the PATH AS A UNIT (model.py:107-108): Encoder.forward + Decoder.decode of the reference on
                       W.synthetic_images(32) under synthetic_checkpoint(0) — ids, lengths, token log-probs, first-steps
                       logits, top-1/top-2 margins of every step, atom positions, bond matrices — for one batch of 6
                       and one of 32, plus the plain-random decoder (molecule_like=False) cut at 64 steps
  crop_pad.json        CropWhite.update_params/apply + PadToSquare.apply of the reference's data_aug.py on ragged pages

    python tools/gen_golden.py [name ...]      # only the named groups: swin decoder edges tokenizer e2e beam pixels crop
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

from ref_import import have_reference, import_reference, reference_args  # noqa: E402
from molnextr_amd import weights as W  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
TINY = W.EncoderDims(img_size=96, patch=4, embed_dim=32, depths=(2, 2), heads=(1, 2), window=12)


def sha(t):
    return hashlib.sha256(np.ascontiguousarray(t).tobytes()).hexdigest()[:16]


def gen_swin_tiny(VT):
    sd = W.synthetic_encoder_state(0, TINY)
    model = VT(img_size=96, patch_size=4, embed_dim=32, depths=(2, 2), num_heads=(1, 2), window_size=12,
               num_classes=0).eval()
    strict = {k[len("transformer."):]: v for k, v in sd.items()}
    missing, unexpected = model.load_state_dict(strict, strict=False)
    assert not unexpected and all(m.startswith("head") for m in missing), (missing, unexpected)
    img = W.hash_normal("swin_tiny_img", (2, 3, 96, 96), 1.0)
    taps = {}
    hooks = []
    for si, layer in enumerate(model.layers):
        for bi, blk in enumerate(layer.blocks):
            hooks.append(blk.register_forward_hook(
                lambda m, i, o, name=f"s{si}b{bi}": taps.__setitem__(name, o.detach().clone())))
        if layer.downsample is not None:
            hooks.append(layer.downsample.register_forward_hook(
                lambda m, i, o, name=f"merge{si}": taps.__setitem__(name, o[0].detach().clone())))
    hooks.append(model.patch_embed.register_forward_hook(
        lambda m, i, o: taps.__setitem__("patch_embed", o[0].detach().clone())))
    with torch.no_grad():
        feats, hiddens = model(img)
    for h in hooks:
        h.remove()
    out = {k: v.numpy() for k, v in taps.items()}
    out["features"] = feats.numpy()
    np.savez_compressed(os.path.join(GOLD, "swin_tiny.npz"), **out)
    print("swin_tiny:", {k: v.shape for k, v in out.items()})


def gen_swin_full(Encoder, args, ck):
    enc = Encoder(args).eval()
    enc.load_state_dict(ck["encoder"], strict=True)
    img = W.synthetic_images(2)
    with torch.no_grad():
        f, hiddens = enc(img)
    f = f.numpy()
    out = {
        "features_head": f[:, :4, :].copy(),                 # first 4 tokens, all channels
        "features_strided": f[:, ::9, ::16].copy(),          # 16 tokens x 64 channels
        "features_mean": f.mean(axis=(1, 2)), "features_std": f.std(axis=(1, 2)),
        "features_abs_sum": np.abs(f).sum(axis=(1, 2)),
        "hidden0_strided": hiddens[0].numpy()[:, ::512, ::8].copy(),
        "hidden2_strided": hiddens[2].numpy()[:, ::36, ::32].copy(),
    }
    np.savez_compressed(os.path.join(GOLD, "swin_full.npz"), **out)
    print("swin_full: features", f.shape, "rms", float(np.sqrt((f ** 2).mean())))
    return torch.from_numpy(f)


def run_greedy(dec, feats, max_length, n_logit_steps=4):
    ar = dec.decoder["chartok_coords"]
    logits = []
    h = ar.output_layer.register_forward_hook(lambda m, i, o: logits.append(o.detach().clone()))
    with torch.no_grad():
        preds, scores, token_scores, hidden = ar.decode(feats, 1, 1, max_length=max_length)
    h.remove()
    B = feats.shape[0]
    ids = np.full((B, max_length), -1, dtype=np.int32)
    logp = np.zeros((B, max_length), dtype=np.float32)
    lens = np.zeros(B, dtype=np.int32)
    hid_head = np.zeros((B, 8, 256), dtype=np.float32)
    hid_sum = np.zeros((B, 256), dtype=np.float64)
    for b in range(B):
        t = preds[b][0].numpy()
        lens[b] = len(t)
        ids[b, :len(t)] = t
        logp[b, :len(t)] = np.log(np.array(token_scores[b][0], dtype=np.float64)).astype(np.float32)
        hb = hidden[b][0].numpy()
        hid_head[b, :min(8, len(t))] = hb[:8]
        hid_sum[b] = hb.astype(np.float64).sum(0)
    out = {"ids": ids, "lens": lens, "token_logp": logp, "scores": np.array([s[0] for s in scores], dtype=np.float64),
           "hidden_head": hid_head, "hidden_sum": hid_sum}
    for s in range(min(n_logit_steps, len(logits))):
        out[f"logits_step{s}"] = logits[s].squeeze(1).numpy()
    return out, preds, hidden


def gen_decoder(Decoder, args, tok, ck):
    dec = Decoder(args, tok).eval()
    dec.load_state_dict(ck["decoder"], strict=True)
    feats = W.hash_normal("decoder_greedy_features", (6, 144, 1024), 0.5)
    out, _, _ = run_greedy(dec, feats, 480)
    np.savez_compressed(os.path.join(GOLD, "decoder_greedy.npz"), **out)
    print("decoder_greedy: lens", out["lens"].tolist())
    feats3 = W.hash_normal("decoder_short_features", (3, 144, 1024), 0.5)
    out3, _, _ = run_greedy(dec, feats3, 24)
    np.savez_compressed(os.path.join(GOLD, "decoder_short.npz"), **out3)
    print("decoder_short: lens", out3["lens"].tolist())
    # row-indexed PE
    ar = dec.decoder["chartok_coords"]
    ids = torch.tensor([1, 57, 130, 200, 2, 57, 57]).view(-1, 1, 1)
    with torch.no_grad():
        emb, _ = ar.dec_embedding(ids)
    np.savez_compressed(os.path.join(GOLD, "embedding_pe.npz"), ids=ids.view(-1).numpy().astype(np.int32),
                        emb=emb.squeeze(1).numpy())
    return dec


def gen_edges(dec, get_edge_prediction):
    gp = dec.decoder["edges"]
    cases = {}
    for name, T, k in (("a", 40, 9), ("b", 90, 30), ("c", 12, 1), ("d", 20, 2)):
        hidden = W.hash_normal(f"edges_hidden_{name}", (T, 256), 1.0)
        u = W.hash_uniform(f"edges_idx_{name}", k)
        idx = np.sort((u * T).astype(np.int64))
        with torch.no_grad():
            pred = gp(hidden.unsqueeze(0), torch.from_numpy(idx).unsqueeze(0))
            prob = torch.softmax(pred["edges"].squeeze(0).permute(1, 2, 0), dim=2)
        e, s = get_edge_prediction(prob.tolist())
        cases[f"{name}_idx"] = idx.astype(np.int32)
        cases[f"{name}_logits"] = pred["edges"].squeeze(0).permute(1, 2, 0).numpy()
        cases[f"{name}_edges"] = np.array(e, dtype=np.int32).reshape(k, k)
        cases[f"{name}_scores"] = np.array(s, dtype=np.float64).reshape(k, k)
    # symmetrisation on raw (non-normalised) tables
    p = W.hash_uniform("edge_prob_raw", 7 * 7 * 7).reshape(7, 7, 7)
    e, s = get_edge_prediction(p.tolist())
    cases["raw_prob"] = p
    cases["raw_edges"] = np.array(e, dtype=np.int32)
    cases["raw_scores"] = np.array(s, dtype=np.float64)
    np.savez_compressed(os.path.join(GOLD, "edges.npz"), **cases)
    print("edges: cases a,b,c,d,raw")


def gen_tokenizer(tok, decoded_ids):
    t = tok["chartok_coords"]
    masks = ["".join("1" if m else "0" for m in t.get_output_mask(i)) for i in range(len(t))]
    # run-length: store per id the (first_forbidden, last_forbidden+1) since masks are one contiguous run or empty
    cases = []
    s = t.stoi
    hand = [
        [],
        [2],
        [s["C"], 110, 180, 2],
        [s["C"], 110, 180, s["C"], 111, 181, 2],
        [s["C"], s["l"], 120, 190, s["B"], s["r"], 121, 191, s["O"], 2],
        [s["["], s["C"], s["@"], s["H"], s["]"], 130, 200, s["("], s["N"], 131, 201, s[")"], 2],
        [s["["], s["N"], s["H"], 130, 200, s["O"], 2],          # unterminated bracket stops at a coordinate id
        [s["["], s["P"], s["h"], s["]"], 110, 187, s["*"], 140, 170, 3, 150, 175, s["="], 2],
        [s["C"], 110, 180],                                     # x y at the very end: no following position -> atom dropped
        [s["C"], 110],                                          # x without y
        [s["c"], s["1"], 115, 166, s["c"], 116, 167, s["1"], 2],
        [1, 4, s["C"], 110, 180, 0, s["N"]],                     # <sos>/<mask> mid-sequence, stop at <pad>
        [s["R"], 101, 228, s["ŕ"], 164, 165, s["~"], s["C"], 2],
    ]
    for seq in hand + [ids for ids in decoded_ids]:
        seq = [int(v) for v in seq]
        cases.append({"ids": seq, "out": t.sequence_to_smiles(seq)})
    with open(os.path.join(GOLD, "tokenizer.json"), "w") as f:
        json.dump({"vocab_size": len(t), "offset": t.offset, "masks": masks, "cases": cases}, f)
    print("tokenizer:", len(cases), "cases")


def gen_e2e(dec, feats):
    with torch.no_grad():
        preds = dec.decode(feats)
    out = []
    for p in preds:
        c = p["chartok_coords"]
        out.append({"smiles": c["smiles"], "symbols": c["symbols"], "coords": c["coords"], "indices": c["indices"],
                    "edges": p["edges"]})
    with open(os.path.join(GOLD, "predict_e2e.json"), "w") as f:
        json.dump({"features": "hash_normal('e2e_features',(4,144,1024),0.5)", "preds": out}, f)
    print("predict_e2e: atoms", [len(o["symbols"]) for o in out])


def gen_e2e_confidence(dec, feats):
    """Decoder.decode with compute_confidence=True (reference components.py:456-469,485-491)."""
    dec.compute_confidence = True
    with torch.no_grad():
        preds = dec.decode(feats)
    dec.compute_confidence = False
    out = []
    for p in preds:
        c = p["chartok_coords"]
        es = np.array(p["edge_scores"], dtype=np.float64)
        out.append({"smiles": c["smiles"], "symbols": c["symbols"], "indices": c["indices"],
                    "atom_scores": c["atom_scores"], "overall_score": p["overall_score"],
                    # full matrix for the first (small) sample only; row sums + log-product for the rest
                    "edge_scores": p["edge_scores"] if len(out) == 0 else None,
                    "edge_score_row_sums": es.sum(axis=1).tolist(),
                    "edge_score_log_sum": float(np.log(es).sum())})
    with open(os.path.join(GOLD, "predict_e2e_conf.json"), "w") as f:
        json.dump({"features": "hash_normal('conf_features',(3,144,1024),0.5)", "preds": out}, f)
    print("predict_e2e_conf: atoms", [len(o["symbols"]) for o in out])


def beam_script(name, steps, B, K, V, eos_boost):
    """Scripted per-(step, image, beam) log-prob rows: log_softmax of hash-normal logits, EOS (id 2) boosted."""
    z = W.hash_normal(name, (steps, B, K, V), 1.0).clone()
    z[..., 2] += eos_boost
    return torch.log_softmax(z, dim=-1)


def gen_beam_strategy():
    """The reference's BeamSearch class driven DIRECTLY with scripted log-probs (its decode loop cannot run: the
    constructor passes max_length/return_attention to DecodeStrategy in swapped positions and decode() calls advance
    with the wrong arity). The swapped positions are compensated at the call site below, so that every line of
    advance/_pick/update_finished executes as written."""
    from MolNexTR.decoding.beam_search import BeamSearch
    out = {}
    cases = [("a", 4, 3, 2, 7, 24, 1.5), ("b", 3, 5, 1, 9, 32, 0.2), ("c", 2, 2, 2, 4, 16, -4.0), ("d", 5, 4, 4, 12, 20, 1.0)]
    for name, B, K, NB, ML, V, boost in cases:
        table = beam_script("beam_script_" + name, ML, B, K, V, boost)
        bs = BeamSearch(pad=0, bos=1, eos=2, batch_size=B, beam_size=K, n_best=NB, min_length=1,
                        return_attention=ML, max_length=False)   # lands as max_length=ML, return_attention=False
        assert bs.max_length == ML and bs.return_attention is False
        bs.initialize(torch.zeros(B, 3, 4))
        sel, toks, scs, offs = [], [], [], []
        for step in range(ML):
            origin = bs.batch_offset.tolist()
            lp = torch.stack([table[step, b, j] for b in origin for j in range(K)]).clone()
            bs.advance(lp, None)
            sc_step = bs.topk_scores.clone()
            if bs.is_finished.any():
                bs.update_finished()
                if bs.done:
                    scs.append(sc_step.numpy()); sel.append(np.zeros(0, np.int64)); toks.append(np.zeros(0, np.int64))
                    offs.append(np.zeros(0, np.int64))
                    break
            scs.append(sc_step.numpy())
            sel.append(bs.select_indices.numpy().copy())
            toks.append(bs.topk_ids.reshape(-1).numpy().copy())
            offs.append(bs.batch_offset.numpy().copy())
        assert bs.done, name
        out[name + "_cfg"] = np.array([B, K, NB, ML, V], dtype=np.int64)
        out[name + "_boost"] = np.array([boost])
        out[name + "_nsteps"] = np.array([len(sel)])
        for i, (a, b, c, d) in enumerate(zip(sel, toks, scs, offs)):
            out[f"{name}_sel{i}"], out[f"{name}_tok{i}"], out[f"{name}_score{i}"], out[f"{name}_origin{i}"] = a, b, c, d
        for b in range(B):
            for r in range(NB):
                out[f"{name}_pred{b}_{r}"] = bs.predictions[b][r].numpy()
                out[f"{name}_final{b}_{r}"] = np.array([bs.scores[b][r]], dtype=np.float64)
        print("beam_strategy", name, "steps", len(sel), "lens", [[len(p) for p in bs.predictions[b]] for b in range(B)])
    np.savez_compressed(os.path.join(GOLD, "beam_strategy.npz"), **out)


def _greedy_with_margins(dec, tok, feats, max_length, n_logit_steps=4):
    """TransformerDecoderAR.decode (greedy) on `feats`, capturing every step's raw logits through a forward hook on the
    reference's own output_layer. Returns ids / lens / token log-probs as the reference reports them, the raw logits of
    the first steps, and per (row, step) the top-1 minus top-2 margin of the masked log-probs the strategy saw
    (log_softmax -> grammar mask -10000 -> EOS ban at step 0), with the argmax cross-checked against the emitted id."""
    ar = dec.decoder["chartok_coords"]
    t = tok["chartok_coords"]
    logits = []
    h = ar.output_layer.register_forward_hook(lambda m, i, o: logits.append(o.detach().squeeze(1).clone()))
    with torch.no_grad():
        preds, scores, token_scores, hidden = ar.decode(feats, 1, 1, max_length=max_length)
    h.remove()
    B = feats.shape[0]
    lens = np.array([len(preds[b][0]) for b in range(B)], dtype=np.int32)
    T = int(lens.max())
    ids = np.full((B, T), -1, dtype=np.int32)
    logp = np.zeros((B, T), dtype=np.float32)
    margin = np.full((B, T), np.inf, dtype=np.float32)
    for b in range(B):
        ids[b, :lens[b]] = preds[b][0].numpy()
        logp[b, :lens[b]] = np.log(np.array(token_scores[b][0], dtype=np.float64)).astype(np.float32)
    for s, lg in enumerate(logits):
        alive = [b for b in range(B) if lens[b] > s]            # compaction keeps the original order
        assert lg.shape[0] == len(alive), (s, lg.shape, len(alive))
        lp = torch.log_softmax(lg, dim=-1)
        prev = [1 if s == 0 else int(ids[b, s - 1]) for b in alive]
        mask = torch.tensor([t.get_output_mask(p) for p in prev])
        lp = lp.masked_fill(mask, -10000.0)
        if s == 0:
            lp[:, 2] = -1e20
        top2 = lp.topk(2, dim=-1)
        for r, b in enumerate(alive):
            assert int(top2.indices[r, 0]) == int(ids[b, s]), (s, b)
            margin[b, s] = float(top2.values[r, 0] - top2.values[r, 1])
    out = {"ids": ids, "lens": lens, "token_logp": logp, "margin": margin}
    for s in range(min(n_logit_steps, len(logits))):
        out[f"logits_step{s}"] = logits[s].numpy()
    return out, preds, hidden


def gen_pixels(Encoder, Decoder, args, tok, ck):
    """The composition the suite never checked in round 1: pixels -> Encoder.forward -> Decoder.decode, run by the
    reference's own classes. Inputs are W.synthetic_images (pure functions of the image index), weights
    synthetic_checkpoint(0) (molecule-like decoder) and synthetic_checkpoint(0, molecule_like=False)."""
    N = 32
    enc = Encoder(reference_args()).eval()
    enc.load_state_dict(ck["encoder"], strict=True)
    img = W.synthetic_images(N)
    with torch.no_grad():
        feats = torch.cat([enc(img[i:i + 4])[0] for i in range(0, N, 4)])
    out = {"feat_strided": feats[:, ::9, ::16].numpy().copy(),
           "feat_rms": np.array([float(feats.pow(2).mean().sqrt())])}
    dec = Decoder(args, tok).eval()
    dec.load_state_dict(ck["decoder"], strict=True)
    plain = W.synthetic_checkpoint(0, molecule_like=False)
    dec_p = Decoder(args, tok).eval()
    dec_p.load_state_dict(plain["decoder"], strict=True)
    for name, d, B, ml in (("m6", dec, 6, 480), ("m32", dec, 32, 480), ("p6", dec_p, 6, 64), ("p32", dec_p, 32, 64)):
        o, _, _ = _greedy_with_margins(d, tok, feats[:B], ml)
        for k, v in o.items():
            out[f"{name}_{k}"] = v
        fin = np.isfinite(o["margin"])
        print(f"pixels {name}: lens {o['lens'].tolist()} margin min {o['margin'][fin].min():.4f} "
              f"median {np.median(o['margin'][fin]):.3f}")
    np.savez_compressed(os.path.join(GOLD, "pixels_e2e.npz"), **out)
    # Decoder.decode end to end (detokenise + bond head) on the batch of 32 and the batch of 6
    js = {}
    for name, B in (("m32", 32), ("m6", 6)):
        with torch.no_grad():
            preds = dec.decode(feats[:B])
        js[name] = [{"smiles": p["chartok_coords"]["smiles"], "symbols": p["chartok_coords"]["symbols"],
                     "coords": p["chartok_coords"]["coords"], "indices": p["chartok_coords"]["indices"],
                     "edges": p["edges"]} for p in preds]
    with open(os.path.join(GOLD, "pixels_e2e.json"), "w") as f:
        json.dump({"images": "W.synthetic_images(32)", "checkpoint": "W.synthetic_checkpoint(0)", "preds": js}, f)
    print("pixels_e2e: atoms", [len(p["symbols"]) for p in js["m32"]])


def gen_autocast(Encoder, Decoder, args, tok, ck):
    """The reference's own evaluation runs its encoder and decoder under fp16 autocast (main.py:277, exps/eval.sh:29
    `--fp16`). This records how far THAT moves the result from the fp32 path of pixels_e2e (same images, same weights):
    strided features, and which of the 32 molecules keep symbols / coordinates / bonds. torch.autocast('cpu', float16) stands
    in for torch.cuda.amp.autocast (no CUDA here): the same op list is cast, accumulation details differ."""
    N = 32
    enc = Encoder(reference_args()).eval()
    enc.load_state_dict(ck["encoder"], strict=True)
    dec = Decoder(args, tok).eval()
    dec.load_state_dict(ck["decoder"], strict=True)
    img = W.synthetic_images(N)
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.float16):
        feats = torch.cat([enc(img[i:i + 4])[0] for i in range(0, N, 4)])
        preds = dec.decode(feats)
    ref = np.load(os.path.join(GOLD, "pixels_e2e.npz"))["feat_strided"]
    with open(os.path.join(GOLD, "pixels_e2e.json")) as f:
        ref_preds = json.load(f)["preds"]["m32"]
    fs = feats.float()[:, ::9, ::16].numpy().copy()
    same = [bool(p["chartok_coords"]["symbols"] == r["symbols"] and p["chartok_coords"]["coords"] == r["coords"] and
                 p["edges"] == r["edges"]) for p, r in zip(preds, ref_preds)]
    np.savez_compressed(os.path.join(GOLD, "pixels_autocast_fp16.npz"), feat_strided=fs.astype(np.float16))
    js = {"what": "reference Encoder.forward + Decoder.decode under torch.autocast('cpu', float16) vs its own fp32 run (pixels_e2e)",
          "feature_max_err_vs_fp32": float(np.abs(fs - ref).max()), "feature_rms_err_vs_fp32": float(np.sqrt(((fs - ref) ** 2).mean())),
          "molecule_identical_to_fp32": same, "molecules_identical": int(sum(same)),
          "smiles_identical": int(sum(p["chartok_coords"]["smiles"] == r["smiles"] for p, r in zip(preds, ref_preds)))}
    with open(os.path.join(GOLD, "pixels_autocast_fp16.json"), "w") as f:
        json.dump(js, f)
    print("pixels autocast fp16:", {k: v for k, v in js.items() if k not in ("what", "molecule_identical_to_fp32")})


def gen_stress(Encoder, Decoder, args, tok):
    """The path from pixels on a SECOND, hostile checkpoint — W.synthetic_checkpoint(1, stress=True): LayerNorm gains in
    [0.1, 8] with x50 outlier channels, per-matrix weight scales from 0.1 to 4 times the variance-preserving one,
    relative-position biases up to +-8, outlier channels in the residual stream (molnextr_amd/weights.py::_stress_encoder) —
    run by the reference's own Encoder / Decoder classes on 16 synthetic images (a different image range than pixels_e2e).
    Everything test_gpu_pixels.py asserts for the exact operand modes on pixels_e2e is asserted on this fixture too."""
    N = 16
    ck = W.synthetic_checkpoint(1, stress=True)
    enc = Encoder(reference_args()).eval()
    enc.load_state_dict(ck["encoder"], strict=True)
    img = W.synthetic_images(N, first_index=700)
    with torch.no_grad():
        feats = torch.cat([enc(img[i:i + 4])[0] for i in range(0, N, 4)])
    out = {"feat_strided": feats[:, ::9, ::16].numpy().copy(), "feat_rms": np.array([float(feats.pow(2).mean().sqrt())]),
           "feat_absmax": np.array([float(feats.abs().max())])}
    dec = Decoder(args, tok).eval()
    dec.load_state_dict(ck["decoder"], strict=True)
    o, _, _ = _greedy_with_margins(dec, tok, feats, 480)
    for k, v in o.items():
        out[f"s16_{k}"] = v
    fin = np.isfinite(o["margin"])
    print(f"stress s16: feature rms {out['feat_rms'][0]:.3f} absmax {out['feat_absmax'][0]:.1f} lens {o['lens'].tolist()} "
          f"margin min {o['margin'][fin].min():.4f} median {np.median(o['margin'][fin]):.3f}")
    np.savez_compressed(os.path.join(GOLD, "pixels_stress.npz"), **out)
    with torch.no_grad():
        preds = dec.decode(feats)
    js = [{"smiles": p["chartok_coords"]["smiles"], "symbols": p["chartok_coords"]["symbols"],
           "coords": p["chartok_coords"]["coords"], "indices": p["chartok_coords"]["indices"], "edges": p["edges"]}
          for p in preds]
    with open(os.path.join(GOLD, "pixels_stress.json"), "w") as f:
        json.dump({"images": "W.synthetic_images(16, first_index=700)", "checkpoint": "W.synthetic_checkpoint(1, stress=True)",
                   "preds": {"s16": js}}, f)
    print("pixels_stress: atoms", [len(p["symbols"]) for p in js])


def gen_crop_pad():
    """CropWhite(pad=50) and PadToSquare of the reference's own data_aug.py (imported through a minimal
    albumentations / cv2 stand-in: both classes are pure numpy apart from a constant-border pad) on ragged pages:
    crop parameters, output shapes and content hashes. Resize / ToGray stay unpinned (OpenCV is not installed)."""
    from MolNexTR.data_aug import CropWhite, PadToSquare
    cw, ps = CropWhite(pad=50), PadToSquare()
    cases = []
    for i in range(len(W.PAGE_CASES)):
        page = W.synthetic_page(i)
        params = cw.update_params({}, image=page)
        out = cw.apply(page, **{k: v for k, v in params.items() if k.startswith("crop_")})
        sq = ps.apply(out)
        cases.append({"case": i, "page_shape": list(page.shape), "page_sha": sha(page),
                      "crop": [int(params.get(k, 0)) for k in ("crop_top", "crop_bottom", "crop_left", "crop_right")],
                      "cropped_shape": list(out.shape), "cropped_sha": sha(out),
                      "square_shape": list(sq.shape), "square_sha": sha(sq)})
    with open(os.path.join(GOLD, "crop_pad.json"), "w") as f:
        json.dump({"pages": "W.synthetic_page(case)", "cases": cases}, f, indent=0)
    print("crop_pad:", [(c["crop"], c["square_shape"][:2]) for c in cases])


def main():
    if not have_reference():
        raise SystemExit("/root/reference is not mounted: fixtures can only be regenerated in the build container")
    torch.set_num_threads(8)
    os.makedirs(GOLD, exist_ok=True)
    import_reference()
    from MolNexTR.components import Encoder, Decoder, get_edge_prediction
    from MolNexTR.models.transformers import Vision_Transformer
    from MolNexTR.tokenization import get_tokenizer
    args = reference_args()
    tok = get_tokenizer(args)
    with open(os.path.join(ROOT, "molnextr_amd", "vocab", "vocab_chars.json")) as f:
        assert json.load(f) == tok["chartok_coords"].stoi, "vocab drifted from the reference"
    ck = W.synthetic_checkpoint(0)
    want = set(sys.argv[1:]) or {"swin", "decoder", "edges", "tokenizer", "e2e", "beam", "pixels", "stress", "crop"}
    if "swin" in want:
        gen_swin_tiny(Vision_Transformer)
        gen_swin_full(Encoder, args, ck)
    args.encoder_dim = 1024
    dec = None
    if want & {"decoder", "edges", "e2e"}:
        dec = gen_decoder(Decoder, args, tok, ck)           # (cheap; also the model the next groups run)
    if "edges" in want:
        gen_edges(dec, get_edge_prediction)
    if "tokenizer" in want:
        g = np.load(os.path.join(GOLD, "decoder_greedy.npz"))
        decoded = [g["ids"][b, :g["lens"][b]].tolist() for b in range(g["ids"].shape[0])]
        gen_tokenizer(tok, decoded)
    if "e2e" in want:
        gen_e2e(dec, W.hash_normal("e2e_features", (4, 144, 1024), 0.5))
        gen_e2e_confidence(dec, W.hash_normal("conf_features", (3, 144, 1024), 0.5))
    if "beam" in want:
        gen_beam_strategy()
    if "pixels" in want:
        gen_pixels(Encoder, Decoder, args, tok, ck)
    if want & {"pixels", "autocast"}:
        gen_autocast(Encoder, Decoder, args, tok, ck)
    if "stress" in want:
        gen_stress(Encoder, Decoder, args, tok)
    if "crop" in want:
        gen_crop_pad()
    sizes = {f: os.path.getsize(os.path.join(GOLD, f)) for f in sorted(os.listdir(GOLD))}
    print("fixture bytes:", sizes, "total", sum(sizes.values()))


if __name__ == "__main__":
    main()
