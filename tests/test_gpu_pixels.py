"""GPU (-m gpu): the path AS A UNIT, from pixels — `features = encoder(images); predictions = decoder.decode(features)`
(reference MolNexTR/model.py:107-108) — against outputs of the reference's own classes on the same images
(tests/golden/pixels_e2e.*, written by tools/gen_golden.py from /root/reference in the build container).

Round 1 only ever compared tokens with the oracle fed the GPU's own features; an argmax flipped by the encoder's
operand rounding would have gone unnoticed. Here nothing of the GPU's output is handed to the checker:

  * fp32 parity mode (compute_dtype FP32: every encoder operand fp32 on the exact-fp32 MFMA): logits of steps 0..3 and
    the log-prob of every emitted token within 1e-3 (north_star's tolerance), every token id, length, atom position,
    coordinate and bond class EXACT for all 32 + 6 images, molecule-like and plain-random decoder;
  * bf16 / fp16 throughput modes: the same comparison, but an argmax decision whose top-1/top-2 margin is smaller than
    the logit error of the mode may legitimately flip (the reference's margins go down to 2e-4 on this workload). The
    test measures the logit error, requires every row to agree with the reference up to its first near-tie (margin
    below MARGIN_FACTOR x the measured error), and reports how many rows / steps that concerns. Numbers land in
    gpurun_out/pixels_parity.json and DESIGN.md §6.
"""
import json
import os

import numpy as np
import pytest
import torch

from molnextr_amd import weights as W

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MARGIN_FACTOR = 10.0
CASES = [("m6", 6, 480, True), ("m32", 32, 480, True), ("p6", 6, 64, False), ("p32", 32, 64, False)]
# logit / log-prob tolerance per mode: fp32 = north_star's 1e-3; the 16-bit modes state what their operand rounding gives
LOGIT_TOL = {"fp32": 1e-3, "fp16": 2e-2, "bf16": 1.5e-1}
FEAT_TOL = {"fp32": 2e-4, "fp16": 1e-2, "bf16": 6e-2}


@pytest.fixture(scope="module")
def gold(golden_dir):
    g = dict(np.load(os.path.join(golden_dir, "pixels_e2e.npz")))
    with open(os.path.join(golden_dir, "pixels_e2e.json")) as f:
        g["preds"] = json.load(f)["preds"]
    return g


@pytest.fixture(scope="module")
def images():
    return W.synthetic_images(32)


def _engines(mode, synth_ckpt):
    from molnextr_amd.engine import Engine
    plain = W.synthetic_checkpoint(0, molecule_like=False)
    mol = Engine(synth_ckpt["encoder"], synth_ckpt["decoder"], device=0, max_batch=32, dtype=mode, dec_slots=64)
    pln = Engine(synth_ckpt["encoder"], plain["decoder"], device=0, max_batch=32, dtype=mode, dec_slots=64)
    return mol, pln


def _report(name, rec):
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, "pixels_parity.json")
        cur = {}
        if os.path.exists(path):
            with open(path) as f:
                cur = json.load(f)
        cur[name] = rec
        with open(path, "w") as f:
            json.dump(cur, f, indent=1)
    except OSError:
        pass
    print("pixels parity", name, json.dumps(rec))


@pytest.mark.parametrize("mode", ["fp32", "fp16", "bf16"])
def test_path_from_pixels_vs_reference(mode, gold, images, synth_ckpt):
    from molnextr_amd.model import predict_pipeline
    dev = torch.device("cuda:0")
    mol, pln = _engines(mode, synth_ckpt)
    try:
        x = images.to(dev)
        feats = mol.encode(x)
        f = feats.cpu().numpy()
        ferr = float(np.abs(f[:, ::9, ::16] - gold["feat_strided"]).max())
        frms = float(np.sqrt(((f[:, ::9, ::16] - gold["feat_strided"]) ** 2).mean()))
        assert ferr < FEAT_TOL[mode], (mode, ferr)
        feats_p = pln.encode(x)
        assert torch.equal(feats_p, feats), "same encoder weights, same kernels: features must be bit-equal"
        summary = {"feature_max_err": ferr, "feature_rms_err": frms, "feature_rms": float(gold["feat_rms"][0])}
        for name, B, max_len, is_mol in CASES:
            eng = mol if is_mol else pln
            out = eng.decode_greedy(feats[:B].contiguous(), max_len=max_len, trace_logits=True)
            lens = out["lengths"].cpu().numpy()
            toks = out["tokens"].cpu().numpy()
            lp = out["token_logp"].cpu().numpy()
            lg = out["logits"].cpu().numpy()                       # [max_len, B, V]
            g_ids, g_lens, g_lp, g_margin = (gold[f"{name}_{k}"] for k in ("ids", "lens", "token_logp", "margin"))
            # (1) logits of the first steps (all rows are still in the batch: shortest sequence > 4 tokens); a row is
            #     compared at step s only while its own history agrees with the reference (after a flipped token the
            #     inputs differ, not just the rounding)
            logit_err = 0.0
            for s in range(4):
                gl = gold[f"{name}_logits_step{s}"]
                assert gl.shape[0] == B
                same_hist = np.array([np.array_equal(toks[b, :s], g_ids[b, :s]) for b in range(B)])
                if same_hist.any():
                    logit_err = max(logit_err, float(np.abs(lg[s][same_hist] - gl[same_hist]).max()))
            assert logit_err < LOGIT_TOL[mode], (mode, name, logit_err)
            # (2) tokens: first divergence per row, log-prob error of every emitted token up to there
            first_div, lp_err, n_steps = {}, 0.0, 0
            for b in range(B):
                n = int(min(lens[b], g_lens[b]))
                neq = np.nonzero(toks[b, :n] != g_ids[b, :n])[0]
                d = int(neq[0]) if neq.size else (n if lens[b] != g_lens[b] else None)
                upto = n if d is None else d
                if upto:
                    lp_err = max(lp_err, float(np.abs(lp[b, :upto] - g_lp[b, :upto]).max()))
                n_steps += upto
                if d is not None:
                    first_div[b] = d
            err = max(logit_err, lp_err)
            fin = np.isfinite(g_margin)
            rec = {"rows": B, "rows_exact": B - len(first_div), "steps_compared": n_steps,
                   "logit_max_err_steps0_3": logit_err, "token_logp_max_err": lp_err,
                   "ref_margin_min": float(g_margin[fin].min()), "ref_margin_median": float(np.median(g_margin[fin])),
                   "ref_steps_with_margin_below_10x_err": int((g_margin[fin] < MARGIN_FACTOR * err).sum()),
                   "first_divergence": {str(b): [d, float(g_margin[b, min(d, g_margin.shape[1] - 1)])]
                                        for b, d in first_div.items()}}
            summary[name] = rec
            if mode == "fp32":
                assert not first_div, (name, rec["first_divergence"])
                assert lp_err < 1e-3, (name, lp_err)
            elif first_div:
                # the earliest flip (later ones can be knock-on effects of the batch-row positional encoding: a row that
                # ends at another step renumbers the rows behind it) must sit on a near-tie of the reference
                b0 = min(first_div, key=lambda b: first_div[b])
                d0 = first_div[b0]
                m0 = float(g_margin[b0, d0]) if d0 < g_margin.shape[1] else 0.0
                assert m0 < MARGIN_FACTOR * err, (mode, name, b0, d0, m0, err)
        # (3) atoms / bonds through the pipeline path (mnx_predict) from pixels, against Decoder.decode's own output
        for name, B in (("m32", 32), ("m6", 6)):
            preds = predict_pipeline(mol, x[:B].contiguous(), ref_batch_size=B)
            exact, bond_only = 0, 0
            for b, (p, g) in enumerate(zip(preds, gold["preds"][name])):
                c = p["chartok_coords"]
                atoms_same = (c["smiles"] == g["smiles"] and c["symbols"] == g["symbols"] and c["indices"] == g["indices"]
                              and c["coords"] == g["coords"])
                same = atoms_same and p["edges"] == g["edges"]
                exact += bool(same)
                bond_only += bool(atoms_same and not same)
                if mode == "fp32":
                    assert same, (name, b)
                elif str(b) not in summary[name]["first_divergence"]:
                    assert atoms_same, (mode, name, b, "tokens agree with the reference but the atom set does not")
            summary[name]["molecules_with_same_atoms_but_a_flipped_bond"] = bond_only   # 7-class argmax near-ties
            summary[name]["molecules_exact_atoms_bonds"] = exact
        _report(mode, summary)
    finally:
        mol.close()
        pln.close()
