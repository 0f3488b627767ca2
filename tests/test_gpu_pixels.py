"""GPU (-m gpu): the path AS A UNIT, from pixels — `features = encoder(images); predictions = decoder.decode(features)`
(reference MolNexTR/model.py:107-108) — against outputs of the reference's own classes on the same images
(tests/golden/pixels_e2e.*, written by tools/gen_golden.py from /root/reference in the build container).

Nothing of the GPU's output is handed to the checker. Per encoder operand mode:

  * EXACT modes — `fp16x3` (split fp16 operands, three MFMA terms per product: the default of the facade and of bench.py)
    and `fp32` (exact-fp32 MFMA): logits of steps 0..3 and the log-prob of every emitted token within 1e-3 (north_star's
    tolerance), every token id, length, atom position, coordinate and bond class EXACT for all 32 + 6 images,
    molecule-like and plain-random decoder, free-running AND teacher-forced;
  * `fp16x3m` (fp16x3 with the Linears of engine.FP16X3M_TWO_TERM on TWO terms — the activation's lo plane dropped; opt-in):
    the exact-mode assertions unchanged (0 flips over every teacher-forced step, every free-running row and every molecule
    exact, both checkpoints), log-probs AND raw logits within 5e-4 on these fixtures (measured 1.8e-4 / 4.997e-4: the round-5
    review's gate for a default mode, met here to the letter — deterministic numbers: the encoder is bit-reproducible and a
    one-tile decode always takes the same tick form). Why it is opt-in all the same: tools/extended_parity.py on 384 FURTHER
    images against the oracle gives 7.2e-4 / 8.7e-4 raw-logit error, 384 more of the hostile checkpoint 1.2e-3 (0 flips, every row
    exact) — at and beyond north_star's 1e-3,
    without the headroom the gate was there to guarantee (profiles/r06_extended_parity_*.json);
  * `bf16x3` (split bf16 operands): the same logit gate (1e-3); token flips are allowed only where the teacher-forced
    trace proves a near-tie (below);
  * `fp16` / `bf16` (one 16-bit plane per operand, the fastest modes): measured, with gates at 2x the values committed in
    profiles/r03_pixels_parity.json; `fp16` additionally against the reference's own evaluation precision (fp16 autocast,
    tests/golden/pixels_autocast_fp16.*): it has to stay at least as close to the fp32 reference as the reference itself does.

Teacher forcing (mnx_decode_forced) is what makes the 16-bit numbers mean something: the engine is fed the REFERENCE ids,
so at every step its history, the finish steps of all rows and hence the positional-encoding rows are the
reference's. The log-prob error is then pure operand rounding at every one of the ~3700 steps, and a flip (argmax !=
reference id at a step) is an independent event: the test asserts that every flip sits on a reference top-1/top-2 margin
below 2 x the measured teacher-forced error, and reports flips per 1000 tokens. The free-running decode is checked
against it: the earliest divergence of a batch must be one of those flips.
"""
import json
import os

import numpy as np
import pytest
import torch

from molnextr_amd import weights as W

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [("m6", 6, 480, True), ("m32", 32, 480, True), ("p6", 6, 64, False), ("p32", 32, 64, False)]
EXACT_MODES = ("fp32", "fp16x3", "fp16x3m")
# max |logit error| over steps 0..3 and max |log-prob error| along the reference trajectory; max |feature error| (features
# have unit rms). Exact modes and bf16x3: north_star's 1e-3. 16-bit modes: 2x the measured values (profiles/r03_pixels_parity.json).
# fp16x3m: raw logits and log-probs 5e-4 on these fixtures (measured 4.997e-4 / 1.8e-4, profiles/r06_two_term_tables_gpu.json),
# features 2x the CPU emulation's 1.3e-3 max (profiles/r06_two_term_study*.json).
LOGIT_TOL = {"fp32": 1e-3, "fp16x3": 1e-3, "fp16x3m": 5e-4, "bf16x3": 1e-3, "fp16": 2e-2, "bf16": 1.2e-1}
LOGP_TOL = dict(LOGIT_TOL, fp16x3m=5e-4)
FEAT_TOL = {"fp32": 5e-5, "fp16x3": 5e-5, "fp16x3m": 2.6e-3, "bf16x3": 3e-4, "fp16": 7e-3, "bf16": 4.5e-2}
FLIP_MARGIN_FACTOR = 2.0


@pytest.fixture(scope="module")
def gold(golden_dir):
    g = dict(np.load(os.path.join(golden_dir, "pixels_e2e.npz")))
    with open(os.path.join(golden_dir, "pixels_e2e.json")) as f:
        g["preds"] = json.load(f)["preds"]
    with open(os.path.join(golden_dir, "pixels_autocast_fp16.json")) as f:
        g["autocast_fp16"] = json.load(f)
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


def _teacher_forced(eng, feats, g_ids, g_lens, g_lp, g_margin, max_len):
    """Feeds the reference ids; returns (max |log-prob error| over every step, flips [(row, step, margin)], steps)."""
    B, L = g_ids.shape
    forced = torch.zeros(B, max_len, dtype=torch.int32)
    forced[:, :L] = torch.from_numpy(g_ids.astype(np.int32))
    out = eng.decode_forced(feats, forced, max_len=max_len)
    lens = out["lengths"].cpu().numpy()
    am = out["argmax"].cpu().numpy()
    lp = out["forced_logp"].cpu().numpy()
    assert np.array_equal(lens, g_lens), "teacher-forced rows must stop exactly where the reference rows stop"
    err, flips, steps = 0.0, [], 0
    for b in range(B):
        n = int(g_lens[b])
        err = max(err, float(np.abs(lp[b, :n] - g_lp[b, :n]).max()))
        steps += n
        for t in np.nonzero(am[b, :n] != g_ids[b, :n])[0]:
            flips.append((b, int(t), float(g_margin[b, t])))
    return err, flips, steps


@pytest.mark.parametrize("mode", ["fp16x3", "fp16x3m", "fp32", "bf16x3", "fp16", "bf16"])
def test_path_from_pixels_vs_reference(mode, gold, images, synth_ckpt):
    from molnextr_amd.model import predict_pipeline
    dev = torch.device("cuda:0")
    exact = mode in EXACT_MODES
    mol, pln = _engines(mode, synth_ckpt)
    try:
        x = images.to(dev)
        feats = mol.encode(x)
        assert not mol.encoder_nonfinite()
        f = feats.cpu().numpy()
        ferr = float(np.abs(f[:, ::9, ::16] - gold["feat_strided"]).max())
        frms = float(np.sqrt(((f[:, ::9, ::16] - gold["feat_strided"]) ** 2).mean()))
        assert ferr < FEAT_TOL[mode], (mode, ferr)
        feats_p = pln.encode(x)
        assert torch.equal(feats_p, feats), "same encoder weights, same kernels: features must be bit-equal"
        summary = {"feature_max_err": ferr, "feature_rms_err": frms, "feature_rms": float(gold["feat_rms"][0])}
        tot_flips, tot_steps = 0, 0
        for name, B, max_len, is_mol in CASES:
            eng = mol if is_mol else pln
            g_ids, g_lens, g_lp, g_margin = (gold[f"{name}_{k}"] for k in ("ids", "lens", "token_logp", "margin"))
            fb = feats[:B].contiguous()
            # (1) teacher-forced along the reference trajectory: log-prob error at EVERY step, independent flips
            tf_err, flips, tf_steps = _teacher_forced(eng, fb, g_ids, g_lens, g_lp, g_margin, max_len)
            assert tf_err < LOGP_TOL[mode], (mode, name, tf_err)
            for (b, t, m) in flips:
                assert m < FLIP_MARGIN_FACTOR * tf_err, (mode, name, "flip away from a near-tie", b, t, m, tf_err)
            if exact:
                assert not flips, (mode, name, flips)
            # (2) free-running decode: logits of the first steps (a row is compared at step s only while its own history
            #     agrees with the reference), first divergence per row, log-prob error up to there
            out = eng.decode_greedy(fb, max_len=max_len, trace_logits=True)
            lens = out["lengths"].cpu().numpy()
            toks = out["tokens"].cpu().numpy()
            lp = out["token_logp"].cpu().numpy()
            lg = out["logits"].cpu().numpy()                       # [max_len, B, V]
            logit_err = 0.0
            for s in range(4):
                gl = gold[f"{name}_logits_step{s}"]
                assert gl.shape[0] == B
                same_hist = np.array([np.array_equal(toks[b, :s], g_ids[b, :s]) for b in range(B)])
                if same_hist.any():
                    logit_err = max(logit_err, float(np.abs(lg[s][same_hist] - gl[same_hist]).max()))
            assert logit_err < LOGIT_TOL[mode], (mode, name, logit_err)
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
            fin = np.isfinite(g_margin)
            rec = {"rows": B, "rows_exact": B - len(first_div), "steps_compared": n_steps,
                   "logit_max_err_steps0_3": logit_err, "token_logp_max_err_free_running": lp_err,
                   "teacher_forced": {"steps": tf_steps, "logp_max_err": tf_err, "flips": len(flips),
                                      "flips_per_1000_tokens": round(1000.0 * len(flips) / tf_steps, 3),
                                      "flip_margins": [round(m, 6) for (_, _, m) in flips][:40]},
                   "ref_margin_min": float(g_margin[fin].min()), "ref_margin_median": float(np.median(g_margin[fin])),
                   "first_divergence": {str(b): [d, float(g_margin[b, min(d, g_margin.shape[1] - 1)])]
                                        for b, d in first_div.items()}}
            summary[name] = rec
            tot_flips += len(flips)
            tot_steps += tf_steps
            if exact:
                assert not first_div, (mode, name, rec["first_divergence"])
                assert lp_err < min(1e-3, LOGP_TOL[mode]), (mode, name, lp_err)
            elif first_div:
                # the earliest divergence of the batch (later ones can be knock-on effects of the batch-row positional
                # encoding) happens with the reference history intact, so it must be one of the teacher-forced flips
                b0 = min(first_div, key=lambda b: (first_div[b], b))
                assert (b0, first_div[b0]) in {(b, t) for (b, t, _) in flips}, (mode, name, b0, first_div[b0], flips[:8])
        summary["teacher_forced_total"] = {"steps": tot_steps, "flips": tot_flips,
                                           "flips_per_1000_tokens": round(1000.0 * tot_flips / tot_steps, 3)}
        # (3) atoms / bonds through the pipeline path (mnx_predict) from pixels, against Decoder.decode's own output
        for name, B in (("m32", 32), ("m6", 6)):
            preds = predict_pipeline(mol, x[:B].contiguous(), ref_batch_size=B)
            exact_n, bond_only = 0, 0
            for b, (p, g) in enumerate(zip(preds, gold["preds"][name])):
                c = p["chartok_coords"]
                atoms_same = (c["smiles"] == g["smiles"] and c["symbols"] == g["symbols"] and c["indices"] == g["indices"]
                              and c["coords"] == g["coords"])
                same = atoms_same and p["edges"] == g["edges"]
                exact_n += bool(same)
                bond_only += bool(atoms_same and not same)
                if exact:
                    assert same, (mode, name, b)
                elif str(b) not in summary[name]["first_divergence"]:
                    assert atoms_same, (mode, name, b, "tokens agree with the reference but the atom set does not")
            summary[name]["molecules_with_same_atoms_but_a_flipped_bond"] = bond_only   # 7-class argmax near-ties
            summary[name]["molecules_exact_atoms_bonds"] = exact_n
        if mode == "fp16":
            # yardstick for the one-plane fp16 mode: the reference's OWN evaluation precision (fp16 autocast, main.py:277,
            # exps/eval.sh --fp16; tests/golden/pixels_autocast_fp16.*, torch.autocast('cpu') standing in for CUDA's).
            # The engine's fp16 operand mode must stay at least as close to the reference's fp32 result as that.
            ac = gold["autocast_fp16"]
            summary["reference_fp16_autocast"] = {k: ac[k] for k in ("feature_max_err_vs_fp32", "feature_rms_err_vs_fp32",
                                                                     "molecules_identical")}
            assert ferr <= ac["feature_max_err_vs_fp32"] and frms <= ac["feature_rms_err_vs_fp32"], (ferr, frms, ac)
            assert summary["m32"]["molecules_exact_atoms_bonds"] >= ac["molecules_identical"], (summary["m32"], ac)
        _report(mode, summary)
    finally:
        mol.close()
        pln.close()


# feature tolerances on the stress checkpoint: 4x what tools/study_split_terms.py --ckpt stress predicts on the CPU
# (profiles/r04_stress_emulation.json: fp16x3 max err 2.6e-5 / rms 1.2e-6, bf16x3 2.7e-4 / 1.4e-5 on unit-rms features; the
# largest operands it sees: 499 into fc1, 183 into fc2, 79 in the residual stream — far from the fp16 limit)
# fp16x3m: 2x the emulation's --two fc1,fc2 row (profiles/r06_two_term_study_stress.json: max 9.0e-3 — the outlier channels —, rms 4.2e-4)
STRESS_FEAT_TOL = {"fp32": 1e-4, "fp16x3": 1e-4, "fp16x3m": 1.8e-2, "bf16x3": 1.2e-3}
STRESS_LOGIT_TOL = {"fp32": 1e-3, "fp16x3": 1e-3, "fp16x3m": 5e-4, "bf16x3": 1e-3}
STRESS_LOGP_TOL = dict(STRESS_LOGIT_TOL, fp16x3m=5e-4)


@pytest.mark.parametrize("mode", ["fp16x3", "fp16x3m", "bf16x3", "fp32"])
def test_stress_checkpoint_from_pixels_vs_reference(mode, golden_dir):
    """The exact-mode assertions of test_path_from_pixels_vs_reference on a SECOND, hostile checkpoint
    (W.synthetic_checkpoint(1, stress=True): LayerNorm gains 0.1..8 with x50 outliers, per-matrix weight scales over a 40x
    range, relative-position biases up to +-8, outlier channels in the residual stream) and other images — the value ranges
    the fp16 split's assumptions (activations < 65504, per-matrix power-of-two weight scale, 2^10 softmax scale) have to
    survive on a trained Swin-B. Fixture: the reference's own Encoder / Decoder on the same pixels (tools/gen_golden.py stress)."""
    from molnextr_amd.engine import Engine
    from molnextr_amd.model import predict_pipeline
    g = dict(np.load(os.path.join(golden_dir, "pixels_stress.npz")))
    with open(os.path.join(golden_dir, "pixels_stress.json")) as f:
        gpreds = json.load(f)["preds"]["s16"]
    ck = W.synthetic_checkpoint(1, stress=True)
    dev = torch.device("cuda:0")
    eng = Engine(ck["encoder"], ck["decoder"], device=0, max_batch=16, dtype=mode, dec_slots=64)
    try:
        x = W.synthetic_images(16, first_index=700).to(dev)
        feats = eng.encode(x)
        assert not eng.encoder_nonfinite(), f"{mode}: an activation of the stress checkpoint left the operand range"
        f = feats.cpu().numpy()[:, ::9, ::16]
        ferr = float(np.abs(f - g["feat_strided"]).max())
        frms = float(np.sqrt(((f - g["feat_strided"]) ** 2).mean()))
        assert ferr < STRESS_FEAT_TOL[mode], (mode, ferr)
        ids, lens, lp, margin = (g[f"s16_{k}"] for k in ("ids", "lens", "token_logp", "margin"))
        tf_err, flips, steps = _teacher_forced(eng, feats, ids, lens, lp, margin, 480)
        assert tf_err < STRESS_LOGP_TOL[mode], (mode, tf_err)
        for (b, t, m) in flips:
            assert m < FLIP_MARGIN_FACTOR * tf_err, (mode, "flip away from a near-tie", b, t, m, tf_err)
        out = eng.decode_greedy(feats, max_len=480, trace_logits=True)
        lg = out["logits"].cpu().numpy()
        logit_err = max(float(np.abs(lg[s] - g[f"s16_logits_step{s}"]).max()) for s in range(4))
        assert logit_err < STRESS_LOGIT_TOL[mode], (mode, logit_err)
        rec = {"feature_max_err": ferr, "feature_rms_err": frms, "feature_rms": float(g["feat_rms"][0]),
               "logit_max_err_steps0_3": logit_err,
               "teacher_forced": {"steps": steps, "logp_max_err": tf_err, "flips": len(flips)},
               "ref_margin_min": float(margin[np.isfinite(margin)].min())}
        if mode in EXACT_MODES:
            assert not flips, (mode, flips)
            toks, ln = out["tokens"].cpu().numpy(), out["lengths"].cpu().numpy()
            assert np.array_equal(ln, lens)
            for b in range(16):
                assert np.array_equal(toks[b, :lens[b]], ids[b, :lens[b]]), (mode, "row", b)
            preds = predict_pipeline(eng, x, ref_batch_size=16)
            for b, (p, q) in enumerate(zip(preds, gpreds)):
                c = p["chartok_coords"]
                assert (c["smiles"] == q["smiles"] and c["symbols"] == q["symbols"] and c["indices"] == q["indices"]
                        and c["coords"] == q["coords"] and p["edges"] == q["edges"]), (mode, "molecule", b)
            rec["molecules_exact_atoms_bonds"] = 16
            rec["free_running_rows_exact"] = 16
        else:
            # bf16x3 — the mode the range fallback lands in (model.py::_with_fallback): free-running rows and molecules are
            # asserted too. A row may leave the reference's path only where the REFERENCE's own top-1 / top-2 gap is a
            # near-tie by the teacher-forced yardstick (margin < FLIP_MARGIN_FACTOR x the measured log-prob error); on this
            # fixture (minimum margin 2.7e-3, log-prob error 4e-5) that never happens: 16 / 16 rows and molecules.
            toks, ln = out["tokens"].cpu().numpy(), out["lengths"].cpu().numpy()
            rows_exact, exact_rows = 0, set()
            for b in range(16):
                if ln[b] == lens[b] and np.array_equal(toks[b, :lens[b]], ids[b, :lens[b]]):
                    rows_exact += 1
                    exact_rows.add(b)
                    continue
                n = int(min(ln[b], lens[b]))
                diff = np.nonzero(toks[b, :n] != ids[b, :n])[0]
                t = int(diff[0]) if len(diff) else n - 1
                assert margin[b, t] < FLIP_MARGIN_FACTOR * tf_err, (mode, "row leaves the reference away from a near-tie", b, t,
                                                                    float(margin[b, t]), tf_err)
            preds = predict_pipeline(eng, x, ref_batch_size=16)
            mol_exact = 0
            for b, (p, q) in enumerate(zip(preds, gpreds)):
                c = p["chartok_coords"]
                atoms_same = (c["smiles"] == q["smiles"] and c["symbols"] == q["symbols"] and c["indices"] == q["indices"]
                              and c["coords"] == q["coords"])
                mol_exact += int(atoms_same and p["edges"] == q["edges"])
                if b in exact_rows:         # a row whose tokens are the reference's gives the reference's atoms
                    assert atoms_same, (mode, "molecule", b, "tokens agree with the reference but the atom set does not")
            rec["free_running_rows_exact"] = rows_exact
            rec["molecules_exact_atoms_bonds"] = mol_exact
            if not flips:
                assert rows_exact == 16 and mol_exact == 16, (mode, rows_exact, mol_exact)
        _report("stress_" + mode, rec)
    finally:
        eng.close()


def test_split_mode_error_budget_by_op_class(gold, images, synth_ckpt):
    """Which op classes need the three-term products? fp16x3 engine, 8 images: the feature error vs the reference with
    every class on three terms, with ONE class at a time reduced to the hi.hi term (= that class computed as the plain
    fp16 mode would), and with every class reduced. Each single reduction must visibly cost accuracy (otherwise the class
    would not need to pay 3x), and all-reduced must land at the plain fp16 mode's error."""
    from molnextr_amd.engine import Engine, SPLIT_CLASSES
    dev = torch.device("cuda:0")
    eng = Engine(synth_ckpt["encoder"], synth_ckpt["decoder"], device=0, max_batch=8, dtype="fp16x3", dec_slots=64)
    try:
        x = images[:8].to(dev)
        g = gold["feat_strided"][:8]

        def err():
            f = eng.encode(x).cpu().numpy()[:, ::9, ::16]
            return float(np.abs(f - g).max()), float(np.sqrt(((f - g) ** 2).mean()))

        table = {}
        eng.set_split_terms(None)
        table["all classes x3"] = err()
        for c in SPLIT_CLASSES:
            eng.set_split_terms([k for k in SPLIT_CLASSES if k != c])
            table[f"{c} x1"] = err()
        eng.set_split_terms([])
        table["all classes x1"] = err()
        eng.set_split_terms(None)
        assert table["all classes x3"] == err(), "restoring the mask must restore the result bit for bit"
        _report("fp16x3_error_budget", {k: {"feature_max_err": v[0], "feature_rms_err": v[1]} for k, v in table.items()})
        base = table["all classes x3"][1]
        assert base < 1e-5
        for c in SPLIT_CLASSES:
            assert table[f"{c} x1"][1] > 5 * base, (c, table[f"{c} x1"], base)
        assert 2e-4 < table["all classes x1"][1] < 3e-3
    finally:
        eng.close()


def test_one_term_attention_measured_in_flips_on_both_checkpoints(gold, images, synth_ckpt, golden_dir):
    """Could the window attention alone run on ONE product term (hi.hi: no lo planes of q / k / v / P to move or multiply)?
    Its class has the smallest feature error of the six when reduced (test_split_mode_error_budget_by_op_class). Measured the
    way that decides it — teacher-forced log-prob error and argmax flips along the reference ids, on both checkpoints — and
    recorded; the gate for shipping it would be 0 flips and <= 3e-4 on the log-probs. It is not shipped: this test pins WHY
    (the error is an order of magnitude above the three-term engine's) and that the three-term engine stays the exact one."""
    from molnextr_amd.engine import Engine, SPLIT_CLASSES
    dev = torch.device("cuda:0")
    rec = {}
    cases = [("e2e", synth_ckpt, images, gold, "m32", 32, 0)]
    gs = dict(np.load(os.path.join(golden_dir, "pixels_stress.npz")))
    cases.append(("stress", W.synthetic_checkpoint(1, stress=True), W.synthetic_images(16, first_index=700), gs, "s16", 16, 700))
    for tag, ck, imgs, g, case, n, _ in cases:
        eng = Engine(ck["encoder"], ck["decoder"], device=0, max_batch=n, dtype="fp16x3", dec_slots=64)
        try:
            ids, lens, lp, margin = (g[f"{case}_{k}"] for k in ("ids", "lens", "token_logp", "margin"))
            out = {}
            for name, classes in (("x3", None), ("attn_x1", [k for k in SPLIT_CLASSES if k != "attn"])):
                eng.set_split_terms(classes)
                feats = eng.encode(imgs.to(dev))
                f = feats.cpu().numpy()[:n, ::9, ::16]
                err, flips, steps = _teacher_forced(eng, feats, ids, lens, lp, margin, 480)
                out[name] = {"feature_max_err": float(np.abs(f - g["feat_strided"][:n]).max()),
                             "feature_rms_err": float(np.sqrt(((f - g["feat_strided"][:n]) ** 2).mean())),
                             "logp_max_err": err, "flips": len(flips), "steps": steps,
                             "flip_margins": [round(m, 6) for (_, _, m) in flips[:10]]}
            eng.set_split_terms(None)
            rec[tag] = out
            assert out["x3"]["flips"] == 0 and out["x3"]["logp_max_err"] < 1e-4, (tag, out["x3"])
            # one-term attention costs accuracy that the gate would have to pay for
            assert out["attn_x1"]["feature_rms_err"] > 5 * out["x3"]["feature_rms_err"], (tag, out)
        finally:
            eng.close()
    _report("attention_one_term", rec)


# Two-term tables measured on the GPU (tags as Engine.set_op_terms / tools/study_split_terms.py --two). The first entry is the
# table compute_dtype FP16X3M ships (engine.FP16X3M_TWO_TERM = include/molnextr_hip.h MNX_FP16X3M_TWO_TERM_BY_STAGE).
TWO_TERM_TABLES = {
    "fp16x3m (shipped)": None,
    "fc1,fc2 every stage (round-5 review's proposal)": ("fc1", "fc2"),
    "stage 3: fc1 fc2": ("fc1.s2", "fc2.s2"),
    "stage 3: qkv fc1 fc2": ("qkv.s2", "fc1.s2", "fc2.s2"),
    "stage 3: proj fc1 fc2": ("proj.s2", "fc1.s2", "fc2.s2"),
    "stage 3: qkv proj fc1 fc2": ("qkv.s2", "proj.s2", "fc1.s2", "fc2.s2"),
    "stages 3+4: fc1 fc2": ("fc1.s2", "fc2.s2", "fc1.s3", "fc2.s3"),
    "stage 3: qkv fc1 fc2 + stage 4: fc1 fc2": ("qkv.s2", "fc1.s2", "fc2.s2", "fc1.s3", "fc2.s3"),
    "stages 3+4: qkv fc1 fc2": ("qkv.s2", "fc1.s2", "fc2.s2", "qkv.s3", "fc1.s3", "fc2.s3"),
    "stages 3+4: qkv proj fc1 fc2": ("qkv.s2", "proj.s2", "fc1.s2", "fc2.s2", "qkv.s3", "proj.s3", "fc1.s3", "fc2.s3"),
    "every Linear of every stage": ("qkv", "proj", "fc1", "fc2", "merge"),
    # a table restricted to a block range of the stage: (tags, {stage: (first_block, last_block)})
    "stage 3 blocks 9-17: qkv fc1 fc2": (("qkv.s2", "fc1.s2", "fc2.s2"), {2: (9, 17)}),
    "stage 3 blocks 0-8: qkv fc1 fc2": (("qkv.s2", "fc1.s2", "fc2.s2"), {2: (0, 8)}),
    "stage 3 blocks 6-17: qkv fc1 fc2": (("qkv.s2", "fc1.s2", "fc2.s2"), {2: (6, 17)}),
    "stage 3 blocks 9-17: qkv proj fc1 fc2": (("qkv.s2", "proj.s2", "fc1.s2", "fc2.s2"), {2: (9, 17)}),
    "stage 3 blocks 12-17: qkv proj fc1 fc2": (("qkv.s2", "proj.s2", "fc1.s2", "fc2.s2"), {2: (12, 17)}),
    "stage 3 blocks 2-17: qkv fc1 fc2": (("qkv.s2", "fc1.s2", "fc2.s2"), {2: (2, 17)}),
    "stage 3 blocks 3-17: qkv fc1 fc2": (("qkv.s2", "fc1.s2", "fc2.s2"), {2: (3, 17)}),
    "stage 3 blocks 4-17: qkv fc1 fc2": (("qkv.s2", "fc1.s2", "fc2.s2"), {2: (4, 17)}),
    "stage 3 blocks 5-17: qkv fc1 fc2": (("qkv.s2", "fc1.s2", "fc2.s2"), {2: (5, 17)}),
    "stage 3 blocks 6-17: qkv fc1 fc2 + stage 4: fc1 fc2": (("qkv.s2", "fc1.s2", "fc2.s2", "fc1.s3", "fc2.s3"), {2: (6, 17)}),
    "stage 3 blocks 6-17: qkv fc1 fc2 + stage 4: qkv fc1 fc2": (("qkv.s2", "fc1.s2", "fc2.s2", "qkv.s3", "fc1.s3", "fc2.s3"), {2: (6, 17)}),
    "stage 3 blocks 6-17: qkv proj fc1 fc2": (("qkv.s2", "proj.s2", "fc1.s2", "fc2.s2"), {2: (6, 17)}),
    "stage 3 blocks 4-17: fc1 fc2": (("fc1.s2", "fc2.s2"), {2: (4, 17)}),
}


def test_two_term_tables_measured_on_both_checkpoints(gold, images, synth_ckpt, golden_dir):
    """Which Linear layers can drop the activation's lo plane (two MFMA terms instead of three)? fp16x3 engines, the table
    switched with mnx_set_op_terms (same weights, same kernels): features vs the reference, raw logits of steps 0..3 and
    teacher-forced log-prob error + argmax flips along the reference ids — on the four cases of the molecule-like fixture (6 and
    32 rows, trained-like and plain decoder: 6855 steps) and on the hostile checkpoint (16 rows, 6008 steps). Recorded for every
    table (profiles/r06_two_term_tables_gpu.json is this test's report); asserted for the SHIPPED table of the opt-in mode: 0
    flips, log-probs and raw logits within 5e-4 everywhere (measured 1.8e-4 / 4.997e-4; north_star: 1e-3) — and for every table 0
    flips. The record shows what each costs; the maximum over 29 000 raw logits is a noisy statistic (block ranges of stage 3:
    2.4e-4 .. 4.7e-4 with no order in it), the feature rms error (1.0e-4 .. 1.5e-4) is the smooth one.
    The CPU emulation of the same tables: profiles/r06_two_term_study*.json."""
    from molnextr_amd.engine import Engine, FP16X3M_BLOCKS, FP16X3M_TWO_TERM
    dev = torch.device("cuda:0")
    gs = dict(np.load(os.path.join(golden_dir, "pixels_stress.npz")))
    stress_ck = W.synthetic_checkpoint(1, stress=True)
    mol, pln = _engines("fp16x3", synth_ckpt)
    st = Engine(stress_ck["encoder"], stress_ck["decoder"], device=0, max_batch=16, dtype="fp16x3", dec_slots=64)
    rec = {}
    try:
        x = images.to(dev)
        xs = W.synthetic_images(16, first_index=700).to(dev)

        def measure(eng, feats, g, case, n, max_len):
            ids, lens, lp, margin = (g[f"{case}_{k}"] for k in ("ids", "lens", "token_logp", "margin"))
            fb = feats[:n].contiguous()
            err, flips, steps = _teacher_forced(eng, fb, ids, lens, lp, margin, max_len)
            lg = eng.decode_greedy(fb, max_len=max_len, trace_logits=True)["logits"].cpu().numpy()
            logit_err = max(float(np.abs(lg[s] - g[f"{case}_logits_step{s}"]).max()) for s in range(4))
            return {"logit_max_err_steps0_3": logit_err, "logp_max_err": err, "flips": len(flips), "steps": steps}

        for name, table in TWO_TERM_TABLES.items():
            tags, blocks = (FP16X3M_TWO_TERM, FP16X3M_BLOCKS) if table is None else (table if isinstance(table[1], dict) else (table, {}))
            for e in (mol, pln, st):
                e.set_op_terms(tags, blocks)
            feats = mol.encode(x)
            f = feats.cpu().numpy()[:, ::9, ::16]
            r = {"two_term": list(tags), "blocks": {str(k): list(v) for k, v in blocks.items()}, "e2e": {"feature_max_err": float(np.abs(f - gold["feat_strided"]).max()),
                                                 "feature_rms_err": float(np.sqrt(((f - gold["feat_strided"]) ** 2).mean()))}}
            for case, n, max_len, is_mol in CASES:
                r["e2e"][case] = measure(mol if is_mol else pln, feats, gold, case, n, max_len)
            fs = st.encode(xs)
            f = fs.cpu().numpy()[:, ::9, ::16]
            r["stress"] = {"feature_max_err": float(np.abs(f - gs["feat_strided"]).max()),
                           "feature_rms_err": float(np.sqrt(((f - gs["feat_strided"]) ** 2).mean())),
                           "s16": measure(st, fs, gs, "s16", 16, 480)}
            r["logit_max_err"] = max([r["e2e"][c]["logit_max_err_steps0_3"] for c, *_ in CASES] + [r["stress"]["s16"]["logit_max_err_steps0_3"]])
            r["logp_max_err"] = max([r["e2e"][c]["logp_max_err"] for c, *_ in CASES] + [r["stress"]["s16"]["logp_max_err"]])
            r["flips"] = sum(r["e2e"][c]["flips"] for c, *_ in CASES) + r["stress"]["s16"]["flips"]
            rec[name] = r
        mol.set_op_terms(())
        again = mol.encode(x)
        eng2 = Engine(synth_ckpt["encoder"], synth_ckpt["decoder"], device=0, max_batch=32, dtype="fp16x3m", dec_slots=64)
        try:
            mol.set_op_terms(FP16X3M_TWO_TERM, FP16X3M_BLOCKS)
            assert torch.equal(eng2.encode(x), mol.encode(x)), "FP16X3M == FP16X3 + its table (same weights, same kernels)"
        finally:
            eng2.close()
        mol.set_op_terms(())
        assert torch.equal(again, mol.encode(x)), "clearing the table restores the three-term result bit for bit"
    finally:
        mol.close()
        pln.close()
        st.close()
    _report("two_term_tables", rec)
    for name, r in rec.items():
        assert r["flips"] == 0, (name, r)
        if name.startswith("fp16x3m"):
            assert r["logp_max_err"] < 5e-4 and r["logit_max_err"] < 5e-4, (name, r["logp_max_err"], r["logit_max_err"])


def test_default_mode_on_images_beyond_the_fixtures_vs_the_oracle(synth_ckpt):
    """The fixtures are 38 + 16 images; a gate met on them is a statement about them (fp16x3m met its 5e-4 gate on the fixtures
    and reaches 8.7e-4 .. 1.2e-3 on further images: profiles/r06_extended_parity_*.json). So every run of the suite also asks the
    from-pixels questions of images NO fixture holds, against the CPU oracle (bit-equal to the reference on every fixture): one
    reference batch per checkpoint (32 images; 16 of the hostile one, whose rows all run to 480 tokens) in the DEFAULT operand mode — free-running rows exact, 0 argmax flips along the
    oracle's ids, log-probs within 2e-4 and the raw logits of EVERY step within 5e-4 (measured 2.5e-4 over 192 images: half of
    north_star's 1e-3 is the bound a default has to hold with room to spare)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import extended_parity as XP
    from molnextr_amd.engine import DEFAULT_DTYPE, Engine
    assert DEFAULT_DTYPE == "fp16x3"
    recs = {}
    for name, ck, first, rows in (("0", synth_ckpt, 5000, 32), ("stress", W.synthetic_checkpoint(1, stress=True), 6000, 16)):
        eng = Engine(ck["encoder"], ck["decoder"], device=0, max_batch=32, dec_slots=64)
        try:
            r = XP.check(eng, ck, first, 1, rows=rows, verbose=False)
        finally:
            eng.close()
        recs[name] = {k: v for k, v in r.items() if k != "batches"}
        assert r["rows_exact_free_running"] == rows and r["flips"] == 0, (name, r)
        assert r["logit_max_err_all_steps"] < 5e-4 and r["logp_max_err"] < 2e-4, (name, r["logit_max_err_all_steps"], r["logp_max_err"])
        assert r["feature_max_err"] < 1e-4, (name, r["feature_max_err"])
    _report("default_mode_beyond_fixtures", recs)
