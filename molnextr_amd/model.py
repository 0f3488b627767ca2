"""Model facade: mirrors the reference's `molnextr` class and `Decoder.decode` on top of the HIP engine.

  decode_batch(...)        <-> Decoder.decode                       reference MolNexTR/components.py:443-492
  molnextr.predict_images  <-> molnextr.predict_images              reference MolNexTR/model.py:97-146
  molnextr.predict_image / predict_image_files / predict_final_results   reference MolNexTR/model.py:148-196

Same method names, argument meaning and output dict keys. Everything between "normalised image batch" and "token
ids / hidden states / bond matrix" runs in libmolnextr_hip.so; tokens are turned into symbols / coordinates /
atom positions on the host (tokenizer.py), as in the reference. No CPU fallback exists for the device part.
"""
from __future__ import annotations

import argparse
from typing import List, Optional

import numpy as np
import torch

from . import weights as W
from .engine import DEFAULT_DTYPE, Engine, MnxError
from .preprocess import load_image_rgb, transform_image
from .tokenizer import get_tokenizer

BOND_TYPES = ["", "single", "double", "triple", "aromatic", "solid wedge", "dashed wedge"]  # reference model.py:30
ROWS = Engine.ROWS_PER_DECODE


def decode_batch(engine: Engine, features: torch.Tensor, tokenizer=None, ref_batch_size: Optional[int] = None,
                 compute_confidence: bool = False, max_len: Optional[int] = None, beam_size: int = 1,
                 n_best: int = 1) -> List[dict]:
    """Decoder.decode for formats ['chartok_coords', 'edges'] (reference components.py:443-492).

    features [B,144,1024] on the GPU. `ref_batch_size`: rows are numbered as if the reference had decoded them in
    consecutive batches of this size (its positional encoding is indexed by the row inside the batch and finished
    rows are compacted away, so results depend on the batch composition); None = one batch of B (B <= 32) or
    batches of 32. Returns one dict per image: {'chartok_coords': {smiles, symbols, coords, indices[, atom_scores]},
    'edges': [[...]] [, 'edge_scores', 'overall_score']}.

    beam_size > 1 (reference signature components.py:443; its own beam branch cannot run): the
    best hypothesis of each image is detokenised, as the reference does with `pred[0]` (components.py:453-455), and
    the bond head runs on the decoder outputs along that hypothesis; `beam_scores` lists the n_best average
    log-probs. Token confidences are not tracked by beam search (nor by the reference's BeamSearch).
    """
    tok = (tokenizer or get_tokenizer())["chartok_coords"]
    B = features.shape[0]
    rbs = ref_batch_size or ROWS
    if rbs > ROWS:
        raise ValueError(f"reference batches larger than {ROWS} rows are not supported by one engine call")
    group = (ROWS // rbs) * rbs          # rows per engine call: whole reference batches only
    if beam_size > 1:
        if compute_confidence:
            raise NotImplementedError("beam search does not track token scores (neither does the reference's)")
        group = rbs                      # one reference batch per beam call
    preds: List[dict] = []
    for g0 in range(0, B, group):
        feats = features[g0:g0 + group].contiguous()
        n = feats.shape[0]
        beam_scores = None
        if beam_size > 1:
            bo = engine.decode_beam(feats, beam=beam_size, n_best=n_best, max_len=max_len)
            out = {"lengths": bo["lengths"][:, 0].contiguous(), "tokens": bo["tokens"][:, 0].contiguous(),
                   "hidden": bo["hidden"][:, 0].contiguous()}
            beam_scores = bo["scores"].cpu().numpy()
        else:
            chunk = torch.arange(n, dtype=torch.int32) // rbs
            out = engine.decode_greedy(feats, chunk_id=chunk, max_len=max_len, want_logp=True)
        lens = out["lengths"].cpu().numpy()
        toks = out["tokens"].cpu().numpy()
        logp = out["token_logp"].cpu().numpy() if compute_confidence else None
        rows = [tok.sequence_to_smiles(toks[b, :lens[b]].tolist()) for b in range(n)]
        kmax = engine.max_atoms
        n_atoms = np.array([len(r["indices"]) for r in rows], dtype=np.int32)
        if n_atoms.max(initial=0) > kmax:
            raise RuntimeError(f"{int(n_atoms.max())} atoms exceed the engine capacity max_atoms={kmax}")
        atom_idx = np.zeros((n, kmax), dtype=np.int32)
        for b, r in enumerate(rows):
            atom_idx[b, :n_atoms[b]] = r["indices"]
        edges, scores = engine.edges(out["hidden"], torch.from_numpy(atom_idx), torch.from_numpy(n_atoms),
                                     want_scores=compute_confidence)
        edges = edges.cpu().numpy()
        scores = scores.cpu().numpy() if scores is not None else None
        for b, r in enumerate(rows):
            k = int(n_atoms[b])
            pred = {"chartok_coords": r, "edges": edges[b, :k, :k].astype(int).tolist()}
            if beam_scores is not None:
                pred["beam_scores"] = beam_scores[b].tolist()
            if compute_confidence:   # reference components.py:456-469, 485-491
                ts = np.exp(logp[b, :lens[b]].astype(np.float64))
                idx = np.array(r["indices"]) - 3
                r["atom_scores"] = [float(np.prod(ts[i - len(s) + 1:i + 1]) ** (1 / len(s)))
                                    for s, i in zip(r["symbols"], idx)]
                avg = float(np.exp(np.mean(logp[b, :lens[b]].astype(np.float64))))
                es = scores[b, :k, :k]
                pred["edge_scores"] = es.tolist()
                pred["overall_score"] = avg * float(np.sqrt(np.prod(es)))
            preds.append(pred)
    return preds


def predict_pipeline(engine: Engine, images: torch.Tensor, tokenizer=None, ref_batch_size: int = 16,
                     max_len: Optional[int] = None, beam_size: int = 1) -> List[dict]:
    """Encoder + Decoder.decode for MANY images through the engine's continuous-batching path (mnx_predict):
    same per-image dicts as `decode_batch`, identical results (the on-device atom scan equals
    sequence_to_smiles' indices), much higher throughput. Confidences are not available on this path.
    beam_size > 1: mnx_predict_beam (best hypothesis per image, 'beam_scores' = [its average log-prob])."""
    tok = (tokenizer or get_tokenizer())["chartok_coords"]
    out = engine.predict(images, ref_batch=ref_batch_size, max_len=max_len, beam=beam_size)
    scores = out["scores"].cpu().numpy() if beam_size > 1 else None
    lens = out["lengths"].cpu().numpy()
    toks = out["tokens"].cpu().numpy()
    n_atoms = out["n_atoms"].cpu().numpy()
    edges = out["edges"].cpu().numpy()
    preds = []
    for b in range(len(lens)):
        r = tok.sequence_to_smiles(toks[b, :lens[b]].tolist())
        k = int(n_atoms[b])
        assert k == len(r["indices"]), "device atom scan disagrees with the host tokenizer"
        preds.append({"chartok_coords": r, "edges": edges[b, :k, :k].astype(int).tolist()})
        if scores is not None:
            preds[-1]["beam_scores"] = [float(scores[b])]
    return preds


class _RestartCall(Exception):
    """Private: the facade's engine was rebuilt in the range-fallback mode after part of a call had been computed."""


class molnextr:
    """Main interface (reference MolNexTR/model.py:33-196).

    model_path: a checkpoint in the reference's format ({'encoder','decoder','args'}, `.pth`) or our `.safetensors`;
    the literal 'synthetic' opts into the deterministic hash-generated checkpoint (tests / bench only: its predictions
    are meaningless as chemistry). There is no default: like the reference, the model cannot run without weights.
    device: torch.device('cuda', i) — an MI355X is required.
    dtype: encoder operand mode. 'fp16x3' (the default, engine.DEFAULT_DTYPE: split fp16 operands, three MFMA terms per product:
    features equal the reference's to fp32 rounding level, logits within 2.5e-4), 'fp16x3m' (opt-in: the Linear layers of
    engine.FP16X3M_TWO_TERM on two terms, +8-11 % throughput; every token / atom / bond still the reference's on everything
    measured, raw logits within 5e-4 on the fixtures, 7.2e-4 on further images and up to 1.2e-3 on a hostile checkpoint — at and
    beyond north_star's 1e-3),
    'bf16x3' (three terms with the fp32 exponent range), 'fp32' (exact-fp32 MFMA, slowest), 'bf16' / 'fp16' (fastest; argmax
    decisions near a tie can differ)."""

    def __init__(self, model_path, device=None, max_batch: int = 32, dtype: str = DEFAULT_DTYPE,
                 device_preprocess: bool = True):
        if model_path is None:
            raise ValueError("molnextr(model_path): a checkpoint path is required (pass 'synthetic' explicitly for the "
                             "deterministic test checkpoint)")
        if model_path == "synthetic":
            states = W.synthetic_checkpoint(0)
        else:
            from .checkpoint import load_checkpoint      # .pth in the reference format or our .safetensors; strict
            states = load_checkpoint(model_path)
        args = self._get_args(states.get("args"))
        if device is None:
            device = torch.device("cuda", 0)
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("molnextr_amd runs the model on an MI355X only (no CPU path); pass device='cuda:N'")
        self.device = device
        self.args = args
        self.tokenizer = get_tokenizer(args)
        self._states, self._max_batch = states, max_batch      # kept for the operand-range fallback (_with_fallback)
        self.engine = Engine(states["encoder"], states["decoder"], device=device.index or 0, max_batch=max_batch,
                             dtype=dtype)
        self.input_size = args.input_size
        self.device_preprocess = device_preprocess
        self.group_images = 1024          # images per engine call of the throughput path (whole reference batches)

    _groups_done = 0          # groups of the running predict_images call that have produced predictions

    def _with_fallback(self, job):
        """job(engine) -> result. When the engine reports that an activation left the fp16 range of its operand mode
        (MNX_ERR_RANGE: possible with fp16x3 / fp16 on checkpoints with very large activations), the engine is rebuilt ONCE
        in the corresponding bf16 mode (fp32 exponent range) with a warning and the job runs again; the instance keeps the
        new engine. The reference runs such a checkpoint without complaint, so the drop-in must too."""
        from .engine import range_fallback_dtype
        try:
            return job(self.engine)
        except MnxError as e:
            to = range_fallback_dtype(e, self.engine.dtype)
            if to is None:
                raise
            import warnings
            warnings.warn(f"molnextr_amd: an encoder activation left the fp16 range of operand mode '{self.engine.dtype}' "
                          f"({e}); rebuilding the engine with dtype='{to}' and repeating the batch", RuntimeWarning)
            dev = self.engine.device
            self._join_prefetch()       # the helper of group g + 1 may still be inside mnx_preprocess on this handle
            self.engine.close()
            self.engine = Engine(self._states["encoder"], self._states["decoder"], device=dev, max_batch=self._max_batch,
                                 dtype=to)
            if self._groups_done:
                # earlier groups of this predict_images call were computed in the fp16 mode: one call, one operand mode —
                # the call starts again from its first image (predict_images catches this)
                raise _RestartCall()
            return job(self.engine)

    @staticmethod
    def _get_args(args_states=None):
        """Inference defaults of the reference (model.py:50-81) overridden by the checkpoint's saved args."""
        a = argparse.Namespace(encoder="swin_base", decoder="transformer", enc_pos_emb=False, dec_num_layers=6,
                               dec_hidden_size=256, dec_attn_heads=8, continuous_coords=False,
                               compute_confidence=False, input_size=384, vocab_file=None, coord_bins=64, sep_xy=True,
                               formats=["chartok_coords", "edges"])
        for k, v in (args_states or {}).items():
            setattr(a, k, v)
        if a.encoder != "swin_base" or a.input_size != 384 or a.continuous_coords:
            raise NotImplementedError("engine is built for the swin_base / 384 / discrete-coordinate configuration")
        return a

    def _transform(self, images: List, engine=None) -> torch.Tensor:
        """CropWhite + Resize + ToGray + Normalize (reference model.py:104): on the device (mnx_preprocess), or with the
        bit-identical host restatement when `device_preprocess` is off. The result is a torch tensor of integer-exact
        arithmetic: it does not depend on the engine's operand mode and outlives the engine that made it."""
        if self.device_preprocess:
            return (engine or self.engine).preprocess(images)
        return torch.from_numpy(np.stack([transform_image(im, self.input_size) for im in images])).to(self.device)

    _prefetch_thread = None       # the helper thread of the running _prefetched generator, if one is in flight

    def _join_prefetch(self):
        """Waits for the prefetch helper (if any). Called before the engine it works on is closed (_with_fallback)."""
        t = self._prefetch_thread
        if t is not None:
            t.join()

    def _side_context(self):
        """The context the prefetch helper runs in: this device, a side stream (a seam for the CPU tests)."""
        import contextlib
        stack = contextlib.ExitStack()
        stack.enter_context(torch.cuda.device(self.device))
        stack.enter_context(torch.cuda.stream(torch.cuda.Stream(device=self.device)))
        return stack

    def _prefetched(self, groups: List[List]):
        """Yields the transformed tensor of every group; group g+1 is uploaded (pinned staging -> H2D) and transformed on a
        side stream by a helper thread while the caller runs the engine on group g (reference main.py gets the same
        overlap from DataLoader workers + pin_memory). `mnx_preprocess` is the one entry point that may run beside
        another call on the same handle (include/molnextr_hip.h) — ONE such call: at most one helper exists at a time, it
        works on the engine that was current when it was started, and whoever replaces that engine joins the helper first
        (_with_fallback -> _join_prefetch); closing the generator (a call that is abandoned and restarted) joins it too."""
        if len(groups) <= 1 or not self.device_preprocess:
            for g in groups:
                yield self._transform(g)
            return
        import threading

        def work(g, box, engine):
            try:
                with self._side_context():
                    box.append(self._transform(g, engine))      # Engine.preprocess synchronises the side stream before returning
            except BaseException as e:  # noqa: BLE001 - re-raised in the caller's thread
                box.append(e)

        def start(g):
            box: list = []
            t = threading.Thread(target=work, args=(g, box, self.engine), daemon=True)
            self._prefetch_thread = t
            t.start()
            return t, box

        t, box = start(groups[0])
        try:
            for gi in range(len(groups)):
                t.join()
                item = box.pop()
                if isinstance(item, BaseException):
                    raise item
                if gi + 1 < len(groups):
                    t, box = start(groups[gi + 1])
                yield item
        finally:
            t.join()
            self._prefetch_thread = None

    def predict_images(self, input_images: List, return_atoms_bonds=False, return_confidence=False, batch_size=16):
        if len(input_images) == 0:
            return []                                      # reference model.py:101-102: empty loop, empty list
        try:
            self._groups_done = 0
            preds = self._predict_all(input_images, return_confidence, batch_size)
        except _RestartCall:                               # the engine was rebuilt in the bf16 split mode after some groups
            self._groups_done = 0
            preds = self._predict_all(input_images, return_confidence, batch_size)
        finally:
            self._groups_done = 0
        return self._assemble(preds, input_images, return_atoms_bonds, return_confidence)

    def _predict_all(self, input_images: List, return_confidence: bool, batch_size: int) -> List[dict]:
        """The engine part of predict_images: one prediction dict per image, every group in ONE operand mode."""
        preds: List[dict] = []
        cap = min(ROWS, self.engine.max_batch)
        batch_size = min(batch_size, len(input_images))     # a batch larger than the job is the whole job: same numbering
        if batch_size < 1 or batch_size > cap:
            # results depend on the row inside the reference batch (positional-encoding quirk), so a silently
            # different batch size would silently change tokens
            raise ValueError(f"batch_size must be 1..{cap} (one reference batch per {ROWS}-row decode tile); got {batch_size}")
        if not return_confidence:
            # throughput path: many images per engine call, reference batches of `batch_size` kept as numbering units;
            # the group is a whole number of reference batches so that batch boundaries do not drift between groups
            group = (self.group_images // batch_size) * batch_size
            groups = [input_images[i:i + group] for i in range(0, len(input_images), group)]
            gen = self._prefetched(groups)
            try:
                for x in gen:
                    preds += self._with_fallback(
                        lambda eng: predict_pipeline(eng, x, self.tokenizer, ref_batch_size=batch_size))
                    self._groups_done += 1
            finally:
                if hasattr(gen, "close"):
                    gen.close()         # a call abandoned by _RestartCall leaves no helper thread behind
        else:
            step = max(self.engine.max_batch // batch_size, 1) * batch_size
            for i in range(0, len(input_images), step):
                x = self._transform(input_images[i:i + step])

                def conf_job(eng):
                    feats = eng.encode(x)
                    if eng.encoder_nonfinite():          # fp16 operand range exceeded (mnx_predict reports it by itself)
                        from .engine import MNX_ERR_RANGE
                        raise MnxError("encoder features are not finite: an activation left the fp16 range of the operand "
                                       f"mode '{eng.dtype}'", code=MNX_ERR_RANGE)
                    return decode_batch(eng, feats, self.tokenizer, ref_batch_size=batch_size, compute_confidence=True)
                preds += self._with_fallback(conf_job)
                self._groups_done += 1
        return preds

    def _assemble(self, preds: List[dict], input_images: List, return_atoms_bonds: bool, return_confidence: bool):
        """Output dicts of predict_images (reference model.py:111-196) from the per-image predictions."""
        from .chem import convert_graph_to_smiles
        smiles_list, molblock_list, _ = convert_graph_to_smiles(
            [p["chartok_coords"]["coords"] for p in preds], [p["chartok_coords"]["symbols"] for p in preds],
            [p["edges"] for p in preds], images=input_images)
        outputs = []
        for smiles, molfile, pred in zip(smiles_list, molblock_list, preds):
            d = {"predicted_smiles": smiles, "predicted_molfile": molfile}
            if return_atoms_bonds:
                c = pred["chartok_coords"]
                atoms = []
                for i, (sym, xy) in enumerate(zip(c["symbols"], c["coords"])):
                    a = {"atom_number": f"{i}", "atom_symbol": sym, "coords": (round(xy[0], 3), round(xy[1], 3))}
                    if return_confidence:
                        a["confidence"] = c["atom_scores"][i]
                    atoms.append(a)
                d["atom_sets"] = atoms
                bonds = []
                k = len(c["symbols"])
                for i in range(k - 1):
                    for j in range(i + 1, k):
                        t = pred["edges"][i][j]
                        if t != 0:
                            bd = {"atom_number": f"{i}", "bond_type": BOND_TYPES[t], "endpoints": (i, j)}
                            if return_confidence:
                                bd["confidence"] = pred["edge_scores"][i][j]
                            bonds.append(bd)
                d["bond_sets"] = bonds
            outputs.append(d)
        return outputs

    def predict_image(self, image, return_atoms_bonds=False, return_confidence=False):
        return self.predict_images([image], return_atoms_bonds, return_confidence)[0]

    def predict_image_files(self, image_files: List, return_atoms_bonds=False, return_confidence=False):
        return self.predict_images([load_image_rgb(p) for p in image_files], return_atoms_bonds, return_confidence)

    def predict_final_results(self, image_file: str, return_atoms_bonds=False, return_confidence=False):
        return self.predict_image_files([image_file], return_atoms_bonds, return_confidence)[0]
