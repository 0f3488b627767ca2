"""Evaluation harness around the device path: the reference's `inference()` / `valid_fn()` driver
(reference main.py:259-303, 429-532) and the prediction CSV that `evaluate.py` consumes (reference evaluate.py:157-218,
MolNexTR/utils.py:145-163).

What is mirrored:
  * sharding: `DistributedSampler(dataset, shuffle=False)` (main.py:440-441) — rank r takes indices r, r+W, r+2W, ...
    of the list padded (by wrapping around) to a multiple of W; per-rank batches of `batch_size * 2` (main.py:445).
    These batches ARE part of the parity contract (the decoder's positional encoding depends on the row inside the
    batch), so the harness hands exactly them to the engine as reference batches;
  * gather: fixed-size records through one all-gather (molnextr_amd/shard.py) instead of `all_gather_object`
    (main.py:295-301); padded duplicates overwrite themselves, as in the reference;
  * output: `prediction_<file>.csv` with image_id, SMILES, node_coords, node_symbols, edges (+ graph_SMILES /
    post_SMILES when RDKit is importable), lists serialised like `format_df` (3-decimal floats, no spaces), and
    `eval_scores_<file>.json`.
Scores: with RDKit the reference's canonicalised exact-match family; without it (this image) only a raw string match
is reported and labelled as such.

    python -m molnextr_amd.evaluate --data_path data --test_file real/acs.csv --save_path out --batch_size 4 \\
           [--load_path ckpt.pth] (under torchrun for several GPUs)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

from . import shard
from .tokenizer import get_tokenizer


def sampler_indices(n: int, rank: int, world: int) -> List[int]:
    """torch DistributedSampler(shuffle=False, drop_last=False): pad by wrapping around, then stride by world."""
    if n == 0:
        return []
    per = -(-n // world)
    total = per * world
    idx = list(range(n))
    pad = total - n
    if pad:
        idx += (idx * (-(-pad // n)))[:pad]
    return idx[rank:total:world]


def reference_batches(indices: Sequence[int], batch_size: int) -> List[List[int]]:
    """DataLoader(batch_size=batch_size * 2, drop_last=False) over the rank's indices (main.py:443-451)."""
    step = batch_size * 2
    return [list(indices[i:i + step]) for i in range(0, len(indices), step)]


def round_floats(o):
    if isinstance(o, float):
        return round(o, 3)
    if isinstance(o, dict):
        return {k: round_floats(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [round_floats(x) for x in o]
    return o


def dumps_field(obj) -> Optional[str]:
    """One cell of node_coords / node_symbols / edges, as `format_df` writes it (utils.py:155-163)."""
    if obj is None:
        return None
    return json.dumps(round_floats(obj)).replace(" ", "")


PAD_TO_SQUARE_FILES = ("real/acs.csv", "real/UOB.csv")     # reference dataset.py:163-164


class PeerFailed(RuntimeError):
    """Raised by run_inference on the ranks that were fine when ANOTHER rank failed with an error that has no fallback
    (capacity, I/O, a HIP error): every rank leaves together instead of waiting in the gather for a peer that is gone."""


class RangeFallback(RuntimeError):
    """Raised by run_inference on EVERY rank when any rank's encoder left the fp16 range of the operand mode: the caller
    rebuilds its engine in the fallback mode (engine.RANGE_FALLBACK) and repeats the whole evaluation."""


def run_inference(engine, load_image: Callable[[int], np.ndarray], n_items: int, batch_size: int, rank: int = 0,
                  world: int = 1, tokenizer=None, group: int = 512, pad_to_square: bool = False) -> Dict[int, dict]:
    """valid_fn for this rank's shard, then the gather: returns {dataset index: prediction dict} on every rank
    (the reference keeps it on all ranks too, main.py:295-301). `engine`: molnextr_amd.engine.Engine.
    pad_to_square: the PadToSquare step `get_transforms` inserts for the test files in PAD_TO_SQUARE_FILES."""
    tok = (tokenizer or get_tokenizer())["chartok_coords"]
    ref_batch = batch_size * 2
    if ref_batch > engine.ROWS_PER_DECODE:
        raise ValueError(f"per-rank batches of {ref_batch} exceed the engine's reference-batch capacity "
                         f"({engine.ROWS_PER_DECODE}); use --batch_size <= {engine.ROWS_PER_DECODE // 2}")
    mine = sampler_indices(n_items, rank, world)
    kmax = engine.max_atoms
    recs = []
    step = max(group // ref_batch, 1) * ref_batch          # whole reference batches per engine call
    dev = getattr(engine, "torch_device", None) or torch.device("cuda", engine.device)
    # An activation beyond the fp16 range of the operand mode ends THIS rank's shard with MNX_ERR_RANGE. Every rank must then
    # repeat its shard in the fallback mode — one predictions table must not mix operand modes, and a rank that restarts
    # alone would leave its peers waiting in the gather —, so the ranks agree on "somebody saw a range error" (one scalar
    # all-reduce MAX) BEFORE the gather and raise together.
    # ANY other failure of one rank (capacity, a missing image file, a HIP error) must not leave its peers in that collective
    # either: the ranks exchange a status code — 0 ok, 1 range error, 2 fatal — and the fatal rank re-raises its own error
    # after the exchange, the others raise PeerFailed.
    from .engine import MnxError, range_fallback_dtype
    range_err, fatal = None, None
    try:
        for g0 in range(0, len(mine), step):
            ids = mine[g0:g0 + step]
            x = engine.preprocess([load_image(i) for i in ids], pad_to_square=pad_to_square)
            out = engine.predict(x, ref_batch=ref_batch)
            recs.append(shard.pack_records_device(out["tokens"], out["lengths"], out["atom_idx"], out["n_atoms"],
                                                  out["edges"]))
    except MnxError as err:
        if range_fallback_dtype(err, getattr(engine, "dtype", None)) is None:
            fatal = err
        else:
            range_err = err
    except BaseException as err:  # noqa: BLE001 - re-raised below, after the peers have been told
        fatal = err
    mine_status = 2 if fatal is not None else (1 if range_err is not None else 0)
    status = shard.max_over_ranks(mine_status, dev) if world > 1 else mine_status
    if fatal is not None:
        raise fatal
    if status == 2:
        raise PeerFailed("another rank failed during inference; this rank leaves before the gather")
    if status == 1:
        raise RangeFallback(str(range_err) if range_err is not None else "another rank's encoder left the fp16 operand range")
    rec = torch.cat(recs) if recs else torch.zeros(0, shard.record_words(kmax), dtype=torch.int32, device=dev)
    index = torch.tensor(mine, dtype=torch.int32, device=dev).view(-1, 1)
    rec = torch.cat([index, rec], dim=1).contiguous()      # the dataset index travels with its record
    if world > 1:
        rec = shard.gather_records(rec)
    rec = rec.cpu().numpy()
    preds: Dict[int, dict] = {}
    rows = shard.unpack_records(torch.from_numpy(np.ascontiguousarray(rec[:, 1:])), kmax)
    for r, row in enumerate(rows):
        preds[int(rec[r, 0])] = {"chartok_coords": tok.sequence_to_smiles(row["tokens"]), "edges": row["edges"]}
    return preds


def predictions_table(image_ids: Sequence, preds: Dict[int, dict]) -> Dict[str, list]:
    """Columns of the reference's pred_df (main.py:466-487), already serialised with `format_df`."""
    from .chem import convert_graph_to_smiles, have_rdkit
    rows = [preds[i] for i in range(len(image_ids))]
    coords = [p["chartok_coords"]["coords"] for p in rows]
    symbols = [p["chartok_coords"]["symbols"] for p in rows]
    edges = [p["edges"] for p in rows]
    table = {"image_id": list(image_ids), "SMILES": [p["chartok_coords"]["smiles"] for p in rows],
             "node_coords": [dumps_field(c) for c in coords], "node_symbols": [dumps_field(s) for s in symbols],
             "edges": [dumps_field(e) for e in edges]}
    if have_rdkit():
        table["graph_SMILES"] = convert_graph_to_smiles(coords, symbols, edges)[0]
    return table


def smiles_scores(gold: Sequence[str], pred: Sequence[str]) -> Dict[str, float]:
    """evaluate.py's SmilesEvaluator needs RDKit canonicalisation; without RDKit only the raw string match exists."""
    from .chem import have_rdkit
    gold, pred = list(gold), list(pred)
    out = {"raw_string_match": float(np.mean([g == p for g, p in zip(gold, pred)])) if gold else 0.0}
    if have_rdkit():  # pragma: no cover - rdkit is absent in the build image
        from rdkit import Chem

        def canon(s, chiral):
            try:
                m = Chem.MolFromSmiles(s)
                return Chem.MolToSmiles(m, isomericSmiles=chiral) if m is not None else ""
            except Exception:  # noqa: BLE001
                return ""
        g1 = [canon(s, True) or "<empty>" for s in gold]
        out["canon_smiles"] = float(np.mean([a == canon(b, True) for a, b in zip(g1, pred)]))
        g0 = [canon(s, False) or "<empty>" for s in gold]
        out["graph"] = float(np.mean([a == canon(b, False) for a, b in zip(g0, pred)]))
    return out


def write_predictions(save_path: str, file_name: str, table: Dict[str, list], scores: Optional[dict] = None,
                      tag: str = "best") -> str:
    import pandas as pd
    os.makedirs(save_path, exist_ok=True)
    base = os.path.basename(file_name)
    out_csv = os.path.join(save_path, f"prediction_{base}")
    pd.DataFrame(table).to_csv(out_csv, index=False)
    if scores is not None:
        with open(os.path.join(save_path, f"eval_scores_{os.path.splitext(base)[0]}_{tag}.json"), "w") as f:
            json.dump(scores, f)
    return out_csv


def main(argv=None):
    import pandas as pd
    import torch.distributed as dist
    from . import weights as W
    from .engine import DEFAULT_DTYPE, DTYPES, Engine
    from .preprocess import load_image_rgb
    ap = argparse.ArgumentParser(description="MolNexTR test-set inference on MI355X (reference main.py --do_test)")
    ap.add_argument("--data_path", default=".")
    ap.add_argument("--test_file", required=True, help="CSV with file_path (and SMILES / image_id) columns")
    ap.add_argument("--save_path", default="predict_output")
    ap.add_argument("--load_path", required=True,
                    help="checkpoint: reference .pth or .safetensors; 'synthetic' = deterministic test weights")
    ap.add_argument("--batch_size", type=int, default=4, help="per-GPU batch size; inference uses twice that")
    ap.add_argument("--dtype", default=None, choices=sorted(DTYPES),
                    help="encoder operand mode; default engine.DEFAULT_DTYPE = fp16x3 (every token / atom / bond as the reference's fp32 path; "
                         "fp16x3m = qkv / fc1 / fc2 of Swin stage 3 on two product terms: faster, tokens exact on everything measured, raw logits up to "
                         "1.2e-3 off on a hostile checkpoint)")
    ap.add_argument("--fp16", action="store_true",
                    help="the reference's flag (main.py:40, exps/eval.sh: fp16 autocast): without --dtype it selects the one-plane "
                         "fp16 operand mode, which stays closer to the fp32 result than the reference's autocast path does "
                         "(tests/test_gpu_pixels.py, tests/golden/pixels_autocast_fp16.json)")
    args = ap.parse_args(argv)
    dtype = args.dtype or ("fp16" if args.fp16 else DEFAULT_DTYPE)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    from .checkpoint import load_checkpoint               # strict validation, no optimizer state, safetensors-aware
    states = W.synthetic_checkpoint(0) if args.load_path == "synthetic" else load_checkpoint(args.load_path)
    engine = Engine(states["encoder"], states["decoder"], device=local, max_batch=64, dtype=dtype)
    df = pd.read_csv(os.path.join(args.data_path, args.test_file))
    paths = [os.path.join(args.data_path, p) for p in df["file_path"]]
    def infer(e):
        return run_inference(e, lambda i: load_image_rgb(paths[i]), len(df), args.batch_size, rank, world,
                             pad_to_square=args.test_file in PAD_TO_SQUARE_FILES)
    try:
        preds = infer(engine)
    except RangeFallback as err:
        # the reference evaluates any checkpoint; an activation beyond the fp16 range must not end the run. All ranks arrive
        # here together (run_inference agrees on the flag before its gather): one table, one operand mode.
        from .engine import RANGE_FALLBACK
        to = RANGE_FALLBACK.get(dtype)
        if to is None:
            raise
        print(f"[rank {rank}] {err}: repeating the evaluation with --dtype {to} on every rank", file=sys.stderr, flush=True)
        engine.close()
        engine = Engine(states["encoder"], states["decoder"], device=local, max_batch=64, dtype=to)
        preds = infer(engine)
    if rank == 0:
        if "image_id" not in df.columns:    # main.py:461-462
            df["image_id"] = [p.split("/")[-1].split(".")[0] for p in df["file_path"]]
        table = predictions_table(df["image_id"], preds)
        scores = smiles_scores(df["SMILES"], table["SMILES"]) if "SMILES" in df.columns else None
        print(write_predictions(args.save_path, args.test_file, table, scores), json.dumps(scores))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
