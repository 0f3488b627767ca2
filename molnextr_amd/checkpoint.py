"""Checkpoint tooling: read the reference's `.pth` ({'encoder','decoder','args', + optimizer /
scheduler / scaler state}, reference main.py:389-398), validate it STRICTLY against the weight contract (the reference
loads with strict=False and silently drops mismatches, MolNexTR/model.py:17-28), and write an inference-only
safetensors file (weights + the saved args as metadata) that loads without unpickling.

    python -m molnextr_amd.checkpoint molnextr_best.pth molnextr_best.safetensors
"""
from __future__ import annotations

import json
import sys
from typing import Dict

import torch

from . import weights as W

_PREFIX = {"encoder": "encoder/", "decoder": "decoder/"}


def _strip_module(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """DistributedDataParallel prefixes every key with 'module.' (reference model.py:19-24 removes it the same way)."""
    return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}


def validate(states: dict, enc: W.EncoderDims = None, dec: W.DecoderDims = None) -> dict:
    enc = enc or W.EncoderDims()
    dec = dec or W.DecoderDims()
    out = {"encoder": _strip_module(states["encoder"]), "decoder": _strip_module(states["decoder"]),
           "args": dict(states.get("args") or {})}
    W.validate_state(out["encoder"], W.encoder_spec(enc), "encoder")
    W.validate_state(out["decoder"], W.decoder_spec(dec), "decoder")
    return out


def load_checkpoint(path: str) -> dict:
    """`.pth` (reference format) or `.safetensors` (ours) -> {'encoder': sd, 'decoder': sd, 'args': dict}, validated."""
    if path.endswith(".safetensors"):
        from safetensors import safe_open
        states = {"encoder": {}, "decoder": {}, "args": {}}
        with safe_open(path, framework="pt", device="cpu") as f:
            meta = f.metadata() or {}
            states["args"] = json.loads(meta.get("args", "{}"))
            for k in f.keys():
                part, name = k.split("/", 1)
                states[part][name] = f.get_tensor(k)
        return validate(states)
    return validate(torch.load(path, map_location="cpu"))


def convert(src: str, dst: str) -> dict:
    """Reference `.pth` -> inference-only safetensors (no optimizer / scheduler state). Returns a size summary."""
    from safetensors.torch import save_file
    states = load_checkpoint(src)
    flat = {}
    for part in ("encoder", "decoder"):
        for k, v in states[part].items():
            flat[_PREFIX[part] + k] = v.contiguous()
    args = {k: v for k, v in states["args"].items() if isinstance(v, (int, float, str, bool, list, type(None)))}
    save_file(flat, dst, metadata={"args": json.dumps(args), "format": "molnextr_amd/1"})
    return {"tensors": len(flat), "parameters": int(sum(v.numel() for v in flat.values() if v.is_floating_point()))}


if __name__ == "__main__":
    if len(sys.argv) != 3:
        raise SystemExit(__doc__)
    print(json.dumps(convert(sys.argv[1], sys.argv[2])))
