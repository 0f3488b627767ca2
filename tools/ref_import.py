"""Container-only helper: import the reference's hot-path modules from /root/reference.

Used by tools/gen_golden.py to generate fixtures. /root/reference does not exist
on the GPU box; nothing under tests/, bench.py or the product imports this file.
"""
import argparse
import os
import sys
import types
import warnings

REF_ROOT = "/root/reference"


def have_reference():
    return os.path.isdir(os.path.join(REF_ROOT, "MolNexTR"))


def import_reference():
    """Returns the namespace-package `MolNexTR` bound to the reference tree (read-only)."""
    if "MolNexTR" in sys.modules and getattr(sys.modules["MolNexTR"], "_mnx_ref", False):
        return sys.modules["MolNexTR"]
    shim_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_shims")
    if shim_dir not in sys.path:
        sys.path.insert(0, shim_dir)
    sys.dont_write_bytecode = True  # /root/reference is read-only
    warnings.filterwarnings("ignore")
    pkg = types.ModuleType("MolNexTR")
    pkg.__path__ = [os.path.join(REF_ROOT, "MolNexTR")]  # skip MolNexTR/__init__.py (needs cv2/rdkit/pystow)
    pkg._mnx_ref = True
    sys.modules["MolNexTR"] = pkg
    return pkg


def reference_args(**over):
    """Effective inference config of the reference (MolNexTR/model.py:50-81, exps/eval.sh:19-22)."""
    a = argparse.Namespace(
        encoder="swin_base", decoder="transformer", use_checkpoint=True, dropout=0.5, embed_dim=256,
        enc_pos_emb=False, dec_num_layers=6, dec_hidden_size=256, dec_attn_heads=8, dec_num_queries=128,
        hidden_dropout=0.1, attn_dropout=0.1, max_relative_positions=0, continuous_coords=False,
        compute_confidence=False, input_size=384, vocab_file=None, coord_bins=64, sep_xy=True,
        formats=["chartok_coords", "edges"])
    for k, v in over.items():
        setattr(a, k, v)
    return a
