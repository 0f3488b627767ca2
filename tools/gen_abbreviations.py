#!/usr/bin/env python3
"""Build container only: writes molnextr_amd/vocab/abbreviations.json — the SYMBOL TABLES (no code) that the graph ->
SMILES step consults before it treats an atom token as a chemical element (reference MolNexTR/chemical.py:886-898 tests
`symbol in RGROUP_SYMBOLS` and `symbol in ABBREVIATIONS` first; 'Ac', 'Ts', 'Pr', 'Ar', 'Y' ... would otherwise parse as
actinium, tennessine, praseodymium, argon, yttrium). Source of the data: MolNexTR/abbrs.py:8-10 (R-group symbols) and the
keys of ABBREVIATIONS (:26-218). Only names are taken, as data, like vocab_chars.json."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("ref_abbrs", "/root/reference/MolNexTR/abbrs.py")
m = importlib.util.module_from_spec(spec)
spec.loader.exec_module(m)
out = {"source": "CYF2000127/MolNexTR MolNexTR/abbrs.py: RGROUP_SYMBOLS and the keys of ABBREVIATIONS (names only)",
       "rgroup_symbols": list(m.RGROUP_SYMBOLS), "abbreviations": sorted(m.ABBREVIATIONS.keys())}
path = os.path.join(ROOT, "molnextr_amd", "vocab", "abbreviations.json")
with open(path, "w") as f:
    json.dump(out, f, indent=0, ensure_ascii=False)
print(path, len(out["rgroup_symbols"]), "R-group symbols,", len(out["abbreviations"]), "abbreviations")
