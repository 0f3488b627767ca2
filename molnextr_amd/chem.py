"""Graph -> SMILES / molfile post-processing (reference MolNexTR/chemical.py:880-975) — host side, RDKit-bound.

Out of scope for the device path and BLOCKED in this environment: RDKit is not installed and
cannot be installed, so nothing below can be executed or tested here. Behaviour:

  * without RDKit: `predicted_smiles` / `predicted_molfile` are None (logged once). The caller still receives atoms,
    bonds and the decoder's raw token SMILES; BASELINE's "SMILES exact-match" is therefore checkable on the raw token
    SMILES + atom / bond sets only — every result that reports a match says so (evaluate.smiles_scores);
  * with RDKit: the molecule graph is built exactly as `_convert_graph_to_smiles` builds it (chemical.py:880-925:
    the de-bracketed symbol is looked up in the R-group and abbreviation tables FIRST (:886-898 — 'Ac', 'Ts', 'Pr', 'Ar',
    'Y' ... are shorthand there, not actinium / tennessine / praseodymium / argon / yttrium; the tables are data,
    vocab/abbreviations.json), then `Chem.AtomFromSmiles(symbol)` so charges / isotopes / explicit H survive, chiral tag
    cleared, '*' + alias for R-groups / abbreviations / unparsable symbols, BondDir BEGINWEDGE / BEGINDASH for bond
    classes 5 / 6). The
    reference then runs `_verify_chirality` (:212-287) and `_expand_functional_group` (:565-877), ~400 lines of RDKit
    calls that are NOT restated (they cannot be validated here). A molecule that needs either — it has a wedge bond
    or an alias atom — is reported as failed (`None`, success False) instead of a plausible but different SMILES;
    only molecules that need neither get a SMILES.
"""
import json
import logging
import os
from typing import List, Tuple

import numpy as np

try:  # pragma: no cover - rdkit is absent in the build container
    from rdkit import Chem
    _HAVE_RDKIT = True
except Exception:  # noqa: BLE001
    Chem = None
    _HAVE_RDKIT = False

logger = logging.getLogger("molnextr")
_warned = False


def have_rdkit() -> bool:
    return _HAVE_RDKIT


with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "vocab", "abbreviations.json")) as _f:
    _tables = json.load(_f)
RGROUP_SYMBOLS = frozenset(_tables["rgroup_symbols"])       # reference abbrs.py:8-10
ABBREVIATIONS = frozenset(_tables["abbreviations"])         # keys of the reference's ABBREVIATIONS (abbrs.py:218)


def classify_symbol(symbol: str) -> str:
    """'rgroup' | 'abbreviation' | 'atom' for one atom token, in the reference's order of tests (chemical.py:886-898):
    brackets stripped, R-group table, abbreviation table, and only then a chemical element / SMILES atom."""
    inner = symbol[1:-1] if symbol[:1] == "[" and symbol[-1:] == "]" else symbol
    if inner in RGROUP_SYMBOLS:
        return "rgroup"
    if inner in ABBREVIATIONS:
        return "abbreviation"
    return "atom"


def _graph_to_smiles(coords, symbols, edges) -> Tuple[object, object, bool]:  # pragma: no cover - needs RDKit
    mol = Chem.RWMol()
    n = len(symbols)
    needs_unported_pass = False
    for i, sym in enumerate(symbols):
        inner = sym[1:-1] if sym[0] == "[" else sym
        atom = None
        if classify_symbol(sym) == "atom":               # shorthand is never handed to AtomFromSmiles (it would parse)
            try:
                atom = Chem.AtomFromSmiles(sym)
                if atom is not None:
                    atom.SetChiralTag(Chem.rdchem.ChiralType.CHI_UNSPECIFIED)
            except Exception:  # noqa: BLE001
                atom = None
        if atom is None or atom.GetSymbol() == "*":      # R-group, abbreviation or condensed formula
            atom = Chem.Atom("*")
            if inner[:1] == "R" and inner[1:].isdigit():
                atom.SetIsotope(int(inner[1:]))
            Chem.SetAtomAlias(atom, inner)
            atom.SetProp("molFileAlias", inner)
            needs_unported_pass = True                   # _expand_functional_group
        assert mol.AddAtom(atom) == i
    kinds = {1: Chem.BondType.SINGLE, 2: Chem.BondType.DOUBLE, 3: Chem.BondType.TRIPLE, 4: Chem.BondType.AROMATIC,
             5: Chem.BondType.SINGLE, 6: Chem.BondType.SINGLE}
    for i in range(n):
        for j in range(i + 1, n):
            t = edges[i][j]
            if t in kinds:
                mol.AddBond(i, j, kinds[t])
                if t == 5:
                    mol.GetBondBetweenAtoms(i, j).SetBondDir(Chem.BondDir.BEGINWEDGE)
                    needs_unported_pass = True           # _verify_chirality
                elif t == 6:
                    mol.GetBondBetweenAtoms(i, j).SetBondDir(Chem.BondDir.BEGINDASH)
                    needs_unported_pass = True
    if needs_unported_pass:
        return None, None, False
    try:
        smiles = Chem.MolToSmiles(mol, isomericSmiles=True, canonical=True)
        m2 = Chem.MolFromSmiles(smiles)
        return smiles, Chem.MolToMolBlock(m2), True
    except Exception:  # noqa: BLE001
        return "<invalid>", "", False


def convert_graph_to_smiles(coords: List, symbols: List, edges: List, images=None, num_workers: int = 16):
    """Same signature and return triple as the reference (chemical.py:960-975): (smiles list, molblock list, success
    rate). Without RDKit: ([None]*n, [None]*n, 0.0)."""
    global _warned
    n = len(symbols)
    if not _HAVE_RDKIT:
        if not _warned:
            logger.warning("RDKit is not installed: predicted_smiles / predicted_molfile are None; atoms, bonds and the "
                           "raw token SMILES are still returned (molnextr_amd/chem.py)")
            _warned = True
        return [None] * n, [None] * n, 0.0
    out = [_graph_to_smiles(c, s, e) for c, s, e in zip(coords, symbols, edges)]  # pragma: no cover
    if not _warned and any(o[0] is None for o in out):  # pragma: no cover
        logger.warning("molecules with wedge bonds or abbreviations need the reference's _verify_chirality / "
                       "_expand_functional_group, which are not restated: their SMILES are reported as None")
        _warned = True
    return [o[0] for o in out], [o[1] for o in out], float(np.mean([o[2] for o in out])) if out else 0.0  # pragma: no cover
