"""Graph -> SMILES / molfile post-processing (reference MolNexTR/chemical.py:880-975) — host side, RDKit-bound.

Out of scope for the device path. RDKit is not installed in this environment, so only the
first-pass conversion is provided when RDKit happens to be importable (atoms + bonds -> canonical SMILES, without the
reference's abbreviation expansion and wedge-based chirality repair); otherwise the SMILES fields are None and the
caller still receives atoms, bonds and the decoder's raw token SMILES. This is documented as a gap in DESIGN.md.
"""
from typing import List, Tuple

try:  # pragma: no cover - rdkit is absent in the build container
    from rdkit import Chem
    _HAVE_RDKIT = True
except Exception:  # noqa: BLE001
    Chem = None
    _HAVE_RDKIT = False


def have_rdkit() -> bool:
    return _HAVE_RDKIT


def _graph_to_smiles(coords, symbols, edges) -> Tuple[str, str, bool]:  # pragma: no cover
    mol = Chem.RWMol()
    n = len(symbols)
    ids = []
    for sym in symbols:
        s = sym[1:-1] if sym[0] == "[" else sym
        atom = None
        try:
            m = Chem.MolFromSmiles(sym if sym[0] == "[" else f"[{s}]" if len(s) > 2 else s)
            if m is not None and m.GetNumAtoms() == 1:
                atom = Chem.Atom(m.GetAtomWithIdx(0).GetSymbol())
        except Exception:  # noqa: BLE001
            atom = None
        if atom is None:
            atom = Chem.Atom("*")
            atom.SetProp("molFileAlias", s)
        ids.append(mol.AddAtom(atom))
    order = {1: Chem.BondType.SINGLE, 2: Chem.BondType.DOUBLE, 3: Chem.BondType.TRIPLE, 4: Chem.BondType.AROMATIC,
             5: Chem.BondType.SINGLE, 6: Chem.BondType.SINGLE}
    for i in range(n):
        for j in range(i + 1, n):
            if edges[i][j] in order:
                mol.AddBond(ids[i], ids[j], order[edges[i][j]])
    try:
        smiles = Chem.MolToSmiles(mol)
        block = Chem.MolToMolBlock(Chem.MolFromSmiles(smiles))
        return smiles, block, True
    except Exception:  # noqa: BLE001
        return "<invalid>", "", False


def convert_graph_to_smiles(coords: List, symbols: List, edges: List, images=None, num_workers: int = 16):
    """Same signature and return triple as the reference (chemical.py:960-975)."""
    if not _HAVE_RDKIT:
        n = len(symbols)
        return [None] * n, [None] * n, [False] * n
    out = [_graph_to_smiles(c, s, e) for c, s, e in zip(coords, symbols, edges)]
    return [o[0] for o in out], [o[1] for o in out], [o[2] for o in out]
