"""CPU: the symbol classification in front of the RDKit graph builder (molnextr_amd/chem.py) — the part of the graph ->
SMILES step that can be checked without RDKit. Reference order of tests: MolNexTR/chemical.py:886-898."""
from molnextr_amd import chem


def test_shorthand_that_parses_as_an_element_is_not_an_element():
    # every one of these is a valid element symbol for RDKit, and shorthand for the reference
    for s in ("[Ac]", "[Ts]", "[Pr]", "Ac", "Ts", "Pr"):
        assert chem.classify_symbol(s) == "abbreviation", s
    for s in ("[Ar]", "[Y]", "[Ra]", "[Rb]", "[Rf]", "Ar", "Y", "R", "R1", "[R12]", "X", "[R']"):
        assert chem.classify_symbol(s) == "rgroup", s


def test_real_atoms_and_common_groups():
    for s in ("C", "N", "O", "Cl", "Br", "[NH3+]", "[C@@H]", "[O-]", "[13C]", "[Si]"):
        assert chem.classify_symbol(s) == "atom", s
    for s in ("Ph", "OMe", "[OMe]", "Boc", "NO2", "CO2Et", "[Me]", "Et", "tBu"):
        assert chem.classify_symbol(s) == "abbreviation", s


def test_tables_are_the_reference_tables():
    assert len(chem.RGROUP_SYMBOLS) >= 50 and len(chem.ABBREVIATIONS) >= 200
    assert not (chem.RGROUP_SYMBOLS & {"C", "N", "O"})


def test_without_rdkit_results_are_none_not_guesses():
    if chem.have_rdkit():
        import pytest
        pytest.skip("RDKit present")
    s, m, r = chem.convert_graph_to_smiles([[[0, 0]]], [["C"]], [[[0]]])
    assert s == [None] and m == [None] and r == 0.0
