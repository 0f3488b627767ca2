"""CPU: host tokenizer (molnextr_amd/tokenizer.py) against golden outputs of the reference's CharTokenizer."""
import json
import os

from molnextr_amd.tokenizer import CharTokenizer, get_tokenizer, EOS_ID


def _gold(golden_dir):
    with open(os.path.join(golden_dir, "tokenizer.json")) as f:
        return json.load(f)


def test_sizes(golden_dir):
    g = _gold(golden_dir)
    t = get_tokenizer()["chartok_coords"]
    assert len(t) == g["vocab_size"] == 229 and t.offset == g["offset"] == 101
    assert t.output_constraint


def test_output_mask_truth_table(golden_dir):
    g = _gold(golden_dir)
    t = CharTokenizer(64)
    for i, row in enumerate(g["masks"]):
        assert "".join("1" if v else "0" for v in t.get_output_mask(i)) == row, f"prev id {i}"


def test_sequence_to_smiles_cases(golden_dir):
    g = _gold(golden_dir)
    t = CharTokenizer(64)
    assert len(g["cases"]) >= 15
    for c in g["cases"]:
        assert t.sequence_to_smiles(c["ids"]) == c["out"], c["ids"][:20]


def test_coordinates_round_trip():
    t = CharTokenizer(64)
    for b in range(64):
        assert t.x_to_id(t.id_to_x(101 + b)) == 101 + b and t.y_to_id(t.id_to_y(165 + b)) == 165 + b
    assert t.id_to_x(110) == 9 / 63 and t.id_to_y(187) == 22 / 63       # README atom coords 0.143, 0.349
    out = t.sequence_to_smiles([t.stoi["C"], 110, 187, EOS_ID])
    assert out["symbols"] == ["C"] and out["indices"] == [3] and out["coords"] == [[9 / 63, 22 / 63]]
