"""CPU: the evaluation harness' host logic (sharding like DistributedSampler, reference batches, CSV serialisation)."""
import json

import pytest
import torch
from torch.utils.data.distributed import DistributedSampler

from molnextr_amd import evaluate as E


@pytest.mark.parametrize("n,world", [(10, 1), (10, 2), (10, 4), (7, 3), (1, 4), (64, 8), (3, 8)])
def test_sampler_indices_equal_torch_distributed_sampler(n, world):
    """The reference shards its test set with DistributedSampler(shuffle=False) (main.py:440-441)."""
    data = list(range(n))
    for rank in range(world):
        ref = list(DistributedSampler(data, num_replicas=world, rank=rank, shuffle=False))
        assert E.sampler_indices(n, rank, world) == ref


def test_reference_batches_are_twice_the_batch_size():
    b = E.reference_batches(list(range(0, 40, 2)), batch_size=4)      # main.py:445 batch_size * 2
    assert [len(x) for x in b] == [8, 8, 4] and b[0][:3] == [0, 2, 4]


def test_field_serialisation_matches_format_df():
    assert E.dumps_field([[0.123456, 1.0], [0.5, 0.25]]) == "[[0.123,1.0],[0.5,0.25]]"     # utils.py:145-163
    assert E.dumps_field(["C", "[OH]"]) == '["C","[OH]"]'
    assert E.dumps_field([[0, 1], [1, 0]]) == "[[0,1],[1,0]]"
    assert E.dumps_field(None) is None


def test_prediction_table_and_csv(tmp_path):
    preds = {i: {"chartok_coords": {"smiles": "CC", "symbols": ["C", "C"], "coords": [[0.0, 0.1], [1 / 3, 0.9]],
                                    "indices": [3, 6]}, "edges": [[0, 1], [1, 0]]} for i in range(3)}
    table = E.predictions_table(["a", "b", "c"], preds)
    assert table["node_coords"][0] == "[[0.0,0.1],[0.333,0.9]]" and table["edges"][1] == "[[0,1],[1,0]]"
    scores = E.smiles_scores(["CC", "CO", "CC"], table["SMILES"])
    assert scores["raw_string_match"] == pytest.approx(2 / 3)
    out = E.write_predictions(str(tmp_path), "real/acs.csv", table, scores)
    import pandas as pd
    df = pd.read_csv(out)
    assert out.endswith("prediction_acs.csv") and list(df["image_id"]) == ["a", "b", "c"]
    assert json.loads(df["node_symbols"][0]) == ["C", "C"]
    assert json.load(open(tmp_path / "eval_scores_acs_best.json"))["raw_string_match"] == pytest.approx(2 / 3)
