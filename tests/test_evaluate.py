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


# ---- run_inference's index bookkeeping across ranks (gloo, CPU): padded duplicates, n not divisible -----------------
class _FakeEngine:
    """Stands in for molnextr_amd.engine.Engine on the CPU: 'decodes' image i into the token sequence [5 + i % 90, 2]
    and records the reference batches it was handed (they are part of the parity contract)."""
    ROWS_PER_DECODE = 32
    max_atoms = 8
    torch_device = torch.device("cpu")

    def __init__(self):
        self.batches = []

    def preprocess(self, images, pad_to_square=False):
        return torch.tensor([int(im[0, 0, 0]) * 256 + int(im[0, 0, 1]) for im in images], dtype=torch.int32)

    def predict(self, x, ref_batch=32, max_len=None):
        n = x.shape[0]
        self.batches += [x[i:i + ref_batch].tolist() for i in range(0, n, ref_batch)]
        tokens = torch.zeros(n, 480, dtype=torch.int32)
        tokens[:, 0] = 5 + x % 90
        tokens[:, 1] = 2
        return {"tokens": tokens, "lengths": torch.full((n,), 2, dtype=torch.int32),
                "n_atoms": torch.zeros(n, dtype=torch.int32), "atom_idx": torch.zeros(n, 8, dtype=torch.int32),
                "edges": torch.zeros(n, 8, 8, dtype=torch.uint8)}


def _page(i):
    import numpy as np
    p = np.zeros((2, 2, 3), np.uint8)
    p[0, 0, 0], p[0, 0, 1] = i // 256, i % 256
    return p


def _eval_worker(rank, world, n, port, q):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = _FakeEngine()
    preds = E.run_inference(eng, _page, n, batch_size=2, rank=rank, world=world, group=8)
    q.put((rank, {i: p["chartok_coords"]["smiles"] for i, p in preds.items()}, eng.batches))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 7), (3, 10), (3, 2)])
def test_run_inference_index_bookkeeping_gloo(world, n):
    """Every rank ends up with exactly one prediction per dataset index (the sampler's wrap-around duplicates overwrite
    themselves, as in the reference main.py:295-301), each from ITS image, and the engine saw exactly the reference
    batches DistributedSampler + DataLoader(batch_size*2) would have produced on that rank."""
    import os
    import torch.multiprocessing as mp
    from molnextr_amd.tokenizer import get_tokenizer
    tok = get_tokenizer()["chartok_coords"]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31000 + (os.getpid() * 7 + world * 13 + n) % 2000
    procs = [ctx.Process(target=_eval_worker, args=(r, world, n, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = {i: tok.sequence_to_smiles([5 + i % 90, 2])["smiles"] for i in range(n)}
    for rank, smiles, batches in got:
        assert smiles == want, f"rank {rank}"
        assert batches == E.reference_batches(E.sampler_indices(n, rank, world), batch_size=2), f"rank {rank}"


# ---- the operand-range fallback is collective: one rank's MNX_ERR_RANGE makes EVERY rank raise before the gather --------
class _RangeEngine(_FakeEngine):
    """fp16x3 engine whose rank-1 instance reports an activation beyond the fp16 range; bf16x3 engines decode normally."""
    def __init__(self, dtype, fail):
        super().__init__()
        self.dtype, self.fail = dtype, fail

    def predict(self, x, ref_batch=32, max_len=None):
        if self.fail and self.dtype == "fp16x3":
            from molnextr_amd.engine import MNX_ERR_RANGE, MnxError
            raise MnxError("mnx_predict failed (-6): encoder features are not finite", code=MNX_ERR_RANGE)
        return super().predict(x, ref_batch, max_len)


def _range_worker(rank, world, n, port, q):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    raised = False
    try:
        E.run_inference(_RangeEngine("fp16x3", fail=(rank == 1)), _page, n, batch_size=2, rank=rank, world=world, group=8)
    except E.RangeFallback:
        raised = True
    # every rank rebuilds in the fallback mode and repeats: one table, one operand mode
    preds = E.run_inference(_RangeEngine("bf16x3", fail=(rank == 1)), _page, n, batch_size=2, rank=rank, world=world, group=8)
    q.put((rank, raised, sorted(preds)))
    dist.barrier()
    dist.destroy_process_group()


def test_range_fallback_is_agreed_on_by_all_ranks_before_the_gather():
    """Rank 1's encoder leaves the fp16 range, rank 0's does not: both must raise RangeFallback (a rank that restarted alone
    would leave its peer waiting in the all-gather, and the table would mix operand modes), and the repeated run completes."""
    import os
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33000 + (os.getpid() * 11) % 2000
    procs = [ctx.Process(target=_range_worker, args=(r, 2, 7, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == [(0, True, list(range(7))), (1, True, list(range(7)))]


def test_range_error_without_a_process_group_raises_rangefallback_and_other_errors_pass_through():
    from molnextr_amd.engine import MnxError
    with pytest.raises(E.RangeFallback):
        E.run_inference(_RangeEngine("fp16x3", fail=True), _page, 5, batch_size=2)

    class _Cap(_FakeEngine):
        dtype = "fp16x3"

        def predict(self, x, ref_batch=32, max_len=None):
            raise MnxError("capacity", code=-5)
    with pytest.raises(MnxError, match="capacity"):
        E.run_inference(_Cap(), _page, 5, batch_size=2)


# ---- any OTHER failure of one rank must not leave its peers waiting in a collective -------------------------
class _CapEngine(_FakeEngine):
    dtype = "fp16x3"

    def __init__(self, fail):
        super().__init__()
        self.fail = fail

    def predict(self, x, ref_batch=32, max_len=None):
        if self.fail:
            from molnextr_amd.engine import MnxError
            raise MnxError("mnx_predict failed (-5): capacity", code=-5)
        return super().predict(x, ref_batch, max_len)


def _fatal_worker(rank, world, n, port, q):
    import os
    import torch.distributed as dist
    from molnextr_amd.engine import MnxError
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    what = "returned"
    try:
        E.run_inference(_CapEngine(fail=(rank == 1)), _page, n, batch_size=2, rank=rank, world=world, group=8)
    except MnxError as e:
        what = f"own:{e}"
    except E.PeerFailed:
        what = "peer"
    q.put((rank, what))
    dist.barrier()
    dist.destroy_process_group()


def test_a_fatal_error_on_one_rank_ends_every_rank_before_the_gather():
    """Rank 1 fails with an error that has no fallback (capacity): it re-raises ITS error after telling its peer, rank 0 raises
    PeerFailed instead of blocking in the status all-reduce / the all-gather for a rank that is gone."""
    import os
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35000 + (os.getpid() * 17) % 2000
    procs = [ctx.Process(target=_fatal_worker, args=(r, 2, 7, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] == "peer" and got[1].startswith("own:") and "capacity" in got[1]
