"""CPU: sharding + result gather with world_size 2 over gloo (the N>1 path of bench.py / the eval driver)."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from molnextr_amd import shard


def test_shard_range_partitions():
    for n in (0, 1, 31, 32, 255, 256):
        for world in (1, 2, 3, 8):
            spans = [shard.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1


def _fake_results(lo, hi, kmax):
    rng = np.random.default_rng(1234)
    B = 8
    lengths = rng.integers(1, 480, size=B).astype(np.int32)
    tokens = rng.integers(0, 229, size=(B, 480)).astype(np.int32)
    n_atoms = rng.integers(0, kmax + 1, size=B).astype(np.int32)
    atom_idx = rng.integers(0, 480, size=(B, kmax)).astype(np.int32)
    edges = rng.integers(0, 7, size=(B, kmax, kmax)).astype(np.uint8)
    return tuple(a[lo:hi] for a in (tokens, lengths, atom_idx, n_atoms, edges))


def test_pack_unpack_round_trip():
    kmax = 23
    tokens, lengths, atom_idx, n_atoms, edges = _fake_results(0, 8, kmax)
    rec = shard.pack_records(tokens, lengths, atom_idx, n_atoms, edges, kmax)
    assert rec.shape == (8, shard.record_words(kmax)) and rec.dtype == torch.int32
    out = shard.unpack_records(rec, kmax)
    for b, o in enumerate(out):
        assert o["tokens"] == tokens[b, :lengths[b]].tolist()
        assert o["atom_idx"] == atom_idx[b, :n_atoms[b]].tolist()
        assert o["edges"] == edges[b, :n_atoms[b], :n_atoms[b]].astype(int).tolist()


def test_device_packing_matches_host_packing():
    kmax = 24
    tokens, lengths, atom_idx, n_atoms, edges = _fake_results(0, 8, kmax)
    # the engine zero-fills beyond lengths / n_atoms; host packing keeps whatever is passed: use identical inputs
    a = shard.pack_records(tokens, lengths, atom_idx, n_atoms, edges, kmax)
    b = shard.pack_records_device(torch.from_numpy(tokens), torch.from_numpy(lengths), torch.from_numpy(atom_idx),
                                  torch.from_numpy(n_atoms), torch.from_numpy(edges))
    assert torch.equal(a, b)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    kmax = 23
    lo, hi = shard.shard_range(8, rank, world)
    tokens, lengths, atom_idx, n_atoms, edges = _fake_results(lo, hi, kmax)
    rec = shard.pack_records(tokens, lengths, atom_idx, n_atoms, edges, kmax)
    allrec = shard.gather_records(rec)
    # records sized by the largest molecule of the job: both ranks must agree on the capacity (scalar all-reduce MAX)
    k = shard.common_atom_capacity(torch.from_numpy(n_atoms), 23)
    ai, ed = shard.trim_atoms(torch.from_numpy(atom_idx), torch.from_numpy(edges), k)
    small = shard.gather_records(shard.pack_records(tokens, lengths, ai.numpy(), n_atoms, ed.numpy(), k))
    full_atoms = _fake_results(0, 8, 23)[3]
    assert k == min(23, max(4, (int(full_atoms.max()) + 3) // 4 * 4)), "capacity must be the job-wide maximum"
    assert [d["edges"] for d in shard.unpack_records(small, k)] == [d["edges"] for d in shard.unpack_records(allrec, 23)]
    assert [d["atom_idx"] for d in shard.unpack_records(small, k)] == \
        [d["atom_idx"] for d in shard.unpack_records(allrec, 23)]
    dist.barrier()
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)          # the max-over-ranks timing reduction bench.py uses
    q.put((rank, allrec.numpy().tobytes(), float(t)))
    dist.destroy_process_group()


def test_gather_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = shard.pack_records(*_fake_results(0, 8, 23), 23).numpy().tobytes()
    for rank, blob, tmax in got:
        assert blob == full, f"rank {rank}: gathered records differ from the unsharded result"
        assert tmax == 2.0
