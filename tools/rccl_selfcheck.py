#!/usr/bin/env python3
"""Row (e) on the hardware at hand: the multi-GPU result path of bench.py / evaluate.py with the REAL collective backend.

Run under torch.distributed.run with any number of ranks on one node (one rank is what a 1-GPU box can do; the driver's
8-GPU node runs the same code with 8):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/rccl_selfcheck.py

Every rank initialises the `nccl` process group (= RCCL on ROCm) on its GPU, runs the engine on its own 64 images
(mnx_predict: Swin-B encode, greedy decode, atom scan, bond head), and hands the device tensors to bench.land_records with the
gather forced: shard.common_atom_capacity (scalar all-reduce MAX) + shard.pack_records_device + ONE all_gather_into_tensor
(reference main.py:295-296 does this with all_gather_object on pickled dicts) + landing in pinned host memory. Rank 0 then
checks EVERY gathered record of its own shard against shard.unpack_records of its local tensors packed on the host
(pack_records: an independent implementation of the layout), and every rank checks that the collective returned world x n
records. Prints `RCCL_SELFCHECK_OK ranks=<world> records=<n>` on rank 0; any mismatch raises."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench                                                               # noqa: E402
from molnextr_amd import shard                                             # noqa: E402
from molnextr_amd import weights as W                                      # noqa: E402
from molnextr_amd.engine import Engine                                     # noqa: E402


def main():
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)       # nccl == RCCL on ROCm
    assert dist.get_backend() == "nccl"
    n = 64
    ck = W.synthetic_checkpoint(0)
    eng = Engine(ck["encoder"], ck["decoder"], device=local, max_batch=n, dec_slots=128)
    imgs = W.synthetic_images(n, first_index=rank * n).to(dev)
    out = eng.predict(imgs, ref_batch=32)
    kmax = eng.max_atoms
    pinned = torch.empty(world * n * shard.record_words(kmax), dtype=torch.int32, pin_memory=True)
    landed, k = bench.land_records(out, kmax, rank, world, n, pinned, gather=True)
    torch.cuda.current_stream().synchronize()
    W_ = shard.record_words(k)
    want_rows = world * n if rank == 0 else n
    assert tuple(landed.shape) == (want_rows, W_), (tuple(landed.shape), want_rows, W_)
    # every rank: its own shard as the collective returned it == the host packing of its local tensors
    tokens, lengths, atom_idx, n_atoms, edges = (out[key].cpu().numpy() for key in ("tokens", "lengths", "atom_idx", "n_atoms", "edges"))
    assert int(n_atoms.max()) <= k <= kmax and k % 4 == 0
    host = shard.pack_records(tokens, lengths, atom_idx, n_atoms, edges[:, :k, :k], k)
    mine = landed[rank * n:(rank + 1) * n] if rank == 0 else landed
    assert torch.equal(mine.cpu(), host), "gathered records differ from the host packing of the local tensors"
    recs = shard.unpack_records(mine, k)
    for i, d in enumerate(recs):
        assert d["tokens"] == tokens[i, :lengths[i]].tolist()
        assert d["atom_idx"] == atom_idx[i, :n_atoms[i]].tolist()
        assert d["edges"] == edges[i, :n_atoms[i], :n_atoms[i]].astype(int).tolist()
    assert sum(len(d["atom_idx"]) for d in recs) > 0 and sum(len(d["tokens"]) for d in recs) > n
    eng.close()
    dist.barrier()
    if rank == 0:
        print(f"RCCL_SELFCHECK_OK ranks={world} records={want_rows} atom_capacity={k}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
