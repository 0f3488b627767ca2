"""Multi-GPU sharding of the predict path: images are independent units, so a batch is split contiguously by image
index, one process per GPU, and the only collective is a gather of fixed-size result records.

Mirrors the reference's data-parallel inference (reference main.py:440-446 DistributedSampler + valid_fn, and
`dist.all_gather_object(gathered_preds, predictions)` main.py:295-296) — but instead of pickling Python dicts through
the CPU, every image's result is a fixed-size int32 record gathered with ONE all-gather (RCCL over xGMI when the
tensors live on the GPU, gloo for the CPU tests). The payload is ~28 KB per image, i.e. latency-bound: a single direct
all-gather is the right shape, ring/bucket tuning is irrelevant at this size.

Parity contract: the reference's outputs depend on the row index inside its per-rank batch (positional-encoding
quirk), so a sharded run equals the reference run with the SAME per-rank batches — `shard_range` is part of it.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np
import torch
import torch.distributed as dist

MAX_LEN = 480


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split; the first (n % world) ranks get one extra item."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def record_words(kmax: int) -> int:
    return 2 + MAX_LEN + kmax + (kmax * kmax + 3) // 4


def pack_records(tokens: np.ndarray, lengths: np.ndarray, atom_idx: np.ndarray, n_atoms: np.ndarray,
                 edges: np.ndarray, kmax: int) -> torch.Tensor:
    """[B, record_words] int32: length, n_atoms, tokens[480], atom_idx[kmax], edges (uint8[kmax*kmax] packed 4/word)."""
    B = len(lengths)
    rec = np.zeros((B, record_words(kmax)), dtype=np.int32)
    rec[:, 0] = lengths
    rec[:, 1] = n_atoms
    T = min(tokens.shape[1], MAX_LEN)
    rec[:, 2:2 + T] = tokens[:, :T]
    rec[:, 2 + MAX_LEN:2 + MAX_LEN + kmax] = atom_idx[:, :kmax]
    e = np.zeros((B, ((kmax * kmax + 3) // 4) * 4), dtype=np.uint8)
    e[:, :kmax * kmax] = edges.reshape(B, -1)[:, :kmax * kmax]
    rec[:, 2 + MAX_LEN + kmax:] = e.view(np.int32)
    return torch.from_numpy(rec)


def pack_records_device(tokens: torch.Tensor, lengths: torch.Tensor, atom_idx: torch.Tensor, n_atoms: torch.Tensor,
                        edges: torch.Tensor) -> torch.Tensor:
    """Same record layout as pack_records, built on the device from the engine's output tensors
    (tokens int32 [n,480], atom_idx int32 [n,kmax], edges uint8 [n,kmax,kmax]); no host round trip before the gather."""
    n, kmax = atom_idx.shape
    assert tokens.shape[1] == MAX_LEN and (kmax * kmax) % 4 == 0
    e32 = edges.reshape(n, kmax * kmax).contiguous().view(torch.int32)
    return torch.cat([lengths.view(n, 1), n_atoms.view(n, 1), tokens, atom_idx, e32], dim=1).contiguous()


def common_atom_capacity(n_atoms: torch.Tensor, kmax: int) -> int:
    """Smallest record capacity (a multiple of 4, <= kmax) that holds every molecule of EVERY rank: the bond matrix is
    kmax^2 bytes per record but molecules have ~30 atoms, so the exchange is sized by the largest molecule of the job
    (one scalar all-reduce MAX) instead of by the engine's capacity — 4-5x fewer bytes over xGMI and PCIe."""
    k = n_atoms.max().reshape(1).to(torch.int32) if n_atoms.numel() else torch.zeros(1, dtype=torch.int32,
                                                                                        device=n_atoms.device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(k, op=dist.ReduceOp.MAX)
    return min(kmax, max(4, (int(k.item()) + 3) // 4 * 4))


def trim_atoms(atom_idx: torch.Tensor, edges: torch.Tensor, k: int):
    """Views of the engine outputs cut to `k` atoms per molecule (made contiguous for packing)."""
    return atom_idx[:, :k].contiguous(), edges[:, :k, :k].contiguous()


def unpack_records(rec: torch.Tensor, kmax: int) -> List[dict]:
    r = rec.cpu().numpy()
    out = []
    for row in r:
        n, k = int(row[0]), int(row[1])
        e = np.ascontiguousarray(row[2 + MAX_LEN + kmax:]).view(np.uint8)[:kmax * kmax].reshape(kmax, kmax)
        out.append({"tokens": row[2:2 + n].tolist(), "atom_idx": row[2 + MAX_LEN:2 + MAX_LEN + k].tolist(),
                    "edges": e[:k, :k].astype(int).tolist()})
    return out


def gather_records(rec: torch.Tensor, force: bool = False) -> torch.Tensor:
    """All-gather equal-sized [b, W] int32 records from every rank -> [world*b, W] (rank order = image order)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return rec
    world = dist.get_world_size()
    out = torch.empty((world * rec.shape[0], rec.shape[1]), dtype=rec.dtype, device=rec.device)
    dist.all_gather_into_tensor(out, rec.contiguous())
    return out


def max_over_ranks(value: int, device) -> int:
    """The largest `value` of any rank (one scalar all-reduce MAX; the value itself without a process group): the ranks of
    evaluate.run_inference agree on a status code with it before they enter the gather."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


def any_rank(flag: bool, device) -> bool:
    """True on every rank when `flag` is true on at least one (one scalar all-reduce MAX; a plain bool without a process
    group). evaluate.run_inference uses it so that all ranks leave the fp16 operand mode together."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(t.item())
