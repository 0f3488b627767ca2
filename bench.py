#!/usr/bin/env python3
"""bench.py — throughput of the MolNexTR predict hot path on MI355X (molecules/s at 384x384, batch 32 per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path over one batch of 32 synthetic 384x384x3 images per GPU, inputs already
resident in HBM: Swin-B encode (bf16 MFMA GEMMs) -> enc_transform + cross-KV -> greedy decode until EOS / 480 tokens
(reference default max_length) -> host detokenisation -> bond head; with N > 1 the batch of N*32 images is sharded
by image across the ranks and the fixed-size result records are all-gathered with RCCL inside the step.
Weights: deterministic synthetic checkpoint in the reference's exact state-dict layout (no pretrained checkpoint
exists offline). The K timed batches are submitted to the engine's continuous-batching entry point (mnx_predict):
every batch of 32 stays ONE reference batch (its own positional-encoding numbering), but up to 8 batches are
resident in the decoder at once and finished rows are refilled with the next batch (`--mode batch` runs the
batches strictly one after the other through mnx_encode / mnx_decode_greedy / host detokenise / mnx_edges).

Rank 0 prints ONE JSON line (contract in the task statement), including
  roofline      the dominant FLOP kernel (gemm_tn_kernel, bf16 MFMA): algorithmic FLOP (2*M*N*K per launch) divided
                by its event-bracketed duration, measured with HIP events on the engine's stream over a replay of the
                timed steps; peak = 2500 TFLOP/s dense bf16 (MI355X_MICROARCH.md)
  cpu_baseline  the CPU oracle (oracle/, bit-equal to the reference in the build container) on a bounded sample of
                the same workload on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from molnextr_amd import shard  # noqa: E402
from molnextr_amd import weights as W  # noqa: E402
from molnextr_amd.tokenizer import get_tokenizer  # noqa: E402

BATCH = 32
PEAK_BF16_TFLOPS = 2500.0


def run_batch(eng, tok, images, kmax, max_len):
    """Encoder.forward + Decoder.decode for one batch on the current stream. Returns packed result records (CPU)."""
    feats = eng.encode(images)
    out = eng.decode_greedy(feats, max_len=max_len, want_logp=False)
    lens = out["lengths"].cpu().numpy()
    toks = out["tokens"].cpu().numpy()
    B = len(lens)
    n_atoms = np.zeros(B, dtype=np.int32)
    atom_idx = np.zeros((B, kmax), dtype=np.int32)
    for b in range(B):
        idx = tok.sequence_to_smiles(toks[b, :lens[b]].tolist())["indices"]
        n_atoms[b] = len(idx)
        atom_idx[b, :len(idx)] = idx
    edges, _ = eng.edges(out["hidden"], torch.from_numpy(atom_idx), torch.from_numpy(n_atoms))
    rec = shard.pack_records(toks, lens, atom_idx, n_atoms, edges.cpu().numpy(), kmax)
    return rec, lens, n_atoms


def gemm_algorithmic_bytes(batch=BATCH):
    """Average algorithmic HBM bytes per encoder GEMM launch (A + W read once, output written once, residual read
    for the two residual epilogues) for Swin-B @384: the figure `roofline.traffic` is compared with."""
    total, launches = 0, 0
    for s, (L, C, depth) in enumerate([(9216, 128, 2), (2304, 256, 2), (576, 512, 18), (144, 1024, 2)]):
        M = batch * L
        per_block = [(M, 3 * C, C, 2, 0), (M, C, C, 4, 4), (M, 4 * C, C, 2, 0), (M, C, 4 * C, 4, 4)]
        for (m, n, k, out_b, res_b) in per_block:
            total += depth * (m * k * 2 + n * k * 2 + m * n * (out_b + res_b))
            launches += depth
        if s < 3:
            total += (M // 4) * 4 * C * 2 + 2 * C * 4 * C * 2 + (M // 4) * 2 * C * 4
            launches += 1
    return total / launches


def cpu_baseline(ck, seconds_budget=25.0):
    """Oracle on host cores: B=2 images through encoder + greedy decode + bond head, repeated within the budget."""
    from oracle.decoder import greedy_decode
    from oracle.edges import predict_edges
    from oracle.swin import encoder_forward
    tok = get_tokenizer()["chartok_coords"]
    threads = torch.get_num_threads()
    img = W.synthetic_images(2)
    t0 = time.time()
    n = 0
    lens_all = []
    while True:
        f = encoder_forward(img, ck["encoder"])
        g = greedy_decode(f, ck["decoder"])
        for b in range(2):
            idx = tok.sequence_to_smiles(g.tokens[b])["indices"]
            predict_edges(g.hidden[b], idx, ck["decoder"])
        lens_all += [len(t) for t in g.tokens]
        n += 2
        if time.time() - t0 > seconds_budget * 0.6 or n >= 8:
            break
    dt = time.time() - t0
    return {"value": round(n / dt, 3), "unit": "molecules/s", "cores": threads, "kind": "port",
            "sample": f"{n} synthetic images (indices 0,1 repeated) through the CPU oracle (fp32 torch ops), encoder + "
                      f"greedy decode to EOS (mean length {np.mean(lens_all):.0f}) + bond head, {dt:.1f} s wall"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--mode", default="pipeline", choices=["pipeline", "batch"])
    ap.add_argument("--max-len", type=int, default=480)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--encode-batch", type=int, default=int(os.environ.get("MNX_ENCODE_BATCH", "64")),
                    help="images per encoder launch group (a multiple of 32; decode batches stay 32)")
    ap.add_argument("--slots", type=int, default=int(os.environ.get("MNX_SLOTS", "3072")),
                    help="sequences resident in the decoder (multiple of 32, <= 4096)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-gather", action="store_true", help="run the RCCL record gather even with one rank")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or ("RANK" in os.environ and args.force_gather)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm
    from molnextr_amd.engine import Engine

    ck = W.synthetic_checkpoint(0)
    tok = get_tokenizer()["chartok_coords"]
    eng = Engine(ck["encoder"], ck["decoder"], device=local, max_batch=max(BATCH, args.encode_batch), dtype=args.dtype,
                 dec_slots=args.slots)
    kmax = eng.max_atoms
    # step s, rank r owns images [(s*world + r)*32, +32): every step has its own images (8 distinct batches cycle)
    n_distinct = 8
    pool = [W.synthetic_images(BATCH, first_index=(s * world + rank) * BATCH).to(dev) for s in range(n_distinct)]

    def images_for(first_step, count):
        return torch.cat([pool[(first_step + i) % n_distinct] for i in range(count)]).contiguous()

    stats = {}
    host_buf = {}

    def run(first_step, count):
        imgs = images_for(first_step, count)
        torch.cuda.synchronize()
        return imgs

    def process(imgs, count):
        """`count` steps over resident images; returns the gathered result records on the host."""
        if args.mode == "pipeline":
            out = eng.predict(imgs, ref_batch=BATCH, max_len=args.max_len)
            stats["lens"] = out["lengths"].cpu().numpy()
            stats["atoms"] = out["n_atoms"].cpu().numpy()
            rec = None
            if args.max_len == shard.MAX_LEN:
                # records sized by the largest molecule of the whole job (one scalar all-reduce), not by max_atoms
                k = shard.common_atom_capacity(out["n_atoms"], kmax)
                ai, ed = shard.trim_atoms(out["atom_idx"], out["edges"], k)
                rec = shard.pack_records_device(out["tokens"], out["lengths"], ai, out["n_atoms"], ed)
        else:
            recs, lens, atoms = [], [], []
            for i in range(count):
                r, l, a = run_batch(eng, tok, imgs[i * BATCH:(i + 1) * BATCH], kmax, args.max_len)
                recs.append(r)
                lens += l.tolist()
                atoms += a.tolist()
            stats["lens"], stats["atoms"] = np.array(lens), np.array(atoms)
            rec = torch.cat(recs).to(dev)
        if rec is not None:
            if world > 1 or args.force_gather:   # result gather over xGMI: fixed-size records, one RCCL all-gather
                rec = shard.gather_records(rec, force=args.force_gather)
            # results land in pinned host memory: every rank keeps its own shard, rank 0 the gathered whole
            mine = rec if (rank == 0 or world == 1) else rec[rank * count * BATCH:(rank + 1) * count * BATCH]
            land = host_buf["pinned"][:mine.numel()].view(mine.shape)
            land.copy_(mine, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            rec = land
        return rec

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # pinned landing buffer for the result records, allocated once outside the timed region (capacity: every record
    # at the engine's max_atoms; the records actually exchanged are sized by the largest molecule of the job)
    gathered = world if (rank == 0 and (world > 1 or args.force_gather)) else 1
    if args.steps < 1:
        raise SystemExit("--steps must be >= 1")
    host_buf["pinned"] = torch.empty(max(args.steps, args.warmup) * BATCH * gathered * shard.record_words(kmax),
                                     dtype=torch.int32, pin_memory=True)
    eb = max(BATCH, args.encode_batch)   # images per encoder launch group
    live = args.mode == "pipeline"
    groups = max(1, args.steps * BATCH // eb)
    stride = max(1, groups // 12)        # ~12 encoder launch groups of the timed region get their GEMMs bracketed
    if args.warmup > 0:
        imgs = run(0, args.warmup)
        if live:
            eng.profile(max(1, args.warmup * BATCH // eb // 12))    # creates the event pool outside the timed region
        process(imgs, args.warmup)
        if live:
            eng.profile_read()
    imgs = run(args.warmup, args.steps)
    if live:
        eng.profile(stride)              # HIP events on the encoder stream, live inside the timed region
    barrier()
    t0 = time.perf_counter()
    process(imgs, args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    live_ms, live_flop, live_n = eng.profile_read() if live else (0.0, 0.0, 0)
    eng.profile(False)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    batches = [imgs[i * eb:(i + 1) * eb].contiguous() for i in range(min(args.steps * BATCH // eb, 4))]

    out = None
    if rank == 0:
        # ---- roofline of the dominant FLOP kernel (all encoder GEMM launches). `achieved` is measured LIVE: HIP events
        # around every GEMM of ~12 encoder launch groups spread over the timed region, i.e. next to the decoder; the
        # same launches replayed afterwards on an otherwise idle GPU are reported as `isolated`.
        eng.profile(True)
        for b in batches:
            eng.encode(b)
        iso_ms, iso_flop, iso_n = eng.profile_read()
        eng.profile(False)
        isolated = iso_flop / (iso_ms * 1e-3) / 1e12 if iso_ms > 0 else 0.0
        if live_n > 0:
            gemm_ms, gemm_flop, launches = live_ms, live_flop, live_n
        else:
            gemm_ms, gemm_flop, launches = iso_ms, iso_flop, iso_n
        achieved = gemm_flop / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", f"r01_gemm_traffic_b{eb}.json")
        if os.path.exists(tpath):       # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/collect_traffic.py)
            with open(tpath) as f:
                traffic = round(json.load(f)["hbm_bytes_per_launch"])
            traffic_src = (f"profiles/r01_gemm_traffic_b{eb}.json (separate rocprofv3 --pmc passes over the same "
                           "encoder launches)")
        roofline = {"kernel": "mnx::gemm_tn_glds_kernel (bf16 MFMA 16x16x32, all encoder Linear layers)",
                    "bound": "mfma", "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": round(gemm_algorithmic_bytes(eb)),
                    "images_per_launch": eb,
                    "launches": int(launches), "avg_launch_us": round(gemm_ms * 1e3 / max(launches, 1), 2),
                    "flop_per_launch_avg": round(gemm_flop / max(launches, 1)),
                    "measured": ("live: HIP events on the encoder stream inside the timed region" if live_n > 0
                                 else "replay after the timed region"),
                    "isolated": {"achieved": round(isolated, 1), "avg_launch_us": round(iso_ms * 1e3 / max(iso_n, 1), 2),
                                 "launches": int(iso_n)}}
        cpu = None if (args.no_cpu_baseline or world > 1) else cpu_baseline(ck)   # host baseline: rank 0 at N=1 only
        total = args.steps * BATCH * world
        out = {
            "metric": "molecules/sec (384x384, bs32 per GPU), full predict hot path",
            "value": round(total / elapsed, 2), "unit": "molecules/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "batch=32 synthetic 384x384x3 images per GPU, synthetic_checkpoint(0) in the "
                                   "reference state-dict layout (no pretrained weights offline), Swin-B encode + greedy "
                                   f"decode to EOS (max_length {args.max_len}) + atom positions + bond head"
                                   + (", RCCL all-gather of result records" if world > 1 else ""),
                       "batch_per_gpu": BATCH, "global_batch": BATCH * world,
                       "mode": (f"continuous batching: up to {args.slots // 32} reference batches ({args.slots} sequences) resident in the decoder"
                                if args.mode == "pipeline" else "one batch at a time"),
                       "decoded_len_mean": round(float(np.mean(stats["lens"])), 1),
                       "decoded_len_max": int(np.max(stats["lens"])),
                       "atoms_mean": round(float(np.mean(stats["atoms"])), 1),
                       "parallelism": f"dp{world} (shard by image, no data-path collective)"},
            "roofline": roofline, "cpu_baseline": cpu,
        }
    eng.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
