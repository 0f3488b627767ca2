"""CPU: `python bench.py --gpus N` must produce N ranks or fail loudly — the decision logic runs
before anything touches a GPU and is tested here with stub environments."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_single_gpu_runs_in_process():
    assert bench.plan_launch(1, {}, 1) == ("run", 1)
    assert bench.plan_launch(1, {"WORLD_SIZE": "1", "RANK": "0"}, 8) == ("run", 1)


def test_multi_gpu_without_torchrun_spawns_the_ranks_itself():
    assert bench.plan_launch(8, {}, 8) == ("spawn", None)
    assert bench.plan_launch(2, {}, 8) == ("spawn", None)


def test_under_torchrun_world_size_must_equal_gpus():
    assert bench.plan_launch(4, {"WORLD_SIZE": "4"}, 8) == ("run", 4)
    with pytest.raises(SystemExit, match="ranks and --gpus must agree"):
        bench.plan_launch(8, {"WORLD_SIZE": "1"}, 8)
    with pytest.raises(SystemExit, match="ranks and --gpus must agree"):
        bench.plan_launch(1, {"WORLD_SIZE": "2"}, 8)


def test_fewer_devices_than_requested_is_an_error_not_a_smaller_run():
    with pytest.raises(SystemExit, match="exposes 1 GPU"):
        bench.plan_launch(2, {}, 1)
    with pytest.raises(SystemExit, match="exposes 1 GPU"):
        bench.plan_launch(8, {"WORLD_SIZE": "8"}, 1)
    with pytest.raises(SystemExit):
        bench.plan_launch(0, {}, 1)


def test_spawn_command_is_one_rank_per_gpu_on_loopback(monkeypatch):
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    rc = bench.spawn_ranks(4, ["--gpus", "4", "--steps", "3"])
    assert rc == 7
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_bench_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "n_gpus" not in r.stdout


# ---- N > 1 dry run: bench.py's real record path with two ranks over gloo and a stub engine ----------------
def _stub_engine_out(first, n, kmax):
    """What Engine.predict returns for images [first, first + n): deterministic, zero-filled beyond lengths / n_atoms as the
    engine's outputs are."""
    import numpy as np
    import torch
    tokens = np.zeros((n, 480), np.int32)
    lengths = np.zeros(n, np.int32)
    atom_idx = np.zeros((n, kmax), np.int32)
    n_atoms = np.zeros(n, np.int32)
    edges = np.zeros((n, kmax, kmax), np.uint8)
    for i in range(n):
        g = np.random.default_rng(1000 + first + i)
        L = int(g.integers(5, 480))
        k = min(int(g.integers(0, 40)), L)
        lengths[i], n_atoms[i] = L, k
        tokens[i, :L] = g.integers(3, 229, size=L)
        atom_idx[i, :k] = np.sort(g.choice(L, size=k, replace=False)) if k else []
        edges[i, :k, :k] = g.integers(0, 7, size=(k, k))
    return {k_: torch.from_numpy(v) for k_, v in dict(tokens=tokens, lengths=lengths, atom_idx=atom_idx, n_atoms=n_atoms,
                                                      edges=edges).items()}


def _rank_main(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from molnextr_amd import shard
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    kmax, steps = 160, 2
    n_local = steps * bench.BATCH
    # step s, rank r owns images [(s * world + r) * 32, + 32) — bench.py's shard rule
    outs = [_stub_engine_out((s * world + rank) * bench.BATCH, bench.BATCH, kmax) for s in range(steps)]
    out = {k: torch.cat([o[k] for o in outs]) for k in outs[0]}
    landing = torch.empty(n_local * world * shard.record_words(kmax), dtype=torch.int32)
    rec, k = bench.land_records(out, kmax, rank, world, n_local, landing)
    q.put((rank, k, tuple(rec.shape), rec.numpy().tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_land_the_job_through_the_real_record_path():
    import numpy as np
    import torch
    import torch.multiprocessing as mp
    from molnextr_amd import shard
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    world, kmax, steps = 2, 160, 2
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {r: (k, shp, blob) for r, k, shp, blob in (q.get(timeout=180) for _ in procs)}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    k0 = got[0][0]
    assert got[1][0] == k0 and k0 % 4 == 0 and k0 < kmax, "both ranks size the records by the job's largest molecule"
    n_local = steps * bench.BATCH
    W_ = shard.record_words(k0)
    assert got[0][1] == (world * n_local, W_) and got[1][1] == (n_local, W_), "rank 0 lands the whole job, rank 1 its shard"
    whole = torch.from_numpy(np.frombuffer(got[0][2], dtype=np.int32).reshape(world * n_local, W_).copy())
    mine1 = np.frombuffer(got[1][2], dtype=np.int32).reshape(n_local, W_)
    assert np.array_equal(whole[n_local:].numpy(), mine1), "rank order = record order"
    recs = shard.unpack_records(whole, k0)
    for r in range(world):
        for s in range(steps):
            ref = _stub_engine_out((s * world + r) * bench.BATCH, bench.BATCH, kmax)
            for i in range(bench.BATCH):
                d = recs[r * n_local + s * bench.BATCH + i]
                L, k = int(ref["lengths"][i]), int(ref["n_atoms"][i])
                assert d["tokens"] == ref["tokens"][i, :L].tolist()
                assert d["atom_idx"] == ref["atom_idx"][i, :k].tolist()
                assert d["edges"] == ref["edges"][i, :k, :k].int().tolist()


def test_rank_cpu_sets_follow_the_device_topology():
    # two NUMA nodes with SMT siblings, 8 devices, 4 per node (the usual MI355X node): every rank stays on its device's socket
    n0 = bench.parse_cpulist("0-63,128-191")
    n1 = bench.parse_cpulist("64-127,192-255")
    lists = [n0] * 4 + [n1] * 4
    avail = set(range(256))
    sets = [bench.rank_cpus(r, lists, avail) for r in range(8)]
    assert all(len(s) == 32 for s in sets)
    assert all(set(sets[r]) <= set(lists[r]) for r in range(8))
    assert len(set().union(*map(set, sets))) == 256, "disjoint: no two ranks share a core"
    # a cgroup that exposes only part of the node: only usable CPUs are handed out
    sets = [bench.rank_cpus(r, lists, set(range(0, 32)) | set(range(64, 96))) for r in range(8)]
    assert [len(s) for s in sets] == [8] * 8 and set(sets[5]) <= set(range(64, 96))
    # no topology information (sysfs unreadable): contiguous slices of what the process may use
    sets = [bench.rank_cpus(r, [None] * 4, set(range(16))) for r in range(4)]
    assert sets == [list(range(0, 4)), list(range(4, 8)), list(range(8, 12)), list(range(12, 16))]
    # fewer usable CPUs on the node than ranks sharing it: fall back to slices instead of an empty set
    assert bench.rank_cpus(3, [n0] * 4, {0, 1}) != []
    assert bench.pin_rank(0, 1) is None


def test_traffic_figure_is_only_taken_from_a_file_made_with_these_kernel_sources(tmp_path, monkeypatch):
    """roofline.traffic comes from rocprofv3 --pmc passes that cannot run inside a timed bench; bench.py quotes the file only
    when it carries the digest of the kernel sources THIS tree has (tools/collect_traffic.py stamps the same digest)."""
    import json
    import runpy
    have = bench.library_sha16()
    assert len(have) == 16 and have == bench.library_sha16()
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "library_sha16", lambda: have)
    assert bench.gemm_traffic("fp16x3", 512)[0] is None                      # no file
    (prof / "r06_gemm_traffic_fp16x3_b512.json").write_text(json.dumps({"hbm_bytes_per_launch": 123.4, "library_sha16": "0" * 16}))
    val, why = bench.gemm_traffic("fp16x3", 512)
    assert val is None and "refused" in why
    (prof / "r06_gemm_traffic_fp16x3_b512.json").write_text(json.dumps({"hbm_bytes_per_launch": 123.4, "library_sha16": have}))
    val, why = bench.gemm_traffic("fp16x3", 512)
    assert val == 123 and have in why
    # the collector computes the digest the same way (it is a script: run its digest lines on this tree)
    src = open(os.path.join(ROOT, "tools", "collect_traffic.py")).read()
    ns = {"__file__": os.path.join(ROOT, "tools", "collect_traffic.py"), "out": {}}
    exec(src[src.index("# the GEMM kernel sources the counters were collected on"):src.index("json.dump(out, open(sys.argv[3]")], ns)
    assert ns["out"]["library_sha16"] == have
