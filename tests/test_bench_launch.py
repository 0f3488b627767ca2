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
