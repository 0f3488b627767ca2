"""GPU (-m gpu): SURVEY 8(e) on the hardware a 1-GPU box has — the result gather of the multi-GPU path with the REAL
collective backend (`nccl` = RCCL), one rank under torch.distributed.run. The N > 1 logic is covered on the CPU with two gloo
ranks (tests/test_shard.py, tests/test_bench_launch.py); what those cannot show is that the RCCL calls themselves
(all_reduce MAX of the atom capacity, all_gather_into_tensor of the records) run on device tensors produced by the engine.
Reference: main.py:295-296 (all_gather_object of prediction dicts), :577-581 (process-group init)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torchrun(script_and_args, timeout):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_and_args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_rccl_gathers_the_records_of_real_engine_output_on_one_rank():
    r = _torchrun([os.path.join(ROOT, "tools", "rccl_selfcheck.py")], 600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "RCCL_SELFCHECK_OK ranks=1 records=64" in r.stdout, r.stdout[-2000:]


def test_bench_force_gather_runs_the_rccl_path_with_one_rank():
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-gather", "--steps", "2", "--warmup", "1",
                   "--no-cpu-baseline", "--no-sub"], 900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["rccl_ranks"] == 1 and line["n_gpus"] == 1 and line["value"] > 0
