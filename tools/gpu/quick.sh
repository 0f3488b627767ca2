#!/bin/bash
# scratch script for one-off GPU experiments (edit, run with gpurun, do not rely on its contents)
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --force-gather --steps 4 --warmup 1 --no-sub --no-cpu-baseline > gpurun_out/bench_torchrun.log 2>&1
echo "rc=$?"; tail -1 gpurun_out/bench_torchrun.log | cut -c1-400
