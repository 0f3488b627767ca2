#!/bin/bash
# scratch script for one-off GPU experiments (edit, run with gpurun, do not rely on its contents)
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sub > gpurun_out/b_head.log 2>&1; tail -1 gpurun_out/b_head.log | cut -c1-160
