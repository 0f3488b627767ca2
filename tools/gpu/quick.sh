#!/bin/bash
# scratch script for one-off GPU experiments (edit, run with gpurun, do not rely on its contents)
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python bench.py --steps 20 --warmup 5 --no-sub --no-cpu-baseline > gpurun_out/bench_quick.log 2>&1
tail -1 gpurun_out/bench_quick.log | cut -c1-300
