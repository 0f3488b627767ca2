#!/bin/bash
# scratch script for one-off GPU experiments (edit, run with gpurun, do not rely on its contents)
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pipeline or predict or grouped" > gpurun_out/t_pipe.log 2>&1; echo "pytest pipeline rc=$?"; tail -3 gpurun_out/t_pipe.log | cut -c1-300
for eb in 224 128; do
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sub --encode-batch $eb > gpurun_out/bench20_eb$eb.log 2>&1; echo "bench20 eb=$eb rc=$?"; tail -1 gpurun_out/bench20_eb$eb.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(d['value'], d['roofline']['achieved'], d['roofline']['stage34']['achieved'])
"
done
timeout 300 python bench.py --gpus 1 --no-cpu-baseline --no-sub > gpurun_out/bench512.log 2>&1; echo "bench512 rc=$?"; tail -1 gpurun_out/bench512.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(d['value'], d['roofline']['achieved'], d['roofline']['stage34']['achieved'])
"
