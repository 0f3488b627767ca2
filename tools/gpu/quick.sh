#!/bin/bash
# scratch script for one-off GPU experiments (edit, run with gpurun, do not rely on its contents)
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "decode or beam or end_to_end or bond or predict or confidence or facade" > gpurun_out/t_dec.log 2>&1; echo "pytest dec rc=$?"; tail -3 gpurun_out/t_dec.log | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_pixels.py -x -q -m gpu -k "fp16x3 and not budget" > gpurun_out/t_pixels.log 2>&1; echo "pytest pixels rc=$?"; tail -3 gpurun_out/t_pixels.log | cut -c1-600
for i in 1 2; do
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sub > gpurun_out/bench20.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench20.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(d['value'], d['roofline']['achieved'], d['roofline']['stage34']['achieved'])
"
done
timeout 300 python bench.py --no-cpu-baseline --no-sub > gpurun_out/bench512.log 2>&1; tail -1 gpurun_out/bench512.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('default', d['value'], d['steps'])
"
