#!/bin/bash
# scratch script for one-off GPU experiments (edit, run with gpurun, do not rely on its contents)
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
for eb in 128 160 224 256; do
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sub --encode-batch $eb > gpurun_out/bench20_eb$eb.log 2>&1; echo "bench20 eb=$eb rc=$?"; tail -1 gpurun_out/bench20_eb$eb.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(d['value'], d['roofline']['achieved'], d['roofline']['stage34']['achieved'], d['roofline']['isolated'])
"
done
for eb in 128 224; do
timeout 300 python bench.py --gpus 1 --no-cpu-baseline --no-sub --encode-batch $eb > gpurun_out/bench512_eb$eb.log 2>&1; echo "bench512 eb=$eb rc=$?"; tail -1 gpurun_out/bench512_eb$eb.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(d['value'], d['roofline']['achieved'], d['roofline']['stage34']['achieved'])
"
done
