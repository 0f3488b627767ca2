#!/bin/bash
# scratch script for one-off GPU experiments (edit, run with gpurun, do not rely on its contents)
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "swin_tiny or batch32 or swin_full" > gpurun_out/t_attn.log 2>&1; echo "pytest enc rc=$?"; tail -3 gpurun_out/t_attn.log | cut -c1-300
for i in 1 2; do
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sub > gpurun_out/bench20.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench20.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(d['value'], d['roofline']['achieved'], d['roofline']['stage34']['achieved'], d['roofline'].get('isolated'))
for e in d['roofline_extra']:
    if 'dec_attn' not in e['kernel']: print(e['kernel'][:50], e['measured'][:10], e['achieved'], e['frac'], e['avg_launch_us'])
"
done
