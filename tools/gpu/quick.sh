#!/bin/bash
# scratch script for one-off GPU experiments (edit, run with gpurun, do not rely on its contents)
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
MNX_ENC_CHUNK=8 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "swin_tiny or batch32 or swin_full or grouped" > gpurun_out/t_attn.log 2>&1; echo "pytest enc (chunk 8) rc=$?"; tail -3 gpurun_out/t_attn.log | cut -c1-300
MNX_ENC_CHUNK=8 timeout 600 python -m pytest tests/test_gpu_pixels.py -x -q -m gpu -k "fp16x3 and not budget" > gpurun_out/t_pixels.log 2>&1; echo "pytest pixels (chunk 8) rc=$?"; tail -3 gpurun_out/t_pixels.log | cut -c1-300
for ch in 0 8 16 32 56; do
MNX_ENC_CHUNK=$ch timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sub > gpurun_out/bench20_ch$ch.log 2>&1; echo "chunk $ch bench rc=$?"; tail -1 gpurun_out/bench20_ch$ch.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(d['value'], d['roofline']['achieved'], d['roofline'].get('isolated'))
for e in d['roofline_extra']:
    if 'dec_attn' not in e['kernel'] and e['measured']=='isolated': print('   ', e['kernel'][:40], e['achieved'], e['frac'], e['avg_launch_us'], e['launches'])
"
done
