#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --durations=25 > gpurun_out/t_parity.log 2>&1; echo "parity rc=$?"
tail -32 gpurun_out/t_parity.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-sub --no-cpu-baseline > gpurun_out/bench20.log 2>&1
echo "bench rc=$?"; tail -1 gpurun_out/bench20.log | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | head -2
