#!/bin/bash
# scratch: A/B of one vs two encoder streams
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
for n in 1 2 1 2; do
  MNX_ENC_STREAMS=$n timeout 300 python bench.py --steps 20 --warmup 5 --no-sub --no-cpu-baseline > gpurun_out/bench_es$n.log 2>&1
  echo "enc_streams=$n rc=$? $(tail -1 gpurun_out/bench_es$n.log | grep -o '"value": [0-9.]*')"
done
MNX_ENC_STREAMS=2 timeout 300 python bench.py --steps 256 --warmup 16 --no-sub --no-cpu-baseline > gpurun_out/bench_es2_256.log 2>&1; echo "2 streams, 256 steps: $(tail -1 gpurun_out/bench_es2_256.log | grep -o '"value": [0-9.]*')"
MNX_ENC_STREAMS=1 timeout 300 python bench.py --steps 256 --warmup 16 --no-sub --no-cpu-baseline > gpurun_out/bench_es1_256.log 2>&1; echo "1 stream, 256 steps: $(tail -1 gpurun_out/bench_es1_256.log | grep -o '"value": [0-9.]*')"
MNX_ENC_STREAMS=2 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pipeline or grouped or neighbouring or reproducible" > gpurun_out/t_es2.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/t_es2.log
