#!/bin/bash
# round-2 validation pass: GEMM shape tables, the whole GPU suite, bench at three encoder group sizes
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
for b in 64 96 128; do timeout 120 tools/gemm_lab/lab $b 20 > gpurun_out/gemm_shapes_b$b.txt 2>&1; done
grep -c FAIL gpurun_out/gemm_shapes_b*.txt
cat gpurun_out/gemm_shapes_b64.txt | cut -c1-125
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/t_gpu.log 2>&1; echo "pytest rc=$?"
tail -8 gpurun_out/t_gpu.log
for eb in 64 96 128; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-sub --no-cpu-baseline --encode-batch $eb > gpurun_out/bench20_eb$eb.log 2>&1
  echo "eb=$eb rc=$?"; tail -c 600 gpurun_out/bench20_eb$eb.log | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"frac": [0-9.]*' | head -4
done
