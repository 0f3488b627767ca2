#!/bin/bash
# validation pass: the whole GPU suite, smoke(), GEMM shape tables (tools/gemm_lab) for both operand modes
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/t_gpu.log 2>&1; echo "pytest rc=$?"
tail -8 gpurun_out/t_gpu.log | cut -c1-400
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
for b in 128 224; do
  timeout 150 tools/gemm_lab/lab $b 20 > gpurun_out/gemm_shapes_b$b.txt 2>&1
  timeout 150 tools/gemm_lab/lab $b 20 - fp16x3 > gpurun_out/gemm_shapes_fp16x3_b$b.txt 2>&1
done
grep -c FAIL gpurun_out/gemm_shapes_b128.txt gpurun_out/gemm_shapes_b224.txt
true
