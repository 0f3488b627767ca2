#!/bin/bash
# validation pass: GEMM shape tables (tools/gemm_lab), the whole GPU suite
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
for b in 64 96 128; do timeout 120 tools/gemm_lab/lab $b 20 > gpurun_out/gemm_shapes_b$b.txt 2>&1; done
grep -c FAIL gpurun_out/gemm_shapes_b*.txt
cat gpurun_out/gemm_shapes_b64.txt | cut -c1-125
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/t_gpu.log 2>&1; echo "pytest rc=$?"
tail -8 gpurun_out/t_gpu.log
