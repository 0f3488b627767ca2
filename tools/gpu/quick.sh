#!/bin/bash
# scratch script for one-off GPU experiments (edit, run with gpurun, do not rely on its contents)
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
for b in 224 448; do
  timeout 200 tools/gemm_lab/lab $b 20 - fp16x3 > gpurun_out/r04_gemm_shapes_fp16x3_b$b.txt 2>&1
  grep -c FAIL gpurun_out/r04_gemm_shapes_fp16x3_b$b.txt
  python tools/gemm_shapes_report.py gpurun_out/r04_gemm_shapes_fp16x3_b$b.txt | tail -1
done
