#!/bin/bash
# scratch script for one-off GPU experiments (edit, run with gpurun, do not rely on its contents)
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 60 tools/probes/mfma_f16_subnormal > gpurun_out/subnormal_probe.txt 2>&1; echo "probe rc=$?"; tail -3 gpurun_out/subnormal_probe.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "split_operand or swin_tiny or native_library or swin_full or batch32" > gpurun_out/t_split.log 2>&1; echo "pytest split rc=$?"; tail -15 gpurun_out/t_split.log
timeout 900 python -m pytest tests/test_gpu_pixels.py -x -q -m gpu > gpurun_out/t_pixels.log 2>&1; echo "pytest pixels rc=$?"; tail -25 gpurun_out/t_pixels.log | cut -c1-600
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench20.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench20.log | cut -c1-1500
timeout 200 tools/gemm_lab/lab 128 10 - fp16x3 > gpurun_out/gemm_shapes_fp16x3_b128.txt 2>&1; cat gpurun_out/gemm_shapes_fp16x3_b128.txt | cut -c1-110
