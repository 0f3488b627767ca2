#!/bin/bash
# round 6, call 1: the two-term GEMM forms + FP16X3M from pixels (gates), then same-box bench A/B fp16x3 vs fp16x3m
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm_two_term or gemm_split_operand or native_library or persistent_residual or swin_tiny or batch_invariant" > gpurun_out/r06_c1_gemm.log 2>&1
tail -5 gpurun_out/r06_c1_gemm.log
timeout 1200 python -m pytest tests/test_gpu_pixels.py -x -q -m gpu -k "fp16x3" > gpurun_out/r06_c1_pixels.log 2>&1
tail -8 gpurun_out/r06_c1_pixels.log
for i in 1 2; do
  for dt in fp16x3 fp16x3m; do
    timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sub --dtype $dt > gpurun_out/r06_c1_bench_$dt.log 2>&1
    echo "$dt $(tail -1 gpurun_out/r06_c1_bench_$dt.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['mfma_terms'], r['clock']['gemm_shader_mhz_live'])")"
  done
done
for dt in fp16x3 fp16x3m; do
  timeout 600 python bench.py --gpus 1 --no-cpu-baseline --no-sub --dtype $dt > gpurun_out/r06_c1_bench512_$dt.log 2>&1
  echo "512 $dt $(tail -1 gpurun_out/r06_c1_bench512_$dt.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['achieved'], r['frac'])")"
done
