#!/bin/bash
# round 6, call 2: two-term tables measured on both checkpoints, FP16X3M gates from pixels, RCCL on one rank, bench A/B
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pixels.py -q -m gpu -k "two_term_tables" > gpurun_out/r06_c2_tables.log 2>&1
tail -5 gpurun_out/r06_c2_tables.log
timeout 1200 python -m pytest tests/test_gpu_pixels.py -q -m gpu -k "fp16x3m" > gpurun_out/r06_c2_pixels.log 2>&1
tail -6 gpurun_out/r06_c2_pixels.log
timeout 1200 python -m pytest tests/test_gpu_rccl.py -q -m gpu > gpurun_out/r06_c2_rccl.log 2>&1
tail -6 gpurun_out/r06_c2_rccl.log
for i in 1 2; do
  for dt in fp16x3 fp16x3m; do
    timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sub --dtype $dt > gpurun_out/r06_c2_bench_$dt.log 2>&1
    echo "$dt $(tail -1 gpurun_out/r06_c2_bench_$dt.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['mfma_terms'], r['clock']['gemm_shader_mhz_live'])")"
  done
done
for dt in fp16x3 fp16x3m; do
  timeout 600 python bench.py --gpus 1 --no-cpu-baseline --no-sub --dtype $dt > gpurun_out/r06_c2_bench512_$dt.log 2>&1
  echo "512 $dt $(tail -1 gpurun_out/r06_c2_bench512_$dt.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['achieved'], r['frac'])")"
done
