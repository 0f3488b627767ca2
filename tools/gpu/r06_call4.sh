#!/bin/bash
# round 6, call 4: 24-bit block fixed point K / V cache — decoder parity, tick times, same-box A/B against the fp32 cache
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not gemm" > gpurun_out/r06_c4_parity.log 2>&1
tail -6 gpurun_out/r06_c4_parity.log
timeout 900 python -m pytest tests/test_gpu_pixels.py -x -q -m gpu -k "fp16x3 and not tables and not one_term and not budget" > gpurun_out/r06_c4_pixels.log 2>&1
tail -4 gpurun_out/r06_c4_pixels.log
cp molnextr_amd/lib/libmolnextr_hip.so /tmp/mnx_cur.so
for v in cur prev; do
  if [ $v = cur ]; then cp /tmp/mnx_cur.so molnextr_amd/lib/libmolnextr_hip.so; else cp tools/ab/libmolnextr_hip_prev.so molnextr_amd/lib/libmolnextr_hip.so; fi
  timeout 900 python tools/tick_time.py 64,128,256,384,640,1024 unfused,fused > gpurun_out/r06_c4_tick_$v.txt 2>&1
  echo "== $v"; tail -8 gpurun_out/r06_c4_tick_$v.txt
done
for i in 1 2; do
  for v in cur prev; do
    if [ $v = cur ]; then cp /tmp/mnx_cur.so molnextr_amd/lib/libmolnextr_hip.so; else cp tools/ab/libmolnextr_hip_prev.so molnextr_amd/lib/libmolnextr_hip.so; fi
    timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sub > gpurun_out/r06_c4_bench_$v.log 2>&1
    echo "$v 20: $(tail -1 gpurun_out/r06_c4_bench_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
    timeout 600 python bench.py --gpus 1 --no-cpu-baseline --no-sub > gpurun_out/r06_c4_bench512_$v.log 2>&1
    echo "$v 512: $(tail -1 gpurun_out/r06_c4_bench512_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
  done
done
cp /tmp/mnx_cur.so molnextr_amd/lib/libmolnextr_hip.so
