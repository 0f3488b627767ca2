#!/bin/bash
# round 5, GPU call 7: 128x128 kernel with all residual loads of an epilogue pass in flight at once — correctness + timing
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05_c7; mkdir -p $OUT
export TMPDIR=/tmp
L=tools/gemm_lab
echo "=== bf16 (element check vs fp32 reference), 64 images" | tee $OUT/lab.txt
timeout 300 $L/lab 64 10 - bf16 2>&1 | tee -a $OUT/lab.txt
echo "=== fp16x3, 512 images: base = 128x128 kernel, disp = product dispatch" | tee -a $OUT/lab.txt
timeout 300 $L/lab 512 20 - fp16x3 2>&1 | tee -a $OUT/lab.txt
grep -c "FAIL\|MISMATCH" $OUT/lab.txt | sed 's/^/failures: /'
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm or swin or encoder" 2>&1 | tail -3 | tee $OUT/pytest.txt
cp molnextr_amd/lib/libmolnextr_hip.so /tmp/mnx_cur.so
for r in 1 2; do
  for v in cur prev; do
    if [ $v = cur ]; then cp /tmp/mnx_cur.so molnextr_amd/lib/libmolnextr_hip.so; else cp tools/ab/libmolnextr_hip_prev.so molnextr_amd/lib/libmolnextr_hip.so; fi
    timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sub > $OUT/bench_${v}_$r.log 2>&1
    echo "$v $(tail -1 $OUT/bench_${v}_$r.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)" | tee -a $OUT/bench.txt
  done
done
cp /tmp/mnx_cur.so molnextr_amd/lib/libmolnextr_hip.so
echo done
