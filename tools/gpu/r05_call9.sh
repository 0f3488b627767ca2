#!/bin/bash
# round 5, GPU call 9: LayerNorm with all loads of a row in flight (A/B), MNX_ENC_CUS invariance test, 128-row kernel tile widths at stage 1
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05_c9; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "encoder or swin or persistent or end_to_end or grouped" 2>&1 | tail -3 | tee $OUT/pytest.txt
timeout 600 python -m pytest tests/test_gpu_pixels.py -x -q -m gpu 2>&1 | tail -3 | tee -a $OUT/pytest.txt
L=tools/gemm_lab
for b in lab lab_bn64 lab_bn128; do
  echo "=== $b 512 images" | tee -a $OUT/lab.txt
  MNX_LAB_NOBASE=0 timeout 300 $L/$b 512 20 "qkv s0,proj s0,fc2 s0,qkv s1,merge s0" fp16x3 2>&1 | tee -a $OUT/lab.txt
done
cp molnextr_amd/lib/libmolnextr_hip.so /tmp/mnx_cur.so
for r in 1 2 3; do
  for v in cur prev; do
    if [ $v = cur ]; then cp /tmp/mnx_cur.so molnextr_amd/lib/libmolnextr_hip.so; else cp tools/ab/libmolnextr_hip_prev.so molnextr_amd/lib/libmolnextr_hip.so; fi
    timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sub > $OUT/bench_${v}_$r.log 2>&1
    echo "$v $(tail -1 $OUT/bench_${v}_$r.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], [ (e['measured'], e['achieved']) for e in d['roofline_extra'] if 'layernorm' in e['kernel']])" 2>&1 | tail -1)" | tee -a $OUT/bench.txt
  done
done
cp /tmp/mnx_cur.so molnextr_amd/lib/libmolnextr_hip.so
echo done
