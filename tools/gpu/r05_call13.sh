#!/bin/bash
# round 5, GPU call 13: window attention softmax in the log2 domain (MNX_ATTN_EXP2) — attn_lab timing + bit-compare of the two split
# kernels, encoder / from-pixels tests, bench A/B
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05_c13; mkdir -p $OUT
export TMPDIR=/tmp
for st in 1 2 3 4; do for b in lab_exp0 lab; do
  echo "=== $b stage $st" | tee -a $OUT/attn_lab.txt
  timeout 200 tools/attn_lab/$b 224 $st 20 2>&1 | grep -E "differ|us / launch" | tee -a $OUT/attn_lab.txt
done; done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "swin or encoder or end_to_end or grouped or persistent" 2>&1 | tail -3 | tee $OUT/pytest.txt
timeout 600 python -m pytest tests/test_gpu_pixels.py -x -q -m gpu 2>&1 | tail -3 | tee -a $OUT/pytest.txt
cp gpurun_out/pixels_parity.json $OUT/pixels_parity.json 2>/dev/null
cp molnextr_amd/lib/libmolnextr_hip.so /tmp/mnx_cur.so
for r in 1 2 3; do
  for v in cur prev; do
    if [ $v = cur ]; then cp /tmp/mnx_cur.so molnextr_amd/lib/libmolnextr_hip.so; else cp tools/ab/libmolnextr_hip_prev.so molnextr_amd/lib/libmolnextr_hip.so; fi
    timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sub > $OUT/bench_${v}_$r.log 2>&1
    echo "$v $(tail -1 $OUT/bench_${v}_$r.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], [ (e['measured'], e['achieved']) for e in d['roofline_extra'] if 'window' in e['kernel']])" 2>&1 | tail -1)" | tee -a $OUT/bench.txt
  done
done
cp /tmp/mnx_cur.so molnextr_amd/lib/libmolnextr_hip.so
echo done
