#!/bin/bash
# round 5, GPU call 4: the mid tick form (dec_ma + dec_mb) — bit-equality tests, tick time by capacity and form, bench A/B
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05_c4; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused or greedy or decode or predict or pipeline or chunk" 2>&1 | tail -8 | tee $OUT/pytest_dec.txt
timeout 600 python tools/tick_time.py 64,128,192,256,384,512,640 2>&1 | tail -12 | tee $OUT/tick_time.txt
bench() { local label=$1; shift
    env "$@" timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sub > $OUT/bench_$label.log 2>&1
    echo "$label $(tail -1 $OUT/bench_$label.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)" | tee -a $OUT/bench.txt
}
for r in 1 2; do
  bench mid640_$r MNX_DEC_MID_MAX=640
  bench nomid_$r MNX_DEC_MID_MAX=0
  bench mid1024_$r MNX_DEC_MID_MAX=1024
  bench mid384_$r MNX_DEC_MID_MAX=384
done
bench mid4096_1 MNX_DEC_MID_MAX=4096
env MNX_DEC_MID_MAX=640 timeout 600 python bench.py --no-cpu-baseline --no-sub > $OUT/bench512_mid640.log 2>&1; echo "512 steps mid640 $(tail -1 $OUT/bench512_mid640.log | cut -c1-120)" | tee -a $OUT/bench.txt
env MNX_DEC_MID_MAX=0 timeout 600 python bench.py --no-cpu-baseline --no-sub > $OUT/bench512_nomid.log 2>&1; echo "512 steps nomid $(tail -1 $OUT/bench512_nomid.log | cut -c1-120)" | tee -a $OUT/bench.txt
env MNX_DEC_MID_MAX=4096 timeout 600 python bench.py --no-cpu-baseline --no-sub > $OUT/bench512_mid4096.log 2>&1; echo "512 steps mid4096 $(tail -1 $OUT/bench512_mid4096.log | cut -c1-120)" | tee -a $OUT/bench.txt
env MNX_ENC_CUS=224 timeout 600 python bench.py --no-cpu-baseline --no-sub > $OUT/bench512_cus224.log 2>&1; echo "512 steps enc_cus224 $(tail -1 $OUT/bench512_cus224.log | cut -c1-120)" | tee -a $OUT/bench.txt
echo done
