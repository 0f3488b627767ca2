#!/bin/bash
# round 5, GPU call 3: fp32 epilogue form 2 — correctness (lab word compare, GEMM / encoder GPU tests), A/B on the bench,
# and MNX_ENC_CUS (persistent encoder grids leave CUs to the decode stream)
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05_c3; mkdir -p $OUT
export TMPDIR=/tmp
L=tools/gemm_lab
ALL32="proj s1,fc2 s1,merge s1,proj s2,fc2 s2,merge s2,proj s3,fc2 s3,merge s0"
run() { local name=$1 bin=$2 shapes=$3 it=$4; shift 4
    echo "=== $name: $* $bin 448 $it [$shapes]" | tee -a $OUT/lab.txt
    env "$@" timeout 200 $L/$bin 448 $it "$shapes" fp16x3 2>&1 | tee -a $OUT/lab.txt; }
run e1 lab_e1 "$ALL32" 10 MNX_LAB_NOBASE=1
for r in 1 2; do for e in e0 e2; do run ${e}_r$r lab_$e "$ALL32" 30 MNX_LAB_NOBASE=1; done; done
grep -c MISMATCH $OUT/lab.txt | sed 's/^/lab mismatching shapes: /'
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm or swin or encoder or end_to_end or pipeline_equals" 2>&1 | tail -5 | tee $OUT/pytest_gemm.txt
timeout 600 python -m pytest tests/test_gpu_pixels.py -x -q -m gpu 2>&1 | tail -5 | tee $OUT/pytest_pixels.txt
bench() { # label, lib, env...
    local label=$1 lib=$2; shift 2
    cp $lib molnextr_amd/lib/libmolnextr_hip.so
    env "$@" timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sub > $OUT/bench_$label.log 2>&1
    echo "$label $(tail -1 $OUT/bench_$label.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('stage34',{}).get('frac_of_peak_executed'))" 2>&1 | tail -1)" | tee -a $OUT/bench.txt
}
cp molnextr_amd/lib/libmolnextr_hip.so /tmp/mnx_cur.so
for r in 1 2; do
  bench cur256_$r /tmp/mnx_cur.so MNX_ENC_CUS=256
  bench prev_$r tools/ab/libmolnextr_hip_prev.so
  bench cur224_$r /tmp/mnx_cur.so MNX_ENC_CUS=224
  bench cur208_$r /tmp/mnx_cur.so MNX_ENC_CUS=208
done
bench cur240_1 /tmp/mnx_cur.so MNX_ENC_CUS=240
cp /tmp/mnx_cur.so molnextr_amd/lib/libmolnextr_hip.so
echo done
