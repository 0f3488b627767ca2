#!/bin/bash
# round 5, GPU call 6: full GPU suite on the current build + encoder launch-group sizes (tile-count quantisation: 512 images
# give every stage-3 layer a whole number of rounds of 256 tiles)
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05_c6; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.txt
bench() { local label=$1; shift
    timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sub "$@" > $OUT/bench_$label.log 2>&1
    echo "$label $(tail -1 $OUT/bench_$label.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['clock'])" 2>&1 | tail -1)" | tee -a $OUT/bench.txt
}
for r in 1 2; do
  bench eb448_$r --encode-batch 448
  bench eb512_$r --encode-batch 512
  bench eb640_$r --encode-batch 640
  bench eb320_$r --encode-batch 320
done
for eb in 448 512 1024; do
  timeout 600 python bench.py --no-cpu-baseline --no-sub --encode-batch $eb > $OUT/bench512_eb$eb.log 2>&1; echo "512 steps eb$eb $(tail -1 $OUT/bench512_eb$eb.log | cut -c1-130)" | tee -a $OUT/bench.txt
done
echo done
