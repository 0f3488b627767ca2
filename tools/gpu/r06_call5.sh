#!/bin/bash
# round 6, call 5: beam search of several reference batches per step sequence
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "beam" > gpurun_out/r06_c5_beam.log 2>&1
tail -6 gpurun_out/r06_c5_beam.log
for g in 1 2 4; do
  MNX_BEAM_GROUPS=$g timeout 600 python bench.py --gpus 1 --beam 5 --steps 16 --warmup 4 --no-cpu-baseline --no-sub > gpurun_out/r06_c5_bench_beam_g$g.log 2>&1
  echo "groups $g: $(tail -1 gpurun_out/r06_c5_bench_beam_g$g.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
done
