#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "several_reference_batches or persistent_encoder_grids" > gpurun_out/r06_c6_beam.log 2>&1
tail -4 gpurun_out/r06_c6_beam.log
for g in 4 6 8; do
  MNX_BEAM_GROUPS=$g timeout 600 python bench.py --gpus 1 --beam 5 --steps 32 --warmup 8 --no-cpu-baseline --no-sub > gpurun_out/r06_c6_bench_beam_g$g.log 2>&1
  echo "groups $g: $(tail -1 gpurun_out/r06_c6_bench_beam_g$g.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
done
