#!/bin/bash
# scratch: a handful of tests + a short bench (edit as needed)
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "predict_beam or beam_matches or pipeline_facade" > gpurun_out/t_quick.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/t_quick.log
timeout 300 python bench.py --beam 5 --steps 6 --warmup 1 --no-sub --no-cpu-baseline > gpurun_out/bench_beam.log 2>&1; echo "bench rc=$?"
tail -1 gpurun_out/bench_beam.log | cut -c1-400
