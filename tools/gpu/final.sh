#!/bin/bash
# round-end pass: the whole GPU suite, smoke(), the driver's bench command (full line, with sub-results and the CPU baseline),
# the default bench, then the rocprofv3 profiles of tools/gpu/profiles.sh
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/t_gpu.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/t_gpu.log | cut -c1-300
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_full20.log 2>&1; echo "bench20 rc=$?"; tail -1 gpurun_out/bench_full20.log | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline --no-sub > gpurun_out/bench_default.log 2>&1; echo "bench default rc=$?"; tail -1 gpurun_out/bench_default.log | cut -c1-200
bash tools/gpu/profiles.sh
