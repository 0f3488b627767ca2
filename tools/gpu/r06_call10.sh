#!/bin/bash
# default mode switched to fp16x3m: the whole GPU suite, smoke(), the driver's bench command (full line), the default bench
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/pixels_parity.json
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r06_c10_suite.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r06_c10_suite.log | cut -c1-300
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_c10_bench20.log 2>&1; echo "bench20 rc=$?"; tail -1 gpurun_out/r06_c10_bench20.log | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline --no-sub > gpurun_out/r06_c10_bench512.log 2>&1; echo "bench default rc=$?"; tail -1 gpurun_out/r06_c10_bench512.log | cut -c1-200
