#!/bin/bash
# full default-contract bench runs (what the driver executes), plus smoke()
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/bench_full20.log 2>&1; echo "bench20 rc=$?"
tail -3 gpurun_out/bench_full20.log | cut -c1-300
( time timeout 900 python bench.py ) > gpurun_out/bench_default.log 2>&1; echo "bench default rc=$?"
tail -3 gpurun_out/bench_default.log | cut -c1-300
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
