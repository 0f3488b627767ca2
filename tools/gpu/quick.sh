#!/bin/bash
# scratch script for one-off GPU experiments (edit, run with gpurun, do not rely on its contents)
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/pixels_parity.json
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/t_gpu.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/t_gpu.log | cut -c1-300
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
