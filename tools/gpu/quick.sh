#!/bin/bash
# scratch script for one-off GPU experiments (edit, run with gpurun, do not rely on its contents)
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "4-3-2-160" > gpurun_out/t_dec.log 2>&1; echo "pytest beam case rc=$?"; tail -3 gpurun_out/t_dec.log | cut -c1-400
