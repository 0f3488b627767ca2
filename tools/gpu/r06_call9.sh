#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_pixels.py -q -m gpu -k "two_term_tables" > gpurun_out/r06_c9_tables.log 2>&1
tail -5 gpurun_out/r06_c9_tables.log
