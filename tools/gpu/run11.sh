#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "facade or public_api or preprocess or crop or eval_harness or checkpoint" > gpurun_out/t_facade.log 2>&1; echo "facade rc=$?"
tail -5 gpurun_out/t_facade.log
