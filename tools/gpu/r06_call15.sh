#!/bin/bash
# from-pixels parity of the default mode (and of fp16x3) on images beyond the fixtures, against the CPU oracle (tools/extended_parity.py)
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python tools/extended_parity.py --ckpt 0 --batches 8 --first 1000 --out gpurun_out/r06_extended_parity_0_fp16x3m.json 2>&1 | grep -v amdgpu.ids | tail -10
timeout 1500 python tools/extended_parity.py --ckpt stress --batches 4 --first 2000 --out gpurun_out/r06_extended_parity_stress_fp16x3m.json 2>&1 | grep -v amdgpu.ids | tail -6
timeout 1500 python tools/extended_parity.py --ckpt 0 --batches 2 --first 1000 --dtype fp16x3 --out gpurun_out/r06_extended_parity_0_fp16x3.json 2>&1 | grep -v amdgpu.ids | tail -4
