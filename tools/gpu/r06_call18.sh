#!/bin/bash
# the default mode on MANY images beyond the fixtures against the CPU oracle (tools/extended_parity.py): 512 images of checkpoint 0,
# 256 of the hostile one
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python tools/extended_parity.py --ckpt 0 --batches 16 --first 10000 --out gpurun_out/r06_extended_parity_0_fp16x3_512.json 2>&1 | grep EXTENDED_PARITY | cut -c1-700
timeout 2400 python tools/extended_parity.py --ckpt stress --batches 8 --first 20000 --out gpurun_out/r06_extended_parity_stress_fp16x3_256.json 2>&1 | grep EXTENDED_PARITY | cut -c1-700
