#!/bin/bash
# round 5, GPU call 11: the whole GPU suite with every greedy tick on the fused / mid arithmetic (MNX_DEC_MID_MAX=4096), and a
# cleaner A/B of the polynomial GELU (40 launches, alternating three times)
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05_c11; mkdir -p $OUT
export TMPDIR=/tmp
MNX_DEC_MID_MAX=4096 timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee $OUT/pytest_mid4096.txt
L=tools/gemm_lab
for r in 1 2 3; do for b in lab_e0 lab_gelu; do
  echo "=== $b run $r" | tee -a $OUT/lab.txt
  MNX_LAB_NOBASE=1 timeout 300 $L/$b 512 40 "fc1 s2,fc1 s3,fc1 s1" fp16x3 2>&1 | grep -E "^fc1|stamps" | tee -a $OUT/lab.txt
done; done
echo done
