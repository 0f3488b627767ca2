#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05_c5; mkdir -p $OUT
export TMPDIR=/tmp
tools/probes/mfma_power 2>&1 | tee $OUT/mfma_power.txt
for c in 240 208 192; do
  env MNX_ENC_CUS=$c timeout 600 python bench.py --no-cpu-baseline --no-sub > $OUT/bench512_cus$c.log 2>&1; echo "512 steps enc_cus$c $(tail -1 $OUT/bench512_cus$c.log | cut -c1-130)" | tee -a $OUT/bench.txt
done
timeout 600 python bench.py --no-cpu-baseline --no-sub > $OUT/bench512_cus256.log 2>&1; echo "512 steps enc_cus256 $(tail -1 $OUT/bench512_cus256.log | cut -c1-130)" | tee -a $OUT/bench.txt
env MNX_ENC_CUS=224 timeout 600 python bench.py --no-cpu-baseline --no-sub > $OUT/bench512_cus224.log 2>&1; echo "512 steps enc_cus224 $(tail -1 $OUT/bench512_cus224.log | cut -c1-130)" | tee -a $OUT/bench.txt
echo done
