#!/bin/bash
# which op class of the stage-3 two-term table costs the logit error on images beyond the fixtures (tools/extended_parity.py --two)
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
for two in fc1.s2 fc2.s2 qkv.s2 qkv.s2,fc1.s2 qkv.s2,fc2.s2 fc1.s2,fc2.s2; do
  for ck in stress 0; do
    f=$([ $ck = stress ] && echo 2000 || echo 1000)
    echo "== $two ckpt $ck"
    timeout 600 python tools/extended_parity.py --ckpt $ck --batches 2 --first $f --two $two 2>&1 | grep EXTENDED_PARITY | cut -c1-600
  done
done | tee gpurun_out/r06_c16_two_term_classes.txt
