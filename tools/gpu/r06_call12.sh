#!/bin/bash
# which piece of the window attention's instruction diet breaks the word compare: tools/attn_lab built with -DMNX_ATTN_MIX=<m>
# -DMNX_ATTN_PK=<p> into lab_m<m>_p<p> (record of the call; the two switches and the packed-fp32 paths they selected were removed from
# encoder.hip after this measurement, profiles/r06_attn_lab_diet.txt has the output)
cd /root/repo
mkdir -p gpurun_out
for v in m0_p0 m1_p0 m0_p1 m0_p2 m1_p3 m0_p0 m1_p0 m0_p1 m0_p2 m1_p3; do
  echo "== $v"
  timeout 200 tools/attn_lab/lab_$v 128 3 3 2>&1 | grep -v amdgpu.ids | grep "differ\|first at\|dispatched"
done > gpurun_out/r06_c12_attn_bisect.txt 2>&1
cat gpurun_out/r06_c12_attn_bisect.txt | cut -c1-220
