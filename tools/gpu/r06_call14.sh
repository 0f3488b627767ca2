#!/bin/bash
# window attention, second step: per-lane byte offsets from LDS tables in the unmasked kernel (encoder.hip): word compare against the
# per-item kernel at every stage (tools/attn_lab), output digests against the previous library, bench A/B on this box
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
for st in 1 2 3 4; do timeout 200 tools/attn_lab/lab 512 $st 10 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06_c14_attn_lab.txt
cat gpurun_out/r06_c14_attn_lab.txt | cut -c1-200
cp molnextr_amd/lib/libmolnextr_hip.so /tmp/mnx_cur.so
timeout 300 python tools/features_hash.py 2>&1 | grep sha256 | sed "s/^/cur  /" | tee gpurun_out/r06_c14_hash.txt
cp tools/ab/libmolnextr_hip_prev.so molnextr_amd/lib/libmolnextr_hip.so
timeout 300 python tools/features_hash.py 2>&1 | grep sha256 | sed "s/^/prev /" | tee -a gpurun_out/r06_c14_hash.txt
cp /tmp/mnx_cur.so molnextr_amd/lib/libmolnextr_hip.so
bash tools/gpu/ab_run.sh 2>&1 | tee gpurun_out/r06_c14_ab.txt
