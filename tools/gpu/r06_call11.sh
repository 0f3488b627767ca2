#!/bin/bash
# window attention: split of P by v_cvt_pk_f16_f32 + v_fma_mix{lo,hi}_f16, packed fp32 scores / exponent arguments (encoder.hip):
# word compare of the dispatched kernel against the per-item kernel (tools/attn_lab), digests of the library's outputs against
# the previous library's (tools/gpu/ab_build.sh HEAD), bench A/B on this box
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
for st in 1 2 3 4; do timeout 200 tools/attn_lab/lab 512 $st 10 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06_c11_attn_lab.txt
cat gpurun_out/r06_c11_attn_lab.txt | cut -c1-200
cp molnextr_amd/lib/libmolnextr_hip.so /tmp/mnx_cur.so
timeout 300 python tools/features_hash.py 2>&1 | grep sha256 | sed "s/^/cur  /" | tee gpurun_out/r06_c11_hash.txt
cp tools/ab/libmolnextr_hip_prev.so molnextr_amd/lib/libmolnextr_hip.so
timeout 300 python tools/features_hash.py 2>&1 | grep sha256 | sed "s/^/prev /" | tee -a gpurun_out/r06_c11_hash.txt
cp /tmp/mnx_cur.so molnextr_amd/lib/libmolnextr_hip.so
bash tools/gpu/ab_run.sh 2>&1 | tee gpurun_out/r06_c11_ab.txt
