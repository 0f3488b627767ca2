#!/bin/bash
# patch embedding with three patches per thread (encoder.hip): bit-identity of the encoder output against the previous
# library (tools/gpu/ab_build.sh HEAD~ -> tools/ab/libmolnextr_hip_prev.so), then the bench A/B of tools/gpu/ab_run.sh
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
cp molnextr_amd/lib/libmolnextr_hip.so /tmp/mnx_cur.so
timeout 300 python tools/features_hash.py 2>&1 | grep sha256 | sed "s/^/cur  /"
cp tools/ab/libmolnextr_hip_prev.so molnextr_amd/lib/libmolnextr_hip.so
timeout 300 python tools/features_hash.py 2>&1 | grep sha256 | sed "s/^/prev /"
cp /tmp/mnx_cur.so molnextr_amd/lib/libmolnextr_hip.so
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "swin_full or encoder_batch32 or end_to_end or greedy_decode or beam1 or natural_lengths or bond_head" 2>&1 | tail -2
bash tools/gpu/ab_run.sh
