#!/bin/bash
# scratch script for one-off GPU experiments (edit, run with gpurun, do not rely on its contents)
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
# phase stamps of the fused tick: needs the lab build (make STAMPS=1 BUILD=build_stamps OUT=../../tools/ab/libmolnextr_hip_stamps.so)
cp molnextr_amd/lib/libmolnextr_hip.so /tmp/mnx_cur.so
cp tools/ab/libmolnextr_hip_stamps.so molnextr_amd/lib/libmolnextr_hip.so
for cfg in "64 250 2 4" "128 250 4 4" "128 250 2 4"; do
  set -- $cfg
  MNX_FUSED_STAMPS=/tmp/st_$1_$3.bin timeout 300 python tools/fused_stamps.py run $1 $2 $3 $4 2>&1 | grep -v amdgpu.ids
  python tools/fused_stamps.py show /tmp/st_$1_$3.bin > gpurun_out/stamps_$1_r$3.txt
done
cp /tmp/mnx_cur.so molnextr_amd/lib/libmolnextr_hip.so
head -36 gpurun_out/stamps_64_r2.txt
