#!/bin/bash
# patch embedding with two patch rows per workgroup: output digests against the previous library (tools/gpu/ab_build.sh HEAD) and the
# kernel's live / isolated fraction of 8 TB/s from bench.py's roofline_extra, alternating on this box
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
cp molnextr_amd/lib/libmolnextr_hip.so /tmp/mnx_cur.so
timeout 300 python tools/features_hash.py 2>&1 | grep sha256 | sed "s/^/cur  /" | tee gpurun_out/r06_c17_hash.txt
cp tools/ab/libmolnextr_hip_prev.so molnextr_amd/lib/libmolnextr_hip.so
timeout 300 python tools/features_hash.py 2>&1 | grep sha256 | sed "s/^/prev /" | tee -a gpurun_out/r06_c17_hash.txt
for i in 1 2; do
  for v in cur prev; do
    if [ $v = cur ]; then cp /tmp/mnx_cur.so molnextr_amd/lib/libmolnextr_hip.so; else cp tools/ab/libmolnextr_hip_prev.so molnextr_amd/lib/libmolnextr_hip.so; fi
    timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sub > gpurun_out/bench_ab_$v.log 2>&1
    echo "$v $(tail -1 gpurun_out/bench_ab_$v.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
pe=[e for e in d['roofline_extra'] if 'patch_embed' in e['kernel']]
print(d['value'], d['ms_per_step'], ' '.join(f\"{e['measured'][:8]} {e['frac']} {e['avg_launch_us']}us\" for e in pe))")"
  done
done | tee gpurun_out/r06_c17_ab.txt
cp /tmp/mnx_cur.so molnextr_amd/lib/libmolnextr_hip.so
