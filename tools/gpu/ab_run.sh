#!/bin/bash
# GPU side of tools/gpu/ab_build.sh: alternate the current library and tools/ab/libmolnextr_hip_prev.so on this box
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
ARGS=${*:---gpus 1 --steps 20 --warmup 5}
[ -f tools/ab/libmolnextr_hip_prev.so ] || { echo "run tools/gpu/ab_build.sh <git-ref> first"; exit 1; }
cp molnextr_amd/lib/libmolnextr_hip.so /tmp/mnx_cur.so
for i in 1 2; do
  for v in cur prev; do
    if [ $v = cur ]; then cp /tmp/mnx_cur.so molnextr_amd/lib/libmolnextr_hip.so; else cp tools/ab/libmolnextr_hip_prev.so molnextr_amd/lib/libmolnextr_hip.so; fi
    timeout 400 python bench.py $ARGS --no-cpu-baseline --no-sub > gpurun_out/bench_ab_$v.log 2>&1
    echo "$v $(tail -1 gpurun_out/bench_ab_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
  done
done
cp /tmp/mnx_cur.so molnextr_amd/lib/libmolnextr_hip.so
