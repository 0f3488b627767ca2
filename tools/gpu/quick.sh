#!/bin/bash
# scratch script for one-off GPU experiments (edit, run with gpurun, do not rely on its contents)
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
cp molnextr_amd/lib/libmolnextr_hip.so /tmp/mnx_cur.so
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sub"
run() { # name, lib, env
  n=$1; cp $2 molnextr_amd/lib/libmolnextr_hip.so; shift; shift
  env "$@" timeout 400 $B > gpurun_out/b_$n.log 2>&1
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/b_$n.log").read().strip().splitlines()[-1])
    print("$n", d["value"], "mol/s", d["ms_per_step"], "ms/step")
except Exception as e:
    print("$n FAILED", e)
PY
}
run cur_auto /tmp/mnx_cur.so A=1
run occ4_r2 tools/ab/libmolnextr_hip_prev.so MNX_DEC_TILE=2
run cur_r2 /tmp/mnx_cur.so MNX_DEC_TILE=2
run occ4_r2_256 tools/ab/libmolnextr_hip_prev.so MNX_DEC_TILE=2 MNX_DEC_FUSED_MAX=256
run occ4_auto tools/ab/libmolnextr_hip_prev.so A=1
cp tools/ab/libmolnextr_hip_prev.so molnextr_amd/lib/libmolnextr_hip.so
(cd /tmp && env MNX_DEC_TILE=2 MNX_DEC_FUSED_MAX=256 timeout 400 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_tick_x -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sub > $GRAFT_REPO_ROOT/gpurun_out/prof_tick_x.log 2>&1)
cp /tmp/mnx_cur.so molnextr_amd/lib/libmolnextr_hip.so
DB=$(find gpurun_out/prof_tick_x -name "*.db" | head -1)
python tools/tick_profile.py $DB gpurun_out/tick_profile_x.txt | head -6
grep -n "rows_cap 128:" -A5 gpurun_out/tick_profile_x.txt
rm -f $DB
