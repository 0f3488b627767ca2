#!/bin/bash
# scratch script for one-off GPU experiments (edit, run with gpurun, do not rely on its contents)
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fused or greedy or decode or predict or chunk or beam1 or end_to_end or pipeline" > gpurun_out/t_dec.log 2>&1; echo "pytest decoder subset rc=$?"; tail -4 gpurun_out/t_dec.log | cut -c1-400
cp molnextr_amd/lib/libmolnextr_hip.so /tmp/mnx_cur.so
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline"
run() { # name, lib, args
  n=$1; cp $2 molnextr_amd/lib/libmolnextr_hip.so
  timeout 400 $B $3 > gpurun_out/b_$n.log 2>&1
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/b_$n.log").read().strip().splitlines()[-1])
    s = d.get("sub_results") or {}
    print("$n", d["value"], "mol/s", d["ms_per_step"], "ms/step", (s.get("latency_mode") or {}).get("ms_per_batch"))
except Exception as e:
    print("$n FAILED", e)
PY
}
run cur /tmp/mnx_cur.so ""
run prev tools/ab/libmolnextr_hip_prev.so ""
run cur_b /tmp/mnx_cur.so "--no-sub"
run prev_b tools/ab/libmolnextr_hip_prev.so "--no-sub"
cp /tmp/mnx_cur.so molnextr_amd/lib/libmolnextr_hip.so
(cd /tmp && timeout 400 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_tick_x -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sub > $GRAFT_REPO_ROOT/gpurun_out/prof_tick_x.log 2>&1)
DB=$(find gpurun_out/prof_tick_x -name "*.db" | head -1)
python tools/tick_profile.py $DB gpurun_out/tick_profile_x.txt | head -4
rm -f $DB
