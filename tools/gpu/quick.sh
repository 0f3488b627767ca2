#!/bin/bash
# scratch script for one-off GPU experiments (edit, run with gpurun, do not rely on its contents)
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fused" > gpurun_out/t_dec.log 2>&1; echo "pytest fused rc=$?"; tail -8 gpurun_out/t_dec.log | cut -c1-600
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline"
run() { # name, args, env...
  n=$1; shift; x=$1; shift
  env "$@" timeout 400 $B $x > gpurun_out/b_$n.log 2>&1
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
run xcd0 "" MNX_DEC_XCD=0
run xcd1 "" MNX_DEC_XCD=1
run xcd0_b "--no-sub" MNX_DEC_XCD=0
run xcd1_b "--no-sub" MNX_DEC_XCD=1
(cd /tmp && env MNX_DEC_XCD=1 timeout 400 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_tick_xcd -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sub > $GRAFT_REPO_ROOT/gpurun_out/prof_tick_xcd.log 2>&1)
DB=$(find gpurun_out/prof_tick_xcd -name "*.db" | head -1)
python tools/tick_profile.py $DB gpurun_out/tick_profile_xcd.txt | head -36
rm -f $DB
