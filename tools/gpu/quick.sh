#!/bin/bash
# scratch script for one-off GPU experiments (edit, run with gpurun, do not rely on its contents)
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fused or greedy or decode or predict or chunk or beam1 or end_to_end or pipeline" > gpurun_out/t_dec.log 2>&1; echo "pytest decoder subset rc=$?"; tail -15 gpurun_out/t_dec.log | cut -c1-600
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline"
run() { # name, extra bench args, env...
  n=$1; shift; x=$1; shift
  env "$@" timeout 400 $B $x > gpurun_out/b_$n.log 2>&1
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/b_$n.log").read().strip().splitlines()[-1])
    s = d.get("sub_results") or {}
    print("$n", d["value"], "mol/s", d["ms_per_step"], "ms/step", {k: v.get("ms_per_batch", v.get("molecules_per_s")) for k, v in s.items()})
except Exception as e:
    print("$n FAILED", e)
PY
}
run auto128 "" MNX_DEC_TILE=-1
run unfused "" MNX_DEC_TILE=0
run auto128b --no-sub MNX_DEC_TILE=-1
run unfused_b --no-sub MNX_DEC_TILE=0
run auto256 --no-sub MNX_DEC_FUSED_MAX=256
run auto64 --no-sub MNX_DEC_FUSED_MAX=64
run r2_128 --no-sub MNX_DEC_TILE=2
run auto192_ff8 --no-sub MNX_DEC_FUSED_MAX=192 MNX_DEC_TILE_FF=8
for cfg in "all MNX_DEC_FUSED_MAX=4096"; do
  set -- $cfg
  (cd /tmp && env ${2//,/ } timeout 400 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_tick_$1 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sub > $GRAFT_REPO_ROOT/gpurun_out/prof_tick_$1.log 2>&1)
  DB=$(find gpurun_out/prof_tick_$1 -name "*.db" | head -1)
  python tools/tick_profile.py $DB gpurun_out/tick_profile_$1.txt | head -12
  rm -f $DB
done
