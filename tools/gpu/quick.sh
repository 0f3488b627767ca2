#!/bin/bash
# scratch script for one-off GPU experiments (edit, run with gpurun, do not rely on its contents)
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fused or greedy or decode or predict or chunk or beam1 or end_to_end or pipeline" > gpurun_out/t_dec.log 2>&1; echo "pytest decoder subset rc=$?"; tail -12 gpurun_out/t_dec.log | cut -c1-600
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sub"
run() { # name, env...
  n=$1; shift
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
run f128 MNX_DEC_FUSED_MAX=128
run unfused MNX_DEC_TILE=0
run f256 MNX_DEC_FUSED_MAX=256
run f512 MNX_DEC_FUSED_MAX=512
run f1024 MNX_DEC_FUSED_MAX=1024
run f128_b MNX_DEC_FUSED_MAX=128
run f512_ff8 MNX_DEC_FUSED_MAX=512 MNX_DEC_TILE_FF=8
(cd /tmp && env MNX_DEC_FUSED_MAX=4096 timeout 400 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_tick_all -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sub > $GRAFT_REPO_ROOT/gpurun_out/prof_tick_all.log 2>&1)
DB=$(find gpurun_out/prof_tick_all -name "*.db" | head -1)
python tools/tick_profile.py $DB gpurun_out/tick_profile_all.txt | head -12
rm -f $DB
