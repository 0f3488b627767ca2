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
    print("$n FAILED", e, open("gpurun_out/b_$n.log").read()[-400:])
PY
}
run br128x4 MNX_DEC_BRANCH_ROWS=128
run br0 MNX_DEC_BRANCH_ROWS=0
run br128x4_b MNX_DEC_BRANCH_ROWS=128
run br0_b MNX_DEC_BRANCH_ROWS=0
run br128x8 MNX_DEC_BRANCH_ROWS=128 MNX_DEC_BRANCH_MAX=8
run br128x2 MNX_DEC_BRANCH_ROWS=128 MNX_DEC_BRANCH_MAX=2
run br64x8 MNX_DEC_BRANCH_ROWS=64 MNX_DEC_BRANCH_MAX=8
run br256x4_old MNX_DEC_BRANCH_ROWS=256 MNX_DEC_BRANCH_MAX=4
(cd /tmp && timeout 400 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_tick_br -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sub > $GRAFT_REPO_ROOT/gpurun_out/prof_tick_br.log 2>&1)
DB=$(find gpurun_out/prof_tick_br -name "*.db" | head -1)
python tools/tick_profile.py $DB gpurun_out/tick_profile_br.txt | head -12
rm -f $DB
