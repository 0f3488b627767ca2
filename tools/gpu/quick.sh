#!/bin/bash
# scratch script for one-off GPU experiments (edit, run with gpurun, do not rely on its contents)
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
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
run pre A=1
run nopre MNX_NO_PRECAPTURE=1
run pre_b A=1
run nopre_b MNX_NO_PRECAPTURE=1
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "predict or pipeline or fused" > gpurun_out/t_dec.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/t_dec.log
