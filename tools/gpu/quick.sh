#!/bin/bash
# scratch script for one-off GPU experiments (edit, run with gpurun, do not rely on its contents)
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pixels.py -q -m gpu -k "fused or greedy or decode or predict or chunk or beam1 or end_to_end or pipeline or stress or falls_back or range_flag or gemm_split or persistent_256 or gemm_all" > gpurun_out/t_dec.log 2>&1; echo "pytest subset rc=$?"; tail -25 gpurun_out/t_dec.log | cut -c1-900
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
run base MNX_DEC_CAP_FINE=0
run fine64 MNX_DEC_CAP_FINE=64
run base_b MNX_DEC_CAP_FINE=0
run fine64_b MNX_DEC_CAP_FINE=64
run fine128 MNX_DEC_CAP_FINE=128
