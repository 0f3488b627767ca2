#!/bin/bash
# scratch script for one-off GPU experiments (edit, run with gpurun, do not rely on its contents)
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sub"
run() { # name, args, env...
  n=$1; shift; x=$1; shift
  env "$@" timeout 400 $B $x > gpurun_out/b_$n.log 2>&1
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/b_$n.log").read().strip().splitlines()[-1])
    print("$n", d["value"], "mol/s", d["ms_per_step"], "ms/step")
except Exception as e:
    print("$n FAILED", e)
PY
}
run eb320 "--encode-batch 320" A=1
run eb448 "--encode-batch 448" A=1
run eb640 "--encode-batch 640" A=1
run eb320_b "--encode-batch 320" A=1
run eb224 "--encode-batch 224" A=1
timeout 600 python bench.py --no-cpu-baseline --no-sub --encode-batch 320 > gpurun_out/b_def320.log 2>&1; tail -1 gpurun_out/b_def320.log | cut -c1-140
timeout 600 python bench.py --no-cpu-baseline --no-sub --encode-batch 224 > gpurun_out/b_def224.log 2>&1; tail -1 gpurun_out/b_def224.log | cut -c1-140
timeout 600 python bench.py --no-cpu-baseline --no-sub --encode-batch 448 > gpurun_out/b_def448.log 2>&1; tail -1 gpurun_out/b_def448.log | cut -c1-140
