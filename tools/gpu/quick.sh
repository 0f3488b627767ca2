#!/bin/bash
# scratch script for one-off GPU experiments (edit, run with gpurun, do not rely on its contents)
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pixels.py -q -m gpu -x -k "fused or greedy or decode or predict or chunk or beam1 or end_to_end or pipeline or stress or falls_back or range_flag or gemm_split or persistent_256" > gpurun_out/t_dec.log 2>&1; echo "pytest subset rc=$?"; tail -15 gpurun_out/t_dec.log | cut -c1-600
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sub"
run() { # name, env...
  n=$1; shift
  env "$@" timeout 400 $B > gpurun_out/b_$n.log 2>&1
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/b_$n.log").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("$n", d["value"], "mol/s", d["ms_per_step"], "ms/step; gemm frac", r["frac"], "s34 exec", r["stage34"]["frac_of_peak_executed"], "iso", r["isolated"]["achieved"])
except Exception as e:
    print("$n FAILED", e)
PY
}
run base MNX_X3_STAGGER_US=0
run stag4 MNX_X3_STAGGER_US=4
run stag8 MNX_X3_STAGGER_US=8
run base_b MNX_X3_STAGGER_US=0
run stag12 MNX_X3_STAGGER_US=12
run stag20 MNX_X3_STAGGER_US=20
(cd /tmp && env MNX_DEC_FUSED_MAX=4096 timeout 400 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_tick_all -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sub > $GRAFT_REPO_ROOT/gpurun_out/prof_tick_all.log 2>&1)
DB=$(find gpurun_out/prof_tick_all -name "*.db" | head -1)
python tools/tick_profile.py $DB gpurun_out/tick_profile_all.txt | head -12
rm -f $DB
cp molnextr_amd/lib/libmolnextr_hip.so /tmp/mnx_cur.so
cp tools/ab/libmolnextr_hip_stamps.so molnextr_amd/lib/libmolnextr_hip.so
for cfg in "64 250 2 4" "128 250 4 4"; do
  set -- $cfg
  MNX_FUSED_STAMPS=/tmp/st_$1_$3.bin timeout 300 python tools/fused_stamps.py run $1 $2 $3 $4 2>&1 | grep -v amdgpu.ids
  python tools/fused_stamps.py show /tmp/st_$1_$3.bin > gpurun_out/stamps_$1_r$3.txt
done
cp /tmp/mnx_cur.so molnextr_amd/lib/libmolnextr_hip.so
