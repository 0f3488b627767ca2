#!/bin/bash
# scratch: rocprofv3 trace of the default (512-step) bench -> tick profile at steady state
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_def -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-sub > $GRAFT_REPO_ROOT/gpurun_out/prof_def.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/prof_def -name "*.db" | head -1)
ls -la $DB
python tools/tick_profile.py $DB gpurun_out/tick_profile_default.txt > /dev/null
python tools/rocpd_stats.py $DB gpurun_out/kernel_stats_default.txt | head -24
rm -f $DB
head -16 gpurun_out/tick_profile_default.txt
grep -n "tick at rows_cap" -A51 gpurun_out/tick_profile_default.txt | tail -52 | head -20
tail -1 gpurun_out/prof_def.log | cut -c1-200
