#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp && timeout 400 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_tick -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sub > $GRAFT_REPO_ROOT/gpurun_out/prof_tick.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/prof_tick -name "*.db" | head -1)
python tools/tick_profile.py $DB gpurun_out/tick_profile.txt
python tools/rocpd_stats.py $DB gpurun_out/kernel_stats_tick.txt | head -3
rm -f $DB
