#!/bin/bash
# after the patch-embedding change (bit-identical output, tools/gpu/r05_patch_embed.sh): the driver's bench command (full line),
# the rocprofv3 kernel stats / tick profile of the same command, the default bench. The PMC traffic file stays: its digest covers
# the GEMM sources, which did not change.
cd /root/repo
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_full20.log 2>&1; echo "bench20 rc=$?"; tail -1 gpurun_out/bench_full20.log | cut -c1-300
rm -rf gpurun_out/prof_stats
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-sub --no-cpu-baseline > $R/gpurun_out/prof_stats.log 2>&1; echo "stats rc=$?")
DB=$(find gpurun_out/prof_stats -name "*.db" | head -1)
if [ -n "$DB" ]; then
  python tools/rocpd_stats.py $DB gpurun_out/r05_kernel_stats_bench20.txt | head -14
  python tools/tick_profile.py $DB gpurun_out/r05_tick_profile_bench20.txt | head -4
  for f in $(find gpurun_out/prof_stats -name "*kernel_stats*.csv"); do cp $f gpurun_out/r05_rocprofv3_kernel_stats_bench20.csv; done
  rm -f $DB
fi
timeout 150 python bench.py --no-cpu-baseline --no-sub > gpurun_out/bench_default.log 2>&1; echo "bench default rc=$?"; tail -1 gpurun_out/bench_default.log | cut -c1-200
