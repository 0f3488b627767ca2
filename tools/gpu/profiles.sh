#!/bin/bash
# profiles for the round: rocprofv3 kernel-trace stats of the driver's bench command, the two PMC passes (FETCH_SIZE,
# WRITE_SIZE: separate passes) over plain encodes of one launch group, and the decode-tick view of the same trace
cd /root/repo
R=$GRAFT_REPO_ROOT
EB=${EB:-512}
DT=${DT:-fp16x3}
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-sub --no-cpu-baseline > $R/gpurun_out/prof_stats.log 2>&1
echo "stats rc=$?"
BATCH=$EB ENCODES=2 DTYPE=$DT timeout 600 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o enc -- python $R/tools/encode_once.py > $R/gpurun_out/pmc_fetch.log 2>&1
echo "fetch rc=$?"
BATCH=$EB ENCODES=2 DTYPE=$DT timeout 600 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o enc -- python $R/tools/encode_once.py > $R/gpurun_out/pmc_write.log 2>&1
echo "write rc=$?"
cd $R
F=$(find gpurun_out/pmc_fetch -name "*.db" | head -1); W=$(find gpurun_out/pmc_write -name "*.db" | head -1)
# 99 GEMM layers per encode x 2 encodes
python tools/collect_traffic.py $F $W gpurun_out/r06_gemm_traffic_${DT}_b$EB.json $EB 198
cp gpurun_out/r06_gemm_traffic_${DT}_b$EB.json profiles/      # on this box: a bench run after this one reports it (same kernels)
DB=$(find gpurun_out/prof_stats -name "*.db" | head -1)
python tools/rocpd_stats.py $DB gpurun_out/r06_kernel_stats_bench20.txt | head -14
python tools/tick_profile.py $DB gpurun_out/r06_tick_profile_bench20.txt | head -12
for f in $(find gpurun_out/prof_stats -name "*kernel_stats*.csv"); do cp $f gpurun_out/r06_rocprofv3_kernel_stats_bench20.csv; done
rm -f $F $W $DB
tail -1 gpurun_out/prof_stats.log | cut -c1-200
