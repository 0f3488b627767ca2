set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pixels.py -q -m gpu -s > gpurun_out/t_pixels.log 2>&1; echo "pixels rc=$?"
tail -4 gpurun_out/t_pixels.log
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "greedy or beam or chunk or decode or predict or pipeline or e2e or confidence or facade or public or checkpoint or crop or preprocess" > gpurun_out/t_parity.log 2>&1; echo "parity rc=$?"
tail -8 gpurun_out/t_parity.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench20.log 2>&1; tail -c 1700 gpurun_out/bench20.log | head -c 500
cd /tmp && timeout 400 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_r2b -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r2b.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/prof_r2b -name "*.db" | head -1)
python tools/tick_profile.py $DB gpurun_out/tick_profile_r2b.txt
python tools/rocpd_stats.py $DB gpurun_out/kernel_stats_r2b.txt | head -24
rm -f $DB
