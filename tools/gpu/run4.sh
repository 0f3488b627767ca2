export TMPDIR=/tmp
for l in g:2:294912:128:128 encode; do LOAD=$l ITERS=10 timeout 600 python tools/gpu/diag_load.py 2>&1 | grep -E "^iter|LOAD=|Error|error" | tail -2; done
ITERS=40 timeout 600 python tools/gpu/diag_pipe.py 2>&1 | grep -E "iterations with|repeat equal: False|Error|error"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "greedy or beam or chunk or decode or predict or pipeline or e2e or confidence or facade or public or checkpoint or neighbouring or grouped" > gpurun_out/t_parity.log 2>&1; echo "parity rc=$?"; tail -4 gpurun_out/t_parity.log
timeout 600 python -m pytest tests/test_gpu_pixels.py -q -m gpu -k fp32 > gpurun_out/t_pixels.log 2>&1; echo "pixels(fp32) rc=$?"; tail -2 gpurun_out/t_pixels.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench20.log 2>&1; tail -c 1700 gpurun_out/bench20.log | head -c 330; echo
for m in 0 3; do
cd /tmp && MNX_DEC_OLD=$m timeout 400 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_r2c$m -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r2c$m.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/prof_r2c$m -name "*.db" | head -1)
echo "== tick profile MNX_DEC_OLD=$m"; python tools/tick_profile.py $DB gpurun_out/tick_profile_r2c$m.txt
python tools/rocpd_stats.py $DB gpurun_out/kernel_stats_r2c$m.txt | grep -E "dec_|kernel " | head -12
rm -f $DB
done
