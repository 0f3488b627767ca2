#!/bin/bash
# round-end pass: the whole GPU suite, smoke(), the rocprofv3 profiles of tools/gpu/profiles.sh (kernel stats + tick profile of
# the driver's command, the two PMC passes whose traffic figure the bench lines AFTER them report — same kernels, same box),
# the driver's bench command (full line, with sub-results and the CPU baseline), the default bench, the GEMM shape tables of
# tools/gemm_lab and the tick-time table of tools/tick_time.py, then the fused tick's phase stamps (lab build of the library)
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/pixels_parity.json
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/t_gpu.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/t_gpu.log | cut -c1-300
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
bash tools/gpu/profiles.sh
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_full20.log 2>&1; echo "bench20 rc=$?"; tail -1 gpurun_out/bench_full20.log | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline --no-sub > gpurun_out/bench_default.log 2>&1; echo "bench default rc=$?"; tail -1 gpurun_out/bench_default.log | cut -c1-200
timeout 600 python bench.py --no-cpu-baseline --no-sub --dtype fp16x3m > gpurun_out/bench_default_fp16x3m.log 2>&1; echo "bench default fp16x3m rc=$?"; tail -1 gpurun_out/bench_default_fp16x3m.log | cut -c1-200
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sub --dtype fp16x3m > gpurun_out/bench20_fp16x3m.log 2>&1; echo "bench20 fp16x3m rc=$?"; tail -1 gpurun_out/bench20_fp16x3m.log | cut -c1-200
timeout 300 python bench.py --gpus 1 --force-gather --steps 2 --warmup 1 --no-cpu-baseline --no-sub > gpurun_out/bench_force_gather.log 2>&1; echo "bench force-gather rc=$?"; tail -1 gpurun_out/bench_force_gather.log | cut -c1-200
timeout 600 python bench.py --gpus 1 --beam 5 --steps 32 --warmup 8 --no-cpu-baseline --no-sub > gpurun_out/bench_beam5.log 2>&1; echo "bench beam5 rc=$?"; tail -1 gpurun_out/bench_beam5.log | cut -c1-200
for m in fp16x3 fp16x2; do
  timeout 600 tools/gemm_lab/lab 512 20 - $m > gpurun_out/r06_gemm_shapes_${m}_b512.txt 2>&1; echo "lab $m rc=$?"
done
timeout 600 python tools/extended_parity.py --ckpt 0 --batches 4 --first 3000 --out gpurun_out/r06_extended_parity_0_fp16x3_final.json 2>&1 | grep EXTENDED_PARITY | cut -c1-500
timeout 600 python tools/tick_time.py 64,128,192,256,384,512,640 unfused,fused 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_tick_time.txt; echo "tick_time rc=$?"
if [ -f tools/ab/libmolnextr_hip_stamps.so ]; then
  cp molnextr_amd/lib/libmolnextr_hip.so /tmp/mnx_cur.so
  cp tools/ab/libmolnextr_hip_stamps.so molnextr_amd/lib/libmolnextr_hip.so
  for cfg in "64 250 2 4" "128 250 4 4"; do
    set -- $cfg
    MNX_FUSED_STAMPS=/tmp/st_$1_$3.bin timeout 300 python tools/fused_stamps.py run $1 $2 $3 $4 2>&1 | grep -v amdgpu.ids
    python tools/fused_stamps.py show /tmp/st_$1_$3.bin > gpurun_out/r06_fused_stamps_rows$1_tile$3.txt
  done
  cp /tmp/mnx_cur.so molnextr_amd/lib/libmolnextr_hip.so
fi
