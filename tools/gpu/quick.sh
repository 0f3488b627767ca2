#!/bin/bash
# scratch script for one-off GPU experiments (edit, run with gpurun, do not rely on its contents)
cd /root/repo
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 python tools/gpu/diag_split_gemm.py > gpurun_out/diag_split.txt 2>&1; echo "diag rc=$?"; grep "^epi" gpurun_out/diag_split.txt | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm or swin_tiny" > gpurun_out/t_gemm.log 2>&1; echo "pytest gemm rc=$?"; tail -5 gpurun_out/t_gemm.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_pixels.py -x -q -m gpu -k "fp16x3 or budget" > gpurun_out/t_pixels.log 2>&1; echo "pytest pixels rc=$?"; tail -3 gpurun_out/t_pixels.log | cut -c1-600
cd /tmp
SH="qkv s2,proj s2,fc2 s2"
for mode in bf16 fp16x3; do
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d $R/gpurun_out/pmc_sq_$mode -o lab -- $R/tools/gemm_lab/lab 128 2 "$SH" $mode > $R/gpurun_out/pmc_sq_$mode.log 2>&1; echo "pmc sq $mode rc=$?"
timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_tcc_$mode -o lab -- $R/tools/gemm_lab/lab 128 2 "$SH" $mode > $R/gpurun_out/pmc_tcc_$mode.log 2>&1; echo "pmc tcc $mode rc=$?"
done
cd $R
python tools/pmc_summary.py gpurun_out/pmc_gemm_summary.txt $(find gpurun_out/pmc_sq_bf16 gpurun_out/pmc_tcc_bf16 gpurun_out/pmc_sq_fp16x3 gpurun_out/pmc_tcc_fp16x3 -name "*.db") | cut -c1-200
find gpurun_out/pmc_* -name "*.db" -delete
tail -3 gpurun_out/pmc_sq_bf16.log | cut -c1-300
