#!/bin/bash
# round 5: SQ / GRBM / TCC counters of gemm256x3_kernel (product build) on the stage-3 shapes and 8192^3 — matrix-pipe busy cycles
# against the kernel's duration in cycles (separate --pmc passes, kernel-trace only)
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05_pmc; mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
cd /tmp
SH="qkv s2,proj s2,fc1 s2,fc2 s2,sq8192"
MNX_LAB_NOBASE=1 timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $R/$OUT/sq -o lab -- $R/tools/gemm_lab/lab 512 3 "$SH" fp16x3 > $R/$OUT/sq.log 2>&1; echo "sq rc=$?"
MNX_LAB_NOBASE=1 timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $R/$OUT/tcc -o lab -- $R/tools/gemm_lab/lab 512 3 "$SH" fp16x3 > $R/$OUT/tcc.log 2>&1; echo "tcc rc=$?"
cd $R
python tools/pmc_summary.py $OUT/r05_pmc_gemm_summary.txt $(find $OUT -name "*.db") | grep -A12 "gemm256x3" | head -80
# per-dispatch durations (ns) of the same run, to turn cycles into fractions
python - <<'PY'
import sqlite3, glob
for db in glob.glob('gpurun_out/r05_pmc/sq/**/*.db', recursive=True):
    c = sqlite3.connect(db)
    try:
        names = [r[0] for r in c.execute("select name from sqlite_master where type='table' or type='view'")]
        t = [n for n in names if 'kernel_dispatch' in n.lower() or n == 'kernels']
        print('tables', t[:6])
        for n in t[:1]:
            cols = [r[1] for r in c.execute(f"pragma table_info({n})")]
            print(n, cols)
    except Exception as e:
        print(e)
PY
find $OUT -name "*.db" -size +8M -delete
echo done
