#!/bin/bash
# round 5, GPU call 1: what bounds gemm256x3_kernel? (clock vs fills vs issue), fp32 epilogue forms, polynomial GELU.
# Everything is tools/gemm_lab (built in the container); output -> gpurun_out/r05_lab/
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05_lab; mkdir -p $OUT
L=tools/gemm_lab
S2="qkv s2,proj s2,fc1 s2,fc2 s2"
run() { # name, binary, shapes, env...
    local name=$1 bin=$2 shapes=$3; shift 3
    echo "=== $name: $* $bin 448 10 [$shapes]" | tee -a $OUT/all.txt
    env "$@" timeout 120 $L/$bin 448 10 "$shapes" fp16x3 2>&1 | tee -a $OUT/all.txt
}
# 1. epilogue forms (correctness = 0 words differing) on every fp32-output shape the x3 kernel takes, + clocks
for e in e0 e1 e2; do run epi_$e lab_$e "proj s1,fc2 s1,merge s1,proj s2,fc2 s2,merge s2,proj s3,fc2 s3,merge s0" MNX_LAB_NOBASE=1; done
# 2. the 16-bit shapes + sq8192 with clocks (reference for the ablations), polynomial GELU
run base16 lab_e0 "qkv s2,fc1 s2,fc1 s0,fc1 s1,fc1 s3,sq8192" MNX_LAB_NOBASE=1
run gelu lab_gelu "fc1 s2,fc1 s0,fc1 s1,fc1 s3" MNX_LAB_NOBASE=1
# 3. fewer CUs, zero operands: clock vs contention
for c in 128 192 224; do run cus$c lab_e0 "$S2,sq8192" MNX_LAB_NOBASE=1 MNX_LAB_CUS=$c; done
run zero lab_e0 "$S2,sq8192" MNX_LAB_NOBASE=1 MNX_LAB_ZERO=1
run zero128 lab_e0 "fc2 s2,sq8192" MNX_LAB_NOBASE=1 MNX_LAB_ZERO=1 MNX_LAB_CUS=128
# 4. ablations
for a in nofill nomfma noepi noread noglb nofill_noepi mfmaonly; do run abl_$a lab_$a "$S2,sq8192" MNX_LAB_NOBASE=1; done
run abl_mfmaonly128 lab_mfmaonly "fc2 s2,sq8192" MNX_LAB_NOBASE=1 MNX_LAB_CUS=128
run abl_nofill128 lab_nofill "fc2 s2,sq8192" MNX_LAB_NOBASE=1 MNX_LAB_CUS=128
# 5. PMC cross-check of the effective clock: GRBM_GUI_ACTIVE / wall per launch
export TMPDIR=/tmp
for c in 256 128; do
    MNX_LAB_NOBASE=1 MNX_LAB_CUS=$c timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_cus$c -- $L/lab_e0 448 5 "fc2 s2,sq8192" fp16x3 > $OUT/pmc_cus$c.log 2>&1
done
python3 - <<'PY' > $OUT/pmc_clock.txt 2>&1
import csv, glob, collections
for c in (256, 128):
    cnt = collections.defaultdict(list); dur = collections.defaultdict(list)
    for f in glob.glob(f'gpurun_out/r05_lab/pmc_cus{c}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get('Counter_Name') == 'GRBM_GUI_ACTIVE': cnt[(r['Kernel_Name'][:60], r['Dispatch_Id'])].append(float(r['Counter_Value']))
    for f in glob.glob(f'gpurun_out/r05_lab/pmc_cus{c}/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            dur[(r['Kernel_Name'][:60], r['Dispatch_Id'])] = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
    agg = collections.defaultdict(list)
    for k, v in cnt.items():
        if k in dur and dur[k] > 0: agg[k[0]].append((max(v), dur[k]))
    for k, v in agg.items():
        if 'gemm256x3' not in k: continue
        g = sum(a for a, _ in v) / len(v); d = sum(b for _, b in v) / len(v)
        print(f'cus {c} {k}: launches {len(v)}  GRBM_GUI_ACTIVE {g:.0f}  wall {d/1000:.1f} us  -> {g/d*1000:.0f} MHz')
PY
cat $OUT/pmc_clock.txt
find $OUT -name "*.csv" -size +2M -delete
echo done
