#!/bin/bash
# round 5, GPU call 2: fixed fp32 epilogue forms, polynomial GELU without stores, staggered cohorts, MFMA power by instruction shape
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05_lab2; mkdir -p $OUT
L=tools/gemm_lab
S2="qkv s2,proj s2,fc1 s2,fc2 s2"
run() { local name=$1 bin=$2 shapes=$3; shift 3
    echo "=== $name: $* $bin 448 10 [$shapes]" | tee -a $OUT/all.txt
    env "$@" timeout 120 $L/$bin 448 10 "$shapes" fp16x3 2>&1 | tee -a $OUT/all.txt; }
tools/probes/mfma_power 2>&1 | tee $OUT/mfma_power.txt
ALL32="proj s1,fc2 s1,merge s1,proj s2,fc2 s2,merge s2,proj s3,fc2 s3,merge s0"
for e in e0 e1 e2; do run epi_$e lab_$e "$ALL32" MNX_LAB_NOBASE=1; done
run gelu_noglb lab_gelu_noglb "fc1 s2,qkv s2" MNX_LAB_NOBASE=1
run gelu lab_gelu "fc1 s2,qkv s2" MNX_LAB_NOBASE=1
for v in p p_st3 p_st6 p_st12; do run $v lab_$v "$S2,fc1 s0,fc1 s1,proj s1,fc2 s1" MNX_LAB_NOBASE=1; done
run e0_again lab_e0 "$S2,fc1 s0,fc1 s1,proj s1,fc2 s1" MNX_LAB_NOBASE=1
echo done
