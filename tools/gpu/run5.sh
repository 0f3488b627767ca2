#!/bin/bash
cd /root/repo
timeout 120 tools/gemm_lab/lab 112 5 0 32 "qkv s2,fc1 s2" 2>&1 | cut -c1-700
timeout 120 tools/gemm_lab/lab 112 20 0 0 "qkv s2,fc1 s2,fcb s2,qkv s3,fc1 s3,qkv s1,fc1 s1" 2>&1 | cut -c1-150 | grep -E "fc|qkv"
timeout 120 tools/gemm_lab/lab 64 20 0 0 "qkv s2,fc1 s2,fcb s2,qkv s3,fc1 s3,qkv s1,fc1 s1" 2>&1 | cut -c1-150 | grep -E "fc|qkv"
