#!/bin/bash
cd /root/repo
timeout 120 tools/gemm_lab/lab 64 20 "fc1 s0,fc1 s1,qkv s1" 2>&1 | cut -c1-140
