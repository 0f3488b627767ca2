#!/bin/bash
# robustness passes of the final tree: every tick enqueued kernel by kernel (MNX_NO_GRAPH), and the whole GPU suite with every greedy
# tick on the fused / mid arithmetic (MNX_DEC_MID_MAX=4096: the bit-reproducible switch of INTEGRATION.md section 5)
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/gpu/nograph_check.sh > gpurun_out/r06_c13_nograph.txt 2>&1; tail -6 gpurun_out/r06_c13_nograph.txt
MNX_DEC_MID_MAX=4096 timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06_gpu_suite_mid4096.txt 2>&1; echo "mid suite rc=$?"; tail -3 gpurun_out/r06_gpu_suite_mid4096.txt
