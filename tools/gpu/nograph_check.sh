#!/bin/bash
# the eager (MNX_NO_GRAPH=1) tick path: smoke() and three decoder / pipeline tests without hipGraph replay
cd /root/repo
export TMPDIR=/tmp
MNX_NO_GRAPH=1 timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
MNX_NO_GRAPH=1 timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "greedy_decode_vs_reference or chunk_ids or predict_pipeline_equals" 2>&1 | tail -2
