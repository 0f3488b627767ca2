for d in 0 16; do echo "== uniform wave id, dbg=$d"; MNX_EXP_DBG=$d LOAD=g:2:294912:128:128 ITERS=10 timeout 600 python tools/gpu/diag_load.py 2>&1 | grep -E "^iter|LOAD=|Error|error" | tail -2; done
