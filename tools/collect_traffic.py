#!/usr/bin/env python3
"""HBM traffic of the MFMA GEMM kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE — separate passes, as
MI355X_MICROARCH.md prescribes). gfx950 correction: FETCH_SIZE counts 128-B requests as 64 B for wide coalesced
streams (calibrated in this repo on a 512 MiB copy: FETCH_SIZE read exactly half, WRITE_SIZE exact), so reads are
doubled. Units are KiB.   usage: collect_traffic.py fetch.db write.db out.json [batch]"""
import json
import sqlite3
import sys


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, sum(value), count(*) from counters_collection where counter_name=? "
                     "group by kernel_name", (counter,)).fetchall()
    return {r[0]: (r[1], r[2]) for r in rows}


f = per_kernel(sys.argv[1], "FETCH_SIZE")
w = per_kernel(sys.argv[2], "WRITE_SIZE")
def is_gemm(k):
    return "gemm_tn" in k or "gemm256" in k or "gemm_res" in k


tot_f = sum(v[0] for k, v in f.items() if is_gemm(k))
n_f = sum(v[1] for k, v in f.items() if is_gemm(k))
tot_w = sum(v[0] for k, v in w.items() if is_gemm(k))
n_w = sum(v[1] for k, v in w.items() if is_gemm(k))
# launches = GEMM kernel launches of the pass; a layer whose rows are split between gemm256x3 and the 128-tile kernel
# counts twice, so per-LAYER figures use the layer count given on the command line when present
layers = int(sys.argv[5]) if len(sys.argv) > 5 else None
if layers:
    n_f = n_w = layers
out = {"kernel": f"mnx::gemm_tn_* / gemm256* / gemm_res kernels (all encoder GEMM launches of Swin-B @ B={sys.argv[4] if len(sys.argv) > 4 else 32})", "launches_per_pass": n_f,
       "read_bytes_per_launch": 2 * tot_f * 1024 / n_f, "write_bytes_per_launch": tot_w * 1024 / n_w,
       "hbm_bytes_per_launch": (2 * tot_f / n_f + tot_w / n_w) * 1024,
       "note": "FETCH_SIZE doubled (gfx950 undercount, calibrated); Infinity-Cache hits are counted as traffic"}
# the GEMM kernel sources the counters were collected on (bench.py::library_sha16 computes the same digest over the same files
# and refuses the file when its own differs)
import hashlib
import os
_root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
_h = hashlib.sha256()
for _name in ("gemm.hip", "gemm256.hip", "gemm_res.hip", "common.h", "kernels.h"):
    _h.update(_name.encode())
    with open(os.path.join(_root, "molnextr_amd", "csrc", _name), "rb") as _f:
        _h.update(_f.read())
out["library_sha16"] = _h.hexdigest()[:16]
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
