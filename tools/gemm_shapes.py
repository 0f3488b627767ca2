#!/usr/bin/env python3
"""Per-shape timing of the encoder kernels from a rocprofv3 rocpd database (B=32 Swin-B shapes)."""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
rows = c.execute("select name, grid_x/workgroup_x as wgs, count(*), avg(duration), min(duration) from kernels where "
                 "name like '%gemm_tn%' or name like '%window_attn%' or name like '%layernorm%' or name like '%patch_embed%' "
                 "group by name, wgs order by name, wgs").fetchall()


def tiles(M, N, bn):
    return ((M + 127) // 128) * ((N + bn - 1) // bn)


cand = []
for s, (L, C) in enumerate([(9216, 128), (2304, 256), (576, 512), (144, 1024)]):
    M = B * L
    for bn in (128, 64):
        cand += [(0, tiles(M, 3 * C, bn), M, 3 * C, C, f'qkv s{s}', bn), (2, tiles(M, C, bn), M, C, C, f'proj s{s}', bn),
                 (1, tiles(M, 4 * C, bn), M, 4 * C, C, f'fc1 s{s}', bn), (2, tiles(M, C, bn), M, C, 4 * C, f'fc2 s{s}', bn)]
        if s < 3:
            cand.append((3, tiles(M // 4, 2 * C, bn), M // 4, 2 * C, 4 * C, f'merge s{s}', bn))
for n, w, cnt, avg, mn in rows:
    label = ''
    if 'gemm' in n:
        m = re.search(r'Li(\d)ELi(\d+)E', n) or re.search(r'Li(\d)E', n)
        epi = int(m.group(1)) if m else None
        bn = int(m.group(2)) if m and m.lastindex and m.lastindex >= 2 else None
        if 'bool _Accum' in n:
            epi = None
        for e, t, M, N, K, l, cb in cand:
            if (epi is None or e == epi) and t == w and (bn is None or bn == cb):
                fl = 2 * M * N * K
                label += f" {l} bn{cb} M{M} N{N} K{K}: {fl / avg / 1e3:.0f} TF |"
    print(f"{n[:52]:52s} wgs {w:6d} n {cnt:5d} avg {avg / 1e3:8.1f} us min {mn / 1e3:8.1f} {label}")
