#!/usr/bin/env python3
"""Per-tick view of a rocprofv3 rocpd database of a bench run: a decode tick starts with dec_begin_kernel; for every
tick, its wall time (start of this begin -> start of the next one on the same stream), the sum of its kernels'
durations, its launch count and the row capacity it was launched for (grid of the first skinny linear). Grouped by
capacity: how long a tick takes when few rows are alive (the drain tail) vs many.   usage: tick_profile.py db [out]"""
import sqlite3
import statistics
import sys
from collections import defaultdict


def main(path, out=None):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    q = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else "0")
    rows = c.execute(f"select name, start, end, grid_x, grid_y, workgroup_x, workgroup_y, {q} from kernels order by start").fetchall()
    ticks = []
    cur = None
    for name, st, en, gx, gy, wx, wy, sid in rows:
        if "dec_begin_kernel" in name or "beam_begin_kernel" in name:
            if cur:
                ticks.append(cur)
            cur = {"start": st, "sid": sid, "kern": 0.0, "n": 0, "cap": None, "last_end": en, "seq": []}
        if cur is None or sid != cur["sid"]:
            continue
        if not ("dec_" in name or "beam_" in name):
            continue
        cur["kern"] += en - st
        cur["n"] += 1
        cur["seq"].append((name.split("(")[0][-40:], (en - st) / 1e3, gx // max(wx, 1) * (gy // max(wy, 1))))
        cur["last_end"] = max(cur["last_end"], en)
        if cur["cap"] is None and "dec_linear" in name:
            cur["cap"] = (gy // max(wy, 1)) * 32
        if "dec_head" in name:                       # one workgroup per row of capacity (fused ticks have no dec_linear)
            cur["cap"] = gx // max(wx, 1)
    if cur:
        ticks.append(cur)
    by = defaultdict(list)
    for a, b in zip(ticks, ticks[1:]):
        wall = b["start"] - a["start"]
        if wall < 5e6:                      # back-to-back ticks only (skip host gaps)
            by[a["cap"]].append((wall / 1e3, a["kern"] / 1e3, a["n"], (a["last_end"] - a["start"]) / 1e3))
    lines = [f"{'rows_cap':>8s} {'ticks':>6s} {'wall_us_med':>11s} {'wall_us_p10':>11s} {'busy_us_med':>11s} {'span_us_med':>11s} {'launches':>8s}"]
    for cap in sorted(k for k in by if k is not None):
        v = by[cap]
        w = sorted(x[0] for x in v)
        lines.append(f"{cap:8d} {len(v):6d} {statistics.median(w):11.1f} {w[len(w) // 10]:11.1f} "
                     f"{statistics.median(x[1] for x in v):11.1f} {statistics.median(x[3] for x in v):11.1f} {v[0][2]:8d}")
    # per-launch view of a tick: median duration of the i-th kernel of the tick, for the smallest and largest capacity
    caps_shown = sorted({c for c in by if c and c <= 256} | {min(k for k in by if k), max(k for k in by if k)})
    for cap in caps_shown:
        nl = statistics.mode(len(t["seq"]) for t in ticks if t["cap"] == cap)
        sel = [t for t in ticks if t["cap"] == cap and len(t["seq"]) == nl]
        if not sel:
            continue
        lines.append("")
        lines.append(f"tick at rows_cap {cap}: i-th launch, median us over {len(sel)} ticks, workgroups")
        for i in range(nl):
            d = sorted(t["seq"][i][1] for t in sel)
            lines.append(f"  {i:2d} {sel[0]['seq'][i][0]:40s} {d[len(d) // 2]:7.2f} {sel[0]['seq'][i][2]:6d}")
    # timeline: 10 ms bins — decode ticks (count, median row capacity, busy) and encoder busy time per bin
    t0 = rows[0][1]
    nb = int((rows[-1][2] - t0) / 1e7) + 1
    enc = [0.0] * nb
    dec = [0.0] * nb
    for name, st, en, gx, gy, wx, wy, sid in rows:
        b = int((st - t0) / 1e7)
        if "dec_" in name or "beam_" in name or "head_" in name:
            dec[b] += (en - st) / 1e3
        else:
            enc[b] += (en - st) / 1e3
    tk = defaultdict(list)
    for t in ticks:
        tk[int((t["start"] - t0) / 1e7)].append(t["cap"] or 0)
    lines.append("")
    lines.append(f"{'t_ms':>6s} {'ticks':>6s} {'cap_med':>8s} {'cap_max':>8s} {'dec_busy_us':>12s} {'other_busy_us':>14s}")
    for b in range(nb):
        caps = tk.get(b, [])
        lines.append(f"{b * 10:6d} {len(caps):6d} {int(statistics.median(caps)) if caps else 0:8d} {max(caps) if caps else 0:8d} "
                     f"{dec[b]:12.0f} {enc[b]:14.0f}")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
