#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (--kernel-trace --stats) as a per-kernel table (text)."""
import re
import sqlite3
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*$", "", n)
    n = n.replace("void ", "").replace("mnx::", "")
    return n[:78]


def main(path, out=None):
    c = sqlite3.connect(path)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                     "max(grid_x*grid_y*grid_z/(workgroup_x*workgroup_y*workgroup_z)), max(vgpr_count), max(lds_size) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = [f"{'kernel':78s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>9s} {'%':>6s} {'wgs':>7s} {'vgpr':>5s} {'lds':>6s}"]
    for n, cnt, tot, avg, mn, mx, wgs, vg, lds in rows:
        lines.append(f"{short(n):78s} {cnt:7d} {tot / 1e6:10.3f} {avg / 1e3:9.2f} {mn / 1e3:8.2f} {mx / 1e3:9.2f} "
                     f"{100 * tot / total:6.2f} {int(wgs or 0):7d} {int(vg or 0):5d} {int(lds or 0):6d}")
    lines.append(f"TOTAL kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
