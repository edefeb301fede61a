#!/usr/bin/env python3
"""Per-kernel statistics + timeline utilisation from a rocprofv3 (ROCm 7.2, rocpd sqlite) kernel trace.
usage: python tools/rocpd_stats.py run_results.db [--csv out.csv] [--window t0_ms t1_ms]"""
import argparse
import re
import sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--csv", default=None)
ap.add_argument("--top", type=int, default=40)
args = ap.parse_args()
con = sqlite3.connect(args.db)
rows = con.execute("select name, start, end, queue_id, stream_id, grid_x, grid_y, grid_z, workgroup_x, vgpr_count, lds_size from kernels order by start").fetchall()


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*$", "", n)
    n = re.sub(r"^void ", "", n)
    return n[:90]


agg = {}
for name, s, e, q, st, gx, gy, gz, wx, vg, lds in rows:
    k = short(name)
    a = agg.setdefault(k, [0, 0, 10 ** 18, 0])
    a[0] += 1
    a[1] += e - s
    a[2] = min(a[2], e - s)
    a[3] = max(a[3], e - s)
tot = sum(a[1] for a in agg.values())
t0, t1 = rows[0][1], max(r[2] for r in rows)
# union of busy intervals
ev = sorted((r[1], r[2]) for r in rows)
busy, cs, ce = 0, ev[0][0], ev[0][1]
for s, e in ev[1:]:
    if s > ce:
        busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print(f"{len(rows)} dispatches, span {(t1 - t0) / 1e6:.1f} ms, sum of kernel durations {tot / 1e6:.1f} ms, union busy {busy / 1e6:.1f} ms, "
      f"mean concurrency {tot / max(busy, 1):.2f}")
print(f"{'kernel':90s} {'calls':>7s} {'total ms':>9s} {'avg us':>8s} {'min us':>8s} {'max us':>8s} {'%':>6s}")
lines = []
for k, (n, d, mn, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append((k, n, d / 1e6, d / n / 1e3, mn / 1e3, mx / 1e3, 100.0 * d / tot))
for l in lines[:args.top]:
    print(f"{l[0]:90s} {l[1]:7d} {l[2]:9.2f} {l[3]:8.1f} {l[4]:8.1f} {l[5]:8.1f} {l[6]:6.2f}")
if args.csv:
    with open(args.csv, "w") as f:
        f.write("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage\n")
        for l in lines:
            f.write(f"\"{l[0]}\",{l[1]},{int(l[2] * 1e6)},{l[3] * 1e3:.1f},{l[4] * 1e3:.1f},{l[5] * 1e3:.1f},{l[6]:.3f}\n")
