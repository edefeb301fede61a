#!/usr/bin/env python3
"""One tracking window of the one-window schedule, from a rocprofv3 --kernel-trace CSV: union busy time, concurrency histogram, per-queue
busy time, kernel classes by total time and by count, and the gaps of the busiest queue.  usage: wb1_timeline.py <kernel_trace.csv> [ms=24]"""
import csv
import collections
import re
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
rows.sort()
span_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 24.0
t1 = max(r[1] for r in rows)
t0 = t1 - int(span_ms * 1e6) - int(30e6)          # one window somewhere in the steady state: 30 ms before the end
w = [r for r in rows if r[0] >= t0 and r[1] <= t0 + int(span_ms * 1e6)]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"([A-Za-z0-9_]+(<[^>]*>)?)", n)
    return (m.group(1) if m else n)[:60]


ev = []
for s, e, q, n in w:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
hist = collections.Counter()
cur, last = 0, ev[0][0]
for t, d in ev:
    hist[cur] += t - last
    cur += d
    last = t
tot = sum(hist.values())
print(f"window of {span_ms} ms: {len(w)} kernels; concurrency histogram [ms]: " + ", ".join(f"{k}: {v / 1e6:.2f}" for k, v in sorted(hist.items())))
byq = collections.defaultdict(float)
cnt = collections.Counter()
tim = collections.Counter()
for s, e, q, n in w:
    byq[q] += (e - s) / 1e6
    cnt[short(n)] += 1
    tim[short(n)] += (e - s) / 1e3
print("busy per queue [ms]:", {q: round(v, 2) for q, v in sorted(byq.items(), key=lambda kv: -kv[1])})
print("kernel classes by total time [us] (count, avg us):")
for k, v in tim.most_common(28):
    print(f"  {v:9.1f}  {cnt[k]:5d}  {v / cnt[k]:7.2f}  {k}")
