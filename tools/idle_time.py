#!/usr/bin/env python3
"""GPU idle time (no kernel running) inside the steady state of a rocprofv3 kernel trace (rocpd sqlite): the last `frac` of the
trace's span.  usage: python tools/idle_time.py run_results.db [frac=0.4]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
rows = con.execute("select start, end from kernels order by start").fetchall()
t0, t1 = rows[0][0], max(r[1] for r in rows)
w0 = t1 - frac * (t1 - t0)
ev = [(max(s, w0), e) for s, e in rows if e > w0]
busy, cs, ce = 0, ev[0][0], ev[0][1]
gaps = []
for s, e in ev[1:]:
    if s > ce:
        busy += ce - cs
        gaps.append(s - ce)
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
span = t1 - w0
tot = sum(e - s for s, e in ev)
gaps.sort(reverse=True)
print(f"window {span / 1e6:.1f} ms: busy {busy / 1e6:.1f} ms ({100 * busy / span:.1f} %), idle {100 * (1 - busy / span):.1f} %, mean concurrency {tot / busy:.2f}; "
      f"{len(gaps)} gaps, largest [us]: {[round(g / 1e3, 1) for g in gaps[:8]]}, gaps > 20 us: {sum(1 for g in gaps if g > 20e3)} totalling {sum(g for g in gaps if g > 20e3) / 1e6:.2f} ms")
