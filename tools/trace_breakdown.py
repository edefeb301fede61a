#!/usr/bin/env python3
"""Per-class / per-queue kernel time of the steady-state tail of a rocprofv3 --kernel-trace database.
usage: python tools/trace_breakdown.py <results.db> [tail_seconds] [ms_per_step]"""
import collections, sqlite3, sys

db = sqlite3.connect(sys.argv[1])
tail = float(sys.argv[2]) if len(sys.argv) > 2 else 0.45
step_ms = float(sys.argv[3]) if len(sys.argv) > 3 else None
c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(c.execute(f"select k.start, k.end, k.queue_id, k.grid_size_x, k.grid_size_y, k.workgroup_size_x, s.kernel_name "
                      f"from {kd} k join {ks} s on k.kernel_id = s.id order by k.start"))
end = rows[-1][1]
lo = end - int(tail * 1e9)


def cls(n, gx, gy):
    if "gemm256" in n:
        return "gemm256"
    if "gemm_kernel" in n:
        if gy >= 100 and gy <= 130:
            return "enc.gemm"
        if "Li64E" in n:
            return "mem/pose.gemm64"
        if gx <= 32 and 40 <= gy <= 60 and "Li128" in n:
            return "dec.gemm"
        if gy in (8, 16, 48) and gx >= 12:
            return "mem.gemm128"
        return "head.gemm"
    for key, name in (("attn_kernelILi64ELi4", "enc.attn"), ("attn_kernelILi64ELi2", "dec.attn"), ("attn_kernelILi48", "dec.attn"),
                      ("attn_kernelILi128", "mem.attn"), ("layernorm_fast_kernelILi4", "enc.ln"), ("layernorm_fast_kernelILi3", "dec.ln"),
                      ("layernorm_fast_kernelILi6", "mem.ln"), ("rope", "rope"), ("win_", "chain"), ("logdepth", "chain"),
                      ("upsample", "head.misc"), ("dpt_final", "head.misc"), ("colmean", "head.misc"), ("postprocess", "head.misc")):
        if key in n:
            return name
    return "other:" + n.split("(")[0][-36:]


agg = collections.defaultdict(lambda: [0, 0.0])
iv = []
for a, b, q, gx, gy, wx, n in rows:
    if a < lo:
        continue
    k = cls(n, gx // max(wx, 1), gy)
    agg[(k, q)][0] += 1
    agg[(k, q)][1] += (b - a) / 1e6
    iv.append((a, b))
iv.sort()
busy, (cs, ce) = 0, iv[0]
for a, b in iv[1:]:
    if a > ce:
        busy += ce - cs
        cs, ce = a, b
    else:
        ce = max(ce, b)
busy += ce - cs
span = (iv[-1][1] - iv[0][0]) / 1e6
steps = span / step_ms if step_ms else 1.0
print(f"span {span:.1f} ms, GPU busy (union) {busy / 1e6:.1f} ms = {100 * busy / 1e6 / span:.1f} %" + (f", {steps:.2f} steps" if step_ms else ""))
for (k, q), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:32]:
    print(f"{k:40s} q{q}  launches/step {n / steps:8.1f}  ms/step {ms / steps:8.2f}")
