#!/usr/bin/env python3
"""Reduce rocprofv3 CSV outputs on the GPU box to the small summaries that are committed under profiles/:
  pmc_reduce.py traffic <fetch_dir> <write_dir> <out.json>       per-kernel HBM-side bytes per launch (FETCH_SIZE x2 + WRITE_SIZE)
  pmc_reduce.py sq <sq_dir> <out.json>                           per-kernel SQ counter averages
gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE reports half of a wide
coalesced read (TCC_EA0_RDREQ tallied at 64 B) -> doubled.  Separate --pmc passes with --kernel-trace only."""
import collections
import csv
import glob
import json
import re
import subprocess
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*$", "", n)
    return re.sub(r"^void ", "", n)[:80]


def table(dirname):
    f = glob.glob(f"{dirname}/*counter_collection.csv") + glob.glob(f"{dirname}/*/*counter_collection.csv")
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(f[0])):
        a = acc[short(r["Kernel_Name"])][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    return acc


def sha():
    try:
        return subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        return None


if sys.argv[1] == "traffic":
    fd, wd, out = sys.argv[2:5]
    F, W = table(fd), table(wd)
    res = {"note": "bytes per launch = 2 * FETCH_SIZE[KiB] * 1024 + WRITE_SIZE[KiB] * 1024 (gfx950 correction); separate --pmc passes",
           "git_sha": sha(), "kernels": {}}
    for k in F:
        if "FETCH_SIZE" not in F[k] or k not in W or "WRITE_SIZE" not in W[k]:
            continue
        f, nf = F[k]["FETCH_SIZE"]
        w, nw = W[k]["WRITE_SIZE"]
        res["kernels"][k] = {"launches": nf, "fetch_kib_raw_avg": f / nf, "write_kib_raw_avg": w / max(nw, 1),
                             "traffic_bytes_per_launch": 2 * f / nf * 1024 + w / max(nw, 1) * 1024}
    json.dump(res, open(out, "w"), indent=1)
else:
    sd, out = sys.argv[2:4]
    S = table(sd)
    res = {"note": "average per launch of each SQ counter (SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles, "
                   "SQ_VALU_MFMA_BUSY_CYCLES in cycles; MI355X_MICROARCH.md)", "git_sha": sha(), "kernels": {}}
    for k, c in S.items():
        row = {name: v[0] / v[1] for name, v in c.items()}
        row["launches"] = max(v[1] for v in c.values())
        wc = row.get("SQ_WAVE_CYCLES")
        if wc:
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if n in row:
                    row[n + "_frac_of_wave_cycles"] = row[n] / wc
        res["kernels"][k] = row
    json.dump(res, open(out, "w"), indent=1)
print("wrote", sys.argv[-1])
