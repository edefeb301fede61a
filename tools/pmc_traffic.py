#!/usr/bin/env python3
"""Reduce two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes) to the
average HBM traffic per launch of the dominant kernel.  gfx950 corrections: FETCH_SIZE/WRITE_SIZE are in KiB;
FETCH_SIZE under-reports wide coalesced reads by exactly 2x (TCC_EA0_RDREQ tallied at 64 B), so it is doubled.

usage: pmc_traffic.py <fetch_dir> <write_dir> <kernel substring> <out.json>"""
import csv, glob, json, sys


def avg(dirname, counter, sub):
    f = glob.glob(f"{dirname}/*/*counter_collection.csv")[0]
    tot, n = 0.0, 0
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and sub in r["Kernel_Name"]:
            tot += float(r["Counter_Value"]); n += 1
    return tot / max(n, 1), n


if __name__ == "__main__":
    fd, wd, sub, out = sys.argv[1:5]
    fetch_kib, nf = avg(fd, "FETCH_SIZE", sub)
    write_kib, nw = avg(wd, "WRITE_SIZE", sub)
    res = {"kernel": sub, "launches_fetch_pass": nf, "launches_write_pass": nw,
           "fetch_size_kib_raw": fetch_kib, "write_size_kib_raw": write_kib,
           "fetch_bytes_corrected": 2 * fetch_kib * 1024, "write_bytes": write_kib * 1024,
           "traffic_bytes_per_launch": 2 * fetch_kib * 1024 + write_kib * 1024,
           "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (HBM section); separate --pmc passes with --kernel-trace only"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))
