#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB per dispatch) of the same command, next
to the kernel durations of a --kernel-trace pass: every kernel of the run, not a fixed list (tools/pmc_hbm.py covers the two
2-D N-scale kernels only).

usage: pmc_all.py <fetch_dir> <write_dir> <kernel_stats.csv or ''> <out.json> [min_share_percent]
FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM / rocprofv3 section: gfx950 tallies wide coalesced reads at half their bytes);
hbm_bytes_per_launch = 2 * FETCH + WRITE.  `achieved_GBs` = that traffic / the kernel's average duration."""
import csv
import glob
import json
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("efgp::", "").replace("pcg::", "").strip()


def collect(root, counter):
    acc = defaultdict(list)
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] == counter:
                    acc[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    for f in glob.glob(root + "/**/*_results.db", recursive=True):
        db = sqlite3.connect(f)
        for name, value in db.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
            acc[short(name)].append(float(value))
    return acc


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
dur = {}
if sys.argv[3]:
    with open(sys.argv[3]) as fh:
        for row in csv.DictReader(fh):
            dur[short(row["Name"])] = (float(row["AverageNs"]), int(row["Calls"]), float(row["Percentage"]))
min_share = float(sys.argv[5]) if len(sys.argv) > 5 else 0.5
out = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (no trace domains with --pmc); FETCH_SIZE doubled "
               "per MI355X_MICROARCH.md; hbm_bytes_per_launch = 2*FETCH+WRITE; durations from a --kernel-trace pass of the same "
               "command; kernels below %.1f %% of the run's GPU time omitted" % min_share, "kernels": {}}
for k in sorted(set(fetch) & set(write), key=lambda k: -(dur.get(k, (0, 0, 0))[2])):
    f = sum(fetch[k]) / len(fetch[k])
    w = sum(write[k]) / len(write[k])
    d = dur.get(k)
    if d is not None and d[2] < min_share:
        continue
    ent = {"launches_counted": len(fetch[k]), "FETCH_SIZE_KiB_avg": round(f, 2), "WRITE_SIZE_KiB_avg": round(w, 2),
           "hbm_bytes_per_launch": round((2.0 * f + w) * 1024.0)}
    if d is not None:
        ent.update({"avg_us": round(d[0] / 1e3, 2), "calls": d[1], "share_percent": d[2],
                    "achieved_GBs": round(ent["hbm_bytes_per_launch"] / d[0], 1), "frac_of_8TBs": round(ent["hbm_bytes_per_launch"] / d[0] / 8000.0, 4)})
    out["kernels"][k] = ent
json.dump(out, open(sys.argv[4], "w"), indent=1)
for k, e in out["kernels"].items():
    print(f"{k[:70]:70s} {e.get('avg_us', 0):9.1f} us  {e['hbm_bytes_per_launch'] / 1e6:9.2f} MB/launch  {e.get('achieved_GBs', 0):8.1f} GB/s")
