#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB per dispatch) into per-kernel HBM bytes per launch.

usage: pmc_hbm.py <fetch_dir> <write_dir> <out.json>
FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 tallies wide coalesced reads at half their bytes);
hbm_bytes_per_launch = 2*FETCH + WRITE.  The spread entry is copied to the top level for bench.py."""
import csv
import glob
import json
import sys
from collections import defaultdict

KERNELS = {"spread_kernel": ("spread_mfma_kernel", "spread_pad_kernel", "spread_kernel", "spread_tile_kernel"),
           "interp_kernel": ("interp_real2_pair_kernel", "interp_real_halo_kernel", "interp_kernel"),
           "reduce_slabs_kernel": ("reduce_slabs_kernel",),
           "cg_persistent_2d64_kernel": ("cg_persistent_2d64_kernel",)}


def _match(name):
    for key, pats in KERNELS.items():
        if any((p + "<") in name or (p + "(") in name or name.endswith(p) for p in pats):
            return key
    return None


def collect(root, counter):
    """counter values per kernel group from the rocprofv3 output under `root`: counter_collection CSVs, or the rocpd
    SQLite file (`counters_collection` view) that ROCm 7.2 writes by default"""
    import sqlite3
    acc = defaultdict(list)
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] == counter and _match(row["Kernel_Name"]):
                    acc[_match(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    for f in glob.glob(root + "/**/*_results.db", recursive=True):
        db = sqlite3.connect(f)
        for name, value in db.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
            key = _match(name)
            if key:
                acc[key].append(float(value))
    return acc


fetch = collect(sys.argv[1], "FETCH_SIZE")
write = collect(sys.argv[2], "WRITE_SIZE")
out = {}
for key in KERNELS:
    if not fetch.get(key) or not write.get(key):
        continue
    f = sum(fetch[key]) / len(fetch[key])
    w = sum(write[key]) / len(write[key])
    out[key] = {"launches": len(fetch[key]), "FETCH_SIZE_KiB_avg": f, "WRITE_SIZE_KiB_avg": w,
                "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0}
out["note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of `bench.py --main-only --steps 10 --warmup 2` (global N as given on the command line, d=2: every launch belongs to the fit step); "
               "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts wide coalesced reads at half). "
               "hbm_bytes_per_launch = 2*FETCH+WRITE.")
if "spread_kernel" in out:
    out["hbm_bytes_per_launch"] = out["spread_kernel"]["hbm_bytes_per_launch"]
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
