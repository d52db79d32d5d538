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

KERNELS = {"spread_kernel": ("spread_pad_kernel", "spread_kernel", "spread_tile_kernel"),
           "interp_kernel": ("interp_real_halo_kernel", "interp_kernel"),
           "reduce_slabs_kernel": ("reduce_slabs_kernel",),
           "cg_persistent_2d64_kernel": ("cg_persistent_2d64_kernel",)}


def collect(root, counter):
    acc = defaultdict(list)
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                for key, pats in KERNELS.items():
                    if any(("::" + p + "<") in row["Kernel_Name"] or ("::" + p + "(") in row["Kernel_Name"] for p in pats):
                        acc[key].append(float(row["Counter_Value"]))
                        break
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
out["note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of `bench.py --main-only --steps 10 --warmup 2` (N=1e6, d=2: every launch belongs to the fit step); "
               "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts wide coalesced reads at half). "
               "hbm_bytes_per_launch = 2*FETCH+WRITE.")
if "spread_kernel" in out:
    out["hbm_bytes_per_launch"] = out["spread_kernel"]["hbm_bytes_per_launch"]
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
