#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output (counter_collection CSVs or the rocpd SQLite file of ROCm 7.2): mean counter value per kernel name substring.
usage: pmc_summary.py <dir> <kernel-substring>"""
import csv
import glob
import sys
from collections import defaultdict

root, key = sys.argv[1], sys.argv[2]
acc = defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if key in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
import sqlite3
for f in glob.glob(root + "/**/*_results.db", recursive=True):
    db = sqlite3.connect(f)
    for name, cname, value in db.execute("select kernel_name, counter_name, value from counters_collection"):
        if key in name:
            acc[cname].append(float(value))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:32s} n={len(v):3d} mean={sum(v) / len(v):.4g}")
