#!/usr/bin/env python3
"""Per-kernel statistics (the table `rocprofv3 --kernel-trace --stats` prints) from the rocpd SQLite file rocprofv3
writes by default.  usage: kernel_stats_from_db.py <results.db> [out.csv]"""
import csv
import sqlite3
import statistics
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select s.display_name, s.kernel_name, d.end - d.start from rocpd_kernel_dispatch d "
                  "join rocpd_info_kernel_symbol s on s.id = d.kernel_id").fetchall()
per = {}
for disp, name, ns in rows:
    per.setdefault(disp or name, []).append(int(ns))
total = sum(sum(v) for v in per.values()) or 1
out = open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout
w = csv.writer(out, quoting=csv.QUOTE_NONNUMERIC)
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    w.writerow([name, len(v), sum(v), round(sum(v) / len(v), 3), round(100.0 * sum(v) / total, 2), min(v), max(v),
                round(statistics.pstdev(v), 3)])
