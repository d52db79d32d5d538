#!/usr/bin/env python3
"""The launch sequence of ONE step out of a rocprofv3 --kernel-trace database: gaps, durations, names, and the per-step totals.
The step boundary is a kernel that runs exactly once per step (given by a substring).
usage: launch_sequence.py <results.db> <substring of the once-per-step kernel> [steps to average over]"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
key = sys.argv[2]
take = int(sys.argv[3]) if len(sys.argv) > 3 else 50
rows = list(db.execute("select name, start, end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if key in r[0]]
take = min(take, len(idx) - 2)
i0, i1 = idx[-take - 1], idx[-1]
seg = rows[i0:i1]
busy = sum(e - s for _, s, e in seg)
print(f"{(i1 - i0) / take:.1f} launches per step, device busy {busy / take / 1e3:.1f} us per step (kernel durations under the profiler), "
      f"averaged over {take} steps")
agg = collections.defaultdict(lambda: [0, 0])
for n, s, e in seg:
    agg[n][0] += e - s
    agg[n][1] += 1
print("per step, by kernel:")
for n, (t, k) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"  {t / take / 1e3:8.1f} us  {k / take:5.1f} x  {re.sub('at::native::', '', n)[:120]}")
print("one step in order (duration us):")
for n, s, e in rows[idx[-2]:idx[-1]]:
    print(f"  {(e - s) / 1e3:7.1f}  {re.sub('at::native::', '', n)[:120]}")
