"""Device timeline of one hyper-gradient step from a rocprofv3 --kernel-trace CSV: every kernel with its duration and the idle gap
before it.  usage: step_timeline.py <kernel_trace.csv> [first-kernel-substring]   (a step = from one weights kernel to the next)"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
key = sys.argv[2] if len(sys.argv) > 2 else "spectral_weights"
starts = [i for i, r in enumerate(rows) if key in r[2]]
steps = []
for a, b in zip(starts[:-1], starts[1:]):
    seg = rows[a:b]
    busy = sum(e - s for s, e, _ in seg)
    span = seg[-1][1] - seg[0][0]
    steps.append((span, busy, a, b))
steps_sorted = sorted(steps)
span, busy, a, b = steps_sorted[len(steps_sorted) // 2]
print(f"{len(steps)} steps; median step: {b - a} kernels, first start -> last end {span / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, "
      f"gaps {(span - busy) / 1e3:.1f} us; period to the next step {(rows[b][0] - rows[a][0]) / 1e3:.1f} us")
prev = None
for s, e, n in rows[a:b]:
    gap = 0.0 if prev is None else (s - prev) / 1e3
    print(f"  gap {gap:6.1f}  run {(e - s) / 1e3:7.1f}  {n[:110]}")
    prev = e
