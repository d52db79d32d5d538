#!/usr/bin/env python3
"""Fused (y, ones) type-1 pass at the bench grid: plain plan vs plan on the per-model point layout (MFMA spreader).
usage: spread_compare.py N [reps]   -> kernel microseconds (HIP events around the spread launch) and the difference."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
import torch  # noqa: E402
from efgp_hip import NufftPlan, PointSet, kernel_timing, kernel_timing_read  # noqa: E402

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
h = float(sys.argv[3]) if len(sys.argv) > 3 else 0.31
nm = int(sys.argv[4]) if len(sys.argv) > 4 else 23
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
x = (torch.rand(N, 2, generator=g, dtype=torch.float64) * 2 - 1).to(dev)
y = torch.randn(N, generator=g, dtype=torch.float64).to(dev)
t0 = time.perf_counter()
pts = PointSet(x, values=y)
torch.cuda.synchronize()
t1 = time.perf_counter()
res = {}
for name, plan in (("plain", NufftPlan(x, h, 6e-8)), ("layout", NufftPlan(x, h, 6e-8, points=pts))):
    out = plan.type1_pair(y, (nm, nm), (2 * nm - 1, 2 * nm - 1))
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    out = plan.type1_pair(y, (nm, nm), (2 * nm - 1, 2 * nm - 1))
    torch.cuda.synchronize()
    kernel_timing(True)
    for _ in range(reps):
        out = plan.type1_pair(y, (nm, nm), (2 * nm - 1, 2 * nm - 1))
    torch.cuda.synchronize()
    ms, n = kernel_timing_read("spread")
    kernel_timing(False)
    tw0 = time.perf_counter()
    for _ in range(reps):
        out = plan.type1_pair(y, (nm, nm), (2 * nm - 1, 2 * nm - 1))
    torch.cuda.synchronize()
    wall = (time.perf_counter() - tw0) / reps
    res[name] = out
    print(f"{name:7s} N={N:.0e} spread kernel {1e3 * ms / max(n, 1):8.1f} us  ({n} launches)   whole type1_pair {1e6 * wall:8.1f} us wall")
e1 = float((res["plain"][0] - res["layout"][0]).abs().max() / res["plain"][0].abs().max())
e2 = float((res["plain"][1] - res["layout"][1]).abs().max() / res["plain"][1].abs().max())
print(f"layout creation (bbox + attach) {1e3 * (t1 - t0):.2f} ms; rel diff F*y {e1:.2e}, v {e2:.2e}")
