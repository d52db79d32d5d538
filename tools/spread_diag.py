#!/usr/bin/env python3
"""Diagnostic: duration of the MFMA spread launch (fused (y, 1) pass on the per-model layout) with phases switched off
(EFGP_MFMA_DIAG: 1 no MFMA phase, 2 no window polynomials, 4 no LDS operand writes; results are then wrong on purpose).
usage: spread_diag.py [N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
import torch  # noqa: E402
from efgp_hip import NufftPlan, PointSet, kernel_timing, kernel_timing_read  # noqa: E402

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
x = (torch.rand(N, 2, generator=g, dtype=torch.float64) * 2 - 1).to(dev)
y = torch.randn(N, generator=g, dtype=torch.float64).to(dev)
pts = PointSet(x, values=y)
plan = NufftPlan(x, 0.346, 6e-8, points=pts)
for diag in (0, 1, 2, 4, 3, 7):
    os.environ["EFGP_MFMA_DIAG"] = str(diag)
    for _ in range(3):
        plan.type1_pair(y, (23, 23), (45, 45))
    kernel_timing(True, only="spread")
    for _ in range(10):
        plan.type1_pair(y, (23, 23), (45, 45))
    ms, n = kernel_timing_read("spread")
    kernel_timing(False)
    print(f"N={N} EFGP_MFMA_DIAG={diag}: spread launch {1e3 * ms / n:.1f} us", flush=True)
