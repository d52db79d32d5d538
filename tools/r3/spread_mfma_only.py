#!/usr/bin/env python3
"""Diagnostic: a few fused (y, ones) type-1 passes over the per-model layout (MFMA spreader) at the bench grid, for rocprofv3
--pmc runs.  usage: spread_mfma_only.py N [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
import torch  # noqa: E402
from efgp_hip import NufftPlan, PointSet  # noqa: E402

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
g = torch.Generator().manual_seed(0)
x = (torch.rand(N, 2, generator=g, dtype=torch.float64) * 2 - 1).cuda()
y = torch.randn(N, generator=g, dtype=torch.float64).cuda()
pts = PointSet(x, values=y)
plan = NufftPlan(x, 0.346, 1e-7, points=pts)
for _ in range(reps):
    Fy, v = plan.type1_pair(y, (23, 23), (45, 45))
torch.cuda.synchronize()
print("done", float(Fy.abs().sum()))
