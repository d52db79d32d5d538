#!/usr/bin/env python3
"""Diagnostic: a few real-output type-2 passes at the bench grid, for rocprofv3 --pmc runs.  usage: interp_only.py N [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
import torch  # noqa: E402
from efgp_hip import NufftPlan  # noqa: E402

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
x = (torch.rand(N, 2, generator=g, dtype=torch.float64) * 2 - 1).to(dev)
f = torch.complex(torch.randn(23, 23, generator=g, dtype=torch.float64), torch.randn(23, 23, generator=g, dtype=torch.float64)).to(dev)
plan = NufftPlan(x, 0.346, 1e-7)
for _ in range(reps):
    out = plan.type2(f, (23, 23), real_only=True)
torch.cuda.synchronize()
print("done", float(out.abs().sum()))
