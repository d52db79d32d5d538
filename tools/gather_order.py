#!/usr/bin/env python3
"""Diagnostic: the two-copy gather (real output, bench grid) on points in random order, in the per-model layout order and
in Morton order -- how much of its time is LDS bank conflicts between lanes whose stencils are unrelated.
usage: gather_order.py [N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
import torch  # noqa: E402
from efgp_hip import NufftPlan, kernel_timing, kernel_timing_read  # noqa: E402

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
x = (torch.rand(N, 2, generator=g, dtype=torch.float64) * 2 - 1).to(dev)
f = torch.complex(torch.randn(23, 23, generator=g, dtype=torch.float64), torch.randn(23, 23, generator=g, dtype=torch.float64)).to(dev)
h = 0.346
orders = {"random (caller's order)": x}
# layout-like order: bands of x_1 (8 fine cells ~ 1/12 of the box), then x_0
band = torch.floor((x[:, 1] + 1) * 6).to(torch.int64)
key = band * 4 + 0
idx = torch.argsort(band.to(torch.float64) * 4.0 + (x[:, 0] + 1))
orders["(band of x_1, x_0)"] = x[idx].contiguous()
# cell order: fine-grid cell of both coordinates (96 x 96)
c0 = torch.floor((x[:, 0] + 1) * 48).to(torch.int64)
c1 = torch.floor((x[:, 1] + 1) * 48).to(torch.int64)
orders["fine cell (x_0 cell, x_1 cell)"] = x[torch.argsort(c0 * 96 + c1)].contiguous()
for name, xs in orders.items():
    plan = NufftPlan(xs, h, 6e-8)
    for _ in range(3):
        plan.type2(f, (23, 23), real_only=True)
    kernel_timing(True, only="interp")
    for _ in range(10):
        plan.type2(f, (23, 23), real_only=True)
    ms, n = kernel_timing_read("interp")
    kernel_timing(False)
    print(f"N={N:.0e} {name:34s}: gather launch {1e3 * ms / n:7.1f} us", flush=True)
