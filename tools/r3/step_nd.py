#!/usr/bin/env python3
"""Round 3: the bench's timed step (forced refit + posterior mean at the N points) on a 1-D or 3-D model, wall time per step; run
it under rocprofv3 --kernel-trace --stats for the kernel table.  usage: step_nd.py d N [lengthscale] [steps]"""
import math
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
import torch
from efgpnd import EFGPND
from kernels.squared_exponential import SquaredExponential

d = int(sys.argv[1])
N = int(float(sys.argv[2]))
ls = float(sys.argv[3]) if len(sys.argv) > 3 else {1: 0.05, 2: 0.2, 3: 0.3}[d]
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
g = torch.Generator().manual_seed(7)
x = torch.rand(N, d, dtype=torch.float64, generator=g) * 2 - 1
y = torch.sin(3 * x).sum(dim=1) + math.sqrt(0.2) * torch.randn(N, dtype=torch.float64, generator=g)
x, y = x.cuda(), y.cuda()
kern = SquaredExponential(dimension=d, init_lengthscale=ls, init_variance=2.0)
m = EFGPND(x, y, kern, sigmasq=0.2, eps=1e-4, nufft_eps=1e-7, estimate_params=False, opts={"cg_tolerance": 1e-4, "mean_cg_warm_start": False})
for _ in range(3):
    m._compute_common_parameters(force_recompute=True)
    m.predict(x, return_variance=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    m._compute_common_parameters(force_recompute=True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
for _ in range(steps):
    m.predict(x, return_variance=False)
torch.cuda.synchronize()
t3 = time.perf_counter()
print(f"d={d} N={N} ls={ls}: mtot {int(m.last_fit_stats['mtot'])}, mean iters {int(m.last_fit_stats['mean_cg_iters'])}; refit {1e3 * (t2 - t0) / steps:.3f} ms "
      f"(host returns after {1e3 * (t1 - t0) / steps:.3f}), predict {1e3 * (t3 - t2) / steps:.3f} ms", flush=True)
