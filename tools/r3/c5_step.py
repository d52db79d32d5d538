#!/usr/bin/env python3
"""Round 3: BASELINE configs[4] at full size (3-D Matern-3/2, N = 5e6, eps 1e-3 -> mtot 57, 128^3 circulant grid): one fit and
a few hyper-gradient steps (T = 2 probes), wall time per step; run it under rocprofv3 --kernel-trace --stats for the table.
usage: c5_step.py [N] [steps] [eps]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
import torch
from efgpnd import EFGPND
from kernels.matern import Matern

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 5_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
eps = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3
d = 3
g = torch.Generator(device="cuda").manual_seed(21)
x = torch.rand(N, d, generator=g, dtype=torch.float64, device="cuda") * 2 - 1
y = torch.sin(3 * x[:, 0]) * torch.cos(4 * x[:, 1]) * torch.cos(2 * x[:, 2]) + 0.3 * torch.randn(N, generator=g, dtype=torch.float64, device="cuda")
k = Matern(dimension=d, nu=1.5, init_lengthscale=0.2, init_variance=1.5)
m = EFGPND(x, y, k, sigmasq=0.2, eps=eps, nufft_eps=1e-6, estimate_params=False, opts={"cg_tolerance": 1e-5, "mean_cg_warm_start": False})
torch.cuda.synchronize()
t0 = time.perf_counter()
m.fit()
torch.cuda.synchronize()
print(f"N={N} eps={eps}: first fit {1e3 * (time.perf_counter() - t0):.1f} ms, mtot {m.last_fit_stats['mtot']}, mean iters {int(m.last_fit_stats['mean_cg_iters'])}", flush=True)
for i in range(2):
    t0 = time.perf_counter()
    m.fit()
    torch.cuda.synchronize()
    print(f"  refit {1e3 * (time.perf_counter() - t0):.2f} ms", flush=True)
M = m.last_fit_stats["feature_count"]
V = torch.ones(2, M, dtype=torch.float64)
V[1, ::2] = -1
for i in range(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gr = m.compute_gradients(trace_samples=2, cg_tol=1e-3, probe_seed=99, probes_V=V)
    torch.cuda.synchronize()
    st = m.last_gradient_stats
    print(f"  gradient step {i}: {1e3 * (time.perf_counter() - t0):.2f} ms (mean iters {int(st['mean_cg_iters'])}, trace iters {int(st['trace_cg_iters'])})", flush=True)
