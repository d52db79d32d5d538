#!/usr/bin/env python3
"""Cold-start latency of 1-D and 3-D models in a fresh process (tools/cold_start.py is the 2-D bench model): first and second
call of fit, prediction and gradient step.  usage: cold_start_nd.py d N [lengthscale]"""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
t0 = time.perf_counter()
import torch  # noqa: E402
t_torch = time.perf_counter() - t0
from efgpnd import EFGPND  # noqa: E402
from kernels.squared_exponential import SquaredExponential  # noqa: E402

d = int(sys.argv[1])
N = int(float(sys.argv[2]))
ls = float(sys.argv[3]) if len(sys.argv) > 3 else {1: 0.05, 2: 0.2, 3: 0.3}[d]
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
torch.cuda.synchronize()
g = torch.Generator().manual_seed(7)
x = torch.rand(N, d, dtype=torch.float64, generator=g) * 2 - 1
y = torch.sin(3 * x).sum(dim=1) + math.sqrt(0.2) * torch.randn(N, dtype=torch.float64, generator=g)
x, y = x.to(dev), y.to(dev)
torch.cuda.synchronize()


def lap(label, fn):
    torch.cuda.synchronize()
    t = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    print(f"{label:34s} {1e3 * (time.perf_counter() - t):10.2f} ms", flush=True)
    return r


print(f"d = {d}, N = {N}, lengthscale {ls}; import torch {1e3 * t_torch:.0f} ms")
kern = SquaredExponential(dimension=d, init_lengthscale=ls, init_variance=2.0)
model = lap("construct model", lambda: EFGPND(x, y, kern, sigmasq=0.2, eps=1e-4, nufft_eps=1e-7, estimate_params=False,
                                              opts={"cg_tolerance": 1e-4}))
lap("first fit", lambda: model.fit())
print(f"   mtot = {int(model.last_fit_stats['mtot'])}, CG iterations {int(model.last_fit_stats['mean_cg_iters'])}")
lap("second fit (forced)", lambda: model._compute_common_parameters(force_recompute=True))
lap("first predict at the N points", lambda: model.predict(x, return_variance=False))
lap("second predict", lambda: model.predict(x, return_variance=False))
lap("first gradient step (T=5)", lambda: model.compute_gradients(trace_samples=5, cg_tol=1e-3))
lap("second gradient step", lambda: model.compute_gradients(trace_samples=5, cg_tol=1e-3))
