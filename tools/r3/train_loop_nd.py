#!/usr/bin/env python3
"""Round 3: a training loop with MOVING hyper-parameters on a 1-D or 3-D model (EFGPND.optimize_hyperparameters, reference
efgpnd.py:1068-1226) in a FRESH process: total time, time per step, and how many distinct mode counts the walk visited -- every new
size is a new set of transform lengths (run-time compilation with hipFFT: EFGP_FFT_ROCFFT=1).  usage: train_loop_nd.py d N [iters] [lr]"""
import contextlib
import io
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
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lr = float(sys.argv[4]) if len(sys.argv) > 4 else 0.05
g = torch.Generator().manual_seed(7)
x = torch.rand(N, d, dtype=torch.float64, generator=g) * 2 - 1
y = torch.sin(3 * x).sum(dim=1) + math.sqrt(0.2) * torch.randn(N, dtype=torch.float64, generator=g)
x, y = x.cuda(), y.cuda()
kern = SquaredExponential(dimension=d, init_lengthscale={1: 0.03, 3: 0.2}.get(d, 0.2), init_variance=2.0)
model = EFGPND(x, y, kern, sigmasq=0.5, eps=1e-4, nufft_eps=1e-7, estimate_params=False, opts={"cg_tolerance": 1e-4})
seen = set()
inner = model.compute_gradients


def counted(*a, **k):
    r = inner(*a, **k)
    seen.add(int(model.last_gradient_stats["feature_count"]))
    return r


model.compute_gradients = counted
sink = io.StringIO()
torch.cuda.synchronize()
t0 = time.perf_counter()
with contextlib.redirect_stdout(sink):
    model.optimize_hyperparameters(lr=lr, max_iters=iters, trace_samples=3, cg_tol=1e-3, log_interval=10 ** 9)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"d={d} N={N}: {iters} training steps (+ final refit) in {dt:.2f} s = {1e3 * dt / iters:.1f} ms/step; {len(seen)} distinct feature counts "
      f"{sorted(seen)[:3]}..{sorted(seen)[-1]}; back end {'hipFFT' if os.environ.get('EFGP_FFT_ROCFFT') else 'in-house'}", flush=True)
