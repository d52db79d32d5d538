#!/usr/bin/env python3
"""Training-loop rate: EFGPND.optimize_hyperparameters (reference: efgpnd.py:1068-1226) at the bench configuration,
steps per second and the share spent outside compute_gradients.  usage: train_loop_rate.py [N] [iters] [T]"""
import contextlib
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from efgpnd import EFGPND  # noqa: E402
from kernels.squared_exponential import SquaredExponential  # noqa: E402

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
T = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda", 0)
x, y = bench.synth(N, bench.DIM, 1000, dev)
kern = SquaredExponential(dimension=bench.DIM, init_lengthscale=bench.LS, init_variance=bench.VAR)
model = EFGPND(x, y, kern, sigmasq=bench.SIG2, eps=bench.EPS, nufft_eps=bench.NUFFT_TOL, estimate_params=False,
               opts={"cg_tolerance": bench.CG_TOL})
sink = io.StringIO()
with contextlib.redirect_stdout(sink):
    model.optimize_hyperparameters(lr=0.01, max_iters=10, trace_samples=T, cg_tol=1e-3)
torch.cuda.synchronize()
t_grad = [0.0]
inner = model.compute_gradients


def timed(*a, **k):
    t0 = time.perf_counter()
    r = inner(*a, **k)
    t_grad[0] += time.perf_counter() - t0
    return r


model.compute_gradients = timed
t0 = time.perf_counter()
with contextlib.redirect_stdout(sink):
    model.optimize_hyperparameters(lr=0.01, max_iters=iters, trace_samples=T, cg_tol=1e-3, log_interval=10 ** 9)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"N={N} T={T}: {iters} training steps in {dt * 1e3:.1f} ms = {1e3 * dt / iters:.3f} ms/step ({iters / dt:.0f} steps/s); "
      f"compute_gradients {1e3 * t_grad[0] / iters:.3f} ms/step, loop around it {1e3 * (dt - t_grad[0]) / iters:.3f} ms/step "
      f"(includes the final refit)")
print("final hypers:", {k_: round(v_[-1], 5) for k_, v_ in model.training_log.items() if k_ in ("lengthscale", "variance", "sigmasq")})
