#!/usr/bin/env python3
"""Per-step times of a training loop with MOVING hyper-parameters (every step has a new grid spacing h, some a new mtot):
what a step costs when plans / windows / FFT sizes cannot be reused.  usage: train_loop_steps.py [N] [iters] [T] [lr]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.optim import Adam  # noqa: E402
import bench  # noqa: E402
from efgpnd import EFGPND  # noqa: E402
from kernels.squared_exponential import SquaredExponential  # noqa: E402

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
T = int(sys.argv[3]) if len(sys.argv) > 3 else 5
lr = float(sys.argv[4]) if len(sys.argv) > 4 else 0.01
dev = torch.device("cuda", 0)
x, y = bench.synth(N, bench.DIM, 1000, dev)
kern = SquaredExponential(dimension=bench.DIM, init_lengthscale=bench.LS, init_variance=bench.VAR)
model = EFGPND(x, y, kern, sigmasq=bench.SIG2, eps=bench.EPS, nufft_eps=bench.NUFFT_TOL, estimate_params=False,
               opts={"cg_tolerance": bench.CG_TOL})
opt = Adam(model._gp_params.parameters(), lr=lr)
rows = []
for it in range(iters):
    opt.zero_grad()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.compute_gradients(trace_samples=T, cg_tol=1e-3, do_profiling=(os.environ.get("STAGES") == "1"))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = model.last_gradient_stats
    rows.append((it, dt * 1e3, st["mtot"], int(st["mean_cg_iters"]), int(st["trace_cg_iters"])))
    opt.step()
for r in rows:
    print(f"step {r[0]:3d}: {r[1]:8.3f} ms  mtot {r[2]:3d}  mean iters {r[3]:4d}  trace iters {r[4]:4d}")
ts = sorted(r[1] for r in rows[5:])
print(f"median {ts[len(ts) // 2]:.3f} ms, mean {sum(ts) / len(ts):.3f} ms, max {ts[-1]:.3f} ms (steps 5..)")
