#!/usr/bin/env python3
"""Diagnostic: host-side (Python + launch) profile of one bench step (fit + mean at N points)."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from efgpnd import EFGPND  # noqa: E402
from kernels.squared_exponential import SquaredExponential  # noqa: E402

dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
x, y = bench.synth(N, 2, 1000, dev)
kern = SquaredExponential(dimension=2, init_lengthscale=bench.LS, init_variance=bench.VAR)
model = EFGPND(x, y, kern, sigmasq=bench.SIG2, eps=bench.EPS, nufft_eps=bench.NUFFT_TOL, estimate_params=False,
               opts={"cg_tolerance": bench.CG_TOL, "mean_cg_warm_start": False})


def step():
    model._compute_common_parameters(force_recompute=True)
    return model.predict(x, return_variance=False)[0]


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    step()
torch.cuda.synchronize()
print(f"N={N}: {1e3 * (time.perf_counter() - t0) / 50:.3f} ms per step")
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
