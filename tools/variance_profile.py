#!/usr/bin/env python3
"""Diagnostic: posterior mean + stochastic (Hutchinson) variance at the bench model; stage times."""
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

dev = torch.device("cuda", 0)
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
B = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000
J = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
x, y = bench.synth(N, 2, 1000, dev)
xn, _ = bench.synth(B, 2, 7, dev)
kern = SquaredExponential(dimension=2, init_lengthscale=bench.LS, init_variance=bench.VAR)
model = EFGPND(x, y, kern, sigmasq=bench.SIG2, eps=bench.EPS, estimate_params=False)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mean, var = model.predict(xn, variance_method="stochastic", hutchinson_probes=J)
    torch.cuda.synchronize()
    print(f"N={N} B={B} J={J}: predict mean+variance {1e3 * (time.perf_counter() - t0):.2f} ms; var range [{float(var.min()):.3e}, {float(var.max()):.3e}]")
