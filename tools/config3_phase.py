#!/usr/bin/env python3
"""Diagnostic: BASELINE configs[2] (2-D Matern-5/2 on the PRISM usa_temp points: fit + posterior mean + stochastic
variance) split into its phases (synchronised wall times).  usage: config3_phase.py [eps]"""
import sys
import time

sys.path.insert(0, "gp-quadrature_amd")
sys.path.insert(0, ".")
import numpy as np  # noqa: E402
import torch  # noqa: E402
from efgpnd import EFGPND  # noqa: E402
from kernels.matern import Matern  # noqa: E402

eps = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-2
g = np.load("tests/golden/c3_matern52_usatemp.npz")
x = torch.from_numpy(g["x"]).cuda()
y = torch.from_numpy(g["y"]).cuda()
k = Matern(dimension=2, nu=2.5, init_lengthscale=0.1, init_variance=1.0)
m = EFGPND(x, y, k, sigmasq=0.1, eps=eps, estimate_params=False)
xn = torch.rand(2000, 2, dtype=torch.float64).cuda()


def timed(f):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = f()
    torch.cuda.synchronize()
    return r, 1e3 * (time.perf_counter() - t0)


for rep in range(3):
    _, t_fit = timed(lambda: m._compute_common_parameters(force_recompute=True))
    _, t_mean = timed(lambda: m.predict(xn, return_variance=False))
    _, t_var = timed(lambda: m.predict(xn, variance_method="stochastic", hutchinson_probes=200))
    print(f"eps={eps} mtot={m.last_fit_stats['mtot']} N={x.shape[0]}: fit {t_fit:.2f} ms (CG iters {m.last_fit_stats['mean_cg_iters']}), "
          f"mean at 2000 points {t_mean:.2f} ms, mean + stochastic variance (200 probes) {t_var:.2f} ms", flush=True)
