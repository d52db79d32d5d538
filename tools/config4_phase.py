#!/usr/bin/env python3
"""Diagnostic: BASELINE configs[3]'s hard single-GPU case (2-D SE, l = 0.05 -> mtot = 71, circulant grid 256^2) at N = 1e6:
fit + posterior mean at the N points, phases synchronised.  usage: config4_phase.py [N]"""
import sys
import time

sys.path.insert(0, "gp-quadrature_amd")
sys.path.insert(0, ".")
import torch  # noqa: E402
import bench  # noqa: E402
from efgpnd import EFGPND  # noqa: E402
from kernels.squared_exponential import SquaredExponential  # noqa: E402

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda", 0)
x, y = bench.synth(N, 2, 1000, dev)
k = SquaredExponential(dimension=2, init_lengthscale=0.05, init_variance=3.0)
m = EFGPND(x, y, k, sigmasq=0.2, eps=1e-4, nufft_eps=1e-7, estimate_params=False, opts={"cg_tolerance": 1e-4, "mean_cg_warm_start": False})


def timed(f):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = f()
    torch.cuda.synchronize()
    return r, 1e3 * (time.perf_counter() - t0)


for rep in range(4):
    _, t_fit = timed(lambda: m._compute_common_parameters(force_recompute=True))
    _, t_mean = timed(lambda: m.predict(x, return_variance=False))
    print(f"N={N} mtot={m.last_fit_stats['mtot']}: fit {t_fit:.3f} ms ({m.last_fit_stats['mean_cg_iters']} CG iterations), mean at the N points {t_mean:.3f} ms", flush=True)
