"""Wall times (synchronised, 12 samples each) of the fits whose mean solve is a cooperative launch: configs[3]-hard (N = 1e6, mtot 71)
and configs[2] usa_temp Matern-5/2 at eps 1e-3 / 1e-4."""
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (".", "gp-quadrature_amd"):
    sys.path.insert(0, os.path.join(R, p))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from bench import synth  # noqa: E402
from efgpnd import EFGPND  # noqa: E402
from kernels.squared_exponential import SquaredExponential  # noqa: E402
from kernels.matern import Matern  # noqa: E402

dev = torch.device("cuda", 0)


def samples(m, n=12):
    out = []
    for _ in range(3):
        m._compute_common_parameters(force_recompute=True)
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m._compute_common_parameters(force_recompute=True)
        torch.cuda.synchronize()
        out.append(1e3 * (time.perf_counter() - t0))
    return " ".join(f"{v:.3f}" for v in out), int(m.last_fit_stats["mean_cg_iters"])


x, y = synth(1_000_000, 2, 1000, dev)
m = EFGPND(x, y, SquaredExponential(dimension=2, init_lengthscale=0.05, init_variance=2.0), sigmasq=0.2, eps=1e-4, nufft_eps=1e-7,
           estimate_params=False, opts={"cg_tolerance": 1e-4, "mean_cg_warm_start": False})
print("configs3-hard fit ms:", *samples(m))
del m, x, y
z = np.load(os.path.join(R, "tests", "golden", "c3_matern52_usatemp.npz"))
xt = torch.from_numpy(z["x"]).to(dev)
yt = torch.from_numpy(z["y"]).to(dev)
for eps in (1e-3, 1e-4):
    m = EFGPND(xt, yt, Matern(dimension=2, nu=2.5, init_lengthscale=0.1, init_variance=1.0), sigmasq=0.05, eps=eps, nufft_eps=1e-7,
               estimate_params=False, opts={"cg_tolerance": 1e-4, "mean_cg_warm_start": False})
    print(f"usa_temp eps {eps:g} fit ms:", *samples(m), "mtot", m.last_fit_stats["mtot"])
