import sys, time
sys.path.insert(0, "gp-quadrature_amd"); sys.path.insert(0, ".")
import numpy as np, torch
from efgpnd import EFGPND
from kernels.matern import Matern
g = np.load("tests/golden/c3_matern52_usatemp.npz")
x = torch.from_numpy(g["x"]).cuda(); y = torch.from_numpy(g["y"]).cuda()
for eps in (1e-2, 1e-3, 1e-4):
    k = Matern(dimension=2, nu=2.5, init_lengthscale=0.1, init_variance=1.0)
    m = EFGPND(x, y, k, sigmasq=0.1, eps=eps, estimate_params=False)
    xn = torch.rand(2000, 2, dtype=torch.float64).cuda()
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m._compute_common_parameters(force_recompute=True)
        mean, var = m.predict(xn, variance_method="stochastic", hutchinson_probes=200)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"eps={eps}: mtot={m.last_fit_stats['mtot']} iters={m.last_fit_stats['mean_cg_iters']} fit+mean+variance(200 probes) {1e3*dt:.1f} ms")
