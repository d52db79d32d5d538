"""Stochastic-variance solves of BASELINE configs[2] (usa_temp-shaped, Matern-5/2, J = 500 Hutchinson systems on the 128 x 128 grid):
time of predict(mean + stochastic variance) under the launch shapes of the one-workgroup-per-system cooperative kernel."""
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (".", "gp-quadrature_amd"):
    sys.path.insert(0, os.path.join(R, p))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from efgpnd import EFGPND  # noqa: E402
from kernels.matern import Matern  # noqa: E402

dev = torch.device("cuda", 0)
gz = np.load(os.path.join(R, "tests", "golden", "c3_matern52_usatemp.npz"))
x, y = torch.from_numpy(gz["x"]).to(dev), torch.from_numpy(gz["y"]).to(dev)
for eps in (1e-3, 1e-4):
    m = EFGPND(x, y, Matern(dimension=2, nu=2.5, init_lengthscale=0.1, init_variance=1.0), sigmasq=0.05, eps=eps, nufft_eps=1e-7,
               estimate_params=False, opts={"cg_tolerance": 1e-4, "mean_cg_warm_start": False})
    m.fit()
    M = int(m.last_fit_stats["feature_count"])
    g = torch.Generator().manual_seed(1)
    probes = (torch.randint(0, 2, (500, M), generator=g) * 2 - 1).to(torch.float64).to(dev)
    ref = None
    for setting in (os.environ.get("SHAPES", "default,2,4,8,16").split(",")):
        if setting == "default":
            os.environ.pop("EFGP_COOP_GMIN", None)
        else:
            os.environ["EFGP_COOP_GMIN"] = setting               # at least this many workgroups per system
        ts = []
        for _ in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, var = m.predict(x, variance_method="stochastic", variance_probes=probes)
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
        if ref is None:
            ref = var.clone()
        err = float((var - ref).abs().max() / ref.abs().max())
        print(f"eps {eps:g} mtot {m.last_fit_stats['mtot']} G_min {setting}: {sorted(ts)[1]:.2f} ms (min {min(ts):.2f}); max dev from first setting {err:.1e}", flush=True)
