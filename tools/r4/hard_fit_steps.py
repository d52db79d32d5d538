"""Fits of the hard case of configs[3] (2-D SE l = 0.05, N = 1e6, mtot 71, cooperative solve on the 192 x 192 grid), synchronised:
the workload for a kernel-trace timeline (tools/r4/step_timeline.py ... spectral_weights)."""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (".", "gp-quadrature_amd"):
    sys.path.insert(0, os.path.join(R, p))
import torch  # noqa: E402
from bench import synth  # noqa: E402
from efgpnd import EFGPND  # noqa: E402
from kernels.squared_exponential import SquaredExponential  # noqa: E402

dev = torch.device("cuda", 0)
x, y = synth(1_000_000, 2, 1000, dev)
m = EFGPND(x, y, SquaredExponential(dimension=2, init_lengthscale=0.05, init_variance=2.0), sigmasq=0.2, eps=1e-4, nufft_eps=1e-7,
           estimate_params=False, opts={"cg_tolerance": 1e-4, "mean_cg_warm_start": False})
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    m._compute_common_parameters(force_recompute=True)
    torch.cuda.synchronize()
