"""cProfile of the Python side of 200 hyper-gradient steps (T = 5, N = 1e6), sorted by cumulative time."""
import cProfile
import os
import pstats
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (".", "gp-quadrature_amd"):
    sys.path.insert(0, os.path.join(R, p))
import torch  # noqa: E402
from bench import synth, LS, VAR, SIG2, EPS, NUFFT_TOL  # noqa: E402
from efgpnd import EFGPND  # noqa: E402
from kernels.squared_exponential import SquaredExponential  # noqa: E402

dev = torch.device("cuda", 0)
x, y = synth(1_000_000, 2, 1000, dev)
m = EFGPND(x, y, SquaredExponential(dimension=2, init_lengthscale=LS, init_variance=VAR), sigmasq=SIG2, eps=EPS, nufft_eps=NUFFT_TOL,
           estimate_params=False)
for _ in range(100):
    m.compute_gradients(trace_samples=5, cg_tol=1e-3)
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    m.compute_gradients(trace_samples=5, cg_tol=1e-3)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
