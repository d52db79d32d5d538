"""Is the first process on a fresh box slower, and for how long?  Times consecutive blocks of 25 headline steps (fit + mean at N = 1e6)
from process start; prints ms per step of every block with the wall-clock offset of the block."""
import os
import sys
import time

T0 = time.perf_counter()
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (".", "gp-quadrature_amd"):
    sys.path.insert(0, os.path.join(R, p))
import torch  # noqa: E402
from bench import synth, LS, VAR, SIG2, EPS, NUFFT_TOL, CG_TOL  # noqa: E402
from efgpnd import EFGPND  # noqa: E402
from kernels.squared_exponential import SquaredExponential  # noqa: E402

dev = torch.device("cuda", 0)
x, y = synth(1_000_000, 2, 1000, dev)
model = EFGPND(x, y, SquaredExponential(dimension=2, init_lengthscale=LS, init_variance=VAR), sigmasq=SIG2, eps=EPS, nufft_eps=NUFFT_TOL,
               estimate_params=False, opts={"cg_tolerance": CG_TOL, "mean_cg_warm_start": False})
print(f"model ready at {time.perf_counter() - T0:.2f} s after process start", flush=True)
out = []
for blk in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(25):
        model._compute_common_parameters(force_recompute=True)
        model.predict(x, return_variance=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    out.append((t - T0, 1e3 * dt / 25))
print(" ".join(f"{a:.2f}s:{b:.3f}" for a, b in out))
