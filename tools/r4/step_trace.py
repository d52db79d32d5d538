"""Host time of every entry inside efgp_gradient_step (EFGP_STEP_TRACE=1), one traced call after 300 warm steps."""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (".", "gp-quadrature_amd"):
    sys.path.insert(0, os.path.join(R, p))
os.environ["EFGP_STEP_TRACE"] = "1"
import torch  # noqa: E402
from bench import synth, LS, VAR, SIG2, EPS, NUFFT_TOL  # noqa: E402
from efgpnd import EFGPND  # noqa: E402
from kernels.squared_exponential import SquaredExponential  # noqa: E402

dev = torch.device("cuda", 0)
x, y = synth(1_000_000, 2, 1000, dev)
m = EFGPND(x, y, SquaredExponential(dimension=2, init_lengthscale=LS, init_variance=VAR), sigmasq=SIG2, eps=EPS, nufft_eps=NUFFT_TOL,
           estimate_params=False)
fd = os.dup(2)
null = os.open(os.devnull, os.O_WRONLY)
os.dup2(null, 2)
for _ in range(300):
    m.compute_gradients(trace_samples=5, cg_tol=1e-3)
torch.cuda.synchronize()
os.dup2(fd, 2)
import time  # noqa: E402
import efgp_hip.ops as ops  # noqa: E402
real = ops.gradient_step


def timed(*a, **k):
    t0 = time.perf_counter()
    out = real(*a, **k)
    sys.stderr.write(f"ops.gradient_step host time {1e6 * (time.perf_counter() - t0):.1f} us\n")
    return out


ops.gradient_step = timed
for _ in range(3):
    t0 = time.perf_counter()
    m.compute_gradients(trace_samples=5, cg_tol=1e-3)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    sys.stderr.write(f"---- compute_gradients {1e6 * (t1 - t0):.1f} us\n")
