"""A/B of an environment switch on the synchronised hyper-gradient step (T = 5, N = 1e6): alternating blocks of 100 steps in ONE
process, median step of every block.  usage: grad_step_ab.py ENV_NAME [blocks]"""
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (".", "gp-quadrature_amd"):
    sys.path.insert(0, os.path.join(R, p))
import torch  # noqa: E402
from bench import synth, LS, VAR, SIG2, EPS, NUFFT_TOL  # noqa: E402
from efgpnd import EFGPND  # noqa: E402
from kernels.squared_exponential import SquaredExponential  # noqa: E402

name = sys.argv[1]
dev = torch.device("cuda", 0)
x, y = synth(1_000_000, 2, 1000, dev)
m = EFGPND(x, y, SquaredExponential(dimension=2, init_lengthscale=LS, init_variance=VAR), sigmasq=SIG2, eps=EPS, nufft_eps=NUFFT_TOL,
           estimate_params=False)
for _ in range(300):
    m.compute_gradients(trace_samples=5, cg_tol=1e-3)
res = {0: [], 1: []}
for blk in range(int(sys.argv[2]) if len(sys.argv) > 2 else 8):
    on = blk & 1
    if on:
        os.environ[name] = "1"
    else:
        os.environ.pop(name, None)
    ts = []
    for _ in range(100):
        t0 = time.perf_counter()
        m.compute_gradients(trace_samples=5, cg_tol=1e-3)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    res[on].append(1e3 * sorted(ts)[50])
os.environ.pop(name, None)
print(f"{name} unset: median step per block {[round(v, 4) for v in res[0]]} ms;  {name}=1: {[round(v, 4) for v in res[1]]} ms")
