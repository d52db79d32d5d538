"""Where the wall time of a synchronised one-call gradient step goes on the host: before the library call, inside it (enqueue),
waiting for the read-back, after it.  Medians over 300 steps."""
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
import efgp_hip.ops as ops  # noqa: E402

dev = torch.device("cuda", 0)
x, y = synth(1_000_000, 2, 1000, dev)
m = EFGPND(x, y, SquaredExponential(dimension=2, init_lengthscale=LS, init_variance=VAR), sigmasq=SIG2, eps=EPS, nufft_eps=NUFFT_TOL,
           estimate_params=False)
for _ in range(300):
    m.compute_gradients(trace_samples=5, cg_tol=1e-3)
marks = {}
real_step = ops.gradient_step
real_cpu = torch.Tensor.cpu


def timed_step(*a, **k):
    marks["call0"] = time.perf_counter()
    out = real_step(*a, **k)
    marks["call1"] = time.perf_counter()
    return out


def timed_cpu(self, *a, **k):
    t0 = time.perf_counter()
    out = real_cpu(self, *a, **k)
    if "call1" in marks and "cpu0" not in marks:
        marks["cpu0"], marks["cpu1"] = t0, time.perf_counter()
    return out


ops.gradient_step = timed_step
torch.Tensor.cpu = timed_cpu
rows = []
for _ in range(300):
    marks.clear()
    t0 = time.perf_counter()
    m.compute_gradients(trace_samples=5, cg_tol=1e-3)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    rows.append((marks["call0"] - t0, marks["call1"] - marks["call0"], marks["cpu0"] - marks["call1"], marks["cpu1"] - marks["cpu0"],
                 t1 - marks["cpu1"], t2 - t1, t2 - t0))
med = [sorted(r[i] for r in rows)[len(rows) // 2] * 1e6 for i in range(7)]
print("median us: before call %.1f | library call %.1f | call -> read-back %.1f | read-back wait %.1f | after %.1f | final sync %.1f | "
      "step %.1f" % tuple(med))
