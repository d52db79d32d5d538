#!/usr/bin/env python3
"""Cold-start latency in a fresh process: library load, first fit, first prediction, first gradient step, and the second of
each.  usage: cold_start.py [N]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
sys.path.insert(0, ROOT)
t0 = time.perf_counter()
import torch  # noqa: E402
t_torch = time.perf_counter() - t0
import bench  # noqa: E402
from efgpnd import EFGPND  # noqa: E402
from kernels.squared_exponential import SquaredExponential  # noqa: E402

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
torch.cuda.synchronize()
x, y = bench.synth(N, bench.DIM, 1000, dev)
torch.cuda.synchronize()


def lap(label, fn):
    torch.cuda.synchronize()
    t = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    print(f"{label:34s} {1e3 * (time.perf_counter() - t):10.2f} ms")
    return r


print(f"import torch {1e3 * t_torch:.0f} ms")
kern = SquaredExponential(dimension=bench.DIM, init_lengthscale=bench.LS, init_variance=bench.VAR)
model = lap("construct model", lambda: EFGPND(x, y, kern, sigmasq=bench.SIG2, eps=bench.EPS, nufft_eps=bench.NUFFT_TOL,
                                              estimate_params=False, opts={"cg_tolerance": bench.CG_TOL}))
lap("first fit", lambda: model.fit())
lap("second fit (forced)", lambda: model._compute_common_parameters(force_recompute=True))
lap("first predict at the N points", lambda: model.predict(x, return_variance=False))
lap("second predict", lambda: model.predict(x, return_variance=False))
lap("first gradient step (T=5)", lambda: model.compute_gradients(trace_samples=5, cg_tol=1e-3))
lap("second gradient step", lambda: model.compute_gradients(trace_samples=5, cg_tol=1e-3))
xq = x[:2000].contiguous()
import contextlib  # noqa: E402
import io  # noqa: E402
with contextlib.redirect_stdout(io.StringIO()):
    lap_v1 = lambda: model.predict(xq, return_variance=True, variance_method="stochastic", hutchinson_probes=100)  # noqa: E731
    t = time.perf_counter(); lap_v1(); torch.cuda.synchronize(); t1 = time.perf_counter() - t  # noqa: E702
    t = time.perf_counter(); lap_v1(); torch.cuda.synchronize(); t2 = time.perf_counter() - t  # noqa: E702
print(f"{'first stochastic variance (100 probes)':34s} {1e3 * t1:10.2f} ms")
print(f"{'second stochastic variance':34s} {1e3 * t2:10.2f} ms")
