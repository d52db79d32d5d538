#!/usr/bin/env python3
"""Is the hyper-gradient step host-bound?  Host time of K steps (no synchronisation in the loop) against device-complete time,
and a cProfile of the Python side.  usage: grad_host_ahead.py [N] [K]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from efgpnd import EFGPND  # noqa: E402
from kernels.squared_exponential import SquaredExponential  # noqa: E402

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda", 0)
x, y = bench.synth(N, bench.DIM, 1000, dev)
kern = SquaredExponential(dimension=bench.DIM, init_lengthscale=bench.LS, init_variance=bench.VAR)
model = EFGPND(x, y, kern, sigmasq=bench.SIG2, eps=bench.EPS, nufft_eps=bench.NUFFT_TOL, estimate_params=False,
               opts={"cg_tolerance": bench.CG_TOL})
for _ in range(5):
    model.compute_gradients(trace_samples=5, cg_tol=1e-3)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    g = model.compute_gradients(trace_samples=5, cg_tol=1e-3)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"N={N}: host loop {1e3 * (t1 - t0) / K:.3f} ms/step, device complete {1e3 * (t2 - t0) / K:.3f} ms/step")
import cProfile
import pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(K):
    model.compute_gradients(trace_samples=5, cg_tol=1e-3)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)

# the gradient function alone, nothing read back (stats_out=None): the host's enqueue time per step against the device's
from efgpnd import efgpnd_gradient_batched  # noqa: E402
dd = model._device_data()
kw = dict(sigmasq=model._gp_params.sig2, kernel=kern, eps=bench.EPS, trace_samples=5, nufft_eps=bench.EPS * 0.1, cg_tol=1e-3,
          domain_length=dd["L"], y_norm_sq=dd["yy"], points=dd["points"])
for _ in range(3):
    efgpnd_gradient_batched(dd["x"], dd["y"], **kw)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    efgpnd_gradient_batched(dd["x"], dd["y"], **kw)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"no read-back: host enqueue {1e3 * (t1 - t0) / K:.3f} ms/step, device complete {1e3 * (t2 - t0) / K:.3f} ms/step")

# host clock per stage (no synchronisation inside the step: these are ENQUEUE times, the last one includes the wait)
acc = {}
for _ in range(K):
    st = {}
    efgpnd_gradient_batched(dd["x"], dd["y"], stats_out=st, **kw)
    for k_, v_ in st["stage_sec"].items():
        acc[k_] = acc.get(k_, 0.0) + v_
print("host enqueue per stage (us):", {k_: round(1e6 * v_ / K, 1) for k_, v_ in acc.items()})
