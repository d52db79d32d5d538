#!/usr/bin/env python3
"""Diagnostic: host-side time of the pieces of one fit step (no synchronisation inside the loop: the Python cost only).
usage: host_breakdown.py [N]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import efgpnd as E  # noqa: E402
from efgp_hip import ops  # noqa: E402
from kernels.squared_exponential import SquaredExponential  # noqa: E402

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda", 0)
x, y = bench.synth(N, bench.DIM, 1000, dev)
kern = SquaredExponential(dimension=bench.DIM, init_lengthscale=bench.LS, init_variance=bench.VAR)
model = E.EFGPND(x, y, kern, sigmasq=bench.SIG2, eps=bench.EPS, nufft_eps=bench.NUFFT_TOL, estimate_params=False,
                 opts={"cg_tolerance": bench.CG_TOL, "mean_cg_warm_start": False})
acc = {}


def wrap(obj, name, label):
    f = getattr(obj, name)

    def g(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
        return r
    setattr(obj, name, g)


wrap(E, "_Grid", "grid (get_xis, ws, upload)")
wrap(E, "NufftPlan", "NufftPlan()")
wrap(E, "_normal_equations", "type1_pair (enqueue)")
wrap(E, "ToeplitzOp", "ToeplitzOp()") if hasattr(E, "ToeplitzOp") else None
wrap(E, "ToeplitzND", "ToeplitzND()")
wrap(E, "cg_solve_mean_async", "cg_solve_mean_async (enqueue)")
for _ in range(10):
    model._compute_common_parameters(force_recompute=True)
    model.predict(x, return_variance=False)
torch.cuda.synchronize()
acc.clear()
K = 200
t0 = time.perf_counter()
tf = tp = 0.0
for _ in range(K):
    a = time.perf_counter()
    model._compute_common_parameters(force_recompute=True)
    b = time.perf_counter()
    model.predict(x, return_variance=False)
    c = time.perf_counter()
    tf += b - a
    tp += c - b
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host loop {1e3 * (t1 - t0) / K:.3f} ms/step (fit {1e3 * tf / K:.3f}, predict {1e3 * tp / K:.3f}); device complete {1e3 * (t2 - t0) / K:.3f} ms/step")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:36s} {1e6 * v / K:8.1f} us/step")
