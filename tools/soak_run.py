#!/usr/bin/env python3
"""Soak run: many fits / predictions / gradient steps with changing hyper-parameters (changing grids, windows,
FFT plans, tile binnings); prints device memory in use by the process at intervals to expose leaks in the
library's own allocations (torch does not see them)."""
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
import torch  # noqa: E402
from efgpnd import EFGPND  # noqa: E402
from kernels.squared_exponential import SquaredExponential  # noqa: E402
from kernels.matern import Matern  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
N = 200_000
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
x2 = torch.rand(N, 2, generator=g, dtype=torch.float64, device=dev) * 2 - 1
y2 = torch.sin(3 * x2[:, 0]) * torch.cos(4 * x2[:, 1]) + 0.3 * torch.randn(N, generator=g, dtype=torch.float64, device=dev)
x3 = torch.rand(50_000, 3, generator=g, dtype=torch.float64, device=dev) * 2 - 1
y3 = torch.sin(2 * x3[:, 0]) * torch.cos(3 * x3[:, 1]) * torch.cos(x3[:, 2]) + 0.3 * torch.randn(50_000, generator=g, dtype=torch.float64, device=dev)
xn = torch.rand(5000, 2, generator=g, dtype=torch.float64, device=dev) * 2 - 1


def used():
    free, total = torch.cuda.mem_get_info(dev)
    return (total - free) / 2 ** 20


t0 = time.perf_counter()
base = None
for r in range(rounds):
    nell = int(os.environ.get("SOAK_NELL", "101"))             # distinct lengthscales (-> distinct grid / FFT sizes)
    ell = 0.08 + 0.4 * ((r * 37) % nell) / nell
    k = SquaredExponential(dimension=2, init_lengthscale=ell, init_variance=1.5) if r % 3 else \
        Matern(dimension=2, nu=2.5, init_lengthscale=ell + 0.1, init_variance=1.0)
    m = EFGPND(x2, y2, k, sigmasq=0.2, eps=1e-3 if r % 2 else 1e-4, estimate_params=False)
    mean, var = m.predict(xn, variance_method="stochastic", hutchinson_probes=16)
    assert torch.isfinite(mean).all() and torch.isfinite(var).all()
    m.compute_gradients(trace_samples=3)
    if r % 10 == 0 and not os.environ.get("SOAK_NO_3D"):
        k3 = Matern(dimension=3, nu=1.5, init_lengthscale=0.5 + 0.01 * (r % 7), init_variance=1.0)
        m3 = EFGPND(x3, y3, k3, sigmasq=0.2, eps=1e-2, estimate_params=False)
        mean3, _ = m3.predict(x3[:2000], return_variance=False)
        assert torch.isfinite(mean3).all()
        del m3
    del m
    if r % 25 == 0 or r == rounds - 1:
        gc.collect()
        torch.cuda.synchronize()
        u = used()
        if r == 25:
            base = u
        print(f"round {r:4d}: {u:9.1f} MiB in use, torch allocated {torch.cuda.memory_allocated(dev) / 2 ** 20:8.1f} MiB, "
              f"torch reserved {torch.cuda.memory_reserved(dev) / 2 ** 20:8.1f} MiB, {time.perf_counter() - t0:6.1f} s", flush=True)
torch.cuda.synchronize()
final = used()
print(f"growth after warm-up (round 25 -> end): {final - base:.1f} MiB")
