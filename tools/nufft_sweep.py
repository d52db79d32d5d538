#!/usr/bin/env python3
"""Diagnostic: timing of the NUFFT entry points over grid sizes / dimensions (HIP events in the library)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
import torch  # noqa: E402
from efgp_hip import NufftPlan, kernel_timing, kernel_timing_read  # noqa: E402

dev = torch.device("cuda", 0)
cases = [(2, 1_000_000, 23, 1e-7), (2, 1_000_000, 71, 1e-7), (2, 10_000_000, 71, 1e-7), (3, 1_000_000, 19, 1e-5),
         (3, 1_000_000, 29, 1e-5), (1, 10_000_000, 35, 1e-7)]
if len(sys.argv) > 1:
    cases = [tuple(float(v) if "e-" in v else int(float(v)) for v in a.split(",")) for a in sys.argv[1:]]
for d, N, mtot, tol in cases:
    g = torch.Generator().manual_seed(0)
    x = (torch.rand(N, d, generator=g, dtype=torch.float64) * 2 - 1).to(dev)
    y = torch.randn(N, generator=g, dtype=torch.float64).to(dev)
    if os.environ.get("SWEEP_PRESORT"):                                          # spatially pre-sorted points (row-major cells)
        q = int(os.environ["SWEEP_PRESORT"])
        key = ((x[:, 0] + 1) * (q / 2)).long().clamp_(0, q - 1)
        for j in range(1, d):
            key = key * q + ((x[:, j] + 1) * (q / 2)).long().clamp_(0, q - 1)
        x = x[torch.argsort(key)].contiguous()
    m = (mtot - 1) // 2
    NufftPlan(x, 0.45, tol).type1_pair(y, (mtot,) * d, (4 * m + 1,) * d)      # warm caches (windows, FFT plans, scratch)
    torch.cuda.synchronize()
    tb = time.perf_counter()
    plan = NufftPlan(x, 0.45, tol)
    plan.type1_pair(y, (mtot,) * d, (4 * m + 1,) * d)
    torch.cuda.synchronize()
    first_ms = 1e3 * (time.perf_counter() - tb)                                 # includes the per-plan tile binning
    for _ in range(2):
        Fy, v = plan.type1_pair(y, (mtot,) * d, (4 * m + 1,) * d)
        out = plan.type2(Fy, (mtot,) * d, real_only=True)
    torch.cuda.synchronize()
    kernel_timing(True)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        Fy, v = plan.type1_pair(y, (mtot,) * d, (4 * m + 1,) * d)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(reps):
        out = plan.type2(Fy, (mtot,) * d, real_only=True)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    sp, ns = kernel_timing_read("spread")
    ip, ni = kernel_timing_read("interp")
    kernel_timing(False)
    print(f"d={d} N={N:.0e} mtot={mtot} tol={tol}: first call on a new plan {first_ms:.3f} ms; type1_pair {1e3 * (t1 - t0) / reps:.3f} ms (spread kernel {1e3 * sp / max(ns, 1):.1f} us)  "
          f"type2 {1e3 * (t2 - t1) / reps:.3f} ms (interp kernel {1e3 * ip / max(ni, 1):.1f} us)")
