#!/usr/bin/env python3
"""Diagnostic: stage timings of one hyper-gradient step (reference driver: test_timing_profiling.py:94-111)."""
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

dev = torch.device("cuda", 0)
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 5
D = int(sys.argv[3]) if len(sys.argv) > 3 else 2
if D == 2:
    x, y = bench.synth(N, 2, 1000, dev)
    kern = SquaredExponential(dimension=2, init_lengthscale=bench.LS, init_variance=bench.VAR)
    model = EFGPND(x, y, kern, sigmasq=bench.SIG2, eps=bench.EPS, estimate_params=False)
else:      # BASELINE configs[4]: 3-D Matern-3/2, hyper-gradient step
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from sweep_bench import CASES, synth
    x, y = synth(N, 3, 103, dev)
    model = EFGPND(x, y, CASES[3]["kernel"](), sigmasq=CASES[3]["sig"], eps=1e-3, estimate_params=False)
for _ in range(3):
    model.compute_gradients(trace_samples=T, cg_tol=1e-3)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 10
for _ in range(reps):
    g = model.compute_gradients(trace_samples=T, cg_tol=1e-3)
torch.cuda.synchronize()
print(f"N={N} T={T}: {1e3 * (time.perf_counter() - t0) / reps:.3f} ms per gradient step; grad={g.tolist()}")
model.compute_gradients(trace_samples=T, cg_tol=1e-3, do_profiling=True)
st = model.last_gradient_stats
print({k: v for k, v in st.items() if k not in ("stage_sec", "term1", "term2")})
for k, v in st["stage_sec"].items():
    print(f"  {k:28s} {1e3 * v:9.3f} ms")
