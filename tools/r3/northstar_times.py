#!/usr/bin/env python3
"""Round 3: spread / gather microseconds per launch of the fit + mean step (HIP events inside the library), as bench.py's
north_star leg measures them.  usage: northstar_times.py [N] [reps]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
import torch
import bench
from efgpnd import EFGPND
from efgp_hip import kernel_timing, kernel_timing_read
from kernels.squared_exponential import SquaredExponential

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
x, y = bench.synth(N, 2, 2000, dev)
kern = SquaredExponential(dimension=2, init_lengthscale=bench.LS, init_variance=bench.VAR)
model = EFGPND(x, y, kern, sigmasq=bench.SIG2, eps=bench.EPS, nufft_eps=bench.NUFFT_TOL, estimate_params=False,
               opts={"cg_tolerance": bench.CG_TOL, "mean_cg_warm_start": False})

def step():
    model._compute_common_parameters(force_recompute=True)
    return model.predict(x, return_variance=False)[0]

for _ in range(3):
    step()
torch.cuda.synchronize()
kernel_timing(True)
t0 = time.perf_counter()
for _ in range(reps):
    m = step()
torch.cuda.synchronize()
el = time.perf_counter() - t0
sp, ns = kernel_timing_read("spread")
ip, ni = kernel_timing_read("interp")
kernel_timing(False)
tag = os.environ.get("EFGP_NO_DENSE_SIGMA")
print(f"N={N} dense_sigma={'off' if tag else 'on'}: spread {1e3 * sp / ns:.1f} us, gather {1e3 * ip / ni:.1f} us, sum {1e3 * (sp / ns + ip / ni):.1f} us; "
      f"step {1e3 * el / reps:.3f} ms; iters {model.last_fit_stats['mean_cg_iters']}; mean checksum {float(m.sum()):.10e}", flush=True)
