#!/usr/bin/env python3
"""Diagnostic: the drop-in scenario with HOST tensors (as the reference's scripts pass them): model construction
(upload of x, y), fit, posterior mean at the N training points returned to the host.  Prints ms and the implied rate."""
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
x, y = bench.synth(N, 2, 1000, "cpu")
for pinned in (False, True):
    xs, ys = (x.pin_memory(), y.pin_memory()) if pinned else (x, y)
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        kern = SquaredExponential(dimension=2, init_lengthscale=bench.LS, init_variance=bench.VAR)
        m = EFGPND(xs, ys, kern, sigmasq=bench.SIG2, eps=bench.EPS, nufft_eps=bench.NUFFT_TOL, estimate_params=False,
                   opts={"cg_tolerance": bench.CG_TOL})
        mean, _ = m.predict(xs, return_variance=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    assert mean.device.type == "cpu"
    print(f"N={N} host tensors ({'pinned' if pinned else 'pageable'}): construct + fit + mean back on the host {1e3 * dt:.2f} ms "
          f"({1.0 / dt:.0f} fits/s; {N * 40 / dt / 1e9:.1f} GB/s of x, y in and mean out)")
