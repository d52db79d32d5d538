#!/usr/bin/env python3
"""North-star sweep: GP-fits/s, CG-iter/s and NUFFT spread/gather rates for N = 1e5..1e7, d = 1..3 on one MI355X.

One JSON line per case on stdout.  A case = fit + posterior mean at the N training points (same step as bench.py).
Bytes for the roofline fractions: SURVEY 8(d): type-1 / type-2 = N (8 d + 16) B (complex strengths; the fused real
pair of the fit reads N (8 d + 8) B once for two transforms -- both accountings are printed)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
import torch  # noqa: E402
from efgpnd import EFGPND  # noqa: E402
from efgp_hip import kernel_timing, kernel_timing_read  # noqa: E402
from kernels.squared_exponential import SquaredExponential  # noqa: E402
from kernels.matern import Matern  # noqa: E402

HBM = 8000.0   # GB/s
CASES = {1: dict(kernel=lambda: SquaredExponential(dimension=1, init_lengthscale=0.1, init_variance=2.0), sig=0.1),
         2: dict(kernel=lambda: SquaredExponential(dimension=2, init_lengthscale=0.2, init_variance=2.0), sig=0.2),
         3: dict(kernel=lambda: Matern(dimension=3, nu=1.5, init_lengthscale=0.5, init_variance=1.0), sig=0.2)}


def synth(N, d, seed, dev):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.rand(N, d, generator=g, dtype=torch.float64, device=dev) * 2 - 1
    f = torch.sin(3 * x[:, 0])
    if d > 1:
        f = f * torch.cos(4 * x[:, 1]) + 0.7 * torch.sin(2 * torch.pi * (x[:, 0] ** 2 + x[:, 1] ** 2))
    if d > 2:
        f = f * torch.cos(2 * x[:, 2])
    return x, f + 0.2 ** 0.5 * torch.randn(N, generator=g, dtype=torch.float64, device=dev)


def main():
    dev = torch.device("cuda", 0)
    dims = [int(a) for a in os.environ.get("SWEEP_DIMS", "1,2,3").split(",")]
    sizes = [int(float(a)) for a in os.environ.get("SWEEP_N", "1e5,1e6,1e7").split(",")]
    eps = {1: 1e-4, 2: 1e-4, 3: 1e-3}
    for d in dims:
        for N in sizes:
            x, y = synth(N, d, 100 + d, dev)
            c = CASES[d]
            model = EFGPND(x, y, c["kernel"](), sigmasq=c["sig"], eps=eps[d], nufft_eps=1e-7, estimate_params=False,
                           opts={"cg_tolerance": 1e-4, "mean_cg_warm_start": False})

            def step():
                model._compute_common_parameters(force_recompute=True)
                return model.predict(x, return_variance=False)[0]

            for _ in range(2):
                step()
            torch.cuda.synchronize()
            steps = 10 if N <= 1_000_000 else 5
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            # kernel durations from a separate short pass: the HIP-event timers add host work per launch
            kernel_timing(True)
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            sp_ms, sp_n = kernel_timing_read("spread")
            ip_ms, ip_n = kernel_timing_read("interp")
            cg_ms, cg_n = kernel_timing_read("cg_persistent")
            kernel_timing(False)
            st = model.last_fit_stats
            iters = int(st["mean_cg_iters"])
            sp_us = 1e3 * sp_ms / max(sp_n, 1)
            ip_us = 1e3 * ip_ms / max(ip_n, 1)
            rec = {"d": d, "N": N, "mtot": st["mtot"], "M": st["feature_count"], "ms_per_fit_plus_mean": 1e3 * el / steps,
                   "fits_per_s": steps / el, "cg_iters": iters,
                   "cg_us_per_iter": (1e3 * cg_ms / max(cg_n, 1)) / max(iters, 1) if cg_n else None,
                   "spread_us": sp_us, "interp_us": ip_us,
                   "spread_GBs_fused_bytes": N * (8 * d + 8) / sp_us / 1e3 if sp_n else None,
                   "spread_frac_hbm_survey_bytes": 2 * N * (8 * d + 16) / sp_us / 1e3 / HBM if sp_n else None,
                   "interp_GBs": N * (8 * d + 8) / ip_us / 1e3 if ip_n else None,
                   "interp_frac_hbm_survey_bytes": N * (8 * d + 16) / ip_us / 1e3 / HBM if ip_n else None}
            print(json.dumps(rec), flush=True)
            del model, x, y
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
