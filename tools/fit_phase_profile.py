#!/usr/bin/env python3
"""Diagnostic: wall time of the phases of one fit + mean prediction (synchronised after each phase).
usage: fit_phase_profile.py d N"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import efgpnd as E  # noqa: E402
from efgp_hip import NufftPlan, cg_solve  # noqa: E402
from sweep_bench import CASES, synth  # noqa: E402

d, N = int(sys.argv[1]), int(float(sys.argv[2]))
dev = torch.device("cuda", 0)
x, y = synth(N, d, 100 + d, dev)
c = CASES[d]
eps = {1: 1e-4, 2: 1e-4, 3: 1e-3}[d]
model = E.EFGPND(x, y, c["kernel"](), sigmasq=c["sig"], eps=eps, nufft_eps=1e-7, estimate_params=False,
                 opts={"cg_tolerance": 1e-4, "mean_cg_warm_start": False})
for rep in range(3):
    T = {}

    def lap(name, t0):
        torch.cuda.synchronize()
        T[name] = T.get(name, 0.0) + 1e3 * (time.perf_counter() - t0)

    dd = model._device_data()
    t0 = time.perf_counter(); grid = E._Grid(model.kernel, model.eps, dd["L"], d, dev); lap("grid (get_xis, ws)", t0)
    t0 = time.perf_counter(); plan = NufftPlan(dd["x"], grid.h, 1e-7); lap("plan", t0)
    t0 = time.perf_counter(); Fy, v = E._normal_equations(plan, dd["y"], grid, model._shards); lap("type-1 pair", t0)
    t0 = time.perf_counter(); toep = E.ToeplitzND(v, force_pow2=True); lap("toeplitz setup", t0)
    t0 = time.perf_counter()
    rhs = grid.ws * Fy
    diag = E._center_value(v) * grid.ws.abs().pow(2).real + c["sig"]
    beta, iters, _ = cg_solve(toep._op, grid.ws, c["sig"], 0, rhs, torch.zeros_like(rhs), 1e-4, diag=diag, batched=False)
    lap(f"cg ({iters} iterations)", t0)
    t0 = time.perf_counter(); plan2 = NufftPlan(x, grid.h, 1e-7); mean = plan2.type2(beta, (grid.mtot,) * d, real_only=True, mode_scale=grid.ws); lap("type-2 mean", t0)
print(f"d={d} N={N} mtot={grid.mtot}")
for k, vv in T.items():
    print(f"  {k:28s} {vv:8.3f} ms")
