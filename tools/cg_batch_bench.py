#!/usr/bin/env python3
"""Diagnostic: batched CG on mid-size 2-D grids (the posterior-variance / trace-probe solves of BASELINE configs[2], [3]):
time per iteration of a whole batch, cooperative launches against the multi-launch iteration.
usage: cg_batch_bench.py mtot nbatch [iters]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
import torch  # noqa: E402
from efgp_hip import ToeplitzOp, cg_solve  # noqa: E402

mtot, nb = int(sys.argv[1]), int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 100
g = torch.Generator().manual_seed(0)
L = 2 * mtot - 1
v = torch.complex(torch.randn(L, L, generator=g, dtype=torch.float64), torch.randn(L, L, generator=g, dtype=torch.float64))
v = ((v + v.flip(0, 1).conj()) / 2).cuda()
M = mtot * mtot
ws = torch.rand(M, generator=g, dtype=torch.float64).to(torch.complex128).cuda()
b = (torch.randint(0, 2, (nb, M), generator=g) * 2 - 1).to(torch.complex128).cuda()
op = ToeplitzOp(v)
res = {}
for mode in ("coop", "multi-launch"):
    if mode == "multi-launch":
        os.environ["EFGP_NO_CG_COOP"] = "1"
    else:
        os.environ.pop("EFGP_NO_CG_COOP", None)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x, it, rows = cg_solve(op, ws, 0.1, 1, b, torch.zeros_like(b), 1e-300, max_iter=iters, early_stop=False, batched=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    res[mode] = (x, dt)
    print(f"mtot={mtot} fft={op.fft_shape} systems={nb} {mode}: {1e3 * dt:.2f} ms for {iters} iterations = {1e6 * dt / iters:.1f} us per iteration of the batch", flush=True)
print("rel diff", float(torch.linalg.norm(res["coop"][0] - res["multi-launch"][0]) / torch.linalg.norm(res["multi-launch"][0])))
