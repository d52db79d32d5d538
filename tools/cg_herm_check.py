#!/usr/bin/env python3
"""Diagnostic: the Hermitian 64 x 64 mean solve (cg_herm64_kernel) against the general complex kernel on the same
system: iteration counts, solution difference, time per iteration.  Run on the GPU box."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from efgp_hip import ToeplitzOp, cg_solve_mean_async  # noqa: E402
from oracle import efgp_oracle as O  # noqa: E402


def herm(t):
    return 0.5 * (t + torch.flip(t, dims=tuple(range(t.dim()))).conj())


def case(mtot, tol, seed=0, iters=None):
    g = torch.Generator().manual_seed(seed)
    N = 900
    x = torch.rand(N, 2, generator=g, dtype=torch.float64) * 2 - 1
    v = O.conv_vector(x, 0.4, (mtot - 1) // 2)
    w = torch.exp(-2.5 * torch.rand(mtot, mtot, generator=g, dtype=torch.float64))
    ws = (0.5 * (w + torch.flip(w, dims=(0, 1)))).reshape(-1).to(torch.complex128)
    fy = herm(torch.complex(torch.randn(mtot, mtot, generator=g, dtype=torch.float64),
                            torch.randn(mtot, mtot, generator=g, dtype=torch.float64))).reshape(-1)
    vd = v.cuda()
    centre = vd[tuple((s - 1) // 2 for s in vd.shape)].real
    op = ToeplitzOp(vd)
    out = {}
    for mode in ("herm", "plain"):
        if mode == "plain":
            os.environ["EFGP_NO_CG_HERM"] = "1"
        else:
            os.environ.pop("EFGP_NO_CG_HERM", None)
        kw = dict(max_iter=iters, early_stop=False) if iters else {}
        beta, lazy = cg_solve_mean_async(op, ws.cuda(), 0.25, centre, fy.cuda(), tol, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            beta, lazy = cg_solve_mean_async(op, ws.cuda(), 0.25, centre, fy.cuda(), tol, **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out[mode] = (beta.cpu(), int(lazy), dt)
    os.environ.pop("EFGP_NO_CG_HERM", None)
    bh, ih, th = out["herm"]
    bp, ip, tp = out["plain"]
    rel = float(torch.linalg.norm(bh - bp) / torch.linalg.norm(bp))
    print(f"mtot {mtot:3d} tol {tol:g}: iters herm {ih} plain {ip}  rel diff {rel:.2e}  "
          f"us/iter herm {1e6 * th / max(ih, 1):.2f} plain {1e6 * tp / max(ip, 1):.2f}  (launch herm {1e6 * th:.0f} us, plain {1e6 * tp:.0f} us)",
          flush=True)


for mtot in (23, 15, 31, 3, 5, 9, 17, 29):
    case(mtot, 1e-8)
case(23, 1e-300, iters=300)
case(23, 1e-4, seed=3)
