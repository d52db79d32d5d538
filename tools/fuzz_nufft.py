#!/usr/bin/env python3
"""Randomised check of the HIP transforms against the exact sums (oracle) over dimensions, mode boxes (odd and even),
tolerances, point counts (1 .. 3e5), coordinate ranges and strength types.  usage: fuzz_nufft.py [cases] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from efgp_hip import NufftPlan  # noqa: E402
from oracle import efgp_oracle as O  # noqa: E402



def run(cases, seed, verbose=True):
    """Returns (worst error / tol over the cases, list of failing case descriptions: error > 2 tol + 1e-12)."""
    g = torch.Generator().manual_seed(seed)

    def ri(lo, hi):
        return int(torch.randint(lo, hi + 1, (1,), generator=g))

    def rel(a, b):
        a, b = a.detach().cpu(), b.detach().cpu()
        return float(torch.linalg.norm((a - b).reshape(-1)) / max(float(torch.linalg.norm(b.reshape(-1))), 1e-300))

    worst = 0.0
    failures = []
    t0 = time.time()
    for case in range(cases):
        d = ri(1, 3)
        nm = [ri(3, 90), ri(3, 70), ri(3, 22)][d - 1]
        tol = 10.0 ** (-ri(3, 11))
        N = [1, 2, 37, 1000, 40_000, 300_000][ri(0, 5)]
        if d == 3 and N > 40_000:
            N = 40_000
        scale = 10.0 ** ri(-2, 2)
        shift = float(torch.randn(1, generator=g)) * scale * 3
        x = (torch.rand(N, d, generator=g, dtype=torch.float64) - 0.5) * scale + shift
        h = 0.45 / scale / max(1.0, abs(shift) / scale) if ri(0, 1) else 0.3 / scale
        cplx = bool(ri(0, 1))
        c = torch.randn(N, generator=g, dtype=torch.float64)
        if cplx:
            c = torch.complex(c, torch.randn(N, generator=g, dtype=torch.float64))
        shape = (nm,) * d
        plan = NufftPlan(x.cuda(), h, tol)
        out = plan.type1(c.cuda() if cplx else c.cuda(), shape)
        sub = slice(0, min(N, 3000))
        # exact type-1 needs all points: bound the work by thinning the POINT set for the reference only when N is large
        if N <= 40_000:
            ref1 = O.nudft_type1(x, h, c.to(torch.complex128), shape)
            e1 = rel(out, ref1)
        else:
            e1 = 0.0
        f = torch.complex(torch.randn(*shape, generator=g, dtype=torch.float64), torch.randn(*shape, generator=g, dtype=torch.float64))
        ro = bool(ri(0, 1))
        o2 = plan.type2(f.cuda(), shape, real_only=ro)
        ref2 = O.nudft_type2(x[sub], h, f, shape)
        # real-only outputs are judged on the scale of the complex sums they are the real part of (for a handful of
        # points the real part alone can be arbitrarily small against the transform's absolute error tol * |sum|)
        # and on at least the typical magnitude |f|_2 of one output (a single point's sum can come out small by chance)
        scale2 = max(float(torch.linalg.norm(ref2)), float(torch.linalg.norm(f)) * ref2.numel() ** 0.5, 1e-300)
        e2 = float(torch.linalg.norm(o2[sub].cpu() - (ref2.real if ro else ref2))) / scale2
        # adjointness at any size
        lhs = torch.vdot(plan.type1(c.cuda(), shape).reshape(-1), f.cuda().reshape(-1))
        Ff = plan.type2(f.cuda(), shape)
        rhs = torch.vdot(c.to(torch.complex128).cuda(), Ff)
        ea = abs(complex(lhs - rhs)) / max(float(torch.linalg.norm(c) * torch.linalg.norm(Ff.cpu())), 1e-300)
        bad = max(e1, e2, ea) > 2 * tol + 1e-12
        worst = max(worst, max(e1, e2, ea) / tol)
        desc = (f"case {case:3d} d={d} nm={nm} tol={tol:.0e} N={N} scale={scale:g} shift={shift:.3g} h={h:.3g} cplx={cplx} real_only={ro}: "
                f"type1 {e1:.2e} type2 {e2:.2e} adjoint {ea:.2e}")
        if bad:
            failures.append(desc)
        if verbose and (bad or case % 10 == 0 or globals().get("_ALL")):
            print(f"case {case:3d} d={d} nm={nm} tol={tol:.0e} N={N} scale={scale:g} shift={shift:.3g} h={h:.3g} cplx={cplx} real_only={ro}: "
                  f"type1 {e1:.2e} type2 {e2:.2e} adjoint {ea:.2e}{'   <-- FAIL' if bad else ''}", flush=True)
    if verbose:
        print(f"{cases} cases in {time.time() - t0:.1f} s; worst error / tol = {worst:.2f}")
    return worst, failures


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
