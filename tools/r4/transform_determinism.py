"""Which type-1 passes are bit-reproducible run to run?  (F*y, Toeplitz vector) pair and +-1 probe transforms, with and without the
per-model point layout, N = 20000 and 1e6, d = 2."""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (".", "gp-quadrature_amd"):
    sys.path.insert(0, os.path.join(R, p))
import torch  # noqa: E402
from efgp_hip import NufftPlan, PointSet  # noqa: E402

dev = torch.device("cuda", 0)
for N in (4766, 20000, 1_000_000):
    g = torch.Generator().manual_seed(1)
    x = torch.rand(N, 2, generator=g, dtype=torch.float64).to(dev)
    y = torch.randn(N, generator=g, dtype=torch.float64).to(dev)
    for label, pts in (("no layout", None), ("layout", PointSet(x, values=y))):
        for tol in (6e-8, 1e-5):
            plan = NufftPlan(x, 0.8, tol, points=pts)
            outs = []
            for _ in range(3):
                fy, v = plan.type1_pair(y, (23, 23), (45, 45))
                fz = plan.type1_rademacher(7, 5, (23, 23))
                f1 = plan.type1(y, (23, 23))
                outs.append((fy.clone(), v.clone(), fz.clone(), f1.clone()))
            same = [all(torch.equal(outs[0][q], o[q]) for o in outs[1:]) for q in range(4)]
            print(f"N {N:8d} {label:9s} tol {tol:g}: pair F*y {same[0]}, pair v {same[1]}, rademacher {same[2]}, type1(y) {same[3]}")
