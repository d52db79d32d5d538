#!/usr/bin/env python3
"""Round 3: the 3-D plane-per-wave tile spreader (spread_tile3_planes_kernel) against the lane-per-point one (EFGP_NO_TILE_PLANES=1):
type-1 outputs must be IDENTICAL (integer sums of the same rounded contributions), and the time per pass.
usage: tile_planes_ab.py [N] [mtot] [tol]
NOTE: the plane-per-wave kernel measured slower and was removed again (profiles/r3_tile_planes_negative.txt); with the current library
both legs of this script run the same kernel.  Kept as the record of how the comparison was made."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
import torch
from efgp_hip import NufftPlan

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 5_000_000
mt = int(sys.argv[2]) if len(sys.argv) > 2 else 57
tol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-6
g = torch.Generator(device="cuda").manual_seed(3)
x = torch.rand(N, 3, generator=g, dtype=torch.float64, device="cuda") * 2 - 1
c = torch.randn(2, N, generator=g, dtype=torch.float64, device="cuda")
h = 0.9 * 3.14159 / mt / 1.0
res = {}
for env in ("1", None):
    if env:
        os.environ["EFGP_NO_TILE_PLANES"] = env
    else:
        os.environ.pop("EFGP_NO_TILE_PLANES", None)
    plan = NufftPlan(x, h, tol)
    outs = []
    for what in ("pair", "complex"):
        cc = c if what == "pair" else torch.complex(c[0], c[1])
        o = plan.type1(cc, (mt,) * 3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            o = plan.type1(cc, (mt,) * 3)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        outs.append(o)
        print(f"{'lane per point' if env else 'plane per wave':>15s}  {what:8s} N={N} mtot={mt} tol={tol:g}: {1e3 * dt:.3f} ms per type-1", flush=True)
    res[env] = outs
for a, b, what in zip(res["1"], res[None], ("pair", "complex")):
    print(f"{what}: identical = {bool(torch.equal(a, b))}, max |diff| = {float((a - b).abs().max()):.3e}")
