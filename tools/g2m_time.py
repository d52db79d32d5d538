#!/usr/bin/env python3
"""Launch times of the type-1 back end (accumulator -> modes) with the one-launch kernel and with the rocFFT sequence:
the fit's pair pass (96 x 96 grid, boxes 23 / 45) and the probe pass (48 x 48, 3 grids).  Needs rocprofv3 around it, or
prints event-timed whole transforms."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from efgp_hip import NufftPlan, PointSet  # noqa: E402

dev = torch.device("cuda", 0)
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
x, y = bench.synth(N, 2, 1000, dev)
pts = PointSet(x, values=y)
for label, env in (("one launch", None), ("rocFFT sequence", "1")):
    if env:
        os.environ["EFGP_NO_GRID_TO_MODES"] = env
    else:
        os.environ.pop("EFGP_NO_GRID_TO_MODES", None)
    plan = NufftPlan(x, 0.3459, 6e-8, points=pts)
    planp = NufftPlan(x, 0.3459, 1e-5, points=pts)
    for name, fn in (("pair (F*y, v)", lambda: plan.type1_pair(y, (23, 23), (45, 45))), ("5 probes", lambda: planp.type1_rademacher(7, 5, (23, 23)))):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"{label:16s} {name:14s} {1e3 * e0.elapsed_time(e1) / 50:7.1f} us per transform (device, pipelined)")
