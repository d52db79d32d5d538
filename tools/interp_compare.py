#!/usr/bin/env python3
"""Real-output type-2 pass at the bench grid (23 x 23 modes -> N points): two-copy aligned gather vs the single-copy
kernel with its per-plan class order.  usage: interp_compare.py N [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
import torch  # noqa: E402
from efgp_hip import NufftPlan, kernel_timing, kernel_timing_read  # noqa: E402

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
x = (torch.rand(N, 2, generator=g, dtype=torch.float64) * 2 - 1).to(dev)
f = torch.complex(torch.randn(23, 23, generator=g, dtype=torch.float64), torch.randn(23, 23, generator=g, dtype=torch.float64)).to(dev)
res = {}
for name, env in (("pair", {}), ("single+order", {"EFGP_NO_PAIR_GATHER": "1"}), ("single", {"EFGP_NO_PAIR_GATHER": "1", "EFGP_NO_CLASS_ORDER": "1"})):
    for k in ("EFGP_NO_PAIR_GATHER", "EFGP_NO_CLASS_ORDER"):
        os.environ.pop(k, None)
    os.environ.update(env)
    plan = NufftPlan(x, 0.31, 1e-7)
    out = plan.type2(f, (23, 23), real_only=True)
    torch.cuda.synchronize()
    kernel_timing(True)
    for _ in range(reps):
        out = plan.type2(f, (23, 23), real_only=True)
    torch.cuda.synchronize()
    ms, n = kernel_timing_read("interp")
    kernel_timing(False)
    res[name] = out
    print(f"{name:13s} N={N:.0e} interp kernel {1e3 * ms / max(n, 1):8.1f} us ({n} launches)")
ref = res["single"]
for k, v in res.items():
    print(k, "max abs diff vs single:", float((v - ref).abs().max()))
