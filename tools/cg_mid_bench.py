#!/usr/bin/env python3
"""Diagnostic: multi-kernel CG iteration time for grids beyond the single-launch kernel.
usage: cg_mid_bench.py d mtot [iters]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
import torch  # noqa: E402
from efgp_hip import ToeplitzOp, cg_solve  # noqa: E402

d, mtot = int(sys.argv[1]), int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 160
g = torch.Generator().manual_seed(0)
L = 2 * mtot - 1
v = torch.complex(torch.randn(*(L,) * d, generator=g, dtype=torch.float64), torch.randn(*(L,) * d, generator=g, dtype=torch.float64)).cuda()
M = mtot ** d
ws = torch.rand(M, generator=g, dtype=torch.float64).to(torch.complex128).cuda()
b = torch.randn(M, generator=g, dtype=torch.float64).to(torch.complex128).cuda()
diag = (ws.abs() ** 2 + 0.1).real
op = ToeplitzOp(v)
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, it, _ = cg_solve(op, ws, 0.1, 0, b, torch.zeros_like(b), 1e-300, max_iter=iters, early_stop=False, diag=diag, batched=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"d={d} mtot={mtot} fft={op.fft_shape}: {it} iterations, {1e6 * dt / it:.1f} us/iteration (graph {'off' if os.environ.get('EFGP_NO_CG_GRAPH') else 'on'})")
