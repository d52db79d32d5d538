#!/usr/bin/env python3
"""Diagnostic (round 3): where does the 1e-6 deviation of beta on c2 at cg_tol 1e-12 come from?  F*y, v and beta against the
golden vectors at several NUFFT tolerances."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from _golden import load_case, rel
from test_gpu_efgp_golden import make_model

name = "c2_se2d_n100000"
g, x, y = load_case(name)
for ne in (1e-9, 1e-12, 1e-13, 1e-14):
    m = make_model(name, g, x.cuda(), y.cuda(), 1e-12, nufft_eps=ne)
    m.fit()
    st = m._fit_state
    print(f"nufft_eps={ne:g}: rel Fy={rel(st['Fy'], g['Fy']):.2e} v={rel(st['v'], g['v']):.2e} beta={rel(st['beta'], g['beta']):.2e} "
          f"iters={int(m.last_fit_stats['mean_cg_iters'])} (golden {int(g['iters_1e12'])})", flush=True)
# the reference's beta through OUR operator: residual of the golden beta vs ours
from efgp_hip import cg_solve
m = make_model(name, g, x.cuda(), y.cuda(), 1e-12, nufft_eps=1e-14)
m.fit()
st = m._fit_state
ws, Fy, v = st["ws"], st["Fy"], st["v"]
T = m._toeplitz
def A(b):
    return ws * T(ws * b) + st["sig"] * b
rhs = ws * Fy
bg = torch.from_numpy(g["beta"]).cuda()
for nm, b in (("ours", st["beta"]), ("golden", bg)):
    r = rhs - A(b)
    print(f"{nm}: |rhs - A beta| / |rhs| = {float(torch.linalg.norm(r) / torch.linalg.norm(rhs)):.3e}")
# exact solve of the M x M system in float64 (dense) for reference
M = ws.numel()
I = torch.eye(M, dtype=torch.complex128, device="cuda")
Ad = torch.stack([A(I[i]) for i in range(M)], dim=1)
bd = torch.linalg.solve(Ad, rhs)
print(f"dense solve: rel to ours {rel(st['beta'], bd):.2e}, rel to golden {rel(bg, bd):.2e}; cond(A) = {float(torch.linalg.cond(Ad)):.3e}")
