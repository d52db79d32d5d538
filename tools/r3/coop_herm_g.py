#!/usr/bin/env python3
"""Round 3: microseconds per iteration of the Hermitian and the general cooperative CG on 128^2 / 256^2 / 512^2 grids for
different numbers of workgroups per system (EFGP_COOP_G).  usage: coop_herm_g.py"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
import torch
from efgp_hip import ToeplitzOp, cg_solve

dev = torch.device("cuda", 0)
gm = torch.Generator().manual_seed(0)
for mt in (41, 71, 131):
    L = 2 * mt - 1
    vv = torch.complex(torch.randn(L, L, generator=gm, dtype=torch.float64), torch.randn(L, L, generator=gm, dtype=torch.float64))
    vv = ((vv + vv.flip(0, 1).conj()) / 2).to(dev)
    wr = torch.rand(mt, mt, generator=gm, dtype=torch.float64)
    wsm = ((wr + wr.flip(0, 1)) / 2).reshape(-1).to(torch.complex128).to(dev)
    br = torch.complex(torch.randn(mt, mt, generator=gm, dtype=torch.float64), torch.randn(mt, mt, generator=gm, dtype=torch.float64))
    bm = ((br + br.flip(0, 1).conj()) / 2).reshape(-1).to(dev)
    dgm = (wsm.abs() ** 2 + 0.1).real
    opm = ToeplitzOp(vv)
    for G in (64, 32, 16, 8, 4, 2):
        os.environ["EFGP_COOP_G"] = str(G)
        out = []
        for herm in (True, False):
            for _ in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                _, it, _ = cg_solve(opm, wsm, 0.1, 0, bm, torch.zeros_like(bm), 1e-300, max_iter=160, early_stop=False, diag=dgm, batched=False, hermitian=herm)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            out.append(1e6 * dt / it)
        print(f"grid {opm.fft_shape[0]}^2 (mtot {mt}) G<={G}: hermitian {out[0]:.1f} us/iter, general {out[1]:.1f} us/iter", flush=True)
