#!/usr/bin/env python3
"""Round 3: microseconds per iteration of the 3-D line iteration, Hermitian (planes k0 >= 0) against general, forced iteration counts.
usage: cg3_iter.py [mtot ...]   (default 19 33 57)"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
import torch
from efgp_hip import ToeplitzOp, cg_solve

dev = torch.device("cuda", 0)
gm = torch.Generator().manual_seed(0)
for mt in [int(a) for a in sys.argv[1:]] or [19, 33, 57]:
    L = 2 * mt - 1
    vv = torch.complex(torch.randn(L, L, L, generator=gm, dtype=torch.float64), torch.randn(L, L, L, generator=gm, dtype=torch.float64))
    vv = ((vv + vv.flip(0, 1, 2).conj()) / 2).to(dev)
    wr = torch.rand(mt, mt, mt, generator=gm, dtype=torch.float64)
    wsm = ((wr + wr.flip(0, 1, 2)) / 2).reshape(-1).to(torch.complex128).to(dev)
    for rows in (1, 3):
        br = torch.complex(torch.randn(rows, mt, mt, mt, generator=gm, dtype=torch.float64), torch.randn(rows, mt, mt, mt, generator=gm, dtype=torch.float64))
        bm = ((br + br.flip(1, 2, 3).conj()) / 2).reshape(rows, -1).to(dev)
        dgm = (wsm.abs() ** 2 + 0.1).real
        opm = ToeplitzOp(vv)
        out = []
        for herm in (True, False):
            for _ in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                _, it, _ = cg_solve(opm, wsm, 0.1, 0, bm if rows > 1 else bm[0], None, 1e-300, max_iter=96, early_stop=False, diag=dgm, batched=rows > 1, hermitian=herm)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            out.append(1e6 * dt / it)
        print(f"grid {opm.fft_shape[0]}^3 (mtot {mt}), {rows} system(s): hermitian {out[0]:.1f} us/iter, general {out[1]:.1f} us/iter", flush=True)
