#!/usr/bin/env python3
"""Round 3: launch-shape sweep of the Hermitian 3-D line iteration (EFGP_CG3_LC / _LS / _LM lines per workgroup, EFGP_CG_NBLK
update workgroups), microseconds per iteration.  usage: cg3_sweep.py mtot"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
import torch
from efgp_hip import ToeplitzOp, cg_solve

mt = int(sys.argv[1]) if len(sys.argv) > 1 else 57
dev = torch.device("cuda", 0)
gm = torch.Generator().manual_seed(0)
L = 2 * mt - 1
vv = torch.complex(torch.randn(L, L, L, generator=gm, dtype=torch.float64), torch.randn(L, L, L, generator=gm, dtype=torch.float64))
vv = ((vv + vv.flip(0, 1, 2).conj()) / 2).to(dev)
wr = torch.rand(mt, mt, mt, generator=gm, dtype=torch.float64)
wsm = ((wr + wr.flip(0, 1, 2)) / 2).reshape(-1).to(torch.complex128).to(dev)
dgm = (wsm.abs() ** 2 + 0.1).real
opm = ToeplitzOp(vv)


def run(rows):
    br = torch.complex(torch.randn(rows, mt, mt, mt, generator=gm, dtype=torch.float64), torch.randn(rows, mt, mt, mt, generator=gm, dtype=torch.float64))
    bm = ((br + br.flip(1, 2, 3).conj()) / 2).reshape(rows, -1).to(dev)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, it, _ = cg_solve(opm, wsm, 0.1, 0, bm if rows > 1 else bm[0], None, 1e-300, max_iter=96, early_stop=False, diag=dgm, batched=rows > 1, hermitian=True)
        torch.cuda.synchronize()
        best = min(best, 1e6 * (time.perf_counter() - t0) / it)
    return best


base = {"EFGP_CG3_LC": "8", "EFGP_CG3_LS": "16", "EFGP_CG3_LM": "16", "EFGP_CG_NBLK": "64"}
for k, vals in (("EFGP_CG_NBLK", (64, 128, 256)), ("EFGP_CG3_LC", (4, 8, 16, 32)), ("EFGP_CG3_LS", (4, 8, 16, 32)), ("EFGP_CG3_LM", (4, 8, 16, 32))):
    for v in vals:
        os.environ.update(base)
        os.environ[k] = str(v)
        print(f"mtot {mt}: {k}={v}: 1 system {run(1):.1f} us/iter, 3 systems {run(3):.1f} us/iter", flush=True)
