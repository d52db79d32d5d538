#!/usr/bin/env python3
"""Round 3 diagnostic: where one iteration of the Hermitian cooperative CG (cg_coop2d_herm_kernel) spends its time: shader-clock
stamps of workgroup 0 (EFGP_COOP_DBG=2; the library prints the table on stderr).  usage: coop_phase_profile.py  (GPU box)"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
import torch  # noqa: E402
from efgp_hip import ToeplitzOp, cg_solve  # noqa: E402

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
    for dbg in ("0", "0", "2"):
        os.environ["EFGP_COOP_DBG"] = dbg
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, it, _ = cg_solve(opm, wsm, 0.1, 0, bm, torch.zeros_like(bm), 1e-300, max_iter=200, early_stop=False, diag=dgm, batched=False, hermitian=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"grid {opm.fft_shape[0]}^2 (mtot {mt}): {1e6 * dt / it:.1f} us per iteration with stamps on", flush=True)
