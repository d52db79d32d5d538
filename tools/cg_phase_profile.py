#!/usr/bin/env python3
"""Diagnostic: where one persistent-CG iteration spends its cycles (in-kernel stamps).

Needs the instrumented build:  python gp-quadrature_amd/efgp_hip/build.py --stamps
Run on the GPU box:            python tools/cg_phase_profile.py
Stamp fences change overlap; read the SHARES, not the absolute length."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("EFGP_HIP_LIBRARY", os.path.join(ROOT, "gp-quadrature_amd", "efgp_hip", "libefgp_hip_stamps.so"))
sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
import torch  # noqa: E402
import efgp_hip  # noqa: E402
from efgp_hip import ToeplitzOp, cg_solve, cg_solve_mean_async  # noqa: E402

mtot, d = 23, 2
g = torch.Generator().manual_seed(0)
v = torch.complex(torch.randn(*(4 * ((mtot - 1) // 2) + 1,) * d, generator=g, dtype=torch.float64),
                  torch.randn(*(4 * ((mtot - 1) // 2) + 1,) * d, generator=g, dtype=torch.float64)).cuda()
M = mtot ** d
ws = torch.rand(M, generator=g, dtype=torch.float64).to(torch.complex128).cuda()
b = torch.randn(M, generator=g, dtype=torch.float64).to(torch.complex128).cuda()
diag = (ws.abs() ** 2 + 0.1).real
op = ToeplitzOp(v)
iters = 300
cg_solve(op, ws, 0.1, 0, b, torch.zeros_like(b), 1e-300, max_iter=iters, early_stop=False, diag=diag, batched=False)
out = (C.c_longlong * 16)()
lib = efgp_hip.lib()
lib.efgp_debug_cg_stamps.argtypes = [C.POINTER(C.c_longlong)]
assert lib.efgp_debug_cg_stamps(out) == 0
names = None
if len(sys.argv) > 1 and sys.argv[1] == "herm":
    # the Hermitian mean-system kernel (cg_herm64_kernel): real-function coefficients, symmetric ws and Toeplitz vector
    flip = lambda t: torch.flip(t, dims=(0, 1))
    vh = 0.5 * (v + flip(v).conj())
    wsq = ws.reshape(mtot, mtot)
    wsh = (0.5 * (wsq + flip(wsq))).reshape(-1)
    bq = torch.complex(torch.randn(mtot, mtot, generator=g, dtype=torch.float64), torch.randn(mtot, mtot, generator=g, dtype=torch.float64)).cuda()
    bh = (0.5 * (bq + flip(bq).conj())).reshape(-1)
    oph = ToeplitzOp(vh)
    beta, lazy = cg_solve_mean_async(oph, wsh, 0.1, vh[tuple((s - 1) // 2 for s in vh.shape)].real, bh, 1e-300, max_iter=iters, early_stop=False)
    assert int(lazy) == iters
    assert lib.efgp_debug_cg_stamps(out) == 0
    names = {0: "A: ws*p, row transforms (h+1 lines)", 1: "B/C: packed column transforms, spectrum", 2: "D: unpack, row transforms, A u",
             3: "<p,Ap>: wave sums, LDS, barrier", 4: "alpha, x r z updates, partial sums", 5: "<r,r>, <r,z>: wave sums, LDS, barrier",
             7: "norm test, beta, p update"}
herm_names = names if len(sys.argv) > 1 and sys.argv[1] == "herm" else None
names = herm_names or {0: "load ws*p -> LDS", 1: "fwd pass 0 (last dim)", 2: "fwd pass 1 (+fused mid)", 3: "fwd pass 2",
         4: "inv pass 0 (rest)", 5: "inv pass 1", 6: "inv pass 2", 7: "vector updates + reductions", 8: "crop + A u"}
tot = sum(out[i] for i in range(9))
for i in range(9):
    if out[i]:
        print(f"{names[i]:44s} {out[i] / (iters + 1):10.0f} cycles/iter  {100.0 * out[i] / tot:5.1f}%")
print(f"total {tot / (iters + 1):.0f} cycles per iteration (shader clock ticks)")
