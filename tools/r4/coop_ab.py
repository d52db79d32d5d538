"""A/B of two builds of the library on the cooperative single-system solves (Hermitian and general kernels, mtot 41 / 71 / 131,
400 forced iterations, 5 repeats, median): usage coop_ab.py  (run once per library with EFGP_HIP_LIBRARY set; symbols a build
lacks are skipped when binding)."""
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "gp-quadrature_amd"))
import ctypes as C  # noqa: E402
import efgp_hip  # noqa: E402,F401
L = sys.modules["efgp_hip.lib"]

path = L.library_path()
h = C.CDLL(path) if os.path.exists(path) else None
for name in list(L._SIGNATURES):
    if h is not None and not hasattr(h, name):
        del L._SIGNATURES[name]
import torch  # noqa: E402
from efgp_hip import ToeplitzOp, cg_solve  # noqa: E402

dev = torch.device("cuda", 0)
gm = torch.Generator().manual_seed(0)
out = []
for mt in (41, 71, 131):
    Lg = 2 * mt - 1
    vv = torch.complex(torch.randn(Lg, Lg, generator=gm, dtype=torch.float64), torch.randn(Lg, Lg, generator=gm, dtype=torch.float64))
    vv = ((vv + vv.flip(0, 1).conj()) / 2).to(dev)
    wr = torch.rand(mt, mt, generator=gm, dtype=torch.float64)
    wsm = ((wr + wr.flip(0, 1)) / 2).reshape(-1).to(torch.complex128).to(dev)
    br = torch.complex(torch.randn(mt, mt, generator=gm, dtype=torch.float64), torch.randn(mt, mt, generator=gm, dtype=torch.float64))
    bm = ((br + br.flip(0, 1).conj()) / 2).reshape(-1).to(dev)
    dgm = (wsm.abs() ** 2 + 0.1).real
    opm = ToeplitzOp(vv)
    for herm in (True, False):
        ts = []
        for _ in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, itm, _ = cg_solve(opm, wsm, 0.1, 0, bm, torch.zeros_like(bm), 1e-300, max_iter=400, early_stop=False, diag=dgm,
                                 batched=False, hermitian=herm)
            torch.cuda.synchronize()
            ts.append(1e6 * (time.perf_counter() - t0) / itm)
        out.append(f"mtot {mt} {'herm' if herm else 'general'} {sorted(ts[1:])[2]:.2f}")
print(os.path.basename(path), " | ".join(out))
