"""Cooperative 2-D solves on the smallest in-wave grid (96 / 192 / 384) against the reference's power-of-two grid (128 / 256 / 512,
EFGP_NO_COOP_SMALL=1): microseconds per iteration of one system (Hermitian and general kernel), of a batch of 200 general systems,
and the agreement of the solutions."""
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (".", "gp-quadrature_amd"):
    sys.path.insert(0, os.path.join(R, p))
import torch  # noqa: E402
from efgp_hip import ToeplitzOp, cg_solve  # noqa: E402

dev = torch.device("cuda", 0)
gm = torch.Generator().manual_seed(0)
for mt in (41, 47, 67, 71, 95, 131):
    L = 2 * mt - 1
    # Toeplitz vector of 3000 random points (positive semi-definite, as a model's): v[k] = sum_n exp(-2 pi i h k . x_n)
    xp = (torch.rand(3000, 2, generator=gm, dtype=torch.float64) * 2 - 1).to(dev)
    kk = torch.arange(-(mt - 1), mt, dtype=torch.float64, device=dev)
    E0 = torch.exp(-2j * torch.pi * 0.3 * kk[:, None] * xp[None, :, 0])
    E1 = torch.exp(-2j * torch.pi * 0.3 * kk[:, None] * xp[None, :, 1])
    vv = (E0 @ E1.T).contiguous()
    wr = torch.rand(mt, mt, generator=gm, dtype=torch.float64)
    wsm = ((wr + wr.flip(0, 1)) / 2).reshape(-1).to(torch.complex128).to(dev)
    br = torch.complex(torch.randn(mt, mt, generator=gm, dtype=torch.float64), torch.randn(mt, mt, generator=gm, dtype=torch.float64))
    bm = ((br + br.flip(0, 1).conj()) / 2).reshape(-1).to(dev)
    dgm = (3000.0 * wsm.abs() ** 2 + 0.1).real
    res = {}
    for small in (True, False):
        if small:
            os.environ.pop("EFGP_NO_COOP_SMALL", None)
        else:
            os.environ["EFGP_NO_COOP_SMALL"] = "1"
        opm = ToeplitzOp(vv)
        for herm in (True, False):
            for _ in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                x, it, _ = cg_solve(opm, wsm, 0.1, 0, bm, torch.zeros_like(bm), 1e-300, max_iter=160, early_stop=False, diag=dgm, batched=False,
                                    hermitian=herm)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            xs, its, _ = cg_solve(opm, wsm, 0.1, 0, bm, torch.zeros_like(bm), 1e-9, diag=dgm, batched=False, hermitian=herm)
            res[(small, herm)] = (1e6 * dt / it, xs, its)
        if mt <= 71:
            Bn = 200
            rb = bm[None, :].repeat(Bn, 1) * torch.linspace(0.5, 1.5, Bn, device=dev, dtype=torch.float64)[:, None]
            for _ in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                xb, itb, _ = cg_solve(opm, wsm, 0.1, 1, rb, torch.zeros_like(rb), 1e-300, max_iter=60, early_stop=False, batched=True)
                torch.cuda.synchronize()
                dtb = time.perf_counter() - t0
            res[(small, "batch")] = 1e6 * dtb / itb
        del opm
    line = f"mtot {mt:3d} (2n-1 = {L}):"
    for herm, name in ((True, "herm"), (False, "general")):
        a, b = res[(True, herm)], res[(False, herm)]
        dev_ = float(torch.linalg.norm(a[1] - b[1]) / torch.linalg.norm(b[1]))
        line += f"  {name} {a[0]:.2f} vs {b[0]:.2f} us/iter (iters to 1e-9: {a[2]} vs {b[2]}, solutions differ {dev_:.1e})"
    if (True, "batch") in res:
        line += f"  200 systems {res[(True, 'batch')]:.1f} vs {res[(False, 'batch')]:.1f} us per batch iteration"
    print(line, flush=True)
