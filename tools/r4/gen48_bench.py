"""General (non-Hermitian) persistent solves of blocks <= 23 x 23: cg_gen48_kernel (48 x 48 grid) against cg_persistent_2d64_kernel
(EFGP_NO_CG48G=1): microseconds per iteration, 400 forced iterations, batches of 1 / 10 / 64 / 256 systems."""
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
for mt in (23, 17, 13):
    L = 2 * mt - 1
    xp = (torch.rand(3000, 2, generator=gm, dtype=torch.float64) * 2 - 1).to(dev)
    kk = torch.arange(-(mt - 1), mt, dtype=torch.float64, device=dev)
    E0 = torch.exp(-2j * torch.pi * 0.3 * kk[:, None] * xp[None, :, 0])
    E1 = torch.exp(-2j * torch.pi * 0.3 * kk[:, None] * xp[None, :, 1])
    vv = (E0 @ E1.T).contiguous()
    wr = torch.rand(mt, mt, generator=gm, dtype=torch.float64)
    wsm = ((wr + wr.flip(0, 1)) / 2).reshape(-1).to(torch.complex128).to(dev)
    opm = ToeplitzOp(vv)
    dgm = (3000.0 * wsm.abs() ** 2 + 0.1).real
    out = []
    for Bn in (1, 10, 64, 256):
        rb = torch.complex(torch.randn(Bn, mt * mt, generator=gm, dtype=torch.float64), torch.randn(Bn, mt * mt, generator=gm, dtype=torch.float64)).to(dev)
        res = []
        for env in (None, "1"):
            if env:
                os.environ["EFGP_NO_CG48G"] = env
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                xb, itb, _ = cg_solve(opm, wsm, 0.1, 0, rb, None, 1e-300, max_iter=400, early_stop=False, diag=dgm, batched=True)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            os.environ.pop("EFGP_NO_CG48G", None)
            res.append(1e6 * dt / itb)
        out.append(f"{Bn} systems: {res[0]:.2f} vs {res[1]:.2f}")
    print(f"mtot {mt}: us/iter on 48 x 48 vs 64 x 64 | " + " | ".join(out))
