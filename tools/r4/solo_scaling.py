"""How the per-iteration time of the batched cooperative general solve (one system per workgroup, G = 1) depends on the number of
systems in the launch: latency of one workgroup's iteration vs contention.  mtot 41 (96 x 96 grid)."""
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
for mt in (41, 31):
    L = 2 * mt - 1
    xp = (torch.rand(3000, 2, generator=gm, dtype=torch.float64) * 2 - 1).to(dev)
    kk = torch.arange(-(mt - 1), mt, dtype=torch.float64, device=dev)
    E0 = torch.exp(-2j * torch.pi * 0.3 * kk[:, None] * xp[None, :, 0])
    E1 = torch.exp(-2j * torch.pi * 0.3 * kk[:, None] * xp[None, :, 1])
    vv = (E0 @ E1.T).contiguous()
    wr = torch.rand(mt, mt, generator=gm, dtype=torch.float64)
    wsm = ((wr + wr.flip(0, 1)) / 2).reshape(-1).to(torch.complex128).to(dev)
    br = torch.complex(torch.randn(mt, mt, generator=gm, dtype=torch.float64), torch.randn(mt, mt, generator=gm, dtype=torch.float64))
    bm = br.reshape(-1).to(dev)
    opm = ToeplitzOp(vv)
    out = []
    for Bn in (1, 8, 64, 128, 256, 512):
        rb = bm[None, :].repeat(Bn, 1) * torch.linspace(0.5, 1.5, Bn, device=dev, dtype=torch.float64)[:, None]
        if Bn == 1:
            os.environ["EFGP_COOP_G"] = "1"
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            xb, itb, _ = cg_solve(opm, wsm, 0.1, 1, rb, torch.zeros_like(rb), 1e-300, max_iter=100, early_stop=False, batched=True)
            torch.cuda.synchronize()
            dtb = time.perf_counter() - t0
        os.environ.pop("EFGP_COOP_G", None)
        out.append(f"{Bn} systems: {1e6 * dtb / itb:.1f} us/iter")
    print(f"mtot {mt} grid {opm.cg_shape()}: " + " | ".join(out))
