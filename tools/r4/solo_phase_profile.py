"""Phase shares of one iteration of the batched general cooperative solve (one system per workgroup) at mtot 41 (96 x 96 grid):
EFGP_COOP_DBG=2 stamps of workgroup 0, system 0, 256 systems in the launch."""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (".", "gp-quadrature_amd"):
    sys.path.insert(0, os.path.join(R, p))
os.environ["EFGP_COOP_DBG"] = "2"
import torch  # noqa: E402
from efgp_hip import ToeplitzOp, cg_solve  # noqa: E402

dev = torch.device("cuda", 0)
gm = torch.Generator().manual_seed(0)
mt = int(sys.argv[1]) if len(sys.argv) > 1 else 41
xp = (torch.rand(3000, 2, generator=gm, dtype=torch.float64) * 2 - 1).to(dev)
kk = torch.arange(-(mt - 1), mt, dtype=torch.float64, device=dev)
E0 = torch.exp(-2j * torch.pi * 0.3 * kk[:, None] * xp[None, :, 0])
E1 = torch.exp(-2j * torch.pi * 0.3 * kk[:, None] * xp[None, :, 1])
vv = (E0 @ E1.T).contiguous()
wr = torch.rand(mt, mt, generator=gm, dtype=torch.float64)
wsm = ((wr + wr.flip(0, 1)) / 2).reshape(-1).to(torch.complex128).to(dev)
opm = ToeplitzOp(vv)
Bn = 256
rb = torch.complex(torch.randn(Bn, mt * mt, generator=gm, dtype=torch.float64), torch.randn(Bn, mt * mt, generator=gm, dtype=torch.float64)).to(dev)
cg_solve(opm, wsm, 0.1, 1, rb, None, 1e-300, max_iter=100, early_stop=False, batched=True)
torch.cuda.synchronize()
