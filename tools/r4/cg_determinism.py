"""Is a fused CG solve bit-reproducible run to run?  The 'hard' usa_temp system of tests/test_gpu_realdata_csv.py (mtot 49, 128 x 128
grid), no preconditioner (600-700 iterations: the most sensitive), five solves per path: iteration count and a checksum of x."""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("tests", "gp-quadrature_amd"):
    sys.path.insert(0, os.path.join(R, p))
import torch  # noqa: E402
import _realdata as RD  # noqa: E402
from efgp_hip import cg_solve  # noqa: E402
from efgpnd import NUFFT, ToeplitzND, compute_convolution_vector_vectorized_dD, _cmplx  # noqa: E402
from utils.kernels import get_xis  # noqa: E402

regime = "hard"
x, y = RD.usa_temp()
x, y = x.cuda(), y.cuda()
k = RD.kernel(regime)
sig = RD.REGIMES[regime][2]
dtype, cdtype, d = x.dtype, _cmplx(x.dtype), 2
xis_1d, h, mtot = get_xis(k, eps=RD.EPS, L=RD.domain_length(x), use_integral=True, l2scaled=False)
xis_1d = xis_1d.to(device=x.device, dtype=dtype)
xis = torch.stack(torch.meshgrid(*(xis_1d for _ in range(d)), indexing="ij"), dim=-1).reshape(-1, d)
ws = torch.sqrt(k.spectral_density(xis).to(dtype=cdtype) * h ** d)
nufft = NUFFT(x, torch.zeros(d, dtype=dtype, device=x.device), h, RD.NUFFT_EPS, cdtype=cdtype)
rhs_runs = [ws * nufft.type1(y, out_shape=(mtot,) * d).reshape(-1) for _ in range(3)]
print("rhs bit-identical over 3 transforms:", all(torch.equal(rhs_runs[0], r) for r in rhs_runs[1:]))
v_runs = [compute_convolution_vector_vectorized_dD((mtot - 1) // 2, x, h).to(dtype=cdtype) for _ in range(3)]
print("Toeplitz vector bit-identical:", all(torch.equal(v_runs[0], r) for r in v_runs[1:]))
rhs = rhs_runs[0]
top = ToeplitzND(v_runs[0], force_pow2=True)
for label, env, herm in (("cooperative general", {}, False), ("cooperative Hermitian", {}, True),
                         ("multi-launch", {"EFGP_NO_CG_COOP": "1"}, False)):
    for kk, vv in env.items():
        os.environ[kk] = vv
    out = []
    for _ in range(5):
        xs, it, _ = cg_solve(top._op, ws, sig, 0, rhs, torch.zeros_like(rhs), RD.CG_TOL, early_stop=True, batched=False, hermitian=herm)
        out.append((int(it), float(xs.abs().sum())))
    for kk in env:
        os.environ.pop(kk)
    print(f"{label:24s}", out)
