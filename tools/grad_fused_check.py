#!/usr/bin/env python3
"""A/B of the native gradient tail against the torch sequence (EFGP_NO_FUSED_GRADIENT=1) with injected probes, and of
efgp_toeplitz_apply_scaled against efgp_toeplitz_apply on several grids."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
import torch  # noqa: E402
from efgp_hip import ToeplitzOp  # noqa: E402
from efgpnd import efgpnd_gradient_batched  # noqa: E402
from kernels.squared_exponential import SquaredExponential  # noqa: E402
from kernels.matern import Matern  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(3)
for shape in [(45, 45), (29, 29), (61, 61), (33, 33), (141, 141), (21, 21, 21), (45, 37)]:
    n = [(L + 1) // 2 for L in shape]
    M = 1
    for a in n:
        M *= a
    v = torch.complex(torch.randn(*shape, generator=g, dtype=torch.float64), torch.randn(*shape, generator=g, dtype=torch.float64)).to(dev)
    op = ToeplitzOp(v)
    x = torch.complex(torch.randn(3, M, generator=g, dtype=torch.float64), torch.randn(3, M, generator=g, dtype=torch.float64)).to(dev)
    pre = torch.complex(torch.randn(M, generator=g, dtype=torch.float64), torch.randn(M, generator=g, dtype=torch.float64)).to(dev)
    post = torch.complex(torch.randn(M, generator=g, dtype=torch.float64), torch.randn(M, generator=g, dtype=torch.float64)).to(dev)
    ref = post * op.apply(pre * x)
    got = op.apply_scaled(x, pre=pre, post=post)
    xr = x.real.contiguous()
    ref_r = op.apply(xr.to(torch.complex128))
    got_r = op.apply_scaled(xr)
    e1 = float((got - ref).abs().max() / ref.abs().max())
    e2 = float((got_r - ref_r).abs().max() / ref_r.abs().max())
    print(f"apply_scaled {shape}: rel err {e1:.2e} (complex, pre+post) {e2:.2e} (real, plain)")
    assert e1 < 1e-12 and e2 < 1e-12

for name, kern, d, N in [("SE 2-D", SquaredExponential(dimension=2, init_lengthscale=0.1, init_variance=1.0), 2, 20000),
                         ("Matern 2-D", Matern(dimension=2, nu=1.5, init_lengthscale=0.3, init_variance=1.3), 2, 20000),
                         ("SE 3-D", SquaredExponential(dimension=3, init_lengthscale=0.4, init_variance=0.7), 3, 5000),
                         ("SE 1-D", SquaredExponential(dimension=1, init_lengthscale=0.05, init_variance=0.7), 1, 5000)]:
    x = torch.rand(N, d, generator=g, dtype=torch.float64).to(dev)
    y = (torch.sin(3 * x.sum(1)) + 0.3 * torch.randn(N, generator=g, dtype=torch.float64).to(dev)).contiguous()
    T = 4
    sig = torch.tensor(0.09, dtype=torch.float64)
    outs = {}
    for mode in ("native", "torch"):
        if mode == "torch":
            os.environ["EFGP_NO_FUSED_GRADIENT"] = "1"
        else:
            os.environ.pop("EFGP_NO_FUSED_GRADIENT", None)
        st = {}
        # probes: the grid size is not known here -> first call discovers M
        gr0 = efgpnd_gradient_batched(x, y, sig, kern, 1e-4, T, stats_out=st, probe_seed=11, cg_tol=1e-11)
        M = st["feature_count"]
        pv = (torch.randint(0, 2, (T, M), generator=torch.Generator().manual_seed(5)) * 2 - 1).to(torch.float64).to(dev)
        st = {}
        gr, lm = efgpnd_gradient_batched(x, y, sig, kern, 1e-4, T, stats_out=st, probe_seed=11, cg_tol=1e-11, probes_V=pv, compute_log_marginal=True,
                                         log_marginal_probes=8, log_marginal_steps=10,
                                         log_marginal_probe_vectors=(torch.randint(0, 2, (8, M), generator=torch.Generator().manual_seed(6)) * 2 - 1).to(torch.float64).to(dev))
        outs[mode] = (gr.cpu(), st["term1"], st["term2"], float(lm), int(st["mean_cg_iters"]), int(st["trace_cg_iters"]))
    a, b = outs["native"], outs["torch"]
    print(name, "grad native", a[0].tolist(), "torch", b[0].tolist(), "iters", a[4:], b[4:])
    for q in range(3):
        rel = float((a[q] - b[q]).abs().max() / b[q].abs().max())
        print(f"   {['grad', 'term1', 'term2'][q]} rel diff {rel:.2e}")
        assert rel < 1e-8, (rel, a[q], b[q])
    assert abs(a[3] - b[3]) <= 1e-9 * abs(b[3]), (a[3], b[3])
os.environ.pop("EFGP_NO_FUSED_GRADIENT", None)
print("OK")
