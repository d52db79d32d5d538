"""GPU parity of the native tail of the hyper-parameter gradient (csrc/gradient_ops.hip, efgp_toeplitz_apply_scaled).

Reference: efgpnd_gradient_batched, efgpnd.py:17-317.  The package's default (adjoint) estimator runs steps 4-8 through
efgp_gradient_prepare / efgp_toeplitz_apply_scaled / efgp_gradient_assemble; the torch sequence it replaces is still in
the package (literal and pointwise modes, EFGP_NO_FUSED_GRADIENT=1) and the oracle restates the reference's algebra.
Checked here: each entry against a float64 torch restatement on the same inputs, the whole gradient against the torch
sequence and (through the existing golden tests) against the reference's fixtures, and the argument checks.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

_CD = torch.complex128


def _crandn(shape, g):
    return torch.complex(torch.randn(shape, generator=g, dtype=torch.float64), torch.randn(shape, generator=g, dtype=torch.float64))


# (45,45): the 64 x 64 single-launch kernel; (29,29), (5,5): its 64 x 64 embedding of small blocks; (63,63): n = 32, the largest
# block of that kernel; (45,37): unequal block sizes, (141,141) / 3-D / 1-D: the pad | FFT | crop sequence with folded diagonals
@pytest.mark.parametrize("shape", [(45, 45), (29, 29), (5, 5), (63, 63), (45, 37), (141, 141), (21, 21, 21), (257,)])
def test_apply_scaled_matches_plain_apply(shape):
    from efgp_hip import ToeplitzOp
    g = torch.Generator().manual_seed(len(shape) * 1000 + shape[0])
    v = _crandn(shape, g).cuda()
    op = ToeplitzOp(v)
    M = op.size
    x = _crandn((3, M), g).cuda()
    pre, post = _crandn((M,), g).cuda(), _crandn((M,), g).cuda()
    for p, q in [(pre, post), (pre, None), (None, post), (None, None)]:
        ref = op.apply(x if p is None else p * x)
        if q is not None:
            ref = q * ref
        got = op.apply_scaled(x, pre=p, post=q)
        assert float((got - ref).abs().max() / ref.abs().max()) < 1e-13
    xr = x.real.contiguous()                                   # real input (the +-1 probes): no complex copy is made
    ref = post * op.apply(pre * xr.to(_CD))
    got = op.apply_scaled(xr, pre=pre, post=post)
    assert float((got - ref).abs().max() / ref.abs().max()) < 1e-13
    # written into a slice of a larger buffer (the gradient's B_all), single vector
    buf = torch.zeros((5, M), dtype=_CD, device="cuda")
    op.apply_scaled(x[:2], pre=pre, out=buf[2:4])
    assert float((buf[2:4] - op.apply(pre * x[:2])).abs().max()) < 1e-13 * float(buf.abs().max())
    assert float(buf[:2].abs().max()) == 0.0 and float(buf[4].abs().max()) == 0.0
    one = op.apply_scaled(x[0], post=post)
    assert one.shape == x[0].shape and float((one - post * op.apply(x[0])).abs().max() / one.abs().max()) < 1e-13


def test_apply_scaled_refuses_aliasing():
    from efgp_hip import ToeplitzOp
    g = torch.Generator().manual_seed(1)
    op = ToeplitzOp(_crandn((45, 45), g).cuda())
    x = _crandn((1, op.size), g).cuda()
    with pytest.raises(ValueError):
        op.apply_scaled(x, out=x)


def test_prepare_matches_torch():
    from efgp_hip import gradient_prepare
    g = torch.Generator().manual_seed(2)
    M = 23 * 23
    ws = torch.rand(M, generator=g, dtype=torch.float64).to(_CD).cuda()
    fy = _crandn((M,), g).cuda()
    v = _crandn((45, 45), g).cuda()
    vc = v.reshape(-1)[22 * 45 + 22: 22 * 45 + 23]
    diag, rhs = gradient_prepare(ws, fy, vc, 0.37)
    ref_d = v[22, 22].real * ws.abs().pow(2) + 0.37          # efgpnd.py:128-133
    assert float((diag - ref_d).abs().max() / ref_d.abs().max()) < 1e-15
    assert torch.equal(rhs, ws * fy)                          # efgpnd.py:141 (ws is real: the products round identically)
    d2, r2 = gradient_prepare(ws, fy, vc, 0.37, want_diag=False)
    assert d2 is None and torch.equal(r2, rhs)


def _assemble_reference(fy, tg, ws, beta, dp, fz, v, beta_all, variance_idx, trace_idx, sig, n_obs, yy, variance):
    """The adjoint estimator's algebra in plain torch (the sequence of this package's literal code path)."""
    H = dp.shape[1]
    T, M = v.shape
    K = len(trace_idx)
    g = ws * beta
    fa = (fy - tg) / sig
    term2 = torch.zeros(H + 1, dtype=torch.float64)
    term1 = torch.zeros(H + 1, dtype=torch.float64)
    for i in range(H):
        term2[i] = (fa.conj() * (dp[:, i] * fa)).sum().real
    y_z = (fy.conj() * g).sum().real
    z_z = (g.conj() * tg).sum().real
    a_norm = (yy - 2.0 * y_z + z_z) / (sig * sig)
    y_alpha = (yy - y_z) / sig
    term2[H] = a_norm
    bk, bn = beta_all[:K * T], beta_all[K * T:]
    for s, ki in enumerate(trace_idx):
        diff = dp[:, ki] * fz - ws * bk[s * T:(s + 1) * T]
        term1[ki] = (fz.conj() * diff).sum().real / sig / T
    t1_noise = n_obs / sig - ((v.to(_CD).conj() * bn).sum(dim=1).real / sig).mean()
    term1[H] = t1_noise
    if variance_idx is not None:
        term2[variance_idx] = (y_alpha - sig * a_norm) / variance
        term1[variance_idx] = (n_obs - sig * t1_noise) / variance
    return 0.5 * (term1 - term2), term1, term2, y_alpha


@pytest.mark.parametrize("M,T,H,variance_idx,trace_idx", [(529, 5, 2, 1, [0]), (529, 1, 2, 0, [1]), (9261, 3, 3, None, [0, 2]),
                                                          (100000, 2, 4, 3, [0, 1, 2]), (81, 4, 1, 0, []), (70000, 2, 0, None, [])])
def test_assemble_matches_torch_restatement(M, T, H, variance_idx, trace_idx):
    from efgp_hip import gradient_assemble
    g = torch.Generator().manual_seed(M + T)
    K = len(trace_idx)
    fy, tg, beta = _crandn((M,), g), _crandn((M,), g), _crandn((M,), g)
    ws = torch.rand(M, generator=g, dtype=torch.float64).to(_CD)
    dp = _crandn((M, H), g)
    fz = _crandn((T, M), g)
    v = (torch.randint(0, 2, (T, M), generator=g) * 2 - 1).to(torch.float64)
    beta_all = _crandn(((K + 1) * T, M), g)
    sig, n_obs, yy, variance = 0.31, 12345.0, 2.5 * M, 1.7
    ref = _assemble_reference(fy, tg, ws, beta, dp, fz, v, beta_all, variance_idx, trace_idx, sig, n_obs, yy, variance)
    out = gradient_assemble(fy.cuda(), tg.cuda(), ws.cuda(), beta.cuda(), dp.cuda() if H else None, fz.cuda() if K else None, v.cuda(),
                            beta_all.cuda(), variance_idx=variance_idx, trace_idx=trace_idx, sigmasq=sig, n_obs=n_obs, yy=yy,
                            variance=variance)
    out2 = gradient_assemble(fy.cuda(), tg.cuda(), ws.cuda(), beta.cuda(), dp.cuda() if H else None, fz.cuda() if K else None, v.cuda(),
                             beta_all.cuda(), variance_idx=variance_idx, trace_idx=trace_idx, sigmasq=sig, n_obs=n_obs, yy=yy,
                             variance=variance)
    assert torch.equal(out, out2)                              # fixed summation order: reproducible bit for bit
    out = out.cpu()
    nh = H + 1
    assert out.numel() == 3 * nh + 1
    scale = float(max(ref[1].abs().max(), ref[2].abs().max()))
    for q in range(3):                                         # grad | term1 | term2: sums of M (T) terms in another order
        assert float((out[q * nh:(q + 1) * nh] - ref[q]).abs().max()) < 1e-12 * scale
    assert abs(float(out[3 * nh]) - float(ref[3])) < 1e-12 * abs(float(ref[3]))


def test_assemble_argument_checks():
    from efgp_hip import gradient_assemble
    g = torch.Generator().manual_seed(0)
    M, T = 64, 2
    z = _crandn((M,), g).cuda()
    v = torch.ones((T, M), dtype=torch.float64, device="cuda")
    ball = _crandn((2 * T, M), g).cuda()
    fz = _crandn((T, M), g).cuda()
    with pytest.raises(ValueError):                             # five kernel hyper-parameters: beyond the kernel's slots
        gradient_assemble(z, z, z, z, _crandn((M, 5), g).cuda(), fz, v, ball, variance_idx=None, trace_idx=[0], sigmasq=1.0, n_obs=1.0,
                          yy=1.0, variance=1.0)
    with pytest.raises(ValueError):                             # the variance hyper has no trace estimate (:176)
        gradient_assemble(z, z, z, z, _crandn((M, 2), g).cuda(), fz, v, ball, variance_idx=0, trace_idx=[0], sigmasq=1.0, n_obs=1.0,
                          yy=1.0, variance=1.0)
    with pytest.raises(ValueError):
        gradient_assemble(z, z, z, z, _crandn((M, 2), g).cuda(), fz, v, ball, variance_idx=1, trace_idx=[0], sigmasq=0.0, n_obs=1.0,
                          yy=1.0, variance=1.0)


def _problem(d, N, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(N, d, generator=g, dtype=torch.float64).cuda()
    y = (torch.sin(3 * x.sum(1)) + 0.3 * torch.randn(N, generator=g, dtype=torch.float64).cuda()).contiguous()
    return x, y


@pytest.mark.parametrize("case", ["se2", "matern2", "se3", "se1"])
def test_native_tail_equals_torch_sequence(case, monkeypatch):
    """Same probes, CG run to its floor: the two code paths compute the same estimator (differences: summation order)."""
    from efgpnd import efgpnd_gradient_batched
    from kernels.squared_exponential import SquaredExponential
    from kernels.matern import Matern
    kern, d, N = {"se2": (SquaredExponential(dimension=2, init_lengthscale=0.1, init_variance=1.0), 2, 20000),
                  "matern2": (Matern(dimension=2, nu=1.5, init_lengthscale=0.3, init_variance=1.3), 2, 20000),
                  "se3": (SquaredExponential(dimension=3, init_lengthscale=0.4, init_variance=0.7), 3, 5000),
                  "se1": (SquaredExponential(dimension=1, init_lengthscale=0.05, init_variance=0.7), 1, 5000)}[case]
    x, y = _problem(d, N, 3)
    T = 4
    sig = torch.tensor(0.09, dtype=torch.float64)
    st = {}
    efgpnd_gradient_batched(x, y, sig, kern, 1e-4, 1, stats_out=st, probe_seed=11)
    M = st["feature_count"]
    pv = (torch.randint(0, 2, (T, M), generator=torch.Generator().manual_seed(5)) * 2 - 1).to(torch.float64).cuda()
    pl = (torch.randint(0, 2, (8, M), generator=torch.Generator().manual_seed(6)) * 2 - 1).to(torch.float64).cuda()
    outs = []
    for mode in ("native", "torch"):
        if mode == "torch":
            monkeypatch.setenv("EFGP_NO_FUSED_GRADIENT", "1")
        st = {}
        gr, lm = efgpnd_gradient_batched(x, y, sig, kern, 1e-4, T, stats_out=st, probe_seed=11, cg_tol=1e-11, probes_V=pv,
                                         compute_log_marginal=True, log_marginal_probes=8, log_marginal_steps=10,
                                         log_marginal_probe_vectors=pl)
        outs.append((gr.cpu(), st["term1"], st["term2"], float(lm), int(st["mean_cg_iters"]), int(st["trace_cg_iters"]),
                     int(st["trace_num_rhs"])))
    monkeypatch.delenv("EFGP_NO_FUSED_GRADIENT")
    a, b = outs
    assert a[4:] == b[4:]
    for q in range(3):
        assert float((a[q] - b[q]).abs().max() / b[q].abs().max()) < 1e-7
    assert abs(a[3] - b[3]) <= 1e-9 * abs(b[3])


def test_model_gradient_uses_one_read_back(monkeypatch):
    """EFGPND.compute_gradients through the native tail: same raw gradient as through the torch sequence at the default
    tolerances (one CG stopping index may move), stats carry term1 / term2 on the host."""
    from efgpnd import EFGPND
    from kernels.squared_exponential import SquaredExponential
    x, y = _problem(2, 30000, 7)
    res = []
    for mode in ("native", "torch"):
        if mode == "torch":
            monkeypatch.setenv("EFGP_NO_FUSED_GRADIENT", "1")
        kern = SquaredExponential(dimension=2, init_lengthscale=0.2, init_variance=1.0)
        model = EFGPND(x, y, kern, sigmasq=0.09, eps=1e-4, estimate_params=False)
        torch.manual_seed(0)
        g1 = model.compute_gradients(trace_samples=4, probe_seed=5, cg_tol=1e-10)
        st = model.last_gradient_stats
        assert st["term1"].device.type == "cpu" and st["term2"].device.type == "cpu" and "grad_host" not in st
        assert model._gp_params.raw.grad is not None and torch.equal(model._gp_params.raw.grad, g1)
        res.append((g1.clone(), st["term2"].clone(), int(st["mean_cg_iters"])))
    monkeypatch.delenv("EFGP_NO_FUSED_GRADIENT")
    # term 2 has no random probes: identical estimator on both paths
    assert float((res[0][1] - res[1][1]).abs().max() / res[1][1].abs().max()) < 1e-8
    assert res[0][2] == res[1][2]
