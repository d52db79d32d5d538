"""GPU tests of the drop-in call surface: public classes / functions behave like the
reference's -- shapes, devices, dtypes, layouts, error behaviour -- with values checked against the oracle."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a = torch.as_tensor(a).detach().cpu().reshape(-1).to(torch.complex128)
    b = torch.as_tensor(b).detach().cpu().reshape(-1).to(torch.complex128)
    return float(torch.linalg.norm(a - b) / torch.linalg.norm(b))


def _data(N=3000, d=2, seed=0, dtype=torch.float64):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(N, d, generator=g, dtype=torch.float64) * 2 - 1
    y = torch.sin(3 * x[:, 0]) + (0.5 * torch.cos(2 * x[:, 1]) if d > 1 else 0) + 0.1 * torch.randn(N, generator=g, dtype=torch.float64)
    return x.to(dtype), y.to(dtype)


def test_nufft_class_cpu_tensors_roundtrip():
    """CPU tensors in -> CPU tensors out (the reference runs on x.device); flat, shaped and batched layouts."""
    from efgpnd import NUFFT
    from oracle import efgp_oracle as O
    x, y = _data(2000, 2)
    op = NUFFT(x, torch.zeros(2, dtype=torch.float64), 0.3, 1e-9)
    assert op.phi.shape == (2, 2000) and op.phi.device.type == "cpu"
    assert torch.allclose(op.phi, (2 * math.pi * 0.3 * x).T)
    f = op.type1(y, out_shape=(15, 15))
    assert f.device.type == "cpu" and f.dtype == torch.complex128 and f.shape == (15, 15)
    assert _rel(f, O.nudft_type1(x, 0.3, y, (15, 15))) < 1e-8
    fb = op.type1(torch.stack([y, 2 * y, -y]).to(torch.complex128), out_shape=(15, 15))
    assert fb.shape == (3, 15, 15) and _rel(fb[1], 2 * f) < 1e-8
    c = op.type2(f.reshape(-1), out_shape=(15, 15))
    assert c.shape == (2000,) and _rel(c, O.nudft_type2(x, 0.3, f, (15, 15))) < 1e-8
    cb = op.type2(fb.reshape(3, -1), out_shape=(15, 15))
    assert cb.shape == (3, 2000) and _rel(cb[2], -c) < 1e-8
    assert _rel(op.type2(f), c) < 1e-12                       # already shaped, no out_shape
    with pytest.raises(ValueError):
        op.type2(f.reshape(-1))
    # non-zero centre (reference: phi = 2 pi h (x - xcen))
    op2 = NUFFT(x, torch.tensor([0.3, -0.2], dtype=torch.float64), 0.3, 1e-9)
    assert _rel(op2.type1(y, out_shape=(9, 9)), O.nudft_type1(x - torch.tensor([0.3, -0.2], dtype=torch.float64), 0.3, y, (9, 9))) < 1e-8


def test_toeplitz_layouts_and_errors():
    from efgpnd import ToeplitzND, compute_convolution_vector_vectorized_dD
    from oracle import efgp_oracle as O
    x, _ = _data(500, 2)
    v = compute_convolution_vector_vectorized_dD(3, x, 0.4)
    assert v.shape == (13, 13) and v.device.type == "cpu" and v.dtype == torch.complex128
    assert _rel(v, O.conv_vector(x, 0.4, 3)) < 5e-7
    assert abs(v[6, 6].real - 500) < 1e-4
    T = ToeplitzND(v)
    To = O.Toeplitz(O.conv_vector(x, 0.4, 3))
    assert T.ns == [7, 7] and T.Ls == [13, 13] and T.size == 49 and T.d == 2 and T.fft_shape == [16, 16]
    g = torch.Generator().manual_seed(1)
    u = torch.complex(torch.randn(4, 49, generator=g, dtype=torch.float64), torch.randn(4, 49, generator=g, dtype=torch.float64))
    assert _rel(T(u), To(u)) < 1e-6                      # flat batch
    assert T(u).shape == (4, 49) and T(u[0]).shape == (49,)
    assert T(u.reshape(4, 7, 7)).shape == (4, 7, 7)      # block batch
    assert _rel(T(u.reshape(2, 2, 7, 7)), To(u).reshape(2, 2, 7, 7)) < 1e-6
    assert T(u[0].real).dtype == torch.complex128        # real input is promoted
    with pytest.raises(ValueError):
        T(torch.zeros(50, dtype=torch.complex128))
    assert ToeplitzND(v, force_pow2=False).fft_shape == [16, 16] or ToeplitzND(v, force_pow2=False).fft_shape[0] >= 13


def test_operators_and_cg_dispatch():
    """create_A_mean / create_A_var / create_jacobi_precond compose with ConjugateGradients like the reference's
    closures; the solve runs fused (iters and solution match the oracle CG)."""
    from efgpnd import ToeplitzND, create_A_mean, create_A_var, create_Gv, create_jacobi_precond, setup_operators
    from cg import ConjugateGradients
    from oracle import efgp_oracle as O
    x, _ = _data(800, 2, seed=2)
    vo = O.conv_vector(x, 0.35, 5)
    To = O.Toeplitz(vo)
    g = torch.Generator().manual_seed(3)
    ws = torch.exp(-2 * torch.rand(121, generator=g, dtype=torch.float64)).to(torch.complex128)
    b = torch.complex(torch.randn(121, generator=g, dtype=torch.float64), torch.randn(121, generator=g, dtype=torch.float64))
    T = ToeplitzND(vo)
    A = create_A_mean(ws, T, 0.4, torch.complex128)
    Av = create_A_var(ws, T, 0.4, torch.complex128)
    G = create_Gv(ws, T, torch.complex128)
    Ao = O.make_A_mean(ws, To, 0.4)
    assert _rel(A(b), Ao(b)) < 1e-12 and _rel(Av(b), O.make_A_var(ws, To, 0.4)(b)) < 1e-12
    assert _rel(G(b), ws * To(ws * b)) < 1e-12
    assert _rel(A(torch.stack([b, 2 * b])), torch.stack([Ao(b), 2 * Ao(b)])) < 1e-12
    assert len(setup_operators(ws, T, 0.4, torch.complex128)) == 3
    Minv = create_jacobi_precond(ws, 0.4, diag_scale=vo[10, 10].real)
    diag = O.jacobi_diag(ws, 0.4, 800.0)
    assert _rel(Minv(b), b / diag) < 1e-12
    cg = ConjugateGradients(A, b, torch.zeros_like(b), tol=1e-9, early_stopping=True, M_inv_apply=Minv)
    xs = cg.solve()
    xo, ito = O.cg_single(Ao, b, torch.zeros_like(b), 1e-9, diag=diag)
    assert cg.iters_completed == ito and _rel(xs, xo) < 1e-8 and xs.device.type == "cpu"
    B = torch.stack([b, b.conj(), 0.1 * b])
    cgb = ConjugateGradients(Av, B, torch.zeros_like(B), tol=1e-8, max_iter=500)
    xb = cgb.solve()
    xob, itob = O.cg_batched(O.make_A_var(ws, To, 0.4), B, torch.zeros_like(B), 1e-8, max_iter=500)
    # ~300 iterations: the first crossing of the tolerance may move by one with the compiler's FMA contraction
    assert abs(cgb.iters_completed - itob) <= 1 + itob // 200 and _rel(xb, xob) < 1e-7
    # a user-supplied preconditioner falls back to the generic loop with the same semantics
    cg2 = ConjugateGradients(A, b, torch.zeros_like(b), tol=1e-9, M_inv_apply=lambda r: r / diag)
    assert _rel(cg2.solve(), xo) < 1e-8 and cg2.iters_completed == ito


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_model_dtypes_devices_and_shapes(dtype):
    from efgpnd import EFGPND, efgp_nd
    from oracle import efgp_oracle as O
    x, y = _data(4000, 2, seed=4, dtype=dtype)
    m = EFGPND(x, y, "SE", sigmasq=0.05, eps=1e-3, nufft_eps=1e-9, estimate_params=False, opts={"cg_tolerance": 1e-8})
    m.kernel.set_hyper("lengthscale", 0.3)
    m.kernel.set_hyper("variance", 1.2)
    xn, _ = _data(300, 2, seed=5, dtype=dtype)
    mean, var = m.predict(xn, variance_method="regular")
    assert mean.shape == (300,) and var.shape == (300,) and mean.dtype == dtype and mean.device.type == "cpu"
    assert m._beta.dtype == (torch.complex64 if dtype == torch.float32 else torch.complex128)
    assert m._xis.shape[1] == 2 and hasattr(m._xis, "h_float") and m._ws.shape == m._beta.shape
    k = O.KernelSpec("se", 2, m.kernel.get_hyper("lengthscale"), m.kernel.get_hyper("variance"))
    f = O.fit(x.double(), y.double(), k, float(m.sigmasq.detach()), 1e-3, cg_tol=1e-8)
    tol = 1e-5 if dtype == torch.float64 else 2e-4       # float32 inputs are rounded on the way in / out only
    assert _rel(mean, O.predict_mean(f, xn.double())) < tol
    assert _rel(var, O.variance_regular(f, xn.double(), cg_tol=1e-8)) < 10 * tol
    _, nanvar = m.predict(xn, return_variance=False)
    assert torch.isnan(nanvar).all()
    with pytest.raises(ValueError):
        m.predict(xn, variance_method="bogus")
    with pytest.raises(ValueError):
        m.predict(xn[:, :1])
    if dtype == torch.float64:
        beta, xis, ytrg, ws, toep = efgp_nd(x, y, float(m.sigmasq.detach()), m.kernel, 1e-3, xn, nufft_eps=1e-9,
                                            opts={"cg_tolerance": 1e-8, "estimate_variance": True, "variance_method": "regular"})
        assert _rel(ytrg["mean"], mean) < 1e-6 and _rel(ytrg["var"], var) < 1e-5 and beta.shape == ws.shape


def test_one_dimensional_inputs_and_cuda_tensors():
    from efgpnd import EFGPND
    from oracle import efgp_oracle as O
    x, y = _data(3000, 1, seed=6)
    m = EFGPND(x[:, 0].cuda(), y.cuda(), "Matern32", sigmasq=0.1, eps=1e-3, nufft_eps=1e-9, estimate_params=False,
               opts={"cg_tolerance": 1e-9})
    m.kernel.set_hyper("lengthscale", 0.4)
    m.kernel.set_hyper("variance", 0.9)
    xn = torch.linspace(-1, 1, 64, dtype=torch.float64)
    mean, var = m.predict(xn.cuda(), variance_method="regular")
    assert mean.is_cuda and mean.shape == (64,)
    k = O.KernelSpec("matern", 1, m.kernel.get_hyper("lengthscale"), m.kernel.get_hyper("variance"), 1.5)
    f = O.fit(x, y, k, float(m.sigmasq.detach()), 1e-3, cg_tol=1e-9)
    assert _rel(mean, O.predict_mean(f, xn)) < 1e-5
    assert _rel(var, O.variance_regular(f, xn[:, None], cg_tol=1e-9)) < 1e-4


def test_refit_on_changed_hypers_and_training_loop():
    from efgpnd import EFGPND
    x, y = _data(5000, 2, seed=7)
    m = EFGPND(x.cuda(), y.cuda(), "SquaredExponential", eps=1e-3)      # estimate_params=True path
    xn = x[:50].cuda()
    m1, _ = m.predict(xn, return_variance=False)
    it1 = m.last_fit_stats["mean_cg_iters"]
    m.kernel.set_hyper("lengthscale", 2.0 * m.kernel.get_hyper("lengthscale"))
    m2, _ = m.predict(xn, return_variance=False)                          # changed hypers -> refit
    assert not torch.allclose(m1, m2)
    opt = m.register_optimizer(torch.optim.Adam(m.parameters(), lr=0.05))
    before = m._gp_params.raw.detach().clone()
    for _ in range(3):
        opt.zero_grad()
        g = m.compute_gradients(trace_samples=3)
        assert g.shape == (3,) and torch.isfinite(g).all()
        opt.step()
    assert not torch.equal(before, m._gp_params.raw.detach())
    st = m.last_gradient_stats
    assert st["trace_num_rhs"] == 6 and st["trace_samples"] == 3 and "stage_sec" in st
    m.optimize_hyperparameters(max_iters=2, trace_samples=2, log_interval=1)
    assert len(m.training_log["gradients"]) == 2 and m._fitted
    lm = m.predict(xn, return_variance=False, compute_log_marginal=True)[2]
    assert math.isfinite(float(lm))


def test_logdet_slq_against_dense():
    from efgpnd import ToeplitzND, logdet_slq
    from oracle import efgp_oracle as O
    x, _ = _data(300, 1, seed=8)
    vo = O.conv_vector(x, 0.4, 6)
    To = O.Toeplitz(vo)
    M = To.size
    g = torch.Generator().manual_seed(9)
    ws = torch.exp(-torch.rand(M, generator=g, dtype=torch.float64)).to(torch.complex128)
    Tm = torch.stack([To(torch.eye(M, dtype=torch.complex128)[i]) for i in range(M)], dim=1)
    dense = torch.eye(M, dtype=torch.complex128) + (ws[:, None] * Tm * ws[None, :]) / 0.5
    exact = float(torch.linalg.slogdet(dense)[1]) + 300 * math.log(0.5)
    torch.manual_seed(0)
    est = logdet_slq(ws, 0.5, ToeplitzND(vo), probes=200, steps=M, n=300)
    assert abs(est - exact) < 0.05 * abs(exact) + 2.0


def test_sample_posterior_small():
    from efgpnd import EFGPND
    x, y = _data(60, 1, seed=10)
    m = EFGPND(x, y, "SE", sigmasq=0.05, eps=1e-4, estimate_params=False)
    m.kernel.set_hyper("lengthscale", 0.3)
    m.kernel.set_hyper("variance", 1.0)
    s = m.sample_posterior(torch.linspace(-1, 1, 20, dtype=torch.float64)[:, None], 5)
    assert s.shape == (20, 5) and np.isfinite(s).all()


def test_weighted_toeplitz_and_circulant_preconditioners():
    """Row (f)4: the weighted Toeplitz operator F^* diag(w) F (pg_classifier.py:377-384) and the circulant
    preconditioners of benchmark_prism_mean_preconditioners.py:131-191, composed from the HIP operators."""
    from efgpnd import NUFFT, ToeplitzND, create_A_mean
    from cg import ConjugateGradients
    from preconditioners import (weighted_toeplitz, wrap_to_circulant_kernel, scalar_circulant_preconditioner,
                                 sandwich_circulant_preconditioner)
    from oracle import efgp_oracle as O
    x, _ = _data(3000, 2, seed=5)
    h, m = 0.3, 6
    mtot = 2 * m + 1
    g = torch.Generator().manual_seed(4)
    w = torch.rand(3000, generator=g, dtype=torch.float64) * 0.25          # Polya-Gamma style positive weights
    op = NUFFT(x.cuda(), torch.zeros(2, dtype=torch.float64), h, 1e-12)
    Tw = weighted_toeplitz(op, w.cuda(), (mtot, mtot))
    v_exact = O.nudft_type1(x, h, w.to(torch.complex128), (4 * m + 1, 4 * m + 1))
    assert _rel(op.type1(w.cuda().to(torch.complex128), out_shape=(4 * m + 1, 4 * m + 1)), v_exact) < 1e-9
    u = torch.complex(torch.randn(mtot * mtot, generator=g, dtype=torch.float64), torch.randn(mtot * mtot, generator=g, dtype=torch.float64))
    # dense check: F^* diag(w) F u
    k = torch.arange(-m, m + 1, dtype=torch.float64)
    kk = torch.cartesian_prod(k, k)
    F = torch.exp(2j * torch.pi * h * (x @ kk.T))
    dense = F.conj().T @ (w.to(torch.complex128) * (F @ u))
    assert _rel(Tw(u.cuda()), dense) < 1e-9
    # circulant wrap: same as the entry-by-entry definition
    vo = O.conv_vector(x, h, m)
    circ = wrap_to_circulant_kernel(vo)
    ref = torch.zeros(mtot, mtot, dtype=vo.dtype)
    for i in range(4 * m + 1):
        for j in range(4 * m + 1):
            ref[(i - 2 * m) % mtot, (j - 2 * m) % mtot] += vo[i, j]
    assert _rel(circ, ref) < 1e-14
    # preconditioned solves agree with the Jacobi solve and with each other
    ws = torch.exp(-0.04 * (kk ** 2).sum(1)).to(torch.complex128)
    sigmasq = 20.0
    T = ToeplitzND(vo.cuda())
    A = create_A_mean(ws.cuda(), T, sigmasq, torch.complex128)
    b = (ws * (F.conj().T @ torch.randn(3000, generator=g, dtype=torch.float64).to(torch.complex128))).cuda()
    diag = (3000.0 * ws.abs() ** 2 + sigmasq).cuda()
    base = ConjugateGradients(A, b, torch.zeros_like(b), tol=1e-9, M_inv_apply=lambda r: r / diag)
    xj = base.solve()
    assert base.iters_completed < 2 * mtot * mtot                 # converged, not capped
    for make in (lambda: scalar_circulant_preconditioner(vo.cuda(), ws.cuda(), sigmasq, "mean"),
                 lambda: scalar_circulant_preconditioner(vo.cuda(), ws.cuda(), sigmasq, "max"),
                 lambda: sandwich_circulant_preconditioner(vo.cuda(), ws.cuda(), sigmasq, "median"),
                 lambda: sandwich_circulant_preconditioner(vo.cuda(), ws.cuda(), sigmasq, "geom")):
        Minv = make()
        cgp = ConjugateGradients(A, b, torch.zeros_like(b), tol=1e-9, M_inv_apply=Minv)
        xp = cgp.solve()
        assert _rel(xp, xj) < 1e-6 and cgp.iters_completed < 2 * mtot * mtot
        assert _rel(A(xp), b) < 5e-9
        r2 = torch.randn(2, mtot * mtot, generator=g, dtype=torch.float64).to(torch.complex128).cuda()
        assert Minv(r2).shape == r2.shape and _rel(Minv(r2)[1], Minv(r2[1])) < 1e-12


def test_edge_inputs():
    """Empty / single prediction sets, duplicate and lattice points, zero, huge and tiny targets, strided inputs."""
    from efgpnd import EFGPND
    from kernels.squared_exponential import SquaredExponential
    from oracle import efgp_oracle as O
    x, y = _data(4000, 2, seed=9)
    k = lambda: SquaredExponential(dimension=2, init_lengthscale=0.3, init_variance=1.2)  # noqa: E731
    opts = {"cg_tolerance": 1e-10}
    m = EFGPND(x.cuda(), y.cuda(), k(), sigmasq=0.2, eps=1e-4, nufft_eps=1e-9, estimate_params=False, opts=opts)
    # no prediction points / one prediction point
    mean0, var0 = m.predict(torch.zeros(0, 2, dtype=torch.float64).cuda(), return_variance=False)
    assert mean0.shape == (0,) and var0.shape == (0,)
    xn = torch.tensor([[0.1, -0.4]], dtype=torch.float64)
    mean1, _ = m.predict(xn.cuda(), return_variance=False)
    ko = O.KernelSpec("se", 2, 0.3, 1.2)
    fit = O.fit(x, y, ko, 0.2, 1e-4, cg_tol=1e-10)
    assert abs(float(mean1[0]) - float(O.predict_mean(fit, xn)[0])) < 1e-8 * max(1.0, abs(float(mean1[0])))
    # strided (non-contiguous) inputs give the same fit
    xs = torch.zeros(4000, 4, dtype=torch.float64)
    xs[:, ::2] = x
    ms = EFGPND(xs.cuda()[:, ::2], y.cuda(), k(), sigmasq=0.2, eps=1e-4, nufft_eps=1e-9, estimate_params=False, opts=opts)
    assert _rel(ms.predict(xn.cuda(), return_variance=False)[0], mean1) < 1e-12
    # duplicate points and points on a lattice that coincides with fine-grid cell boundaries
    lat = torch.cartesian_prod(torch.linspace(-1, 1, 33, dtype=torch.float64), torch.linspace(-1, 1, 33, dtype=torch.float64))
    xd = torch.cat([lat, lat[:200], lat[:50]])
    g = torch.Generator().manual_seed(3)
    yd = torch.sin(2 * xd[:, 0]) + 0.1 * torch.randn(xd.shape[0], generator=g, dtype=torch.float64)
    md = EFGPND(xd.cuda(), yd.cuda(), k(), sigmasq=0.2, eps=1e-4, nufft_eps=1e-9, estimate_params=False, opts=opts)
    fd = O.fit(xd, yd, ko, 0.2, 1e-4, cg_tol=1e-10)
    xq = _data(64, 2, seed=10)[0]
    assert _rel(md.predict(xq.cuda(), return_variance=False)[0], O.predict_mean(fd, xq)) < 1e-7
    # targets: all zero, huge, tiny (the spreader's fixed-point scale follows max|y|)
    # (tiny targets are limited by the reference's own div_eps = 1e-16 in <p, A p> + eps, cg.py:57)
    for scale in (0.0, 1e60, 1e6, 0.125):            # each channel of the fused pass has its own scale / normalisation
        mz = EFGPND(x.cuda(), (y * scale).cuda(), k(), sigmasq=0.2, eps=1e-4, nufft_eps=1e-9, estimate_params=False, opts=opts)
        mq = mz.predict(xq.cuda(), return_variance=False)[0]
        assert torch.isfinite(mq).all()
        if scale == 0.0:
            assert float(mq.abs().max()) < 1e-20        # rounding leakage of the ones channel in the fused pair
        else:
            ref = m.predict(xq.cuda(), return_variance=False)[0] * scale
            assert _rel(mq, ref) < 1e-7
    # wrong shapes are rejected like the reference does (N, d = x.shape)
    with pytest.raises(ValueError):
        m.predict(torch.zeros(5, 3, dtype=torch.float64).cuda(), return_variance=False)
    with pytest.raises(ValueError):
        m.predict(None)


def test_pointwise_alpha_matches_adjoint_and_nan_targets_poison_the_fit():
    """`pointwise_alpha` (one real type-2 pass for |alpha|^2, y.alpha) agrees with the adjoint identities at moderate SNR; a
    NaN among the targets yields NaN predictions (as the reference's sums do), not finite garbage."""
    from efgpnd import EFGPND
    from kernels.squared_exponential import SquaredExponential
    g = torch.Generator().manual_seed(8)
    N = 40000
    x = torch.rand(N, 2, generator=g, dtype=torch.float64) * 2 - 1
    y = torch.sin(3 * x[:, 0]) * torch.cos(4 * x[:, 1]) + 0.2 * torch.randn(N, generator=g, dtype=torch.float64)

    def model(yy):
        k = SquaredExponential(dimension=2, init_lengthscale=0.2, init_variance=2.0)
        return EFGPND(x.cuda(), yy.cuda(), k, sigmasq=0.2, eps=1e-4, nufft_eps=1e-7, estimate_params=False,
                      opts={"cg_tolerance": 1e-10, "mean_cg_warm_start": False})
    m = model(y)
    V = torch.ones(2, 529, dtype=torch.float64)
    V[1, ::2] = -1
    ga = m.compute_gradients(trace_samples=2, cg_tol=1e-10, probe_seed=5, probes_V=V).detach().cpu()
    gp = m.compute_gradients(trace_samples=2, cg_tol=1e-10, probe_seed=5, probes_V=V, pointwise_alpha=True).detach().cpu()
    assert float((ga - gp).abs().max() / ga.abs().max()) < 1e-5
    ybad = y.clone()
    ybad[17] = float("nan")
    mb = model(ybad)
    mean, _ = mb.predict(x[:100].cuda(), return_variance=False)
    assert torch.isnan(mean).all()


def test_device_spectral_weights_match_the_kernel_classes():
    """efgp_spectral_weights (ws and hyper-derivatives on the tensor grid in one launch) against kernel.spectral_density /
    spectral_grad (efgpnd.py:766-780, kernels/*.py): equal to rounding; and the model's grid uses it."""
    import torch
    import efgpnd as E
    from kernels.squared_exponential import SquaredExponential
    from kernels.matern import Matern
    dev = torch.device("cuda", 0)
    for k, eps, L in ((SquaredExponential(dimension=2, init_lengthscale=0.2, init_variance=2.0), 1e-4, 2.0),
                      (Matern(dimension=3, nu=1.5, init_lengthscale=0.5, init_variance=1.3), 1e-2, 2.0),
                      (Matern(dimension=1, nu=2.5, init_lengthscale=0.1, init_variance=0.7), 1e-5, 1.5)):
        g = E._Grid(k, eps, L, k.dimension, dev, want_grad=True)
        ws_ref = torch.sqrt(k.spectral_density(g.xis).to(torch.complex128) * g.h ** k.dimension)
        dp_ref = (g.h ** k.dimension * k.spectral_grad(g.xis)).to(torch.complex128)
        assert g.ws.is_cuda and g.ws.shape == ws_ref.shape and g.dprime.shape == dp_ref.shape
        assert float((g.ws.cpu() - ws_ref).abs().max() / ws_ref.abs().max()) < 1e-13
        assert float((g.dprime.cpu() - dp_ref).abs().max() / dp_ref.abs().max()) < 1e-12
        assert float(g.ws.imag.abs().max()) == 0.0


def test_kernel_timing_name_filter():
    """efgp_kernel_timing_only: with a name set only that kernel's launches carry HIP events (bench.py keeps the timed region
    free of the other timers)."""
    import torch
    from efgp_hip import NufftPlan, PointSet, kernel_timing, kernel_timing_read
    g = torch.Generator().manual_seed(0)
    x = (torch.rand(50000, 2, generator=g, dtype=torch.float64) * 2 - 1).cuda()
    y = torch.randn(50000, generator=g, dtype=torch.float64).cuda()
    plan = NufftPlan(x, 0.31, 1e-7, points=PointSet(x, values=y))
    beta = torch.complex(torch.randn(23, 23, generator=g, dtype=torch.float64), torch.randn(23, 23, generator=g, dtype=torch.float64)).cuda()
    try:
        kernel_timing(True, only="spread")
        plan.type1_pair(y, (23, 23), (45, 45))
        plan.type2(beta, (23, 23), real_only=True)
        assert kernel_timing_read("spread")[1] == 1 and kernel_timing_read("interp")[1] == 0
        kernel_timing(True)
        plan.type1_pair(y, (23, 23), (45, 45))
        plan.type2(beta, (23, 23), real_only=True)
        assert kernel_timing_read("spread")[1] == 1 and kernel_timing_read("interp")[1] == 1
    finally:
        kernel_timing(False)


def test_user_kernel_with_a_non_even_spectral_density_is_a_value_error():
    """The fused solvers carry every system of a model as coefficients of real functions, which needs `ws` real and even.  The
    built-in kernels are; a user subclass whose spectral_density is not even (not the density of a real stationary kernel) must
    fail at the fit with a ValueError -- not with NaN coefficients from the Hermitian kernel's refusal.  A subclass with an even
    density (here: the parent's, scaled) goes through the same torch path and fits."""
    from efgpnd import EFGPND
    from kernels.squared_exponential import SquaredExponential

    class Skewed(SquaredExponential):
        def spectral_density(self, xid):
            S = super().spectral_density(xid)
            first = xid[..., 0] if xid.ndim > 1 else xid
            return S * (1.0 + 0.1 * torch.tanh(first))

    class Scaled(SquaredExponential):
        def spectral_density(self, xid):
            return 0.5 * super().spectral_density(xid)

    g = torch.Generator().manual_seed(0)
    x = (torch.rand(2000, 2, dtype=torch.float64, generator=g) * 2 - 1).cuda()
    y = torch.sin(3 * x[:, 0]) + 0.1 * torch.randn(2000, dtype=torch.float64, generator=g).cuda()
    bad = EFGPND(x, y, Skewed(dimension=2, init_lengthscale=0.3, init_variance=1.0), sigmasq=0.1, eps=1e-3, estimate_params=False)
    with pytest.raises(ValueError, match="even"):
        bad.fit()
    ok = EFGPND(x, y, Scaled(dimension=2, init_lengthscale=0.3, init_variance=1.0), sigmasq=0.1, eps=1e-3, estimate_params=False)
    ok.fit()
    mean, _ = ok.predict(x[:50], return_variance=False)
    assert torch.isfinite(mean).all() and int(ok.last_fit_stats["mean_cg_iters"]) > 0
