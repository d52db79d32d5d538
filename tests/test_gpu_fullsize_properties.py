"""GPU checks at the BASELINE sizes (N = 1e6 and 1e7, d = 2) through size-independent properties.

The oracle's exact NUDFT is O(N M) and only finishes in seconds for a few thousand points, so at full size the
HIP path is pinned by identities the transforms must satisfy (adjointness, linearity, the k = 0 mode, Hermitian
symmetry, additivity over point shards), by exact evaluation on a random subset of targets / modes, and by the
residual of the solved system.  Tolerances are the requested NUFFT / CG tolerances, stated per assertion."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

H = 0.346             # the grid spacing of BASELINE configs[1] (2-D SE, l = 0.2, eps = 1e-4): mtot = 23
MTOT = 23
TOL = 1e-7


def _rel(a, b):
    return float(torch.linalg.norm((a - b).reshape(-1)) / torch.linalg.norm(b.reshape(-1)))


def _data(N, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.rand(N, 2, generator=g, dtype=torch.float64, device="cuda") * 2 - 1
    y = torch.sin(3 * x[:, 0]) * torch.cos(4 * x[:, 1]) + 0.2 * torch.randn(N, generator=g, dtype=torch.float64, device="cuda")
    return x, y


def _exact_type1_modes(x, c, modes):
    """sum_n c_n exp(-2 pi i h k.x_n) for a few modes k (rows of `modes`), on the device in float64."""
    ph = -2 * math.pi * H * (x @ modes.T.to(x.dtype))                      # (N, K)
    return (torch.complex(torch.cos(ph), torch.sin(ph)) * c[:, None].to(torch.complex128)).sum(0)


@pytest.mark.parametrize("N", [1_000_000, 10_000_000])
def test_type1_type2_identities_at_full_size(N):
    from efgp_hip import NufftPlan
    x, y = _data(N, seed=N % 97)
    plan = NufftPlan(x, H, TOL)
    shape = (MTOT, MTOT)
    conv = (2 * MTOT - 1, 2 * MTOT - 1)
    Fy, v = plan.type1_pair(y, shape, conv)
    # k = 0 mode of the ones channel counts the points; the Toeplitz vector is Hermitian (efgpnd.py:1395-1421)
    c = MTOT - 1
    assert abs(complex(v[c, c]) - N) < 5 * TOL * N
    assert _rel(v.flip(0, 1).conj(), v) < 1e-12
    assert _rel(Fy.flip(0, 1).conj(), Fy) < 1e-12                          # real strengths
    # exact values on a random subset of modes
    g = torch.Generator().manual_seed(1)
    sel = torch.randint(0, MTOT, (6, 2), generator=g)
    k = (sel - (MTOT - 1) // 2).to(torch.float64).cuda()
    ref = _exact_type1_modes(x, y, k)
    got = Fy[sel[:, 0].cuda(), sel[:, 1].cuda()]
    assert float((got - ref).abs().max() / Fy.abs().max()) < 5 * TOL
    # linearity: F*(a y + b z) = a F*y + b F*z
    z = torch.cos(5 * x[:, 0] + x[:, 1])
    Fz = plan.type1(z.to(torch.complex128), shape)
    Fyz = plan.type1((0.3 * y - 1.7 * z).to(torch.complex128), shape)
    assert _rel(Fyz, 0.3 * Fy - 1.7 * Fz) < 5 * TOL
    # adjointness: <F* y, f> = <y, F f>
    gf = torch.Generator().manual_seed(2)
    f = torch.complex(torch.randn(shape, generator=gf, dtype=torch.float64), torch.randn(shape, generator=gf, dtype=torch.float64)).cuda()
    Ff = plan.type2(f, shape)
    lhs = torch.vdot(Fy.reshape(-1), f.reshape(-1))
    rhs = torch.vdot(y.to(torch.complex128), Ff)
    scale = float(torch.linalg.norm(y) * torch.linalg.norm(Ff))
    assert abs(complex(lhs - rhs)) < 5 * TOL * scale
    # type 2 on a subset of targets against the explicit sum, and real_only == real part
    idx = torch.randint(0, N, (64,), generator=g).cuda()
    kk = torch.cartesian_prod(torch.arange(-(MTOT // 2), MTOT // 2 + 1), torch.arange(-(MTOT // 2), MTOT // 2 + 1)).to(torch.float64).cuda()
    ph = 2 * math.pi * H * (x[idx] @ kk.T)
    exact = (torch.complex(torch.cos(ph), torch.sin(ph)) * f.reshape(-1)[None, :]).sum(1)
    assert float((Ff[idx] - exact).abs().max() / Ff.abs().max()) < 5 * TOL
    assert _rel(plan.type2(f, shape, real_only=True), Ff.real) < 5 * TOL
    # additivity over point shards (what the all-reduce of sharded fits relies on)
    halves = [NufftPlan(x[:N // 2], H, TOL).type1_pair(y[:N // 2], shape, conv),
              NufftPlan(x[N // 2:], H, TOL).type1_pair(y[N // 2:], shape, conv)]
    assert _rel(halves[0][0] + halves[1][0], Fy) < 5 * TOL and _rel(halves[0][1] + halves[1][1], v) < 5 * TOL
    # in-kernel probes equal the materialised ones at this size too
    from efgp_hip import rademacher_fill
    Zf = plan.type1_rademacher(3, 2, shape)
    Z = rademacher_fill(x.device, 3, 2, N)
    assert set(Z.unique().tolist()) == {-1.0, 1.0} and abs(float(Z.mean())) < 5 / math.sqrt(N)
    assert _rel(Zf, plan.type1(Z.to(torch.complex128), shape)) < 5 * TOL


def test_fit_residual_and_mean_at_full_size():
    """N = 1e6 fit of BASELINE configs[1]: the solved beta satisfies the normal equations to the CG tolerance, the
    posterior mean equals the explicit feature sum on a subset, and a refit reproduces it bit for bit."""
    from efgpnd import EFGPND, create_A_mean
    from kernels.squared_exponential import SquaredExponential
    N = 1_000_000
    x, y = _data(N, seed=3)
    k = SquaredExponential(dimension=2, init_lengthscale=0.2, init_variance=2.0)
    tol = 1e-6
    m = EFGPND(x, y, k, sigmasq=0.2, eps=1e-4, nufft_eps=TOL, estimate_params=False,
               opts={"cg_tolerance": tol, "mean_cg_warm_start": False, "point_layout": True})   # same spreader in both fits below
    mean, _ = m.predict(x, return_variance=False)
    st = m._fit_state
    assert st["mtot"] == MTOT
    A = create_A_mean(st["ws"], m._toeplitz, st["sig"], torch.complex128)
    rhs = st["ws"] * st["Fy"]
    assert _rel(A(st["beta"]), rhs) < 1.05 * tol                          # cg.py:132 stopping rule
    g = torch.Generator().manual_seed(5)
    idx = torch.randint(0, N, (128,), generator=g).cuda()
    kk = torch.cartesian_prod(torch.arange(-(MTOT // 2), MTOT // 2 + 1), torch.arange(-(MTOT // 2), MTOT // 2 + 1)).to(torch.float64).cuda()
    ph = 2 * math.pi * st["h"] * (x[idx] @ kk.T)
    exact = (torch.complex(torch.cos(ph), torch.sin(ph)) * (st["ws"] * st["beta"]).reshape(-1)[None, :]).sum(1).real
    assert float((mean[idx] - exact).abs().max() / mean.abs().max()) < 5 * TOL
    # the data term dominates: the fit explains most of the signal (sanity of the whole pipeline at scale)
    assert float(((mean - y) ** 2).mean()) < 0.06 and float((y ** 2).mean()) > 0.2
    beta0 = st["beta"].clone()
    m._compute_common_parameters(force_recompute=True)
    assert torch.equal(m._fit_state["beta"], beta0)                       # fixed-point spreader + fixed CG: reproducible


def test_layout_spreader_at_full_size():
    """N = 1e7, d = 2 (the north_star configuration) through the per-model point layout (MFMA spreader): equals the plain
    plan to the tolerance, exact on a subset of modes, counts the points in the k = 0 mode, reproducible bit for bit."""
    from efgp_hip import NufftPlan, PointSet
    N = 10_000_000
    x, y = _data(N, seed=11)
    shape, conv = (MTOT, MTOT), (2 * MTOT - 1, 2 * MTOT - 1)
    pts = PointSet(x, values=y)
    plan = NufftPlan(x, H, TOL, points=pts)
    Fy, v = plan.type1_pair(y, shape, conv)
    c = MTOT - 1
    assert abs(complex(v[c, c]) - N) < 1e-9 * N                                 # exact integer accumulation of the ones channel
    assert _rel(v.flip(0, 1).conj(), v) < 1e-12 and _rel(Fy.flip(0, 1).conj(), Fy) < 1e-12
    g = torch.Generator().manual_seed(1)
    sel = torch.randint(0, MTOT, (6, 2), generator=g)
    k = (sel - (MTOT - 1) // 2).to(torch.float64).cuda()
    ref = _exact_type1_modes(x, y, k)
    assert float((Fy[sel[:, 0].cuda(), sel[:, 1].cuda()] - ref).abs().max() / Fy.abs().max()) < 5 * TOL
    Fy0, v0 = NufftPlan(x, H, TOL).type1_pair(y, shape, conv)
    assert _rel(Fy, Fy0) < 5 * TOL and _rel(v, v0) < 5 * TOL
    Fy2, v2 = plan.type1_pair(y, shape, conv)
    assert torch.equal(Fy, Fy2) and torch.equal(v, v2)
    # generated probes through the sorted copies equal the materialised ones
    from efgp_hip import rademacher_fill
    Zf = plan.type1_rademacher(5, 3, shape)
    Z = rademacher_fill(x.device, 5, 3, N)
    assert _rel(Zf, NufftPlan(x, H, TOL).type1(Z, shape)) < 5 * TOL


def test_c5_fullsize_3d_properties():
    """BASELINE configs[4] at full size: 3-D Matern-3/2 (l = 0.2, eps = 1e-3 -> mtot = 57, circulant grid 128^3), N = 5e6.
    Adjointness, the k = 0 mode, shard additivity, exact values on a subset, residual of the solve, and one
    hyper-gradient step (finite, reproducible for a fixed probe seed)."""
    from efgpnd import EFGPND, create_A_mean
    from efgp_hip import NufftPlan
    from kernels.matern import Matern
    N, d = 5_000_000, 3
    g = torch.Generator(device="cuda").manual_seed(21)
    x = torch.rand(N, d, generator=g, dtype=torch.float64, device="cuda") * 2 - 1
    y = torch.sin(3 * x[:, 0]) * torch.cos(4 * x[:, 1]) * torch.cos(2 * x[:, 2]) + 0.3 * torch.randn(N, generator=g, dtype=torch.float64, device="cuda")
    k = Matern(dimension=d, nu=1.5, init_lengthscale=0.2, init_variance=1.5)
    tol, ntol = 1e-5, 1e-6
    m = EFGPND(x, y, k, sigmasq=0.2, eps=1e-3, nufft_eps=ntol, estimate_params=False,
               opts={"cg_tolerance": tol, "mean_cg_warm_start": False})
    m.fit()
    st = m._fit_state
    mtot, h = st["mtot"], st["h"]
    assert mtot == 57 and list(m._toeplitz.fft_shape) == [128, 128, 128]
    shape = (mtot,) * d
    Fy, v = st["Fy"].reshape(shape), st["v"]
    c = 2 * ((mtot - 1) // 2)
    assert abs(complex(v[c, c, c]) - N) < 5 * 6e-8 * N                              # the Toeplitz vector is always computed to 6e-8
    assert _rel(v.flip(0, 1, 2).conj(), v) < 1e-12
    # exact values of F*y on a few modes
    gs = torch.Generator().manual_seed(2)
    sel = torch.randint(0, mtot, (4, d), generator=gs)
    kk = (sel - (mtot - 1) // 2).to(torch.float64).cuda()
    ph = -2 * math.pi * h * (x @ kk.T)
    ref = (torch.complex(torch.cos(ph), torch.sin(ph)) * y[:, None].to(torch.complex128)).sum(0)
    got = Fy[sel[:, 0].cuda(), sel[:, 1].cuda(), sel[:, 2].cuda()]
    assert float((got - ref).abs().max() / Fy.abs().max()) < 5 * 6e-8
    # adjointness <F* y, f> = <y, F f> through a plan at the prediction tolerance, and additivity over two shards
    plan = NufftPlan(x, h, ntol)
    f = torch.complex(torch.randn(shape, generator=gs, dtype=torch.float64), torch.randn(shape, generator=gs, dtype=torch.float64)).cuda()
    Ff = plan.type2(f, shape)
    Fy_p = plan.type1(y, shape)
    lhs, rhs = torch.vdot(Fy_p.reshape(-1), f.reshape(-1)), torch.vdot(y.to(torch.complex128), Ff)
    assert abs(complex(lhs - rhs)) < 5 * ntol * float(torch.linalg.norm(y) * torch.linalg.norm(Ff))
    half = N // 2
    parts = NufftPlan(x[:half], h, ntol).type1(y[:half], shape) + NufftPlan(x[half:], h, ntol).type1(y[half:], shape)
    assert _rel(parts, Fy_p) < 5 * ntol
    # the solved beta satisfies the normal equations to the CG tolerance (cg.py:132)
    A = create_A_mean(st["ws"], m._toeplitz, st["sig"], torch.complex128)
    assert _rel(A(st["beta"]), st["ws"] * st["Fy"]) < 1.05 * tol
    # one hyper-gradient step of the training loop: finite, and identical when the probe seed is fixed
    V = torch.ones(2, st["ws"].numel(), dtype=torch.float64)
    V[1, ::2] = -1
    g1 = m.compute_gradients(trace_samples=2, cg_tol=1e-3, probe_seed=99, probes_V=V).detach().cpu()
    g2 = m.compute_gradients(trace_samples=2, cg_tol=1e-3, probe_seed=99, probes_V=V).detach().cpu()
    assert torch.isfinite(g1).all() and g1.abs().max() > 0
    assert float((g1 - g2).abs().max()) <= 1e-9 * float(g1.abs().max())          # warm-started mean solve: same to rounding
