"""GPU tests of the M-scale variance entry points (efgp_lag_sums, efgp_variance_rhs, efgp_variance_contract) against the
reference's torch formulas (efgpnd.py:1660-1664, 1805-1820), and of the RCCL collectives of the C ABI on a world of one."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


# the transform runs at the next 2^k / 3 * 2^k length >= 2 mtot - 1 (exact at mtot 1, 2; padded elsewhere): same lags
@pytest.mark.parametrize("d,mtot,J", [(1, 35, 5), (2, 23, 7), (2, 8, 3), (3, 9, 4), (1, 1, 2), (2, 2, 3), (2, 17, 2), (3, 13, 2), (1, 301, 2),
                                      (2, 32, 2), (2, 33, 2)])     # 2-D lag boxes up to 64 x 64 run on the library's own 64 x 64 transforms
def test_lag_sums_match_fft_correlation(d, mtot, J):
    from efgp_hip import lag_sums
    g = torch.Generator().manual_seed(3)
    M = mtot ** d
    gam = torch.complex(torch.randn(J, M, generator=g, dtype=torch.float64), torch.randn(J, M, generator=g, dtype=torch.float64))
    eta = (torch.randint(0, 2, (J, M), generator=g) * 2 - 1).to(torch.float64)
    out = lag_sums(gam.cuda(), eta.cuda(), mtot, d).cpu()
    shp = (J,) + (mtot,) * d
    s = (2 * mtot - 1,) * d
    dims = tuple(range(1, d + 1))
    G = torch.fft.fftn(gam.view(shp), s=s, dim=dims)
    E = torch.fft.fftn(eta.view(shp).to(torch.complex128), s=s, dim=dims)
    ref = torch.fft.ifftn(G * torch.conj(E), s=s, dim=dims).mean(dim=0)            # efgpnd.py:1660-1664
    assert out.shape == ref.shape
    assert float((out - ref).abs().max() / ref.abs().max()) < 1e-12


@pytest.mark.parametrize("d,mtot", [(1, 35), (2, 23), (3, 9)])
def test_variance_rows_and_contraction(d, mtot):
    from efgp_hip import variance_rhs, variance_contract
    g = torch.Generator().manual_seed(4)
    B, h = 37, 0.31
    M = mtot ** d
    x = torch.rand(B, d, generator=g, dtype=torch.float64) * 2 - 1
    ws = torch.complex(torch.rand(M, generator=g, dtype=torch.float64), torch.zeros(M, dtype=torch.float64))
    gamma = torch.complex(torch.randn(B, M, generator=g, dtype=torch.float64), torch.randn(B, M, generator=g, dtype=torch.float64))
    k1 = torch.arange(mtot, dtype=torch.float64) - (mtot - 1) // 2
    xis = h * torch.stack(torch.meshgrid(*([k1] * d), indexing="ij"), dim=-1).view(-1, d)
    ang = 2 * math.pi * (x @ xis.T)
    fx = torch.polar(torch.ones_like(ang), ang)                                       # efgpnd.py:1808-1811
    rhs = variance_rhs(x.cuda(), h, mtot, ws.cuda()).cpu()
    assert float((rhs - ws * fx.conj()).abs().max()) < 1e-12
    out = variance_contract(x.cuda(), h, mtot, ws.cuda(), gamma.cuda()).cpu()
    ref = torch.real((fx * (ws * gamma)).sum(dim=-1)).clamp_min(0.0)                  # efgpnd.py:1817-1820
    assert float((out - ref).abs().max()) < 1e-10 * float(ref.abs().max() + 1.0)
    assert (out >= 0).all() and (out == 0).any() == bool((ref == 0).any())


def test_rccl_comm_world_of_one():
    """efgp_comm_* on a single rank: the collectives must run (RCCL initialises, stream-ordered calls) and be identities."""
    from efgp_hip import RcclComm
    from efgp_hip.dist import PointShards
    dev = torch.device("cuda", 0)
    comm = RcclComm(dev, 0, 1, RcclComm.make_id())
    t = torch.arange(10, dtype=torch.float64, device=dev)
    c = torch.complex(t, -t)
    comm.all_reduce_sum_(t)
    comm.all_reduce_sum_(c)
    lo = torch.tensor([-1.0, 2.0], dtype=torch.float64, device=dev)
    comm.all_reduce_minmax_(lo, False)
    b = torch.arange(7, dtype=torch.int64, device=dev)
    comm.broadcast_(b, 0)
    torch.cuda.synchronize()
    assert torch.equal(t.cpu(), torch.arange(10, dtype=torch.float64)) and torch.equal(c.real.cpu(), t.cpu())
    assert lo.tolist() == [-1.0, 2.0] and b.tolist() == list(range(7))
    sh = PointShards(comm=comm)
    assert sh.world_size == 1 and not sh.active and sh.sum_scalars([1.5], dev) == [1.5]
    comm.close()


def test_stochastic_variance_survives_a_dead_grid_barrier(monkeypatch):
    """The Hutchinson solves of the stochastic variance on a 128 x 128 circulant grid run as ONE cooperative launch
    (cg_coop2d_kernel) whose hand-rolled grid barrier can die when its workgroups are not resident together; the systems it
    hit then hold NaN and -3 iterations.  diag_sums_nd reads no iteration counts, so `_solve_batched` must notice and
    re-solve those systems through the multi-launch iteration (reference: efgpnd.py:1634-1664, 1827-1838).  The test hook
    EFGP_COOP_TEST_DEAD kills every barrier of every cooperative launch."""
    from efgpnd import EFGPND
    from kernels.matern import Matern
    g = torch.Generator().manual_seed(5)
    N = 3000
    x = torch.rand(N, 2, dtype=torch.float64, generator=g)
    y = torch.sin(5 * x[:, 0]) * torch.cos(3 * x[:, 1]) + 0.1 * torch.randn(N, dtype=torch.float64, generator=g)
    kern = Matern(dimension=2, nu=2.5, init_lengthscale=0.1, init_variance=1.0)
    model = EFGPND(x.cuda(), y.cuda(), kern, sigmasq=0.1, eps=1e-3, nufft_eps=1e-8, estimate_params=False,
                   opts={"cg_tolerance": 1e-8})
    model.fit()
    xn = torch.rand(64, 2, dtype=torch.float64, generator=g).cuda()
    Mtot = int(model.last_fit_stats["feature_count"])
    assert tuple(model._toeplitz.fft_shape) == (128, 128)
    probes = (torch.randint(0, 2, (6, Mtot), generator=g) * 2 - 1).to(torch.float64).cuda()
    _, var_ok = model.predict(xn, variance_method="stochastic", variance_probes=probes)
    monkeypatch.setenv("EFGP_COOP_TEST_DEAD", "1")
    _, var_dead = model.predict(xn, variance_method="stochastic", variance_probes=probes, force_recompute=False)
    monkeypatch.delenv("EFGP_COOP_TEST_DEAD")
    assert torch.isfinite(var_dead).all()
    assert float((var_dead - var_ok).abs().max()) < 1e-6 * float(var_ok.abs().max())


def test_fit_survives_a_dead_grid_barrier(monkeypatch):
    """The fit's mean solve on a 128 x 128 circulant grid is ONE cooperative launch (efgp_cg_solve_mean_async: right-hand side,
    Jacobi diagonal and zero start formed in the kernel; reference: efgpnd.py:786-813).  A dead grid barrier leaves -3 and NaN;
    the fit reads the count once and re-solves through the multi-launch iteration: same coefficients, same iteration count."""
    from efgpnd import EFGPND
    from kernels.matern import Matern
    g = torch.Generator().manual_seed(5)
    N = 3000
    x = torch.rand(N, 2, dtype=torch.float64, generator=g)
    y = torch.sin(5 * x[:, 0]) * torch.cos(3 * x[:, 1]) + 0.1 * torch.randn(N, dtype=torch.float64, generator=g)

    def fit():
        kern = Matern(dimension=2, nu=2.5, init_lengthscale=0.1, init_variance=1.0)
        model = EFGPND(x.cuda(), y.cuda(), kern, sigmasq=0.1, eps=1e-3, nufft_eps=1e-8, estimate_params=False,
                       opts={"cg_tolerance": 1e-8, "mean_cg_warm_start": False})
        model.fit()
        assert tuple(model._toeplitz.fft_shape) == (128, 128)
        return model._beta.clone(), int(model.last_fit_stats["mean_cg_iters"])

    beta_ok, it_ok = fit()
    monkeypatch.setenv("EFGP_COOP_TEST_DEAD", "1")
    beta_dead, it_dead = fit()
    monkeypatch.delenv("EFGP_COOP_TEST_DEAD")
    assert torch.isfinite(torch.view_as_real(beta_dead)).all()
    assert abs(it_dead - it_ok) <= 1
    assert float((beta_dead - beta_ok).abs().max()) < 1e-7 * float(beta_ok.abs().max())
