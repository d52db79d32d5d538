"""GPU parity of the whole EFGP path (through efgpnd.EFGPND -> C ABI -> HIP kernels) against golden
vectors produced by the reference's own code (oracle/gen_golden.py).

Tolerance: north_star asks posterior mean / variance within 1e-5 relative (float64).  Quantities
that feed the solve (F*y, Toeplitz vector) are checked to the NUFFT tolerance requested here (1e-9)."""
import os

import numpy as np
import pytest
import torch

from _golden import GOLDEN, load_case, rel

pytestmark = pytest.mark.gpu

NOMINAL = {"c1_se1d_n5000": ("se", 0.1, 2.0, 0.1, 2.5), "c2_se2d_n100000": ("se", 0.2, 2.0, 0.2, 2.5),
           "c3_matern52_usatemp": ("matern", 0.1, 1.0, 0.05, 2.5), "c4_se2d_hard_n100000": ("se", 0.05, 3.0, 0.2, 2.5),
           "c5_matern32_3d_n20000": ("matern", 0.3, 1.5, 0.2, 1.5), "c5b_matern32_3d_l02_n20000": ("matern", 0.2, 1.5, 0.2, 1.5),
           "s1_se2d_n100": ("se", 0.5, 2.0, 0.2, 2.5),
           "s2_matern12_1d_n200": ("matern", 0.3, 1.2, 0.1, 0.5)}


def make_model(name, g, x, y, cg_tol, nufft_eps=1e-9, **opts):
    from efgpnd import EFGPND
    from kernels.squared_exponential import SquaredExponential
    from kernels.matern import Matern
    kind, ls, var, sig2, nu = NOMINAL[name]
    d = x.shape[1]
    k = SquaredExponential(dimension=d, init_lengthscale=ls, init_variance=var) if kind == "se" else \
        Matern(dimension=d, nu=nu, init_lengthscale=ls, init_variance=var)
    o = {"cg_tolerance": cg_tol, "mean_cg_warm_start": False}
    o.update(opts)
    m = EFGPND(x, y, k, sigmasq=sig2, eps=float(g["eps"]), nufft_eps=nufft_eps, estimate_params=False, opts=o)
    # effective hyper-parameters equal what the reference read back
    assert abs(k.get_hyper("lengthscale") - float(g["lengthscale"])) < 1e-15
    assert abs(float(m.sigmasq.detach()) - float(g["sigmasq"])) < 1e-15
    return m


CASES = ["s1_se2d_n100", "s2_matern12_1d_n200", "c1_se1d_n5000", "c2_se2d_n100000", "c3_matern52_usatemp",
         "c4_se2d_hard_n100000", "c5_matern32_3d_n20000", "c5b_matern32_3d_l02_n20000"]


@pytest.mark.parametrize("name", CASES)
def test_fit_pieces_and_mean(name):
    g, x, y = load_case(name)
    m = make_model(name, g, x, y, 1e-12)
    m.fit()
    st = m._fit_state
    assert st["mtot"] == int(g["mtot"])
    assert abs(st["h"] - float(g["h"])) < 1e-12 * float(g["h"])
    assert rel(st["ws"], g["ws"]) < 1e-12
    assert rel(st["Fy"], g["Fy"]) < 5e-8
    assert rel(st["v"], g["v"]) < 5e-8
    # Toeplitz apply on the golden probe vector
    tv = torch.from_numpy(g["toeplitz_in"]).cuda()
    assert rel(m._toeplitz(tv), g["toeplitz_out"]) < 5e-8
    # posterior mean at the golden prediction points: 1e-5 relative (north_star)
    xn = torch.from_numpy(g["x_new"])
    mean, var = m.predict(xn, return_variance=False)
    assert mean.device == x.device and mean.dtype == x.dtype
    assert torch.isnan(var).all()
    assert rel(mean, g["mean"]) < 1e-5


@pytest.mark.parametrize("name", ["s1_se2d_n100", "s2_matern12_1d_n200", "c1_se1d_n5000", "c2_se2d_n100000",
                                  "c3_matern52_usatemp", "c4_se2d_hard_n100000", "c5_matern32_3d_n20000"])
def test_default_tolerance_iterations(name):
    """CG at the reference's default tolerance 1e-4: same iteration count (+-1 for long solves)."""
    g, x, y = load_case(name)
    m = make_model(name, g, x, y, 1e-4)
    m.fit()
    it = m.last_fit_stats["mean_cg_iters"]
    ref_it = int(g["iters_1e4"])
    # Round 3: the band comes from the REFERENCE's own behaviour, stored in the round-3 fixtures (oracle/gen_golden_r3.py): under a
    # 1e-13 perturbation of its Toeplitz vector the reference itself stops at 195 instead of 175 on c3 (|r|/|b| dips below 1e-4 at
    # iterations 175, 178, 182, 195, ...), at 152 instead of 150 on c2, and does not move on c1.  Equal counts elsewhere
    # (+-1 beyond 100 iterations where no such fixture exists); always the defining property: the iterate meets the tolerance.
    spread, beta_bound = 0, 5e-2
    r3_path = os.path.join(GOLDEN, name + "_r3.npz")
    if os.path.exists(r3_path):
        r3 = np.load(r3_path)
        spread = abs(int(r3["wfit_iters_1e4"][0]) - int(r3["sens_wfit_iters_1e4"][0]))
        # beta: 10 x the reference's own move (Toeplitz-vector perturbation, NUFFT results accurate to 1e-11), at most the old 5e-2
        beta_bound = min(5e-2, 10.0 * max(float(r3["sens_wfit_beta1_1e4"]), float(r3["sens_wfit_nufft_beta1_1e4"])))
        assert abs(it - ref_it) <= spread + (max(1, int(0.02 * ref_it)) if spread else 0), (name, it, ref_it, spread)
    else:
        assert abs(it - ref_it) <= (1 if ref_it > 100 else 0), (name, it, ref_it)
    from efgpnd import create_A_mean
    st = m._fit_state
    A = create_A_mean(st["ws"], m._toeplitz, st["sig"], torch.complex128)
    rhs = st["ws"] * st["Fy"]
    res = float(torch.linalg.norm(rhs - A(m._beta)) / torch.linalg.norm(rhs))
    assert res < 1.05e-4
    # at the loose default tolerance the iterate moves by ~tol*cond when the stop flips by one pass (SURVEY section 7 "hard parts")
    dev_beta = rel(m._fit_state["ws"] * m._beta, torch.from_numpy(g["ws"]) * torch.from_numpy(g["beta_1e4"]))
    print(f"\n{name}: iterations hip={it} ref={ref_it} (reference's own spread {spread}); ws*beta deviates by {dev_beta:.2e} (bound {beta_bound:.1e})")
    assert dev_beta < beta_bound


@pytest.mark.parametrize("name", ["s1_se2d_n100", "s2_matern12_1d_n200", "c1_se1d_n5000", "c2_se2d_n100000",
                                  "c3_matern52_usatemp"])
def test_variance_regular(name):
    g, x, y = load_case(name)
    m = make_model(name, g, x, y, 1e-12, max_cg_iterations=4000)
    xn = torch.from_numpy(g["x_new"])[:16]
    _, var = m.predict(xn, variance_method="regular")
    assert rel(var, g["var_regular"]) < 1e-5


@pytest.mark.parametrize("name", CASES)
def test_variance_stochastic_with_recorded_probes(name):
    g, x, y = load_case(name)
    m = make_model(name, g, x, y, 1e-12, max_cg_iterations=4000)
    xn = torch.from_numpy(g["x_new"])
    etas = torch.from_numpy(g["etas"].astype(np.float64))
    _, var = m.predict(xn, variance_method="stochastic", hutchinson_probes=etas.shape[0], variance_probes=etas)
    scale = float(np.abs(g["var_stochastic"]).max())
    assert float(np.abs(var.cpu().numpy() - g["var_stochastic"]).max()) < 1e-5 * scale


@pytest.mark.parametrize("mode", ["adjoint", "reference"])
@pytest.mark.parametrize("name", CASES)
def test_gradient_with_recorded_probes(name, mode):
    """efgpnd_gradient_batched with the reference's own Rademacher draws (efgpnd.py:179-182,199-202):
    the literal operation sequence ("reference") and the adjoint-identity evaluation (default) both match."""
    g, x, y = load_case(name)
    m = make_model(name, g, x, y, 1e-12)
    Z = g["Z"]
    V = torch.from_numpy(g["V"].astype(np.float64))
    raw = m.compute_gradients(trace_samples=V.shape[0], nufft_eps=1e-9, cg_tol=1e-12, probes_Z=Z, probes_V=V,
                              trace_mode=mode)
    pos = torch.tensor([float(g["lengthscale"]), float(g["variance"]), float(g["sigmasq"])], dtype=torch.float64)
    grad = raw.detach().cpu() / pos
    ref = torch.from_numpy(g["grad"])
    # the gradient is a difference of large terms: compare against the size of the terms
    scale = float(m.last_gradient_stats["term1"].abs().max())
    assert float((grad - ref).abs().max()) < 1e-5 * scale
    assert m._gp_params.raw.grad is not None


def test_gradient_in_kernel_probes_match_materialised_probes():
    """Probes generated inside the spread kernel (default path) == the same probes written to memory."""
    from efgp_hip import rademacher_fill
    name = "c2_se2d_n100000"
    g, x, y = load_case(name)
    m = make_model(name, g, x, y, 1e-10)
    T = 3
    V = torch.from_numpy(g["V"].astype(np.float64))[:1].repeat(T, 1)
    g1 = m.compute_gradients(trace_samples=T, nufft_eps=1e-9, cg_tol=1e-10, probes_V=V, probe_seed=12345)
    Z = rademacher_fill(torch.device("cuda", 0), 12345, T, x.shape[0])
    assert set(torch.unique(Z).tolist()) == {-1.0, 1.0}
    assert abs(float(Z.mean())) < 0.01 and abs(float((Z[0] * Z[1]).mean())) < 0.02
    m._last_gradient_beta = None
    g2 = m.compute_gradients(trace_samples=T, nufft_eps=1e-9, cg_tol=1e-10, probes_V=V, probes_Z=Z, trace_mode="reference")
    scale = float(m.last_gradient_stats["term1"].abs().max()) * float(m._gp_params.pos.detach().max())
    assert float((g1 - g2).abs().max()) < 1e-6 * scale
