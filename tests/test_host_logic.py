"""Host-side mirror of the reference interface: kernels, GPParams, get_xis, CG generic path,
model construction and error behaviour.  No GPU needed."""
import math

import numpy as np
import pytest
import torch

from _golden import load_case, oracle_kernel
from oracle import efgp_oracle as O


def test_import_surface():
    import efgpnd
    for name in ["EFGPND", "efgpnd_gradient_batched", "efgp_nd", "NUFFT", "ToeplitzND",
                 "compute_convolution_vector_vectorized_dD", "_cmplx", "create_A_mean", "create_A_var", "create_Gv",
                 "create_jacobi_precond", "setup_operators", "setup_nufft", "get_xis", "diag_sums_nd",
                 "nufft_var_est_nd", "compute_prediction_variance", "logdet_slq", "ConjugateGradients", "GPParams"]:
        assert hasattr(efgpnd, name), name
    from kernels import Kernel, Matern, SquaredExponential, GPParams  # noqa: F401
    from kernels.squared_exponential import SquaredExponential as SE2  # noqa: F401
    from kernels.matern import Matern as M2  # noqa: F401
    from utils.kernels import get_xis, GetTruncationBound  # noqa: F401
    from cg import ConjugateGradients  # noqa: F401


def test_kernel_values_against_oracle():
    from kernels.squared_exponential import SquaredExponential
    from kernels.matern import Matern
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        xi = torch.tensor(np.random.default_rng(0).normal(size=(50, 2)))
        r = torch.linspace(0, 3, 40, dtype=torch.float64)
        for k, spec in [(SquaredExponential(dimension=2, init_lengthscale=0.3, init_variance=1.7), O.KernelSpec("se", 2, 0.3, 1.7)),
                        (Matern(dimension=2, nu=0.5, init_lengthscale=0.3, init_variance=1.7), O.KernelSpec("matern", 2, 0.3, 1.7, 0.5)),
                        (Matern(dimension=2, nu=1.5, init_lengthscale=0.3, init_variance=1.7), O.KernelSpec("matern", 2, 0.3, 1.7, 1.5)),
                        (Matern(dimension=2, nu=2.5, init_lengthscale=0.3, init_variance=1.7), O.KernelSpec("matern", 2, 0.3, 1.7, 2.5))]:
            spec.lengthscale, spec.variance = k.get_hyper("lengthscale"), k.get_hyper("variance")
            assert np.allclose(k.kernel(r).numpy(), spec.k(r.numpy()), rtol=1e-14)
            assert np.allclose(k.spectral_density(xi).numpy(), spec.S(xi.numpy()), rtol=1e-13)
            assert np.allclose(k.spectral_grad(xi).numpy(), spec.dS(xi.numpy()), rtol=1e-12)
            # spectral_grad is the derivative of spectral_density (finite differences)
            e = 1e-6
            l0 = k.get_hyper("lengthscale")
            k._gp_params_ref.raw.data[0] = math.log(l0 + e)
            S1 = k.spectral_density(xi)
            k._gp_params_ref.raw.data[0] = math.log(l0 - e)
            S0 = k.spectral_density(xi)
            k._gp_params_ref.raw.data[0] = math.log(l0)
            fd = (S1 - S0) / (2 * e)
            assert torch.allclose(fd, k.spectral_grad(xi)[:, 0], rtol=1e-5, atol=1e-7)
    finally:
        torch.set_default_dtype(prev)


def test_hyper_storage_quirks():
    """log-space storage in the default dtype; set_hyper rounds through float32 (reference kernel.py:137)."""
    from kernels.squared_exponential import SquaredExponential
    k = SquaredExponential(dimension=2, init_lengthscale=0.2, init_variance=2.0)
    assert k.num_hypers == 3 and k.hypers == ["lengthscale", "variance"]
    assert k.get_hyper("lengthscale") == pytest.approx(0.2, rel=1e-6)
    assert k.get_hyper("lengthscale") != 0.2            # float32 GPParams outside a float64 model
    assert dict(k.iter_hypers())["lengthscale"] == 0.2
    with pytest.raises(ValueError):
        k.set_hyper("nope", 1.0)
    with pytest.raises(ValueError):
        k.get_hyper("nope")
    with pytest.raises(ValueError):
        SquaredExponential(dimension=0)
    with pytest.raises(ValueError):
        SquaredExponential(dimension=1, init_lengthscale=1e-9)
    k.lengthscale = 0.14
    assert k.lengthscale == pytest.approx(0.14, rel=1e-6)


@pytest.mark.parametrize("name", ["s1_se2d_n100", "s2_matern12_1d_n200", "c1_se1d_n5000", "c2_se2d_n100000",
                                  "c3_matern52_usatemp", "c4_se2d_hard_n100000", "c5_matern32_3d_n20000"])
def test_get_xis_matches_reference(name):
    """get_xis inside a float64 model reproduces the reference's h and mtot (golden)."""
    from efgpnd import EFGPND
    from kernels.squared_exponential import SquaredExponential
    from kernels.matern import Matern
    from utils.kernels import get_xis
    from test_gpu_efgp_golden import NOMINAL
    g, x, y = load_case(name)
    kind, ls, var, sig2, nu = NOMINAL[name]
    d = x.shape[1]
    k = SquaredExponential(dimension=d, init_lengthscale=ls, init_variance=var) if kind == "se" else \
        Matern(dimension=d, nu=nu, init_lengthscale=ls, init_variance=var)
    m = EFGPND(x, y, k, sigmasq=sig2, eps=float(g["eps"]), estimate_params=False)
    assert k.get_hyper("lengthscale") == pytest.approx(float(g["lengthscale"]), abs=1e-15)
    assert k.get_hyper("variance") == pytest.approx(float(g["variance"]), abs=1e-15)
    assert float(m.sigmasq.detach()) == pytest.approx(float(g["sigmasq"]), abs=1e-15)
    L = float((x.max(0).values - x.min(0).values).max())
    xis, h, mtot = get_xis(k, float(g["eps"]), L, use_integral=True)
    assert mtot == int(g["mtot"])
    assert h == pytest.approx(float(g["h"]), rel=1e-13)
    assert np.allclose(xis.numpy(), g["xis_1d"], rtol=1e-13, atol=1e-15)
    # generic tensor path (any object with kernel()/spectral_density()) gives the same grid
    class Wrapped:
        dimension = d
        kernel = staticmethod(k.kernel)
        spectral_density = staticmethod(k.spectral_density)
    xis2, h2, mtot2 = get_xis(Wrapped(), float(g["eps"]), L, use_integral=True)
    assert mtot2 == mtot and h2 == pytest.approx(h, rel=1e-12)


def test_get_xis_heuristic_branches():
    from kernels.squared_exponential import SquaredExponential
    from kernels.matern import Matern
    from utils.kernels import get_xis
    for k in (SquaredExponential(dimension=2, init_lengthscale=0.3, init_variance=1.0),
              Matern(dimension=1, nu=1.5, init_lengthscale=0.3, init_variance=1.0)):
        for l2 in (False, True):
            xis, h, mtot = get_xis(k, 1e-3, 2.0, use_integral=False, l2scaled=l2)
            assert mtot % 2 == 1 and xis.numel() == mtot and h > 0
            assert xis[mtot // 2].item() == 0.0


def test_model_construction_and_errors():
    from efgpnd import EFGPND
    x = torch.rand(50, 2, dtype=torch.float64)
    y = torch.rand(50, dtype=torch.float64)
    for name, nu in [("SquaredExponential", None), ("se", None), ("SE", None), ("Matern12", 0.5), ("matern32", 1.5), ("MATERN52", 2.5)]:
        m = EFGPND(x, y, name, eps=1e-3)
        if nu is not None:
            assert m.kernel.nu == nu
        assert len(list(m.parameters())) == 1 and list(m.parameters())[0].shape == (3,)
        assert list(m.parameters())[0].dtype == torch.float64
    with pytest.raises(ValueError):
        EFGPND(x, y, "Periodic")
    m = EFGPND(x, y, "se", sigmasq=None, estimate_params=False)
    assert float(m.sigmasq.detach()) == pytest.approx(0.1)
    opt = torch.optim.Adam(m.parameters(), lr=0.1)
    assert m.register_optimizer(opt) is opt and m.register_optimizer(opt) is opt
    with pytest.raises(ValueError):
        m.predict(None)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_path_fails_loudly_without_gpu():
    """No CPU fallback: every hot entry point raises instead of silently computing elsewhere."""
    from efgpnd import EFGPND, NUFFT, ToeplitzND, compute_convolution_vector_vectorized_dD
    x = torch.rand(50, 2, dtype=torch.float64)
    y = torch.rand(50, dtype=torch.float64)
    m = EFGPND(x, y, "se", sigmasq=0.1, estimate_params=False)
    with pytest.raises(RuntimeError, match="GPU"):
        m.predict(x)
    with pytest.raises(RuntimeError, match="GPU"):
        m.compute_gradients(trace_samples=2)
    with pytest.raises(RuntimeError, match="GPU"):
        NUFFT(x, torch.zeros(2, dtype=torch.float64), 0.3, 1e-6)
    with pytest.raises(RuntimeError, match="GPU"):
        ToeplitzND(torch.ones(5, 5, dtype=torch.complex128))
    with pytest.raises(RuntimeError, match="GPU"):
        compute_convolution_vector_vectorized_dD(2, x, 0.3)


def test_generic_cg_matches_oracle():
    """The compatibility loop for arbitrary operators follows cg.py:86-244 (checked via the oracle)."""
    from cg import ConjugateGradients
    g = torch.Generator().manual_seed(0)
    n = 40
    Q = torch.complex(torch.randn(n, n, generator=g, dtype=torch.float64), torch.randn(n, n, generator=g, dtype=torch.float64))
    A = Q @ Q.conj().T + 0.5 * torch.eye(n, dtype=torch.complex128)
    b = torch.complex(torch.randn(n, generator=g, dtype=torch.float64), torch.randn(n, generator=g, dtype=torch.float64))
    diag = A.diagonal().real.clone()
    cgs = ConjugateGradients(A, b, torch.zeros_like(b), tol=1e-10, M_inv_apply=lambda r: r / diag)
    xs = cgs.solve()
    xo, ito = O.cg_single(lambda v: A @ v, b, torch.zeros_like(b), 1e-10, diag=diag)
    assert cgs.iters_completed == ito and torch.allclose(xs, xo, rtol=1e-12, atol=1e-14)
    B = torch.stack([b, 2 * b.conj(), torch.zeros_like(b)])
    cgb = ConjugateGradients(lambda V: V @ A.T, B, torch.zeros_like(B), tol=1e-9, max_iter=500)
    xb = cgb.solve()
    xob, itob = O.cg_batched(lambda V: V @ A.T, B, torch.zeros_like(B), 1e-9, max_iter=500)
    assert cgb.iters_completed == itob and torch.allclose(xb, xob, rtol=1e-12, atol=1e-14)
    with pytest.raises(ValueError):
        ConjugateGradients(3.0, b, b)


def test_cpu_quota_parsing(tmp_path):
    """efgp_hip.cpu_quota reads the CFS bandwidth limit (cgroup v2 cpu.max, v1 cfs files); 'max' / absent = unlimited."""
    from efgp_hip import cpu_quota
    assert cpu_quota.cpu_quota_cores(str(tmp_path)) is None
    (tmp_path / "cpu.max").write_text("1600000 100000\n")
    assert cpu_quota.cpu_quota_cores(str(tmp_path)) == 16.0
    (tmp_path / "cpu.max").write_text("max 100000\n")
    assert cpu_quota.cpu_quota_cores(str(tmp_path)) is None
    v1 = tmp_path / "v1"
    (v1 / "cpu").mkdir(parents=True)
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("250000\n")
    (v1 / "cpu" / "cpu.cfs_period_us").write_text("100000\n")
    assert cpu_quota.cpu_quota_cores(str(v1)) == 2.5


def test_native_grid_bounds_are_bit_identical(monkeypatch):
    """efgp_grid_bounds runs the two bisections of get_xis (utils/kernels.py:28-69, 94-105) in C with the operations of the
    Python expressions in the same order: h, mtot and the nodes must be IDENTICAL over kernels, dimensions, tolerances."""
    import random
    from utils.kernels import get_xis
    from kernels.squared_exponential import SquaredExponential
    from kernels.matern import Matern
    rnd = random.Random(11)
    for trial in range(400):
        d = rnd.choice([1, 2, 3])
        ell, var = 10 ** rnd.uniform(-1.5, 0.5), 10 ** rnd.uniform(-2, 2)
        eps, L = 10 ** rnd.uniform(-9, -1), rnd.uniform(0.3, 6)
        if trial % 2:
            k = SquaredExponential(dimension=d, init_lengthscale=ell, init_variance=var)
        else:
            k = Matern(dimension=d, nu=rnd.choice([0.5, 1.5, 2.5]), init_lengthscale=ell, init_variance=var)
        monkeypatch.setenv("EFGP_NO_NATIVE_GRID", "1")
        ref = get_xis(kernel_obj=k, eps=eps, L=L, use_integral=True)
        monkeypatch.delenv("EFGP_NO_NATIVE_GRID")
        got = get_xis(kernel_obj=k, eps=eps, L=L, use_integral=True)
        assert got[1] == ref[1] and got[2] == ref[2] and torch.equal(got[0], ref[0]), (trial, d, ell, var, eps, L)


def test_native_spectral_weights_match_the_kernel_classes():
    """efgp_spectral_weights_host against kernel.spectral_density / spectral_grad on the tensor grid (efgpnd.py:766-780):
    equal to rounding (libm vs torch's vectorised exp / pow)."""
    import ctypes as C
    from efgp_hip.lib import lib
    from utils.kernels import kernel_constants
    from kernels.squared_exponential import SquaredExponential
    from kernels.matern import Matern
    for k, h, mtot in ((SquaredExponential(dimension=2, init_lengthscale=0.2, init_variance=2.0), 0.346, 23),
                       (Matern(dimension=3, nu=1.5, init_lengthscale=0.5, init_variance=1.3), 0.21, 9),
                       (Matern(dimension=1, nu=2.5, init_lengthscale=0.1, init_variance=0.7), 0.4, 41),
                       (SquaredExponential(dimension=3, init_lengthscale=0.4, init_variance=0.5), 0.3, 7)):
        d = k.dimension
        ell, var = k.get_hyper("lengthscale"), k.get_hyper("variance")
        kind, nu, c0 = kernel_constants(k, ell, var)
        M = mtot ** d
        ws = torch.empty(M, dtype=torch.complex128)
        dp = torch.empty((M, 2), dtype=torch.complex128)
        assert lib().efgp_spectral_weights_host(kind, d, nu, ell, var, c0, h, mtot, ws.data_ptr(), dp.data_ptr()) == 0
        x1 = torch.arange(-(mtot // 2), mtot // 2 + 1, dtype=torch.float64) * h
        xis = torch.stack(torch.meshgrid(*(x1 for _ in range(d)), indexing="ij"), dim=-1).view(-1, d)
        ws_ref = torch.sqrt(k.spectral_density(xis).to(torch.complex128) * h ** d)
        dp_ref = (h ** d * k.spectral_grad(xis)).to(torch.complex128)
        assert float((ws - ws_ref).abs().max() / ws_ref.abs().max()) < 1e-14
        assert float((dp - dp_ref).abs().max() / dp_ref.abs().max()) < 1e-13
        assert float(ws.imag.abs().max()) == 0.0


def test_fine_grid_size_always_returns():
    """es_fine_size's ladder search must terminate for every tolerance / mode count / dimension (the width model is not
    monotone in the upsampling ratio beyond its calibrated range: tol ~ 0.035, 40 modes has W = 2 only on the dense
    100-cell grid, never on a ladder size -- the search used to double until overflow).  Sweep incl. that band; the result
    is a 2^a 3^b 5^c size >= 2 n whose window is no wider than the ladder's first candidate would need."""
    from efgp_hip.lib import lib
    L = lib()
    tols = [10 ** (-12 + 11.7 * i / 119) for i in range(120)] + [0.0346 + 0.0025 * i / 39 for i in range(40)]
    for dim in (1, 2, 3):
        for dense in (0, 1):
            for n in range(1, 257):
                for tol in tols:
                    c = L.efgp_fine_grid_size_nd(n, tol, dim, dense)
                    assert c >= 2 * n and c >= 32, (n, tol, dim, dense, c)
                    m = c
                    for p in (2, 3, 5):
                        while m % p == 0:
                            m //= p
                    assert m == 1, (n, tol, dim, dense, c)
    # the width a plan of dimension d uses never shrinks with d (the errors of the axes add up) and never with the tolerance
    for sig in (2.0, 2.5, 3.1, 4.0, 7.0):
        for tol in (1e-3, 1e-6, 1e-9, 1e-12):
            ws_ = [L.efgp_window_width_nd(tol, sig, d) for d in (1, 2, 3)]
            assert ws_[0] <= ws_[1] <= ws_[2] and ws_[0] == L.efgp_window_width(tol, sig)
            assert L.efgp_window_width_nd(tol / 10, sig, 2) >= ws_[1]


def test_window_design_matches_round3_outputs():
    """Round 4 took the transcendental-heavy parts out of the host-side window design (a training step with a new mode count
    paid 0.2-0.4 ms per window set): cosine tables shared by the cells, reference values hoisted out of the degree search,
    deconvolution factors by Reinsch's cosine recurrence instead of one cosl per (mode, node).  Against the round-3 library's
    outputs (tests/golden/window_design_r3.npz, written before the change): widths, shape parameters and window polynomial
    values bit-identical, correction factors within one ulp.  Run in a child process under EFGP_WIDTH_MODEL_R3=1: round 4 also
    changed WHICH width a tolerance gets (the error model of es_kernel.cpp); the arithmetic behind a given width is what is
    pinned here."""
    import subprocess
    import sys
    code = r"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from efgp_hip.lib import lib
L = lib()
g = np.load(sys.argv[2])
first, vals, w, beta = C.c_int64(), (C.c_double * 16)(), C.c_int(), C.c_double()
for i, (tol, nf, n) in enumerate(g["cases"]):
    nf, n = int(nf), int(n)
    out = (C.c_double * n)()
    assert L.efgp_window_deconv(float(tol), nf, n, out) == 0
    a, b = np.array(out[:]), g[f"deconv_{i}"]
    assert float((np.abs(a - b) / np.spacing(np.abs(b))).max()) <= 1.0, (tol, nf, n)
    for r, X in enumerate((10.3, 7.77, 21.5, 3.999999)):
        assert L.efgp_window_eval(float(tol), nf / n, X, C.byref(first), vals, C.byref(w), C.byref(beta)) == 0
        assert w.value == int(g[f"w_{i}"][0]) and beta.value == float(g[f"beta_{i}"][0]), (tol, nf, n, w.value)
        assert np.array_equal(np.array(vals[:w.value]), g[f"eval_{i}"][r]), (tol, nf, n, X)
print("ok")
"""
    import os
    from _golden import GOLDEN
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code, os.path.join(root, "gp-quadrature_amd"), f"{GOLDEN}/window_design_r3.npz"],
                         env=dict(os.environ, EFGP_WIDTH_MODEL_R3="1"), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr
