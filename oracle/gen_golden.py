#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own code (build container only).

  python oracle/gen_golden.py [case ...]

Imports /root/reference/{efgpnd,cg,utils.kernels,kernels.*} unmodified, with
`oracle/standin/pytorch_finufft` (exact NUDFT) standing in for the absent
third-party FINUFFT binding, and writes `tests/golden/<case>.npz`.  The
reference never travels to the GPU box; the fixtures (inputs + outputs) do.

What a fixture holds (float64 / complex128): inputs or the generator seed that
reproduces them, effective hyper-parameters as the reference reads them back
(`kernel.get_hyper`, `model.sigmasq`), h, mtot, xis_1d, ws, conv vector v, F*y,
rhs, beta at CG tol 1e-12 (+ iteration counts at 1e-4 and 1e-12), posterior mean
at x_new, 'regular' variance, Hutchinson probes + lag sums + stochastic variance,
and the hyper-gradient with its probes Z, V (drawn by the reference from
torch.manual_seed(seed), re-drawn here in the same order to store them).
"""
import io
import contextlib
import math
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path.insert(0, os.path.join(HERE, "standin"))
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import efgpnd as ref  # noqa: E402  (the reference)
from kernels.squared_exponential import SquaredExponential  # noqa: E402
from kernels.matern import Matern  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(GOLD, "data")


def synth(N, d, seed):
    """Synthetic inputs of the reference's timing driver (test_timing_profiling.py:24-44)."""
    torch.manual_seed(seed)
    x = torch.rand(N, d, dtype=torch.float64) * 2 - 1
    if d == 1:
        f = torch.sin(3 * x[:, 0]) + 0.5 * torch.exp(-((x[:, 0] - 0.3) ** 2) / 0.3) + 0.7 * torch.sin(2 * math.pi * x[:, 0] ** 2)
    else:
        f = (torch.sin(3 * x[:, 0]) * torch.cos(4 * x[:, 1])
             + 0.5 * torch.exp(-((x[:, 0] - 0.3) ** 2 + (x[:, 1] + 0.3) ** 2) / 0.3)
             + 0.7 * torch.sin(2 * math.pi * (x[:, 0] ** 2 + x[:, 1] ** 2)))
        if d == 3:
            f = f * torch.cos(2 * x[:, 2])
    y = f + torch.randn(N, dtype=torch.float64) * math.sqrt(0.2)
    return x, y


def probe_points(x, n):
    """n deterministic prediction points inside the bounding box of x (golden-ratio lattice)."""
    d = x.shape[1]
    lo, hi = x.min(0).values, x.max(0).values
    i = torch.arange(1, n + 1, dtype=torch.float64)
    g = torch.tensor([0.6180339887498949, 0.7548776662466927, 0.8191725133961645][:d], dtype=torch.float64)
    u = torch.remainder(i[:, None] * g[None, :], 1.0)
    return lo + (0.02 + 0.96 * u) * (hi - lo)


def make_kernel(kind, d, ls, var, nu):
    if kind == "se":
        return SquaredExponential(dimension=d, init_lengthscale=ls, init_variance=var)
    return Matern(dimension=d, nu=nu, init_lengthscale=ls, init_variance=var)


def run_case(name, x, y, kind, ls, var, sig2, eps, *, nu=2.5, n_new=48, J=6, T=2, seed=1234,
             store_inputs=True, inputs_note="", do_regular=True):
    d = x.shape[1]
    N = x.shape[0]
    out = dict(kind=kind, d=d, N=N, nu=nu, eps=eps, seed=seed, inputs_note=inputs_note)
    if store_inputs:
        out["x"], out["y"] = x.numpy(), y.numpy()
    out["x_checksum"] = float(x.sum()); out["y_checksum"] = float(y.sum())

    def model(tol):
        k = make_kernel(kind, d, ls, var, nu)
        return EFGPND(x, y, k, sigmasq=sig2, eps=eps, nufft_eps=1e-12, estimate_params=False,
                      opts={"cg_tolerance": tol, "mean_cg_warm_start": False})

    EFGPND = ref.EFGPND
    m = model(1e-12)
    kern = m.kernel
    out["lengthscale"] = kern.get_hyper("lengthscale")
    out["variance"] = kern.get_hyper("variance")
    out["sigmasq"] = float(m.sigmasq.detach())
    m._compute_common_parameters()
    beta, ws, toep = m._beta, m._ws, m._toeplitz
    xis = m._xis
    h = xis.h_float
    M = ws.numel()
    mtot = round(M ** (1.0 / d))
    assert mtot ** d == M
    out.update(h=h, mtot=mtot, xis_1d=np.unique(xis[:, -1].numpy()), ws=ws.numpy(), beta=beta.numpy())
    # pieces of the fit, recomputed through the reference's own operators
    nu_op = ref.NUFFT(x, torch.zeros(d, dtype=torch.float64), torch.tensor(h, dtype=torch.float64), 1e-12)
    Fy = nu_op.type1(y, out_shape=(mtot,) * d).reshape(-1)
    v = ref.compute_convolution_vector_vectorized_dD((mtot - 1) // 2, x, torch.tensor(h, dtype=torch.float64))
    out.update(Fy=Fy.numpy(), v=v.numpy())
    # iteration counts of the reference CG at two tolerances, same operator
    A = ref.create_A_mean(ws, toep, out["sigmasq"], torch.complex128)
    center = tuple((s - 1) // 2 for s in v.shape)
    Minv = ref.create_jacobi_precond(ws, out["sigmasq"], diag_scale=v[center].real)
    for tol, tag in ((1e-4, "1e4"), (1e-12, "1e12")):
        cg = ref.ConjugateGradients(A, ws * Fy, torch.zeros_like(Fy), tol=tol, early_stopping=True, M_inv_apply=Minv)
        b = cg.solve()
        out["iters_" + tag] = cg.iters_completed
        if tag == "1e4":
            out["beta_1e4"] = b.numpy()
    # one Toeplitz apply for a known vector
    tv = torch.polar(torch.ones(M, dtype=torch.float64), torch.arange(M, dtype=torch.float64) * 0.37)
    out["toeplitz_in"], out["toeplitz_out"] = tv.numpy(), toep(tv).numpy()
    # predictions
    xn = probe_points(x, n_new)
    out["x_new"] = xn.numpy()
    sink = io.StringIO()
    if d < 3:
        mean, _ = m.predict(xn, return_variance=False)
    else:  # reference predict() mis-derives mtot in 3-D (efgpnd.py:908); go through its operators directly
        op = ref.NUFFT(xn, torch.zeros(d, dtype=torch.float64), torch.tensor(h, dtype=torch.float64), 1e-12)
        mean = op.type2(ws * beta, out_shape=(mtot,) * d).real
    out["mean"] = mean.numpy()
    Avar = ref.create_A_var(ws, toep, out["sigmasq"], torch.complex128)
    hT = torch.tensor(h, dtype=torch.float64)
    if do_regular:
        with contextlib.redirect_stdout(sink):
            var_reg = ref.compute_prediction_variance(
                x_new=xn[:16], xis=xis, ws=ws, A_var=Avar, cg_tol=1e-12, max_cg_iter=4000,
                variance_method="regular", h=hT, xcen=torch.zeros(d, dtype=torch.float64), hutchinson_probes=0,
                nufft_eps=1e-12, device=x.device, rdtype=torch.float64, cdtype=torch.complex128) if d < 3 else None
        if var_reg is not None:
            out["var_regular"] = var_reg.numpy()
    # Hutchinson lag sums with the probes the reference draws (efgpnd.py:1644)
    torch.manual_seed(seed)
    etas = (torch.randint(0, 2, (J, M)) * 2 - 1).to(torch.float64)
    torch.manual_seed(seed)
    c = ref.diag_sums_nd(Avar, J, xis, 4000, 1e-12, ws)
    out["etas"] = etas.numpy().astype(np.int8)
    out["lag_sums"] = c.numpy()
    out["var_stochastic"] = ref.nufft_var_est_nd(c, hT, torch.zeros(d, dtype=torch.float64), xn, 1e-12).numpy()
    # hyper-gradient with recorded probes (efgpnd.py:179-182, 199-202)
    torch.manual_seed(seed)
    Z = torch.empty((T, N), dtype=torch.float64).bernoulli_(0.5).mul_(2).sub_(1)
    V = torch.empty((T, M), dtype=torch.float64).bernoulli_(0.5).mul_(2).sub_(1)
    out["Z_bits"] = np.packbits((Z.numpy() > 0).astype(np.uint8), axis=1)
    out["V"] = V.numpy().astype(np.int8)
    mg = model(1e-12)
    torch.manual_seed(seed)
    stats = {}
    g = ref.efgpnd_gradient_batched(x, y, sigmasq=mg._gp_params.sig2.detach(), kernel=mg.kernel, eps=eps,
                                    trace_samples=T, x0=None, x1=None, nufft_eps=1e-12, cg_tol=1e-12,
                                    stats_out=stats)
    out["grad"] = g.detach().numpy()
    out["grad_mean_cg_iters"] = stats["mean_cg_iters"]
    out["grad_trace_cg_iters"] = stats["trace_cg_iters"]
    out["grad_beta"] = stats["mean_beta"].numpy()
    os.makedirs(GOLD, exist_ok=True)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print(f"{name}: N={N} d={d} h={h:.6g} mtot={mtot} M={M} iters(1e-4)={out['iters_1e4']} "
          f"iters(1e-12)={out['iters_1e12']} grad={out['grad']}")


def load_pair(fn):
    x, y = torch.load(os.path.join(REF, "data", fn))
    x = x.to(torch.float64)
    if x.ndim == 1:
        x = x[:, None]
    return x, y.to(torch.float64)


def usa_temp():
    dct = torch.load(os.path.join(REF, "data", "usa_temp_data.pt"))
    x = dct["x"].to(torch.float64); y = dct["y"].to(torch.float64)
    # normalisation of verify_efgpnd_exact_small.py:57-63
    x = (x - x.min(0).values) / (x.max(0).values - x.min(0).values)
    y = (y - y.mean()) / y.std()
    return x, y


CASES = {}


def case(fn):
    CASES[fn.__name__] = fn
    return fn


@case
def c1_se1d_n5000():
    x, y = load_pair("gp_samples_5000_0.1_2_0.1.pt")
    run_case("c1_se1d_n5000", x, y, "se", 0.1, 2.0, 0.1, 1e-4, inputs_note="data/gp_samples_5000_0.1_2_0.1.pt")


@case
def c2_se2d_n100000():
    x, y = load_pair("gp_samples_100000_0.2_2_0.2.pt")
    os.makedirs(DATA, exist_ok=True)
    np.savez(os.path.join(DATA, "gp_samples_100000_0.2_2_0.2.npz"), x=x.numpy(), y=y.numpy())
    run_case("c2_se2d_n100000", x, y, "se", 0.2, 2.0, 0.2, 1e-4, store_inputs=False,
             inputs_note="tests/golden/data/gp_samples_100000_0.2_2_0.2.npz (= reference data/gp_samples_100000_0.2_2_0.2.pt)")


@case
def c3_matern52_usatemp():
    x, y = usa_temp()
    run_case("c3_matern52_usatemp", x, y, "matern", 0.1, 1.0, 0.05, 1e-3, nu=2.5,
             inputs_note="data/usa_temp_data.pt, x min-max normalised, y standardised")


@case
def c4_se2d_hard_n100000():
    x, y = synth(100000, 2, 0)
    run_case("c4_se2d_hard_n100000", x, y, "se", 0.05, 3.0, 0.2, 1e-4, store_inputs=False, do_regular=False,
             inputs_note="synth(N=100000,d=2,seed=0) of oracle/gen_golden.py (torch CPU generator)")


@case
def c5_matern32_3d_n20000():
    x, y = synth(20000, 3, 1)
    run_case("c5_matern32_3d_n20000", x, y, "matern", 0.3, 1.5, 0.2, 1e-2, nu=1.5, store_inputs=False, J=4,
             inputs_note="synth(N=20000,d=3,seed=1) of oracle/gen_golden.py (torch CPU generator)")


@case
def c5b_matern32_3d_l02_n20000():
    # SURVEY's C5 hyper-parameters (3-D Matern-3/2, l = 0.2): mtot = 29 at eps = 1e-2 (the c5 case above uses l = 0.3 -> mtot = 19)
    x, y = synth(20000, 3, 1)
    run_case("c5b_matern32_3d_l02_n20000", x, y, "matern", 0.2, 1.5, 0.2, 1e-2, nu=1.5, store_inputs=False, J=4,
             inputs_note="synth(N=20000,d=3,seed=1) of oracle/gen_golden.py (torch CPU generator)")


@case
def s1_se2d_n100():
    x, y = load_pair("gp_samples_100_0.5_2_0.2.pt")
    run_case("s1_se2d_n100", x, y, "se", 0.5, 2.0, 0.2, 1e-5, inputs_note="data/gp_samples_100_0.5_2_0.2.pt")


@case
def s2_matern12_1d_n200():
    x, y = load_pair("gp_samples_200_0.1_2_0.1.pt")
    run_case("s2_matern12_1d_n200", x, y, "matern", 0.3, 1.2, 0.1, 1e-2, nu=0.5,
             inputs_note="data/gp_samples_200_0.1_2_0.1.pt")


if __name__ == "__main__":
    torch.set_num_threads(8)
    names = sys.argv[1:] or list(CASES)
    for n in names:
        CASES[n]()
