#!/usr/bin/env python3
"""How far do the REFERENCE's own CG residual curve and SLQ log-determinant move under a rounding-level perturbation of
their inputs?  (build container only; prints numbers quoted in tests/test_gpu_r2_goldens.py)

For each case the reference's operators are built twice: as they are, and with the Toeplitz vector v multiplied by
(1 + 1e-13 * standard normal) -- smaller than the difference between two correct NUFFTs at tolerance 1e-12.  Printed:
max relative deviation of |r_i|/|b| between the two runs over the first 20 iterations / first half / all iterations, and
the relative change of logdet_slq with the same probes."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402
import gen_golden_r2 as R  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402

ref = G.ref


def study(name, x, y, kind, ls, var, sig2, eps, nu=2.5, seed=1234):
    d = x.shape[1]
    N = x.shape[0]
    k = G.make_kernel(kind, d, ls, var, nu)
    m = ref.EFGPND(x, y, k, sigmasq=sig2, eps=eps, nufft_eps=1e-12, estimate_params=False,
                   opts={"cg_tolerance": 1e-12, "mean_cg_warm_start": False})
    m._compute_common_parameters()
    ws, xis = m._ws, m._xis
    h = xis.h_float
    M = ws.numel()
    mtot = round(M ** (1.0 / d))
    sig = float(m.sigmasq.detach())
    nu_op = ref.NUFFT(x, torch.zeros(d, dtype=torch.float64), torch.tensor(h, dtype=torch.float64), 1e-12)
    Fy = nu_op.type1(y, out_shape=(mtot,) * d).reshape(-1)
    v = ref.compute_convolution_vector_vectorized_dD((mtot - 1) // 2, x, torch.tensor(h, dtype=torch.float64))
    g = torch.Generator().manual_seed(5)
    res = []
    for pert in (0.0, 1e-13):
        vv = v * (1.0 + pert * torch.randn(v.shape, generator=g, dtype=torch.float64))
        toep = ref.ToeplitzND(vv, force_pow2=True)
        A = ref.create_A_mean(ws, toep, sig, torch.complex128)
        center = tuple((s - 1) // 2 for s in v.shape)
        Minv = ref.create_jacobi_precond(ws, sig, diag_scale=vv[center].real)
        cg = ref.ConjugateGradients(A, ws * Fy, torch.zeros_like(Fy), tol=1e-4, early_stopping=True, M_inv_apply=Minv)
        with R.NormRecorder() as rec:
            cg.solve()
        hist = np.array(rec.vals[1:]) / (rec.vals[0] + 1e-16)
        torch.manual_seed(seed)
        ld = ref.logdet_slq(ws, sig, toep, probes=6, steps=25, dtype=torch.float64, device="cpu", n=N)
        res.append((hist, float(ld), cg.iters_completed))
    (h0, l0, i0), (h1, l1, i1) = res
    kk = min(len(h0), len(h1))
    dev = np.abs(h0[:kk] - h1[:kk]) / h0[:kk]
    print(f"{name}: iters {i0} vs {i1}; residual-curve deviation first20={dev[:20].max():.2e} half={dev[:kk // 2].max():.2e} "
          f"all={dev.max():.2e};  logdet {l0:.6f} vs {l1:.6f}: rel change {abs(l0 - l1) / abs(l0):.2e}")


if __name__ == "__main__":
    torch.set_num_threads(8)
    x, y = G.load_pair("gp_samples_100000_0.2_2_0.2.pt")
    study("c2_se2d_n100000", x, y, "se", 0.2, 2.0, 0.2, 1e-4)
    x, y = G.usa_temp()
    study("c3_matern52_usatemp", x, y, "matern", 0.1, 1.0, 0.05, 1e-3, nu=2.5)
    x, y = G.synth(20000, 3, 1)
    study("c5_matern32_3d_n20000", x, y, "matern", 0.3, 1.5, 0.2, 1e-2, nu=1.5)
