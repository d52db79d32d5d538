"""The HIP path against the one stored output of the reference that REAL FINUFFT produced (experiments/
cg_preconditioning_realdata.csv, committed as tests/golden/data/cg_preconditioning_realdata.csv): the two operator bundles of
benchmark_cg_preconditioning_realdata.py:83-147 rebuilt through this package's NUFFT / compute_convolution_vector_vectorized_dD /
ToeplitzND / create_A_mean / ConjugateGradients, with the script's six diagonal preconditioners (:150-171) and its mean solves
(:174-201).  Every other golden goes through the exact-NUDFT stand-in; this one pins sign, mode order, scaling and the centring
of the Toeplitz vector to numbers the third-party transform itself wrote.  CPU twin: tests/test_oracle_realdata.py."""
import pytest
import torch

import _realdata as R

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("regime", ["hard", "very_hard"])
def test_hip_path_reproduces_real_finufft_csv(regime):
    from cg import ConjugateGradients
    from efgp_hip import cg_residual_history
    from efgpnd import NUFFT, ToeplitzND, compute_convolution_vector_vectorized_dD, create_A_mean, create_jacobi_precond, _cmplx
    from utils.kernels import get_xis
    ref = R.rows(regime)
    any_row = ref["none"]
    x, y = R.usa_temp()
    x, y = x.cuda(), y.cuda()
    k = R.kernel(regime)
    sig = R.REGIMES[regime][2]
    dtype, cdtype, d = x.dtype, _cmplx(x.dtype), 2
    xis_1d, h, mtot = get_xis(k, eps=R.EPS, L=R.domain_length(x), use_integral=True, l2scaled=False)          # :98-100
    assert h == float(any_row["h"]) and mtot == int(float(any_row["mtot"])) and mtot ** d == int(float(any_row["M"]))
    xis_1d = xis_1d.to(device=x.device, dtype=dtype)
    xis = torch.stack(torch.meshgrid(*(xis_1d for _ in range(d)), indexing="ij"), dim=-1).reshape(-1, d)
    ws = torch.sqrt(k.spectral_density(xis).to(dtype=cdtype) * h ** d)                                          # :103
    OUT = (mtot,) * d
    nufft = NUFFT(x, torch.zeros(d, dtype=dtype, device=x.device), h, R.NUFFT_EPS, cdtype=cdtype)               # :109
    v_kernel = compute_convolution_vector_vectorized_dD((mtot - 1) // 2, x, h).to(dtype=cdtype)                 # :113
    toeplitz = ToeplitzND(v_kernel, force_pow2=True)
    A = create_A_mean(ws, toeplitz, torch.tensor(sig, dtype=dtype), cdtype)
    rhs = ws * nufft.type1(y, out_shape=OUT).reshape(-1)                                                        # :117-118
    centre = tuple(((torch.tensor(v_kernel.shape) - 1) // 2).tolist())
    diag_t = float(v_kernel[centre].real.item())                                                                # :132-133
    # v[0] = N exactly; FINUFFT at its 6e-8 wrote N (1 - 1.9e-10) / N (1 - 0.9e-10); this transform is asked for the same 6e-8
    assert abs(diag_t - float(any_row["diag_toeplitz"])) < 1e-8 * 4766, diag_t
    got, res = {}, {}
    for name, c in R.PRECS.items():
        want = int(float(ref[name]["iters_completed"]))
        pre = None if c is None else create_jacobi_precond(ws, sig, diag_t if c == "N" else c)                  # :150-171
        with cg_residual_history(x.device, 2048) as rec:
            cg = ConjugateGradients(A, rhs, torch.zeros_like(rhs), tol=0.5 * R.CG_TOL, early_stopping=True, M_inv_apply=pre,
                                    max_iter=int(1.3 * want) + 8)
            cg.solve()
        hist = rec.values()[:int(cg.iters_completed)]
        cg = ConjugateGradients(A, rhs, torch.zeros_like(rhs), tol=R.CG_TOL, early_stopping=True, M_inv_apply=pre)
        beta = cg.solve()
        got[name] = int(cg.iters_completed)
        assert R.count_agrees(name, want, got[name], hist), (regime, name, got[name], want)
        res[name] = float(torch.linalg.norm(rhs - A(beta)) / torch.linalg.norm(rhs))
        # the script's rel_res_mean is the TRUE residual of the returned iterate (diagnose_efgpnd_learning_curve.py); < tol there
        assert float(ref[name]["rel_res_mean"]) < R.CG_TOL and res[name] < 1.05 * R.CG_TOL, (regime, name, res[name])
    print(f"\n{regime}: diag_toeplitz {diag_t!r} (csv {any_row['diag_toeplitz']}); iterations hip {got}; "
          f"csv { {n: int(float(r['iters_completed'])) for n, r in ref.items()} }; true residuals { {n: f'{v:.2e}' for n, v in res.items()} }")
