"""CPU twin of tests/test_gpu_realdata_csv.py: the oracle (exact NUDFT + the reference's CG, cg.py:86-153) against the numbers
real FINUFFT produced in experiments/cg_preconditioning_realdata.csv -- the pin at the third-party boundary that does not go
through oracle/standin.  A convention error (sign, mode order, scaling, centring of the Toeplitz vector) shared by stand-in,
oracle and HIP path would move these iteration counts and the Toeplitz diagonal."""
import pytest
import torch

import _realdata as R
from oracle import efgp_oracle as O


@pytest.mark.parametrize("regime", ["hard", "very_hard"])
def test_oracle_reproduces_real_finufft_csv(regime):
    from utils.kernels import get_xis
    ref = R.rows(regime)
    x, y = R.usa_temp()
    k = R.kernel(regime)
    L = R.domain_length(x)
    # the grid through the product's host-side get_xis: bit for bit the CSV's h, mtot, M
    _, h, mtot = get_xis(k, eps=R.EPS, L=L, use_integral=True, l2scaled=False)
    any_row = ref["none"]
    assert h == float(any_row["h"]) and mtot == int(float(any_row["mtot"])) and mtot * mtot == int(float(any_row["M"]))
    # ... and through the oracle's restatement
    ko = O.KernelSpec("se", 2, k.get_hyper("lengthscale"), k.get_hyper("variance"))
    xis, ho, mo = O.get_xis(ko, R.EPS, L)
    assert mo == mtot and abs(ho - h) <= 1e-15 * h
    sig = R.REGIMES[regime][2]
    ws = torch.from_numpy(O.feature_weights(ko, xis, ho)).to(torch.complex128)
    v = O.conv_vector(x, ho, (mtot - 1) // 2)
    diag_t = float(v[(v.shape[0] - 1) // 2, (v.shape[1] - 1) // 2].real)
    # FINUFFT at 6e-8 wrote 4765.999999107 (hard) / ...557 (very_hard); the exact transform gives N
    assert abs(diag_t - 4766.0) < 1e-8 and abs(diag_t - float(any_row["diag_toeplitz"])) < 6e-8 * 4766
    T = O.Toeplitz(v)
    A = O.make_A_mean(ws, T, sig)
    rhs = ws * O.nudft_type1(x, ho, y, (mtot, mtot)).reshape(-1)
    got = {}
    want_max = int(1.3 * max(float(r["iters_completed"]) for r in ref.values()))
    for name, c in R.PRECS.items():
        diag = None if c is None else O.jacobi_diag(ws, sig, diag_t if c == "N" else c).real
        hist = []
        O.cg_single(A, rhs, torch.zeros_like(rhs), 0.5 * R.CG_TOL, diag=diag, history=hist, max_iter=want_max)
        sol, it = O.cg_single(A, rhs, torch.zeros_like(rhs), R.CG_TOL, diag=diag)
        got[name] = it
        want = int(float(ref[name]["iters_completed"]))
        assert R.count_agrees(name, want, it, hist), (regime, name, it, want)
        res = float(torch.linalg.norm(rhs - A(sol)) / torch.linalg.norm(rhs))
        assert res < 1.05 * R.CG_TOL, (regime, name, res)
    print(f"\n{regime}: oracle iterations {got}; csv { {n: int(float(r['iters_completed'])) for n, r in ref.items()} }")
