"""GPU parity: Toeplitz mat-vec and the fused CG (C ABI) vs the oracle restatement of
ToeplitzND (efgpnd.py:1239-1393) and cg.py:86-244."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a = a.detach().cpu()
    b = b.detach().cpu()
    return float(torch.linalg.norm((a - b).reshape(-1)) / torch.linalg.norm(b.reshape(-1)))


def _setup(d, mtot, N=500, seed=0, h=0.4):
    from oracle import efgp_oracle as O
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(N, d, generator=g, dtype=torch.float64) * 2 - 1
    v = O.conv_vector(x, h, (mtot - 1) // 2)
    return x, v, O.Toeplitz(v)


@pytest.mark.parametrize("d,mtot", [(1, 35), (2, 23), (2, 15), (3, 7), (2, 71)])
def test_toeplitz_apply(d, mtot):
    from efgp_hip import ToeplitzOp
    x, v, T = _setup(d, mtot)
    op = ToeplitzOp(v.cuda())
    assert op.fft_shape == T.fft_shape and op.ns == T.ns
    g = torch.Generator().manual_seed(1)
    u = torch.complex(torch.randn(3, T.size, generator=g, dtype=torch.float64),
                      torch.randn(3, T.size, generator=g, dtype=torch.float64))
    assert _rel(op.apply(u.cuda()), T(u)) < 1e-13
    assert _rel(op.apply(u[0].cuda()), T(u[0])) < 1e-13


@pytest.mark.parametrize("d,mtot,precond", [(1, 35, True), (2, 23, True), (2, 23, False), (3, 7, True)])
def test_cg_single_iterate_parity(d, mtot, precond):
    """Same iteration count and solution as the oracle CG at a moderate tolerance."""
    from efgp_hip import ToeplitzOp, cg_solve
    from oracle import efgp_oracle as O
    x, v, T = _setup(d, mtot, N=800)
    M = T.size
    g = torch.Generator().manual_seed(2)
    ws = torch.exp(-3.0 * torch.rand(M, generator=g, dtype=torch.float64)).to(torch.complex128)
    sig2 = 0.3
    b = torch.complex(torch.randn(M, generator=g, dtype=torch.float64), torch.randn(M, generator=g, dtype=torch.float64))
    diag = O.jacobi_diag(ws, sig2, float(v.reshape(-1)[v.numel() // 2].real)) if precond else None
    A = O.make_A_mean(ws, T, sig2)
    xo, ito = O.cg_single(A, b, torch.zeros_like(b), 1e-8, diag=diag)
    op = ToeplitzOp(v.cuda())
    xg, itg, _ = cg_solve(op, ws.cuda(), sig2, 0, b.cuda(), torch.zeros_like(b).cuda(), 1e-8,
                          diag=diag.cuda() if precond else None, batched=False)
    # hundreds of iterations with a different FFT: the stopping test may flip one iteration late/early
    assert abs(itg - ito) <= (0 if ito < 100 else 1 + ito // 200)
    assert _rel(xg, xo) < 1e-7
    # warm start from the solution: converges in one step (cg.py:94-95 residual with x0)
    xg2, it2, _ = cg_solve(op, ws.cuda(), sig2, 0, b.cuda(), xg, 1e-6, diag=diag.cuda() if precond else None, batched=False)
    xo2, ito2 = O.cg_single(A, b, xo, 1e-6, diag=diag)
    assert it2 == ito2


def test_cg_batched_mask_semantics():
    """Per-row early stop, +1 terminating pass, A_var variant, no preconditioner (efgpnd.py:1648-1655)."""
    from efgp_hip import ToeplitzOp, cg_solve
    from oracle import efgp_oracle as O
    x, v, T = _setup(2, 15, N=600)
    M = T.size
    g = torch.Generator().manual_seed(3)
    ws = torch.exp(-2.0 * torch.rand(M, generator=g, dtype=torch.float64)).to(torch.complex128)
    sig2 = 0.2
    B = 9
    b = torch.complex(torch.randn(B, M, generator=g, dtype=torch.float64), torch.randn(B, M, generator=g, dtype=torch.float64))
    b[3] = 0.0                    # zero rhs: denom falls back to 1, converges at once
    b[5] *= 1e-3
    A = O.make_A_var(ws, T, sig2)
    xo, ito = O.cg_batched(A, b, torch.zeros_like(b), 1e-7, max_iter=1000)
    op = ToeplitzOp(v.cuda())
    xg, itg, rows = cg_solve(op, ws.cuda(), sig2, 1, b.cuda(), torch.zeros_like(b).cuda(), 1e-7, max_iter=1000)
    assert itg == ito
    assert _rel(xg, xo) < 1e-9
    assert rows[3] == 1
    # iteration cap reached without convergence: count equals the cap
    xo3, ito3 = O.cg_batched(A, b, torch.zeros_like(b), 1e-30, max_iter=7)
    xg3, itg3, _ = cg_solve(op, ws.cuda(), sig2, 1, b.cuda(), torch.zeros_like(b).cuda(), 1e-30, max_iter=7)
    assert itg3 == ito3 == 7
    assert _rel(xg3, xo3) < 1e-10


@pytest.mark.parametrize("d,mtot", [(1, 35), (1, 501), (2, 23), (2, 9), (3, 7)])
def test_persistent_and_multikernel_cg_agree(d, mtot, monkeypatch):
    """The single-launch LDS-FFT solver and the rocFFT multi-kernel solver implement the same CG."""
    from efgp_hip import ToeplitzOp, cg_solve
    from oracle import efgp_oracle as O
    x, v, T = _setup(d, mtot, N=700, seed=5)
    M = T.size
    g = torch.Generator().manual_seed(4)
    ws = torch.exp(-2.5 * torch.rand(M, generator=g, dtype=torch.float64)).to(torch.complex128)
    sig2 = 0.25
    B = 5
    b = torch.complex(torch.randn(B, M, generator=g, dtype=torch.float64), torch.randn(B, M, generator=g, dtype=torch.float64))
    diag = O.jacobi_diag(ws, sig2, 700.0)
    op = ToeplitzOp(v.cuda())
    out = {}
    for mode in ("persistent", "multikernel"):
        if mode == "multikernel":
            monkeypatch.setenv("EFGP_NO_PERSISTENT_CG", "1")
        else:
            monkeypatch.delenv("EFGP_NO_PERSISTENT_CG", raising=False)
        out[mode] = cg_solve(op, ws.cuda(), sig2, 0, b.cuda(), torch.zeros_like(b).cuda(), 1e-9, diag=diag.cuda())
        out[mode + "_1"] = cg_solve(op, ws.cuda(), sig2, 1, b[0].cuda(), torch.zeros_like(b[0]).cuda(), 1e-9, batched=False)
    xo, ito = O.cg_batched(O.make_A_mean(ws, T, sig2), b, torch.zeros_like(b), 1e-9, diag=diag)
    for mode in ("persistent", "multikernel"):
        xg, itg, rows = out[mode]
        assert abs(itg - ito) <= 1
        assert _rel(xg, xo) < 1e-7
    assert abs(out["persistent_1"][1] - out["multikernel_1"][1]) <= 1
    assert _rel(out["persistent_1"][0], out["multikernel_1"][0]) < 1e-7


@pytest.mark.parametrize("d,mtot,precond", [(2, 23, True), (2, 23, False), (1, 35, True), (3, 7, True), (2, 15, True)])
def test_fused_mean_system_matches_general_solve(d, mtot, precond):
    """efgp_cg_solve_mean_async forms ws*F*y, the Jacobi diagonal v[0]|ws|^2+sigma^2 and beta_0 = 0 inside the
    kernel: same iterates as the general entry point given the materialised tensors (and as the oracle)."""
    from efgp_hip import ToeplitzOp, cg_solve, cg_solve_mean_async
    from oracle import efgp_oracle as O
    x, v, T = _setup(d, mtot, N=900, seed=4)
    M = T.size
    g = torch.Generator().manual_seed(8)
    # the entry point's contract (include/efgp_hip.h): fy is the transform of REAL data and ws is real and even
    shape = (mtot,) * d
    dims = tuple(range(d))
    w = torch.exp(-2.5 * torch.rand(shape, generator=g, dtype=torch.float64))
    ws = (0.5 * (w + torch.flip(w, dims))).reshape(-1).to(torch.complex128)
    sig2 = 0.25
    fy = torch.complex(torch.randn(shape, generator=g, dtype=torch.float64), torch.randn(shape, generator=g, dtype=torch.float64))
    fy = (0.5 * (fy + torch.flip(fy, dims).conj())).reshape(-1)
    vd = v.cuda()
    centre = vd[tuple((s - 1) // 2 for s in vd.shape)].real            # a view into v: no copy, no kernel
    op = ToeplitzOp(vd)
    res = cg_solve_mean_async(op, ws.cuda(), sig2, centre if precond else None, fy.cuda(), 1e-8)
    assert res is not None
    beta, lazy = res
    rhs = ws * fy
    diag = (float(centre) * ws.abs().pow(2).real + sig2) if precond else None
    xg, itg, _ = cg_solve(op, ws.cuda(), sig2, 0, rhs.cuda(), torch.zeros_like(rhs).cuda(), 1e-8,
                          diag=diag.cuda() if precond else None, batched=False)
    # 2-D blocks up to 31 x 31 run the Hermitian 64 x 64 kernel (real transforms): same recurrences, different rounding
    herm = d == 2 and mtot <= 31
    assert abs(int(lazy) - itg) <= (1 if herm else 0)
    assert _rel(beta, xg) < (1e-9 if herm else 1e-13)
    xo, ito = O.cg_single(O.make_A_mean(ws, T, sig2), rhs, torch.zeros_like(rhs), 1e-8, diag=diag)
    assert abs(int(lazy) - ito) <= (0 if ito < 100 else 1 + ito // 200) and _rel(beta, xo) < 1e-7
    assert beta.shape == fy.shape


@pytest.mark.parametrize("mtot", [41, 71, 131])
def test_line_fft_iteration_matches_generic_and_oracle(mtot, monkeypatch):
    """2-D grids beyond one CU (F = 128, 256, 512): the fused line-FFT iteration (three launches of pruned in-LDS
    transforms) against the generic pad + rocFFT iteration (single and batched systems, both operators) and, at
    the smallest size, the oracle CG.  The Toeplitz vector is a synthetic Hermitian one (an exact conv vector at
    mtot = 131 would dominate the test time on the CPU)."""
    from efgp_hip import ToeplitzOp, cg_solve
    from oracle import efgp_oracle as O
    g = torch.Generator().manual_seed(12)
    L = 2 * mtot - 1
    v = torch.complex(torch.randn(L, L, generator=g, dtype=torch.float64), torch.randn(L, L, generator=g, dtype=torch.float64))
    v = (v + v.flip(0, 1).conj()) / 2
    v[mtot - 1, mtot - 1] = 3.0 * L                       # diagonally dominant: positive definite T
    M = mtot * mtot
    ws = torch.exp(-2.0 * torch.rand(M, generator=g, dtype=torch.float64)).to(torch.complex128)
    sig2 = 0.5
    B = 3
    b = torch.complex(torch.randn(B, M, generator=g, dtype=torch.float64), torch.randn(B, M, generator=g, dtype=torch.float64))
    diag = O.jacobi_diag(ws, sig2, 3.0 * L)
    op = ToeplitzOp(v.cuda())
    assert op.fft_shape[0] in (128, 256, 512)
    res = {}
    for mode in ("lines", "generic"):
        if mode == "generic":
            monkeypatch.setenv("EFGP_NO_CG_LINES", "1")
        else:
            monkeypatch.delenv("EFGP_NO_CG_LINES", raising=False)
        res[mode] = cg_solve(op, ws.cuda(), sig2, 0, b.cuda(), torch.zeros_like(b).cuda(), 1e-9, diag=diag.cuda())
        res[mode + "_var"] = cg_solve(op, ws.cuda(), sig2, 1, b[1].cuda(), 0.1 * b[0].cuda(), 1e-9, batched=False)
    assert abs(res["lines"][1] - res["generic"][1]) <= 1 and _rel(res["lines"][0], res["generic"][0]) < 1e-8
    assert abs(res["lines_var"][1] - res["generic_var"][1]) <= 1 and _rel(res["lines_var"][0], res["generic_var"][0]) < 1e-8
    assert res["lines"][1] < 2 * M                        # converged, not capped
    if mtot == 41:
        T = O.Toeplitz(v)
        xo, ito = O.cg_batched(O.make_A_mean(ws, T, sig2), b, torch.zeros_like(b), 1e-9, diag=diag)
        assert abs(res["lines"][1] - ito) <= 1 and _rel(res["lines"][0], xo) < 1e-7


@pytest.mark.parametrize("mtot", [17, 23, 35])
def test_line_fft_iteration_3d(mtot, monkeypatch):
    """3-D grids (F = 64, 64, 128): five pruned line-transform launches per matvec against the generic pad + rocFFT
    iteration (single and batched systems, both operators) and, at the smallest size, the oracle CG."""
    from efgp_hip import ToeplitzOp, cg_solve
    from oracle import efgp_oracle as O
    g = torch.Generator().manual_seed(31)
    L = 2 * mtot - 1
    v = torch.complex(torch.randn(L, L, L, generator=g, dtype=torch.float64), torch.randn(L, L, L, generator=g, dtype=torch.float64))
    v = (v + v.flip(0, 1, 2).conj()) / 2
    v[mtot - 1, mtot - 1, mtot - 1] = 6.0 * L ** 1.5          # dominant diagonal: positive definite T
    M = mtot ** 3
    ws = torch.exp(-2.0 * torch.rand(M, generator=g, dtype=torch.float64)).to(torch.complex128)
    sig2 = 0.5
    B = 3
    b = torch.complex(torch.randn(B, M, generator=g, dtype=torch.float64), torch.randn(B, M, generator=g, dtype=torch.float64))
    diag = O.jacobi_diag(ws, sig2, 6.0 * L ** 1.5)
    op = ToeplitzOp(v.cuda())
    assert op.fft_shape[0] in (64, 128)
    res = {}
    for mode in ("lines", "generic"):
        if mode == "generic":
            monkeypatch.setenv("EFGP_NO_CG_LINES", "1")
        else:
            monkeypatch.delenv("EFGP_NO_CG_LINES", raising=False)
        res[mode] = cg_solve(op, ws.cuda(), sig2, 0, b.cuda(), torch.zeros_like(b).cuda(), 1e-9, diag=diag.cuda())
        res[mode + "_var"] = cg_solve(op, ws.cuda(), sig2, 1, b[1].cuda(), 0.1 * b[0].cuda(), 1e-9, batched=False)
    assert abs(res["lines"][1] - res["generic"][1]) <= 1 and _rel(res["lines"][0], res["generic"][0]) < 1e-8
    assert abs(res["lines_var"][1] - res["generic_var"][1]) <= 1 and _rel(res["lines_var"][0], res["generic_var"][0]) < 1e-8
    assert res["lines"][1] < 2 * M
    if mtot == 17:
        T = O.Toeplitz(v)
        xo, ito = O.cg_batched(O.make_A_mean(ws, T, sig2), b, torch.zeros_like(b), 1e-9, diag=diag)
        assert abs(res["lines"][1] - ito) <= 1 and _rel(res["lines"][0], xo) < 1e-7


@pytest.mark.parametrize("mtot", [17, 23, 29, 32])
def test_one_launch_toeplitz_spectrum_matches_rocfft_path(mtot, monkeypatch):
    """64 x 64 circulant grids: the single-launch LDS transform of the Toeplitz vector (toeplitz_vhat_2d64_kernel) and
    the pad + rocFFT route (EFGP_NO_VHAT64) give the same operator, and both match the oracle (efgpnd.py:1283-1290)."""
    from efgp_hip import ToeplitzOp
    from oracle import efgp_oracle as O
    g = torch.Generator().manual_seed(3)
    if mtot % 2:
        x, v, T = _setup(2, mtot, N=900, seed=7)
    else:      # even block size: lags box (2 mtot - 1)^2 = 63^2 of arbitrary complex numbers (the operator is generic)
        L = 2 * mtot - 1
        v = torch.complex(torch.randn(L, L, generator=g, dtype=torch.float64), torch.randn(L, L, generator=g, dtype=torch.float64))
        T = O.Toeplitz(v)
    assert tuple(T.fft_shape) == (64, 64)
    u = torch.complex(torch.randn(4, T.size, generator=g, dtype=torch.float64),
                      torch.randn(4, T.size, generator=g, dtype=torch.float64))
    fused = ToeplitzOp(v.cuda()).apply(u.cuda())
    monkeypatch.setenv("EFGP_NO_VHAT64", "1")
    plain = ToeplitzOp(v.cuda()).apply(u.cuda())
    assert _rel(fused, plain) < 1e-14
    assert _rel(fused, T(u)) < 1e-13


@pytest.mark.parametrize("mtot,nb", [(41, 8), (71, 8), (131, 4), (57, 1), (99, 3)])
def test_cooperative_solve_equals_multi_launch_iteration(mtot, nb, monkeypatch):
    """cg_coop2d_kernel (the whole CG loop of a 128^2..512^2 grid in ONE launch over G workgroups per system with grid
    barriers) against the multi-launch line-FFT iteration: same per-row iteration counts, solutions equal to rounding, and
    bit-identical results launch after launch with as many systems in flight as fit the chip (a partial-sum array reused
    by two consecutive reductions -- a cross-workgroup race -- showed up only with several systems resident)."""
    from efgp_hip import ToeplitzOp, cg_solve
    g = torch.Generator().manual_seed(12)
    L = 2 * mtot - 1
    v = torch.complex(torch.randn(L, L, generator=g, dtype=torch.float64), torch.randn(L, L, generator=g, dtype=torch.float64))
    v = (v + v.flip(0, 1).conj()) / 2
    v[mtot - 1, mtot - 1] = 3.0 * L
    M = mtot * mtot
    ws = torch.exp(-2.0 * torch.rand(M, generator=g, dtype=torch.float64)).to(torch.complex128)
    b = torch.complex(torch.randn(nb, M, generator=g, dtype=torch.float64), torch.randn(nb, M, generator=g, dtype=torch.float64))
    b = b * torch.logspace(-2, 1, nb, dtype=torch.float64)[:, None]
    x0 = 0.05 * torch.complex(torch.randn(nb, M, generator=g, dtype=torch.float64), torch.randn(nb, M, generator=g, dtype=torch.float64))
    diag = 3.0 * L * ws.abs().pow(2).real + 0.5
    op = ToeplitzOp(v.cuda())
    args = (op, ws.cuda(), 0.5, 0, b.cuda(), x0.cuda(), 1e-9)
    monkeypatch.setenv("EFGP_NO_CG_COOP", "1")
    ref = cg_solve(*args, diag=diag.cuda(), batched=True)
    ref1 = cg_solve(op, ws.cuda(), 0.5, 1, b[0].cuda(), x0[0].cuda(), 1e-9, batched=False)
    monkeypatch.delenv("EFGP_NO_CG_COOP")
    first = None
    for rep in range(6):
        out = cg_solve(*args, diag=diag.cuda(), batched=True)
        assert out[1] == ref[1] and out[2] == ref[2], (rep, out[1:], ref[1:])
        for r in range(nb):                      # same recurrences; the partial sums are grouped by workgroup, not by 64 blocks
            assert _rel(out[0][r], ref[0][r]) < 1e-12, (rep, r, _rel(out[0][r], ref[0][r]))
        if first is None:
            first = out[0].clone()
        assert torch.equal(out[0], first), rep   # bit-identical from launch to launch: fixed summation order, no race
    out1 = cg_solve(op, ws.cuda(), 0.5, 1, b[0].cuda(), x0[0].cuda(), 1e-9, batched=False)       # A_var, single-system rule, no diagonal
    assert out1[1] == ref1[1] and _rel(out1[0], ref1[0]) < 1e-12


def test_async_solve_on_mid_size_grids():
    """efgp_cg_solve_async on 128^2..512^2 grids enqueues the cooperative launches without any host synchronisation: same
    solution and counts as the synchronous entry."""
    from efgp_hip import ToeplitzOp, cg_solve, cg_solve_async
    mtot = 57
    g = torch.Generator().manual_seed(5)
    L = 2 * mtot - 1
    v = torch.complex(torch.randn(L, L, generator=g, dtype=torch.float64), torch.randn(L, L, generator=g, dtype=torch.float64))
    v = (v + v.flip(0, 1).conj()) / 2
    v[mtot - 1, mtot - 1] = 3.0 * L
    M = mtot * mtot
    ws = torch.exp(-2.0 * torch.rand(M, generator=g, dtype=torch.float64)).to(torch.complex128)
    b = torch.complex(torch.randn(5, M, generator=g, dtype=torch.float64), torch.randn(5, M, generator=g, dtype=torch.float64))
    diag = 3.0 * L * ws.abs().pow(2).real + 0.5
    op = ToeplitzOp(v.cuda())
    res = cg_solve_async(op, ws.cuda(), 0.5, 0, b.cuda(), torch.zeros_like(b).cuda(), 1e-9, diag=diag.cuda(), batched=True)
    assert res is not None
    xa, lazy = res
    xs, its, rows = cg_solve(op, ws.cuda(), 0.5, 0, b.cuda(), torch.zeros_like(b).cuda(), 1e-9, diag=diag.cuda(), batched=True)
    assert int(lazy) == its and list(lazy.rows) == rows
    assert torch.equal(xa, xs)
    # hermitian=True is a CONTRACT on these grids too since round 3 (cg_coop2d_herm_kernel): data that are not the transform of a
    # real function are refused in the result (NaN, -2); conjugate-even data give the general solver's answer
    one = cg_solve_async(op, ws.cuda(), 0.5, 0, b[1].cuda(), torch.zeros_like(b[1]).cuda(), 1e-9, diag=diag.cuda(), batched=False, hermitian=True)
    assert one is not None and bool(torch.isnan(one[0].real).all())
    with pytest.raises(RuntimeError, match="not the transform of real data"):
        int(one[1])
    wsq = ws.reshape(mtot, mtot)
    ws_e = ((wsq + wsq.flip(0, 1)) / 2).reshape(-1)
    bh = b[1].reshape(mtot, mtot)
    bh = ((bh + bh.flip(0, 1).conj()) / 2).reshape(-1)
    dg_e = 3.0 * L * ws_e.abs().pow(2).real + 0.5
    xg, itg, _ = cg_solve(op, ws_e.cuda(), 0.5, 0, bh.cuda(), torch.zeros_like(bh).cuda(), 1e-9, diag=dg_e.cuda(), batched=False)
    xh, lz = cg_solve_async(op, ws_e.cuda(), 0.5, 0, bh.cuda(), torch.zeros_like(bh).cuda(), 1e-9, diag=dg_e.cuda(), batched=False, hermitian=True)
    assert abs(int(lz) - itg) <= 1 and _rel(xh, xg) < 1e-7


def _psd_system(n0, n1, seed, nb):
    """Toeplitz vector of 2500 random points on an (n0, n1) block (positive semi-definite, as a model's), even real ws, right-hand
    sides that are transforms of real data."""
    g = torch.Generator().manual_seed(seed)
    xp = torch.rand(2500, 2, generator=g, dtype=torch.float64) * 2 - 1
    k0 = torch.arange(-(n0 - 1), n0, dtype=torch.float64)
    k1 = torch.arange(-(n1 - 1), n1, dtype=torch.float64)
    E0 = torch.exp(-2j * math.pi * 0.31 * k0[:, None] * xp[None, :, 0])
    E1 = torch.exp(-2j * math.pi * 0.27 * k1[:, None] * xp[None, :, 1])
    v = (E0 @ E1.T).contiguous()
    w = torch.exp(-3.0 * torch.rand(n0, n1, generator=g, dtype=torch.float64))
    ws = (0.5 * (w + w.flip(0, 1))).reshape(-1).to(torch.complex128)
    b = torch.complex(torch.randn(nb, n0, n1, generator=g, dtype=torch.float64), torch.randn(nb, n0, n1, generator=g, dtype=torch.float64))
    b = (0.5 * (b + b.flip(1, 2).conj())).reshape(nb, -1) * torch.logspace(-1, 1, nb, dtype=torch.float64)[:, None]
    return v, ws, b


@pytest.mark.parametrize("n0,n1,grid", [(41, 41, (96, 96)), (47, 47, (96, 96)), (67, 67, (192, 192)), (71, 71, (192, 192)),
                                        (95, 95, (192, 192)), (131, 131, (384, 384)), (41, 71, (96, 192)), (67, 41, (192, 96)),
                                        (57, 57, (128, 128)), (33, 99, (96, 256))])
def test_cooperative_solves_on_the_smallest_grid_equal_the_power_of_two_grid(n0, n1, grid, monkeypatch):
    """Round 4: the cooperative 2-D solves run on the smallest grid of the in-wave transforms that holds 2 n - 1 per axis -- 96, 192,
    384 (48 R lines: `line_fft_inwave<R, MUL, 48>`) between the powers of two; the Toeplitz product is exact on any such grid
    (efgpnd.py:1266-1271).  Against the same solves on the reference's next_pow2 grid (EFGP_NO_COOP_SMALL): iteration counts equal
    (+- 1 in several hundred), solutions within 10 x the CG tolerance -- Hermitian kernel (single system) and general kernel (batch with per-row stopping, single system)."""
    from efgp_hip import ToeplitzOp, cg_solve
    nb = 5
    v, ws, b = _psd_system(n0, n1, 3 + n0, nb)
    sig = 25.0                                            # well conditioned: tens of iterations, counts are crisp
    diag = (2500.0 * ws.abs().pow(2).real + sig).cuda()
    op = ToeplitzOp(v.cuda())
    pow2 = [1 << (2 * n - 2).bit_length() for n in (n0, n1)]
    assert list(op.fft_shape) == pow2 and tuple(op.cg_shape()) == grid and tuple(op.cg_shape(hermitian=True)) == grid
    runs = {}
    for small in (True, False):
        if not small:
            monkeypatch.setenv("EFGP_NO_COOP_SMALL", "1")
            assert list(op.cg_shape()) == pow2
        herm = cg_solve(op, ws.cuda(), sig, 0, b[2].cuda(), None, 1e-8, diag=diag, batched=False, hermitian=True)
        gen1 = cg_solve(op, ws.cuda(), sig, 0, b[1].cuda(), None, 1e-8, diag=diag, batched=False)
        genb = cg_solve(op, ws.cuda(), sig, 0, b.cuda(), None, 1e-8, diag=diag, batched=True)
        # A_var without a preconditioner (the variance's systems): 50 forced iterations -- on the ill-conditioned operator the
        # rounding of two correct transforms separates the iterates along the recurrence, as it does between any two solvers
        genv = cg_solve(op, ws.cuda(), sig, 1, b.cuda(), None, 1e-300, batched=True, max_iter=50, early_stop=False)
        runs[small] = (herm, gen1, genb, genv)
    monkeypatch.delenv("EFGP_NO_COOP_SMALL")
    for a_, b_ in zip(runs[True], runs[False]):
        # same recurrences, different rounding: a stopping index may move by one in a few hundred iterations
        assert all(abs(p - q) <= 1 + q // 200 for p, q in zip(a_[2], b_[2])), (a_[1:], b_[1:])
        assert _rel(a_[0], b_[0]) < 1e-5, _rel(a_[0], b_[0])          # cond x tolerance: two correct solvers on one system
    # the TRUE residual of the small-grid solutions under the operator applied on the REFERENCE's grid (efgp_toeplitz_apply)
    wsd = ws.cuda()
    for x_, rhs in ((runs[True][0][0], b[2].cuda()), (runs[True][1][0], b[1].cuda())):
        Ax = wsd * op.apply(wsd * x_) + sig * x_
        assert float(torch.linalg.norm(Ax - rhs) / torch.linalg.norm(rhs)) < 1.05e-8
    assert len(set(runs[True][2][2])) > 1             # per-row stopping happened in the batch
    assert runs[True][0][1] < 2 * n0 * n1            # the preconditioned mean system converged before its iteration cap
