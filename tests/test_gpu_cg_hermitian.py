"""GPU parity of the Hermitian 64 x 64 CG kernel (csrc/cg_persistent.hip: cg_herm64_kernel).

Every CG system of an EFGP model has a right-hand side D F* (real vector) (efgpnd.py:186-189, 792, 1657): coefficients of
a real function on the symmetric mode grid.  The kernel solves such systems on real transforms (half the line
transforms of the complex kernel).  Checked here: against the general complex kernel and the oracle's cg.py restatement
(iteration counts, solutions), exact conjugate symmetry of the output, both operators, explicit and formed Jacobi
diagonals, non-zero Hermitian start vectors, per-row stopping of batched solves, and the refusal of inputs that break
the contract.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a = a.detach().cpu()
    b = b.detach().cpu()
    return float(torch.linalg.norm((a - b).reshape(-1)) / torch.linalg.norm(b.reshape(-1)))


def _herm(t):
    return 0.5 * (t + torch.flip(t, dims=(-2, -1)).conj())


def _system(mtot, seed, N=700, rows=None):
    from oracle import efgp_oracle as O
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(N, 2, generator=g, dtype=torch.float64) * 2 - 1
    v = O.conv_vector(x, 0.4, (mtot - 1) // 2)
    w = torch.exp(-2.5 * torch.rand(mtot, mtot, generator=g, dtype=torch.float64))
    ws = (0.5 * (w + torch.flip(w, dims=(0, 1)))).reshape(-1).to(torch.complex128)
    shape = (mtot, mtot) if rows is None else (rows, mtot, mtot)
    b = _herm(torch.complex(torch.randn(shape, generator=g, dtype=torch.float64), torch.randn(shape, generator=g, dtype=torch.float64)))
    b = b.reshape(-1) if rows is None else b.reshape(rows, -1)
    return v, O.Toeplitz(v), ws, b


@pytest.mark.parametrize("mtot,precond,tol", [(23, True, 1e-8), (23, False, 1e-6), (17, True, 1e-10), (31, True, 1e-6), (29, True, 1e-4),
                                              (15, True, 1e-8), (7, True, 1e-8), (3, False, 1e-8)])   # small blocks: 64 x 64 embedding
def test_mean_solve_matches_complex_kernel_and_oracle(mtot, precond, tol, monkeypatch):
    from efgp_hip import ToeplitzOp, cg_solve_mean_async
    from oracle import efgp_oracle as O
    v, T, ws, fy = _system(mtot, 3)
    vd = v.cuda()
    centre = vd[tuple((s - 1) // 2 for s in vd.shape)].real
    op = ToeplitzOp(vd)
    beta, lazy = cg_solve_mean_async(op, ws.cuda(), 0.25, centre if precond else None, fy.cuda(), tol)
    it_h = int(lazy)
    monkeypatch.setenv("EFGP_NO_CG_HERM", "1")
    beta_c, lazy_c = cg_solve_mean_async(op, ws.cuda(), 0.25, centre if precond else None, fy.cuda(), tol)
    it_c = int(lazy_c)
    monkeypatch.delenv("EFGP_NO_CG_HERM")
    # Same recurrences, different rounding: the two residual curves separate slowly (fastest without the preconditioner,
    # as the reference's own curve does under a 1e-13 perturbation, oracle/sensitivity_r2.py), so the stopping index may
    # move by a few per cent; both solutions satisfy the stopping rule, i.e. agree to cond(A) * tol.
    slack = 1 + it_c // (200 if precond else 50)
    xtol = (100 if precond else 1e4) * tol
    assert abs(it_h - it_c) <= slack
    assert _rel(beta, beta_c) < xtol
    rhs = ws * fy
    diag = (float(centre) * ws.abs().pow(2).real + 0.25) if precond else None
    xo, ito = O.cg_single(O.make_A_mean(ws, T, 0.25), rhs, torch.zeros_like(rhs), tol, diag=diag)
    assert abs(it_h - ito) <= slack
    assert _rel(beta, xo) < xtol
    # TRUE residual of the returned solution under the oracle's operator: that of the oracle's own solution (at tight
    # tolerances the recursive residual undershoots the true one for every implementation alike)
    A = O.make_A_mean(ws, T, 0.25)
    true_h = float(torch.linalg.norm(A(beta.cpu()) - rhs) / torch.linalg.norm(rhs))
    true_o = float(torch.linalg.norm(A(xo) - rhs) / torch.linalg.norm(rhs))
    assert true_h < 1.05 * true_o + 0.1 * tol, (true_h, true_o)
    # the output is exactly conjugate-even: the mirrored half is written as the conjugate
    bq = beta.cpu().reshape(mtot, mtot)
    assert torch.equal(torch.flip(bq, dims=(0, 1)).conj()[: mtot // 2], bq[: mtot // 2])


@pytest.mark.parametrize("variant", [0, 1])
def test_batched_hermitian_solves(variant, monkeypatch):
    """One system per workgroup with per-row stopping (cg.py:190-241), explicit Jacobi diagonal, non-zero Hermitian start."""
    from efgp_hip import ToeplitzOp, cg_solve
    mtot, B = 23, 9
    v, T, ws, b = _system(mtot, 11, rows=B)
    b = b * torch.logspace(-3, 2, B, dtype=torch.float64)[:, None]        # rows converge at different iterations
    g = torch.Generator().manual_seed(5)
    x0 = 0.1 * _herm(torch.complex(torch.randn(B, mtot, mtot, generator=g, dtype=torch.float64),
                                   torch.randn(B, mtot, mtot, generator=g, dtype=torch.float64))).reshape(B, -1)
    diag = (3.0 * ws.abs().pow(2).real + 0.3) if variant == 0 else None
    op = ToeplitzOp(v.cuda())
    args = (op, ws.cuda(), 0.3, variant, b.cuda(), x0.cuda(), 1e-9)
    kw = dict(diag=diag.cuda() if diag is not None else None, batched=True)
    xh, ith, rows_h = cg_solve(*args, hermitian=True, **kw)
    xc, itc, rows_c = cg_solve(*args, **kw)
    assert abs(ith - itc) <= 1
    assert all(abs(a - c) <= 1 + c // 100 for a, c in zip(rows_h, rows_c)), (rows_h, rows_c)
    assert len(set(rows_h)) > 1                                          # per-row stopping really happened
    for r in range(B):
        assert _rel(xh[r], xc[r]) < 1e-7
    # a single system through the same entry (single-system stopping rule, cg.py:132)
    x1, it1, _ = cg_solve(op, ws.cuda(), 0.3, variant, b[2].cuda(), x0[2].cuda(), 1e-9, diag=kw["diag"], batched=False, hermitian=True)
    x2, it2, _ = cg_solve(op, ws.cuda(), 0.3, variant, b[2].cuda(), x0[2].cuda(), 1e-9, diag=kw["diag"], batched=False)
    assert abs(it1 - it2) <= 1 and _rel(x1, x2) < 1e-7


def test_contract_violations_are_refused():
    from efgp_hip import ToeplitzOp, cg_solve_async, cg_solve_mean_async
    mtot = 23
    v, T, ws, b = _system(mtot, 2)
    op = ToeplitzOp(v.cuda())
    g = torch.Generator().manual_seed(1)
    bad = torch.complex(torch.randn(mtot * mtot, generator=g, dtype=torch.float64), torch.randn(mtot * mtot, generator=g, dtype=torch.float64))
    x, lazy = cg_solve_async(op, ws.cuda(), 0.25, 0, bad.cuda(), torch.zeros_like(bad).cuda(), 1e-8, batched=False, hermitian=True)
    with pytest.raises(RuntimeError, match="not the transform of real data"):
        int(lazy)
    assert bool(torch.isnan(x.real).all())
    ws_bad = ws.clone()
    ws_bad[5] = ws_bad[5] + 0.1j
    beta, lazy = cg_solve_mean_async(op, ws_bad.cuda(), 0.25, None, b.cuda(), 1e-8)
    with pytest.raises(RuntimeError):
        int(lazy)
    # the general entry takes the same data without complaint
    x, lazy = cg_solve_async(op, ws.cuda(), 0.25, 0, bad.cuda(), torch.zeros_like(bad).cuda(), 1e-8, batched=False)
    assert int(lazy) > 0 and bool(torch.isfinite(x.real).all())


@pytest.mark.parametrize("mtot,precond,variant,tol", [(41, True, 0, 1e-8), (41, False, 0, 1e-6), (63, True, 0, 1e-8), (71, True, 0, 1e-8),
                                                       (71, False, 1, 1e-6), (131, True, 0, 1e-6), (33, True, 0, 1e-10)])
def test_cooperative_hermitian_solve_matches_complex_kernel_and_oracle(mtot, precond, variant, tol, monkeypatch):
    """Round 3: the Hermitian specialisation of the cooperative 128^2..512^2 solve (cg_coop2d_herm_kernel: rows k0 >= 0, packed
    column pairs) against the complex cooperative kernel and the oracle's cg.py restatement: iteration counts, solution, exact
    conjugate symmetry of the output, the TRUE residual (a recurrence that drifts shows there)."""
    from efgp_hip import ToeplitzOp, cg_solve_async
    from oracle import efgp_oracle as O
    v, T, ws, b = _system(mtot, 5, N=900)
    op = ToeplitzOp(v.cuda())
    assert min(op.fft_shape) >= 128
    centre = float(v[tuple((s - 1) // 2 for s in v.shape)].real)
    sig = 0.25
    diag = (centre * ws.abs().pow(2).real + sig) if precond else None
    dd = diag.cuda() if diag is not None else None
    xh, lazy_h = cg_solve_async(op, ws.cuda(), sig, variant, b.cuda(), None, tol, diag=dd, batched=False, hermitian=True)
    it_h = int(lazy_h)
    monkeypatch.setenv("EFGP_NO_CG_COOP_HERM", "1")
    xc, lazy_c = cg_solve_async(op, ws.cuda(), sig, variant, b.cuda(), None, tol, diag=dd, batched=False, hermitian=True)
    it_c = int(lazy_c)
    monkeypatch.delenv("EFGP_NO_CG_COOP_HERM")
    # long solves of these random systems decorrelate (the reference's own stopping index moves by several per cent under a
    # 1e-13 perturbation, oracle/sensitivity_r2.py): a few per cent beyond ~400 iterations
    slack = 1 + it_c // (200 if precond else 50) if it_c < 400 else int(0.06 * it_c)
    xtol = (100 if precond else 1e4) * tol
    assert abs(it_h - it_c) <= slack, (it_h, it_c)
    assert _rel(xh, xc) < xtol
    A = O.make_A_mean(ws, T, sig) if variant == 0 else O.make_A_var(ws, T, sig)
    xo, ito = O.cg_single(A, b, torch.zeros_like(b), tol, diag=diag)
    assert abs(it_h - ito) <= slack, (it_h, ito)
    assert _rel(xh, xo) < xtol
    true_h = float(torch.linalg.norm(A(xh.cpu()) - b) / torch.linalg.norm(b))
    true_o = float(torch.linalg.norm(A(xo) - b) / torch.linalg.norm(b))
    assert true_h < 1.05 * true_o + 0.1 * tol, (true_h, true_o)
    xq = xh.cpu().reshape(mtot, mtot)
    assert torch.equal(torch.flip(xq, dims=(0, 1)).conj()[: mtot // 2], xq[: mtot // 2])


def test_cooperative_hermitian_batched_warm_start_and_refusal():
    """Many systems (one workgroup per system, no grid barrier), non-zero Hermitian start vectors, per-row stopping; a
    right-hand side that is not conjugate-even is refused in the data (NaN, -2) while the general entry solves it."""
    from efgp_hip import ToeplitzOp, cg_solve_async
    from oracle import efgp_oracle as O
    mtot, rows = 41, 40
    v, T, ws, B = _system(mtot, 6, N=900, rows=rows)
    op = ToeplitzOp(v.cuda())
    B = B * torch.logspace(-2, 2, rows, dtype=torch.float64)[:, None]           # rows stop at different iterations
    g = torch.Generator().manual_seed(2)
    X0 = 0.1 * _herm(torch.complex(torch.randn(rows, mtot, mtot, generator=g, dtype=torch.float64),
                                   torch.randn(rows, mtot, mtot, generator=g, dtype=torch.float64))).reshape(rows, -1)
    xh, lazy = cg_solve_async(op, ws.cuda(), 0.25, 1, B.cuda(), X0.cuda(), 1e-8, batched=True, hermitian=True)
    its = lazy.rows
    A = O.make_A_var(ws, T, 0.25)
    for r in (0, 7, rows - 1):
        xo, ito = O.cg_batched(A, B[r:r + 1], X0[r:r + 1], 1e-8)
        assert abs(its[r] + 1 - ito) <= 2, (r, its[r], ito)                      # cg.py:243 counts the breaking pass
        assert _rel(xh[r], xo[0]) < 1e-6
    bad = torch.complex(torch.randn(mtot * mtot, generator=g, dtype=torch.float64), torch.randn(mtot * mtot, generator=g, dtype=torch.float64))
    x, lz = cg_solve_async(op, ws.cuda(), 0.25, 0, bad.cuda(), None, 1e-8, batched=False, hermitian=True)
    with pytest.raises(RuntimeError, match="not the transform of real data"):
        int(lz)
    assert bool(torch.isnan(x.real).all())
    x, lz = cg_solve_async(op, ws.cuda(), 0.25, 0, bad.cuda(), None, 1e-8, batched=False)
    assert int(lz) > 0 and bool(torch.isfinite(x.real).all())


def test_residual_history_from_the_hermitian_kernel(monkeypatch):
    """Row 0's |r_i| / |b| per iteration (efgp_cg_record_history) against the oracle's curve: iterate-level parity.  Rounding
    differences grow along the recurrence at a rate set by the system, not the kernel: the Hermitian kernel's curve must
    stay as close to the oracle's as the complex kernel's does (same order of magnitude at every third of the solve)."""
    from efgp_hip import ToeplitzOp, cg_solve_mean_async, cg_residual_history
    from oracle import efgp_oracle as O
    mtot = 23
    v, T, ws, fy = _system(mtot, 6)
    vd = v.cuda()
    centre = vd[tuple((s - 1) // 2 for s in vd.shape)].real
    op = ToeplitzOp(vd)
    curves = {}
    for mode in ("herm", "complex"):
        if mode == "complex":
            monkeypatch.setenv("EFGP_NO_CG_HERM", "1")
        with cg_residual_history(op.dev, 1024) as rec:
            beta, lazy = cg_solve_mean_async(op, ws.cuda(), 0.25, centre, fy.cuda(), 1e-8)
            its = int(lazy)
        curves[mode] = (rec.values()[:its], its)
    monkeypatch.delenv("EFGP_NO_CG_HERM")
    h, its = curves["herm"]
    assert float(h[-1]) < 1e-8 and bool((h[:-1] >= 1e-8).all())
    rhs = ws * fy
    res = []
    diag = float(centre) * ws.abs().pow(2).real + 0.25
    xo, ito = O.cg_single(O.make_A_mean(ws, T, 0.25), rhs, torch.zeros_like(rhs), 1e-8, diag=diag, history=res)
    ref = torch.tensor(res, dtype=torch.float64)

    def dev(curve, lo, hi):
        hi = min(hi, curve.numel(), ref.numel())
        return float(((curve[lo:hi] - ref[lo:hi]).abs() / ref[lo:hi]).max())
    n = min(its, curves["complex"][1], ito)
    print(f"\nits herm {its} complex {curves['complex'][1]} oracle {ito}")
    assert dev(h, 0, 20) < 1e-10
    for lo, hi in ((0, n // 3), (n // 3, 2 * n // 3), (2 * n // 3, n)):
        dh, dc = dev(h, lo, hi), dev(curves["complex"][0], lo, hi)
        print(f"iterations {lo}..{hi}: deviation from the oracle's curve  hermitian {dh:.1e}  complex {dc:.1e}")
        assert dh < 30 * dc + 1e-12, (lo, hi, dh, dc)
    assert abs(its - ito) <= 1 + ito // 100


# ---- round 3: the Hermitian 3-D line iteration (cg3h_* kernels, planes k0 >= 0 only) -------------------------------------------
def _herm3(t):
    return 0.5 * (t + torch.flip(t, dims=(-3, -2, -1)).conj())


def _system3(ns, seed, N=160, rows=None):
    """Toeplitz vector of N random 3-D points on the (2 n_a - 1) lag box (anisotropic boxes: the central crop of the cubic one), a real
    even ws, Hermitian right-hand sides."""
    from oracle import efgp_oracle as O
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(N, 3, generator=g, dtype=torch.float64) * 2 - 1
    m = (max(ns) - 1) // 2
    v = O.conv_vector(x, 0.31, m)
    c = 2 * m
    v = v[tuple(slice(c - (n - 1), c + n) for n in ns)].contiguous()
    w = torch.exp(-2.5 * torch.rand(*ns, generator=g, dtype=torch.float64))
    ws = (0.5 * (w + torch.flip(w, dims=(0, 1, 2)))).reshape(-1).to(torch.complex128)
    shape = tuple(ns) if rows is None else (rows, *ns)
    b = _herm3(torch.complex(torch.randn(shape, generator=g, dtype=torch.float64), torch.randn(shape, generator=g, dtype=torch.float64)))
    b = b.reshape(-1) if rows is None else b.reshape(rows, -1)
    return v, O.Toeplitz(v), ws, b


@pytest.mark.parametrize("ns,precond,variant,tol,sig", [((19, 19, 19), True, 0, 1e-8, 0.3), ((19, 19, 19), False, 1, 1e-6, 0.3),
                                                        ((17, 19, 21), True, 0, 1e-10, 0.3), ((33, 33, 33), True, 0, 1e-8, 30.0),
                                                        ((17, 33, 19), True, 1, 1e-8, 0.3)])
def test_hermitian_3d_iteration_matches_general_iteration_and_oracle(ns, precond, variant, tol, sig, monkeypatch):
    """Half the planes, packed column pairs along dim 0, real centred spectrum -- against the general 3-D line iteration and the
    oracle's cg.py restatement: iteration counts, solutions, exact conjugate symmetry of the result, the true residual."""
    from efgp_hip import ToeplitzOp, cg_solve
    from oracle import efgp_oracle as O
    v, T, ws, b = _system3(ns, 11)
    op = ToeplitzOp(v.cuda())
    assert len(op.fft_shape) == 3 and min(op.fft_shape) >= 64
    centre = float(v[tuple((s - 1) // 2 for s in v.shape)].real)
    diag = (centre * ws.abs().pow(2).real + sig) if precond else None
    dd = diag.cuda() if diag is not None else None
    xh, it_h, _ = cg_solve(op, ws.cuda(), sig, variant, b.cuda(), None, tol, diag=dd, batched=False, hermitian=True)
    xc, it_c, _ = cg_solve(op, ws.cuda(), sig, variant, b.cuda(), None, tol, diag=dd, batched=False)
    monkeypatch.setenv("EFGP_NO_CG_HERM3", "1")
    xg, it_g, _ = cg_solve(op, ws.cuda(), sig, variant, b.cuda(), None, tol, diag=dd, batched=False, hermitian=True)
    monkeypatch.delenv("EFGP_NO_CG_HERM3")
    assert it_g == it_c and torch.equal(xg, xc)                   # the switch falls back to the general iteration
    # long solves decorrelate (the reference's own stopping index moves by several per cent under a 1e-13 perturbation)
    slack = max(1 + it_c // (200 if precond else 50), int(0.03 * it_c)) if it_c < 400 else int(0.06 * it_c)
    xtol = (100 if precond else 1e4) * tol
    print(f"\n3-D {ns}: iterations hermitian {it_h}, general {it_c}; solutions differ by {_rel(xh, xc):.2e}")
    assert abs(it_h - it_c) <= slack, (it_h, it_c)
    assert _rel(xh, xc) < xtol
    A = O.make_A_mean(ws, T, sig) if variant == 0 else O.make_A_var(ws, T, sig)
    xo, ito = O.cg_single(A, b, torch.zeros_like(b), tol, diag=diag)
    assert abs(it_h - ito) <= slack, (it_h, ito)
    assert _rel(xh, xo) < xtol
    true_h = float(torch.linalg.norm(A(xh.cpu()) - b) / torch.linalg.norm(b))
    true_o = float(torch.linalg.norm(A(xo) - b) / torch.linalg.norm(b))
    assert true_h < 1.05 * true_o + 0.1 * tol, (true_h, true_o)
    xq = xh.cpu().reshape(*ns)
    assert torch.equal(torch.flip(xq, dims=(0, 1, 2)).conj()[: ns[0] // 2], xq[: ns[0] // 2])


def test_hermitian_3d_batched_warm_start_and_refusal():
    """Several systems with per-row stopping and non-zero Hermitian start vectors; data that break the promise are refused
    (ValueError, nothing written) while the general entry takes them."""
    from efgp_hip import ToeplitzOp, cg_solve
    from oracle import efgp_oracle as O
    ns, rows = (19, 19, 19), 5
    v, T, ws, B = _system3(ns, 12, rows=rows)
    op = ToeplitzOp(v.cuda())
    B = B * torch.logspace(-2, 2, rows, dtype=torch.float64)[:, None]
    g = torch.Generator().manual_seed(3)
    X0 = 0.1 * _herm3(torch.complex(torch.randn(rows, *ns, generator=g, dtype=torch.float64),
                                    torch.randn(rows, *ns, generator=g, dtype=torch.float64))).reshape(rows, -1)
    xh, it_h, rows_h = cg_solve(op, ws.cuda(), 0.3, 1, B.cuda(), X0.cuda(), 1e-8, batched=True, hermitian=True)
    xc, it_c, rows_c = cg_solve(op, ws.cuda(), 0.3, 1, B.cuda(), X0.cuda(), 1e-8, batched=True)
    assert abs(it_h - it_c) <= 1 and all(abs(a - b) <= 1 for a, b in zip(rows_h, rows_c)), (rows_h, rows_c)
    assert len(set(rows_h)) > 1                                                   # the rows did stop at different iterations
    A = O.make_A_var(ws, T, 0.3)
    for r in (0, rows - 1):
        xo, ito = O.cg_batched(A, B[r:r + 1], X0[r:r + 1], 1e-8)
        assert abs(rows_h[r] + 1 - ito) <= 2, (r, rows_h[r], ito)
        assert _rel(xh[r], xo[0]) < 1e-6
        assert _rel(xh[r], xc[r]) < 1e-7
    bad = torch.complex(torch.randn(B.shape[1], generator=g, dtype=torch.float64), torch.randn(B.shape[1], generator=g, dtype=torch.float64))
    with pytest.raises(ValueError, match="conjugate-even"):
        cg_solve(op, ws.cuda(), 0.3, 0, bad.cuda(), None, 1e-8, batched=False, hermitian=True)
    ws_bad = ws.clone()
    ws_bad[7] = ws_bad[7] + 0.1
    with pytest.raises(ValueError, match="conjugate-even"):
        cg_solve(op, ws_bad.cuda(), 0.3, 0, B[0].cuda(), None, 1e-8, batched=False, hermitian=True)
    x, it, _ = cg_solve(op, ws.cuda(), 0.3, 0, bad.cuda(), None, 1e-8, batched=False)
    assert it > 0 and bool(torch.isfinite(x.real).all())
    # a grid outside the line kernels' range (16 x 128 x 64): the promise is accepted and the general iteration runs
    v2, T2, ws2, b2 = _system3((5, 33, 17), 13)
    op2 = ToeplitzOp(v2.cuda())
    xa, ia, _ = cg_solve(op2, ws2.cuda(), 0.3, 0, b2.cuda(), None, 1e-8, batched=False, hermitian=True)
    xb, ib, _ = cg_solve(op2, ws2.cuda(), 0.3, 0, b2.cuda(), None, 1e-8, batched=False)
    assert ia == ib and torch.equal(xa, xb)


@pytest.mark.parametrize("mtot,precond,tol", [(23, True, 1e-10), (23, False, 1e-6), (21, True, 1e-8), (19, True, 1e-12), (13, True, 1e-8),
                                              (11, False, 1e-8), (5, True, 1e-10), (1, True, 1e-10), (25, True, 1e-8)])
def test_smallest_circulant_grid_48_equals_64(mtot, precond, tol, monkeypatch):
    """Blocks of up to 23 x 23 modes: the Hermitian solve runs on the 48 x 48 circulant grid (cg_herm48_kernel; any F >= 2 n - 1
    embeds the Toeplitz product exactly, efgpnd.py:1266-1271) -- against the 64 x 64 kernel (EFGP_NO_CG48) and the oracle:
    same iteration counts, solutions to cond x tol, true residual at the oracle's.  mtot 25 (2 n - 1 = 49) must stay on 64."""
    from efgp_hip import ToeplitzOp, cg_solve_mean_async, kernel_timing, kernel_timing_read
    from oracle import efgp_oracle as O
    v, T, ws, fy = _system(mtot, 17)
    vd = v.cuda()
    centre = vd[tuple((s - 1) // 2 for s in vd.shape)].real
    op = ToeplitzOp(vd)
    assert tuple(op.fft_shape) == ((64, 64) if mtot >= 17 else tuple(op.fft_shape))      # the reference's grid is what is reported
    assert op.cg_shape(hermitian=True) == ([48, 48] if mtot <= 23 else [64, 64]) and op.cg_shape() == [64, 64]
    beta, lazy = cg_solve_mean_async(op, ws.cuda(), 0.25, centre if precond else None, fy.cuda(), tol)
    it48 = int(lazy)
    monkeypatch.setenv("EFGP_NO_CG48", "1")
    beta64, lazy64 = cg_solve_mean_async(op, ws.cuda(), 0.25, centre if precond else None, fy.cuda(), tol)
    it64 = int(lazy64)
    monkeypatch.delenv("EFGP_NO_CG48")
    slack = 1 + it64 // (200 if precond else 50)
    xtol = (100 if precond else 1e4) * tol
    assert abs(it48 - it64) <= slack, (it48, it64)
    assert _rel(beta, beta64) < xtol
    if mtot == 25:
        assert torch.equal(beta, beta64)              # same kernel both times
    rhs = ws * fy
    diag = (float(centre) * ws.abs().pow(2).real + 0.25) if precond else None
    A = O.make_A_mean(ws, T, 0.25)
    xo, ito = O.cg_single(A, rhs, torch.zeros_like(rhs), tol, diag=diag)
    assert abs(it48 - ito) <= slack and _rel(beta, xo) < xtol
    true_h = float(torch.linalg.norm(A(beta.cpu()) - rhs) / torch.linalg.norm(rhs))
    true_o = float(torch.linalg.norm(A(xo) - rhs) / torch.linalg.norm(rhs))
    assert true_h < 1.05 * true_o + 0.1 * tol, (true_h, true_o)
    bq = beta.cpu().reshape(mtot, mtot)
    assert torch.equal(torch.flip(bq, dims=(0, 1)).conj()[: mtot // 2], bq[: mtot // 2])
