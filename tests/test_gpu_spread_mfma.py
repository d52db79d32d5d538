"""GPU parity of the MFMA spreader over the per-model point layout (csrc/spread_mfma.hip, points_layout.hip).

Plans made on a `PointSet` (efgp_nufft_create_on) must give the transforms of plain plans: both are checked against
the oracle's exact NUDFT (the sum FINUFFT approximates, efgpnd.py:1496-1499) at the requested tolerance, against
each other, and for bitwise reproducibility (exact integer accumulation across runs of the same layout).
Edge cases: ragged chunk tails, clustered points (all in one cell), shifted centre, narrow windows, one band /
many bands, strengths through the attached sorted copy and through the permutation, generated probes.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a = a.detach().cpu()
    b = b.detach().cpu()
    return float(torch.linalg.norm((a - b).reshape(-1)) / torch.linalg.norm(b.reshape(-1)))


def _data(N, seed, lo=-1.0, hi=1.0, cluster=False):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(N, 2, generator=g, dtype=torch.float64) * (hi - lo) + lo
    if cluster:                                  # half of the points inside one fine-grid cell, duplicates included
        x[: N // 2] = x[0] + 1e-4 * torch.rand(N // 2, 2, generator=g, dtype=torch.float64)
        x[1] = x[0]
    y = torch.randn(N, generator=g, dtype=torch.float64)
    return x, y


@pytest.mark.parametrize("N,h,nm_y,nm_o,tol,kw", [
    (40001, 0.31, 23, 45, 6e-8, {}),                       # the fit-time pair of BASELINE configs[1], ragged tail
    (65536, 0.31, 23, 45, 1e-4, {}),                       # narrow window (W < 8)
    (50000, 0.12, 71, 141, 6e-8, {}),                      # grid beyond LDS (configs[3] hard case), several bands
    (40000, 0.31, 23, 45, 6e-8, {"cluster": True}),
    (40000, 0.05, 23, 45, 6e-8, {"lo": 2.0, "hi": 9.0}),   # shifted box, one band
])
@pytest.mark.parametrize("band_cells", [8, 1])
def test_pair_on_layout_vs_exact(N, h, nm_y, nm_o, tol, kw, band_cells, monkeypatch):
    """band_cells: the two tile geometries of the spreader -- band levels up to 8 fine cells high (16 tile columns) and
    one-cell bands (W + 1 columns, three waves per SIMD; chosen by itself from 192 points per run on, forced here)."""
    from efgp_hip import NufftPlan, PointSet
    from oracle import efgp_oracle as O
    monkeypatch.setenv("EFGP_MFMA_BAND_CELLS", str(band_cells))
    x, y = _data(N, 3, **kw)
    xd, yd = x.cuda(), y.cuda()
    pts = PointSet(xd, values=yd)
    plan = NufftPlan(xd, h, tol, points=pts)
    Fy, v = plan.type1_pair(yd, (nm_y, nm_y), (nm_o, nm_o))
    Fy2, v2 = plan.type1_pair(yd, (nm_y, nm_y), (nm_o, nm_o))
    assert torch.equal(Fy, Fy2) and torch.equal(v, v2)          # exact integer accumulation: bitwise reproducible
    ref_y = O.nudft_type1(x, h, y, (nm_y, nm_y))
    ref_o = O.nudft_type1(x, h, torch.ones(N, dtype=torch.float64), (nm_o, nm_o))
    assert _rel(Fy, ref_y) < 2 * tol
    assert _rel(v, ref_o) < 2 * tol
    plain = NufftPlan(xd, h, tol)
    Fy0, v0 = plain.type1_pair(yd, (nm_y, nm_y), (nm_o, nm_o))
    assert _rel(Fy, Fy0) < 4 * tol and _rel(v, v0) < 4 * tol
    assert abs(float(v[nm_o // 2, nm_o // 2].real) - N) < 1e-6 * N


def test_layout_path_is_the_one_that_runs():
    """The layout path must actually launch (not silently fall back): with it disabled the results differ in the last
    bits (different summation), with it enabled two layouts over the same points agree bit for bit."""
    from efgp_hip import NufftPlan, PointSet
    x, y = _data(60000, 9)
    xd, yd = x.cuda(), y.cuda()
    a = NufftPlan(xd, 0.31, 6e-8, points=PointSet(xd, values=yd)).type1_pair(yd, (23, 23), (45, 45))
    b = NufftPlan(xd, 0.31, 6e-8, points=PointSet(xd, values=yd)).type1_pair(yd, (23, 23), (45, 45))
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    os.environ["EFGP_NO_MFMA_SPREAD"] = "1"
    try:
        c = NufftPlan(xd, 0.31, 6e-8, points=PointSet(xd, values=yd)).type1_pair(yd, (23, 23), (45, 45))
    finally:
        del os.environ["EFGP_NO_MFMA_SPREAD"]
    assert not torch.equal(a[0], c[0])
    assert _rel(a[0], c[0]) < 1e-6


def test_rows_and_probes_on_layout():
    """Real rows in user order (fetched through the permutation), complex rows, generated Rademacher probes."""
    from efgp_hip import NufftPlan, PointSet, rademacher_fill
    from oracle import efgp_oracle as O
    N, h, nm, tol = 45000, 0.31, 23, 1e-7
    x, y = _data(N, 5)
    xd = x.cuda()
    pts = PointSet(xd, values=y.cuda())
    plan = NufftPlan(xd, h, tol, points=pts)
    g = torch.Generator().manual_seed(1)
    Z = torch.randn(3, N, generator=g, dtype=torch.float64)
    out = plan.type1(Z.cuda(), (nm, nm))
    for b in range(3):
        assert _rel(out[b], O.nudft_type1(x, h, Z[b], (nm, nm))) < 2 * tol
    c = torch.complex(Z[0], Z[1])
    outc = plan.type1(c.cuda(), (nm, nm))
    assert _rel(outc, O.nudft_type1(x, h, c, (nm, nm))) < 2 * tol
    one = plan.type1(y.cuda(), (nm, nm))                       # the attached array as a single row: sorted copy
    assert _rel(one, O.nudft_type1(x, h, y, (nm, nm))) < 2 * tol
    seed, off = 1234567, 1000
    R = rademacher_fill(xd.device, seed, 3, N, index_offset=off)
    FR = plan.type1_rademacher(seed, 3, (nm, nm), index_offset=off)
    plain = NufftPlan(xd, h, tol)
    assert _rel(FR, plain.type1(R, (nm, nm))) < 4 * tol
    assert torch.equal(FR, plan.type1_rademacher(seed, 3, (nm, nm), index_offset=off))


def test_model_fit_uses_layout_and_matches_plain(monkeypatch):
    """EFGPND end to end: posterior mean with the layout path on and off agree to the north-star tolerance."""
    from efgpnd import EFGPND
    from kernels.squared_exponential import SquaredExponential
    x, _ = _data(80000, 21)
    g = torch.Generator().manual_seed(2)
    y = torch.sin(3 * x[:, 0]) * torch.cos(4 * x[:, 1]) + 0.2 * torch.randn(x.shape[0], generator=g, dtype=torch.float64)
    xn = torch.rand(500, 2, generator=g, dtype=torch.float64) * 2 - 1

    def run():
        k = SquaredExponential(dimension=2, init_lengthscale=0.2, init_variance=2.0)
        m = EFGPND(x.cuda(), y.cuda(), k, sigmasq=0.2, eps=1e-4, nufft_eps=1e-7, estimate_params=False,
                   opts={"cg_tolerance": 1e-10, "point_layout": True})     # layout from the first fit on ("auto": from the second pass)
        mean, _ = m.predict(xn.cuda(), return_variance=False)
        V = torch.ones(3, m.last_fit_stats["feature_count"], dtype=torch.float64)
        V[1, ::2] = -1
        V[2, ::3] = -1
        grad = m.compute_gradients(trace_samples=3, cg_tol=1e-10, probe_seed=77, probes_V=V)
        return mean.cpu(), grad.detach().cpu(), m
    mean_a, grad_a, m = run()
    assert m._devdata["points"] is not None
    # "auto": a model that is fitted once never builds the layout; its second pass over the points does
    k = SquaredExponential(dimension=2, init_lengthscale=0.2, init_variance=2.0)
    lazy = EFGPND(x.cuda(), y.cuda(), k, sigmasq=0.2, eps=1e-4, nufft_eps=1e-7, estimate_params=False, opts={"cg_tolerance": 1e-10})
    mean_l, _ = lazy.predict(xn.cuda(), return_variance=False)
    assert lazy._devdata["points"] is None
    assert float((mean_l.cpu() - mean_a).abs().max() / mean_a.abs().max()) < 1e-5
    lazy.fit()
    assert lazy._devdata["points"] is not None
    monkeypatch.setenv("EFGP_NO_MFMA_SPREAD", "1")
    mean_b, grad_b, _ = run()
    assert float((mean_a - mean_b).abs().max() / mean_b.abs().max()) < 1e-5
    assert float((grad_a - grad_b).abs().max() / grad_b.abs().max()) < 1e-4


@pytest.mark.parametrize("T", [4, 5, 7, 9])
def test_many_probe_rows_in_one_pass(T):
    """Many probe rows (pairs of real rows per fine grid, an odd last row alone): generated probes and rows from memory, even
    and odd counts, against the plain plan and the exact sums."""
    from efgp_hip import NufftPlan, PointSet, rademacher_fill
    from oracle import efgp_oracle as O
    N, h, nm, tol = 50000, 0.31, 23, 1e-5
    x, y = _data(N, 31 + T)
    xd = x.cuda()
    plan = NufftPlan(xd, h, tol, points=PointSet(xd))
    plain = NufftPlan(xd, h, tol)
    seed, off = 99 + T, 7
    FR = plan.type1_rademacher(seed, T, (nm, nm), index_offset=off)
    R = rademacher_fill(xd.device, seed, T, N, index_offset=off)
    assert _rel(FR, plain.type1(R, (nm, nm))) < 4 * tol
    assert torch.equal(FR, plan.type1_rademacher(seed, T, (nm, nm), index_offset=off))
    g = torch.Generator().manual_seed(T)
    Z = torch.randn(T, N, generator=g, dtype=torch.float64)
    out = plan.type1(Z.cuda(), (nm, nm))
    for b in (0, T // 2, T - 1):
        assert _rel(out[b], O.nudft_type1(x, h, Z[b], (nm, nm))) < 2 * tol
    C = torch.complex(Z[:3], Z[1:4])
    outc = plan.type1(C.cuda(), (nm, nm))
    assert _rel(outc, plain.type1(C.cuda(), (nm, nm))) < 4 * tol


@pytest.mark.parametrize("nm,big,h,tol", [((23, 23), (45, 45), 0.31, 6e-8), ((17, 29), (33, 57), 0.22, 1e-5), ((8, 12), (15, 23), 0.9, 1e-9),
                                           ((63, 63), (63, 63), 0.12, 1e-5)])
def test_grid_to_modes_launch_equals_fft_sequence(nm, big, h, tol, monkeypatch):
    """Small 2-D grids go from the spreader's int64 accumulator to the modes in one launch (pruned dense DFT, Hermitian split,
    correction factors: grid_to_modes_kernel) instead of reduce | rocFFT rows | rocFFT columns | deconvolve.  Same sums in
    another order: every variant must agree with the FFT sequence to rounding -- the fit's (F*y, Toeplitz vector) pair on two
    boxes, real rows in pairs (even and odd counts), a lone real row, complex rows, generated probes, both mode orders -- and
    leave the accumulator clean for the next pass (each call below starts from what the previous one left)."""
    from efgp_hip import NufftPlan, PointSet
    N = 50000
    x, y = _data(N, 17)
    xd, yd = x.cuda(), y.cuda()
    g = torch.Generator().manual_seed(4)
    Z = torch.randn(5, N, generator=g, dtype=torch.float64).cuda()
    Cx = torch.complex(Z[:2], Z[2:4]).contiguous()

    def run():
        plan = NufftPlan(xd, h, tol, points=PointSet(xd, values=yd))
        outs = list(plan.type1_pair(yd, nm, big))
        outs.append(plan.type1(Z, nm))                         # 5 real rows: two pair grids + a lone row
        outs.append(plan.type1(Z[:4], nm, modeord=1))          # pairs only, FFT mode order
        outs.append(plan.type1(Z[0], nm))
        outs.append(plan.type1(Cx, nm))                        # complex rows
        outs.append(plan.type1(Cx, nm, modeord=1, isign=1))
        outs.append(plan.type1_rademacher(99, 5, nm, index_offset=3))
        outs.append(plan.type1_rademacher(99, 1, nm, modeord=1))
        outs.extend(plan.type1_pair(yd, nm, big))              # again, after all of the above
        return outs

    a = run()
    monkeypatch.setenv("EFGP_NO_GRID_TO_MODES", "1")
    b = run()
    monkeypatch.delenv("EFGP_NO_GRID_TO_MODES")
    for u, v in zip(a, b):
        assert u.shape == v.shape
        assert float((u - v).abs().max() / v.abs().max()) < 1e-12
    assert torch.equal(a[0], a[-2]) and torch.equal(a[1], a[-1])       # nothing left behind in the accumulator


@pytest.mark.parametrize("N,nm,big,h,tol", [(3000, (23, 23), (45, 45), 0.31, 6e-8), (20000, (17, 29), (33, 57), 0.22, 1e-5),
                                            (40000, (31, 31), (61, 61), 0.2, 1e-7)])
def test_grid_to_modes_behind_the_other_spreaders(N, nm, big, h, tol, monkeypatch):
    """Plans without a point layout / with few points spread through the LDS-atomic kernels and reduce their slabs to a complex
    grid; small grids then take the same one-launch pruned DFT (reading that grid) instead of rocFFT + deconvolve."""
    from efgp_hip import NufftPlan
    x, y = _data(N, 23)
    xd, yd = x.cuda(), y.cuda()
    g = torch.Generator().manual_seed(6)
    Z = torch.randn(3, N, generator=g, dtype=torch.float64).cuda()
    Cx = torch.complex(Z[0], Z[1]).contiguous()

    def run():
        plan = NufftPlan(xd, h, tol)
        return list(plan.type1_pair(yd, nm, big)) + [plan.type1(Z, nm), plan.type1(Cx, nm, modeord=1), plan.type1_rademacher(5, 3, nm),
                                                     plan.type1(Cx, nm, isign=1)]

    a = run()
    monkeypatch.setenv("EFGP_NO_GRID_TO_MODES", "1")
    b = run()
    monkeypatch.delenv("EFGP_NO_GRID_TO_MODES")
    for u, v in zip(a, b):
        assert u.shape == v.shape and float((u - v).abs().max() / v.abs().max()) < 1e-12
