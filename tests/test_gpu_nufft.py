"""GPU parity: HIP NUFFT (through the C ABI) vs the oracle's exact NUDFT.

Mirrors the reference's notebook checks (efgpnd_sanity_checks.ipynb cell 14: NUFFT vs explicit F).
Tolerance: the requested NUFFT tolerance `tol` (relative l2), stated per test.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a = a.detach().cpu()
    b = b.detach().cpu()
    return float(torch.linalg.norm((a - b).reshape(-1)) / torch.linalg.norm(b.reshape(-1)))


def _points(N, d, seed, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(N, d, generator=g, dtype=torch.float64) * (hi - lo) + lo


@pytest.mark.parametrize("d,nm,tol", [(1, 35, 1e-6), (1, 69, 1e-9), (2, 23, 1e-4), (2, 45, 6e-8), (2, 29, 1e-12),
                                      (3, 11, 1e-5), (3, 21, 1e-9), (2, 141, 1e-7), (2, 24, 1e-6)])
@pytest.mark.parametrize("complex_c", [False, True])
def test_type1_vs_exact(d, nm, tol, complex_c):
    from efgp_hip import NufftPlan
    from oracle import efgp_oracle as O
    N = 3000
    x = _points(N, d, 10 + d)
    h = 0.37
    g = torch.Generator().manual_seed(5)
    c = torch.randn(N, generator=g, dtype=torch.float64)
    if complex_c:
        c = torch.complex(c, torch.randn(N, generator=g, dtype=torch.float64))
    plan = NufftPlan(x.cuda(), h, tol)
    out = plan.type1(c.cuda(), (nm,) * d)
    ref = O.nudft_type1(x, h, c, (nm,) * d)
    assert out.shape == ref.shape
    assert _rel(out, ref) < 2 * tol + 1e-13


@pytest.mark.parametrize("d,nm,tol", [(1, 35, 1e-6), (2, 23, 1e-4), (2, 45, 6e-8), (3, 11, 1e-5), (2, 141, 1e-9), (2, 24, 1e-6),
                                      (1, 8, 1e-8)])
@pytest.mark.parametrize("real_only", [False, True])
def test_type2_vs_exact(d, nm, tol, real_only):
    from efgp_hip import NufftPlan
    from oracle import efgp_oracle as O
    N = 2500
    x = _points(N, d, 20 + d, -3.0, 5.0)
    h = 0.21
    g = torch.Generator().manual_seed(6)
    f = torch.complex(torch.randn(nm ** d, generator=g, dtype=torch.float64),
                      torch.randn(nm ** d, generator=g, dtype=torch.float64))
    plan = NufftPlan(x.cuda(), h, tol)
    out = plan.type2(f.cuda(), (nm,) * d, real_only=real_only)
    ref = O.nudft_type2(x, h, f, (nm,) * d)
    if real_only:
        ref = ref.real
    assert _rel(out, ref) < 2 * tol + 1e-13


def test_type2_fft_order_and_batch():
    """modeord=1 (efgpnd.py:1679) and batched inputs (efgpnd.py:1531-1536)."""
    from efgp_hip import NufftPlan
    from oracle import efgp_oracle as O
    N, d, nm = 1500, 2, 45
    x = _points(N, d, 3)
    h = 0.3
    g = torch.Generator().manual_seed(7)
    f = torch.complex(torch.randn(3, nm, nm, generator=g, dtype=torch.float64),
                      torch.randn(3, nm, nm, generator=g, dtype=torch.float64))
    plan = NufftPlan(x.cuda(), h, 1e-9)
    out = plan.type2(f.cuda(), (nm, nm), modeord=1)
    ref = torch.stack([O.nudft_type2(x, h, f[b], (nm, nm), fft_order=True) for b in range(3)])
    assert out.shape == (3, N)
    assert _rel(out, ref) < 5e-9
    # modes scaled inside the transform (F (ws * beta)): same as scaling first; shared by the batch rows
    sc = torch.complex(torch.randn(nm, nm, generator=g, dtype=torch.float64), torch.randn(nm, nm, generator=g, dtype=torch.float64))
    for ro in (False, True):
        a = plan.type2(f.cuda(), (nm, nm), real_only=ro, mode_scale=sc.cuda())
        b = plan.type2((f * sc).cuda(), (nm, nm), real_only=ro)
        assert _rel(a, b) < 1e-13
    with pytest.raises(ValueError):
        plan.type2(f.cuda(), (nm, nm), mode_scale=sc[:3].cuda())


def test_type1_batched_and_pair():
    """batched strengths (efgpnd.py:183) and the fused (F*y, F*1) pass (efgpnd.py:786,789-790)."""
    from efgp_hip import NufftPlan
    from oracle import efgp_oracle as O
    N, d = 4000, 2
    x = _points(N, d, 4)
    h = 0.346
    g = torch.Generator().manual_seed(8)
    Z = (torch.randint(0, 2, (4, N), generator=g) * 2 - 1).to(torch.float64)
    plan = NufftPlan(x.cuda(), h, 1e-8)
    out = plan.type1(Z.cuda(), (23, 23))
    ref = O.nudft_type1(x, h, Z, (23, 23))
    assert _rel(out, ref) < 5e-8
    y = torch.randn(N, generator=g, dtype=torch.float64)
    Fy, v = plan.type1_pair(y.cuda(), (23, 23), (45, 45))
    assert _rel(Fy, O.nudft_type1(x, h, y, (23, 23))) < 5e-8
    assert _rel(v, O.conv_vector(x, h, 11)) < 5e-8
    v2 = plan.type1_ones((45, 45))
    assert _rel(v2, O.conv_vector(x, h, 11)) < 5e-8


def test_type1_edge_cases():
    """tiny and ragged inputs: N=1, N=0 batch rows, points exactly on grid cells, huge offsets."""
    from efgp_hip import NufftPlan
    from oracle import efgp_oracle as O
    x = torch.tensor([[0.0, 0.0]], dtype=torch.float64)
    plan = NufftPlan(x.cuda(), 0.5, 1e-9)
    out = plan.type1(torch.tensor([2.0], dtype=torch.float64).cuda(), (7, 5))
    assert _rel(out, O.nudft_type1(x, 0.5, torch.tensor([2.0], dtype=torch.float64), (7, 5))) < 1e-8
    x = torch.tensor([[-118.3, 47.1], [151.2, -33.9], [0.0, 1e3]], dtype=torch.float64)   # raw lon/lat-like
    c = torch.tensor([1.0, -2.0, 0.5], dtype=torch.float64)
    plan = NufftPlan(x.cuda(), 0.013, 1e-9)
    assert _rel(plan.type1(c.cuda(), (33, 33)), O.nudft_type1(x, 0.013, c, (33, 33))) < 1e-7


@pytest.mark.parametrize("d,nm,tol,N", [(2, 141, 1e-7, 40000), (3, 21, 1e-6, 36000), (3, 37, 1e-4, 50000), (2, 24, 1e-12, 40000)])
def test_type1_tiled_large_grid(d, nm, tol, N):
    """Fine grids beyond LDS with many points: tile-sorted LDS spreader (fixed point), incl. the fused pair."""
    from efgp_hip import NufftPlan
    from oracle import efgp_oracle as O
    x = _points(N, d, 40 + d, -1.0, 1.0)
    h = 0.45
    g = torch.Generator().manual_seed(9)
    c = torch.complex(torch.randn(2, N, generator=g, dtype=torch.float64), torch.randn(2, N, generator=g, dtype=torch.float64))
    plan = NufftPlan(x.cuda(), h, tol)
    out = plan.type1(c.cuda(), (nm,) * d)
    ref = O.nudft_type1(x, h, c, (nm,) * d)
    assert _rel(out, ref) < 2 * tol + 1e-13
    # bitwise reproducible (integer accumulation): a second call gives identical bits
    out2 = plan.type1(c.cuda(), (nm,) * d)
    assert torch.equal(out, out2)
    y = torch.randn(N, generator=g, dtype=torch.float64)
    small = (nm + 1) // 2 if ((nm + 1) // 2) % 2 == 1 else (nm + 1) // 2 - 1
    Fy, v = plan.type1_pair(y.cuda(), (small,) * d, (nm,) * d)
    assert _rel(Fy, O.nudft_type1(x, h, y, (small,) * d)) < 2 * tol + 1e-13
    assert _rel(v, O.nudft_type1(x, h, torch.ones(N, dtype=torch.float64), (nm,) * d)) < 2 * tol + 1e-13


def test_type1_lds_path_is_bitwise_reproducible():
    from efgp_hip import NufftPlan
    N = 50000
    x = _points(N, 2, 77)
    g = torch.Generator().manual_seed(10)
    y = torch.randn(N, generator=g, dtype=torch.float64)
    plan = NufftPlan(x.cuda(), 0.346, 1e-7)
    a = plan.type1_pair(y.cuda(), (23, 23), (45, 45))
    b = plan.type1_pair(y.cuda(), (23, 23), (45, 45))
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


# odd probe counts: 2-D pads the last row into one more pair grid (one pass); 1-D / 3-D and a lone probe run it as a single row
# whose hash index is shifted by last * row stride -- either way row r is hash(seed, r, index)
@pytest.mark.parametrize("d,nm,N,B", [(2, 23, 5000, 5), (2, 141, 40000, 2), (1, 35, 3000, 3), (3, 9, 2000, 1), (3, 9, 2000, 3),
                                      (2, 23, 5000, 1), (2, 23, 5000, 7)])
def test_type1_rademacher_equals_type1_of_filled_probes(d, nm, N, B):
    from efgp_hip import NufftPlan, rademacher_fill
    from oracle import efgp_oracle as O
    x = _points(N, d, 60 + d)
    plan = NufftPlan(x.cuda(), 0.37, 1e-9)
    out = plan.type1_rademacher(987654321, B, (nm,) * d, index_offset=17)
    Z = rademacher_fill(torch.device("cuda", 0), 987654321, B, N, index_offset=17)
    assert torch.all(Z.abs() == 1.0)
    assert _rel(out, O.nudft_type1(x, 0.37, Z.cpu(), (nm,) * d)) < 5e-9
    # real rows from memory take the same paired path and give the same bits
    assert torch.equal(plan.type1(Z, (nm,) * d), out) or _rel(plan.type1(Z, (nm,) * d), out) < 1e-13


@pytest.mark.parametrize("nm,tol,N", [(45, 6e-8, 600000), (23, 1e-5, 300000), (35, 1e-9, 400000)])
def test_type1_cell_sorted_register_path(nm, tol, N, monkeypatch):
    """2-D, many points per fine-grid cell: register accumulation over base-cell-sorted points (forced on
    here so that moderate N exercises it) against the exact transform and against the LDS path."""
    from efgp_hip import NufftPlan
    from oracle import efgp_oracle as O
    x = _points(N, 2, 91)
    g = torch.Generator().manual_seed(12)
    y = torch.randn(N, generator=g, dtype=torch.float64)
    small = 23 if nm > 23 else 11
    monkeypatch.setenv("EFGP_CELLSORT", "1")
    plan = NufftPlan(x.cuda(), 0.346, tol)
    Fy, v = plan.type1_pair(y.cuda(), (small, small), (nm, nm))
    Zf = plan.type1_rademacher(5, 3, (small, small))
    cplx = plan.type1(torch.complex(y, -2 * y).cuda(), (small, small))
    monkeypatch.setenv("EFGP_CELLSORT", "0")
    plan2 = NufftPlan(x.cuda(), 0.346, tol)
    Fy2, v2 = plan2.type1_pair(y.cuda(), (small, small), (nm, nm))
    Zf2 = plan2.type1_rademacher(5, 3, (small, small))
    assert _rel(Fy, Fy2) < 4 * tol and _rel(v, v2) < 4 * tol and _rel(Zf, Zf2) < 4 * tol
    sub = slice(0, 60000)                                    # exact reference on a sub-sample is enough for the constant
    ref_v = O.nudft_type1(x, 0.346, torch.ones(N, dtype=torch.float64), (nm, nm))
    assert _rel(v, ref_v) < 2 * tol + 1e-13
    assert _rel(cplx, (1 - 2j) * Fy) < 2 * tol + 1e-12


@pytest.mark.parametrize("d,nm,tol", [(3, 21, 1e-9), (3, 23, 1e-6), (2, 141, 1e-7), (3, 12, 1e-5)])
@pytest.mark.parametrize("real_only", [False, True])
def test_type2_tiled_gather_beyond_lds(d, nm, tol, real_only, monkeypatch):
    """Fine grids that do not fit LDS with many points: tile-sorted points + LDS tiles (interp_tile_kernel) against
    the exact transform on a subset and against the untiled L2 gather on every point."""
    from efgp_hip import NufftPlan
    from oracle import efgp_oracle as O
    N = 60000
    x = _points(N, d, 40 + d, -2.0, 3.0)
    h = 0.17
    g = torch.Generator().manual_seed(9)
    f = torch.complex(torch.randn(2, nm ** d, generator=g, dtype=torch.float64),
                      torch.randn(2, nm ** d, generator=g, dtype=torch.float64))
    plan = NufftPlan(x.cuda(), h, tol)
    out = plan.type2(f.cuda(), (nm,) * d, real_only=real_only)
    assert out.shape == (2, N)
    sub = torch.arange(0, N, 97)
    ref = torch.stack([O.nudft_type2(x[sub], h, f[b], (nm,) * d) for b in range(2)])
    if real_only:
        ref = ref.real
    assert _rel(out[:, sub.cuda()], ref) < 2 * tol + 1e-13
    monkeypatch.setenv("EFGP_NO_TILES", "1")
    plan2 = NufftPlan(x.cuda(), h, tol)
    out2 = plan2.type2(f.cuda(), (nm,) * d, real_only=real_only)
    assert _rel(out, out2) < 1e-12


@pytest.mark.parametrize("d,nm,N", [(2, 23, 300_001), (2, 45, 524_288), (3, 9, 400_000)])
def test_bank_balanced_order_is_bitwise_neutral(d, nm, N, monkeypatch):
    """The per-plan processing order of the LDS spreader (class_order_kernel) only permutes exact integer sums:
    results are bit-identical with and without it, for strengths from memory, the fused pair and in-kernel probes."""
    from efgp_hip import NufftPlan
    x = _points(N, d, 77)
    g = torch.Generator().manual_seed(13)
    y = torch.randn(N, generator=g, dtype=torch.float64).cuda()
    res = []
    for off in (False, True):
        if off:
            monkeypatch.setenv("EFGP_NO_CLASS_ORDER", "1")
        else:
            monkeypatch.delenv("EFGP_NO_CLASS_ORDER", raising=False)
        plan = NufftPlan(x.cuda(), 0.31, 1e-7)
        Fy, v = plan.type1_pair(y, (nm,) * d, (2 * nm - 1,) * d)
        Zf = plan.type1_rademacher(11, 3, (nm,) * d)
        c = plan.type1(torch.complex(y, 0.5 * y.flip(0)), (nm,) * d)
        res.append((Fy, v, Zf, c))
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize("d,nm,N", [(2, 141, 200_000), (3, 21, 120_000)])
def test_tile_balancing_is_bitwise_neutral(d, nm, N, monkeypatch):
    """Tiled spreader / gather: the bank-balanced order inside the tiles (3-D: when the binning is built, 2-D: on the
    second pass over it) permutes exact integer sums and independent gathers -- results are bit-identical to the
    unbalanced binning and stable across repeated calls."""
    from efgp_hip import NufftPlan
    x = _points(N, d, 91)
    g = torch.Generator().manual_seed(17)
    y = torch.randn(N, generator=g, dtype=torch.float64).cuda()
    f = torch.complex(torch.randn(nm ** d, generator=g, dtype=torch.float64), torch.randn(nm ** d, generator=g, dtype=torch.float64)).cuda()
    res = []
    for off in (False, True):
        if off:
            monkeypatch.setenv("EFGP_NO_CLASS_ORDER", "1")
        else:
            monkeypatch.delenv("EFGP_NO_CLASS_ORDER", raising=False)
        plan = NufftPlan(x.cuda(), 0.29, 1e-7)
        a1 = plan.type1(y.to(torch.complex128), (nm,) * d)          # first pass over the binning
        a2 = plan.type1(y.to(torch.complex128), (nm,) * d)          # second pass: the 2-D binning is balanced now
        a3 = plan.type1(y.to(torch.complex128), (nm,) * d)
        g1 = plan.type2(f, (nm,) * d, real_only=True)
        g2 = plan.type2(f, (nm,) * d, real_only=True)
        assert torch.equal(a1, a2) and torch.equal(a2, a3) and torch.equal(g1, g2)
        res.append((a1, g1))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


def test_type1_ones_many_slabs_no_fixed_point_overflow():
    """Same-sign strengths over many workgroup slabs: the int64 sum over all slabs must not wrap (1-D, N = 1e6,
    tol 1e-10 so that the 48-bit raw accumulation is off: 512 slabs, each bounded alone would allow 2^61 per slab)."""
    from efgp_hip import NufftPlan
    N, nm = 1_000_000, 45
    x = _points(N, 1, 77)
    plan = NufftPlan(x.cuda(), 0.31, 1e-10)
    v = plan.type1_ones((nm,))
    assert abs(float(v[nm // 2].real) - N) < 1e-6 * N
    assert float(v.abs().max()) <= N * (1 + 1e-9)
    y = torch.full((N,), 3.0, dtype=torch.float64)
    Fy = plan.type1(y.cuda(), (nm,))
    assert _rel(Fy, 3.0 * v) < 1e-9


@pytest.mark.parametrize("bad", [float("nan"), float("inf")])
@pytest.mark.parametrize("N,d,layout", [(3000, 2, False), (50000, 2, True), (3000, 1, False), (40000, 3, False)])
def test_nonfinite_strengths_propagate(bad, N, d, layout):
    """A NaN / Inf among the strengths (missing data in y) must poison the transform as the reference's floating-point sums
    do, on every spreader (LDS fixed point, tiles, MFMA layout) -- not vanish in an fmax or become a finite garbage value."""
    from efgp_hip import NufftPlan, PointSet
    x = _points(N, d, 5)
    y = torch.randn(N, dtype=torch.float64)
    y[N // 3] = bad
    xd, yd = x.cuda(), y.cuda()
    nm = 23 if d < 3 else 9
    plan = NufftPlan(xd, 0.31, 1e-7, points=PointSet(xd, values=yd)) if layout else NufftPlan(xd, 0.31, 1e-7)
    Fy, v = plan.type1_pair(yd, (nm,) * d, (2 * nm - 1,) * d)
    assert not torch.isfinite(Fy).any()
    out = plan.type1(yd, (nm,) * d)
    assert not torch.isfinite(out).any()
    ok = plan.type1(torch.ones(N, dtype=torch.float64).cuda(), (nm,) * d)      # the plan is not left in a bad state
    assert torch.isfinite(ok).all() and abs(float(ok.reshape(-1)[ok.numel() // 2].real) - N) < 1e-6 * N
