"""The C-ABI shared library: loads, exports every symbol of include/efgp_hip.h, and its host-only
window entry points behave (no GPU compute here)."""
import ctypes as C
import math

import numpy as np
import pytest


def _lib():
    import efgp_hip
    return efgp_hip.lib()


def test_library_exports_all_declared_symbols():
    import efgp_hip
    lib = _lib()
    names = efgp_hip.declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), n
    assert lib.efgp_version() >= 1


def test_window_width_monotone():
    lib = _lib()
    prev = 0
    for tol in [1e-1, 1e-2, 1e-4, 1e-6, 1e-8, 1e-10, 1e-12, 1e-14]:
        w = lib.efgp_window_width(tol, 2.0)
        assert 2 <= w <= 16 and w >= prev
        prev = w
    assert lib.efgp_window_width(1e-6, 3.0) <= lib.efgp_window_width(1e-6, 2.0) <= lib.efgp_window_width(1e-6, 1.25)
    assert lib.efgp_fine_grid_size(45, 1e-6) >= 90
    assert lib.efgp_fine_grid_size(23, 1e-6) % 2 == 0


def test_fine_grid_sizes_sit_on_a_short_ladder():
    """Small fine grids take 2^k / 3 * 2^k sizes only: every new FFT length is a runtime compilation in rocFFT (0.6-2.5 s,
    tools/train_loop_steps.py), and a hyper-parameter optimisation walks through many mode counts.  The ladder size is the
    first one >= 2 n whose window is no wider than the best dense (2^a 3^b 5^c in [2 n, 2.5 n]) size would need."""
    lib = _lib()

    def smooth(n):
        n = max(2, n + (n & 1))
        while True:
            m = n
            for p in (2, 3, 5):
                while m % p == 0:
                    m //= p
            if m == 1:
                return n
            n += 2

    for tol in (6e-8, 1e-5, 1e-9):
        seen = set()
        for n in range(1, 257):
            nf = lib.efgp_fine_grid_size(n, tol)
            assert nf >= max(32, 2 * n)
            m = nf
            while m % 2 == 0:
                m //= 2
            assert m in (1, 3), (n, nf)                       # 2^k or 3 * 2^k
            lo = max(32, smooth(2 * n))
            w_dense = min(lib.efgp_window_width_nd(tol, c / n, 2) for c in range(lo, max(lo, 5 * n // 2) + 1, 2) if smooth(c) == c)
            assert lib.efgp_window_width_nd(tol, nf / n, 2) <= w_dense
            assert nf <= 4 * max(16, n)                        # at most two ladder steps above the minimum 2 n
            seen.add(nf)
        assert len(seen) <= 10                                 # 256 mode counts, ten FFT lengths
        # the training run of the bench model (mtot 23 -> 17): two lengths for the Toeplitz boxes, two for the probes
        assert {lib.efgp_fine_grid_size(4 * m + 1, 6e-8) for m in (8, 9, 10, 11)} == {96, 128}
        assert {lib.efgp_fine_grid_size(2 * m + 1, 1e-5) for m in (8, 9, 10, 11)} == {48, 64}
    assert lib.efgp_fine_grid_size(301, 6e-8) % 2 == 0 and lib.efgp_fine_grid_size(301, 6e-8) < 768     # large grids: dense choice


@pytest.mark.parametrize("tol", [1e-3, 1e-6, 6e-8, 1e-10, 1e-13])
def test_window_polynomials_match_closed_form(tol):
    lib = _lib()
    vals = (C.c_double * 16)()
    first = C.c_int64()
    w = C.c_int()
    beta = C.c_double()
    rng = np.random.default_rng(1)
    worst = 0.0
    for X in rng.uniform(0, 90, 200):
        assert lib.efgp_window_eval(tol, 2.0, float(X), C.byref(first), vals, C.byref(w), C.byref(beta)) == 0
        W = w.value
        assert first.value == math.ceil(X - W / 2)
        for j in range(W):
            z = (first.value + j - X) * 2.0 / W
            ref = math.exp(beta.value * (math.sqrt(max(0.0, 1 - z * z)) - 1))
            worst = max(worst, abs(vals[j] - ref))
    assert worst < max(0.1 * tol, 1e-14)


def test_deconvolution_factors_against_quadrature():
    from scipy import integrate
    lib = _lib()
    nf, nm, tol = 96, 45, 6e-8
    out = (C.c_double * nm)()
    assert lib.efgp_window_deconv(tol, nf, nm, out) == 0
    w = lib.efgp_window_width(tol, nf / nm)
    beta = 0.976 * math.pi * w * (1 - 1 / (2 * nf / nm))
    for i in [0, 5, 22, 40, 44]:
        k = i - nm // 2
        val, _ = integrate.quad(lambda z: math.exp(beta * (math.sqrt(1 - z * z) - 1)) * math.cos(k * w * math.pi * z / nf),
                                -1, 1, epsabs=1e-14, epsrel=1e-13, limit=200)
        assert out[i] == pytest.approx(1.0 / (0.5 * w * val), rel=1e-9)
    assert lib.efgp_window_deconv(tol, 10, 20, out) == -1          # n_modes > nf -> EFGP_EINVAL
    assert b"bad sizes" in lib.efgp_last_error()
