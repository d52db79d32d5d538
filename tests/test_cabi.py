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
