"""Equispaced Fourier quadrature grid for a stationary kernel (host-side, scalar work).

`get_xis` returns the 1-D frequency nodes ``xi_j = j h, |j| <= m`` such that the kernel is
approximated to tolerance ``eps`` on a domain of size ``L``.  Behaviour follows the reference's
`utils/kernels.py`: bisection for the eps-support (:28-69), the integral rule (:94-105) and the
closed-form heuristics (:106-134); return convention (:136-143).

For the built-in kernels the ~400 scalar function evaluations of the two bisections run on Python
floats (`Kernel._k_scalar/_S_scalar`) instead of 0-dim torch tensors; any object exposing
``kernel(r)`` and ``spectral_density(r)`` on tensors still works.
"""
import math
from typing import Optional, Tuple

import torch

from kernels.matern import Matern
from kernels.squared_exponential import SquaredExponential


class GetTruncationBound:
    """Finds L with f(L) ~ eps for a decreasing f by doubling + bisection."""

    def __init__(self, eps, kern, initial_upper_bound=1000.0, initial_lower_bound=0.0, max_iterations=200,
                 dtype=torch.float64, _scalar=False):
        self.eps = eps
        self.kern = kern
        self.initial_upper_bound = initial_upper_bound
        self.initial_lower_bound = initial_lower_bound
        self.max_iterations = max_iterations
        self.dtype = dtype
        self._scalar = _scalar

    def _f(self, r):
        if self._scalar:
            return self.kern(r)
        return self.kern(torch.tensor(r, device="cpu", dtype=self.dtype))

    def find_upper_bound_for_bisection(self):
        b = self.initial_upper_bound
        for _ in range(10):
            if self._f(b) > self.eps:
                b *= 2
            else:
                break
        return b

    def find_truncation_bound(self):
        a = self.initial_lower_bound
        b = self.find_upper_bound_for_bisection()
        mid = (a + b) / 2
        for _ in range(self.max_iterations):
            mid = (a + b) / 2
            if mid == a or mid == b:
                # the bracket is two adjacent doubles: every remaining pass of the fixed-count loop
                # recomputes the same midpoint and leaves (a, b) unchanged, so stopping is exact
                break
            if self._f(mid) > self.eps:
                a = mid
            else:
                b = mid
        return mid


def kernel_constants(kernel_obj, ell, var):
    """(kind, nu, c0) of a built-in kernel for the library's host-side grid functions, or None: kind 0 squared exponential
    with c0 = (2 pi l^2)^(d/2) * variance, kind 1 Matern (nu in {1/2, 3/2, 5/2}) with c0 = variance * scaling(l) -- the
    leading factors exactly as the Python expressions of kernels/*.py form them."""
    if type(kernel_obj) is SquaredExponential:
        return 0, 0.0, (2.0 * math.pi * ell ** 2) ** (kernel_obj.dimension / 2) * var
    if type(kernel_obj) is Matern and kernel_obj.nu in (0.5, 1.5, 2.5):
        return 1, kernel_obj.nu, var * kernel_obj._scaling(ell)
    return None


def _native_bounds(kernel_obj, ell, var, S0, eps, trunc_eps):
    """The two bisections in C (efgp_grid_bounds: same operations in the same order, bit-identical bounds) when the library is
    built and the kernel is one of the built-in ones; None otherwise (the Python loops below then run)."""
    import os
    if os.environ.get("EFGP_NO_NATIVE_GRID"):
        return None
    kc = kernel_constants(kernel_obj, ell, var)
    if kc is None or not (1 <= kernel_obj.dimension <= 3):
        return None
    try:
        import ctypes as C
        from efgp_hip.lib import lib
        L = lib()
    except Exception:
        return None
    a, b = C.c_double(0.0), C.c_double(0.0)
    rc = L.efgp_grid_bounds(kc[0], int(kernel_obj.dimension), float(kc[1]), float(ell), float(var), float(kc[2]), float(S0), float(eps),
                            float(trunc_eps), C.byref(a), C.byref(b))
    return (a.value, b.value) if rc == 0 else None


def get_xis(kernel_obj, eps: float, L, use_integral: bool = False, l2scaled: bool = False,
            dtype: torch.dtype = torch.float64, trunc_eps: Optional[float] = None) -> Tuple[torch.Tensor, float, int]:
    dim = kernel_obj.dimension
    if trunc_eps is None:
        trunc_eps = eps
    if torch.is_tensor(L):
        L = float(L)

    if use_integral:
        fast = hasattr(kernel_obj, "_k_scalar") and hasattr(kernel_obj, "_S_scalar") and dtype == torch.float64
        if fast:
            if hasattr(kernel_obj, "get_hypers") and tuple(kernel_obj.hypers) == ("lengthscale", "variance"):
                ell, var = kernel_obj.get_hypers()
            else:
                ell, var = kernel_obj.get_hyper("lengthscale"), kernel_obj.get_hyper("variance")
            k_of = lambda r: kernel_obj._k_scalar(r, ell, var)
            S0 = kernel_obj._S_scalar(0.0, ell, var)
            khat = lambda r: abs(r ** (dim - 1)) * kernel_obj._S_scalar(r, ell, var) / S0
        else:
            k_of = kernel_obj.kernel
            khat = lambda r: abs(r ** (dim - 1)) * kernel_obj.spectral_density(r) / kernel_obj.spectral_density(
                torch.tensor(0, device="cpu", dtype=dtype))
        native = _native_bounds(kernel_obj, ell, var, S0, eps, trunc_eps) if fast else None
        if native is not None:
            Ltime, Lfreq = native
        else:
            Ltime = GetTruncationBound(eps, k_of, dtype=dtype, _scalar=fast).find_truncation_bound()
            Lfreq = GetTruncationBound(trunc_eps, khat, dtype=dtype, _scalar=fast).find_truncation_bound()
        h_spacing = 1 / (L + Ltime)
        hm = math.ceil(Lfreq / h_spacing)
    elif isinstance(kernel_obj, Matern):
        ell, nu = kernel_obj.get_hyper("lengthscale"), kernel_obj.nu
        eps_use = eps / kernel_obj.get_hyper("variance")
        if l2scaled:
            s0 = float(kernel_obj.spectral_density(torch.tensor(0, device="cpu", dtype=dtype)))
            rl2sq = ((2 * nu / math.pi / ell ** 2) ** (dim / 2) * s0 ** 2 / 2
                     * math.gamma(dim / 2 + 2 * nu) / math.gamma(dim + 2 * nu) * 2 ** (-dim / 2))
            eps_use = eps * math.sqrt(rl2sq)
        h_spacing = 1 / (L + 0.85 * ell / math.sqrt(nu) * math.log(1 / eps_use))
        hm = math.ceil((math.pi ** (nu + dim / 2) * ell ** (2 * nu) * eps_use / 0.15) ** (-1 / (2 * nu + dim / 2)) / h_spacing)
    elif isinstance(kernel_obj, SquaredExponential):
        ell = kernel_obj.get_hyper("lengthscale")
        eps_use = eps / kernel_obj.get_hyper("variance")
        if l2scaled:
            k0 = float(kernel_obj.kernel(torch.tensor(0, device="cpu", dtype=dtype)))
            eps_use = eps * math.sqrt(k0 ** 2 * (math.sqrt(math.pi) * ell ** 2) ** dim)
        h_spacing = 1 / (L + ell * math.sqrt(2 * math.log(4 * dim * 3 ** dim / eps_use)))
        hm = math.ceil(math.sqrt(math.log(dim * (4 ** (dim + 1)) / eps_use) / 2) / math.pi / ell / h_spacing)
    else:
        raise ValueError("get_xis without use_integral needs a Matern or SquaredExponential kernel")

    xis = torch.arange(-hm, hm + 1, device="cpu", dtype=dtype) * h_spacing
    return xis, h_spacing, xis.numel()
