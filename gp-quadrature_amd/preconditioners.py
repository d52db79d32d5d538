"""Weighted Toeplitz operators and circulant preconditioners for the EFGP normal equations.

The reference builds these inline in its experiment scripts, out of the same primitives the hot path
exports:
  * weighted convolution vector  v_w[k] = sum_n w_n exp(-2 pi i h k.x_n),  k in [-2m, 2m]^d  -- one type-1
    transform of the weights on the (4m+1)^d grid, then ``ToeplitzND(v_w, force_pow2=True)``
    (polyagamma_classification/pg_classifier.py:377-384);
  * circulant approximations of  D T D + sigma^2 I  used as CG preconditioners
    (prism_experiment/benchmark_prism_mean_preconditioners.py:131-191): wrap the Toeplitz vector onto the
    mtot^d torus, take its (real) eigenvalues by FFT, and invert  sigma^2 + alpha * eig  (scalar |ws|^2
    surrogate) or  ws^-1 (C + tau I)^-1 ws^-1  ("sandwich") in Fourier space.

Everything here stays on the device of its inputs: the transforms go through the HIP NUFFT / Toeplitz
operators, the circulant solves through torch.fft on the same stream.  The returned callables plug into
``ConjugateGradients(..., M_inv_apply=...)`` exactly as the reference's lambdas do.
"""
from __future__ import annotations

from typing import Callable, Sequence

import torch

from efgpnd import NUFFT, ToeplitzND

__all__ = ["weighted_toeplitz", "weighted_feature_operator", "wrap_to_circulant_kernel", "circulant_eigenvalues",
           "make_circulant_inverse", "scalar_circulant_preconditioner", "sandwich_circulant_preconditioner",
           "reference_preconditioners"]


def weighted_toeplitz(nufft_op: NUFFT, weights: torch.Tensor, out_shape: Sequence[int], cdtype=None) -> ToeplitzND:
    """ToeplitzND of  F^* diag(w) F  for per-point weights w (pg_classifier.py:377-384)."""
    conv_shape = tuple(2 * int(n) - 1 for n in out_shape)
    w = weights.flatten()
    v_w = nufft_op.type1(w if w.is_complex() else w.to(nufft_op.cdtype), out_shape=conv_shape)
    if cdtype is not None:
        v_w = v_w.to(cdtype)
    return ToeplitzND(v_w, force_pow2=True)


def weighted_feature_operator(nufft_op: NUFFT, weights: torch.Tensor, ws: torch.Tensor, out_shape: Sequence[int]):
    """u -> u + ws T_w (ws u), the feature-space operator of the Polya-Gamma classifier's solves (pg_classifier.py:398-416 with
    the exact weighted Toeplitz operator).  It is A_var with sigma^2 = 1 on the weighted Toeplitz operator, so
    ``ConjugateGradients(weighted_feature_operator(...), rhs, x0, ...)`` runs inside the fused HIP solver (efgp_cg_solve:
    one-launch persistent / cooperative kernels), not in a Python loop of transforms as the reference's does."""
    from efgpnd import create_A_var
    return create_A_var(ws, weighted_toeplitz(nufft_op, weights, out_shape, cdtype=torch.complex128), 1.0, torch.complex128)


def reference_preconditioners(v_kernel: torch.Tensor, ws: torch.Tensor, sigmasq: float):
    """The six preconditioners of the reference's study, by its names (benchmark_prism_mean_preconditioners.py:158-191)."""
    from efgpnd import create_jacobi_precond
    center = tuple((s - 1) // 2 for s in v_kernel.shape)
    return {"none": None,
            "jacobi_Nws2": create_jacobi_precond(ws, sigmasq, diag_scale=v_kernel[center].real),
            "circ_scalar_meanws2": scalar_circulant_preconditioner(v_kernel, ws, sigmasq, "mean"),
            "circ_scalar_maxws2": scalar_circulant_preconditioner(v_kernel, ws, sigmasq, "max"),
            "circ_sandwich_med": sandwich_circulant_preconditioner(v_kernel, ws, sigmasq, "median"),
            "circ_sandwich_geom": sandwich_circulant_preconditioner(v_kernel, ws, sigmasq, "geom")}


def wrap_to_circulant_kernel(v_kernel: torch.Tensor) -> torch.Tensor:
    """Fold the (2 ns - 1)^d Toeplitz vector onto the ns^d torus: c[r mod ns] += v[r]
    (benchmark_prism_mean_preconditioners.py:131-138), without the reference's Python loop over entries."""
    ns = [(L + 1) // 2 for L in v_kernel.shape]
    circ = v_kernel
    for d, n in enumerate(ns):
        # lags -(n-1)..(n-1) sit at indices 0..2n-2; lag r goes to slot r mod n
        neg = circ.narrow(d, 0, n - 1)                 # lags -(n-1)..-1  -> slots 1..n-1
        pos = circ.narrow(d, n - 1, n)                 # lags 0..n-1      -> slots 0..n-1
        pad_shape = list(pos.shape)
        pad_shape[d] = 1
        circ = pos + torch.cat([torch.zeros(pad_shape, dtype=circ.dtype, device=circ.device), neg], dim=d)
    return circ


def circulant_eigenvalues(circ_kernel: torch.Tensor, floor_rel: float = 1e-10) -> torch.Tensor:
    eigs = torch.fft.fftn(circ_kernel).real
    floor = floor_rel * max(float(eigs.abs().max()), 1.0)
    return eigs.clamp_min(floor)


def make_circulant_inverse(shape: Sequence[int], denom: torch.Tensor) -> Callable[[torch.Tensor], torch.Tensor]:
    """v -> ifft(fft(v) / denom) on the ns^d block, for flat vectors or (B, M) stacks (:141-155)."""
    shape = tuple(int(s) for s in shape)
    dims = tuple(range(-len(shape), 0))

    def M_inv(v: torch.Tensor) -> torch.Tensor:
        x = v.reshape(*v.shape[:-1], *shape) if v.ndim > 1 or len(shape) > 1 else v
        y = torch.fft.ifftn(torch.fft.fftn(x, dim=dims) / denom, dim=dims)
        return y.reshape(v.shape)

    return M_inv


def scalar_circulant_preconditioner(v_kernel: torch.Tensor, ws: torch.Tensor, sigmasq: float, mode: str = "mean"):
    """(sigma^2 + alpha C)^-1 with alpha = mean or max of |ws|^2 (:171-176)."""
    circ = wrap_to_circulant_kernel(v_kernel)
    eigs = circulant_eigenvalues(circ)
    abs2 = ws.abs().pow(2).real
    alpha = float(abs2.mean()) if mode == "mean" else float(abs2.max())
    denom = (sigmasq + alpha * eigs).to(v_kernel.dtype)
    return make_circulant_inverse(circ.shape, denom)


def sandwich_circulant_preconditioner(v_kernel: torch.Tensor, ws: torch.Tensor, sigmasq: float, mode: str = "median"):
    """ws^-1 (C + tau I)^-1 ws^-1 with tau the median / geometric mean of sigma^2 / |ws|^2 (:178-191)."""
    circ = wrap_to_circulant_kernel(v_kernel)
    eigs = circulant_eigenvalues(circ)
    w_abs = ws.abs().real
    w_floor = torch.quantile(w_abs, 0.05).clamp_min(1e-8)
    w_safe = w_abs.clamp_min(w_floor).to(v_kernel.dtype)
    ratio = sigmasq / (w_abs.clamp_min(w_floor) ** 2)
    tau = float(ratio.median()) if mode == "median" else float(torch.exp(torch.log(ratio).mean()))
    inner = make_circulant_inverse(circ.shape, (eigs + tau).to(v_kernel.dtype))
    return lambda v: inner(v / w_safe) / w_safe
