"""EFGP (equispaced-Fourier Gaussian process) regression on MI355X -- host side.

Drop-in for the reference module of the same name (danbider/gp-quadrature `efgpnd.py`): the
public names, argument meanings and error behaviour follow it (`EFGPND`, `efgpnd_gradient_batched`,
`efgp_nd`, `NUFFT`, `ToeplitzND`, `compute_convolution_vector_vectorized_dD`, `create_A_mean`, ...),
but every hot operation -- NUFFT spreading/interpolation, the FFT Toeplitz mat-vec, the
preconditioned CG iterations and the N-length reductions -- runs in hand-written HIP kernels
behind the C ABI of `libefgp_hip.so` (include/efgp_hip.h).  torch tensors carry device memory,
streams and (for multi-GPU) the process group; there is no CPU fallback: without a GPU or without
the built library the solve raises RuntimeError.

Maths (reference conventions): features F[n,k] = exp(2 pi i h k.x_n) on the tensor grid
k in {-m..m}^d, weights ws = sqrt(S(h k) h^d); F* = type-1 NUFFT, F = type-2; F*F is the Toeplitz
operator T with vector v[k] = sum_n exp(-2 pi i h k.x_n), |k| <= 2m; the posterior-mean weights
solve (D T D + sigma^2 I) beta = D F* y  (D = diag ws) by Jacobi-preconditioned CG.
"""
from __future__ import annotations

import math
import os
import time
from math import prod
from typing import Dict, Optional, Tuple, Union

import numpy as np
import torch
from torch import nn
from torch.optim import Adam

from cg import ConjugateGradients
from kernels.kernel_params import GPParams
from utils.kernels import get_xis

from efgp_hip import NufftPlan, PointSet, ToeplitzOp, lanczos, lag_sums, variance_rhs, variance_contract, cg_solve, cg_solve_async, cg_solve_mean_async, vdot_real, compute_device, rademacher_fill, gradient_prepare, gradient_assemble
from efgp_hip.dist import PointShards

TWO_PI = 2.0 * math.pi
_CONV_TOL = 6e-8       # tolerance the reference hard-codes for the Toeplitz vector (efgpnd.py:1418)


class _StageRanges:
    """roctx ranges named after the reference's stage timers (efgpnd.py:61-280 `stage_times`, :888-966 predict), so that a
    `rocprofv3 --marker-trace --kernel-trace` timeline of a gradient step or a prediction reads by stage.  Off unless
    EFGP_ROCTX=1 (a push / pop pair costs about a microsecond of host time each; a 0.4-ms step has ten stages).
    torch.cuda.nvtx IS roctx on ROCm builds.  Stage names are known when a stage ENDS (`lap(name)`), so the range of the
    stage that follows is opened from the fixed order below."""
    ORDER = ["0_book_keeping", "1_frequency_grid_setup", "2_nufft_setup", "3_toeplitz_setup", "4_solve_cg", "5_compute_term2",
             "6_monte_carlo_trace", "7_batch_cg_solve", "7.5_compute_alpha", "8_gradient_calculation", "9_log_marginal_likelihood"]
    enabled = os.environ.get("EFGP_ROCTX", "0") == "1"

    def __init__(self, outer, first=None):
        self.depth = 0
        if self.enabled:
            self._push(outer)
            if first is not None:
                self._push(first)

    def _push(self, name):
        torch.cuda.nvtx.range_push(name)
        self.depth += 1

    def _pop(self):
        if self.depth > 0:
            torch.cuda.nvtx.range_pop()
            self.depth -= 1

    def lap(self, name):
        """Stage `name` has ended: close its range, open the one of the stage that follows it."""
        if not self.enabled:
            return
        if self.depth > 1:
            self._pop()
        k = self.ORDER.index(name) if name in self.ORDER else -1
        if 0 <= k < len(self.ORDER) - 1:
            self._push(self.ORDER[k + 1])

    def stage(self, name):
        """Open a named range inside the outer one (predict: 'predict_mean', 'compute_variance')."""
        if self.enabled:
            if self.depth > 1:
                self._pop()
            self._push(name)

    def close(self):
        while self.depth > 0:
            self._pop()


def _cmplx(real_dtype: torch.dtype) -> torch.dtype:
    """complex dtype matching a real dtype (reference: efgpnd.py:1233-1234)."""
    return torch.complex64 if real_dtype == torch.float32 else torch.complex128


def _as2d(x: torch.Tensor) -> torch.Tensor:
    return x.unsqueeze(-1) if x.ndim == 1 else x


def _dev_points(x: torch.Tensor, dev: torch.device) -> torch.Tensor:
    """(N,d) float64 contiguous copy (or view) of x on the compute device."""
    return _as2d(x).detach().to(device=dev, dtype=torch.float64).contiguous()


# ======================================================================================
# NUFFT  (reference: efgpnd.py:1423-1549)
# ======================================================================================
class NUFFT:
    """Type-1 (points -> modes, F*) and type-2 (modes -> points, F) transforms for fixed points.

    ``NUFFT(x, xcen, h, eps, cdtype=None, device=None)``: x (N,d), phases phi = 2 pi h (x - xcen),
    requested accuracy ``eps``.  Results come back on ``device or x.device`` in ``cdtype``.
    """

    def __init__(self, x, xcen, h, eps, cdtype=None, device=None):
        self.device = device or x.device
        self.dtype = x.dtype
        self.cdtype = cdtype or _cmplx(self.dtype)
        self.eps = eps
        self._hval = float(h)
        x2 = _as2d(x)
        self.d = x2.shape[1]
        if torch.is_tensor(xcen):
            self._xcen = [float(v) for v in xcen.detach().reshape(-1).tolist()]
        elif xcen is None:
            self._xcen = [0.0] * self.d
        else:
            self._xcen = [float(v) for v in np.atleast_1d(xcen)]
        if len(self._xcen) == 1 and self.d > 1:
            self._xcen = self._xcen * self.d
        self._dev = compute_device(x, device=self.device)
        self._x = _dev_points(x2, self._dev)
        self._plan = NufftPlan(self._x, self._hval, float(eps), self._xcen)
        self._x_ref = x2

    @property
    def phi(self) -> torch.Tensor:
        """(d,N) phases, as the reference exposes them (efgpnd.py:1451)."""
        xc = torch.tensor(self._xcen, dtype=self._x_ref.dtype, device=self._x_ref.device)
        return (TWO_PI * self._hval * (self._x_ref - xc)).T.contiguous().to(device=self.device, dtype=self.dtype)

    # device-level entry points (cuda tensors in, cuda complex128 out) used inside this module
    def _type1_dev(self, vals, out_shape):
        return self._plan.type1(vals, tuple(out_shape))

    def _type2_dev(self, fk, out_shape, modeord=0, real_only=False, batched=None):
        return self._plan.type2(fk, tuple(out_shape), modeord=modeord, real_only=real_only, batched=batched)

    def type1(self, vals, out_shape):
        """f[k] = sum_n vals_n exp(-i k.phi_n); vals (N,) or (B,N) -> (*out_shape) or (B,*out_shape)."""
        if isinstance(out_shape, int):
            out_shape = (out_shape,)
        res = self._type1_dev(vals.detach(), out_shape)
        return res.to(device=self.device, dtype=self.cdtype)

    def type2(self, fk, out_shape=None):
        """c_n = sum_k fk[k] exp(+i k.phi_n); fk flat (M,)/(B,M) with out_shape, or already shaped."""
        fk = fk.detach()
        if out_shape is None:
            if fk.ndim == self.d:
                shape, batched = tuple(fk.shape), False
            elif fk.ndim == self.d + 1:
                shape, batched = tuple(fk.shape[1:]), True
            else:
                raise ValueError("type2 needs out_shape for flattened input")
        else:
            if isinstance(out_shape, int):
                out_shape = (out_shape,)
            shape = tuple(out_shape)
            batched = fk.ndim > 1 and tuple(fk.shape) != shape
        res = self._type2_dev(fk, shape, batched=batched)
        return res.to(device=self.device, dtype=self.cdtype)


def setup_nufft(x, xcen, h, nufft_eps, cdtype):
    """Deprecated shim kept for import compatibility (reference: efgpnd.py:1551-1568)."""
    op = NUFFT(x, xcen, h, nufft_eps, cdtype=cdtype)
    return op.phi, (lambda phi_in, vals, OUT=None: op.type1(vals, out_shape=OUT)), \
        (lambda phi_in, fk_flat, OUT=None: op.type2(fk_flat, out_shape=OUT))


def compute_convolution_vector_vectorized_dD(m: int, x: torch.Tensor, h) -> torch.Tensor:
    """v[k] = sum_n exp(-2 pi i h k.x_n) for k in [-2m, 2m]^d (reference: efgpnd.py:1395-1421)."""
    x2 = _as2d(x)
    dev = compute_device(x2)
    plan = NufftPlan(_dev_points(x2, dev), float(h), _CONV_TOL)
    v = plan.type1_ones((4 * m + 1,) * x2.shape[1])
    return v.to(device=x.device, dtype=_cmplx(x.dtype))


# ======================================================================================
# Toeplitz operator (reference: efgpnd.py:1239-1393)
# ======================================================================================
class ToeplitzND:
    """y = T x with T[j,l] = v[j-l]; v has shape (L_1..L_d), blocks n_a = (L_a+1)//2.

    Accepts flat ``(..., prod n)`` or block ``(..., n_1..n_d)`` inputs like the reference; the
    circulant embedding (pad, rocFFT, multiply by the cached transform of v, inverse, crop) runs in
    libefgp_hip.  ``precompute_fft`` is accepted for compatibility; the transform of v is always cached.
    """

    def __init__(self, v: torch.Tensor, *, force_pow2: bool = True, precompute_fft: bool = True):
        if not torch.is_complex(v):
            v = v.to(torch.complex128 if v.dtype == torch.float64 else torch.complex64)
        self.Ls = list(v.shape)
        self.ns = [(L + 1) // 2 for L in self.Ls]
        self.size = prod(self.ns)
        self.d = len(self.Ls)
        self.device = v.device
        self.dtype = v.dtype
        self._dev = compute_device(v)
        self._op = ToeplitzOp(v.detach().to(self._dev), force_pow2=force_pow2)
        self.fft_shape = list(self._op.fft_shape)
        self.starts = [n - 1 for n in self.ns]
        self.ends = [s + n for s, n in zip(self.starts, self.ns)]

    def _apply_dev(self, u_flat):
        return self._op.apply(u_flat)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if x.shape[-1] == self.size:
            flat_in = True
            lead = x.shape[:-1]
        elif list(x.shape[-self.d:]) == self.ns:
            flat_in = False
            lead = x.shape[:-self.d]
        else:
            raise ValueError(f"Expected trailing dim {self.size} or block {tuple(self.ns)}, got {tuple(x.shape)}")
        out_dtype = x.dtype if x.is_complex() else self.dtype
        y = self._op.apply(x.detach().reshape(*lead, self.size))
        y = y if flat_in else y.reshape(*lead, *self.ns)
        return y.to(device=x.device if x.is_cuda else self.device, dtype=out_dtype)


# ======================================================================================
# operators on feature space (reference: efgpnd.py:1572-1631)
# ======================================================================================
class _FeatureOperator:
    """u -> ws*T(ws*u) [kind 'G'], + sigma^2 u ['mean'], or /sigma^2 + u ['var'].

    Callable like the reference's closures; additionally recognised by `ConjugateGradients`,
    which then runs the whole solve inside the fused HIP CG (`efgp_cg_solve`)."""

    def __init__(self, ws, toeplitz, sigmasq, cdtype, kind):
        self.ws = ws
        self.toeplitz = toeplitz
        self.sigmasq = None if sigmasq is None else float(sigmasq)
        self.cdtype = cdtype
        self.kind = kind
        self.variant = {"mean": 0, "var": 1}.get(kind, -1)
        self._efgp_fusable = kind in ("mean", "var")

    def __call__(self, u):
        u = u.to(dtype=self.cdtype)
        ws = self.ws.to(u.device)
        g = ws * self.toeplitz(ws * u)
        if self.kind == "mean":
            return g + self.sigmasq * u
        if self.kind == "var":
            return g / self.sigmasq + u
        return g


def create_Gv(ws, toeplitz, cdtype):
    return _FeatureOperator(ws, toeplitz, None, cdtype, "G")


def create_A_mean(ws, toeplitz, sigmasq_scalar, cdtype):
    return _FeatureOperator(ws, toeplitz, sigmasq_scalar, cdtype, "mean")


def create_A_var(ws, toeplitz, sigmasq_scalar, cdtype):
    return _FeatureOperator(ws, toeplitz, sigmasq_scalar, cdtype, "var")


def setup_operators(ws, toeplitz, sigmasq_scalar, cdtype):
    return (create_A_mean(ws, toeplitz, sigmasq_scalar, cdtype), create_A_var(ws, toeplitz, sigmasq_scalar, cdtype),
            create_Gv(ws, toeplitz, cdtype))


class _JacobiPreconditioner:
    """v -> v / (diag_scale |ws|^2 + sigma^2)  (reference: efgpnd.py:1619-1631)."""
    _efgp_jacobi = True

    def __init__(self, ws, sigmasq_scalar, diag_scale):
        scale = diag_scale if torch.is_tensor(diag_scale) else float(diag_scale)
        sig = sigmasq_scalar.detach() if torch.is_tensor(sigmasq_scalar) else float(sigmasq_scalar)
        self.diag = (scale * ws.abs().pow(2).real + sig).detach()

    def __call__(self, v):
        return v / self.diag.to(v.device)


def create_jacobi_precond(ws, sigmasq_scalar, diag_scale=1.0):
    return _JacobiPreconditioner(ws, sigmasq_scalar, diag_scale)


# ======================================================================================
# shared pieces of fit / gradient
# ======================================================================================
class _PinnedStage:
    """Two reusable pinned staging buffers per device (grown on demand), each guarded by the event of its last copy."""

    def __init__(self):
        self.bufs = [None, None]
        self.events = [None, None]
        self.turn = 0

    def upload(self, t: torch.Tensor, dev: torch.device) -> torch.Tensor:
        src = t.contiguous()
        raw = src.reshape(-1).view(torch.uint8)
        nbytes = raw.numel()
        k = self.turn
        self.turn ^= 1
        if self.events[k] is not None:
            self.events[k].synchronize()          # the copy issued two uploads ago: long finished in steady state
        if self.bufs[k] is None or self.bufs[k].numel() < nbytes:
            self.bufs[k] = torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8).pin_memory()
        self.bufs[k][:nbytes].copy_(raw)
        out = torch.empty(src.shape, dtype=src.dtype, device=dev)
        out.reshape(-1).view(torch.uint8).copy_(self.bufs[k][:nbytes], non_blocking=True)
        if self.events[k] is None:
            self.events[k] = torch.cuda.Event()
        self.events[k].record(torch.cuda.current_stream(dev))
        return out


_STAGES: Dict[int, Optional[_PinnedStage]] = {}


def _upload(t: torch.Tensor, dev: torch.device) -> torch.Tensor:
    """Host -> device copy that does not stall the host: a pageable-memory copy makes the host wait until the stream
    has drained (measured: the weights' upload at the top of every fit exposed ~50 us of launch gaps per step at
    N = 1e6).  Staged through a persistent pinned buffer the copy is just another stream-ordered operation
    (`Tensor.pin_memory()` per call is no alternative: 14 ms per call measured for a 292-KB grid)."""
    if t.device.type == "cpu" and dev.type == "cuda" and t.numel() > 0 and not os.environ.get("EFGP_NO_PINNED_UPLOAD"):
        stage = _STAGES.setdefault(dev.index or 0, _PinnedStage())
        if stage is not None:
            try:
                with torch.cuda.device(dev):
                    return stage.upload(t, dev)
            except RuntimeError:          # no pinned memory to be had (locked-memory limit): plain copies from now on
                _STAGES[dev.index or 0] = None
    return t.to(dev)


class _Grid:
    """Frequency grid and weights for the current hyper-parameters (host scalars + device vectors)."""

    def __init__(self, kernel, eps, L, d, dev, want_grad=False, defer_weights=False):
        """defer_weights=True: only the grid (h, mtot) is set up; the caller enqueues the N-scale pass over the points -- which
        needs nothing else -- and calls `make_weights` behind it (the gradient step waits for its result on the host every step,
        so what the host does before the first big launch is dead time on the device: ~35 us of a 0.4-ms step)."""
        xis_1d, h, mtot = get_xis(kernel_obj=kernel, eps=eps, L=L, use_integral=True, l2scaled=False)
        self.h = float(h)
        self.mtot = int(mtot)
        self.d = d
        self.shape = (self.mtot,) * d
        self.M = self.mtot ** d
        self.xis_1d = xis_1d
        self._xis = None                                                         # (M,d) host float64, built on first use
        # Non-blocking uploads where the whole solve is one launch (circulant grid small enough for the persistent kernel:
        # F^d <= 4096 with F = next_pow2(2 mtot - 1)) -- that is where the host running ahead pays (launch gaps of the
        # N = 1e6 step).  The multi-kernel solves synchronise with the device per burst anyway and keep the plain copy
        # (with torch's CPU pool capped, EFGP_ASYNC_UPLOAD_ALL=1 measures 7.11 vs 7.28 ms on the 3-D 64^3 fit and no
        # difference at 2-D 256^2; with an uncapped pool under a CPU quota the host running ahead made the throttling
        # stalls of efgp_hip/cpu_quota.py appear there too, so the conservative choice stays the default).
        F = 1 << (2 * self.mtot - 2).bit_length()
        async_ok = F ** d <= 4096 or bool(os.environ.get("EFGP_ASYNC_UPLOAD_ALL"))
        up = _upload if async_ok else (lambda t, dv: t.to(dv))
        self.dprime = None
        self.ws = None
        self._weights_args = (kernel, want_grad, dev, up)
        if not defer_weights:
            self.make_weights()

    def make_weights(self):
        if self.ws is not None:
            return
        kernel, want_grad, dev, up = self._weights_args
        d = self.d
        # Built-in kernels: ws (and the hyper-derivatives) in ONE C call on the host (efgp_spectral_weights_host) instead of
        # meshgrid + stack + spectral_density + sqrt + casts, ~10 torch CPU ops whose dispatch (60-150 us per fit) the 0.3-ms
        # step had started to wait for; 3-D grids too (torch's own CPU ops go multi-threaded and slow above 32768 elements)
        native = self._native_weights(kernel, want_grad, dev)
        if native is not None:
            self.ws, self.dprime = native
            return
        # Small grids: evaluate the spectral density on the host (one upload instead of several launches).  Large
        # grids (3-D): on the device -- torch's CPU reductions switch to their threaded path above 32768 elements,
        # which costs tens of milliseconds per call on a many-core host (measured: 89 ms for M = 12167, d = 3).
        # The same host pathology hits torch.pow with a real exponent even on a few hundred elements (12 ms per call
        # measured on the 256-thread host): Matern densities always go to the device.
        host_ok = self.M * d <= 16384 and type(kernel).__name__ != "Matern"
        where = self.xis if host_ok else up(self.xis, dev)
        S = kernel.spectral_density(where).to(torch.float64)
        # The fused solvers treat every system of a model as coefficients of REAL functions (LABNOTES.md section 4.4), which
        # needs ws real and even on the symmetric grid -- true for the spectral density of any real stationary kernel.  The
        # built-in kernels (native path above) are even by construction; a user-supplied spectral_density is checked here, once
        # per grid, so that a broken one is a ValueError at the fit and not NaN coefficients from the kernel's refusal.
        Sf = S.reshape(-1)
        if not bool(torch.isfinite(Sf).all()) or bool((Sf < 0).any()) or \
                float((Sf - Sf.flip(0)).abs().max()) > 1e-12 * float(Sf.abs().max()):
            raise ValueError(f"{type(kernel).__name__}.spectral_density must be finite, non-negative and even (S(-xi) = S(xi)) on "
                             "the frequency grid: it is the spectral density of a real stationary kernel")
        self.ws = up(torch.sqrt(S.to(torch.complex128) * self.h ** d), dev)       # (M,) complex, imag 0
        if want_grad:
            self.dprime = up((self.h ** d * kernel.spectral_grad(where)).to(torch.complex128), dev)   # (M,H)

    @property
    def xis(self):
        if self._xis is None:
            mesh = torch.meshgrid(*(self.xis_1d for _ in range(self.d)), indexing="ij")
            self._xis = torch.stack(mesh, dim=-1).view(-1, self.d)
        return self._xis

    def _native_weights(self, kernel, want_grad, dev):
        """(ws, dprime) as device tensors from ONE launch (efgp_spectral_weights) for the built-in kernels, else None."""
        if os.environ.get("EFGP_NO_NATIVE_GRID") or self.M > (1 << 24) or torch.device(dev).type != "cuda":
            return None
        bk = _builtin_kernel_constants(kernel)
        if bk is None:
            return None
        kc, ell, var = bk
        from efgp_hip.lib import lib
        dev = torch.device(dev)
        ws_d = torch.empty(self.M, dtype=torch.complex128, device=dev)
        dp_d = torch.empty((self.M, 2), dtype=torch.complex128, device=dev) if want_grad else None
        from efgp_hip.ops import _on, _stream
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        with _on(dev):
            rc = lib().efgp_spectral_weights(dev.index, kc[0], int(self.d),
                                             float(kc[1]), float(ell), float(var), float(kc[2]), float(self.h), int(self.mtot),
                                             ws_d.data_ptr(), dp_d.data_ptr() if want_grad else None, _stream(dev))
        return (ws_d, dp_d) if rc == 0 else None


def _builtin_kernel_constants(kernel):
    """((kind, nu, c0), lengthscale, variance) for the kernels efgp_spectral_weights evaluates itself, else None."""
    from utils.kernels import kernel_constants
    if tuple(getattr(kernel, "hypers", ())) != ("lengthscale", "variance") or not hasattr(kernel, "get_hypers"):
        return None
    try:
        ell, var = kernel.get_hypers()
        kc = kernel_constants(kernel, ell, var)
    except Exception:
        return None
    return None if kc is None else (kc, float(ell), float(var))


_NO_ONE_CALL_STEP = set()          # (device, d, mtot) whose solves are not single launches: efgp_gradient_step said so once


def _gradient_one_call(kernel, grid, xd, yd, points, sig, N, cg_tol, early_stopping, mean_cg_init, use_mean_pc, use_trace_pc, tight,
                       nufft_eps, T, trace_idx, variance_idx, probe_seed, y_norm_sq, dev):
    """The whole adjoint-estimator step in one library call (efgp_hip.gradient_step) when it applies: built-in kernel, one GPU,
    generated probes, a circulant grid whose solves are single launches.  Returns what `_gradient_tail_native` returns, or None
    (nothing enqueued that matters) and the caller drives the entry points itself."""
    key = (dev.index, grid.d, grid.mtot)
    F = 1 << (2 * grid.mtot - 2).bit_length()
    if key in _NO_ONE_CALL_STEP or F ** grid.d > 4096 or os.environ.get("EFGP_NO_GRADIENT_STEP") or os.environ.get("EFGP_NO_NATIVE_GRID"):
        return None
    kc = _builtin_kernel_constants(kernel)
    if kc is None or variance_idx != 1 or trace_idx != [0] or T < 1:
        return None
    from efgp_hip.ops import gradient_step
    warm = mean_cg_init is not None and tuple(mean_cg_init.shape) == (grid.M,)
    # the same draws from torch's generator, in the same order, as the entry-by-entry sequence
    if probe_seed is None:
        probe_seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
    v_seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
    yy = float(y_norm_sq) if y_norm_sq is not None else float(vdot_real(yd, yd))
    res = gradient_step(xd, yd, points, h=grid.h, mtot=grid.mtot, kconst=kc[0], lengthscale=kc[1], variance=kc[2], sigmasq=sig,
                        tol_pair=tight, tol_probe=max(tight, float(nufft_eps)) if nufft_eps else tight, cg_tol=cg_tol,
                        early_stop=early_stopping, nprobes=T, probe_seed=probe_seed, v_seed=v_seed, use_mean_pc=use_mean_pc,
                        use_trace_pc=use_trace_pc, variance_idx=variance_idx, trace_idx=trace_idx,
                        beta0=mean_cg_init.detach() if warm else None, n_obs=N, yy=yy)
    if res is None:
        _NO_ONE_CALL_STEP.add(key)
        return None
    out, beta, mean_iters, trace_iters = res
    return out, out[:3], out[3:6], out[6:9], out[9], beta, mean_iters, trace_iters, 2 * T, warm


def _domain_length(xd: torch.Tensor, shards: PointShards) -> float:
    lo, hi = torch.aminmax(xd, dim=0)
    lo, hi = shards.minmax(lo, hi)
    return float((hi - lo).max())


def _normal_equations(plan: NufftPlan, yd, grid: _Grid, shards: PointShards):
    """(F*y (M,), Toeplitz vector v ((4m+1,)*d)) in one pass over the points, all-reduced over shards."""
    m = (grid.mtot - 1) // 2
    Fy, v = plan.type1_pair(yd, grid.shape, (4 * m + 1,) * grid.d)
    shards.sum_many_([Fy, v])
    return Fy.reshape(-1), v


def _center_value(v: torch.Tensor) -> torch.Tensor:
    return v[tuple((s - 1) // 2 for s in v.shape)].real


def _center_flat(v: torch.Tensor) -> int:
    idx = 0
    for s in v.shape:
        idx = idx * s + (s - 1) // 2
    return idx


# ======================================================================================
# hyper-parameter gradient (reference: efgpnd.py:17-317)
# ======================================================================================
def efgpnd_gradient_batched(
        x, y, sigmasq, kernel, eps, trace_samples, x0=None, x1=None,
        *, nufft_eps=6e-8, cg_tol=None, early_stopping=True, device=None,
        do_profiling=False, compute_log_marginal=False,
        noise_floor: Optional[float] = None,
        stats_out: Optional[Dict[str, float]] = None,
        mean_cg_init: Optional[torch.Tensor] = None,
        use_mean_cg_preconditioner: bool = True,
        use_trace_cg_preconditioner: bool = True,
        log_marginal_probes=100, log_marginal_steps=25,
        probes_Z: Optional[torch.Tensor] = None, probes_V: Optional[torch.Tensor] = None,
        shards: Optional[PointShards] = None, trace_mode: str = "adjoint", probe_seed: Optional[int] = None,
        domain_length: Optional[float] = None, y_norm_sq: Optional[float] = None, points: Optional[PointSet] = None,
        log_marginal_probe_vectors: Optional[torch.Tensor] = None, pointwise_alpha: bool = False):
    """d(negative log marginal likelihood)/d(kernel hypers..., sigma^2) = (term1 - term2)/2 with
    Hutchinson trace estimates (data-space probes Z for non-variance kernel hypers, feature-space
    probes V for the noise) and CG solves.  ``x0, x1`` are ignored as in the reference (:72-73).

    Extra keyword arguments (not in the reference): ``probes_Z`` (T,N) / ``probes_V`` (T,M) inject the
    +-1 probes (the reference draws them from torch's generator at :179-182 and :199-202), and
    ``shards`` sums the gridded partials / N-length scalars over point shards; ``domain_length`` passes the box
    length max_a(max x_a - min x_a) when the caller already knows it (EFGPND caches it: the two N-length min/max
    reductions cost 6.6 ms per step at N = 5e6, d = 3); ``y_norm_sq`` likewise the global sum of y^2.  In the
    adjoint mode nothing is read back to the host before the gradient is complete (solves through the asynchronous
    entry points, scalars kept as 0-dim device tensors).
    Stage timers (seconds) are written to ``stats_out['stage_sec']`` with the reference's stage names (host
    clock; with ``do_profiling=True`` the device is synchronised at every stage boundary so they are exact).

    ``trace_mode``:
      * ``"adjoint"`` (default) evaluates every N-length inner product of the reference in feature space
        through the adjoint identity  sum_n z_n (F g)_n = <F* z, g>  and  |F g|^2 = <g, T g>:
        sum_n Z (rhs - F(ws beta))/sigma^2 = Re<F*Z, D'F*Z - ws beta>/sigma^2,  y.z = Re<F*y, g>,  |z|^2 = Re<g, T g>
        (g = ws beta, z = F g).  Same estimator, same probes, but NO type-2 pass over the N points (the
        reference spends 1 + 2T of them, :149, :189, :234) and, without injected probes, the +-1 draws are
        generated inside the spread kernel (``probe_seed``) so Z never exists in memory.  Agrees with the
        literal sequence to the NUFFT tolerance.
      * ``"reference"`` follows the reference's operation sequence literally (type-2 passes included).

    ``pointwise_alpha`` (adjoint mode only): the adjoint form obtains |y - F g|^2 as yy - 2 Re<F*y, g> + <g, T g>, i.e. by
    cancellation: the 6e-8 transform error is amplified by yy / |y - F g|^2 (the reference's pointwise alpha only by the
    square root of that).  For high-SNR data (sigma^2 << signal variance) set it to get alpha = (y - F g)/sigma^2 from ONE
    real type-2 pass and the two N-length reductions of the reference (:163, :170); everything else stays adjoint.
    """
    if trace_mode not in ("adjoint", "reference"):
        raise ValueError(f"trace_mode must be 'adjoint' or 'reference', got {trace_mode!r}")
    adjoint = trace_mode == "adjoint"
    num_hypers = kernel.num_hypers
    if cg_tol is None:
        cg_tol = eps
    out_device = device or x.device
    rdtype = x.dtype
    stages: Dict[str, float] = {}
    tic = [time.perf_counter()]
    ranges = _StageRanges("efgpnd_gradient_batched", "0_book_keeping")

    def lap(name):
        if do_profiling:                      # device-accurate stage times only when asked: a sync per stage is not free
            torch.cuda.synchronize(dev)
        now = time.perf_counter()
        stages[name] = stages.get(name, 0.0) + (now - tic[0])
        tic[0] = now
        ranges.lap(name)

    def finish():
        """Diagnostics, the optional log marginal likelihood and the returned gradient (reads the enclosing step's results)."""
        if stats_out is not None:
            stats_out.update({
                # iteration counts of the asynchronous solves stay on the device until somebody reads them (int(...), comparison,
                # formatting all do): two blocking device-to-host copies per step otherwise
                "mean_cg_iters": mean_iters,
                "trace_cg_iters": trace_iters,
                "trace_num_rhs": int(n_rhs),
                "feature_count": int(M),
                "mtot": int(grid.mtot),
                "trace_samples": int(trace_samples),
                "mean_cg_warm_start_used": bool(warm),
                "mean_cg_preconditioned": bool(use_mean_cg_preconditioner),
                "trace_cg_preconditioned": bool(use_trace_cg_preconditioner),
                "stage_sec": dict(stages),
            })
            stats_out["mean_beta"] = beta_raw.to(out_device)
            if fused:
                # grad | term1 | term2 | y.alpha live in one device vector: ONE read-back serves the diagnostics and the caller's
                # host copy of the gradient (EFGPND.compute_gradients)
                nh_ = int(term1.numel())
                host_out = out_vec.cpu()
                stats_out["term1"] = host_out[nh_:2 * nh_].clone()
                stats_out["term2"] = host_out[2 * nh_:3 * nh_].clone()
                stats_out["grad_host"] = host_out[:nh_].clone()
            else:
                stats_out["term1"] = term1.detach().cpu()
                stats_out["term2"] = term2.detach().cpu()

        log_marginal = None
        if compute_log_marginal:
            det_term = logdet_slq(ws, sig, top, probes=log_marginal_probes, steps=log_marginal_steps,
                                  dtype=torch.float64, device=dev, n=N, probe_vectors=log_marginal_probe_vectors)
            log_marginal = torch.tensor(-0.5 * float(y_alpha) - 0.5 * det_term - 0.5 * N * math.log(TWO_PI), dtype=rdtype)
            lap("9_log_marginal_likelihood")

        if do_profiling:
            print("\n===== stage timings for efgpnd_gradient_batched (seconds) =====")
            for k_, v_ in stages.items():
                print(f"  {k_:28s} {v_:.6f}")

        ranges.close()
        g_out = grad.to(device=out_device, dtype=rdtype)
        return (g_out, log_marginal) if compute_log_marginal else g_out

    # 0) book keeping -------------------------------------------------------------------------
    dev = compute_device(x, device=device)
    shards = shards or PointShards(enabled=False)
    xd = _dev_points(x, dev)
    yd = y.detach().to(device=dev, dtype=torch.float64).contiguous()
    N_local, d = xd.shape
    N = int(shards.sum_scalars([N_local], dev)[0]) if shards.active else N_local
    L = float(domain_length) if domain_length is not None else _domain_length(xd, shards)
    sig = float(sigmasq.detach()) if torch.is_tensor(sigmasq) else float(sigmasq)
    if noise_floor is not None:
        sig = max(sig, float(noise_floor))
    kernel_hypers = list(getattr(kernel, "hypers", []))
    variance_idx = kernel_hypers.index("variance") if "variance" in kernel_hypers else None
    kernel_hyper_count = num_hypers - 1
    trace_idx = [i for i in range(kernel_hyper_count) if i != variance_idx]
    lap("0_book_keeping")

    # 1) frequency grid -----------------------------------------------------------------------
    grid = _Grid(kernel, eps, L, d, dev, want_grad=True, defer_weights=True)
    M = grid.M
    lap("1_frequency_grid_setup")

    # 2) NUFFT plan ---------------------------------------------------------------------------
    # The Toeplitz vector is always computed to 6e-8 (:1418) and F*y rides in that pass; the probe transforms only
    # need the caller's nufft_eps (:186-189), i.e. a narrower window: a second plan over the same points (W^d LDS
    # operations per point: 3-D, eps 1e-4 vs 6e-8 is 216 vs 512 per channel).
    tight = min(float(nufft_eps), _CONV_TOL) if nufft_eps else _CONV_TOL
    if points is not None and (points.x.data_ptr() != xd.data_ptr() or points.npts != N_local):
        points = None
    # Everything from here to the assembled gradient in ONE library call when the step is the common one (adjoint estimator, built-in
    # kernel, one GPU, generated probes, single-launch solves): the ~15 entry points cost more host time driven from Python than
    # their kernels take on the device (csrc/gradient_step.cpp).  Stage timers then carry the whole call under "4_solve_cg".
    one_call = None
    if (adjoint and not pointwise_alpha and not shards.active and probes_Z is None and probes_V is None and not do_profiling
            and not compute_log_marginal and os.environ.get("EFGP_NO_FUSED_GRADIENT") is None and dev.type == "cuda"):
        one_call = _gradient_one_call(kernel, grid, xd, yd, points, sig, N, cg_tol, early_stopping, mean_cg_init,
                                      use_mean_cg_preconditioner, use_trace_cg_preconditioner, tight, nufft_eps, int(trace_samples),
                                      trace_idx, variance_idx, probe_seed, y_norm_sq, dev)
    if one_call is not None:
        fused = True
        for name in ("2_nufft_setup", "3_toeplitz_setup"):
            lap(name)
        (out_vec, grad, term1, term2, y_alpha, beta_raw, mean_iters, trace_iters, n_rhs, warm) = one_call
        for name in ("4_solve_cg", "5_compute_term2", "6_monte_carlo_trace", "7_batch_cg_solve", "7.5_compute_alpha", "8_gradient_calculation"):
            lap(name)
        return finish()
    plan = NufftPlan(xd, grid.h, tight, points=points)
    plan_p = plan if (not nufft_eps or float(nufft_eps) <= tight) else NufftPlan(xd, grid.h, float(nufft_eps), points=points)
    lap("2_nufft_setup")

    # 3) Toeplitz operator, Jacobi diagonal (F*y rides in the same pass over the points) --------
    Fy, v = _normal_equations(plan, yd, grid, shards)
    grid.make_weights()                      # behind the pass over the points: the device is busy while the host sets them up
    ws, Dp = grid.ws, grid.dprime
    top = ToeplitzOp(v)
    # The M-scale tail in native launches when the estimator is the adjoint one: prepare | solve | T g | probes, two scaled
    # Toeplitz products | batched solve | assemble -- about 35 launches per step instead of 118 (the step was bound by the host
    # enqueueing 3 us torch kernels on 529-element vectors).  The literal / pointwise modes keep the torch sequence below.
    fused = (adjoint and not pointwise_alpha and kernel_hyper_count <= 4 and Dp is not None and Dp.ndim == 2
             and Dp.shape[1] == kernel_hyper_count and os.environ.get("EFGP_NO_FUSED_GRADIENT") is None)
    if fused:
        lap("3_toeplitz_setup")
        (out_vec, grad, term1, term2, y_alpha, beta_raw, mean_iters, trace_iters, n_rhs, warm) = _gradient_tail_native(
            kernel, grid, top, Fy, v, sig, N, N_local, cg_tol, early_stopping, mean_cg_init, use_mean_cg_preconditioner,
            use_trace_cg_preconditioner, plan_p, shards, dev, int(trace_samples), trace_idx, variance_idx, probes_Z, probes_V,
            probe_seed, y_norm_sq, yd, lap)
    else:
        # the Jacobi diagonal v[0] |ws|^2 + sigma^2 (:128-133) from the same launch as the native tail: a diagonal that differs in
        # the last bit sends a CG run that ends at its iteration cap (ill-conditioned D T D) to a visibly different iterate
        diag = gradient_prepare(ws, None, v.reshape(-1)[_center_flat(v):_center_flat(v) + 1], sig, want_rhs=False)[0]
        lap("3_toeplitz_setup")

        # 4) mean solve ---------------------------------------------------------------------------
        rhs = ws * Fy
        warm = mean_cg_init is not None and tuple(mean_cg_init.shape) == tuple(rhs.shape)
        b0 = mean_cg_init.detach().to(device=dev, dtype=torch.complex128) if warm else None          # None: the solver starts from zeros it allocates itself (no copy)
        # rhs = D F*y of the real y (and a warm start from an earlier solve of the same kind): coefficients of real functions
        res_m = cg_solve_async(top, ws, sig, 0, rhs, b0, cg_tol, early_stop=early_stopping,
                               diag=diag if use_mean_cg_preconditioner else None, batched=False, hermitian=True)
        if res_m is None:
            res_m = cg_solve(top, ws, sig, 0, rhs, b0, cg_tol, early_stop=early_stopping,
                             diag=diag if use_mean_cg_preconditioner else None, batched=False, hermitian=True)[:2]
        beta, mean_iters = res_m
        beta_raw = beta.clone()
        beta_s = ws * beta                                        # g = D beta
        Tg = top.apply(beta_s)
        if not adjoint:
            z = plan_p.type2(beta_s, grid.shape)                 # F g, complex (N,)
            alpha = (yd - z) / sig
        lap("4_solve_cg")

        # 5) term 2 -------------------------------------------------------------------------------
        fadj_alpha = (Fy - Tg) / sig                               # = F* alpha without another pass over N
        term2_kernel = torch.stack([vdot_m(fadj_alpha, Dp[:, i] * fadj_alpha) for i in range(kernel_hyper_count)]) \
            if kernel_hyper_count else torch.zeros(0, dtype=torch.float64, device=dev)
        if adjoint:
            yy = float(y_norm_sq) if y_norm_sq is not None else shards.sum_scalars([vdot_real(yd, yd)], dev)[0]
            y_z = vdot_m(Fy, beta_s)                               # Re sum_n y_n z_n          (0-dim device tensors:
            z_z = vdot_m(beta_s, Tg)                               # |F g|^2 = <g, T g>         no host round trip)
            if pointwise_alpha:
                zr = plan_p.type2(beta_s, grid.shape, real_only=True)          # Re F g at the N points: one real gather
                alpha_r = (yd - zr) / sig
                a_norm, y_alpha = shards.sum_scalars([vdot_real(alpha_r, alpha_r), vdot_real(yd, alpha_r)], dev)
            else:
                a_norm = (yy - 2.0 * y_z + z_z) / (sig * sig)
                y_alpha = (yy - y_z) / sig
        else:
            a_norm, y_alpha = shards.sum_scalars([vdot_real(alpha, alpha), vdot_real(yd, alpha)], dev)
        if variance_idx is not None:
            variance_scalar = float(kernel.get_hyper("variance"))
            term2_kernel[variance_idx] = (y_alpha - sig * a_norm) / variance_scalar
        term2 = torch.cat((term2_kernel, torch.as_tensor(a_norm, dtype=torch.float64, device=dev).reshape(1)))
        lap("5_compute_term2")

        # 6) Monte-Carlo trace probes ---------------------------------------------------------------
        T = int(trace_samples)
        K = len(trace_idx)
        Z = None
        rhs_k = None
        if K > 0:
            if probes_Z is not None:
                Z = probes_Z.detach().to(device=dev, dtype=torch.float64).contiguous()
                FZ = plan_p.type1(Z, grid.shape).reshape(T, M)                            # real rows, two per pass
            elif adjoint:
                if probe_seed is None:
                    probe_seed = shards.shared_seed(dev)              # one draw on rank 0: all shards use one Z stream
                offset = shards.exclusive_offset(N_local, dev)
                FZ = plan_p.type1_rademacher(probe_seed, T, grid.shape, index_offset=offset).reshape(T, M)
            elif shards.active:
                # sharded literal mode: Z[t, n] from (seed, t, GLOBAL index n) so the shards hold slices of one global
                # probe matrix (per-rank torch generators would repeat or decorrelate blocks depending on their seeds)
                if probe_seed is None:
                    probe_seed = shards.shared_seed(dev)
                Z = rademacher_fill(dev, probe_seed, T, N_local, index_offset=shards.exclusive_offset(N_local, dev))
                FZ = plan_p.type1(Z, grid.shape).reshape(T, M)
            else:
                Z = torch.empty((T, N_local), device=dev, dtype=torch.float64).bernoulli_(0.5).mul_(2).sub_(1)
                FZ = plan_p.type1(Z, grid.shape).reshape(T, M)
            shards.sum_(FZ)
            DFZ = torch.stack([Dp[:, i] * FZ for i in trace_idx], dim=0).reshape(K * T, M)
            if not adjoint:
                rhs_k = plan_p.type2(DFZ, grid.shape, batched=True)                       # (K*T, N) complex
            B_k = ws * top.apply(DFZ)
        else:
            DFZ = torch.empty((0, M), dtype=torch.complex128, device=dev)
            B_k = torch.empty((0, M), dtype=torch.complex128, device=dev)
        if probes_V is not None:
            V = probes_V.detach().to(device=dev, dtype=torch.float64).contiguous()
        else:
            V = torch.empty((T, M), device=dev, dtype=torch.float64).bernoulli_(0.5).mul_(2).sub_(1)
            shards.broadcast_(V)                                      # replicated solves need ONE draw (rank 0's)
        Vc = V.to(torch.complex128)
        B_n = ws * top.apply(ws * Vc)
        B_all = torch.cat((B_k, B_n), dim=0)
        lap("6_monte_carlo_trace")

        # 7) batched CG -----------------------------------------------------------------------------
        Beta_all, trace_iters = _solve_batched(shards, top, ws, sig, 0, B_all, cg_tol, early_stop=early_stopping,
                                               diag=diag if use_trace_cg_preconditioner else None)
        lap("7_batch_cg_solve")

        # 7.5) term 1 -------------------------------------------------------------------------------
        term1 = torch.empty(num_hypers, dtype=torch.float64, device=dev)
        Beta_k, Beta_n = Beta_all[:K * T], Beta_all[K * T:]
        if K > 0:
            if adjoint:
                # sum_n Z (F(D'F*Z) - F(ws beta))/sigma^2 = Re <F*Z, D'F*Z - ws beta> / sigma^2   (F*Z is already global)
                diff = (DFZ - ws * Beta_k).reshape(K, T, M)
                sums = [(FZ.conj() * diff[slot]).sum().real / sig for slot in range(K)]
            else:
                fwdB = plan_p.type2(ws * Beta_k, grid.shape, batched=True)
                Alpha = (rhs_k - fwdB) / sig                                            # (K*T, N)
                sums = [vdot_real(Z, Alpha[s_ * T:(s_ + 1) * T]) for s_ in range(K)]    # sum_t sum_n Z*Alpha
                sums = shards.sum_scalars(sums, dev)
            for slot, ki in enumerate(trace_idx):
                term1[ki] = sums[slot] / T
        t1_noise = N / sig - ((Vc.conj() * Beta_n).sum(dim=1).real / sig).mean()
        if variance_idx is not None:
            term1[variance_idx] = (N - sig * t1_noise) / float(kernel.get_hyper("variance"))
        term1[-1] = t1_noise
        lap("7.5_compute_alpha")

        grad = 0.5 * (term1 - term2)
        lap("8_gradient_calculation")

        n_rhs = int(B_all.shape[0])

    return finish()


def _gradient_tail_native(kernel, grid, top, Fy, v, sig, N, N_local, cg_tol, early_stopping, mean_cg_init, use_mean_pc, use_trace_pc,
                          plan_p, shards, dev, T, trace_idx, variance_idx, probes_Z, probes_V, probe_seed, y_norm_sq, yd, lap):
    """Steps 4-8 of efgpnd_gradient_batched (adjoint estimator) in native launches; same quantities, same stage names.
    Returns (out_vec, grad, term1, term2, y_alpha, beta, mean_iters, trace_iters, n_rhs, warm): out_vec is the device vector
    grad | term1 | term2 | y.alpha of efgp_gradient_assemble, the next four are views of it."""
    ws, Dp, M = grid.ws, grid.dprime, grid.M
    K = len(trace_idx)
    H = Dp.shape[1]
    # 4) mean solve (reference :128-153): Jacobi diagonal and rhs = D F*y in one launch, solve, T g
    cidx = _center_flat(v)
    diag, rhs = gradient_prepare(ws, Fy, v.reshape(-1)[cidx:cidx + 1], sig, want_diag=use_mean_pc or use_trace_pc)
    warm = mean_cg_init is not None and tuple(mean_cg_init.shape) == tuple(rhs.shape)
    b0 = mean_cg_init.detach().to(device=dev, dtype=torch.complex128) if warm else None
    res_m = cg_solve_async(top, ws, sig, 0, rhs, b0, cg_tol, early_stop=early_stopping, diag=diag if use_mean_pc else None,
                           batched=False, hermitian=True)
    if res_m is None:
        res_m = cg_solve(top, ws, sig, 0, rhs, b0, cg_tol, early_stop=early_stopping, diag=diag if use_mean_pc else None,
                         batched=False, hermitian=True)[:2]
    beta, mean_iters = res_m
    Tg = top.apply_scaled(beta, pre=ws)                         # T (D beta)
    lap("4_solve_cg")
    lap("5_compute_term2")                                      # term 2 is part of the assemble launch below

    # 6) probes and the right-hand sides of the trace systems (reference :179-203)
    B_all = torch.empty(((K + 1) * T, M), dtype=torch.complex128, device=dev)
    FZ = None
    if K > 0:
        if probes_Z is not None:
            Z = probes_Z.detach().to(device=dev, dtype=torch.float64).contiguous()
            FZ = plan_p.type1(Z, grid.shape).reshape(T, M)
        else:
            if probe_seed is None:
                probe_seed = shards.shared_seed(dev)
            FZ = plan_p.type1_rademacher(probe_seed, T, grid.shape, index_offset=shards.exclusive_offset(N_local, dev)).reshape(T, M)
        shards.sum_(FZ)
        for slot, ki in enumerate(trace_idx):                   # D ws-scaled T (D'_i F*Z): D'_i rides in the pad, ws in the crop
            top.apply_scaled(FZ, pre=Dp[:, ki].contiguous(), post=ws, out=B_all[slot * T:(slot + 1) * T])
    if probes_V is not None:
        V = probes_V.detach().to(device=dev, dtype=torch.float64).contiguous()
    elif shards.active:
        V = torch.empty((T, M), device=dev, dtype=torch.float64).bernoulli_(0.5).mul_(2).sub_(1)
        shards.broadcast_(V)                                    # replicated solves need ONE draw (rank 0's)
    else:
        V = rademacher_fill(dev, int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item()), T, M)
    top.apply_scaled(V, pre=ws, post=ws, out=B_all[K * T:])
    lap("6_monte_carlo_trace")

    # 7) batched CG from zero (reference :205-236)
    Beta_all, trace_iters = _solve_batched(shards, top, ws, sig, 0, B_all, cg_tol, early_stop=early_stopping,
                                           diag=diag if use_trace_pc else None)
    lap("7_batch_cg_solve")

    # 7.5 / 8) every inner product of terms 1 and 2 and the final algebra: two launches, nothing read back
    yy = float(y_norm_sq) if y_norm_sq is not None else float(shards.sum_scalars([vdot_real(yd, yd)], dev)[0])
    variance = 1.0
    if variance_idx is not None:
        variance = dict(zip(kernel.hypers, kernel.get_hypers()))["variance"] if hasattr(kernel, "get_hypers") \
            else float(kernel.get_hyper("variance"))
    out = gradient_assemble(Fy, Tg, ws, beta, Dp, FZ, V, Beta_all.reshape(-1, M), variance_idx=variance_idx, trace_idx=trace_idx,
                            sigmasq=sig, n_obs=N, yy=yy, variance=variance)
    nh = H + 1
    lap("7.5_compute_alpha")
    lap("8_gradient_calculation")
    return out, out[:nh], out[nh:2 * nh], out[2 * nh:3 * nh], out[3 * nh], beta, mean_iters, trace_iters, (K + 1) * T, warm


def _rows_over_ranks(shards, top, R):
    """Should the R independent systems of a batched solve be split over the ranks?  On the 64 x 64 grid every system is ONE
    workgroup on one CU and up to num_CU of them run side by side on a GPU: splitting pays from more rows than CUs on.  Larger
    grids take many CUs per system (cooperative launch, multi-kernel 3-D iteration) and go through the rows in slabs:
    splitting pays as soon as every rank gets a row.  opts / env EFGP_SHARD_ROWS=0 keeps every solve replicated."""
    import os
    if shards is None or not shards.active or R < shards.world_size:
        return False
    cells = 1
    for f in top.fft_shape:
        cells *= int(f)
    if os.environ.get("EFGP_SHARD_ROWS", "1") == "0":
        split = False
    elif cells <= 4096:                                 # circulant grid of one workgroup (64 x 64, short 1-D lines)
        split = R > torch.cuda.get_device_properties(top.dev).multi_processor_count
    else:
        split = True
    # the decision selects the collectives that follow: it must be identical on all ranks (checked once per shape)
    return shards.agree(("rows_over_ranks", cells, R), split, top.dev)


def _solve_batched(shards, top, ws, sig, variant, B_all, tol, *, early_stop, diag, max_iter=None, hermitian=False):
    """The batched solves of the gradient and of the variance: asynchronous persistent / cooperative kernels where the grid
    allows, the multi-launch solver otherwise; rows split over the ranks when that pays (`_rows_over_ranks`).  Returns
    (X (R, M), iteration count: int-like)."""
    from efgp_hip.ops import LazyIterations
    from efgp_hip.dist import solve_rows_sharded
    cells = 1
    for f in top.fft_shape:
        cells *= int(f)

    def solve(block):
        res = cg_solve_async(top, ws, sig, variant, block, None, tol, max_iter=max_iter, early_stop=early_stop, diag=diag,
                             batched=True, hermitian=hermitian)
        if res is not None:
            if cells <= 4096:
                return res[0], res[1]._rows_dev          # one workgroup per system: no grid barrier, nothing to re-solve
            # cooperative launch (128^2..512^2 grids): a grid barrier that could not get its workgroups resident together
            # leaves -3 in the row counts and NaN in those systems.  Nobody downstream reads the counts before using the
            # solutions (diag_sums_nd, the gradient's assemble launch), so they are read HERE (one host wait behind a
            # solve of milliseconds) and dead systems go through the synchronous solver's multi-launch iteration.
            X, rows_dev = res[0], res[1]._rows_dev
            rows = [int(v) for v in rows_dev.tolist()]
            if any(v == -2 for v in rows):
                res[1].rows                                # raises the Hermitian refusal
            dead = [i for i, v in enumerate(rows) if v == -3]
            if dead:
                ix = torch.tensor(dead, device=X.device)
                xd, _, rd = cg_solve(top, ws, sig, variant, block.reshape(len(rows), -1)[ix], None, tol, max_iter=max_iter,
                                     early_stop=early_stop, diag=diag, batched=True, hermitian=hermitian)
                X.reshape(len(rows), -1)[ix] = xd.reshape(len(dead), -1)
                for i, v in zip(dead, rd):
                    rows[i] = int(v)
                rows_dev = torch.tensor(rows, dtype=torch.int32, device=X.device)
            return X, rows_dev
        x, _, rows = cg_solve(top, ws, sig, variant, block, None, tol, max_iter=max_iter, early_stop=early_stop, diag=diag,
                              batched=True, hermitian=hermitian)
        return x, rows

    R = B_all.shape[0]
    mi = int(max_iter) if max_iter is not None else 2 * top.size
    if _rows_over_ranks(shards, top, R):
        X, rows = solve_rows_sharded(shards, B_all.reshape(R, -1), solve)
        return X.reshape(B_all.shape), LazyIterations(rows, True, mi)
    X, rows = solve(B_all)
    if torch.is_tensor(rows):
        return X, LazyIterations(rows, True, mi)
    mx = max(rows)
    return X, (mx + 1 if mx < mi else mx)


def vdot_m(a, b):
    """Re<a,b> for M-length device vectors, kept on the device (glue on tiny vectors)."""
    return (a.conj() * b).sum().real


# ======================================================================================
# variance helpers (reference: efgpnd.py:1634-1679, 1761-1841) and SLQ log-det (:1686-1759)
# ======================================================================================
def _unwrap_operator(A_apply):
    if isinstance(A_apply, _FeatureOperator) and A_apply._efgp_fusable:
        return A_apply
    raise TypeError("expected an operator made by create_A_mean / create_A_var")


def diag_sums_nd(A_apply, J, xis_flat, max_cg_iter, cg_tol, ws, probes: Optional[torch.Tensor] = None, shards=None):
    """Hutchinson estimate of the lag sums c[r] = sum_{k-l=r} (A^-1)_{kl}-weighted products used by the
    stochastic variance (reference: efgpnd.py:1634-1664).  ``probes`` (J,M) of +-1 may be injected;
    otherwise they are drawn with torch.randint as in the reference (:1644).  ``shards`` (a PointShards of a multi-GPU
    model; the probes must then be identical on all ranks): the J systems are split by rows over the ranks."""
    Mtot, d_loc = xis_flat.shape
    m_loc = round(Mtot ** (1 / d_loc))
    assert m_loc ** d_loc == Mtot, "xis must lie on tensor grid"
    op = _unwrap_operator(A_apply)
    dev = op.toeplitz._dev
    if probes is None:
        etas = (torch.randint(0, 2, (J, Mtot), device=dev) * 2 - 1).to(torch.float64)
    else:
        etas = probes.detach().to(device=dev, dtype=torch.float64)
    wsd = ws.to(device=dev, dtype=torch.complex128)
    rhs = wsd[None, :] * etas
    us, _ = _solve_batched(shards, op.toeplitz._op, wsd, op.sigmasq, op.variant, rhs, cg_tol, early_stop=True, diag=None,
                           max_iter=max_cg_iter)
    # zero-padded correlation of every probe pair and the mean over probes (:1660-1664): hipFFT + three small kernels
    return lag_sums(wsd[None, :] * us, etas, m_loc, d_loc)


def nufft_var_est_nd(est_sums, h_val, x_center, pts, eps_val):
    """s^2(x*) = Re sum_r c[r] exp(2 pi i h r.x*), lags in FFT order (reference: efgpnd.py:1666-1679)."""
    B_loc, d_loc = pts.shape
    if est_sums.ndim != d_loc:
        raise ValueError("est_sums wrong dimensionality")
    op = NUFFT(pts, x_center, h_val, eps=eps_val)
    out = op._type2_dev(est_sums.detach(), tuple(est_sums.shape), modeord=1, real_only=True, batched=False)
    return out.to(device=pts.device, dtype=pts.dtype)


@torch.no_grad()
def logdet_slq(ws, sigma2, toeplitz, *, probes=1000, steps=100, dtype=torch.float64, device="cpu", eps=1e-18, n=None,
               probe_vectors: Optional[torch.Tensor] = None):
    """Stochastic Lanczos quadrature estimate of log det(sigma^2 I + D T D) restricted to feature space,
    i.e. log det(I + D T D / sigma^2) + n log sigma^2 (reference: efgpnd.py:1686-1759).

    All probes run their Lanczos recurrences inside ONE launch (`efgp_lanczos`: one workgroup per probe, alpha / beta
    stay on the device); grids beyond the single-launch kernel run the same recurrence for all probes at once as batched
    device operations.  Either way nothing is read back before the tridiagonal eigen-problems (one batched `eigh` of
    (probes, steps, steps) at the end): the reference's loop structure costs two host reads per Lanczos step.
    ``probe_vectors`` (probes, M) of +-1 may be injected; otherwise they are drawn as the reference does (:1716-1717)."""
    if n is None:
        raise ValueError("logdet_slq needs n (number of observations)")
    top = toeplitz._op if isinstance(toeplitz, ToeplitzND) else toeplitz
    dev = top.dev
    w = ws.real.to(device=dev, dtype=torch.float64)
    wc = w.to(torch.complex128)
    m = w.numel()
    s2 = float(sigma2)
    if probe_vectors is None:
        zs = torch.empty((int(probes), m), dtype=torch.float64, device=dev).bernoulli_(0.5).mul_(2).sub_(1)
    else:
        zs = probe_vectors.detach().to(device=dev, dtype=torch.float64).reshape(-1, m)
    P, K = zs.shape[0], int(steps)
    res = lanczos(top, wc, s2, 1, zs, K)
    if res is not None:
        alphas, betas, norm2, taken = res
        k_idx = torch.arange(K, device=dev)[None, :]
        live = k_idx < taken[:, None]                               # steps actually taken per probe
    else:
        q = (zs / zs.norm(dim=1, keepdim=True)).to(torch.complex128)
        norm2 = (zs * zs).sum(dim=1)
        q_prev = torch.zeros_like(q)
        beta_prev = torch.zeros(P, dtype=torch.float64, device=dev)
        alive = torch.ones(P, dtype=torch.bool, device=dev)
        a_list, b_list, l_list = [], [], []
        for _ in range(K):
            vv = q + (wc * top.apply(wc * q)) / s2 - beta_prev[:, None] * q_prev
            a = (q.conj() * vv).sum(dim=1).real
            vv = vv - a[:, None] * q
            b = vv.norm(dim=1)
            a_list.append(a)
            b_list.append(b)
            l_list.append(alive)
            nxt = alive & (b >= 1e-12)                               # the reference breaks after recording this step (:1733)
            safe = torch.where(nxt, b, torch.ones_like(b))
            q_prev = torch.where(nxt[:, None], q, q_prev)
            q = torch.where(nxt[:, None], vv / safe[:, None], q)
            beta_prev = torch.where(nxt, b, beta_prev)
            alive = nxt
        alphas, betas, live = torch.stack(a_list, 1), torch.stack(b_list, 1), torch.stack(l_list, 1)
    # tridiagonal matrices, padded with decoupled unit diagonal entries behind the steps taken (log 1 = 0, zero weight)
    alphas = torch.where(live, alphas, torch.ones_like(alphas))
    off = torch.where(live[:, 1:], betas[:, :-1], torch.zeros_like(betas[:, :-1]))      # beta_i couples steps i and i+1
    Tm = torch.diag_embed(alphas) + torch.diag_embed(off, offset=1) + torch.diag_embed(off, offset=-1)
    evals, evecs = torch.linalg.eigh(Tm.cpu())                       # (P, K, K) tiny: one transfer, LAPACK on the host
    evals = evals.clamp_min(eps)
    quad = ((evecs[:, 0, :] ** 2) * torch.log(evals)).sum(dim=1) * norm2.cpu()
    return float(quad.sum() / P) + n * math.log(s2)


def compute_prediction_variance(x_new, xis, ws, A_var, cg_tol, max_cg_iter, variance_method, h, xcen,
                                hutchinson_probes, nufft_eps, device, rdtype, cdtype, probes=None, shards=None):
    """Latent posterior variance at x_new: 'regular' (one CG solve per point, microbatched) or
    'stochastic' (Hutchinson lag sums + FFT-ordered type-2).  Reference: efgpnd.py:1761-1841."""
    method = variance_method.lower()
    if method == "regular":
        op = _unwrap_operator(A_var)
        dev = op.toeplitz._dev
        wsd = ws.to(device=dev, dtype=torch.complex128)
        xn = x_new.to(device=dev, dtype=torch.float64)
        M_loc = wsd.numel()
        mtot_loc = round(M_loc ** (1.0 / xn.shape[1]))
        assert mtot_loc ** xn.shape[1] == M_loc, "ws must lie on the tensor grid"
        hval = float(h)
        out = []
        for xb in torch.split(xn, 8192, dim=0):
            rhs = variance_rhs(xb, hval, mtot_loc, wsd)                  # ws * conj(f(x*)): explicit feature rows (b, M)
            gamma, _, _ = cg_solve(op.toeplitz._op, wsd, op.sigmasq, op.variant, rhs, None, cg_tol,
                                   max_iter=max_cg_iter, early_stop=True, diag=None, batched=True,
                                   hermitian=True)     # feature rows of real points: conjugate-even
            out.append(variance_contract(xb, hval, mtot_loc, wsd, gamma))
        return torch.cat(out, dim=0).to(device=device, dtype=rdtype)
    if method == "stochastic":
        t1 = time.time()
        est = diag_sums_nd(A_var, hutchinson_probes, xis, max_cg_iter, cg_tol, ws, probes=probes, shards=shards)
        print(f"Time to compute diag sums: {time.time() - t1:.4f} seconds")
        return nufft_var_est_nd(est, h, xcen, x_new, nufft_eps).to(device=device, dtype=rdtype)
    raise ValueError(f"Variance method '{variance_method}' not implemented. Choose 'regular' or 'stochastic'.")


# ======================================================================================
# the model (reference: efgpnd.py:336-1226)
# ======================================================================================
class EFGPND(nn.Module):
    """Equispaced-Fourier GP regression in d dimensions.

    ``EFGPND(x, y, kernel, sigmasq=None, eps=1e-2, nufft_eps=1e-4, opts=None, estimate_params=True)``
    with ``kernel`` a kernel object or one of "SquaredExponential"/"SE"/"Matern12"/"Matern32"/"Matern52"
    (case-insensitive).  Recognised ``opts``: cg_tolerance (1e-4), max_cg_iterations (1000),
    mean_cg_preconditioner (True), trace_cg_preconditioner (True), mean_cg_warm_start (True),
    noise_floor, log_marginal_probes (100), log_marginal_steps (25); additionally
    ``shard_points`` (bool): x, y hold THIS rank's block of the observations and gridded partial
    sums are all-reduced over the default process group (one process per GPU); ``point_layout``
    ("auto" | True | False): when the model builds its sorted point layout (see ``_layout``).
    """

    def __init__(self, x, y, kernel, sigmasq: float = None, eps: float = 1e-2, nufft_eps: float = 1e-4,
                 opts: Optional[Dict] = None, estimate_params: bool = True):
        super().__init__()
        self.x = x
        self.y = y
        self.device = x.device
        self.eps = eps
        self.nufft_eps = nufft_eps
        self.opts = {} if opts is None else opts.copy()
        dimension = 1 if x.ndim == 1 else x.shape[1]

        if isinstance(kernel, str):
            from kernels.squared_exponential import SquaredExponential
            from kernels.matern import Matern
            name = kernel.lower()
            if name in ("squaredexponential", "se"):
                kernel = SquaredExponential(dimension=dimension)
            elif name in ("matern12", "matern32", "matern52"):
                kernel = Matern(dimension=dimension, nu={"matern12": 0.5, "matern32": 1.5, "matern52": 2.5}[name])
            else:
                raise ValueError(f"Unknown kernel type: {kernel}")
        self.kernel = kernel

        if estimate_params:
            try:
                ls, var, noise = kernel.estimate_hyperparameters(x, y)
                if hasattr(kernel, "set_hyper"):
                    kernel.set_hyper("lengthscale", ls)
                    kernel.set_hyper("variance", var)
                else:
                    print(f"Warning: Could not set hyperparameters on kernel of type {type(kernel)}")
                if sigmasq is None:
                    sigmasq = noise
            except Exception as e:   # same forgiving behaviour as the reference (:437-441)
                print(f"Warning: Failed to estimate hyperparameters: {e}")
                if sigmasq is None:
                    sigmasq = 0.1

        # hyper-parameters live in log space in the data's dtype (reference :444-457)
        prev = torch.get_default_dtype()
        try:
            torch.set_default_dtype(x.dtype)
            self._gp_params = GPParams(kernel=kernel, init_sig2=(sigmasq or 0.1))
            self.register_parameter("gp_params", self._gp_params.raw)
            if hasattr(self.kernel, "_gp_params_ref") and self.kernel._gp_params_ref is None:
                self.kernel._gp_params_ref = self._gp_params
        finally:
            torch.set_default_dtype(prev)

        self._beta = None
        self._xis = None
        self._ws = None
        self._toeplitz = None
        self._fitted = False
        self._cached_params = {}
        self._registered_optimizers = []
        self._last_gradient_stats = {}
        self._last_gradient_beta = None
        self._last_fit_stats = {}
        self._devdata = None
        self._fit_state = None
        self._predict_plan = None
        self._nan_scalar = None
        # shard_points: True = torch.distributed default group; an efgp_hip.RcclComm = the library's own RCCL communicator
        sp = self.opts.get("shard_points", False)
        self._shards = PointShards(comm=sp) if (sp is not None and not isinstance(sp, bool)) else PointShards(enabled=bool(sp))
        self._update_param_cache()

    # -- parameter bookkeeping ------------------------------------------------------------------
    def register_optimizer(self, optimizer):
        """Wrap optimizer.step so the hyper-parameter cache is refreshed after every step."""
        if optimizer in self._registered_optimizers:
            return optimizer
        inner = optimizer.step

        def step_and_sync(*a, **k):
            res = inner(*a, **k)
            self._update_param_cache()
            return res

        optimizer.step = step_and_sync
        self._registered_optimizers.append(optimizer)
        return optimizer

    @property
    def sigmasq(self) -> torch.Tensor:
        return self._gp_params.sig2

    @property
    def last_gradient_stats(self) -> Dict:
        """Diagnostics of the last gradient step (reading them waits for its asynchronous solves)."""
        st = self._last_gradient_stats
        for key in ("mean_cg_iters", "trace_cg_iters"):
            if key in st and not isinstance(st[key], int):
                st[key] = int(st[key])
        return st

    @property
    def last_fit_stats(self) -> Dict:
        """Diagnostics of the last fit (reading them waits for an asynchronous solve to finish)."""
        st = dict(self._last_fit_stats)
        if "mean_cg_iters" in st:
            st["mean_cg_iters"] = int(st["mean_cg_iters"])
        return st

    def _current_hypers(self):
        vals = {name: float(self.kernel.get_hyper(name)) for name in getattr(self.kernel, "hypers", [])}
        vals["sigmasq"] = float(self.sigmasq.detach())
        return vals

    def _update_param_cache(self):
        if isinstance(self.kernel, str):
            self._cached_params["kernel_type"] = str(type(self.kernel))
        else:
            for name, value in self.kernel.iter_hypers():
                self._cached_params[name] = float(value)
        self._cached_params["sigmasq"] = self._gp_params.host_pos()[-1]
        return self

    def _params_changed(self):
        if not self._cached_params:
            return True
        try:
            for name, value in self.kernel.iter_hypers():
                if name not in self._cached_params or abs(self._cached_params[name] - float(value)) > 1e-8:
                    return True
        except (AttributeError, TypeError):
            if self._cached_params.get("kernel_type") != str(type(self.kernel)):
                return True
        if "sigmasq" not in self._cached_params or \
                abs(self._cached_params["sigmasq"] - float(self.sigmasq.detach())) > 1e-8:
            return True
        # the fit itself remembers the hyper-parameters it was computed with
        if self._fit_state is not None:
            now = self._current_hypers()
            then = self._fit_state["hypers"]
            return any(abs(now[k] - then.get(k, float("nan"))) > 1e-8 for k in now)
        return False

    # -- device data ------------------------------------------------------------------------------
    def _device_data(self):
        if self._devdata is None:
            dev = compute_device(self.x)
            xd = _dev_points(self.x, dev)
            yd = self.y.detach().to(device=dev, dtype=torch.float64).contiguous()
            L = _domain_length(xd, self._shards)
            if L <= 1e-9:
                L = 1.0
            n_glob = int(self._shards.sum_scalars([xd.shape[0]], dev)[0]) if self._shards.active else xd.shape[0]
            yy = self._shards.sum_scalars([vdot_real(yd, yd)], dev)[0]          # global sum of y^2 (gradient, once)
            self._devdata = dict(dev=dev, x=xd, y=yd, L=L, N=n_glob, yy=yy, points=None, passes=0)
        return self._devdata

    def _layout(self):
        """The model's point layout (grid-independent sorted copies for the type-1 pass, bounding box, max|y| once per model), or
        None while it is not worth building.  opts["point_layout"]: "auto" (default) -- from the SECOND pass over the points
        on: a model that is fitted once (efgp_nd, predict-only use; efgpnd_variance_shootout.py:129-137) never pays the sort
        (bounding box + key + radix sort + two gathers: 0.5 ms at N = 1e6, 8-13 ms at 1e7), a training loop pays it at its second
        step and streams the sorted copies from then on; True -- from the first pass; False -- never."""
        dd = self._device_data()
        want = self.opts.get("point_layout", "auto")
        dd["passes"] += 1
        if want is False or dd["x"].shape[0] == 0 or (want == "auto" and dd["passes"] < 2):
            return None
        if dd["points"] is None:
            dd["points"] = PointSet(dd["x"], values=dd["y"])
        return dd["points"]

    # -- gradient -----------------------------------------------------------------------------------
    def compute_gradients(self, *, trace_samples: int = 10, do_profiling: bool = False,
                          nufft_eps: Optional[float] = None, cg_tol: Optional[float] = None,
                          noise_floor: Optional[float] = None, apply_gradients: bool = True,
                          compute_log_marginal: bool = False, log_marginal_probes: int = 100,
                          log_marginal_steps: int = 25, verbose: bool = False, **kwargs):
        """Gradient of the negative log marginal likelihood w.r.t. the log-space parameters
        (kernel hypers..., sigma^2); written to ``self._gp_params.raw.grad`` when apply_gradients."""
        self._update_param_cache()
        if nufft_eps is None:
            nufft_eps = self.eps * 0.1
        if noise_floor is None:
            noise_floor = self.opts.get("noise_floor")
        if cg_tol is None:
            cg_tol = 0.1 * self.eps
        warm = self.opts.get("mean_cg_warm_start", True)
        stats: Dict = {}
        dd = self._device_data()
        res = efgpnd_gradient_batched(
            dd["x"], dd["y"], sigmasq=self._gp_params.host_pos()[-1], kernel=self.kernel, eps=self.eps,
            trace_samples=trace_samples, do_profiling=do_profiling, nufft_eps=nufft_eps, cg_tol=cg_tol,
            noise_floor=noise_floor, stats_out=stats,
            mean_cg_init=self._last_gradient_beta if warm else None,
            use_mean_cg_preconditioner=self.opts.get("mean_cg_preconditioner", True),
            use_trace_cg_preconditioner=self.opts.get("trace_cg_preconditioner", True),
            compute_log_marginal=compute_log_marginal, log_marginal_probes=log_marginal_probes,
            log_marginal_steps=log_marginal_steps, shards=self._shards, domain_length=dd["L"], y_norm_sq=dd["yy"],
            points=self._layout(), **kwargs)
        self._last_gradient_beta = stats.pop("mean_beta", None)
        grad_host = stats.pop("grad_host", None)       # the native tail reads grad | term1 | term2 back in one copy
        self._last_gradient_stats = stats
        grads, log_marginal = res if compute_log_marginal else (res, None)
        if grads.ndim == 0:
            grads = grads.unsqueeze(0)
        raw = self._gp_params.raw
        pos = raw.detach().exp()                         # = pos, without recording the exp for autograd
        if grad_host is not None and raw.device.type == "cpu":
            grads = grad_host
        raw_grad = grads.detach().to(device=raw.device, dtype=raw.dtype) * pos               # chain rule d/dlog
        if apply_gradients:
            with torch.no_grad():
                raw.grad = raw_grad.detach().clone()
        return (raw_grad, log_marginal) if compute_log_marginal else raw_grad

    # -- fit -----------------------------------------------------------------------------------------
    def _compute_common_parameters(self, force_recompute: bool = False, nufft_eps: Optional[float] = None) -> None:
        """Fit: grid, one fused pass over the points for (F*y, Toeplitz vector), Jacobi-PCG for beta."""
        self._update_param_cache()
        if self._fitted and not force_recompute and not self._params_changed():
            return
        if nufft_eps is None:
            nufft_eps = self.nufft_eps
        dd = self._device_data()
        dev, xd, yd = dd["dev"], dd["x"], dd["y"]
        d = xd.shape[1]
        sig = self._gp_params.host_pos()[-1]
        rdtype = self.x.dtype
        cdtype = _cmplx(rdtype)

        grid = _Grid(self.kernel, self.eps, dd["L"], d, dev)
        plan = NufftPlan(xd, grid.h, min(float(nufft_eps), _CONV_TOL), points=self._layout())
        Fy, v = _normal_equations(plan, yd, grid, self._shards)
        toeplitz = ToeplitzND(v, force_pow2=True)
        use_precond = self.opts.get("mean_cg_preconditioner", True)
        tol = self.opts.get("cg_tolerance", 1e-4)
        warm = self.opts.get("mean_cg_warm_start", True) and self._beta is not None and \
            tuple(self._beta.shape) == tuple(Fy.shape)
        # Cold start on a grid that fits the single-launch kernel: rhs = ws*F*y (:792), the Jacobi diagonal
        # v[0]|ws|^2 + sigma^2 (:795-799) and beta_0 = 0 are formed inside the solve kernel -- one launch, no host
        # synchronisation (the iteration count stays on the device until somebody reads last_fit_stats).
        res = None
        if not warm:
            res = cg_solve_mean_async(toeplitz._op, grid.ws, sig, _center_value(v) if use_precond else None, Fy, tol,
                                      early_stop=True)
        if res is None:
            # rhs = D F*y (:792) and the Jacobi diagonal (:795-799) in one native launch (four torch launches otherwise, whose first
            # use in a process costs 0.3 s of torch's own lazy kernel loading)
            cidx = _center_flat(v)
            diag, rhs = gradient_prepare(grid.ws, Fy, v.reshape(-1)[cidx:cidx + 1], sig, want_diag=use_precond)
            b0 = self._beta.detach().to(device=dev, dtype=torch.complex128) if warm else None
            res = cg_solve_async(toeplitz._op, grid.ws, sig, 0, rhs, b0, tol, early_stop=True, diag=diag, batched=False,
                                 hermitian=True)
            if res is None:
                res = cg_solve(toeplitz._op, grid.ws, sig, 0, rhs, b0, tol, early_stop=True, diag=diag, batched=False,
                               hermitian=True)[:2]      # 3-D grids: the planes k0 >= 0 only (efgp_cg_solve_hermitian)
        beta, iters = res
        self._beta = beta.to(cdtype) if cdtype != torch.complex128 else beta
        self._xis = (grid, rdtype)                 # the (M, d) node tensor is built when somebody asks for it (property below)
        self._ws = grid.ws.to(cdtype) if cdtype != torch.complex128 else grid.ws
        self._toeplitz = toeplitz
        self._fit_state = dict(h=grid.h, mtot=grid.mtot, d=d, sig=sig, ws=grid.ws, beta=beta,
                               hypers=self._current_hypers(), Fy=Fy, v=v)
        self._last_fit_stats = dict(mean_cg_iters=iters, mtot=grid.mtot, feature_count=grid.M, h=grid.h)
        self._fitted = True
        self._update_param_cache()
        if d == 2 and prod(int(f) for f in toeplitz._op.fft_shape) > 4096 and not isinstance(iters, int):
            # cooperative launch (128^2..512^2): a grid barrier that could not get its workgroups resident together leaves -3 in the
            # count and NaN in beta.  Nothing downstream reads the count before using beta, so it is read HERE -- last, behind the
            # host's own bookkeeping, which thereby still overlaps the solve -- and a dead solve goes through the synchronous
            # multi-launch iteration.
            try:
                iters.rows
            except RuntimeError as err:
                if "cooperative CG" not in str(err):
                    raise
                cidx = _center_flat(v)
                diag, rhs = gradient_prepare(grid.ws, Fy, v.reshape(-1)[cidx:cidx + 1], sig, want_diag=use_precond)
                beta, iters = cg_solve(toeplitz._op, grid.ws, sig, 0, rhs, None, tol, early_stop=True, diag=diag, batched=False,
                                       hermitian=True)[:2]
                self._beta = beta.to(cdtype) if cdtype != torch.complex128 else beta
                self._fit_state["beta"] = beta
                self._last_fit_stats["mean_cg_iters"] = iters

    @property
    def _xis(self):
        """(M, d) frequency nodes of the last fit (reference attribute `_xis`, efgpnd.py:816), materialised on first access."""
        src = self.__dict__.get("_xis_src")
        if src is None:
            return None
        if not torch.is_tensor(src):
            grid, rdtype = src
            xis = grid.xis.to(dtype=rdtype)
            xis.h_float = grid.h
            self.__dict__["_xis_src"] = src = xis
        return src

    @_xis.setter
    def _xis(self, value):
        self.__dict__["_xis_src"] = value

    def fit(self, force_recompute: bool = True):
        """Convenience: run the fit now (the reference fits lazily inside predict)."""
        self._compute_common_parameters(force_recompute=force_recompute)
        return self

    # -- predict ---------------------------------------------------------------------------------------
    def predict(self, x_new: torch.Tensor, *, return_variance: bool = True, variance_method: str = "stochastic",
                hutchinson_probes: int = 1_000, compute_log_marginal: bool = False, force_recompute: bool = False,
                do_profiling: bool = False, nufft_eps: Optional[float] = None, variance_probes: Optional[torch.Tensor] = None):
        """Posterior mean (and latent variance) at x_new -> (mean, var[, log_marginal])."""
        if x_new is None:
            raise ValueError("x_new must be provided for prediction")
        self._compute_common_parameters(force_recompute=force_recompute, nufft_eps=nufft_eps)
        st = self._fit_state
        rdtype = self.x.dtype
        cdtype = _cmplx(rdtype)
        if nufft_eps is None:
            nufft_eps = self.nufft_eps
        dev = self._devdata["dev"]
        xn = _dev_points(x_new, dev)
        B, d = xn.shape
        if d != st["d"]:
            raise ValueError(f"x_new has {d} columns, the model was built on {st['d']}")
        t0 = time.perf_counter()
        ranges = _StageRanges("EFGPND.predict", "predict_mean")
        shape = (st["mtot"],) * d                       # carried explicitly (the reference re-derives it, :908)
        # the plan over x_new is kept while the same tensor (same storage, same version) comes back with the same grid:
        # predicting at the training points after every refit is the reference's own usage (efgpnd_ex.ipynb cell 23)
        pkey = (xn.data_ptr(), tuple(xn.shape), xn._version, st["h"], float(nufft_eps))
        if self._predict_plan is not None and self._predict_plan[0] == pkey:
            plan = self._predict_plan[1]
        else:
            plan = NufftPlan(xn, st["h"], float(nufft_eps))
            self._predict_plan = (pkey, plan)
        mean = plan.type2(st["beta"], shape, real_only=True, mode_scale=st["ws"])     # F (ws * beta), efgpnd.py:919-922
        out_mean = mean.to(device=self.device, dtype=rdtype)
        t1 = time.perf_counter()
        if return_variance:
            ranges.stage("compute_variance")
            if variance_probes is None and self._shards.active and variance_method.lower() == "stochastic":
                # replicas must estimate the SAME lag sums: rank 0's draw (reference draw: efgpnd.py:1644)
                variance_probes = (torch.randint(0, 2, (hutchinson_probes, st["ws"].numel()), device=dev) * 2 - 1).to(torch.float64)
                self._shards.broadcast_(variance_probes)
            A_var = create_A_var(st["ws"], self._toeplitz, st["sig"], torch.complex128)
            var = compute_prediction_variance(
                x_new=xn, xis=self._xis.to(torch.float64), ws=st["ws"], A_var=A_var,
                cg_tol=self.opts.get("cg_tolerance", 1e-4), max_cg_iter=self.opts.get("max_cg_iterations", 1000),
                variance_method=variance_method, h=st["h"], xcen=torch.zeros(d, dtype=torch.float64),
                hutchinson_probes=hutchinson_probes, nufft_eps=nufft_eps, device=self.device, rdtype=rdtype,
                cdtype=cdtype, probes=variance_probes, shards=self._shards if self._shards.active else None)
        else:
            # the reference fills a (B,) tensor with NaN (efgpnd.py:947): same values as a stride-0 view of ONE NaN, without
            # writing 8 B bytes per call (16 us and 80 MB of traffic per predict at N = 1e7)
            if self._nan_scalar is None or self._nan_scalar.device != self.device or self._nan_scalar.dtype != rdtype:
                self._nan_scalar = torch.full((1,), float("nan"), device=self.device, dtype=rdtype)     # once per model
            var = self._nan_scalar.expand(B)
        t2 = time.perf_counter()
        ranges.close()
        if do_profiling:
            torch.cuda.synchronize(dev)
            print(f"predict_mean {t1 - t0:.6f}s  compute_variance {t2 - t1:.6f}s")
        if compute_log_marginal:
            lm = self._compute_log_marginal(beta=st["beta"], ws=st["ws"], sigmasq=st["sig"], toeplitz=self._toeplitz,
                                            device=dev, rdtype=rdtype, n=self._devdata["N"])
            return out_mean, var, lm
        return out_mean, var

    def sample_posterior(self, x_new: torch.Tensor, nsamples: int):
        """Dense O(N^3) posterior samples at x_new (small N only; reference: efgpnd.py:974-1022)."""
        x = _as2d(self.x)
        xn = _as2d(x_new)
        k = self.kernel.kernel
        sig = self.sigmasq.detach()
        K_no = k(torch.cdist(xn, x, p=2))
        K_oo = k(torch.cdist(x, x, p=2)) + sig * torch.eye(x.shape[0], dtype=x.dtype, device=x.device)
        K_nn = k(torch.cdist(xn, xn, p=2))
        cov = K_nn - K_no @ torch.linalg.solve(K_oo, K_no.T)
        cov = cov + 1e-10 * torch.eye(xn.shape[0], dtype=xn.dtype, device=xn.device)
        chol = torch.linalg.cholesky(cov)
        Zs = torch.randn(xn.shape[0], nsamples, dtype=x.dtype, device=x.device)
        mean, _ = self.predict(xn, return_variance=False)
        return (mean.unsqueeze(1) + chol @ Zs).detach().cpu().numpy()

    def _compute_log_marginal(self, beta, ws, sigmasq, toeplitz, device, rdtype, n):
        """-(logdet + sum |ws| |beta|^2)/2, the `predict`-path formula of the reference (:1050-1066)."""
        log_det = logdet_slq(ws=ws, sigma2=sigmasq, toeplitz=toeplitz, probes=self.opts.get("log_marginal_probes", 100),
                             steps=self.opts.get("log_marginal_steps", 25), dtype=rdtype, device=device, n=n,
                             probe_vectors=self.opts.get("log_marginal_probe_vectors"))
        data_fit = float((ws.abs() * (beta.abs() ** 2)).sum().real)
        return -0.5 * (log_det + data_fit)

    # -- training loop -------------------------------------------------------------------------------------
    def optimize_hyperparameters(self, *, optimizer="Adam", lr: Optional[float] = 0.1, max_iters: int = 50,
                                 min_lengthscale: float = 5e-3, log_interval: int = 10,
                                 compute_log_marginal: bool = False, verbose: bool = False,
                                 trace_samples: int = 10, **gkwargs):
        """Adam (or a supplied optimizer) on the log-space parameters using `compute_gradients`;
        history is left in ``self.training_log``.  Reference: efgpnd.py:1068-1226."""
        if isinstance(optimizer, str):
            if optimizer.lower() != "adam":
                raise ValueError(f"Unsupported optimizer string: {optimizer}. Currently supporting: 'adam'")
            opt = Adam(self._gp_params.parameters(), lr=lr)
        else:
            opt = optimizer
        hist = {"log_marginal": [], "gradients": [], "mean_cg_iters": [], "trace_cg_iters": []}

        def record():
            for name, value in self.kernel.iter_hypers():
                hist.setdefault(name, []).append(float(self.kernel.get_hyper(name)))
            hist.setdefault("sigmasq", []).append(float(self.sigmasq.detach()))

        record()
        t_start = time.time()
        print(f"Optimizing hyperparameters using {optimizer if isinstance(optimizer, str) else type(optimizer).__name__}")
        for it in range(max_iters):
            record()
            opt.zero_grad()
            want_lm = compute_log_marginal and (it % log_interval == 0 or it == max_iters - 1)
            res = self.compute_gradients(trace_samples=trace_samples, nufft_eps=self.nufft_eps, apply_gradients=True,
                                         compute_log_marginal=want_lm, verbose=verbose, **gkwargs)
            if want_lm:
                grad, lm = res
                hist["log_marginal"].append(float(lm))
            else:
                grad = res
            hist["gradients"].append([float(g) for g in grad])
            hist["mean_cg_iters"].append(self.last_gradient_stats.get("mean_cg_iters"))
            hist["trace_cg_iters"].append(self.last_gradient_stats.get("trace_cg_iters"))
            if verbose:
                print(f"  Iter {it}: Gradients = {hist['gradients'][-1]}")
            opt.step()
            with torch.no_grad():
                try:
                    idx = self._gp_params.hypers_names.index("lengthscale")
                    floor = torch.tensor(min_lengthscale, device=self._gp_params.raw.device, dtype=self._gp_params.raw.dtype)
                    if torch.exp(self._gp_params.raw[idx]) < floor:
                        self._gp_params.raw[idx].copy_(torch.log(floor))
                except (ValueError, IndexError) as e:
                    if verbose:
                        print(f"Note: Could not apply lengthscale constraint: {e}")
            if it % log_interval == 0 or it == max_iters - 1:
                parts = [f"iter {it}/{max_iters}"]
                for name, values in hist.items():
                    if not values or name == "gradients" or (name == "log_marginal" and not compute_log_marginal):
                        continue
                    if values[-1] is not None:
                        parts.append(f"{name}={values[-1]:.6g}")
                print(", ".join(parts))
        self._fitted = False
        self._cached_params = {}
        self._compute_common_parameters(force_recompute=True)
        print(f"Optimization complete after {time.time() - t_start:.2f} seconds")
        print("\nFinal hyperparameters:")
        for name, _ in self.kernel.iter_hypers():
            print(f"{name} = {float(self.kernel.get_hyper(name)):.6g}")
        print(f"sigmasq = {float(self.sigmasq.detach()):.6g}")
        self.training_log = hist
        return self


# ======================================================================================
# legacy functional entry point (signature from efgpnd_variance_shootout.py:129-137)
# ======================================================================================
def efgp_nd(x, y, sigmasq, kernel, eps, x_new, nufft_eps=1e-4, opts=None, do_profiling=False):
    """One-shot fit + predict -> (beta, xis, ytrg{'mean','var'[,'log_marginal']}, ws, toeplitz).

    ``opts``: cg_tolerance, early_stopping (ignored: the mean solve always stops early, as in the
    reference), estimate_variance, variance_method, hutchinson_probes, max_cg_iter, compute_log_marginal."""
    opts = dict(opts or {})
    mopts = {"cg_tolerance": opts.get("cg_tolerance", 1e-4), "mean_cg_warm_start": False}
    if "max_cg_iter" in opts:
        mopts["max_cg_iterations"] = opts["max_cg_iter"]
    model = EFGPND(_as2d(x), y, kernel, sigmasq=float(sigmasq), eps=eps, nufft_eps=nufft_eps, opts=mopts,
                   estimate_params=False)
    want_var = bool(opts.get("estimate_variance", False))
    res = model.predict(_as2d(x_new), return_variance=want_var,
                        variance_method=opts.get("variance_method", "stochastic"),
                        hutchinson_probes=opts.get("hutchinson_probes", 1000),
                        compute_log_marginal=bool(opts.get("compute_log_marginal", False)), do_profiling=do_profiling)
    ytrg = {"mean": res[0], "var": res[1]}
    if len(res) > 2:
        ytrg["log_marginal"] = res[2]
    return model._beta, model._xis, ytrg, model._ws, model._toeplitz
