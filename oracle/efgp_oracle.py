"""CPU oracle for the EFGP solve path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file is a from-scratch CPU restatement (numpy + torch-CPU, float64) of the
algorithm of danbider/gp-quadrature's `efgpnd.py` hot path.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it;
the product package (`gp-quadrature_amd/`) never does and fails loudly when its
HIP library is missing.

Parity status: PINNED.  `oracle/gen_golden.py` runs the reference's own
`efgpnd.py` / `cg.py` / `utils/kernels.py` / `kernels/*.py` in the build
container (with the exact-NUDFT stand-in of `oracle/standin` substituted for the
absent third-party `pytorch_finufft`, see that package's docstring) and stores
its outputs under `tests/golden/`; `tests/test_oracle_golden.py` checks every
function below against those vectors.  The third-party NUFFT (FINUFFT,
`setup.py:19-20`, ">=1.2.0", pytorch-finufft unpinned) is restated as the exact
transform it approximates; the reference holds no stored vectors for it.

Each function cites the reference lines it follows (paths relative to
/root/reference).  The structure is deliberately different from the reference
(plain functions over arrays, no classes) -- it restates behaviour, not code.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Optional, Sequence, Tuple

import numpy as np
import torch

TWO_PI = 2.0 * math.pi
_CD = torch.complex128
_RD = torch.float64


# --------------------------------------------------------------------------------------
# covariance kernels and their spectral densities
#   kernels/squared_exponential.py:46-123, kernels/matern.py:53-168
# --------------------------------------------------------------------------------------
@dataclass
class KernelSpec:
    """Effective (already exponentiated) hyper-parameters of a stationary kernel."""
    kind: str            # "se" | "matern"
    dim: int
    lengthscale: float
    variance: float
    nu: float = 2.5      # Matern only (0.5, 1.5, 2.5 closed forms)

    # k(r) -- squared_exponential.py:46-63, matern.py:53-69
    def k(self, r):
        r = np.asarray(r, dtype=np.float64)
        s = np.abs(r) / self.lengthscale
        if self.kind == "se":
            return self.variance * np.exp(-0.5 * s ** 2)
        if self.nu == 0.5:
            return self.variance * np.exp(-s)
        if self.nu == 1.5:
            return self.variance * (1 + math.sqrt(3) * s) * np.exp(-math.sqrt(3) * s)
        if self.nu == 2.5:
            return self.variance * (1 + math.sqrt(5) * s + 5 * s ** 2 / 3) * np.exp(-math.sqrt(5) * s)
        raise ValueError("only nu in {0.5,1.5,2.5}")

    # S(xi) -- squared_exponential.py:65-93, matern.py:100-123
    def S(self, xi):
        xi = np.asarray(xi, dtype=np.float64)
        if xi.ndim <= 1:
            xi = xi.reshape(-1, 1)
        q = np.sum(xi ** 2, axis=-1)
        l, d = self.lengthscale, self.dim
        if self.kind == "se":
            pref = (TWO_PI * l ** 2) ** (d / 2) * self.variance
            return pref * np.exp(-(TWO_PI ** 2) * l ** 2 * q / 2)
        nu = self.nu
        scaling = ((2 * math.sqrt(math.pi)) ** d * math.gamma(nu + d / 2) * (2 * nu) ** nu
                   / (math.gamma(nu) * l ** (2 * nu)))
        return self.variance * scaling * (2 * nu / l ** 2 + (4 * math.pi ** 2) * q) ** (-(nu + d / 2))

    # dS/d(lengthscale, variance) -- squared_exponential.py:95-123, matern.py:125-168
    def dS(self, xi):
        xi = np.asarray(xi, dtype=np.float64)
        if xi.ndim <= 1:
            xi = xi.reshape(-1, 1)
        q = np.sum(xi ** 2, axis=-1)
        S = self.S(xi)
        l, d = self.lengthscale, self.dim
        if self.kind == "se":
            dl = S * (d / l - (TWO_PI ** 2) * l * q)
        else:
            nu = self.nu
            den = 2 * nu / l ** 2 + (4 * math.pi ** 2) * q
            dl = S * (-2 * nu / l + (-(nu + d / 2)) * (-4 * nu / l ** 3) / den)
        return np.stack([dl, S / self.variance], axis=-1)


# --------------------------------------------------------------------------------------
# quadrature grid -- utils/kernels.py:7-69 (bisection), :72-143 (get_xis, use_integral=True)
# --------------------------------------------------------------------------------------
def truncation_bound(eps: float, f: Callable[[float], float], hi: float = 1000.0, lo: float = 0.0,
                     iters: int = 200) -> float:
    a, b = lo, hi
    for _ in range(10):                       # utils/kernels.py:39-43
        if f(b) > eps:
            b *= 2
        else:
            break
    mid = 0.5 * (a + b)
    for _ in range(iters):                    # utils/kernels.py:58-67
        mid = (a + b) / 2
        if f(mid) > eps:
            a = mid
        else:
            b = mid
    return mid


def get_xis(kern: KernelSpec, eps: float, L: float, trunc_eps: Optional[float] = None):
    """-> (xis_1d float64 (mtot,), h, mtot); utils/kernels.py:94-105,136-143."""
    if trunc_eps is None:
        trunc_eps = eps
    Ltime = truncation_bound(eps, lambda r: float(kern.k(r)))
    h = 1.0 / (L + Ltime)
    S0 = float(kern.S(np.zeros(1))[0])
    d = kern.dim
    Lfreq = truncation_bound(trunc_eps, lambda r: abs(r ** (d - 1)) * float(kern.S(np.array([r]))[0]) / S0)
    hm = math.ceil(Lfreq / h)
    xis = np.arange(-hm, hm + 1, dtype=np.float64) * h
    return xis, h, int(xis.size)


def tensor_grid(xis_1d: np.ndarray, d: int) -> np.ndarray:
    """(M,d) row-major grid, last dim fastest -- efgpnd.py:767-768."""
    g = np.meshgrid(*([xis_1d] * d), indexing="ij")
    return np.stack(g, axis=-1).reshape(-1, d)


def feature_weights(kern: KernelSpec, xis_1d: np.ndarray, h: float) -> np.ndarray:
    """ws = sqrt(S(xi) h^d) (real, >=0) -- efgpnd.py:778-780."""
    return np.sqrt(kern.S(tensor_grid(xis_1d, kern.dim)) * h ** kern.dim)


# --------------------------------------------------------------------------------------
# exact non-uniform DFTs (what FINUFFT approximates; efgpnd.py:1454-1549, :1679)
# --------------------------------------------------------------------------------------
def _modes(n: int, fft_order: bool) -> torch.Tensor:
    if fft_order:
        k = torch.cat([torch.arange(0, (n - 1) // 2 + 1), torch.arange(-(n // 2), 0)])
    else:
        k = torch.arange(-(n // 2), (n - 1) // 2 + 1)
    return k.to(_RD)


def _phase_tables(x: torch.Tensor, h: float, shape: Sequence[int], sign: float, fft_order: bool):
    tabs = []
    for a, n in enumerate(shape):
        ang = (TWO_PI * h * x[:, a])[:, None] * _modes(n, fft_order)[None, :]
        tabs.append(torch.polar(torch.ones_like(ang), sign * ang))
    return tabs


def nudft_type1(x, h, c, shape, chunk: int = 1 << 15) -> torch.Tensor:
    """f[k] = sum_n c_n exp(-2 pi i h k.x_n), CMCL order; c (N,) or (B,N) -> (B?,*shape).
    efgpnd.py:1451 (phi = 2 pi h (x - 0)), :1496-1499 (isign=-1, modeord=False)."""
    x = torch.as_tensor(x, dtype=_RD)
    if x.ndim == 1:
        x = x[:, None]
    c = torch.as_tensor(c)
    batched = c.ndim > 1
    cc = c.reshape(-1, x.shape[0]).to(_CD)
    d = x.shape[1]
    out = torch.zeros((cc.shape[0],) + tuple(shape), dtype=_CD)
    for lo in range(0, x.shape[0], chunk):
        hi = min(x.shape[0], lo + chunk)
        t = _phase_tables(x[lo:hi], h, shape, -1.0, False)
        cb = cc[:, lo:hi]
        if d == 1:
            out += cb @ t[0]
        elif d == 2:
            out += torch.einsum("bn,nk,nl->bkl", cb, t[0], t[1])
        else:
            out += torch.einsum("bn,nk,nl,nm->bklm", cb, t[0], t[1], t[2])
    return out if batched else out[0]


def nudft_type2(x, h, f, shape, fft_order: bool = False, chunk: int = 1 << 15) -> torch.Tensor:
    """c_n = sum_k f[k] exp(+2 pi i h k.x_n); f (prod,)|(*shape)|(B,...) -> (N,)|(B,N).
    efgpnd.py:1533-1549 (isign=+1, modeord=False), :1679 (modeord=True)."""
    x = torch.as_tensor(x, dtype=_RD)
    if x.ndim == 1:
        x = x[:, None]
    d = x.shape[1]
    f = torch.as_tensor(f).to(_CD)
    batched = not (f.ndim == 1 or tuple(f.shape) == tuple(shape))
    ff = f.reshape((-1,) + tuple(shape))
    out = torch.empty((ff.shape[0], x.shape[0]), dtype=_CD)
    for lo in range(0, x.shape[0], chunk):
        hi = min(x.shape[0], lo + chunk)
        t = _phase_tables(x[lo:hi], h, shape, +1.0, fft_order)
        if d == 1:
            out[:, lo:hi] = ff @ t[0].T
        elif d == 2:
            out[:, lo:hi] = torch.einsum("bkl,nk,nl->bn", ff, t[0], t[1])
        else:
            out[:, lo:hi] = torch.einsum("bklm,nk,nl,nm->bn", ff, t[0], t[1], t[2])
    return out if batched else out[0]


def conv_vector(x, h, m: int) -> torch.Tensor:
    """v[k] = sum_n exp(-2 pi i h k.x_n), k in [-2m,2m]^d -- efgpnd.py:1395-1421."""
    x = torch.as_tensor(x, dtype=_RD)
    if x.ndim == 1:
        x = x[:, None]
    return nudft_type1(x, h, torch.ones(x.shape[0], dtype=_RD), (4 * m + 1,) * x.shape[1])


# --------------------------------------------------------------------------------------
# Toeplitz mat-vec by circulant embedding -- efgpnd.py:1244-1301 (setup), :1331-1393 (apply)
# --------------------------------------------------------------------------------------
class Toeplitz:
    def __init__(self, v: torch.Tensor, pow2: bool = True):
        v = v.to(_CD)
        self.Ls = list(v.shape)
        self.ns = [(L + 1) // 2 for L in self.Ls]
        self.size = int(np.prod(self.ns))
        self.d = len(self.Ls)
        self.fft_shape = [1 << (L - 1).bit_length() for L in self.Ls] if pow2 else list(self.Ls)
        dims = tuple(range(-self.d, 0))
        self.vhat = torch.fft.fftn(v, s=self.fft_shape, dim=dims)      # zero-pad at the end, :1275-1284

    def __call__(self, u: torch.Tensor) -> torch.Tensor:
        if u.shape[-1] == self.size:                       # flat layout wins, efgpnd.py:1345
            flat, batch = True, u.shape[:-1]
            ub = u.reshape(*batch, *self.ns)
        elif list(u.shape[-self.d:]) == self.ns:
            flat, batch, ub = False, u.shape[:-self.d], u
        else:
            raise ValueError("shape mismatch")
        dims = tuple(range(-self.d, 0))
        U = torch.fft.fftn(ub.to(_CD), s=self.fft_shape, dim=dims)
        y = torch.fft.ifftn(U * self.vhat, dim=dims)
        sl = [slice(None)] * len(batch) + [slice(n - 1, 2 * n - 1) for n in self.ns]   # :1289-1290
        y = y[tuple(sl)]
        return y.reshape(*batch, self.size) if flat else y


def make_A_mean(ws: torch.Tensor, T: Toeplitz, sigmasq: float):
    """beta -> ws*T(ws*beta) + sigma^2 beta -- efgpnd.py:1572-1600."""
    return lambda b: ws * T(ws * b) + sigmasq * b


def make_A_var(ws: torch.Tensor, T: Toeplitz, sigmasq: float):
    """gamma -> ws*T(ws*gamma)/sigma^2 + gamma -- efgpnd.py:1602-1609."""
    return lambda g: ws * T(ws * g) / sigmasq + g


def jacobi_diag(ws: torch.Tensor, sigmasq: float, diag_scale: float) -> torch.Tensor:
    """diag_scale*|ws|^2 + sigma^2 -- efgpnd.py:1619-1631."""
    return diag_scale * ws.abs().pow(2) + sigmasq


# --------------------------------------------------------------------------------------
# preconditioned conjugate gradients -- cg.py:86-153 (single), :155-244 (batched)
# --------------------------------------------------------------------------------------
DIV_EPS = 1e-16


def cg_single(A, b, x0, tol, max_iter=None, early=True, diag=None, history: Optional[list] = None):
    b = b.to(_CD)
    x = x0.to(_CD).clone()
    if max_iter is None:
        max_iter = 2 * b.numel()                         # cg.py:59-65
    r = b - A(x)
    z = r / diag if diag is not None else r.clone()
    p = z.clone()
    rz = torch.vdot(r, z).real
    bn = torch.linalg.norm(b)
    den = bn if bn > 0 else torch.tensor(1.0, dtype=_RD)
    it = 0
    for i in range(max_iter):
        it = i + 1
        Ap = A(p)
        alpha = rz / (torch.vdot(p, Ap).real + DIV_EPS)
        x = x + alpha * p
        r = r - alpha * Ap
        rel = torch.linalg.norm(r) / (den + DIV_EPS)
        if history is not None:
            history.append(float(rel))
        if early and rel < tol:                          # cg.py:132 -- before the preconditioner
            break
        z = r / diag if diag is not None else r
        rz_new = torch.vdot(r, z).real
        p = z + (rz_new / (rz + DIV_EPS)) * p
        rz = rz_new
    return x, it


def cg_batched(A, b, x0, tol, max_iter=None, early=True, diag=None):
    b = b.to(_CD)
    x = x0.to(_CD).clone()
    B, n = b.shape
    if max_iter is None:
        max_iter = 2 * n
    r = b - A(x)
    z = r / diag if diag is not None else r.clone()
    p = z.clone()
    rz = (r.conj() * z).sum(1).real
    bn = torch.linalg.norm(b, dim=1)
    den = torch.where(bn > 0, bn, torch.ones_like(bn))
    active = torch.ones(B, dtype=torch.bool)
    it = 0
    for i in range(max_iter):
        it = i + 1                                       # cg.py:243 -- counts the breaking pass too
        idx = torch.where(active)[0]
        if idx.numel() == 0:
            break
        Ap = A(p[idx])
        alpha = rz[idx] / ((p[idx].conj() * Ap).sum(1).real + DIV_EPS)
        x[idx] += alpha[:, None] * p[idx]
        r[idx] -= alpha[:, None] * Ap
        zn = r[idx] / diag if diag is not None else r[idx]
        rzn = (r[idx].conj() * zn).sum(1).real
        p[idx] = zn + (rzn / (rz[idx] + DIV_EPS))[:, None] * p[idx]
        rz[idx] = rzn
        if early:                                        # cg.py:229-241 -- after the p update
            rn = torch.linalg.norm(r[idx], dim=1)
            conv = (rn / (den[idx] + DIV_EPS) < tol) | (rn < 1e-12)
            active[idx[conv]] = False
    return x, it


# --------------------------------------------------------------------------------------
# fit / predict / variance -- efgpnd.py:710-822, 824-972, 1634-1679, 1761-1841
# --------------------------------------------------------------------------------------
@dataclass
class Fit:
    kern: KernelSpec
    sigmasq: float
    eps: float
    h: float
    mtot: int
    xis_1d: np.ndarray
    ws: torch.Tensor         # (M,) complex128 (imag 0)
    v: torch.Tensor          # conv vector (4m+1,)*d
    Fy: torch.Tensor         # F* y (M,)
    rhs: torch.Tensor
    beta: torch.Tensor
    iters: int
    T: Toeplitz


def fit(x, y, kern: KernelSpec, sigmasq: float, eps: float, cg_tol: float = 1e-4,
        precond: bool = True, x0: Optional[torch.Tensor] = None, max_iter=None) -> Fit:
    x = torch.as_tensor(x, dtype=_RD)
    if x.ndim == 1:
        x = x[:, None]
    y = torch.as_tensor(y, dtype=_RD)
    d = x.shape[1]
    L = float((x.max(0).values - x.min(0).values).max())       # efgpnd.py:751
    if L <= 1e-9:
        L = 1.0
    xis_1d, h, mtot = get_xis(kern, eps, L)
    ws = torch.from_numpy(feature_weights(kern, xis_1d, h)).to(_CD)
    shape = (mtot,) * d
    Fy = nudft_type1(x, h, y, shape).reshape(-1)
    rhs = ws * Fy                                               # efgpnd.py:786
    m = (mtot - 1) // 2
    v = conv_vector(x, h, m)                                    # efgpnd.py:789-790
    T = Toeplitz(v)
    A = make_A_mean(ws, T, sigmasq)
    center = tuple((s - 1) // 2 for s in v.shape)
    diag = jacobi_diag(ws, sigmasq, float(v[center].real)) if precond else None
    b0 = torch.zeros_like(rhs) if x0 is None else x0
    beta, it = cg_single(A, rhs, b0, cg_tol, max_iter=max_iter, diag=diag)
    return Fit(kern, sigmasq, eps, h, mtot, xis_1d, ws, v, Fy, rhs, beta, it, T)


def predict_mean(f: Fit, x_new) -> torch.Tensor:
    """Re F_new (ws*beta) -- efgpnd.py:918-922."""
    x_new = torch.as_tensor(x_new, dtype=_RD)
    if x_new.ndim == 1:
        x_new = x_new[:, None]
    return nudft_type2(x_new, f.h, f.ws * f.beta, (f.mtot,) * x_new.shape[1]).real


def variance_regular(f: Fit, x_new, cg_tol=1e-4, max_iter=1000) -> torch.Tensor:
    """efgpnd.py:1805-1820 (dense feature rows, batched unpreconditioned CG on A_var)."""
    x_new = torch.as_tensor(x_new, dtype=_RD)
    d = x_new.shape[1]
    xis = torch.from_numpy(tensor_grid(f.xis_1d, d))
    A = make_A_var(f.ws, f.T, f.sigmasq)
    out = []
    for xb in torch.split(x_new, 8192, dim=0):
        ang = TWO_PI * (xb @ xis.T)
        fx = torch.polar(torch.ones_like(ang), ang)
        rhs = f.ws * fx.conj()
        g, _ = cg_batched(A, rhs, torch.zeros_like(rhs), cg_tol, max_iter=max_iter)
        out.append((fx * (f.ws * g)).sum(-1).real.clamp_min(0.0))
    return torch.cat(out)


def lag_sums(f: Fit, etas: torch.Tensor, cg_tol=1e-4, max_iter=1000):
    """Hutchinson lag sums c[r] -- efgpnd.py:1634-1664; etas (J,M) of +-1 supplied by the caller."""
    d = f.kern.dim
    A = make_A_var(f.ws, f.T, f.sigmasq)
    etas = etas.to(_RD)
    rhs = f.ws[None, :] * etas
    us, it = cg_batched(A, rhs, torch.zeros_like(rhs), cg_tol, max_iter=max_iter)
    gam = (f.ws[None, :] * us).reshape((-1,) + (f.mtot,) * d)
    eta = etas.reshape((-1,) + (f.mtot,) * d)
    s = (2 * f.mtot - 1,) * d
    dims = tuple(range(1, d + 1))
    R = torch.fft.ifftn(torch.fft.fftn(gam, s=s, dim=dims) * torch.conj(torch.fft.fftn(eta, s=s, dim=dims)),
                        s=s, dim=dims)
    return R.mean(0), it


def variance_stochastic(f: Fit, x_new, etas, cg_tol=1e-4, max_iter=1000) -> torch.Tensor:
    """efgpnd.py:1822-1838 -> :1666-1679 (type-2, FFT mode order, real part)."""
    x_new = torch.as_tensor(x_new, dtype=_RD)
    c, _ = lag_sums(f, etas, cg_tol, max_iter)
    return nudft_type2(x_new, f.h, c, tuple(c.shape), fft_order=True).real


# --------------------------------------------------------------------------------------
# hyper-parameter gradient -- efgpnd.py:17-317 with externally supplied probes
# --------------------------------------------------------------------------------------
def gradient(x, y, kern: KernelSpec, sigmasq: float, eps: float, Z: torch.Tensor, V: torch.Tensor,
             cg_tol: Optional[float] = None, mean_x0=None, precond_mean=True, precond_trace=True):
    """Returns (grad (3,), stats).  Z (T,N) and V (T,M) are the +-1 probes the reference draws
    at efgpnd.py:179-182 and :199-202."""
    x = torch.as_tensor(x, dtype=_RD)
    if x.ndim == 1:
        x = x[:, None]
    y = torch.as_tensor(y, dtype=_RD)
    N, d = x.shape
    if cg_tol is None:
        cg_tol = eps
    L = float((x.max(0).values - x.min(0).values).max())        # efgpnd.py:72-79
    xis_1d, h, mtot = get_xis(kern, eps, L)
    grid = tensor_grid(xis_1d, d)
    ws = torch.from_numpy(np.sqrt(kern.S(grid) * h ** d)).to(_CD)
    Dp = torch.from_numpy(h ** d * kern.dS(grid)).to(_CD)        # (M,2)  efgpnd.py:99
    shape = (mtot,) * d
    fadj = lambda c: nudft_type1(x, h, c, shape).reshape(*(c.shape[:-1]), -1)
    fwd = lambda f: nudft_type2(x, h, f, shape)
    v = conv_vector(x, h, (mtot - 1) // 2)
    T = Toeplitz(v)
    A = make_A_mean(ws, T, sigmasq)
    center = tuple((s - 1) // 2 for s in v.shape)
    diag = jacobi_diag(ws, sigmasq, float(v[center].real))
    # mean solve, efgpnd.py:132-152
    Fy = fadj(y)
    rhs = ws * Fy
    b0 = torch.zeros_like(rhs) if mean_x0 is None else mean_x0
    beta, it_mean = cg_single(A, rhs, b0, cg_tol, diag=diag if precond_mean else None)
    beta_raw = beta.clone()
    beta = beta * ws
    alpha = (y.to(_CD) - fwd(beta)) / sigmasq
    # term 2, efgpnd.py:156-172
    fa = (Fy - T(beta)) / sigmasq
    term2 = torch.zeros(3, dtype=_RD)
    term2[0] = torch.vdot(fa, Dp[:, 0] * fa).real
    an = torch.vdot(alpha, alpha).real
    ya = torch.vdot(y.to(_CD), alpha).real
    term2[1] = (ya - sigmasq * an) / kern.variance
    term2[2] = an
    # term 1, efgpnd.py:175-256 (lengthscale by data-space probes, noise by feature-space probes)
    Tn = Z.shape[0]
    Zc = Z.to(_CD)
    FZ = fadj(Zc).reshape(Tn, -1)
    DFZ = Dp[:, 0] * FZ
    rhs_k = fwd(DFZ).reshape(Tn, -1)
    B_k = ws * T(DFZ)
    Vc = V.to(_CD)
    B_n = ws * T(ws * Vc)
    Ball = torch.cat([B_k, B_n], 0)
    Beta_all, it_tr = cg_batched(A, Ball, torch.zeros_like(Ball), cg_tol,
                                 diag=diag if precond_trace else None)
    Bk, Bn = Beta_all[:Tn] * ws, Beta_all[Tn:]
    Alpha = (rhs_k - fwd(Bk).reshape(Tn, -1)) / sigmasq
    term1 = torch.zeros(3, dtype=_RD)
    term1[0] = (Zc * Alpha).sum(1).mean().real
    t1n = N / sigmasq - ((Vc.conj() * Bn).sum(1).real / sigmasq).mean()
    term1[1] = (N - sigmasq * t1n) / kern.variance
    term1[2] = t1n
    grad = 0.5 * (term1 - term2)
    return grad, dict(mean_cg_iters=it_mean, trace_cg_iters=it_tr, term1=term1, term2=term2,
                      beta=beta_raw, h=h, mtot=mtot)
