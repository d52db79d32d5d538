"""Matern kernel (closed forms for nu in {1/2, 3/2, 5/2}) and its spectral density.

Behaviour follows the reference (kernels/matern.py:53-69 kernel, :100-123 spectral density,
:125-168 gradient, :210-264 median-distance initialisation).  The reference's general-nu branch
(:73-98) references undefined names and cannot run; here it raises a clear error instead.
"""
import math

import torch

from .kernel import Kernel


class Matern(Kernel):
    hypers = ["lengthscale", "variance"]
    num_hypers = 3

    def __init__(self, *, dimension, nu=2.5, **kwargs):
        nu = float(nu)
        if not (0.1 <= nu <= 10.0):
            raise ValueError(f"nu must lie in [0.1, 10], got {nu}")
        self.nu = nu
        super().__init__(dimension=dimension, **kwargs)

    lengthscale = property(lambda self: self.get_hyper("lengthscale"),
                           lambda self, v: self.set_hyper("lengthscale", v))
    variance = property(lambda self: self.get_hyper("variance"),
                        lambda self, v: self.set_hyper("variance", v))

    def kernel(self, distance):
        ell, var = self.lengthscale, self.variance
        s = torch.abs(distance) / ell
        if self.nu == 0.5:
            return var * torch.exp(-s)
        if self.nu == 1.5:
            return var * (1 + math.sqrt(3) * s) * torch.exp(-math.sqrt(3) * s)
        if self.nu == 2.5:
            return var * (1 + math.sqrt(5) * s + 5 * s ** 2 / 3) * torch.exp(-math.sqrt(5) * s)
        raise NotImplementedError("Matern kernel values are implemented for nu in {0.5, 1.5, 2.5}")

    def _scaling(self, ell):
        nu, d = self.nu, self.dimension
        return ((2 * math.sqrt(math.pi)) ** d * math.gamma(nu + d / 2) * (2 * nu) ** nu
                / (math.gamma(nu) * ell ** (2 * nu)))

    def spectral_density(self, xid):
        if xid.ndim == 1:
            xid = xid.unsqueeze(-1)
        ell, var = self.lengthscale, self.variance
        q = torch.sum(xid ** 2, dim=-1)
        return var * self._scaling(ell) * (2 * self.nu / ell ** 2 + (4 * math.pi ** 2) * q) ** (-(self.nu + self.dimension / 2))

    def spectral_grad(self, xid):
        if xid.ndim == 1:
            xid = xid.unsqueeze(-1)
        ell, var = self.lengthscale, self.variance
        nu, d = self.nu, self.dimension
        S = self.spectral_density(xid)
        q = torch.sum(xid ** 2, dim=-1)
        den = 2 * nu / ell ** 2 + (4 * math.pi ** 2) * q
        d_ell = S * (-2 * nu / ell + (-(nu + d / 2)) * (-4 * nu / ell ** 3) / den)
        return torch.stack([d_ell, S / var], dim=-1)

    def _k_scalar(self, r, ell, var):
        s = abs(r) / ell
        if self.nu == 0.5:
            return var * math.exp(-s)
        if self.nu == 1.5:
            return var * (1 + math.sqrt(3) * s) * math.exp(-math.sqrt(3) * s)
        if self.nu == 2.5:
            return var * (1 + math.sqrt(5) * s + 5 * s ** 2 / 3) * math.exp(-math.sqrt(5) * s)
        raise NotImplementedError("Matern kernel values are implemented for nu in {0.5, 1.5, 2.5}")

    def _S_scalar(self, r, ell, var):
        return var * self._scaling(ell) * (2 * self.nu / ell ** 2 + (4 * math.pi ** 2) * (r * r)) ** (-(self.nu + self.dimension / 2))

    def log_marginal(self, x, y, sigmasq):
        return self._dense_log_marginal(x, y, sigmasq)

    def estimate_hyperparameters(self, x, y, K=1000):
        y_var = torch.var(y).item()
        dists, mask = self._median_distance(x, K)
        med = torch.median(dists[mask]).item() if mask.sum() > 0 else 1.0
        return med, y_var, 0.2 * y_var
