"""Squared-exponential kernel k(r) = variance * exp(-r^2 / (2 l^2)) and its spectral density.

Behaviour follows the reference (kernels/squared_exponential.py:46-123 for k, S, dS;
:165-216 for the 0.5 x median-distance initialisation heuristic).
"""
import math

import torch

from .kernel import Kernel

_TWO_PI = 2.0 * math.pi


class SquaredExponential(Kernel):
    hypers = ["lengthscale", "variance"]
    num_hypers = 3          # lengthscale, variance, noise variance

    lengthscale = property(lambda self: self.get_hyper("lengthscale"),
                           lambda self, v: self.set_hyper("lengthscale", v))
    variance = property(lambda self: self.get_hyper("variance"),
                        lambda self, v: self.set_hyper("variance", v))

    def kernel(self, distance):
        ell, var = self.get_hyper("lengthscale"), self.get_hyper("variance")
        return var * torch.exp(-0.5 * (distance / ell) ** 2)

    @staticmethod
    def _sqnorm(xid):
        if xid.ndim == 1:
            xid = xid.unsqueeze(-1)
        return torch.sum(xid ** 2, dim=-1)

    def spectral_density(self, xid):
        ell, var = self.get_hyper("lengthscale"), self.get_hyper("variance")
        q = self._sqnorm(xid)
        amp = (_TWO_PI * ell ** 2) ** (self.dimension / 2) * var
        return amp * torch.exp(-(_TWO_PI ** 2) * ell ** 2 * q / 2)

    def spectral_grad(self, xid):
        ell, var = self.get_hyper("lengthscale"), self.get_hyper("variance")
        q = self._sqnorm(xid)
        S = (_TWO_PI * ell ** 2) ** (self.dimension / 2) * var * torch.exp(-(_TWO_PI ** 2) * ell ** 2 * q / 2)
        d_ell = S * (self.dimension / ell - (_TWO_PI ** 2) * ell * q)
        return torch.stack([d_ell, S / var], dim=-1)

    # scalar fast paths for the grid bisection (utils/kernels.py); same formulas in Python floats
    def _k_scalar(self, r, ell, var):
        return var * math.exp(-0.5 * (r / ell) ** 2)

    def _S_scalar(self, r, ell, var):
        return (_TWO_PI * ell ** 2) ** (self.dimension / 2) * var * math.exp(-(_TWO_PI ** 2) * ell ** 2 * (r * r) / 2)

    def log_marginal(self, x, y, sigmasq):
        return self._dense_log_marginal(x, y, sigmasq)

    def estimate_hyperparameters(self, x, y, K=1000):
        y_var = torch.var(y).item()
        dists, mask = self._median_distance(x, K)
        return 0.5 * torch.median(dists[mask]).item(), y_var, 0.2 * y_var
