"""Covariance kernels with spectral densities, and the log-space hyper-parameter container.

Same import surface as the reference package (`kernels/__init__.py:1-11`):
``from kernels import Kernel, Matern, SquaredExponential, GPParams``.
"""
from .kernel import Kernel
from .matern import Matern
from .squared_exponential import SquaredExponential
from .kernel_params import GPParams

__all__ = ["Kernel", "Matern", "SquaredExponential", "GPParams"]
