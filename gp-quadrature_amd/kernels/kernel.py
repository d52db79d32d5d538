"""Base class of the stationary kernels used by EFGP.

Keeps the public behaviour of the reference's pydantic `Kernel` model (kernels/kernel.py:6-241)
with plain Python: keyword construction (``dimension=``, ``init_lengthscale=``, ``init_variance=``),
hyper-parameters living in a `GPParams` that the kernel reads through, `get_hyper` / `set_hyper` /
`iter_hypers`, and the kernel-matrix helper.  Unknown keyword arguments are ignored, as the
reference's models ignore them.
"""
import math

import torch


class Kernel:
    hypers = []
    num_hypers = 1
    _init_defaults = {}

    def __init__(self, *, dimension, **kwargs):
        if not isinstance(dimension, int) or isinstance(dimension, bool) or dimension < 1:
            raise ValueError(f"dimension must be an integer >= 1, got {dimension!r}")
        self.dimension = dimension
        self.hypers = list(type(self).hypers)
        self.num_hypers = type(self).num_hypers
        self._gp_params_ref = None
        self._params_dict = {}
        for name in self.hypers:
            key = "init_" + name
            val = float(kwargs.get(key, self._init_defaults.get(key, float("nan"))))
            if not math.isnan(val) and val < 1e-6:          # reference: Field(ge=1e-6)
                raise ValueError(f"{key} must be >= 1e-6, got {val}")
            setattr(self, key, val)
            self._params_dict[name] = val
        # every kernel owns a GPParams from birth (reference: model_post_init, kernel.py:67-95)
        from .kernel_params import GPParams
        GPParams(kernel=self, init_sig2=0.1)

    # -- hyper-parameter access ------------------------------------------------------------
    def get_hyper(self, name):
        ref = self._gp_params_ref
        if ref is None:
            if name in self._params_dict:
                return self._params_dict[name]
            raise RuntimeError(f"No GPParams reference available and unknown parameter: {name}")
        if name in ref.hypers_names:
            return float(ref.pos[ref.hypers_names.index(name)].item())
        raise ValueError(f"Unknown hyperparameter: {name}")

    def get_hypers(self):
        """All hyper-parameters as Python floats in the order of `hypers`, from ONE exp of the raw parameter vector (get_hyper
        pays a tracked torch.exp, an index and an item() per name: 10 us each on the host path of every fit)."""
        ref = self._gp_params_ref
        if ref is None:
            return tuple(self._params_dict[name] for name in self.hypers)
        vals = ref.host_pos() if hasattr(ref, "host_pos") else ref.raw.detach().exp().tolist()
        return tuple(float(vals[ref.hypers_names.index(name)]) for name in self.hypers)

    def set_hyper(self, name, value):
        if name not in self.hypers:
            raise ValueError(f"Unknown hyperparameter: {name}")
        self._params_dict[name] = float(value)
        ref = self._gp_params_ref
        if ref is not None and name in ref.hypers_names:
            # the reference takes the log in the default dtype (kernel.py:137); keep that rounding
            new_val = torch.log(torch.tensor(float(value)))
            with torch.no_grad():
                ref.raw.data[ref.hypers_names.index(name)] = new_val

    def iter_hypers(self):
        for name in self.hypers:
            yield name, self._params_dict.get(name, 1.0)

    # -- to be provided by subclasses ------------------------------------------------------
    def kernel(self, distance):
        raise NotImplementedError("Subclasses must implement kernel()")

    def spectral_density(self, xid):
        raise NotImplementedError("Subclasses must implement spectral_density()")

    def spectral_grad(self, xid):
        raise NotImplementedError("Subclasses must implement spectral_grad()")

    def log_marginal(self, x, y, sigmasq):
        raise NotImplementedError("Subclasses must implement log_marginal()")

    def estimate_hyperparameters(self, x, y, K=1000):
        raise NotImplementedError("Subclasses should implement their own hyperparameter estimation strategy")

    # -- shared helpers ---------------------------------------------------------------------
    def kernel_matrix(self, x, y):
        if x.ndim == 1:
            x = x.unsqueeze(-1)
        if y.ndim == 1:
            y = y.unsqueeze(-1)
        return self.kernel(torch.cdist(x, y))

    def _dense_log_marginal(self, x, y, sigmasq):
        """-(1/2 y^T K^-1 y + sum log diag chol + n/2 log 2pi) by Cholesky; -inf if it fails."""
        if x.ndim == 1:
            x = x.unsqueeze(-1)
        n = x.shape[0]
        Kn = self.kernel_matrix(x, x) + sigmasq * torch.eye(n, device=x.device)
        try:
            chol = torch.linalg.cholesky(Kn)
            alpha = torch.cholesky_solve(y.unsqueeze(-1), chol).squeeze(-1)
            val = 0.5 * torch.sum(y * alpha) + torch.sum(torch.log(torch.diag(chol))) + 0.5 * n * math.log(2 * math.pi)
            return -val.item()
        except RuntimeError:
            return float("-inf")

    @staticmethod
    def _median_distance(x, K):
        """median non-zero pairwise distance of <= K randomly chosen rows (uses torch.randperm)."""
        if x.ndim == 1:
            x = x.unsqueeze(-1)
        n = x.shape[0]
        xs = x[torch.randperm(n)[:K]] if n > K else x
        dists = torch.cdist(xs, xs)
        return dists, dists > 0

    def __repr__(self):
        hy = ", ".join(f"{k}={v}" for k, v in self.iter_hypers())
        return f"{type(self).__name__}(dimension={self.dimension}, {hy})"
