"""Log-space container for kernel hyper-parameters plus the noise variance.

Mirrors the behaviour of the reference's `GPParams` (kernels/kernel_params.py:9-55): one
``nn.Parameter`` ``raw = log([kernel hypers..., sigma^2])`` in the *current default dtype*,
``pos = exp(raw)``, ``sig2 = pos[-1]``; constructing it binds it to the kernel so that
``kernel.get_hyper`` reads through it.
"""
import torch
from torch import nn


class GPParams(nn.Module):
    def __init__(self, kernel, init_sig2):
        super().__init__()
        self.kernel = kernel
        names, values = [], []
        if hasattr(kernel, "iter_hypers"):
            for name, val in kernel.iter_hypers():
                names.append(name)
                values.append(float(val))
        self.hypers_names = names
        values.append(float(init_sig2))
        self.raw = nn.Parameter(torch.log(torch.tensor(values, dtype=torch.get_default_dtype())))
        if hasattr(kernel, "_gp_params_ref"):
            kernel._gp_params_ref = self

    @property
    def pos(self):
        return self.raw.exp()

    @property
    def sig2(self):
        return self.pos[-1]
