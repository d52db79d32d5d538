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

    def host_pos(self):
        """exp(raw) as Python floats -- the values `pos[i].item()` gives, rounded as torch rounds them in raw's dtype -- from ONE
        read of the parameter vector: the tuple is kept and handed out again as long as raw holds the same numbers (checked by
        value on every call: in-place writes through `.data` do not move a version counter).  A fit or gradient step asks for the
        hyper-parameters four or five times; each exp + item on the tracked parameter costs 5-10 us of host time."""
        now = self.raw.detach().tolist()
        kept = self.__dict__.get("_host_pos")
        if kept is None or kept[0] != now:
            kept = (now, tuple(float(v) for v in self.raw.detach().exp().tolist()))
            self.__dict__["_host_pos"] = kept
        return kept[1]

    @property
    def sig2(self):
        return self.pos[-1]
