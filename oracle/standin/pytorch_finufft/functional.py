"""Exact NUDFT with the call signature of pytorch_finufft.functional (oracle only).

Semantics restated from the FINUFFT documentation (the algorithm itself is a
third-party dependency absent from /root/reference; `setup.py:19-20` pins
"finufft>=1.2.0", "pytorch-finufft" unpinned):

  type 1:  f[k]  = sum_n c_n exp(isign * i * k . p_n)      (no 1/N factor)
  type 2:  c_n   = sum_k f[k] exp(isign * i * k . p_n)

with p_n in radians, and per dimension of size n the mode set
  modeord=False (CMCL): k = -(n//2) ... (n-1)//2  (increasing)
  modeord=True  (FFT) : k = 0 ... (n-1)//2, -(n//2) ... -1

Call sites in the reference that this must serve:
  efgpnd.py:1496-1499 (type1, isign=-1, modeord=False),
  efgpnd.py:1533-1536 and 1546-1549 (type2, isign=+1, modeord=False),
  efgpnd.py:1679 (type2, isign=+1, modeord=True).

The sums are evaluated in separable form (one complex exponential table per
dimension, contracted with einsum) in chunks over the points, in complex128.
"""
import torch

_CHUNK = 1 << 15


def _modes(n, modeord, device):
    if modeord:
        k = torch.cat([torch.arange(0, (n - 1) // 2 + 1), torch.arange(-(n // 2), 0)])
    else:
        k = torch.arange(-(n // 2), (n - 1) // 2 + 1)
    return k.to(device=device, dtype=torch.float64)


def _tables(points, shape, isign, modeord, lo, hi):
    tabs = []
    for a, n in enumerate(shape):
        k = _modes(n, modeord, points.device)
        ang = points[a, lo:hi].to(torch.float64)[:, None] * k[None, :]
        tabs.append(torch.polar(torch.ones_like(ang), float(isign) * ang))
    return tabs


def finufft_type1(points, values, output_shape, *, eps=1e-6, isign=-1, modeord=False, **_):
    if points.ndim == 1:
        points = points[None, :]
    if isinstance(output_shape, int):
        output_shape = (output_shape,)
    output_shape = tuple(int(s) for s in output_shape)
    d, N = points.shape
    assert d == len(output_shape)
    batched = values.ndim > 1
    v = values.reshape(-1, N)
    cdt = torch.complex64 if v.dtype in (torch.float32, torch.complex64) else torch.complex128
    v = v.to(torch.complex128)
    out = torch.zeros((v.shape[0],) + output_shape, dtype=torch.complex128, device=points.device)
    for lo in range(0, N, _CHUNK):
        hi = min(N, lo + _CHUNK)
        t = _tables(points, output_shape, isign, modeord, lo, hi)
        c = v[:, lo:hi]
        if d == 1:
            out += c @ t[0]
        elif d == 2:
            out += torch.einsum('bn,nk,nl->bkl', c, t[0], t[1])
        elif d == 3:
            out += torch.einsum('bn,nk,nl,nm->bklm', c, t[0], t[1], t[2])
        else:
            raise ValueError("only d<=3")
    out = out.to(cdt)
    return out if batched else out[0]


def finufft_type2(points, targets, *, eps=1e-6, isign=+1, modeord=False, **_):
    if points.ndim == 1:
        points = points[None, :]
    d, N = points.shape
    shape = tuple(targets.shape[-d:])
    batched = targets.ndim > d
    f = targets.reshape((-1,) + shape)
    cdt = torch.complex64 if f.dtype in (torch.float32, torch.complex64) else torch.complex128
    f = f.to(torch.complex128)
    out = torch.empty((f.shape[0], N), dtype=torch.complex128, device=points.device)
    for lo in range(0, N, _CHUNK):
        hi = min(N, lo + _CHUNK)
        t = _tables(points, shape, isign, modeord, lo, hi)
        if d == 1:
            out[:, lo:hi] = f @ t[0].T
        elif d == 2:
            out[:, lo:hi] = torch.einsum('bkl,nk,nl->bn', f, t[0], t[1])
        elif d == 3:
            out[:, lo:hi] = torch.einsum('bklm,nk,nl,nm->bn', f, t[0], t[1], t[2])
        else:
            raise ValueError("only d<=3")
    out = out.to(cdt)
    return out if batched else out[0]
