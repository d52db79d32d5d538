#!/usr/bin/env python3
"""Round 3: rounding error of the in-house FFT and of hipFFT against an 80-bit (numpy longdouble) direct DFT, 1-D lines.
usage: fft_accuracy.py"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
import numpy as np
import torch
from efgp_hip import lib
from efgp_hip.lib import check

torch.zeros(1, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(0)
for n in (128, 240, 512, 1024, 2400, 4096):
    x = rng.standard_normal((4, n)) + 1j * rng.standard_normal((4, n))
    k = np.arange(n, dtype=np.longdouble)
    ang = -2 * np.pi * np.outer(k, k).astype(np.longdouble) / np.longdouble(n)   # pi in double only: reduce the argument exactly instead
    kk = (np.outer(np.arange(n), np.arange(n)) % n).astype(np.longdouble)
    ang = -2 * np.longdouble(3.14159265358979323846264338327950288) * kk / np.longdouble(n)
    Wr, Wi = np.cos(ang), np.sin(ang)
    xr, xi = x.real.astype(np.longdouble), x.imag.astype(np.longdouble)
    tr = xr @ Wr - xi @ Wi
    ti = xr @ Wi + xi @ Wr
    out = []
    for rocfft in (0, 1):
        y = torch.from_numpy(x).cuda().contiguous()
        nn = (C.c_longlong * 1)(n)
        check(lib().efgp_fft_c2c(0, 1, nn, 4, C.c_void_p(y.data_ptr()), 1, rocfft, st), "fft")
        g = y.cpu().numpy()
        err = np.sqrt(np.sum((g.real - tr) ** 2 + (g.imag - ti) ** 2) / np.sum(tr ** 2 + ti ** 2))
        out.append(float(err))
    yt = torch.fft.fft(torch.from_numpy(x), dim=-1).numpy()
    et = float(np.sqrt(np.sum((yt.real - tr) ** 2 + (yt.imag - ti) ** 2) / np.sum(tr ** 2 + ti ** 2)))
    print(f"n = {n:5d}: relative l2 error  in-house {out[0]:.2e}   hipFFT {out[1]:.2e}   torch (CPU, pocketfft) {et:.2e}", flush=True)
