#!/usr/bin/env python3
"""Round 3: the in-house FFT (line_fft.hip) against hipFFT on the sizes of the EFGP path, microseconds per transform (warm), and
the first-call time of each (hipFFT: run-time compilation).  usage: fft_bench.py"""
import ctypes as C
import math
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
import torch
from efgp_hip import lib
from efgp_hip.lib import check

torch.zeros(1, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(x, rank, rocfft):
    n = (C.c_longlong * rank)(*x.shape[-rank:])
    batch = x.numel() // math.prod(x.shape[-rank:])
    check(lib().efgp_fft_c2c(0, rank, n, batch, C.c_void_p(x.data_ptr()), 1, int(rocfft), st), "fft")


for rank, shape, batch in [(1, (384,), 1), (1, (2400,), 2), (2, (180, 180), 2), (2, (512, 512), 2), (2, (1024, 1024), 1), (3, (48, 48, 48), 2),
                           (3, (64, 64, 64), 1), (3, (96, 96, 96), 2), (3, (128, 128, 128), 1), (3, (128, 128, 128), 6), (3, (240, 240, 240), 2),
                           (3, (256, 256, 256), 1)]:
    x = torch.zeros(batch, *shape, dtype=torch.complex128, device="cuda")
    out = []
    for rocfft in (False, True):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(x, rank, rocfft)
        torch.cuda.synchronize()
        first = time.perf_counter() - t0
        for _ in range(3):
            run(x, rank, rocfft)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            run(x, rank, rocfft)
        torch.cuda.synchronize()
        out.append((1e3 * first, 1e6 * (time.perf_counter() - t0) / 20))
    print(f"{'x'.join(map(str, shape)):>12s} batch {batch}: in-house {out[0][1]:8.1f} us (first call {out[0][0]:7.1f} ms) | hipFFT {out[1][1]:8.1f} us (first call {out[1][0]:7.1f} ms)", flush=True)
