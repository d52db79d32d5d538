#!/usr/bin/env python3
"""Host cost of one wrapped native call, split: raw ctypes call | + stream lookup | + device context | full wrapper."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
import torch  # noqa: E402
import efgp_hip  # noqa: E402
from efgp_hip import ToeplitzOp  # noqa: E402
from efgp_hip.ops import _ptr, _stream  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
v = torch.complex(torch.randn(45, 45, generator=g, dtype=torch.float64), torch.randn(45, 45, generator=g, dtype=torch.float64)).to(dev)
op = ToeplitzOp(v)
x = torch.complex(torch.randn(1, 529, generator=g, dtype=torch.float64), torch.randn(1, 529, generator=g, dtype=torch.float64)).to(dev)
y = torch.empty_like(x)
lib = efgp_hip.lib()
n = 2000


def bench(fn, label):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn()
        if i % 64 == 63:
            torch.cuda.synchronize()          # keep the queue short: enqueue cost, not back-pressure
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"{label:50s} {1e6 * (t1 - t0) / n:6.2f} us")


st = _stream(dev)
px, py = _ptr(x), _ptr(y)
bench(lambda: lib.efgp_toeplitz_apply_scaled(op._h, px, 0, 1, None, None, py, st), "raw ctypes call (prebuilt arguments)")
bench(lambda: lib.efgp_toeplitz_apply_scaled(op._h, _ptr(x), 0, 1, None, None, _ptr(y), _stream(dev)), "+ data_ptr, stream lookup")


def with_ctx():
    with torch.cuda.device(dev):
        lib.efgp_toeplitz_apply_scaled(op._h, _ptr(x), 0, 1, None, None, _ptr(y), _stream(dev))


bench(with_ctx, "+ torch.cuda.device context")
bench(lambda: torch.empty_like(x), "torch.empty_like alone")
bench(lambda: op.apply_scaled(x), "ToeplitzOp.apply_scaled (allocates its output)")
bench(lambda: op.apply_scaled(x, out=y), "ToeplitzOp.apply_scaled(out=)")
bench(lambda: x.mul_(1.0), "a torch elementwise op for comparison")
