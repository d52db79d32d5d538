"""Per-step wall times of bench.train_loop's shape (50 Adam steps, N = 1e6): which steps carry the loop's total."""
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (".", "gp-quadrature_amd"):
    sys.path.insert(0, os.path.join(R, p))
import torch  # noqa: E402
from bench import synth  # noqa: E402
from efgpnd import EFGPND  # noqa: E402
from torch.optim import Adam  # noqa: E402

dev = torch.device("cuda", 0)
x, y = synth(1_000_000, 2, 1000, dev)
for rep in range(2):
    torch.manual_seed(1234)
    model = EFGPND(x, y, kernel="SquaredExponential", eps=1e-4)
    opt = Adam(model.parameters(), lr=0.1)
    torch.cuda.synchronize()
    ms, info = [], []
    t0 = time.perf_counter()
    for it in range(50):
        t1 = time.perf_counter()
        opt.zero_grad()
        if it > 40:
            model.compute_gradients(trace_samples=10)
        else:
            model.compute_gradients(trace_samples=5, cg_tol=1e-3)
        t2 = time.perf_counter()
        opt.step()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        ms.append((1e3 * (t3 - t1), 1e3 * (t2 - t1), 1e3 * (t3 - t2)))
        st = model.last_gradient_stats
        info.append((int(st["mtot"]), int(st["mean_cg_iters"]), int(st["trace_cg_iters"])))
    total = time.perf_counter() - t0
    print(f"--- loop {rep}: total {1e3 * total:.2f} ms")
    for i, (m, inf) in enumerate(zip(ms, info)):
        print(f"step {i:2d}: {m[0]:7.3f} ms (gradient {m[1]:6.3f}, optimizer + sync {m[2]:6.3f})  mtot {inf[0]:3d}  mean iters {inf[1]:4d}  trace iters {inf[2]:4d}")
