"""The second training step of a model (the one that builds the per-model point layout) takes 4.2-4.6 ms at N = 1e6 against 0.54 ms
for a steady step: where?  cProfile of steps 0, 1, 2 (host side; every device wait shows as time inside the call that waits)."""
import cProfile
import os
import pstats
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (".", "gp-quadrature_amd"):
    sys.path.insert(0, os.path.join(R, p))
import torch  # noqa: E402
from torch.optim import Adam  # noqa: E402
from bench import synth  # noqa: E402
from efgpnd import EFGPND  # noqa: E402

dev = torch.device("cuda", 0)
x, y = synth(1_000_000, 2, 1000, dev)
for rep in range(2):
    torch.manual_seed(1234)
    model = EFGPND(x, y, kernel="SquaredExponential", eps=1e-4)
    opt = Adam(model.parameters(), lr=0.1)
    for it in range(4):
        pr = cProfile.Profile()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pr.enable()
        opt.zero_grad()
        model.compute_gradients(trace_samples=5, cg_tol=1e-3)
        opt.step()
        torch.cuda.synchronize()
        pr.disable()
        dt = 1e3 * (time.perf_counter() - t0)
        print(f"model {rep} step {it}: {dt:.3f} ms", flush=True)
        if rep == 1 and it in (0, 1):
            pstats.Stats(pr).sort_stats("tottime").print_stats(8)
