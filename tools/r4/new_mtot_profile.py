"""Where does a training step with a NEW mode count spend its extra time?  (bench.py train_loop: 0.53 ms per step at a known mtot,
1.4 ms at a new one.)  Runs the loop of test_timing_profiling.py:83-111 at N = 1e6 and profiles (cProfile, host side) the steps
whose mtot differs from the previous step's, next to the steps at a repeated mtot."""
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
torch.manual_seed(1234)
model = EFGPND(x, y, kernel="SquaredExponential", eps=1e-4)
opt = Adam(model.parameters(), lr=0.1)
prof_new, prof_same = cProfile.Profile(), cProfile.Profile()
prev = None
ms_new, ms_same = [], []
for it in range(41):
    opt.zero_grad()
    # which profile this step belongs to is known only afterwards: profile into a fresh one and merge
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pr.enable()
    model.compute_gradients(trace_samples=5, cg_tol=1e-3)
    opt.step()
    torch.cuda.synchronize()
    pr.disable()
    dt = 1e3 * (time.perf_counter() - t0)
    mt = int(model.last_gradient_stats["mtot"])
    if it >= 2:
        pr.create_stats()
        tgt = prof_new if mt != prev else prof_same
        (ms_new if mt != prev else ms_same).append(dt)
        if not hasattr(tgt, "_merged"):
            tgt._merged = pstats.Stats(pr)
        else:
            tgt._merged.add(pr)
    prev = mt
print(f"steps at a new mtot: {len(ms_new)}, median {sorted(ms_new)[len(ms_new) // 2]:.3f} ms (under cProfile); "
      f"at a repeated mtot: {len(ms_same)}, median {sorted(ms_same)[len(ms_same) // 2]:.3f} ms")
for name, tgt, n in (("NEW mtot", prof_new, len(ms_new)), ("repeated mtot", prof_same, len(ms_same))):
    print(f"\n==== {name}: cumulative host time over {n} steps (top 25 by tottime) ====")
    tgt._merged.sort_stats("tottime").print_stats(25)
