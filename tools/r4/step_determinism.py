"""Is the entry-by-entry gradient step bit-reproducible run to run, and does the one-call step reproduce it?  (per case)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "gp-quadrature_amd"))
import torch
from efgpnd import efgpnd_gradient_batched, EFGPND
from kernels.squared_exponential import SquaredExponential

def problem(d, N, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(N, d, generator=g, dtype=torch.float64).cuda()
    y = (torch.sin(3 * x.sum(1)) + 0.3 * torch.randn(N, generator=g, dtype=torch.float64).cuda()).contiguous()
    return x, y

x, y = problem(2, 20000, 3)
sig = torch.tensor(0.09, dtype=torch.float64)
def run(env, points=None):
    if env: os.environ["EFGP_NO_GRADIENT_STEP"] = "1"
    else: os.environ.pop("EFGP_NO_GRADIENT_STEP", None)
    kern = SquaredExponential(dimension=2, init_lengthscale=0.1, init_variance=1.0)
    torch.manual_seed(17)
    st = {}
    g = efgpnd_gradient_batched(x, y, sig, kern, 1e-4, 5, stats_out=st, nufft_eps=1e-5, points=points)
    return g.cpu(), st["term1"], st["term2"], st["mean_beta"].cpu()
for label, pts in (("no layout", None),):
    a, b, c = run(True, pts), run(True, pts), run(False, pts)
    for q, name in enumerate(("grad", "term1", "term2", "beta")):
        print(label, name, "entries twice:", float((a[q] - b[q]).abs().max()), " one-call vs entries:", float((a[q] - c[q]).abs().max()))
from efgp_hip import PointSet
pts = PointSet(x, values=y)
a, b, c = run(True, pts), run(True, pts), run(False, pts)
for q, name in enumerate(("grad", "term1", "term2", "beta")):
    print("layout", name, "entries twice:", float((a[q] - b[q]).abs().max()), " one-call vs entries:", float((a[q] - c[q]).abs().max()))
