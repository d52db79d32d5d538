"""se3 case of tests/test_gpu_gradient_step.py: entry-by-entry twice against the one-call step (run-to-run spread of beta)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "gp-quadrature_amd"))
import torch
from efgpnd import efgpnd_gradient_batched
from kernels.squared_exponential import SquaredExponential

g = torch.Generator().manual_seed(3)
x = torch.rand(5000, 3, generator=g, dtype=torch.float64).cuda()
y = (torch.sin(3 * x.sum(1)) + 0.3 * torch.randn(5000, generator=g, dtype=torch.float64).cuda()).contiguous()
sig = torch.tensor(0.09, dtype=torch.float64)

def run(env):
    if env: os.environ["EFGP_NO_GRADIENT_STEP"] = "1"
    else: os.environ.pop("EFGP_NO_GRADIENT_STEP", None)
    kern = SquaredExponential(dimension=3, init_lengthscale=0.6, init_variance=0.7)
    torch.manual_seed(17)
    st = {}
    gr = efgpnd_gradient_batched(x, y, sig, kern, 1e-2, 5, stats_out=st, nufft_eps=1e-5, cg_tol=1e-11)
    return gr.cpu(), st["term1"], st["term2"], st["mean_beta"].cpu(), int(st["mean_cg_iters"]), int(st["trace_cg_iters"]), st["mtot"]

a, b, c, d = run(True), run(True), run(False), run(False)
print("mtot", a[6], "iters entries", a[4], a[5], "one-call", c[4], c[5])
for q, name in enumerate(("grad", "term1", "term2", "beta")):
    s = float(a[q].abs().max())
    print(name, "entries twice:", float((a[q] - b[q]).abs().max()) / s, " one-call twice:", float((c[q] - d[q]).abs().max()) / s,
          " one-call vs entries:", float((a[q] - c[q]).abs().max()) / s)
