"""Which phase of a fresh process burns the CPU quota (and gets the process throttled ~100 ms later)?  Prints the cgroup's CPU usage
and throttle counters between the phases of the headline bench's start-up."""
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def stat():
    d = {}
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            k, v = line.split()
            d[k] = int(v)
    except OSError:
        pass
    return d


last = [stat(), time.perf_counter()]


def mark(name):
    s, t = stat(), time.perf_counter()
    du = (s.get("usage_usec", 0) - last[0].get("usage_usec", 0)) / 1e6
    print(f"{name:28s} wall {t - last[1]:6.3f} s  cpu {du:6.3f} s  ({du / max(t - last[1], 1e-9):5.1f} cores)  throttled periods "
          f"{s.get('nr_throttled', 0)}", flush=True)
    last[0], last[1] = s, t


for p in (".", "gp-quadrature_amd"):
    sys.path.insert(0, os.path.join(R, p))
import torch  # noqa: E402
mark("import torch")
print("torch threads", torch.get_num_threads(), "OMP_NUM_THREADS", os.environ.get("OMP_NUM_THREADS"), "cpus", os.cpu_count())
import efgp_hip  # noqa: E402,F401
from efgpnd import EFGPND  # noqa: E402
mark("import efgp_hip, efgpnd")
print("torch threads", torch.get_num_threads())
from bench import synth, LS, VAR, SIG2, EPS, NUFFT_TOL, CG_TOL  # noqa: E402
from kernels.squared_exponential import SquaredExponential  # noqa: E402
torch.cuda.init()
_ = torch.zeros(1, device="cuda")
torch.cuda.synchronize()
mark("cuda init")
x, y = synth(1_000_000, 2, 1000, torch.device("cuda", 0))
torch.cuda.synchronize()
mark("synth (CPU) + upload")
model = EFGPND(x, y, SquaredExponential(dimension=2, init_lengthscale=LS, init_variance=VAR), sigmasq=SIG2, eps=EPS, nufft_eps=NUFFT_TOL,
               estimate_params=False, opts={"cg_tolerance": CG_TOL, "mean_cg_warm_start": False})
mark("model")
for blk in range(12):
    for _ in range(25):
        model._compute_common_parameters(force_recompute=True)
        model.predict(x, return_variance=False)
    torch.cuda.synchronize()
    mark(f"steps {25 * blk}..{25 * blk + 24}")
