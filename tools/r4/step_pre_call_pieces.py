"""Host microseconds of the Python pieces in front of efgp_gradient_step (20000 repetitions each, N = 1e6 model)."""
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (".", "gp-quadrature_amd"):
    sys.path.insert(0, os.path.join(R, p))
import torch  # noqa: E402
from bench import synth, LS, VAR, SIG2, EPS, NUFFT_TOL  # noqa: E402
import efgpnd as E  # noqa: E402
from efgpnd import EFGPND  # noqa: E402
from kernels.squared_exponential import SquaredExponential  # noqa: E402
from efgp_hip import compute_device  # noqa: E402

dev = torch.device("cuda", 0)
x, y = synth(1_000_000, 2, 1000, dev)
m = EFGPND(x, y, SquaredExponential(dimension=2, init_lengthscale=LS, init_variance=VAR), sigmasq=SIG2, eps=EPS, nufft_eps=NUFFT_TOL,
           estimate_params=False)
for _ in range(50):
    m.compute_gradients(trace_samples=5, cg_tol=1e-3)
torch.cuda.synchronize()
dd = m._device_data()
k = m.kernel


def t(label, fn, n=5000):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    print(f"{label:48s} {1e6 * (time.perf_counter() - t0) / n:7.2f} us")


t("_update_param_cache", m._update_param_cache)
t("_device_data + _layout", lambda: (m._device_data(), m._layout()))
t("host_pos", m._gp_params.host_pos)
t("kernel.get_hypers", k.get_hypers)
t("compute_device + _dev_points + y.to", lambda: (compute_device(x, device=None), E._dev_points(x, dev), y.detach().to(device=dev, dtype=torch.float64).contiguous()))
t("_StageRanges", lambda: E._StageRanges("efgpnd_gradient_batched", "0_book_keeping"))
t("_Grid(defer_weights)", lambda: E._Grid(k, EPS, dd["L"], 2, dev, want_grad=True, defer_weights=True))
t("get_xis", lambda: E.get_xis(kernel_obj=k, eps=EPS, L=dd["L"], use_integral=True, l2scaled=False))
t("_builtin_kernel_constants", lambda: E._builtin_kernel_constants(k))
t("torch.randint(...).item()", lambda: int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item()))
t("3 x torch.empty on the device", lambda: (torch.empty(10, dtype=torch.float64, device=dev), torch.empty(529, dtype=torch.complex128, device=dev), torch.empty(11, dtype=torch.int32, device=dev)))
t("raw.detach().exp() * grads; clone; assign", lambda: setattr(m._gp_params.raw, "grad", (torch.ones(3) * m._gp_params.raw.detach().exp()).clone()))
st = {}
t("stats dict update", lambda: st.update({"a": 1, "b": 2, "c": 3, "d": 4, "e": 5, "f": 6, "g": 7, "h": 8, "i": 9, "stage_sec": dict(a=1, b=2, c=3)}))
