"""ctypes binding of libefgp_hip.so (the HIP/gfx950 back end of the EFGP solve path).

`efgp_hip.lib()` returns the loaded C-ABI library (declared in include/efgp_hip.h) and raises
RuntimeError when it has not been built; the product path has no CPU fallback.
"""
from .lib import lib, library_path, EfgpError, declared_symbols  # noqa: F401
from .ops import lanczos, lag_sums, variance_rhs, variance_contract, RcclComm, NufftPlan, PointSet, ToeplitzOp, cg_solve, cg_solve_async, cg_solve_mean_async, LazyIterations, vdot_real, compute_device, require_gpu, kernel_timing, kernel_timing_read, rademacher_fill, cg_residual_history, gradient_prepare, gradient_assemble  # noqa: F401
from . import cpu_quota  # noqa: E402

TORCH_THREADS_SET = cpu_quota.apply()     # None unless torch's CPU pool exceeded the cgroup CPU quota (see cpu_quota.py)
