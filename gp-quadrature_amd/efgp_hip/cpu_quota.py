"""Keep torch's CPU thread pool inside the container's CPU quota.

The host side of a fit is a few hundred tiny torch-CPU operations (truncation bisection, frequency grid, spectral
weights).  torch sizes its intra-op pool from the visible cores (128 threads on the 256-core MI355X hosts), and after
any parallel region those OpenMP threads spin-wait.  Under a CFS bandwidth limit (cgroup `cpu.max`, 16 CPUs on the
GPU boxes) that burns the whole quota of a 100-ms period in a few milliseconds and the kernel then freezes EVERY thread
of the process until the period ends: measured on MI355X as 90-ms stalls on every third or fourth fit of a 2-D
71 x 71-mode model (2.1 ms median, 20 ms mean; `cpu.stat` nr_throttled +21 over 40 fits), landing wherever the host
happened to be (the CG poll, torch.exp, torch.arange).  With the pool capped the same run has no throttled period and
a mean of 2.06 ms.

`apply()` runs once at import: if the user has not chosen a thread count (OMP_NUM_THREADS / MKL_NUM_THREADS unset) and
torch's pool is larger than the quota share of this process, the pool is reduced to half that share (at least 1).
EFGP_KEEP_TORCH_THREADS=1 disables it.
"""
import math
import os


def cpu_quota_cores(root="/sys/fs/cgroup"):
    """CPUs this cgroup may use per period (float), or None when unlimited / unknown."""
    try:                                              # cgroup v2
        with open(os.path.join(root, "cpu.max")) as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            return float(quota) / float(period)
        return None
    except (OSError, ValueError):
        pass
    try:                                              # cgroup v1
        with open(os.path.join(root, "cpu", "cpu.cfs_quota_us")) as f:
            quota = float(f.read())
        with open(os.path.join(root, "cpu", "cpu.cfs_period_us")) as f:
            period = float(f.read())
        if quota > 0 and period > 0:
            return quota / period
    except (OSError, ValueError):
        pass
    return None


def apply():
    """-> the thread count set, or None when nothing was changed."""
    if os.environ.get("EFGP_KEEP_TORCH_THREADS") or os.environ.get("OMP_NUM_THREADS") or os.environ.get("MKL_NUM_THREADS"):
        return None
    cores = cpu_quota_cores()
    if cores is None:
        return None
    try:
        ranks = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))      # one process per GPU shares the quota
    except ValueError:
        ranks = 1
    want = max(1, int(math.floor(cores / ranks / 2.0)))
    import torch
    before = torch.get_num_threads()
    if before <= want:
        return None
    torch.set_num_threads(want)
    # a process-wide setting changed at import: say so once (logging, level INFO; EFGP_KEEP_TORCH_THREADS=1 opts out)
    import logging
    logging.getLogger("efgp_hip").info("torch CPU threads %d -> %d (cgroup CPU quota %.1f cores, %d rank(s) per node); "
                                       "set EFGP_KEEP_TORCH_THREADS=1 or OMP_NUM_THREADS to keep your own value",
                                       before, want, cores, ranks)
    return want
