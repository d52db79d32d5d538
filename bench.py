#!/usr/bin/env python3
"""Benchmark of the EFGP solve path on MI355X: GP-fits/s and CG-iter/s.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json metric "GP-fits/sec + CG-iter/sec, N=1e6 d=2 SE kernel"; inputs as in the
reference's timing driver test_timing_profiling.py:18-44): x ~ U[-1,1]^2 (float64, seeded),
y = f(x) + N(0, 0.2); SE kernel l=0.2, sigma_f^2=2, sigma^2=0.2, eps=1e-4 (-> mtot=23, M=529,
Toeplitz FFT 64^2), NUFFT tol 1e-7, CG tol 1e-4 (the reference's default).

One step = one GP fit + posterior mean at the training points:
  grid construction (host) -> ONE fused spread pass over the points for (F*y, Toeplitz vector) ->
  rocFFT + deconvolve -> Toeplitz setup -> Jacobi-PCG to tolerance -> type-2 interpolation at x.
Inputs are resident in HBM before the timed region.  With N GPUs every rank holds its own 1e6
points (weak scaling: global N = n_gpus * 1e6), the gridded partial sums are all-reduced (RCCL) and
CG is replicated; `value` counts 1e6-point fit equivalents per second, i.e. n_gpus * fits/s.

The JSON line also carries `roofline` for the dominant N-scale kernel (the fused spread launch,
timed with HIP events on its launch stream inside the library) and `cpu_baseline` (the CPU oracle
of oracle/efgp_oracle.py timed on this box's host cores on a bounded sample; rank 0, N=1 only).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)
N_PER_GPU = 1_000_000
DIM = 2
LS, VAR, SIG2, EPS = 0.2, 2.0, 0.2, 1e-4
NUFFT_TOL, CG_TOL = 1e-7, 1e-4


def synth(N, d, seed, device):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.rand(N, d, dtype=torch.float64, generator=g) * 2 - 1
    f = (torch.sin(3 * x[:, 0]) * torch.cos(4 * x[:, 1])
         + 0.5 * torch.exp(-((x[:, 0] - 0.3) ** 2 + (x[:, 1] + 0.3) ** 2) / 0.3)
         + 0.7 * torch.sin(2 * math.pi * (x[:, 0] ** 2 + x[:, 1] ** 2)))
    y = f + torch.randn(N, dtype=torch.float64, generator=g) * math.sqrt(0.2)
    return x.to(device), y.to(device)


def cpu_baseline(seed):
    """CPU oracle (port of the reference algorithm, exact NUDFT) on a bounded sample."""
    from oracle import efgp_oracle as O
    # a one-GPU box owns a 16-core share of the host; more threads only oversubscribe it
    ncores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(ncores)
    Ns = N_PER_GPU                  # the whole workload: the separable exact NUDFT takes a few seconds at N = 1e6
    x, y = synth(Ns, DIM, seed, "cpu")
    kern = O.KernelSpec("se", DIM, LS, VAR)
    t0 = time.perf_counter()
    fit = O.fit(x, y, kern, SIG2, EPS, cg_tol=CG_TOL)
    t1 = time.perf_counter()
    O.predict_mean(fit, x)
    t2 = time.perf_counter()
    # CG iteration rate of the oracle's loop on the same operator
    A = O.make_A_mean(fit.ws, fit.T, SIG2)
    diag = O.jacobi_diag(fit.ws, SIG2, float(Ns))
    t3 = time.perf_counter()
    _, its = O.cg_single(A, fit.rhs, torch.zeros_like(fit.rhs), 1e-30, max_iter=200, diag=diag)
    t4 = time.perf_counter()
    return {
        "value": 1.0 / (t2 - t0), "unit": "GP-fits/s (fit + mean at the N points, N=1e6)", "cores": ncores, "kind": "port",
        "sample": f"the full N={Ns} workload, same kernel/eps, one step: fit {t1 - t0:.2f}s ({fit.iters} CG iterations) + mean at "
                  f"N points {t2 - t1:.2f}s (exact-NUDFT oracle, torch CPU, {ncores} threads)",
        "cg_iters_per_s": its / (t4 - t3),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n-per-gpu", type=int, default=N_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--main-only", action="store_true",
                    help="timed steps only (no CG-rate / gradient / CPU-baseline extras): for rocprofv3 --pmc passes, so "
                         "that every profiled launch belongs to the fit step")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X GPU (no CPU fallback in the product path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    import efgp_hip
    from efgpnd import EFGPND
    from efgp_hip import cg_solve, kernel_timing, kernel_timing_read
    from kernels.squared_exponential import SquaredExponential

    N = args.n_per_gpu
    x, y = synth(N, DIM, 1000 + rank, dev)
    kern = SquaredExponential(dimension=DIM, init_lengthscale=LS, init_variance=VAR)
    model = EFGPND(x, y, kern, sigmasq=SIG2, eps=EPS, nufft_eps=NUFFT_TOL, estimate_params=False,
                   opts={"cg_tolerance": CG_TOL, "mean_cg_warm_start": False, "shard_points": distributed})

    def step():
        model._compute_common_parameters(force_recompute=True)
        mean, _ = model.predict(x, return_variance=False)
        return mean

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    kernel_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    spread_ms, spread_n = kernel_timing_read("spread")
    interp_ms, interp_n = kernel_timing_read("interp")
    cgs_ms, cgs_n = kernel_timing_read("cg_solve")
    kernel_timing(False)
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    fits_per_s = args.steps / elapsed
    mean_iters = model.last_fit_stats["mean_cg_iters"]
    mtot = model.last_fit_stats["mtot"]

    if args.main_only:
        if rank == 0:
            print(json.dumps({"metric": "GP-fits/sec (main-only profiling run)", "value": fits_per_s * world,
                              "ms_per_step": ms_per_step, "steps": args.steps, "warmup": args.warmup, "n_gpus": world}))
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
        return

    # fit-only time and CG iteration rate (forced 200 iterations, no early stop), untimed region
    barrier()
    t1 = time.perf_counter()
    for _ in range(5):
        model._compute_common_parameters(force_recompute=True)
    torch.cuda.synchronize(dev)
    fit_only_ms = 1e3 * (time.perf_counter() - t1) / 5
    st = model._fit_state
    rhs = st["ws"] * st["Fy"]
    diag = (float(N * world) * st["ws"].abs().pow(2).real + SIG2)
    cg_solve(model._toeplitz._op, st["ws"], SIG2, 0, rhs, torch.zeros_like(rhs), 1e-30, max_iter=50, diag=diag, batched=False)
    torch.cuda.synchronize(dev)
    t2 = time.perf_counter()
    _, its, _ = cg_solve(model._toeplitz._op, st["ws"], SIG2, 0, rhs, torch.zeros_like(rhs), 1e-300, max_iter=400,
                         early_stop=False, diag=diag, batched=False)
    torch.cuda.synchronize(dev)
    cg_iter_per_s = its / (time.perf_counter() - t2)
    # batched rate: 64 independent right-hand sides
    B = 64
    rb = rhs[None, :].repeat(B, 1) * torch.linspace(0.5, 1.5, B, device=dev, dtype=torch.float64)[:, None]
    torch.cuda.synchronize(dev)
    t3 = time.perf_counter()
    _, itb, _ = cg_solve(model._toeplitz._op, st["ws"], SIG2, 0, rb, torch.zeros_like(rb), 1e-300, max_iter=200,
                         early_stop=False, diag=diag, batched=True)
    torch.cuda.synchronize(dev)
    cg_rhs_iter_per_s = B * itb / (time.perf_counter() - t3)

    # one hyper-gradient step (the reference's training-loop unit, test_timing_profiling.py:94-111), T = 5 probes
    for _ in range(2):
        model.compute_gradients(trace_samples=5, cg_tol=1e-3)
    barrier()
    gts = []
    for _ in range(30):      # median of synchronised iterations: a single 10-ms host hiccup used to move a 10-iteration mean by 1 ms
        t4 = time.perf_counter()
        model.compute_gradients(trace_samples=5, cg_tol=1e-3)
        torch.cuda.synchronize(dev)
        gts.append(time.perf_counter() - t4)
    barrier()
    grad_step_ms = 1e3 * sorted(gts)[len(gts) // 2]

    if rank == 0:
        m = (mtot - 1) // 2
        out_bytes = 16 * (mtot ** DIM + (4 * m + 1) ** DIM)
        spread_bytes = N * (8 * DIM + 8) + out_bytes                  # x + real y read once, two mode boxes written
        unfused_bytes = 2 * N * (8 * DIM + 16) + out_bytes           # two complex-strength type-1 passes
        spread_avg_s = (spread_ms / max(spread_n, 1)) * 1e-3
        achieved = spread_bytes / spread_avg_s / 1e9 if spread_n else None
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "round1_spread_pmc.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        interp_bytes = N * (8 * DIM + 8) + 16 * mtot ** DIM          # real-only output
        ftot = 1
        for _ in range(DIM):
            ftot *= 1 << (4 * m + 1 - 1).bit_length()
        cg_bytes = 16 * (ftot + 8 * mtot ** DIM) * mean_iters        # fused-ideal bytes per iteration
        rec = {
            "metric": "GP-fits/sec (fit + posterior mean at the N training points), N=1e6 d=2 SE kernel",
            "value": fits_per_s * world,
            "unit": "1e6-point GP fits/s (whole job: n_gpus x fits/s, each fit over global N = n_gpus x 1e6)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "2D squared-exponential l=0.2 var=2 sigma2=0.2, eps=1e-4, N=1e6 per GPU synthetic "
                                   "(BASELINE configs[1] at the metric's N=1e6)",
                       "n_per_gpu": N, "global_n": N * world, "d": DIM, "mtot": mtot, "M": mtot ** DIM,
                       "nufft_tol": NUFFT_TOL, "cg_tol": CG_TOL, "mean_cg_iters": mean_iters,
                       "parallelism": f"points sharded over {world} GPU(s), CG replicated"},
            "fit_only_ms": fit_only_ms,
            "cg_iter_per_s": cg_iter_per_s,
            "cg_us_per_iter": 1e6 / cg_iter_per_s,
            "cg_rhs_iter_per_s_batch64": cg_rhs_iter_per_s,
            "points_per_s": N * world * fits_per_s,
            "gradient_step_ms_T5": grad_step_ms,
            "roofline": {"bound": "hbm", "kernel": "spread_kernel (fused F*y + Toeplitz-vector pass, LDS-resident fine grid)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "bytes_per_launch": spread_bytes, "avg_launch_us": spread_avg_s * 1e6, "launches": spread_n,
                         "achieved_vs_unfused_survey_figure": unfused_bytes / spread_avg_s / 1e9 if spread_n else None},
            "interp": {"avg_launch_us": 1e3 * interp_ms / max(interp_n, 1),
                       "achieved_GBs": interp_bytes / (1e-3 * interp_ms / max(interp_n, 1)) / 1e9 if interp_n else None},
            # the mean solve is ONE persistent launch (one CU): largest share of the step, latency/VALU bound by design,
            # priced here against the survey's per-iteration bytes 16*(F_tot + 8M) x the iterations of the launch
            "cg_solve": {"kernel": "cg_persistent_2d64_kernel (whole mean solve, one launch)",
                         "avg_launch_us": 1e3 * cgs_ms / max(cgs_n, 1), "iterations": mean_iters,
                         "bytes_per_launch": cg_bytes,
                         "achieved_GBs": cg_bytes / (1e-3 * cgs_ms / max(cgs_n, 1)) / 1e9 if cgs_n else None,
                         "frac_hbm": cg_bytes / (1e-3 * cgs_ms / max(cgs_n, 1)) / 1e9 / HBM_PEAK_GBS if cgs_n else None,
                         "share_of_step": (cgs_ms / max(cgs_n, 1)) / ms_per_step if cgs_n else None},
        }
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(1000)
        print(json.dumps(rec))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
