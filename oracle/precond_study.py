#!/usr/bin/env python3
"""Iteration counts of the REFERENCE's own preconditioners (prism_experiment/benchmark_prism_mean_preconditioners.py:
none / Jacobi / circulant scalar / circulant sandwich) on the golden cases, run with the reference's own code (build
container only).  Output kept in profiles/r2_reference_preconditioner_iterations.txt; it decides SURVEY 8 row f4:
the circulant preconditioners need 2x-70x MORE iterations than Jacobi on every case, so they are not fused into the HIP CG."""
import sys, os, io, contextlib
sys.path.insert(0, '/root/repo/oracle')
import gen_golden as G
import torch
sys.path.insert(0, '/root/reference/prism_experiment')
import importlib.util
spec = importlib.util.spec_from_file_location("bpm", "/root/reference/prism_experiment/benchmark_prism_mean_preconditioners.py")
bpm = importlib.util.module_from_spec(spec); sys.modules["bpm"] = bpm
try:
    spec.loader.exec_module(bpm)
except Exception as e:
    print("import error", e); raise
ref = G.ref
def run(name, x, y, kind, ls, var, sig2, eps, nu=2.5):
    d = x.shape[1]
    k = G.make_kernel(kind, d, ls, var, nu)
    m = ref.EFGPND(x, y, k, sigmasq=sig2, eps=eps, nufft_eps=1e-12, estimate_params=False, opts={"cg_tolerance": 1e-12, "mean_cg_warm_start": False})
    m._compute_common_parameters()
    ws, toep, xis = m._ws, m._toeplitz, m._xis
    h = xis.h_float; M = ws.numel(); mtot = round(M ** (1.0 / d)); sig = float(m.sigmasq.detach())
    nu_op = ref.NUFFT(x, torch.zeros(d, dtype=torch.float64), torch.tensor(h, dtype=torch.float64), 1e-12)
    Fy = nu_op.type1(y, out_shape=(mtot,) * d).reshape(-1)
    v = ref.compute_convolution_vector_vectorized_dD((mtot - 1) // 2, x, torch.tensor(h, dtype=torch.float64))
    A = ref.create_A_mean(ws, toep, sig, torch.complex128)
    center = tuple((s - 1) // 2 for s in v.shape)
    st = bpm.State(name=name, lengthscale=ls, variance=var, sigmasq=sig, mtot=mtot, M=M, ws=ws, v_kernel=v, rhs=ws * Fy, A_apply=A, diag_scale=float(v[center].real))
    pre = bpm.make_preconditioners(st)
    for tol in (1e-4, 1e-8):
        out = []
        for nm, Minv in pre.items():
            cg = ref.ConjugateGradients(A, ws * Fy, torch.zeros_like(Fy), tol=tol, early_stopping=True, M_inv_apply=Minv)
            cg.solve()
            out.append(f"{nm}={cg.iters_completed}")
        print(name, "mtot", mtot, "tol", tol, " ".join(out))
torch.set_num_threads(8)
x, y = G.usa_temp(); run("c3", x, y, "matern", 0.1, 1.0, 0.05, 1e-3, nu=2.5)
x, y = G.load_pair("gp_samples_100000_0.2_2_0.2.pt"); run("c2", x, y, "se", 0.2, 2.0, 0.2, 1e-4)
x, y = G.synth(100000, 2, 0); run("c4", x, y, "se", 0.05, 3.0, 0.2, 1e-4)
