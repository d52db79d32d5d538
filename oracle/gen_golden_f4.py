#!/usr/bin/env python3
"""SURVEY 8 row f4 goldens, produced by running the REFERENCE's own code (build container only):

  python oracle/gen_golden_f4.py      -> tests/golden/f4_preconditioners.npz

  * pre_<case>_<name>      iterations of the reference's mean solve at tol 1e-4 with each of the reference's preconditioners
                           (prism_experiment/benchmark_prism_mean_preconditioners.py:158-191: none, Jacobi, circulant scalar
                           mean / max, circulant sandwich median / geometric) on the golden systems c2, c3, c4, and the same
                           under a 1e-13 perturbation of the Toeplitz vector (`prep_*`: how far the reference's own count moves);
  * wt_*                   the weighted-Toeplitz system of the Polya-Gamma classifier
                           (polyagamma_classification/pg_classifier.py:377-420): v_w = type1(delta) on the (4m+1)^d box,
                           T_w = ToeplitzND(v_w), A u = u + ws T_w (ws u), rhs = ws F*z, solved by the reference's CG at
                           tol 1e-10 -- stored: delta, z, v_w, rhs, the solution, the iteration count.
"""
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402

ref = G.ref
sys.path.insert(0, "/root/reference/prism_experiment")
spec = importlib.util.spec_from_file_location("bpm", "/root/reference/prism_experiment/benchmark_prism_mean_preconditioners.py")
bpm = importlib.util.module_from_spec(spec)
sys.modules["bpm"] = bpm
spec.loader.exec_module(bpm)

NAMES = ["none", "jacobi_Nws2", "circ_scalar_meanws2", "circ_scalar_maxws2", "circ_sandwich_med", "circ_sandwich_geom"]


def state_of(name, x, y, kind, ls, var, sig2, eps, nu, amp, gen):
    d = x.shape[1]
    k = G.make_kernel(kind, d, ls, var, nu)
    m = ref.EFGPND(x, y, k, sigmasq=sig2, eps=eps, nufft_eps=1e-12, estimate_params=False,
                   opts={"cg_tolerance": 1e-12, "mean_cg_warm_start": False})
    m._compute_common_parameters()
    ws, xis = m._ws, m._xis
    h = xis.h_float
    M = ws.numel()
    mtot = round(M ** (1.0 / d))
    sig = float(m.sigmasq.detach())
    nu_op = ref.NUFFT(x, torch.zeros(d, dtype=torch.float64), torch.tensor(h, dtype=torch.float64), 1e-12)
    Fy = nu_op.type1(y, out_shape=(mtot,) * d).reshape(-1)
    v = ref.compute_convolution_vector_vectorized_dD((mtot - 1) // 2, x, torch.tensor(h, dtype=torch.float64))
    if amp:
        v = v * (1.0 + amp * torch.randn(v.shape, generator=gen, dtype=torch.float64))
    toep = ref.ToeplitzND(v, force_pow2=True)
    A = ref.create_A_mean(ws, toep, sig, torch.complex128)
    center = tuple((s - 1) // 2 for s in v.shape)
    st = bpm.State(name=name, lengthscale=ls, variance=var, sigmasq=sig, mtot=mtot, M=M, ws=ws, v_kernel=v, rhs=ws * Fy, A_apply=A,
                   diag_scale=float(v[center].real))
    return st, nu_op, h, mtot


def counts(st, tol=1e-4):
    pre = bpm.make_preconditioners(st)
    out = []
    for nm in NAMES:
        cg = ref.ConjugateGradients(st.A_apply, st.rhs, torch.zeros_like(st.rhs), tol=tol, early_stopping=True, M_inv_apply=pre[nm])
        cg.solve()
        out.append(int(cg.iters_completed))
    return out


def main():
    torch.set_num_threads(8)
    out = {"names": np.array(NAMES), "tol": 1e-4}
    gen = torch.Generator().manual_seed(5)
    cases = {"c2_se2d_n100000": (G.load_pair("gp_samples_100000_0.2_2_0.2.pt"), "se", 0.2, 2.0, 0.2, 1e-4, 2.5),
             "c3_matern52_usatemp": (G.usa_temp(), "matern", 0.1, 1.0, 0.05, 1e-3, 2.5),
             "c4_se2d_hard_n100000": (G.synth(100000, 2, 0), "se", 0.05, 3.0, 0.2, 1e-4, 2.5)}
    for name, ((x, y), kind, ls, var, sig2, eps, nu) in cases.items():
        st, _, _, mtot = state_of(name, x, y, kind, ls, var, sig2, eps, nu, 0.0, gen)
        c0 = counts(st)
        stp, _, _, _ = state_of(name, x, y, kind, ls, var, sig2, eps, nu, 1e-13, gen)
        c1 = counts(stp)
        out[f"pre_{name}"] = np.array(c0)
        out[f"prep_{name}"] = np.array(c1)
        out[f"mtot_{name}"] = mtot
        print(name, "mtot", mtot, dict(zip(NAMES, c0)), "| under a 1e-13 perturbation:", c1)
    # weighted Toeplitz system (pg_classifier.py:377-420) on the c3 points (N = 4766, 2-D)
    (x, y), kind, ls, var, sig2, eps, nu = cases["c3_matern52_usatemp"]
    st, nu_op, h, mtot = state_of("c3", x, y, kind, ls, var, sig2, eps, nu, 0.0, gen)
    g2 = torch.Generator().manual_seed(17)
    delta = 0.05 + 0.2 * torch.rand(x.shape[0], generator=g2, dtype=torch.float64)        # Polya-Gamma means lie in (0, 1/4]
    z = torch.randn(x.shape[0], generator=g2, dtype=torch.float64)
    ws = st.ws
    conv_shape = (2 * mtot - 1,) * 2
    v_w = nu_op.type1(delta.to(torch.complex128), out_shape=conv_shape)
    Tw = ref.ToeplitzND(v_w.to(torch.complex128), force_pow2=True)
    rhs = ws * nu_op.type1(z.to(torch.complex128), out_shape=(mtot, mtot)).reshape(-1)

    def A_feat(u):
        return u + ws * Tw(ws * u)
    cg = ref.ConjugateGradients(A_feat, rhs, torch.zeros_like(rhs), tol=1e-10, early_stopping=True)
    u = cg.solve()
    out.update(wt_delta=delta.numpy(), wt_z=z.numpy(), wt_vw=v_w.numpy(), wt_rhs=rhs.numpy(), wt_u=u.numpy(), wt_iters=int(cg.iters_completed),
               wt_mtot=mtot, wt_h=h)
    print("weighted Toeplitz system: M", ws.numel(), "iterations", cg.iters_completed)
    np.savez_compressed(os.path.join(G.GOLD, "f4_preconditioners.npz"), **out)


if __name__ == "__main__":
    main()
