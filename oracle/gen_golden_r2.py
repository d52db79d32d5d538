#!/usr/bin/env python3
"""Round-2 additions to the golden vectors, again produced by running the REFERENCE's own code (build container only):

  python oracle/gen_golden_r2.py [case ...]      -> tests/golden/<case>_r2.npz

  * resid_1e4        relative residual |r_i| / |b| of every iteration of the reference's mean solve at its default
                     tolerance 1e-4 (cg.py:132) -- recorded by observing the torch.linalg.norm calls the reference's loop
                     makes, nothing of the reference is edited or copied;
  * slq_*            logdet_slq (efgpnd.py:1686-1759) with the probes it drew (re-drawn here in the same order to store
                     them) and both log-marginal formulas (efgpnd.py:1063-1066 and :288-289);
  * adam_*           a 3-step Adam trajectory of optimize_hyperparameters (efgpnd.py:1068-1226) with the probes of every
                     step recorded (Z, V per step) and the hyper-parameters / gradients after every step.
The existing <case>.npz files are not touched.
"""
import contextlib
import io
import math
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402  (sets up the reference import path)
import numpy as np  # noqa: E402
import torch  # noqa: E402

ref = G.ref


class NormRecorder:
    """Wraps torch.linalg.norm while the reference's CG runs: call 0 is |b| (cg.py:107), call i >= 1 is |r_i| (cg.py:132)."""

    def __init__(self):
        self.vals = []
        self._orig = None

    def __enter__(self):
        self._orig = torch.linalg.norm

        def rec(*a, **k):
            v = self._orig(*a, **k)
            self.vals.append(float(v.real if torch.is_complex(v) else v))
            return v
        torch.linalg.norm = rec
        return self

    def __exit__(self, *exc):
        torch.linalg.norm = self._orig
        return False


def extras(name, x, y, kind, ls, var, sig2, eps, *, nu=2.5, seed=1234, slq_probes=6, slq_steps=25, adam_steps=3, T=2,
           do_adam=True):
    d = x.shape[1]
    N = x.shape[0]
    out = dict(kind=kind, d=d, N=N, nu=nu, eps=eps, seed=seed)
    k = G.make_kernel(kind, d, ls, var, nu)
    m = ref.EFGPND(x, y, k, sigmasq=sig2, eps=eps, nufft_eps=1e-12, estimate_params=False,
                   opts={"cg_tolerance": 1e-12, "mean_cg_warm_start": False})
    m._compute_common_parameters()
    beta, ws, toep, xis = m._beta, m._ws, m._toeplitz, m._xis
    h = xis.h_float
    M = ws.numel()
    mtot = round(M ** (1.0 / d))
    sig = float(m.sigmasq.detach())
    out.update(h=h, mtot=mtot, sigmasq=sig, lengthscale=k.get_hyper("lengthscale"), variance=k.get_hyper("variance"))
    # --- residual history of the default-tolerance mean solve -------------------------------------------------
    nu_op = ref.NUFFT(x, torch.zeros(d, dtype=torch.float64), torch.tensor(h, dtype=torch.float64), 1e-12)
    Fy = nu_op.type1(y, out_shape=(mtot,) * d).reshape(-1)
    v = ref.compute_convolution_vector_vectorized_dD((mtot - 1) // 2, x, torch.tensor(h, dtype=torch.float64))
    A = ref.create_A_mean(ws, toep, sig, torch.complex128)
    center = tuple((s - 1) // 2 for s in v.shape)
    Minv = ref.create_jacobi_precond(ws, sig, diag_scale=v[center].real)
    cg = ref.ConjugateGradients(A, ws * Fy, torch.zeros_like(Fy), tol=1e-4, early_stopping=True, M_inv_apply=Minv)
    with NormRecorder() as rec:
        cg.solve()
    bn = rec.vals[0]
    hist = np.array(rec.vals[1:]) / (bn + 1e-16)
    assert len(hist) == cg.iters_completed, (len(hist), cg.iters_completed)
    out["resid_1e4"] = hist
    out["iters_1e4"] = cg.iters_completed
    # --- SLQ log-determinant with recorded probes, both log-marginal formulas ---------------------------------
    torch.manual_seed(seed)
    zs = torch.stack([torch.empty(M, dtype=torch.float64).bernoulli_(0.5).mul_(2).sub_(1) for _ in range(slq_probes)])
    torch.manual_seed(seed)
    ld = ref.logdet_slq(ws, sig, toep, probes=slq_probes, steps=slq_steps, dtype=torch.float64, device="cpu", n=N)
    out["slq_probes"] = zs.numpy().astype(np.int8)
    out["slq_steps"] = slq_steps
    out["slq_logdet"] = float(ld)
    out["log_marginal_predict"] = float(-0.5 * (ld + float((ws.abs() * (beta.abs() ** 2)).sum().real)))   # efgpnd.py:1063-1066
    # gradient-path formula (efgpnd.py:288-289): -y.alpha/2 - logdet/2 - N log(2 pi)/2 with alpha = (y - F(ws beta))/sigma^2
    z = nu_op.type2(ws * beta, out_shape=(mtot,) * d)
    alpha = (y - z) / sig
    out["y_alpha"] = float(torch.vdot(y.to(torch.complex128), alpha).real)
    out["log_marginal_gradient"] = float(-0.5 * out["y_alpha"] - 0.5 * ld - 0.5 * N * math.log(2 * math.pi))
    # --- 3-step Adam trajectory ----------------------------------------------------------------------------
    if do_adam:
        k2 = G.make_kernel(kind, d, ls, var, nu)
        m2 = ref.EFGPND(x, y, k2, sigmasq=sig2, eps=eps, nufft_eps=1e-12, estimate_params=False,
                        opts={"cg_tolerance": 1e-12, "mean_cg_warm_start": False})
        # the probes optimize_hyperparameters will draw, step by step, from this seed (Z then V per step: efgpnd.py:179-182, 199-202);
        # the grid (hence M) changes with the hyper-parameters, so they are recorded by replaying the generator per step
        torch.manual_seed(seed)
        opt = torch.optim.Adam(m2._gp_params.parameters(), lr=0.05)
        m2.register_optimizer(opt)
        traj, grads, Zs, Vs, Ms = [], [], [], [], []
        for it in range(adam_steps):
            state = torch.get_rng_state()
            opt.zero_grad()
            stats = {}
            with contextlib.redirect_stdout(io.StringIO()):
                g = m2.compute_gradients(trace_samples=T, nufft_eps=1e-12, cg_tol=1e-12, apply_gradients=True)
            Mi = int(m2.last_gradient_stats["feature_count"]) if hasattr(m2, "last_gradient_stats") else None
            # replay the draws of this step
            after = torch.get_rng_state()
            torch.set_rng_state(state)
            Z = torch.empty((T, N), dtype=torch.float64).bernoulli_(0.5).mul_(2).sub_(1)
            V = torch.empty((T, Mi), dtype=torch.float64).bernoulli_(0.5).mul_(2).sub_(1)
            torch.set_rng_state(after)
            Zs.append(np.packbits((Z.numpy() > 0).astype(np.uint8), axis=1))
            Vs.append(V.numpy().astype(np.int8))
            Ms.append(Mi)
            grads.append(g.detach().numpy().copy())
            opt.step()
            traj.append(np.array([float(m2.kernel.get_hyper(n)) for n in m2.kernel.hypers] + [float(m2.sigmasq.detach())]))
        out["adam_lr"] = 0.05
        out["adam_T"] = T
        out["adam_hypers"] = np.array(list(m2.kernel.hypers) + ["sigmasq"])
        out["adam_traj"] = np.stack(traj)
        out["adam_grads"] = np.stack(grads)
        out["adam_M"] = np.array(Ms)
        for i in range(adam_steps):
            out[f"adam_Z{i}"] = Zs[i]
            out[f"adam_V{i}"] = Vs[i]
    np.savez_compressed(os.path.join(G.GOLD, name + "_r2.npz"), **out)
    print(f"{name}_r2: iters(1e-4)={out['iters_1e4']} resid_last={hist[-1]:.3e} slq={out['slq_logdet']:.6f} "
          f"lm_pred={out['log_marginal_predict']:.6f} lm_grad={out['log_marginal_gradient']:.6f}"
          + (f" adam_traj_last={out['adam_traj'][-1]}" if do_adam else ""))


CASES = {}


def case(fn):
    CASES[fn.__name__] = fn
    return fn


@case
def c1_se1d_n5000():
    x, y = G.load_pair("gp_samples_5000_0.1_2_0.1.pt")
    extras("c1_se1d_n5000", x, y, "se", 0.1, 2.0, 0.1, 1e-4)


@case
def c2_se2d_n100000():
    x, y = G.load_pair("gp_samples_100000_0.2_2_0.2.pt")
    extras("c2_se2d_n100000", x, y, "se", 0.2, 2.0, 0.2, 1e-4)


@case
def c3_matern52_usatemp():
    x, y = G.usa_temp()
    extras("c3_matern52_usatemp", x, y, "matern", 0.1, 1.0, 0.05, 1e-3, nu=2.5)


@case
def c4_se2d_hard_n100000():
    x, y = G.synth(100000, 2, 0)
    extras("c4_se2d_hard_n100000", x, y, "se", 0.05, 3.0, 0.2, 1e-4, do_adam=False)


@case
def c5_matern32_3d_n20000():
    x, y = G.synth(20000, 3, 1)
    extras("c5_matern32_3d_n20000", x, y, "matern", 0.3, 1.5, 0.2, 1e-2, nu=1.5, do_adam=False)


if __name__ == "__main__":
    torch.set_num_threads(8)
    for n in (sys.argv[1:] or list(CASES)):
        CASES[n]()
