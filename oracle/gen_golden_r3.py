#!/usr/bin/env python3
"""Round-3 additions to the golden vectors, produced by running the REFERENCE's own code (build container only):

  python oracle/gen_golden_r3.py [case ...]      -> tests/golden/<case>_r3.npz

The reference's DEFAULT is `mean_cg_warm_start=True` (efgpnd.py:639, 803-806; reuse at :136-139, 674): every real training
loop (test_timing_profiling.py:94-111) starts the mean solve of a step from the previous step's beta whenever the grid kept
its shape.  The round-1/2 goldens switch it off; these pin the warm-started paths:

  * wfit_*       fit -> change the hyper-parameters slightly (same grid) -> refit; the second solve starts from the first
                 beta (efgpnd.py:803-806).  Per tolerance (1e-4 = the reference default, 1e-12): both betas, both iteration
                 counts (read off ConjugateGradients.iters_completed of the solves the reference itself runs), the posterior
                 mean of the refit at fixed points.
  * wadam_*      a 4-step Adam trajectory of compute_gradients (efgpnd.py:573-708) with warm starts on, at cg_tol 1e-12 and at
                 the reference's default tolerance (0.1 eps, :652-653): gradients, hyper-parameters, mean-CG iteration counts
                 and `mean_cg_warm_start_used` of every step, the probes of every step.
  * sens_*       how far the REFERENCE's own results (warm refit, cold 3-step trajectory of the r2 goldens, warm 4-step
                 trajectories) move (a) when its Toeplitz vector is perturbed by 1e-13 relative -- less than the difference between
                 two correct NUFFTs -- and (b) when every NUFFT result carries a relative error of 1e-11 / 1e-10 (what another
                 correct NUFFT asked for 1e-12 may deliver).  The GPU tests derive their bounds from these numbers instead of
                 asserting them.
"""
import contextlib
import io
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402  (sets up the reference import path)
import numpy as np  # noqa: E402
import torch  # noqa: E402

ref = G.ref


class SolveRecorder:
    """Observes the ConjugateGradients solves the reference runs (iteration counts), nothing of the reference is edited."""

    def __init__(self):
        self.iters = []

    def __enter__(self):
        self._orig = ref.ConjugateGradients.solve
        rec = self

        def solve(cg_self, *a, **k):
            out = rec._orig(cg_self, *a, **k)
            rec.iters.append(int(cg_self.iters_completed))
            return out
        ref.ConjugateGradients.solve = solve
        return self

    def __exit__(self, *exc):
        ref.ConjugateGradients.solve = self._orig
        return False


class PerturbedToeplitzVector:
    """Multiplies the reference's Toeplitz vector by (1 + amp * N(0,1)) while active (amp = 0: identity)."""

    def __init__(self, amp, seed=5):
        self.amp = amp
        self.gen = torch.Generator().manual_seed(seed)

    def __enter__(self):
        self._orig = ref.compute_convolution_vector_vectorized_dD
        if self.amp:
            def pert(*a, **k):
                v = self._orig(*a, **k)
                return v * (1.0 + self.amp * torch.randn(v.shape, generator=self.gen, dtype=torch.float64))
            ref.compute_convolution_vector_vectorized_dD = pert
        return self

    def __exit__(self, *exc):
        ref.compute_convolution_vector_vectorized_dD = self._orig
        return False


class PerturbedNufft:
    """Every NUFFT result of the reference gets a relative l2 error `amp` (independent normal noise) while active: what the
    reference would see if its FINUFFT calls were replaced by ANOTHER correct NUFFT of that accuracy."""

    def __init__(self, amp, seed=7):
        self.amp = amp
        self.gen = torch.Generator().manual_seed(seed)

    def _noisy(self, fn):
        def wrapped(*a, **k):
            out = fn(*a, **k)
            n = torch.randn(out.shape, generator=self.gen, dtype=torch.float64)
            if out.is_complex():
                n = torch.complex(n, torch.randn(out.shape, generator=self.gen, dtype=torch.float64)) / 2 ** 0.5
            scale = torch.linalg.norm(out.reshape(-1)) / max(1, out.numel()) ** 0.5
            return out + self.amp * scale * n.to(out.dtype)
        return wrapped

    def __enter__(self):
        self._t1, self._t2 = ref.pff.finufft_type1, ref.pff.finufft_type2
        if self.amp:
            ref.pff.finufft_type1 = self._noisy(self._t1)
            ref.pff.finufft_type2 = self._noisy(self._t2)
        return self

    def __exit__(self, *exc):
        ref.pff.finufft_type1, ref.pff.finufft_type2 = self._t1, self._t2
        return False


def set_hypers(model, ls, var, sig2):
    """Writes (lengthscale, variance, sigma^2) into the reference model's raw log-parameters (kernel_params.py:39-55)."""
    with torch.no_grad():
        model._gp_params.raw.copy_(torch.log(torch.tensor([ls, var, sig2], dtype=model._gp_params.raw.dtype)))


def adam_run(x, y, kind, ls, var, sig2, eps, nu, *, steps, T, warm, cg_tol, seed, lr=0.05, amp=0.0, nufft_amp=0.0):
    d, N = x.shape[1], x.shape[0]
    k = G.make_kernel(kind, d, ls, var, nu)
    m = ref.EFGPND(x, y, k, sigmasq=sig2, eps=eps, nufft_eps=1e-12, estimate_params=False,
                   opts={"cg_tolerance": 1e-12 if cg_tol is not None else 1e-4, "mean_cg_warm_start": warm})
    torch.manual_seed(seed)
    opt = torch.optim.Adam(m._gp_params.parameters(), lr=lr)
    m.register_optimizer(opt)
    res = dict(traj=[], grads=[], Z=[], V=[], M=[], mean_iters=[], warm_used=[])
    with PerturbedToeplitzVector(amp), PerturbedNufft(nufft_amp):
        for _ in range(steps):
            state = torch.get_rng_state()
            opt.zero_grad()
            with contextlib.redirect_stdout(io.StringIO()):
                g = m.compute_gradients(trace_samples=T, nufft_eps=1e-12, cg_tol=cg_tol, apply_gradients=True)
            st = m.last_gradient_stats
            Mi = int(st["feature_count"])
            after = torch.get_rng_state()
            torch.set_rng_state(state)
            Z = torch.empty((T, N), dtype=torch.float64).bernoulli_(0.5).mul_(2).sub_(1)
            V = torch.empty((T, Mi), dtype=torch.float64).bernoulli_(0.5).mul_(2).sub_(1)
            torch.set_rng_state(after)
            res["Z"].append(np.packbits((Z.numpy() > 0).astype(np.uint8), axis=1))
            res["V"].append(V.numpy().astype(np.int8))
            res["M"].append(Mi)
            res["mean_iters"].append(int(st["mean_cg_iters"]))
            res["warm_used"].append(bool(st["mean_cg_warm_start_used"]))
            res["grads"].append(g.detach().numpy().copy())
            opt.step()
            res["traj"].append(np.array([float(m.kernel.get_hyper(n)) for n in m.kernel.hypers] + [float(m.sigmasq.detach())]))
    return res


def rel_move(a, b):
    a, b = np.stack(a), np.stack(b)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(a), 1e-300)))


def grad_move(a, b):
    """Move of the gradients relative to the largest gradient entry of the step (the scale the tests use)."""
    a, b = np.stack(a), np.stack(b)
    return float(np.max(np.abs(a - b) / np.abs(a).max(axis=1, keepdims=True)))


def extras(name, x, y, kind, ls, var, sig2, eps, *, nu=2.5, seed=1234, T=2, bump=(1.03, 0.97, 1.05)):
    d, N = x.shape[1], x.shape[0]
    out = dict(kind=kind, d=d, N=N, nu=nu, eps=eps, seed=seed, T=T, lr=0.05)
    xn = G.probe_points(x, 48)
    out["x_new"] = xn.numpy()
    # --- warm-started refit -------------------------------------------------------------------------------------
    ls2, var2, sg2 = ls * bump[0], var * bump[1], sig2 * bump[2]
    out["wfit_hypers0"] = np.array([ls, var, sig2])
    out["wfit_hypers1"] = np.array([ls2, var2, sg2])
    for tol, tag in ((1e-4, "1e4"), (1e-12, "1e12")):
        k = G.make_kernel(kind, d, ls, var, nu)
        m = ref.EFGPND(x, y, k, sigmasq=sig2, eps=eps, nufft_eps=1e-12, estimate_params=False, opts={"cg_tolerance": tol})
        with SolveRecorder() as rec:
            m._compute_common_parameters()
            beta0 = m._beta.clone()
            mt0 = int(round(m._ws.numel() ** (1.0 / d)))
            set_hypers(m, ls2, var2, sg2)
            m._compute_common_parameters(force_recompute=True)
        beta1 = m._beta.clone()
        mt1 = int(round(m._ws.numel() ** (1.0 / d)))
        assert mt0 == mt1, (name, mt0, mt1, "the bump must keep the grid so that the warm start is used")
        assert len(rec.iters) == 2, rec.iters
        h = m._xis.h_float
        nu_new = ref.NUFFT(xn, torch.zeros(d, dtype=torch.float64), torch.tensor(h, dtype=torch.float64), 1e-12)
        mean1 = nu_new.type2(m._ws * beta1, out_shape=(mt1,) * d).real
        # the same refit from zeros, for the iteration count a cold start needs
        k = G.make_kernel(kind, d, ls2, var2, nu)
        mc = ref.EFGPND(x, y, k, sigmasq=sg2, eps=eps, nufft_eps=1e-12, estimate_params=False,
                        opts={"cg_tolerance": tol, "mean_cg_warm_start": False})
        with SolveRecorder() as rc:
            mc._compute_common_parameters()
        # the reference's own refit under a 1e-13 perturbation of its Toeplitz vectors (both fits)
        k = G.make_kernel(kind, d, ls, var, nu)
        mp = ref.EFGPND(x, y, k, sigmasq=sig2, eps=eps, nufft_eps=1e-12, estimate_params=False, opts={"cg_tolerance": tol})
        with SolveRecorder() as rp, PerturbedToeplitzVector(1e-13):
            mp._compute_common_parameters()
            set_hypers(mp, ls2, var2, sg2)
            mp._compute_common_parameters(force_recompute=True)
        out[f"sens_wfit_iters_{tag}"] = np.array(rp.iters)
        out[f"sens_wfit_beta1_{tag}"] = float(torch.linalg.norm(mp._beta - beta1) / torch.linalg.norm(beta1))
        meanp = nu_new.type2(mp._ws * mp._beta, out_shape=(mt1,) * d).real
        out[f"sens_wfit_mean1_{tag}"] = float(torch.linalg.norm(meanp - mean1) / torch.linalg.norm(mean1))
        print(f"{name}: warm refit tol={tol:g} under a 1e-13 perturbation: iters {rp.iters}, beta1 moves {out[f'sens_wfit_beta1_{tag}']:.2e}, the mean {out[f'sens_wfit_mean1_{tag}']:.2e}")
        # ... and when every NUFFT result carries a relative error of 1e-11 (F*y and the Toeplitz vector of both fits): what a
        # correct NUFFT asked for 1e-12 may deliver, amplified by the conditioning of D T D + sigma^2
        k = G.make_kernel(kind, d, ls, var, nu)
        mq = ref.EFGPND(x, y, k, sigmasq=sig2, eps=eps, nufft_eps=1e-12, estimate_params=False, opts={"cg_tolerance": tol})
        with PerturbedNufft(1e-11):
            mq._compute_common_parameters()
            set_hypers(mq, ls2, var2, sg2)
            mq._compute_common_parameters(force_recompute=True)
        meanq = nu_new.type2(mq._ws * mq._beta, out_shape=(mt1,) * d).real
        out[f"sens_wfit_nufft_beta1_{tag}"] = float(torch.linalg.norm(mq._beta - beta1) / torch.linalg.norm(beta1))
        out[f"sens_wfit_nufft_mean1_{tag}"] = float(torch.linalg.norm(meanq - mean1) / torch.linalg.norm(mean1))
        print(f"{name}: warm refit tol={tol:g} with NUFFT results perturbed by 1e-11: beta1 moves {out[f'sens_wfit_nufft_beta1_{tag}']:.2e}, "
              f"the mean {out[f'sens_wfit_nufft_mean1_{tag}']:.2e}")
        # true relative residual |rhs - A beta1| / |rhs| the reference's refit attains (at cg_tol 1e-12 its solves end at the
        # iteration cap 2 M, cg.py:59-65: this is the accuracy level of the system, not 1e-12)
        A1 = ref.create_A_mean(m._ws, m._toeplitz, float(m.sigmasq.detach()), torch.complex128)
        nu_tr = ref.NUFFT(x, torch.zeros(d, dtype=torch.float64), torch.tensor(h, dtype=torch.float64), 1e-12)
        rhs1 = m._ws * nu_tr.type1(y, out_shape=(mt1,) * d).reshape(-1)
        out[f"wfit_true_resid1_{tag}"] = float(torch.linalg.norm(rhs1 - A1(beta1)) / torch.linalg.norm(rhs1))
        print(f"{name}: warm refit tol={tol:g}: true residual of the reference's beta1 {out[f'wfit_true_resid1_{tag}']:.2e}")
        out[f"wfit_beta0_{tag}"] = beta0.numpy()
        out[f"wfit_beta1_{tag}"] = beta1.numpy()
        out[f"wfit_iters_{tag}"] = np.array(rec.iters)
        out[f"wfit_cold_iters_{tag}"] = rc.iters[0]
        out[f"wfit_mean1_{tag}"] = mean1.numpy()
        out["wfit_mtot"] = mt1
        print(f"{name}: warm refit tol={tol:g}: iters first={rec.iters[0]} warm={rec.iters[1]} cold={rc.iters[0]} mtot={mt1}")
    # --- warm-started Adam trajectories ----------------------------------------------------------------------------
    for cg_tol, tag in ((1e-12, "tight"), (None, "default")):
        r = adam_run(x, y, kind, ls, var, sig2, eps, nu, steps=4, T=T, warm=True, cg_tol=cg_tol, seed=seed)
        out[f"wadam_{tag}_traj"] = np.stack(r["traj"])
        out[f"wadam_{tag}_grads"] = np.stack(r["grads"])
        out[f"wadam_{tag}_M"] = np.array(r["M"])
        out[f"wadam_{tag}_mean_iters"] = np.array(r["mean_iters"])
        out[f"wadam_{tag}_warm_used"] = np.array(r["warm_used"])
        for i in range(4):
            out[f"wadam_{tag}_Z{i}"] = r["Z"][i]
            out[f"wadam_{tag}_V{i}"] = r["V"][i]
        rp = adam_run(x, y, kind, ls, var, sig2, eps, nu, steps=4, T=T, warm=True, cg_tol=cg_tol, seed=seed, amp=1e-13)
        out[f"sens_wadam_{tag}_traj"] = rel_move(r["traj"], rp["traj"])
        out[f"sens_wadam_{tag}_grad"] = grad_move(r["grads"], rp["grads"])
        out[f"sens_wadam_{tag}_iters"] = np.array(rp["mean_iters"])
        print(f"{name}: warm Adam ({tag}): iters {r['mean_iters']} warm_used {r['warm_used']} M {r['M']}; under a 1e-13 perturbation "
              f"iters {rp['mean_iters']}, traj moves {out[f'sens_wadam_{tag}_traj']:.2e}, grads {out[f'sens_wadam_{tag}_grad']:.2e}")
    # --- sensitivity of the cold 3-step trajectory of the r2 goldens (same settings as gen_golden_r2.extras) ----------------
    r0 = adam_run(x, y, kind, ls, var, sig2, eps, nu, steps=3, T=T, warm=False, cg_tol=1e-12, seed=seed)
    r1 = adam_run(x, y, kind, ls, var, sig2, eps, nu, steps=3, T=T, warm=False, cg_tol=1e-12, seed=seed, amp=1e-13)
    out["sens_cold_traj"] = rel_move(r0["traj"], r1["traj"])
    out["sens_cold_grad"] = grad_move(r0["grads"], r1["grads"])
    # the same trajectory when every NUFFT result carries a relative error of 1e-11 / 1e-10 (a NUFFT asked for 1e-12 delivers a
    # few times its tolerance): the amplification of transform errors by the ill-conditioned solves, from the reference's side
    for amp, tag in ((1e-11, "1e11"), (1e-10, "1e10")):
        rn = adam_run(x, y, kind, ls, var, sig2, eps, nu, steps=3, T=T, warm=False, cg_tol=1e-12, seed=seed, nufft_amp=amp)
        out[f"sens_cold_nufft{tag}_traj"] = rel_move(r0["traj"], rn["traj"])
        out[f"sens_cold_nufft{tag}_grad"] = grad_move(r0["grads"], rn["grads"])
        print(f"{name}: cold 3-step Adam with NUFFT results perturbed by {amp:g}: trajectory moves {out[f'sens_cold_nufft{tag}_traj']:.2e}, "
              f"gradients {out[f'sens_cold_nufft{tag}_grad']:.2e}")
    out["cold_traj"] = np.stack(r0["traj"])
    out["cold_mean_iters"] = np.array(r0["mean_iters"])
    print(f"{name}: cold 3-step Adam (r2 golden settings): mean iters {r0['mean_iters']}; 1e-13 perturbation moves the trajectory by "
          f"{out['sens_cold_traj']:.2e}, the gradients by {out['sens_cold_grad']:.2e}")
    np.savez_compressed(os.path.join(G.GOLD, name + "_r3.npz"), **out)


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


if __name__ == "__main__":
    torch.set_num_threads(8)
    for n in (sys.argv[1:] or list(CASES)):
        CASES[n]()
