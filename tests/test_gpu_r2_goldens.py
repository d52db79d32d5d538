"""Round-2 golden vectors (oracle/gen_golden_r2.py, the reference's own code): iterate-level CG parity at the reference's
default tolerance, SLQ log-determinant and both log-marginal formulas with the reference's recorded probes, and a 3-step
Adam trajectory with the probes of every step."""
import numpy as np
import pytest
import torch

from _golden import GOLDEN, load_case, rel
from test_gpu_efgp_golden import make_model

pytestmark = pytest.mark.gpu

R2 = ["c1_se1d_n5000", "c2_se2d_n100000", "c3_matern52_usatemp", "c4_se2d_hard_n100000", "c5_matern32_3d_n20000"]


# The REFERENCE's own sensitivity to a rounding-level (1e-13 relative) perturbation of the Toeplitz vector, measured by
# oracle/sensitivity_r2.py with the reference's code: (max relative move of the residual curve over all iterations, relative
# move of logdet_slq with the same probes).  A finite-precision Krylov recurrence amplifies rounding differences, on the
# ill-conditioned Matern systems up to O(1) within the first half of the solve; the tolerances below are 3 x these numbers.
# Cases not listed (short or well-conditioned solves) keep tight absolute bounds.
REF_SENSITIVITY = {"c2_se2d_n100000": (2.82e-1, 4.86e-14), "c3_matern52_usatemp": (1.71, 7.33e-5),
                   "c5_matern32_3d_n20000": (2.91e-1, 1.09e-3)}


def load_r2(name):
    return dict(np.load(f"{GOLDEN}/{name}_r2.npz", allow_pickle=False))


@pytest.mark.parametrize("name", R2)
def test_residual_history_matches_reference(name):
    """cg.py:132 tests |r_i| / (|b| + eps) < tol.  The device solve must follow the reference's residual CURVE, not only
    end near the same iteration.  Inputs of the recurrence (F*y, v) agree with the reference's to the NUFFT tolerance
    requested here (1e-12), so the curves start ~1e-12 apart.  Asserted: <= 1e-8 relative over the first 20 iterations
    (measured 1e-13..4e-11: same operator, same recurrence); afterwards the deviation may grow as far as the REFERENCE's
    own curve moves under a 1e-13 perturbation (REF_SENSITIVITY, x3), 1e-5 for the cases without such growth; every
    recorded value before the last is above the tolerance and the last below it (the stopping rule itself); the stopping
    index equals the reference's unless the reference's own curve passes within 1 % of the threshold between the two, or
    the curves have already decorrelated (then within 15 %, the spread of the reference's own index)."""
    from efgp_hip import cg_residual_history
    g, x, y = load_case(name)
    r2 = load_r2(name)
    ref_hist = r2["resid_1e4"]
    m = make_model(name, g, x.cuda(), y.cuda(), 1e-4, nufft_eps=1e-12)
    with cg_residual_history(torch.device("cuda", 0), 4096) as rec:
        m.fit()
        its = int(m.last_fit_stats["mean_cg_iters"])
    hist = rec.values().numpy()[:its]
    n_ref = len(ref_hist)
    assert int(r2["iters_1e4"]) == n_ref
    k = min(its, n_ref)
    dev = np.abs(hist[:k] - ref_hist[:k]) / ref_hist[:k]
    print(f"\n{name}: iters hip={its} ref={n_ref}; residual-curve deviation first20={dev[:min(20, k)].max():.2e} "
          f"half={dev[:max(1, k // 2)].max():.2e} all={dev.max():.2e}")
    assert dev[:min(20, k)].max() < 1e-8, (name, dev[:min(20, k)].max())
    bound = 3.0 * REF_SENSITIVITY[name][0] if name in REF_SENSITIVITY else 1e-5
    assert dev.max() < bound, (name, dev.max(), int(dev.argmax()), bound)
    assert (hist[:-1] >= 1e-4).all() and hist[-1] < 1e-4
    if its != n_ref:
        if dev.max() > 1e-2:
            # the two curves have decorrelated before the stop (see REF_SENSITIVITY: the reference itself stops at 182 instead
            # of 175 on c3, at 152 instead of 150 on c2, under a 1e-13 perturbation): the index may move within 15 %
            assert abs(its - n_ref) <= 0.15 * n_ref, (name, its, n_ref)
        else:
            lo, hi = sorted((its, n_ref))
            near = np.abs(ref_hist[lo - 1:hi] - 1e-4) / 1e-4
            assert near.min() < 1e-2, (name, its, n_ref, near.min())


@pytest.mark.parametrize("name", R2)
def test_slq_logdet_and_log_marginals(name):
    """logdet_slq (efgpnd.py:1686-1759) with the reference's probes; predict-path (:1063-1066) and gradient-path
    (:288-289) log marginals.  All Lanczos steps run inside one launch (efgp_lanczos) where the grid allows."""
    from efgpnd import logdet_slq, efgpnd_gradient_batched
    g, x, y = load_case(name)
    r2 = load_r2(name)
    probes = torch.from_numpy(r2["slq_probes"].astype(np.float64))
    steps = int(r2["slq_steps"])
    m = make_model(name, g, x.cuda(), y.cuda(), 1e-12, log_marginal_probe_vectors=probes, log_marginal_steps=steps)
    m.fit()
    st = m._fit_state
    N = x.shape[0]
    ld = logdet_slq(st["ws"], st["sig"], m._toeplitz, probes=probes.shape[0], steps=steps, n=N, probe_vectors=probes)
    ref_ld = float(r2["slq_logdet"])
    # Lanczos loses orthogonality once extreme Ritz values converge: on the Matern systems the reference's own value moves by
    # 7e-5 / 1e-3 relative under a 1e-13 perturbation (REF_SENSITIVITY); elsewhere the values agree to 1e-7
    # (round 3: 10 x the reference's own move, the rule of the round-3 fixtures -- the in-house FFT, whose rounding error against an
    # 80-bit DFT is below hipFFT's (profiles/r3_fft_inhouse.txt), moved the gradient-path value on c3 by 2.5 x that move)
    tol_ld = max(1e-7, 10.0 * REF_SENSITIVITY.get(name, (0.0, 0.0))[1])
    print(f"\n{name}: logdet hip={ld:.6f} ref={ref_ld:.6f} rel diff {abs(ld - ref_ld) / abs(ref_ld):.2e} (bound {tol_ld:.1e})")
    assert abs(ld - ref_ld) < tol_ld * abs(ref_ld), (ld, ref_ld)
    # predict path
    xn = torch.from_numpy(g["x_new"])
    _, _, lm = m.predict(xn, return_variance=False, compute_log_marginal=True)
    ref_lm = float(r2["log_marginal_predict"])
    assert abs(float(lm) - ref_lm) < 1e-6 * abs(ref_lm) + 0.5 * tol_ld * abs(ref_ld), (float(lm), ref_lm)
    # gradient path (the literal operation order: y.alpha from a type-2 pass; the adjoint form agrees to the NUFFT tolerance)
    for mode, tol in (("reference", 1e-6), ("adjoint", 1e-5)):
        T = g["V"].shape[0]
        _, lmg = efgpnd_gradient_batched(x.cuda(), y.cuda(), sigmasq=st["sig"], kernel=m.kernel, eps=float(g["eps"]), trace_samples=T,
                                         nufft_eps=1e-9, cg_tol=1e-12, compute_log_marginal=True, log_marginal_probes=probes.shape[0],
                                         log_marginal_steps=steps, log_marginal_probe_vectors=probes, probes_Z=g["Z"],
                                         probes_V=torch.from_numpy(g["V"].astype(np.float64)), trace_mode=mode)
        ref_lmg = float(r2["log_marginal_gradient"])
        assert abs(float(lmg) - ref_lmg) < tol * abs(ref_lmg) + 0.5 * tol_ld * abs(ref_ld), (mode, float(lmg), ref_lmg)


@pytest.mark.parametrize("name", ["c1_se1d_n5000", "c2_se2d_n100000", "c3_matern52_usatemp"])
def test_adam_trajectory_matches_reference(name):
    """Three Adam steps of the reference's training loop (efgpnd.py:1068-1226: compute_gradients -> opt.step, lr 0.05)
    with the probes the reference drew at every step: gradients and hyper-parameters after each step.

    Bounds (round 3): 10 x what the REFERENCE's own trajectory moves by -- under a 1e-13 perturbation of its Toeplitz vector and
    with every NUFFT result carrying a relative error of 1e-11 / 1e-10 (oracle/gen_golden_r3.py stores those moves as `sens_cold_*`
    in <case>_r3.npz) -- with floors of 1e-9 (hyper-parameters) and 1e-8 of the gradient scale.  Round 2 had to allow 1e-5 here and
    called it rounding; it was the Hermitian 64 x 64 solver handing the anti-Hermitian rounding noise of its k0 = 0 row to the
    operator (cg_persistent.hip, phase B): the true residual of these cg_tol = 1e-12 solves stalled at 2e-6 where the reference
    reaches 7e-10.  With the projection in place the trajectory sits at the reference's own sensitivity (c2: 2.8e-9 against 4.7e-9)."""
    g, x, y = load_case(name)
    r2 = load_r2(name)
    r3 = dict(np.load(f"{GOLDEN}/{name}_r3.npz", allow_pickle=False))
    s_traj = max(float(r3["sens_cold_traj"]), float(r3["sens_cold_nufft1e11_traj"]), float(r3["sens_cold_nufft1e10_traj"]))
    s_grad = max(float(r3["sens_cold_grad"]), float(r3["sens_cold_nufft1e11_grad"]), float(r3["sens_cold_nufft1e10_grad"]))
    traj, grads, Ms = r2["adam_traj"], r2["adam_grads"], r2["adam_M"]
    assert np.abs(r3["cold_traj"] - traj).max() < 1e-12 * np.abs(traj).max()      # the sensitivity runs restate this trajectory
    T = int(r2["adam_T"])
    N = x.shape[0]
    m = make_model(name, g, x.cuda(), y.cuda(), 1e-12, nufft_eps=1e-12)
    opt = torch.optim.Adam(m._gp_params.parameters(), lr=float(r2["adam_lr"]))
    m.register_optimizer(opt)
    worst = worst_g = 0.0
    for i in range(traj.shape[0]):
        Z = torch.from_numpy(np.unpackbits(r2[f"adam_Z{i}"], axis=1)[:, :N].astype(np.float64) * 2 - 1)
        V = torch.from_numpy(r2[f"adam_V{i}"].astype(np.float64))
        opt.zero_grad()
        grad = m.compute_gradients(trace_samples=T, nufft_eps=1e-12, cg_tol=1e-12, probes_Z=Z, probes_V=V)
        assert int(m.last_gradient_stats["feature_count"]) == int(Ms[i])
        scale = float(np.abs(grads[i]).max())
        worst_g = max(worst_g, float((grad.detach().cpu() - torch.from_numpy(grads[i])).abs().max()) / scale)
        opt.step()
        now = np.array([float(m.kernel.get_hyper(n)) for n in m.kernel.hypers] + [float(m.sigmasq.detach())])
        worst = max(worst, float(np.abs(now - traj[i]).max() / np.abs(traj[i]).max()))
    print(f"\n{name}: 3 cold Adam steps: gradients deviate by at most {worst_g:.2e} of their scale (the reference's own move: "
          f"{s_grad:.2e}), hyper-parameters by {worst:.2e} (the reference's own: {s_traj:.2e})")
    assert worst_g < max(10.0 * s_grad, 1e-8), (name, worst_g, s_grad)
    assert worst < max(10.0 * s_traj, 1e-9), (name, worst, s_traj)
