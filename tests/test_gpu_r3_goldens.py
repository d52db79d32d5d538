"""Round-3 golden vectors (oracle/gen_golden_r3.py, the reference's own code): the WARM-STARTED paths.

The reference's default is `mean_cg_warm_start=True` (efgpnd.py:639, 803-806; reuse at :136-139, 674); the round-1/2 goldens
switch it off.  Here: (i) fit -> changed hyper-parameters on the same grid -> refit from the previous beta, at the reference's
default tolerance and at 1e-12; (ii) a 4-step Adam trajectory of compute_gradients with warm starts on, at cg_tol 1e-12 and at
the reference's default tolerance, with the probes of every step.  Every bound that is not rounding-tight is derived from the
REFERENCE's own sensitivity stored in the fixture (`sens_*`: how far its outputs move when its Toeplitz vector is perturbed by
1e-13 relative) -- never from what this implementation happened to measure."""
import numpy as np
import pytest
import torch

from _golden import GOLDEN, load_case, rel
from test_gpu_efgp_golden import make_model

pytestmark = pytest.mark.gpu

R3 = ["c1_se1d_n5000", "c2_se2d_n100000", "c3_matern52_usatemp"]


def load_r3(name):
    return dict(np.load(f"{GOLDEN}/{name}_r3.npz", allow_pickle=False))


def set_hypers(model, ls, var, sig2):
    with torch.no_grad():
        model._gp_params.raw.copy_(torch.log(torch.tensor([ls, var, sig2], dtype=model._gp_params.raw.dtype)))


@pytest.mark.parametrize("tag,tol", [("1e4", 1e-4), ("1e12", 1e-12)])
@pytest.mark.parametrize("name", R3)
def test_warm_started_refit_matches_reference(name, tag, tol):
    """efgpnd.py:803-806: the second fit of a model starts its mean solve from the first beta when the grid kept its shape.
    Iteration counts of both solves, both betas and the posterior mean of the refit against the reference's."""
    g, x, y = load_case(name)
    r3 = load_r3(name)
    m = make_model(name, g, x.cuda(), y.cuda(), tol, nufft_eps=1e-12, mean_cg_warm_start=True)
    m.fit()
    it0 = int(m.last_fit_stats["mean_cg_iters"])
    beta0 = m._beta.clone()
    set_hypers(m, *r3["wfit_hypers1"])
    m._compute_common_parameters(force_recompute=True)
    it1 = int(m.last_fit_stats["mean_cg_iters"])
    assert int(m.last_fit_stats["mtot"]) == int(r3["wfit_mtot"])
    ref_it = r3[f"wfit_iters_{tag}"]
    xn = torch.from_numpy(r3["x_new"]).cuda()
    mean1, _ = m.predict(xn, return_variance=False)
    e0, e1, em = rel(beta0, r3[f"wfit_beta0_{tag}"]), rel(m._beta, r3[f"wfit_beta1_{tag}"]), rel(mean1, r3[f"wfit_mean1_{tag}"])
    # the reference's own move: under a 1e-13 perturbation of its Toeplitz vector, and with NUFFT results accurate to 1e-11
    s_it = r3[f"sens_wfit_iters_{tag}"]
    s_b = max(float(r3[f"sens_wfit_beta1_{tag}"]), float(r3[f"sens_wfit_nufft_beta1_{tag}"]))
    s_m = max(float(r3[f"sens_wfit_mean1_{tag}"]), float(r3[f"sens_wfit_nufft_mean1_{tag}"]))
    # the TRUE residual of the returned beta through the model's own operator, against what the reference's beta attains through
    # the reference's (a recurrence that drifts away from the true residual ends here with a larger one: round 3 found the
    # Hermitian 64 x 64 kernel at 2e-6 where the reference reaches 7e-10)
    st = m._fit_state
    rhs = st["ws"] * st["Fy"]
    def resid(b):
        return float(torch.linalg.norm(rhs - (st["ws"] * m._toeplitz(st["ws"] * b) + st["sig"] * b)) / torch.linalg.norm(rhs))
    true_res = resid(m._beta)
    gold_res = resid(torch.from_numpy(r3[f"wfit_beta1_{tag}"]).cuda())      # the reference's beta through THIS operator
    ref_res = float(r3[f"wfit_true_resid1_{tag}"])                           # ... and through the reference's own
    print(f"\n{name} tol={tol:g}: true residual hip={true_res:.2e}; the reference's beta: {gold_res:.2e} through this operator, "
          f"{ref_res:.2e} through its own")
    assert true_res < 3.0 * max(gold_res, ref_res), (name, tag, true_res, gold_res, ref_res)
    print(f"{name} tol={tol:g}: iters hip=({it0}, {it1}) ref=({ref_it[0]}, {ref_it[1]}) cold={int(r3[f'wfit_cold_iters_{tag}'])} "
          f"ref-under-1e-13={tuple(s_it)}; rel err beta0={e0:.2e} beta1={e1:.2e} mean1={em:.2e} (reference's own move of beta1: {s_b:.2e}, of the mean: {s_m:.2e})")
    # the warm start must have been used: far fewer iterations than the cold solve at the default tolerance
    if tol == 1e-4:
        assert it1 < 0.6 * int(r3[f"wfit_cold_iters_{tag}"]), (it1, int(r3[f"wfit_cold_iters_{tag}"]))
    # iteration counts: equal, or inside the spread the reference itself shows on this case under a 1e-13 perturbation (+ 2 %)
    spread = int(np.abs(np.asarray(ref_it, dtype=np.int64) - np.asarray(s_it, dtype=np.int64)).max())
    for mine, theirs in ((it0, int(ref_it[0])), (it1, int(ref_it[1]))):
        assert abs(mine - theirs) <= spread + max(1, int(0.02 * theirs)), (name, tag, mine, theirs, spread)
    # results: 10 x the reference's own move under the perturbation, floors 1e-7 (tight solve) / the solve tolerance (default)
    # floors: two iterates whose true residuals are both ~rho differ by up to ~rho * cond; the tight solves end at the iteration
    # cap with rho = the stored residual (7e-10 on c2, cond 6e4), the default ones at rho = tol
    floor_b = 1e3 * ref_res if tol < 1e-8 else 2.0 * tol
    floor_m = 1e-7 if tol < 1e-8 else 2.0 * tol
    assert e1 < max(10.0 * s_b, floor_b), (name, tag, e1, s_b, floor_b)
    assert em < max(10.0 * s_m, floor_m), (name, tag, em, s_m)


@pytest.mark.parametrize("leg", ["tight", "default"])
@pytest.mark.parametrize("name", R3)
def test_warm_started_adam_trajectory_matches_reference(name, leg):
    """Four Adam steps (efgpnd.py:573-708 + optimizer step, lr 0.05) with the reference's default warm starts and the probes it
    drew: warm-start flags and mean-CG iteration counts of every step, gradients, hyper-parameters after every step."""
    g, x, y = load_case(name)
    r3 = load_r3(name)
    traj, grads, Ms = r3[f"wadam_{leg}_traj"], r3[f"wadam_{leg}_grads"], r3[f"wadam_{leg}_M"]
    ref_it, ref_warm, pert_it = r3[f"wadam_{leg}_mean_iters"], r3[f"wadam_{leg}_warm_used"], r3[f"sens_wadam_{leg}_iters"]
    s_traj, s_grad = float(r3[f"sens_wadam_{leg}_traj"]), float(r3[f"sens_wadam_{leg}_grad"])
    T, N = int(r3["T"]), x.shape[0]
    cg_tol = 1e-12 if leg == "tight" else None                      # None: the reference's default 0.1 eps (efgpnd.py:652-653)
    m = make_model(name, g, x.cuda(), y.cuda(), 1e-12 if leg == "tight" else 1e-4, nufft_eps=1e-12, mean_cg_warm_start=True)
    opt = torch.optim.Adam(m._gp_params.parameters(), lr=float(r3["lr"]))
    m.register_optimizer(opt)
    worst_g = worst_t = 0.0
    its = []
    for i in range(traj.shape[0]):
        Z = torch.from_numpy(np.unpackbits(r3[f"wadam_{leg}_Z{i}"], axis=1)[:, :N].astype(np.float64) * 2 - 1)
        V = torch.from_numpy(r3[f"wadam_{leg}_V{i}"].astype(np.float64))
        opt.zero_grad()
        grad = m.compute_gradients(trace_samples=T, nufft_eps=1e-12, cg_tol=cg_tol, probes_Z=Z, probes_V=V)
        st = m.last_gradient_stats
        assert int(st["feature_count"]) == int(Ms[i])
        assert bool(st["mean_cg_warm_start_used"]) == bool(ref_warm[i]), (i, st["mean_cg_warm_start_used"], ref_warm[i])
        its.append(int(st["mean_cg_iters"]))
        scale = float(np.abs(grads[i]).max())
        worst_g = max(worst_g, float((grad.detach().cpu() - torch.from_numpy(grads[i])).abs().max()) / scale)
        opt.step()
        now = np.array([float(m.kernel.get_hyper(n)) for n in m.kernel.hypers] + [float(m.sigmasq.detach())])
        worst_t = max(worst_t, float(np.abs(now - traj[i]).max() / np.abs(traj[i]).max()))
    print(f"\n{name} warm Adam ({leg}): iters hip={its} ref={list(ref_it)} ref-under-1e-13={list(pert_it)}; gradient deviation "
          f"{worst_g:.2e} of scale (reference's own move {s_grad:.2e}), hyper-parameters {worst_t:.2e} (reference's own {s_traj:.2e})")
    # iteration counts: equal, or inside the spread the reference itself shows on this case under a 1e-13 perturbation (+ 2 %)
    spread = int(np.abs(np.asarray(ref_it, dtype=np.int64) - np.asarray(pert_it, dtype=np.int64)).max())
    for mine, theirs in zip(its, ref_it):
        assert abs(mine - int(theirs)) <= spread + max(1, int(0.02 * int(theirs))), (name, leg, its, list(ref_it), list(pert_it))
    # 10 x the reference's own sensitivity; floors: 1e-6 of the gradient scale / 1e-7 of the hyper-parameters
    assert worst_g < max(10.0 * s_grad, 1e-6), (name, leg, worst_g, s_grad)
    assert worst_t < max(10.0 * s_traj, 1e-7), (name, leg, worst_t, s_traj)
