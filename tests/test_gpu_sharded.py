"""GPU test of the point-sharded product path: two ranks (one process each, both on the single test GPU,
gloo process group carrying the cuda tensors) each hold half of the observations; the all-reduced fit must
equal the unsharded fit.  On a multi-GPU node the same code runs with backend "nccl" (RCCL)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data():
    g = torch.Generator().manual_seed(11)
    N = 30001
    x = torch.rand(N, 2, generator=g, dtype=torch.float64) * 2 - 1
    y = torch.sin(3 * x[:, 0]) * torch.cos(2 * x[:, 1]) + 0.2 * torch.randn(N, generator=g, dtype=torch.float64)
    xn = torch.rand(257, 2, generator=g, dtype=torch.float64) * 2 - 1
    return x, y, xn


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from efgpnd import EFGPND
        from efgp_hip.dist import shard_bounds
        from kernels.squared_exponential import SquaredExponential
        x, y, xn = _data()
        lo, hi = shard_bounds(x.shape[0], world, rank)
        k = SquaredExponential(dimension=2, init_lengthscale=0.25, init_variance=1.3)
        m = EFGPND(x[lo:hi].cuda(), y[lo:hi].cuda(), k, sigmasq=0.1, eps=1e-4, nufft_eps=1e-9, estimate_params=False,
                   opts={"cg_tolerance": 1e-10, "shard_points": True})
        mean, _ = m.predict(xn.cuda(), return_variance=False)
        V = torch.ones(2, m.last_fit_stats["feature_count"], dtype=torch.float64)
        V[1, ::2] = -1
        Z = torch.ones(2, hi - lo, dtype=torch.float64)
        Z[0, ::3] = -1
        grad = m.compute_gradients(trace_samples=2, cg_tol=1e-10, probes_Z=Z, probes_V=V)
        # numpy arrays travel through the queue BY VALUE; torch tensors travel as file descriptors the receiver must fetch from a
        # sender that is still alive -- a worker that exits first makes q.get fail with ConnectionResetError
        q.put((rank, mean.cpu().numpy(), grad.detach().cpu().numpy(), int(m.last_fit_stats["mean_cg_iters"])))
    finally:
        dist.destroy_process_group()


def test_two_shards_equal_unsharded():
    sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
    from efgpnd import EFGPND
    from efgp_hip.dist import shard_bounds
    from kernels.squared_exponential import SquaredExponential
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    x, y, xn = _data()
    k = SquaredExponential(dimension=2, init_lengthscale=0.25, init_variance=1.3)
    m = EFGPND(x.cuda(), y.cuda(), k, sigmasq=0.1, eps=1e-4, nufft_eps=1e-9, estimate_params=False,
               opts={"cg_tolerance": 1e-10})
    mean, _ = m.predict(xn.cuda(), return_variance=False)
    V = torch.ones(2, m.last_fit_stats["feature_count"], dtype=torch.float64)
    V[1, ::2] = -1
    Zs = []
    for r in range(world):
        lo, hi = shard_bounds(x.shape[0], world, r)
        z = torch.ones(2, hi - lo, dtype=torch.float64)
        z[0, ::3] = -1
        Zs.append(z)
    grad = m.compute_gradients(trace_samples=2, cg_tol=1e-10, probes_Z=torch.cat(Zs, dim=1), probes_V=V).detach().cpu()
    for rank, smean, sgrad, its in res:
        smean, sgrad = torch.from_numpy(smean), torch.from_numpy(sgrad)
        # both solves stop at |r| < 1e-10 |b|, possibly one iteration apart (asserted below): the means agree to a small
        # multiple of cond(A) * tol, not to rounding
        assert float((smean - mean.cpu()).abs().max() / mean.cpu().abs().max()) < 3e-8
        scale = float(m.last_gradient_stats["term1"].abs().max()) * float(m._gp_params.pos.detach().max())
        assert float((sgrad - grad).abs().max()) < 1e-7 * scale
        assert abs(its - m.last_fit_stats["mean_cg_iters"]) <= 1


def _worker_free(rank, world, port, q):
    """No injected probes, DIFFERENT torch seeds per rank (the usual manual_seed(seed + rank)): the replicas must stay
    identical -- probe seed and feature-space probes come from rank 0 -- through two optimizer steps."""
    sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from efgpnd import EFGPND
        from efgp_hip.dist import shard_bounds
        from kernels.squared_exponential import SquaredExponential
        torch.manual_seed(1000 + rank)
        x, y, xn = _data()
        lo, hi = shard_bounds(x.shape[0], world, rank)
        k = SquaredExponential(dimension=2, init_lengthscale=0.25, init_variance=1.3)
        m = EFGPND(x[lo:hi].cuda(), y[lo:hi].cuda(), k, sigmasq=0.1, eps=1e-4, nufft_eps=1e-9, estimate_params=False,
                   opts={"cg_tolerance": 1e-10, "shard_points": True})
        opt = torch.optim.Adam(m._gp_params.parameters(), lr=0.1)
        m.register_optimizer(opt)
        grads, mtots = [], []
        for _ in range(2):
            opt.zero_grad()
            g = m.compute_gradients(trace_samples=3, cg_tol=1e-10)
            grads.append(g.detach().cpu().clone())
            mtots.append(int(m.last_gradient_stats["mtot"]))
            opt.step()
        _, var = m.predict(xn.cuda(), variance_method="stochastic", hutchinson_probes=8)
        hyp = [float(m.kernel.get_hyper(n)) for n in m.kernel.hypers] + [float(m.sigmasq.detach())]
        q.put((rank, torch.stack(grads).numpy(), mtots, hyp, var.cpu().numpy()))       # by value, see above
    finally:
        dist.destroy_process_group()


def test_two_shards_stay_identical_without_injected_probes():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_free, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, g0, m0, h0, v0), (_, g1, m1, h1, v1) = res
    g0, g1, v0, v1 = (torch.from_numpy(t) for t in (g0, g1, v0, v1))
    assert torch.equal(g0, g1), (g0, g1)               # bit-identical gradients on both ranks, both steps
    assert m0 == m1 and h0 == h1                       # same grids and hyper-parameters after two optimizer steps
    assert torch.equal(v0, v1)                         # the stochastic variance uses rank 0's probes too
    assert torch.isfinite(g0).all()


def _worker_rows(rank, world, port, q, shard_rows):
    """A grid beyond one workgroup (l = 0.1 -> 128 x 128 circulant grid): the 2T trace systems of the gradient and the probes
    of the stochastic variance are split by rows over the ranks (efgp_hip.dist.solve_rows_sharded) or, with EFGP_SHARD_ROWS=0,
    solved by every rank."""
    sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = "0"
    os.environ["EFGP_SHARD_ROWS"] = "1" if shard_rows else "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import efgpnd as E
        from efgp_hip.dist import shard_bounds
        from kernels.squared_exponential import SquaredExponential
        calls = []
        orig = E._rows_over_ranks

        def spy(shards, top, R):
            r = orig(shards, top, R)
            calls.append((int(R), bool(r)))
            return r
        E._rows_over_ranks = spy
        x, y, xn = _data()
        lo, hi = shard_bounds(x.shape[0], world, rank)
        k = SquaredExponential(dimension=2, init_lengthscale=0.1, init_variance=1.3)
        m = E.EFGPND(x[lo:hi].cuda(), y[lo:hi].cuda(), k, sigmasq=0.1, eps=1e-4, nufft_eps=1e-9, estimate_params=False,
                     opts={"cg_tolerance": 1e-10, "shard_points": True})
        m.fit()
        M = int(m.last_fit_stats["feature_count"])
        g = torch.Generator().manual_seed(5)
        V = torch.empty(3, M, dtype=torch.float64).bernoulli_(0.5, generator=g) * 2 - 1
        Z = torch.empty(3, x.shape[0], dtype=torch.float64).bernoulli_(0.5, generator=g) * 2 - 1
        grad = m.compute_gradients(trace_samples=3, cg_tol=1e-10, probes_Z=Z[:, lo:hi], probes_V=V)
        its = int(m.last_gradient_stats["trace_cg_iters"])
        P = torch.empty(6, M, dtype=torch.float64).bernoulli_(0.5, generator=g) * 2 - 1
        _, var = m.predict(xn.cuda(), variance_method="stochastic", hutchinson_probes=6, variance_probes=P)
        q.put((rank, grad.detach().cpu().numpy(), var.cpu().numpy(), its, calls, int(m.last_fit_stats["mtot"])))
    finally:
        dist.destroy_process_group()


def _run_rows(shard_rows):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_rows, args=(r, world, port, q, shard_rows)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_rows_of_batched_solves_split_over_ranks_equal_replicated():
    split, repl = _run_rows(True), _run_rows(False)
    assert split[0][5] > 32                                          # a grid beyond the one-workgroup kernel
    # the split really happened (gradient: 2T = 6 systems, variance: 6 probes) and did not happen in the replicated run
    assert (6, True) in split[0][4] and all(not c[1] for c in repl[0][4])
    for a, b in ((split[0], split[1]), (repl[0], repl[1])):          # identical on both ranks
        assert (a[1] == b[1]).all() and (a[2] == b[2]).all() and a[3] == b[3]
    g_s, g_r = torch.from_numpy(split[0][1]), torch.from_numpy(repl[0][1])
    v_s, v_r = torch.from_numpy(split[0][2]), torch.from_numpy(repl[0][2])
    # the systems are independent; the cooperative launch picks its workgroups per system from the NUMBER of systems, so the
    # order of the reductions (not the iterates' mathematics) differs between 3 and 6 rows: equal to ~cond * rounding at
    # cg_tol 1e-10 (measured 2e-10)
    assert float((g_s - g_r).abs().max() / g_r.abs().max()) < 1e-8
    assert float((v_s - v_r).abs().max() / v_r.abs().max()) < 1e-8
    assert abs(split[0][3] - repl[0][3]) <= 1
