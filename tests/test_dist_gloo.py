"""world_size-2 gloo test of the point-sharding layer (efgp_hip/dist.py): the gridded partial sums of
per-shard type-1 transforms, all-reduced, equal the unsharded transform; scalars and bounds reduce.
The per-shard compute here is the oracle's exact NUDFT (CPU); on the GPU box the same PointShards
object wraps RCCL ("nccl") tensors."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from efgp_hip.dist import PointShards, shard_bounds
        from oracle import efgp_oracle as O
        g = torch.Generator().manual_seed(7)
        N, d, h, mtot = 1001, 2, 0.31, 9
        x = torch.rand(N, d, generator=g, dtype=torch.float64) * 2 - 1
        y = torch.randn(N, generator=g, dtype=torch.float64)
        lo, hi = shard_bounds(N, world, rank)
        sh = PointShards()
        assert sh.active and sh.world_size == world and sh.rank == rank
        xs, ys = x[lo:hi], y[lo:hi]
        Fy = O.nudft_type1(xs, h, ys, (mtot, mtot))
        v = O.conv_vector(xs, h, (mtot - 1) // 2)
        sh.sum_many_([Fy, v])
        Fy_full = O.nudft_type1(x, h, y, (mtot, mtot))
        v_full = O.conv_vector(x, h, (mtot - 1) // 2)
        e1 = float((Fy - Fy_full).abs().max() / Fy_full.abs().max())
        e2 = float((v - v_full).abs().max() / v_full.abs().max())
        # the product path hands sum_many_ two views of ONE buffer (NufftPlan.type1_pair): reduced in place, no packing
        from efgp_hip.dist import _adjacent_span
        Fy2 = O.nudft_type1(xs, h, ys, (mtot, mtot))
        v2 = O.conv_vector(xs, h, (mtot - 1) // 2)
        flat = torch.cat([Fy2.reshape(-1), v2.reshape(-1)])
        Fy_v, v_v = flat[:Fy2.numel()].view(Fy2.shape), flat[Fy2.numel():].view(v2.shape)
        assert _adjacent_span([Fy_v, v_v]) is not None and _adjacent_span([Fy2, v2]) is None
        sh.sum_many_([Fy_v, v_v])
        e1 = max(e1, float((Fy_v - Fy_full).abs().max() / Fy_full.abs().max()))
        e2 = max(e2, float((v_v - v_full).abs().max() / v_full.abs().max()))
        Z = torch.ones(3, hi - lo, dtype=torch.float64)
        FZ = O.nudft_type1(xs, h, Z, (mtot, mtot)).reshape(3, -1)
        sh.sum_(FZ)
        e3 = float((FZ[0] - v_full[4:13, 4:13].reshape(-1)).abs().max())
        # replicated random state must come from one rank: broadcast_ / shared_seed with different per-rank generators
        torch.manual_seed(100 + rank)
        V = torch.empty(4, 9).bernoulli_(0.5)
        sh.broadcast_(V)
        seed = sh.shared_seed("cpu")
        gathered = [None] * world
        dist.all_gather_object(gathered, (V.tolist(), seed))
        assert all(gv == gathered[0] for gv in gathered), "broadcast_/shared_seed differ across ranks"
        tot = sh.sum_scalars([float(hi - lo), float(ys.sum())], "cpu")
        mn, mx = sh.minmax(xs.min(0).values, xs.max(0).values)
        ok = (e1 < 1e-12 and e2 < 1e-12 and e3 < 1e-9 and tot[0] == N and abs(tot[1] - float(y.sum())) < 1e-9
              and torch.equal(mn, x.min(0).values) and torch.equal(mx, x.max(0).values))
        q.put((rank, ok, e1, e2, e3))
    finally:
        dist.destroy_process_group()


def test_sharded_type1_allreduce_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res


def _rows_worker(rank, world, port, q):
    """solve_rows_sharded: the R independent systems of a batched solve split by rows over the ranks, one all-reduce to gather --
    bit for bit the replicated solve (CPU stand-in for the per-block solver: the oracle's batched CG, cg.py:155-244)."""
    sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from efgp_hip.dist import PointShards, solve_rows_sharded, shard_bounds
        from oracle import efgp_oracle as O
        g = torch.Generator().manual_seed(3)
        N, d, h, mtot = 400, 2, 0.31, 7
        x = torch.rand(N, d, generator=g, dtype=torch.float64) * 2 - 1
        v = O.conv_vector(x, h, (mtot - 1) // 2)
        T = O.Toeplitz(v)
        M = mtot ** d
        ws = (0.3 + torch.rand(M, generator=g, dtype=torch.float64)).to(torch.complex128)
        A = O.make_A_mean(ws, T, 100.0)          # well conditioned: every row converges long before the iteration cap
        ok = True
        worst = 0.0
        for R in (5, 2, 1):                                   # ragged blocks, one row per rank, fewer rows than ranks
            rhs = torch.complex(torch.randn(R, M, generator=g, dtype=torch.float64), torch.randn(R, M, generator=g, dtype=torch.float64))
            calls = []

            def solve(block):
                calls.append(block.shape[0])
                xb, its = O.cg_batched(A, block, torch.zeros_like(block), 1e-10)
                return xb, [its] * block.shape[0]
            sh = PointShards()
            X, rows = solve_rows_sharded(sh, rhs, solve)
            Xr, its = O.cg_batched(A, rhs, torch.zeros_like(rhs), 1e-10)
            lo, hi = shard_bounds(R, world, rank)
            ok = ok and calls == ([hi - lo] if hi > lo else [])            # this rank solved its block only
            ok = ok and rows.dtype == torch.int32 and rows.shape == (R,) and int(rows.min()) > 0
            # the batched oracle masks converged rows, so a row's iterate does not depend on its neighbours: equal to rounding
            worst = max(worst, float((X - Xr).abs().max() / Xr.abs().max()))
            gathered = [None] * world
            dist.all_gather_object(gathered, X.numpy().tobytes())
            ok = ok and all(gv == gathered[0] for gv in gathered)          # identical bits on all ranks
        # a rank whose local solve raises must not leave its peers waiting in the all-reduce: every rank raises behind it
        rhs = torch.complex(torch.randn(4, M, generator=g, dtype=torch.float64), torch.randn(4, M, generator=g, dtype=torch.float64))

        def solve_bad(block):
            if rank == 1:
                raise ValueError("refused on rank 1")
            xb, its = O.cg_batched(A, block, torch.zeros_like(block), 1e-10)
            return xb, [its] * block.shape[0]
        try:
            solve_rows_sharded(PointShards(), rhs, solve_bad)
            ok = False
        except RuntimeError as err:
            ok = ok and "1 of 2 ranks" in str(err) and (("refused on rank 1" in str(err)) == (rank == 1))
        # a per-rank decision that selects collectives is checked for agreement (and cached per key)
        sh = PointShards()
        ok = ok and sh.agree("same", True, "cpu") is True and sh.agree("same", True, "cpu") is True
        try:
            sh.agree("differs", rank == 0, "cpu")
            ok = False
        except RuntimeError as err:
            ok = ok and "disagree" in str(err)
        q.put((rank, ok and worst < 1e-9, worst))
    finally:
        dist.destroy_process_group()


def test_rows_of_batched_solves_sharded_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rows_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` with WORLD_SIZE unset starts N ranks itself (torch.distributed.run as a child, before anything
    touches the GPU), and a rank whose WORLD_SIZE differs from --gpus refuses to measure (round-2 verdict: `--gpus 8` used to
    run ONE GPU and print n_gpus: 1)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--check-launch"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = sorted(l for l in out.stdout.splitlines() if l.startswith("bench.py rank"))
    assert lines == ["bench.py rank 0 of 2 (local 0)", "bench.py rank 1 of 2 (local 1)"], out.stdout + out.stderr[-1000:]
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--check-launch"],
                         env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=2" in (bad.stderr + bad.stdout)


def test_shard_bounds_partition():
    sys.path.insert(0, os.path.join(ROOT, "gp-quadrature_amd"))
    from efgp_hip.dist import shard_bounds, PointShards
    for n in [0, 1, 7, 8, 1000003]:
        for w in [1, 2, 3, 8]:
            segs = [shard_bounds(n, w, r) for r in range(w)]
            assert segs[0][0] == 0 and segs[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(segs, segs[1:]))
            sizes = [b - a for a, b in segs]
            assert max(sizes) - min(sizes) <= 1
    sh = PointShards(enabled=False)
    t = torch.ones(3)
    assert not sh.active and sh.sum_(t) is t and sh.sum_scalars([1.5], "cpu") == [1.5]
