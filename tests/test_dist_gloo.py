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
