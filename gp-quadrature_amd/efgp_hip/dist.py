"""Sharding of the N observation points over the GPUs of one node (one process per GPU).

The EFGP solve path touches the N points only in NUFFT passes; CG works on M-sized grids.  So the
points are partitioned into contiguous blocks, every rank runs the spread kernels on its block,
and the small gridded partial sums (F*y: mtot^d, Toeplitz vector: (4m+1)^d, batched F*Z) plus a
handful of N-length scalar reductions are summed with ONE all-reduce each (RCCL over xGMI when the
backend is "nccl"; gloo in the CPU tests).  The mean solve is replicated: no communication inside it.  The
BATCHED solves (2T trace systems of a gradient, J Hutchinson probes of the variance) are independent systems and
are split by rows over the ranks, with one all-reduce to gather the solutions (`solve_rows_sharded`, round 3).
(The reference has no distributed code at all)
"""
import torch
import torch.distributed as dist


def shard_bounds(n, world_size, rank):
    """[lo, hi) of the contiguous block of `n` points owned by `rank` (sizes differ by at most 1)."""
    base, rem = divmod(int(n), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _adjacent_span(tensors):
    """One flat view covering `tensors` when they are contiguous views placed back to back in the same storage
    (same dtype and device), else None.  Saves the pack / unpack copies around the fused all-reduce."""
    t0 = tensors[0]
    off = t0.storage_offset()
    for t in tensors:
        if not t.is_contiguous() or t.dtype != t0.dtype or t.device != t0.device or \
                t.untyped_storage().data_ptr() != t0.untyped_storage().data_ptr() or t.storage_offset() != off:
            return None
        off += t.numel()
    return torch.as_strided(t0, (off - t0.storage_offset(),), (1,), t0.storage_offset())


def rccl_comm_from_env(device):
    """An `RcclComm` (C ABI efgp_comm_*, no torch process group) for the launch environment of torch.distributed.run:
    RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT.  The 128-byte RCCL id travels through a TCPStore on MASTER_PORT + 1."""
    import os
    from datetime import timedelta
    from .ops import RcclComm
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return RcclComm(device, 0, 1, RcclComm.make_id())
    store = dist.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29500")) + 1, world,
                          is_master=(rank == 0), timeout=timedelta(seconds=300))
    if rank == 0:
        uid = RcclComm.make_id()
        store.set("efgp_rccl_id", uid)
    else:
        uid = bytes(store.get("efgp_rccl_id"))
    return RcclComm(device, rank, world, uid)


def solve_rows_sharded(shards, rhs, solve):
    """Independent systems over the GPUs (SURVEY 8e's alternative to replicating every solve; the reference's batched solves are
    efgpnd.py:205-236 -- the 2T trace systems of a gradient -- and :1651-1657 -- the J Hutchinson probes of the variance).

    `rhs` (R, M) is IDENTICAL on every rank (it is built from all-reduced / broadcast data).  Rank r solves the contiguous row
    block shard_bounds(R, world, r) with `solve(block) -> (x (r, M), rows)` (`rows`: per-system iteration counts, a device
    tensor or a list), and ONE all-reduce of a buffer that is zero outside the rank's own rows gathers the solutions: adding
    zeros is exact, so every rank ends with bit-for-bit the result of the replicated solve.  Returns (x (R, M), rows (R,) int32
    device tensor), identical on all ranks."""
    R = rhs.shape[0]
    lo, hi = shard_bounds(R, shards.world_size, shards.rank)
    full = torch.zeros_like(rhs)
    rows = torch.zeros(R + 1, dtype=torch.float64, device=rhs.device)     # [R]: number of ranks whose local solve failed
    failure = None
    if hi > lo:
        # a rank whose solve raises (refused input, out of memory, a HIP error) must still reach the collective -- its peers
        # would wait in it for ever -- so the failure travels as a flag and EVERY rank raises behind the all-reduce
        try:
            x, its = solve(rhs[lo:hi])
            full[lo:hi] = x.reshape(hi - lo, -1)
            rows[lo:hi] = torch.as_tensor(its, device=rhs.device).to(torch.float64).reshape(-1)
        except Exception as err:       # noqa: BLE001 -- re-raised below, on all ranks
            failure = err
            full.zero_()
            rows[R] = 1.0
    shards.sum_many_([full, rows])
    failed = int(rows[R].item()) if (failure is not None or shards.active) else 0
    if failed:
        raise RuntimeError(f"efgp_hip: the row-sharded batched solve failed on {failed} of {shards.world_size} ranks"
                           + (f" (this rank: {failure})" if failure is not None else " (not this one)")) from failure
    return full, rows[:R].to(torch.int32)


class PointShards:
    """All-reduce helper bound to a torch process group, or to an `RcclComm` (C ABI collectives, no process group) when
    `comm` is given; a no-op when world_size == 1."""

    def __init__(self, group=None, enabled=None, comm=None):
        self.comm = comm
        if comm is not None:
            self.enabled = True
            self.group = None
            self.world_size, self.rank = comm.world, comm.rank
            return
        if enabled is None:
            enabled = dist.is_available() and dist.is_initialized()
        self.enabled = bool(enabled) and dist.is_available() and dist.is_initialized()
        self.group = group
        self.world_size = dist.get_world_size(group) if self.enabled else 1
        self.rank = dist.get_rank(group) if self.enabled else 0

    @property
    def active(self):
        return self.enabled and self.world_size > 1

    def agree(self, key, value, device):
        """A per-rank decision that selects which collectives follow (e.g. whether a batched solve is split by rows: it depends
        on the environment and on the local device) must be the same on every rank, or the ranks enter mismatched collectives
        and hang.  Checked once per `key` with one small all-reduce; raises on every rank when the ranks differ."""
        cache = self.__dict__.setdefault("_agreed", {})
        if key in cache or not self.active:
            return value
        v = float(int(value))
        tot = self.sum_scalars([v, v * v], device)
        if abs(tot[0] - v * self.world_size) > 0.5 or abs(tot[1] - v * v * self.world_size) > 0.5:
            raise RuntimeError(f"efgp_hip: ranks disagree on {key!r} (this rank: {value}); set EFGP_SHARD_ROWS identically on all "
                               "ranks and use identical devices")
        cache[key] = value
        return value

    def sum_(self, t):
        """In-place SUM all-reduce of a real or complex tensor (complex goes as interleaved reals)."""
        if not self.active:
            return t
        buf = torch.view_as_real(t) if t.is_complex() else t
        if not buf.is_contiguous():
            tmp = buf.contiguous()
            self._allreduce_sum(tmp)
            buf.copy_(tmp)
        else:
            self._allreduce_sum(buf)
        return t

    def _allreduce_sum(self, buf):
        if self.comm is not None:
            self.comm.all_reduce_sum_(buf)
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)

    def sum_many_(self, tensors):
        """One fused all-reduce for several small tensors (latency-bound messages)."""
        if not self.active or not tensors:
            return tensors
        span = _adjacent_span(tensors)
        if span is not None:          # views laid out back to back in one buffer (NufftPlan.type1_pair): reduce in place
            self._allreduce_sum(torch.view_as_real(span) if span.is_complex() else span)
            return tensors
        flats = [(torch.view_as_real(t) if t.is_complex() else t).reshape(-1) for t in tensors]
        packed = torch.cat(flats)
        self._allreduce_sum(packed)
        off = 0
        for t, f in zip(tensors, flats):
            n = f.numel()
            (torch.view_as_real(t) if t.is_complex() else t).reshape(-1).copy_(packed[off:off + n])
            off += n
        return tensors

    def sum_scalars(self, values, device):
        """SUM-reduce a list of Python floats; returns Python floats."""
        if not self.active:
            return [float(v) for v in values]
        t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
        self._allreduce_sum(t)
        return [float(v) for v in t.tolist()]

    def exclusive_offset(self, n_local, device):
        """Global index of this rank's first point (ranks hold consecutive blocks of the observations)."""
        if not self.active:
            return 0
        if self.comm is not None:      # counts as a one-hot sum (exact in float64 below 2^53)
            t = torch.zeros(self.world_size, dtype=torch.float64, device=device)
            t[self.rank] = float(n_local)
            self._allreduce_sum(t)
            return int(t[:self.rank].sum().item())
        counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(self.world_size)]
        dist.all_gather(counts, torch.tensor([int(n_local)], dtype=torch.int64, device=device), group=self.group)
        return int(sum(int(c.item()) for c in counts[:self.rank]))

    def broadcast_(self, t, src=0):
        """Overwrite `t` on every rank with rank `src`'s copy (group rank).  Replicated state that is DRAWN at random
        -- probe seeds, feature-space probes -- must come from one rank: replicas that draw their own diverge after
        the first optimizer step (different gradients -> different grids -> mismatched all-reduce sizes)."""
        if not self.active:
            return t
        if self.comm is not None:
            if t.is_contiguous():
                self.comm.broadcast_(t, src)
            else:
                tmp = t.contiguous()
                self.comm.broadcast_(tmp, src)
                t.copy_(tmp)
            return t
        root = dist.get_global_rank(self.group, src) if self.group is not None else src
        buf = torch.view_as_real(t) if t.is_complex() else t
        if not buf.is_contiguous():
            tmp = buf.contiguous()
            dist.broadcast(tmp, src=root, group=self.group)
            buf.copy_(tmp)
        else:
            dist.broadcast(buf, src=root, group=self.group)
        return t

    def shared_seed(self, device, generator=None):
        """A 62-bit seed drawn on rank 0 (from torch's generator there) and broadcast: identical on all ranks."""
        seed = torch.randint(0, 2 ** 62, (1,), generator=generator, dtype=torch.int64)
        if not self.active:
            return int(seed.item())
        t = seed.to(device)
        self.broadcast_(t)
        return int(t.item())

    def minmax(self, lo, hi):
        """Global per-dimension min / max of the point coordinates (for the domain length L)."""
        if not self.active:
            return lo, hi
        lo = lo.clone()
        hi = hi.clone()
        if self.comm is not None:
            self.comm.all_reduce_minmax_(lo, False)
            self.comm.all_reduce_minmax_(hi, True)
            return lo, hi
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        return lo, hi
