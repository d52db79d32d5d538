"""Tensor-level wrappers over the C ABI: torch tensors carry the device memory and the stream,
every computation happens in libefgp_hip.so."""
import ctypes as C
import os

import torch

from .lib import lib, check

_CD = torch.complex128
_RD = torch.float64


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("the EFGP HIP path needs an AMD GPU (torch.cuda.is_available() is False); "
                           "there is no CPU fallback in this package")


def compute_device(*tensors, device=None):
    """Device the kernels run on: an explicit cuda device, else the device of the first cuda tensor,
    else cuda:LOCAL_RANK (one process per GPU)."""
    require_gpu()
    if device is not None:
        dev = torch.device(device)
        if dev.type == "cuda":
            return torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
    for t in tensors:
        if torch.is_tensor(t) and t.is_cuda:
            return t.device
    idx = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    return torch.device("cuda", idx)


# The wrappers below run a dozen times per fit / gradient step, and the step had become HOST-bound (round 4: 300 us of kernels in a
# 430-us gradient step): `torch.cuda.current_stream(dev)` builds a Stream object (5.7 us) and `torch.cuda.device(dev)` resolves its
# argument through `_get_device_index` (2.7 us, twice per wrapper) -- ~25 such calls per step.  Both have direct C entry points.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_exchange = getattr(torch._C, "_cuda_exchangeDevice", None)
_maybe_exchange = getattr(torch._C, "_cuda_maybeExchangeDevice", None) or _exchange


def _stream(dev):
    if _raw_stream is not None and dev.index is not None:
        return C.c_void_p(_raw_stream(dev.index))
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class _on:
    """`with _on(dev):` is `with torch.cuda.device(dev):` without the argument resolution and object churn."""
    __slots__ = ("idx", "prev", "ctx", "_dev")

    def __init__(self, dev):
        self._dev = dev
        self.idx = dev.index if (_exchange is not None and getattr(dev, "index", None) is not None) else -1
        self.prev = -1
        self.ctx = None

    def __enter__(self):
        if self.idx >= 0:
            self.prev = _exchange(self.idx)
        else:
            self.ctx = torch.cuda.device(self._dev)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.idx >= 0:
            _maybe_exchange(self.prev)
            return False
        return self.ctx.__exit__(*exc)


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _dc(t, dev, dtype):
    """`t.to(device=dev, dtype=dtype).contiguous()`, returning `t` itself when it already is all that: the no-op conversions cost
    ~2 us each and a step makes some thirty of them."""
    if t.dtype is dtype and t.device == dev and t.is_contiguous():
        return t
    return t.to(device=dev, dtype=dtype).contiguous()


def _i64(vals):
    return (C.c_int64 * len(vals))(*[int(v) for v in vals])


class PointSet:
    """Per-model layout of the observation points (C ABI: efgp_points_*): bounding box and, for d = 2, copies
    sorted by a grid-independent key that the type-1 pass of every later plan streams (csrc/points_layout.hpp).
    `values` (the model's targets) may be attached so that plans read a sorted copy and max|y| is computed once."""

    def __init__(self, x, values=None):
        assert x.is_cuda and x.dtype == _RD and x.ndim == 2 and x.is_contiguous()
        self.x = x                    # keeps the storage alive; the library does not copy
        self.dev = x.device
        self.npts, self.dim = x.shape
        self.values = None
        self._h = C.c_void_p()
        with _on(self.dev):
            check(lib().efgp_points_create(C.byref(self._h), self.dev.index, self.dim, self.npts, _ptr(x), _stream(self.dev)),
                  "efgp_points_create")
        if values is not None:
            self.attach_values(values)

    def attach_values(self, y):
        assert y.is_cuda and y.dtype == _RD and y.is_contiguous() and y.numel() == self.npts
        self.values = y
        with _on(self.dev):
            check(lib().efgp_points_attach_values(self._h, _ptr(y), _stream(self.dev)), "efgp_points_attach_values")

    def bounds(self):
        lo = (C.c_double * 3)()
        hi = (C.c_double * 3)()
        check(lib().efgp_points_bounds(self._h, lo, hi), "efgp_points_bounds")
        return [float(lo[a]) for a in range(self.dim)], [float(hi[a]) for a in range(self.dim)]

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().efgp_points_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NufftPlan:
    """Type-1 / type-2 transforms for a fixed point set (C ABI: efgp_nufft_*).  With `points` (a PointSet over the
    same x) the plan is made on the model's layout (efgp_nufft_create_on)."""

    def __init__(self, x, h, tol, xcen=None, points=None):
        assert x.is_cuda and x.dtype == _RD and x.ndim == 2 and x.is_contiguous()
        self.x = x                    # keeps the storage alive; the library does not copy
        self.points = points          # the layout must outlive the plan
        self.dev = x.device
        self.npts, self.dim = x.shape
        self.h = float(h)
        self.tol = float(tol)
        xc = None
        if xcen is not None:
            vals = [float(v) for v in xcen]
            if any(v != 0.0 for v in vals):
                xc = (C.c_double * self.dim)(*vals)
        self._h = C.c_void_p()
        if points is not None:
            assert points.x.data_ptr() == x.data_ptr() and points.npts == self.npts
            check(lib().efgp_nufft_create_on(C.byref(self._h), points._h, xc, self.h, self.tol), "efgp_nufft_create_on")
        else:
            check(lib().efgp_nufft_create(C.byref(self._h), self.dev.index, self.dim, self.npts, _ptr(x), xc,
                                          self.h, self.tol), "efgp_nufft_create")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().efgp_nufft_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def type1(self, c, n_modes, modeord=0, isign=-1):
        """c (N,) or (B,N) real or complex -> (B?, *n_modes) complex128."""
        batched = c.ndim > 1
        cc = c.reshape(-1, self.npts)
        is_c = cc.is_complex()
        cc = cc.to(device=self.dev, dtype=_CD if is_c else _RD).contiguous()
        B = cc.shape[0]
        out = torch.empty((B,) + tuple(int(m) for m in n_modes), dtype=_CD, device=self.dev)
        with _on(self.dev):
            check(lib().efgp_nufft_type1(self._h, _ptr(cc), int(is_c), B, _i64(n_modes), isign, int(modeord),
                                         _ptr(out), _stream(self.dev)), "efgp_nufft_type1")
        return out if batched else out[0]

    def type1_rademacher(self, seed, nbatch, n_modes, index_offset=0, modeord=0):
        """F* Z for Z[b,n] = +-1 generated in the kernel from (seed, b, n + index_offset) -> (B, *n_modes)."""
        out = torch.empty((int(nbatch),) + tuple(int(m) for m in n_modes), dtype=_CD, device=self.dev)
        with _on(self.dev):
            check(lib().efgp_nufft_type1_rademacher(self._h, int(seed) & (2 ** 64 - 1), int(index_offset), int(nbatch),
                                                    _i64(n_modes), int(modeord), _ptr(out), _stream(self.dev)),
                  "efgp_nufft_type1_rademacher")
        return out

    def type1_pair(self, y, n_modes_y, n_modes_one):
        """One pass over the points: (F* y on n_modes_y, F* 1 on n_modes_one)."""
        yy = _dc(y, self.dev, _RD)
        shape_y, shape_o = tuple(int(m) for m in n_modes_y), tuple(int(m) for m in n_modes_one)
        My = 1
        for m in shape_y:
            My *= m
        Mo = 1
        for m in shape_o:
            Mo *= m
        # one buffer, two views: the sharded fit all-reduces both results in place with a single collective
        flat = torch.empty(My + Mo, dtype=_CD, device=self.dev)
        out_y, out_o = flat[:My].view(shape_y), flat[My:].view(shape_o)
        with _on(self.dev):
            check(lib().efgp_nufft_type1_pair(self._h, _ptr(yy), _i64(n_modes_y), _ptr(out_y), _i64(n_modes_one),
                                              _ptr(out_o), _stream(self.dev)), "efgp_nufft_type1_pair")
        return out_y, out_o

    def type1_ones(self, n_modes):
        out_o = torch.empty(tuple(int(m) for m in n_modes), dtype=_CD, device=self.dev)
        with _on(self.dev):
            check(lib().efgp_nufft_type1_pair(self._h, None, None, None, _i64(n_modes), _ptr(out_o),
                                              _stream(self.dev)), "efgp_nufft_type1_pair")
        return out_o

    def type2(self, f, n_modes, modeord=0, real_only=False, isign=+1, batched=None, mode_scale=None):
        """f (prod,) | (*n_modes) | (B, ...) complex -> (N,) | (B,N) complex128 (float64 if real_only).
        mode_scale (prod,) complex: the modes are multiplied by it inside the transform (F (ws * beta))."""
        M = 1
        for m in n_modes:
            M *= int(m)
        if batched is None:
            batched = not (f.ndim == 1 or tuple(f.shape) == tuple(int(m) for m in n_modes))
        ff = _dc(f.reshape(-1, M), self.dev, _CD)
        B = ff.shape[0]
        out = torch.empty((B, self.npts), dtype=_RD if real_only else _CD, device=self.dev)
        with _on(self.dev):
            if mode_scale is None:
                check(lib().efgp_nufft_type2(self._h, _ptr(ff), B, _i64(n_modes), isign, int(modeord), _ptr(out),
                                             int(bool(real_only)), _stream(self.dev)), "efgp_nufft_type2")
            else:
                sc = _dc(mode_scale.reshape(-1), self.dev, _CD)
                if sc.numel() != M:
                    raise ValueError(f"mode_scale has {sc.numel()} entries, the mode box has {M}")
                check(lib().efgp_nufft_type2_scaled(self._h, _ptr(ff), _ptr(sc), B, _i64(n_modes), isign, int(modeord),
                                                    _ptr(out), int(bool(real_only)), _stream(self.dev)),
                      "efgp_nufft_type2_scaled")
        return out if batched else out[0]


class ToeplitzOp:
    """d-dimensional Toeplitz mat-vec (C ABI: efgp_toeplitz_*)."""

    def __init__(self, v, force_pow2=True):
        assert v.is_cuda
        self.dev = v.device
        self.v = v.to(_CD).contiguous()
        self.d = self.v.ndim
        self.Ls = list(self.v.shape)
        self.ns = [(L + 1) // 2 for L in self.Ls]
        self.size = 1
        for n in self.ns:
            self.size *= n
        self._h = C.c_void_p()
        with _on(self.dev):
            check(lib().efgp_toeplitz_create(C.byref(self._h), self.dev.index, self.d, _i64(self.Ls), _ptr(self.v),
                                             int(bool(force_pow2)), _stream(self.dev)), "efgp_toeplitz_create")
        shp = (C.c_int64 * 3)()
        check(lib().efgp_toeplitz_fft_shape(self._h, shp), "efgp_toeplitz_fft_shape")
        self.fft_shape = [int(shp[a]) for a in range(self.d)]

    def cg_shape(self, hermitian=False):
        """Circulant grid the fused CG solves of this operator run on (efgp_toeplitz_cg_shape): the smallest one the solvers'
        transforms cover that holds 2 n - 1 per axis; `fft_shape` stays the reference's grid."""
        shp = (C.c_int64 * 3)()
        check(lib().efgp_toeplitz_cg_shape(self._h, int(bool(hermitian)), shp), "efgp_toeplitz_cg_shape")
        return [int(shp[a]) for a in range(self.d)]

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().efgp_toeplitz_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def apply(self, u):
        """u (..., size) complex on the device -> same shape."""
        uu = _dc(u.reshape(-1, self.size), self.dev, _CD)
        out = torch.empty_like(uu)
        with _on(self.dev):
            check(lib().efgp_toeplitz_apply(self._h, _ptr(uu), uu.shape[0], _ptr(out), _stream(self.dev)),
                  "efgp_toeplitz_apply")
        return out.reshape(u.shape)

    def apply_scaled(self, u, pre=None, post=None, out=None):
        """post .* T(pre .* u) for u (..., size) complex or real float64 on the device (efgp_toeplitz_apply_scaled); pre / post:
        (size,) complex128 device tensors or None.  `out`: a contiguous complex128 (rows, size) device tensor to write into."""
        real = not u.is_complex()
        uu = u.reshape(-1, self.size).to(device=self.dev, dtype=_RD if real else _CD).contiguous()
        res = out if out is not None else torch.empty(uu.shape, dtype=_CD, device=self.dev)
        assert res.is_contiguous() and res.dtype == _CD and res.numel() == uu.numel()
        for dgl in (pre, post):
            assert dgl is None or (dgl.is_cuda and dgl.dtype == _CD and dgl.is_contiguous() and dgl.numel() == self.size)
        with _on(self.dev):
            check(lib().efgp_toeplitz_apply_scaled(self._h, _ptr(uu), int(real), uu.shape[0], _ptr(pre) if pre is not None else None,
                                                   _ptr(post) if post is not None else None, _ptr(res), _stream(self.dev)),
                  "efgp_toeplitz_apply_scaled")
        return res if out is not None else res.reshape(u.shape)


def gradient_prepare(ws, fy, v_center, sigmasq, want_diag=True, want_rhs=True):
    """(diag, rhs) = (Re v_center |ws|^2 + sigmasq, ws .* fy) in one launch (efgp_gradient_prepare); v_center: a one-element
    complex128 device view of the Toeplitz vector's centre."""
    dev = ws.device
    M = ws.numel()
    assert ws.dtype == _CD and ws.is_contiguous()
    diag = torch.empty(M, dtype=_RD, device=dev) if want_diag else None
    rhs = torch.empty(M, dtype=_CD, device=dev) if want_rhs else None
    ff = fy.reshape(-1).to(dtype=_CD).contiguous() if want_rhs else None
    if want_diag:
        assert v_center.is_cuda and v_center.dtype == _CD and v_center.numel() == 1
    with _on(dev):
        check(lib().efgp_gradient_prepare(dev.index, M, _ptr(ws), _ptr(ff) if ff is not None else None,
                                          _ptr(v_center) if want_diag else None, float(sigmasq), _ptr(diag) if want_diag else None,
                                          _ptr(rhs) if want_rhs else None, _stream(dev)), "efgp_gradient_prepare")
    return diag, rhs


def gradient_assemble(fy, tg, ws, beta, dprime, fz, v, beta_all, *, variance_idx, trace_idx, sigmasq, n_obs, yy, variance):
    """grad | term1 | term2 | y.alpha as ONE device vector of 3 (H + 1) + 1 doubles (efgp_gradient_assemble)."""
    dev = ws.device
    M = ws.numel()
    H = dprime.shape[1] if dprime is not None and dprime.ndim == 2 else 0
    T = v.shape[0]
    K = len(trace_idx)
    for t_ in (fy, tg, ws, beta, beta_all) + ((dprime,) if H else ()) + ((fz,) if K else ()):
        assert t_.is_cuda and t_.dtype == _CD and t_.is_contiguous()
    assert v.dtype == _RD and v.is_contiguous() and v.shape == (T, M)
    assert beta_all.numel() == (K + 1) * T * M and (K == 0 or fz.numel() == T * M)
    out = torch.empty(3 * (H + 1) + 1, dtype=_RD, device=dev)
    tix = (C.c_int * max(1, K))(*[int(i) for i in trace_idx])
    with _on(dev):
        check(lib().efgp_gradient_assemble(dev.index, M, T, H, -1 if variance_idx is None else int(variance_idx), K, tix,
                                           _ptr(fy), _ptr(tg), _ptr(ws), _ptr(beta), _ptr(dprime) if H else None,
                                           _ptr(fz) if K else None, _ptr(v), _ptr(beta_all), float(sigmasq), float(n_obs), float(yy),
                                           float(variance), _ptr(out), _stream(dev)), "efgp_gradient_assemble")
    return out


def _start_vector(x0, bb, op, dev):
    """The solver's in/out buffer: a copy of x0, or zeros when x0 is None (one fill instead of a fill and a copy)."""
    if x0 is None:
        return torch.zeros(bb.shape, dtype=_CD, device=dev)
    return _dc(x0.reshape(-1, op.size), dev, _CD).clone()


def cg_solve(op, ws, sigmasq, variant, b, x0, tol, max_iter=None, early_stop=True, diag=None, batched=None, hermitian=False):
    """Fused device CG on ws*T(ws*.) (+sigma^2 | /sigma^2 + 1).  Returns (x, iters, row_iters).
    hermitian=True: see cg_solve_async (the systems are transforms of real data; grids without a specialised kernel run
    the general solver)."""
    dev = op.dev
    if batched is None:
        batched = b.ndim > 1
    if hermitian:
        res = cg_solve_async(op, ws, sigmasq, variant, b, x0, tol, max_iter=max_iter, early_stop=early_stop, diag=diag,
                             batched=batched, hermitian=True)
        if res is not None:
            try:
                return res[0], int(res[1]), list(res[1].rows)
            except RuntimeError as err:
                # a cooperative launch whose grid barrier died (-3: its systems hold NaN): this entry is the one that retries --
                # fall through to the synchronous solver, which re-solves dead systems through the multi-launch iteration
                if "cooperative CG" not in str(err):
                    raise
    bb = _dc(b.reshape(-1, op.size), dev, _CD)
    x = _start_vector(x0, bb, op, dev)
    wsd = _dc(ws, dev, _CD)
    dg = _dc(diag, dev, _RD) if diag is not None else None
    B = bb.shape[0]
    iters = C.c_int(0)
    rows = (C.c_int * B)()
    with _on(dev):
        # hermitian: the synchronous entry keeps the promise too (3-D grids carry the planes k0 >= 0 only)
        fn = lib().efgp_cg_solve_hermitian if hermitian else lib().efgp_cg_solve
        check(fn(op._h, _ptr(wsd), float(sigmasq), int(variant), _ptr(dg) if dg is not None else None,
                 _ptr(bb), _ptr(x), B, float(tol), int(max_iter) if max_iter is not None else 0,
                 int(bool(early_stop)), int(bool(batched)), C.byref(iters), rows, _stream(dev)),
              "efgp_cg_solve")
    return x.reshape(b.shape), int(iters.value), [int(r) for r in rows]


class LazyIterations:
    """Iteration counts of an asynchronous CG solve; reading them waits for the solve (device tensor -> host)."""

    def __init__(self, rows_dev, batched, max_iter):
        self._rows_dev = rows_dev
        self._batched = batched
        self._max_iter = max_iter
        self._rows = None

    @property
    def rows(self):
        if self._rows is None:
            self._rows = [int(v) for v in self._rows_dev.tolist()]
            if any(v == -3 for v in self._rows):
                raise RuntimeError("efgp_hip: a cooperative CG launch could not get its workgroups resident together (another "
                                   "kernel held the CUs); the affected systems were not solved and hold NaN -- efgp_hip.cg_solve "
                                   "retries them through the multi-launch iteration; EFGP_NO_CG_COOP=1 avoids the cooperative path")
            if any(v == -2 for v in self._rows):
                raise RuntimeError("efgp_hip: a system given to the Hermitian CG kernel is not the transform of real data "
                                   "(right-hand side not conjugate-even, or ws not real and even); its solution is NaN")
        return self._rows

    def __int__(self):
        mx = max(self.rows)
        # cg.py:193-199,243: the batched loop counts the terminating pass too
        return mx + 1 if (self._batched and mx < self._max_iter) else mx

    __index__ = __int__

    def __repr__(self):
        return str(int(self))

    # number-like: callers of the reference's API compare, add and format iteration counts
    def __eq__(self, other):
        return int(self) == other

    def __lt__(self, other):
        return int(self) < other

    def __le__(self, other):
        return int(self) <= other

    def __gt__(self, other):
        return int(self) > other

    def __ge__(self, other):
        return int(self) >= other

    def __hash__(self):
        return hash(int(self))

    def __add__(self, other):
        return int(self) + other

    __radd__ = __add__

    def __format__(self, spec):
        return format(int(self), spec)


def cg_solve_async(op, ws, sigmasq, variant, b, x0, tol, max_iter=None, early_stop=True, diag=None, batched=None,
                   hermitian=False):
    """Like cg_solve but without host synchronisation: returns (x, LazyIterations) or None when the operator's
    grid does not fit the persistent kernel (the caller then uses cg_solve).  hermitian=True: b, x0 are Fourier
    coefficients of real functions and ws is real and even (efgp_cg_solve_hermitian_async; refused with an error when
    the data say otherwise)."""
    from .lib import EFGP_EUNSUPPORTED
    dev = op.dev
    if batched is None:
        batched = b.ndim > 1
    bb = _dc(b.reshape(-1, op.size), dev, _CD)
    wsd = _dc(ws, dev, _CD)
    dg = _dc(diag, dev, _RD) if diag is not None else None
    B = bb.shape[0]
    mi = int(max_iter) if max_iter is not None else 2 * op.size
    rows_dev = torch.empty(B, dtype=torch.int32, device=dev)
    if x0 is None:
        # from zero: the kernels start from x = 0 themselves (no fill launch, no initial operator application)
        x = torch.empty(bb.shape, dtype=_CD, device=dev)
        with _on(dev):
            rc = lib().efgp_cg_solve_from_zero_async(op._h, _ptr(wsd), float(sigmasq), int(variant), _ptr(dg) if dg is not None else None,
                                                     _ptr(bb), _ptr(x), B, float(tol), mi, int(bool(early_stop)), int(bool(batched)),
                                                     int(bool(hermitian)), _ptr(rows_dev), _stream(dev))
    else:
        x = _start_vector(x0, bb, op, dev)
        with _on(dev):
            fn = lib().efgp_cg_solve_hermitian_async if hermitian else lib().efgp_cg_solve_async
            rc = fn(op._h, _ptr(wsd), float(sigmasq), int(variant), _ptr(dg) if dg is not None else None, _ptr(bb), _ptr(x), B,
                    float(tol), mi, int(bool(early_stop)), int(bool(batched)), _ptr(rows_dev), _stream(dev))
    if rc == EFGP_EUNSUPPORTED:
        return None
    check(rc, "efgp_cg_solve_async")
    # keep the operands alive until the stream has consumed them: torch's caching allocator only reuses a block
    # for work enqueued later on the same stream, so dropping the Python references here is safe
    return x.reshape(b.shape), LazyIterations(rows_dev, bool(batched), mi)


def cg_solve_mean_async(op, ws, sigmasq, diag_scale, fy, tol, max_iter=None, early_stop=True):
    """The fit's mean system (D T D + sigmasq I) beta = D fy from beta_0 = 0 in ONE launch and without host
    synchronisation: the right-hand side ws*fy, the Jacobi diagonal diag_scale*|ws|^2 + sigmasq and the zero start
    are formed inside the kernel.  diag_scale: 0-dim float64 device tensor (may be a view, e.g. v[centre].real) or
    None for no preconditioner.  Returns (beta, LazyIterations) or None when the grid does not fit the kernel."""
    from .lib import EFGP_EUNSUPPORTED
    dev = op.dev
    ff = _dc(fy.reshape(-1), dev, _CD)
    if ff.numel() != op.size:
        raise ValueError(f"fy has {ff.numel()} entries, the operator has {op.size}")
    wsd = _dc(ws.reshape(-1), dev, _CD)
    x = torch.empty(op.size, dtype=_CD, device=dev)
    ds = None
    if diag_scale is not None:
        ds = diag_scale.to(device=dev, dtype=_RD)        # no copy for a float64 view on the device
        if ds.numel() != 1:
            raise ValueError("diag_scale must hold one value")
    mi = int(max_iter) if max_iter is not None else 2 * op.size
    rows_dev = torch.empty(1, dtype=torch.int32, device=dev)
    with _on(dev):
        rc = lib().efgp_cg_solve_mean_async(op._h, _ptr(wsd), float(sigmasq), _ptr(ds) if ds is not None else None, _ptr(ff),
                                            _ptr(x), float(tol), mi, int(bool(early_stop)), _ptr(rows_dev), _stream(dev))
    if rc == EFGP_EUNSUPPORTED:
        return None
    check(rc, "efgp_cg_solve_mean_async")
    return x.reshape(fy.shape), LazyIterations(rows_dev, False, mi)


def lanczos(op, ws, sigmasq, variant, z, steps):
    """`steps` Lanczos steps on A (variant 0: ws*T(ws*.) + sigmasq, 1: /sigmasq + 1) from every row of z (P, M), all inside
    one launch (efgp_lanczos).  Returns (alpha (P,steps), beta (P,steps), |z|^2 (P,), steps_taken (P,) int32) as DEVICE
    tensors -- nothing is read back -- or None when the grid does not fit the persistent kernel."""
    from .lib import EFGP_EUNSUPPORTED
    dev = op.dev
    zz = _dc(z.reshape(-1, op.size), dev, _CD)
    P = zz.shape[0]
    wsd = _dc(ws, dev, _CD)
    alpha = torch.zeros((P, int(steps)), dtype=_RD, device=dev)
    beta = torch.zeros((P, int(steps)), dtype=_RD, device=dev)
    norm2 = torch.empty(P, dtype=_RD, device=dev)
    taken = torch.empty(P, dtype=torch.int32, device=dev)
    with _on(dev):
        rc = lib().efgp_lanczos(op._h, _ptr(wsd), float(sigmasq), int(variant), _ptr(zz), P, int(steps), _ptr(alpha), _ptr(beta),
                                _ptr(norm2), _ptr(taken), _stream(dev))
    if rc == EFGP_EUNSUPPORTED:
        return None
    check(rc, "efgp_lanczos")
    return alpha, beta, norm2, taken


def lag_sums(gamma, eta, mtot, dim):
    """c[r] = mean_j sum_{k-l=r} gamma[j,k] eta[j,l] on the (2 mtot - 1)^d lag box in FFT order (efgp_lag_sums)."""
    dev = gamma.device
    M = int(mtot) ** int(dim)
    gg = gamma.reshape(-1, M).to(_CD).contiguous()
    ee = _dc(eta.reshape(-1, M), dev, _RD)
    out = torch.empty((2 * int(mtot) - 1,) * int(dim), dtype=_CD, device=dev)
    with _on(dev):
        check(lib().efgp_lag_sums(dev.index, int(dim), int(mtot), _ptr(gg), _ptr(ee), gg.shape[0], _ptr(out), _stream(dev)),
              "efgp_lag_sums")
    return out


def variance_rhs(x_new, h, mtot, ws):
    """rhs[b, k] = ws[k] conj(f_k(x*_b)) for the 'regular' variance solves (efgp_variance_rhs)."""
    dev = ws.device
    xn = _dc(x_new, dev, _RD)
    B, d = xn.shape
    wsd = ws.to(_CD).contiguous()
    out = torch.empty((B, wsd.numel()), dtype=_CD, device=dev)
    with _on(dev):
        check(lib().efgp_variance_rhs(dev.index, d, int(mtot), float(h), _ptr(xn), B, _ptr(wsd), _ptr(out), _stream(dev)),
              "efgp_variance_rhs")
    return out


def variance_contract(x_new, h, mtot, ws, gamma):
    """s^2[b] = max(0, Re sum_k f_k(x*_b) ws[k] gamma[b, k]) (efgp_variance_contract)."""
    dev = ws.device
    xn = _dc(x_new, dev, _RD)
    B, d = xn.shape
    wsd = ws.to(_CD).contiguous()
    gg = gamma.reshape(B, -1).to(_CD).contiguous()
    out = torch.empty(B, dtype=_RD, device=dev)
    with _on(dev):
        check(lib().efgp_variance_contract(dev.index, d, int(mtot), float(h), _ptr(xn), B, _ptr(wsd), _ptr(gg), _ptr(out),
                                           _stream(dev)), "efgp_variance_contract")
    return out


class RcclComm:
    """Sum / min / max all-reduce and broadcast over RCCL without a torch process group (C ABI: efgp_comm_*).
    `unique_id` (128 bytes) comes from `RcclComm.make_id()` on rank 0 and reaches the other ranks by the caller's channel."""

    @staticmethod
    def make_id():
        buf = (C.c_char * 128)()
        check(lib().efgp_comm_unique_id(buf), "efgp_comm_unique_id")
        return bytes(buf)

    def __init__(self, dev, rank, world, unique_id):
        assert len(unique_id) == 128
        self.dev, self.rank, self.world = dev, int(rank), int(world)
        self._h = C.c_void_p()
        with _on(dev):
            check(lib().efgp_comm_init(C.byref(self._h), dev.index, self.rank, self.world, C.c_char_p(unique_id)), "efgp_comm_init")

    def all_reduce_sum_(self, t):
        buf = torch.view_as_real(t) if t.is_complex() else t
        assert buf.is_cuda and buf.dtype == _RD and buf.is_contiguous()
        with _on(self.dev):
            check(lib().efgp_comm_allreduce_sum(self._h, _ptr(buf), buf.numel(), _stream(self.dev)), "efgp_comm_allreduce_sum")
        return t

    def all_reduce_minmax_(self, t, take_max):
        assert t.is_cuda and t.dtype == _RD and t.is_contiguous()
        with _on(self.dev):
            check(lib().efgp_comm_allreduce_minmax(self._h, _ptr(t), t.numel(), int(bool(take_max)), _stream(self.dev)),
                  "efgp_comm_allreduce_minmax")
        return t

    def broadcast_(self, t, root=0):
        assert t.is_cuda and t.is_contiguous()
        with _on(self.dev):
            check(lib().efgp_comm_broadcast(self._h, _ptr(t), t.numel() * t.element_size(), int(root), _stream(self.dev)),
                  "efgp_comm_broadcast")
        return t

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().efgp_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def vdot_real(a, b):
    """Re <a, b> = Re sum conj(a) b for real or complex device vectors, reduced by the HIP kernel."""
    dev = a.device
    aa = a.reshape(-1).to(_CD if a.is_complex() else _RD).contiguous()
    bb = b.reshape(-1).to(device=dev, dtype=_CD if b.is_complex() else _RD).contiguous()
    out = C.c_double(0.0)
    with _on(dev):
        check(lib().efgp_vdot_real(dev.index, _ptr(aa), int(aa.is_complex()), _ptr(bb), int(bb.is_complex()),
                                   aa.numel(), C.byref(out), _stream(dev)), "efgp_vdot_real")
    return float(out.value)


def gradient_step(xd, yd, points, *, h, mtot, kconst, lengthscale, variance, sigmasq, tol_pair, tol_probe, cg_tol, early_stop,
                  nprobes, probe_seed, v_seed, use_mean_pc, use_trace_pc, variance_idx, trace_idx, beta0, n_obs, yy):
    """One hyper-gradient step of the adjoint estimator in ONE library call (efgp_gradient_step; csrc/gradient_step.cpp).
    kconst = (kind, nu, c0) of utils.kernels.kernel_constants.  Returns (out_vec, beta, mean LazyIterations, trace
    LazyIterations) -- all device-resident, nothing read back -- or None when the grid's solves are not single launches."""
    from .lib import EFGP_EUNSUPPORTED
    dev = xd.device
    npts, d = xd.shape
    M = int(mtot) ** d
    K = len(trace_idx)
    R = (K + 1) * int(nprobes)
    out = torch.empty(3 * 3 + 1, dtype=_RD, device=dev)
    beta = torch.empty(M, dtype=_CD, device=dev)
    its = torch.empty(1 + R, dtype=torch.int32, device=dev)
    tix = (C.c_int * max(1, K))(*[int(i) for i in trace_idx])
    b0 = _dc(beta0.reshape(-1), dev, _CD) if beta0 is not None else None
    with _on(dev):
        rc = lib().efgp_gradient_step(points._h if points is not None else None, dev.index, int(d), int(npts), _ptr(xd), _ptr(yd),
                                      float(h), int(mtot), int(kconst[0]), float(kconst[1]), float(lengthscale), float(variance),
                                      float(kconst[2]), float(sigmasq), float(tol_pair), float(tol_probe), float(cg_tol),
                                      int(bool(early_stop)), int(nprobes), int(probe_seed) & (2 ** 64 - 1), int(v_seed) & (2 ** 64 - 1),
                                      int(bool(use_mean_pc)), int(bool(use_trace_pc)), -1 if variance_idx is None else int(variance_idx),
                                      K, tix, _ptr(b0) if b0 is not None else None, float(n_obs), float(yy), _ptr(beta), _ptr(out),
                                      _ptr(its), _ptr(its[1:]), _stream(dev))
    if rc == EFGP_EUNSUPPORTED:
        return None
    check(rc, "efgp_gradient_step")
    return out, beta, LazyIterations(its[:1], False, 2 * M), LazyIterations(its[1:], True, 2 * M)


def rademacher_fill(dev, seed, nbatch, npts, index_offset=0):
    """The +-1 probes `NufftPlan.type1_rademacher` uses, materialised as a (nbatch, npts) float64 tensor."""
    out = torch.empty((int(nbatch), int(npts)), dtype=_RD, device=dev)
    with _on(dev):
        check(lib().efgp_rademacher_fill(dev.index, int(seed) & (2 ** 64 - 1), int(index_offset), int(nbatch), int(npts),
                                         _ptr(out), _stream(dev)), "efgp_rademacher_fill")
    return out


def kernel_timing(enable, only=None):
    """HIP-event timers around the library's main launches; `only`: record the launches of this name alone (every timed
    launch puts two event records into the stream)."""
    check(lib().efgp_kernel_timing_only(only.encode() if only else None), "efgp_kernel_timing_only")
    check(lib().efgp_kernel_timing(int(bool(enable))), "efgp_kernel_timing")


def kernel_timing_read(name):
    """-> (total milliseconds, launches) of the named kernel since kernel_timing(True)."""
    ms = C.c_double(0.0)
    n = C.c_int64(0)
    check(lib().efgp_kernel_timing_read(name.encode(), C.byref(ms), C.byref(n)), "efgp_kernel_timing_read")
    return float(ms.value), int(n.value)


class cg_residual_history:
    """Context manager: records row 0's relative residuals |r_i| / |b| of the CG solves enqueued inside it
    (efgp_cg_record_history).  `.values()` -> 1-D float64 tensor of the recorded iterations (zeros beyond the last)."""

    def __init__(self, dev, capacity=4096):
        self.buf = torch.zeros(int(capacity), dtype=_RD, device=dev)

    def __enter__(self):
        check(lib().efgp_cg_record_history(_ptr(self.buf), self.buf.numel()), "efgp_cg_record_history")
        return self

    def __exit__(self, *exc):
        torch.cuda.synchronize(self.buf.device)
        check(lib().efgp_cg_record_history(None, 0), "efgp_cg_record_history")
        return False

    def values(self):
        return self.buf.detach().cpu()
