"""A device context of the library is single-stream at any moment (shared scratch, pooled blocks, caches): a caller
that switches streams is ordered behind the work queued on the previous one (common.cpp::stream_handover) instead of racing with
it.  Reference behaviour: one stream (the reference is single-stream Python); here two torch streams alternate on two models that
share every scratch buffer of the device, and the results must be those of the single-stream run."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(seed, N):
    from efgpnd import EFGPND
    from kernels.squared_exponential import SquaredExponential
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(N, 2, generator=g, dtype=torch.float64).cuda()
    y = (torch.sin(4 * x[:, 0]) * torch.cos(3 * x[:, 1]) + 0.2 * torch.randn(N, generator=g, dtype=torch.float64).cuda()).contiguous()
    kern = SquaredExponential(dimension=2, init_lengthscale=0.15 + 0.05 * seed, init_variance=1.0)
    return EFGPND(x, y, kern, sigmasq=0.1, eps=1e-4, estimate_params=False, opts={"mean_cg_warm_start": False}), x


def test_two_streams_alternating_equal_one_stream():
    # A's pass over 2e7 points runs for ~0.5 ms behind a host that has long moved on: B's launches on the other stream land in
    # the middle of it (with EFGP_NO_STREAM_HANDOVER=1 this test fails: both spread into the same accumulator)
    ma, xa = _model(1, 20_000_000)
    mb, xb = _model(2, 50_000)
    ref = []
    for m, x in ((ma, xa), (mb, xb)):
        m.fit()
        ref.append((m._beta.clone(), m.predict(x[:5000], return_variance=False)[0].clone()))
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    out = [None, None]
    for rep in range(6):
        # no synchronisation between the two: without the hand-over the second fit's pair pass would overwrite the accumulator,
        # the fine grid and the pooled blocks the first one's kernels are still reading
        with torch.cuda.stream(s1):
            ma.fit()
            pa = ma.predict(xa[:5000], return_variance=False)[0]
        with torch.cuda.stream(s2):
            mb.fit()
            pb = mb.predict(xb[:5000], return_variance=False)[0]
        out = [(ma._beta, pa), (mb._beta, pb)]
    s1.synchronize()
    s2.synchronize()
    for (b0, p0), (b1, p1) in zip(ref, out):
        assert float((b1 - b0).abs().max()) <= 1e-9 * float(b0.abs().max())
        assert float((p1 - p0).abs().max()) <= 1e-9 * float(p0.abs().max())


def test_gradient_steps_on_alternating_streams():
    ma, _ = _model(3, 100_000)
    for _ in range(2):                      # the model builds its point layout at its second pass: the reference run comes after
        ma.compute_gradients(trace_samples=4, probe_seed=7, cg_tol=1e-10)
    torch.manual_seed(0)
    ma._last_gradient_beta = None
    g_ref = ma.compute_gradients(trace_samples=4, probe_seed=7, cg_tol=1e-10).clone()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for rep in range(4):
        with torch.cuda.stream(s1 if rep % 2 == 0 else s2):
            torch.manual_seed(0)
            ma._last_gradient_beta = None
            outs.append(ma.compute_gradients(trace_samples=4, probe_seed=7, cg_tol=1e-10).clone())
    torch.cuda.synchronize()
    for g in outs:
        assert float((g - g_ref).abs().max()) <= 1e-6 * float(g_ref.abs().max())


def test_two_host_threads_on_one_device():
    """ctypes releases the GIL inside every library call: two Python threads that fit their own models on one device are two
    host threads inside the library at once.  A device's context belongs to one entry point at a time (the per-device lock of
    DeviceGuard) and follows the callers from stream to stream: the results are those of the threads run one after the other."""
    import threading
    ma, xa = _model(4, 300_000)
    mb, xb = _model(5, 120_000)
    ref = []
    for m, x in ((ma, xa), (mb, xb)):
        for _ in range(3):                 # (the point layout is built at a model's second pass: the reference fit comes after)
            m.fit()
        ref.append((m._beta.clone(), m.predict(x[:4000], return_variance=False)[0].clone()))
    torch.cuda.synchronize()
    out, errs = {}, []

    def work(key, m, x, stream):
        try:
            with torch.cuda.stream(stream):
                for _ in range(25):
                    m.fit()
                    p = m.predict(x[:4000], return_variance=False)[0]
                out[key] = (m._beta.clone(), p.clone())
            stream.synchronize()
        except Exception as err:          # noqa: BLE001
            errs.append(err)

    ts = [threading.Thread(target=work, args=(0, ma, xa, torch.cuda.Stream())), threading.Thread(target=work, args=(1, mb, xb, torch.cuda.Stream()))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for i in (0, 1):
        assert float((out[i][0] - ref[i][0]).abs().max()) <= 1e-9 * float(ref[i][0].abs().max())
        assert float((out[i][1] - ref[i][1]).abs().max()) <= 1e-9 * float(ref[i][1].abs().max())
