"""The in-house FFT (csrc/line_fft.hip, no run-time compilation) against torch.fft on the sizes the EFGP path produces: fine grids
2^a 3^b 5^c of the NUFFT (es_fine_size) and the power-of-two circulant grids of ToeplitzND (efgpnd.py:1275-1290), rank 1..3,
batched, both directions.  Tolerance: 5e-15 * sqrt(log2 n) relative (double-precision FFT rounding), written below."""
import ctypes as C
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _fft(x, rank, forward, rocfft=False):
    from efgp_hip import lib
    from efgp_hip.lib import check
    y = x.clone().contiguous()
    n = (C.c_longlong * rank)(*y.shape[-rank:])
    batch = y.numel() // math.prod(y.shape[-rank:])
    check(lib().efgp_fft_c2c(y.device.index or 0, rank, n, batch, C.c_void_p(y.data_ptr()), int(forward), int(rocfft),
                             C.c_void_p(torch.cuda.current_stream().cuda_stream)), "efgp_fft_c2c")
    return y


CASES = [(1, (384,), 3), (1, (4096,), 2), (1, (1,), 1), (1, (2,), 5), (1, (3000,), 1), (1, (1215,), 2), (1, (250,), 4),
         (2, (72, 180), 2), (2, (512, 256), 1), (2, (5, 3), 7), (2, (1024, 6), 1),
         (3, (48, 48, 48), 2), (3, (64, 128, 64), 1), (3, (96, 30, 75), 3), (3, (240, 120, 16), 1), (3, (2, 3, 5), 4)]


@pytest.mark.parametrize("rank,shape,batch", CASES)
@pytest.mark.parametrize("forward", [True, False])
def test_matches_torch_fft(rank, shape, batch, forward):
    g = torch.Generator().manual_seed(sum(shape) + batch)
    x = torch.complex(torch.randn(batch, *shape, generator=g, dtype=torch.float64), torch.randn(batch, *shape, generator=g, dtype=torch.float64)).cuda()
    dims = tuple(range(-rank, 0))
    want = torch.fft.fftn(x, dim=dims) if forward else torch.fft.ifftn(x, dim=dims, norm="forward")
    got = _fft(x, rank, forward)
    tol = 5e-15 * max(1.0, math.sqrt(math.log2(max(2, math.prod(shape)))))
    err = float(torch.linalg.norm((got - want).reshape(-1)) / torch.linalg.norm(want.reshape(-1)))
    assert err < tol, (shape, forward, err, tol)


def test_unsupported_sizes_are_refused_and_the_library_route_still_works():
    from efgp_hip.lib import EFGP_EUNSUPPORTED, lib
    x = torch.zeros(2, 14, dtype=torch.complex128, device="cuda")                 # 14 = 2 * 7
    n = (C.c_longlong * 1)(14)
    rc = lib().efgp_fft_c2c(0, 1, n, 2, C.c_void_p(x.data_ptr()), 1, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == EFGP_EUNSUPPORTED
    g = torch.Generator().manual_seed(5)
    y = torch.complex(torch.randn(3, 14, generator=g, dtype=torch.float64), torch.randn(3, 14, generator=g, dtype=torch.float64)).cuda()
    got = _fft(y, 1, True, rocfft=True)
    assert float(torch.linalg.norm(got - torch.fft.fft(y, dim=-1)) / torch.linalg.norm(got)) < 1e-14
