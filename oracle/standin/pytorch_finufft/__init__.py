"""Stand-in for the third-party `pytorch_finufft` package (TEST INFRASTRUCTURE ONLY).

The reference (danbider/gp-quadrature, efgpnd.py:11) imports
`pytorch_finufft.functional`, which wraps the FINUFFT C++ library.  Neither is
installed in the build container and there is no network.  FINUFFT approximates
the non-uniform discrete Fourier transform to a requested tolerance; this
stand-in evaluates that transform *exactly* (direct summation), so the reference
code can be run unmodified to produce golden vectors (oracle/gen_golden.py).

Never imported by the product path and never shipped to the GPU box as a
dependency of anything but tests.
"""
from . import functional  # noqa: F401
