"""Loader and signatures for libefgp_hip.so."""
import ctypes as C
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

EFGP_EINVAL = -1


class EfgpError(RuntimeError):
    pass


def library_path():
    # EFGP_HIP_LIBRARY lets the diagnostic tools under tools/ load an instrumented build
    return os.environ.get("EFGP_HIP_LIBRARY") or os.path.join(HERE, "libefgp_hip.so")


def header_path():
    return os.path.normpath(os.path.join(HERE, "..", "..", "include", "efgp_hip.h"))


def declared_symbols():
    """Names of every function declared in include/efgp_hip.h."""
    txt = open(header_path()).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(efgp_[a-z0-9_]+)\s*\(", txt)))


_VP, _I, _I64, _D = C.c_void_p, C.c_int, C.c_int64, C.c_double
_PI64 = C.POINTER(C.c_int64)

_SIGNATURES = {
    "efgp_version": (_I, []),
    "efgp_last_error": (C.c_char_p, []),
    "efgp_release_workspaces": (_I, [_I]),
    "efgp_kernel_timing": (_I, [_I]),
    "efgp_kernel_timing_only": (_I, [C.c_char_p]),
    "efgp_kernel_timing_read": (_I, [C.c_char_p, C.POINTER(_D), _PI64]),
    "efgp_window_width": (_I, [_D, _D]),
    "efgp_window_eval": (_I, [_D, _D, _D, _PI64, C.POINTER(_D), C.POINTER(_I), C.POINTER(_D)]),
    "efgp_window_width_nd": (_I, [_D, _D, _I]),
    "efgp_fine_grid_size": (_I64, [_I64, _D]),
    "efgp_fine_grid_size_nd": (_I64, [_I64, _D, _I, _I]),
    "efgp_window_deconv": (_I, [_D, _I64, _I64, C.POINTER(_D)]),
    "efgp_nufft_create": (_I, [C.POINTER(_VP), _I, _I, _I64, _VP, C.POINTER(_D), _D, _D]),
    "efgp_nufft_destroy": (_I, [_VP]),
    "efgp_points_create": (_I, [C.POINTER(_VP), _I, _I, _I64, _VP, _VP]),
    "efgp_points_destroy": (_I, [_VP]),
    "efgp_points_bounds": (_I, [_VP, C.POINTER(_D), C.POINTER(_D)]),
    "efgp_points_attach_values": (_I, [_VP, _VP, _VP]),
    "efgp_nufft_create_on": (_I, [C.POINTER(_VP), _VP, C.POINTER(_D), _D, _D]),
    "efgp_nufft_type1": (_I, [_VP, _VP, _I, _I, _PI64, _I, _I, _VP, _VP]),
    "efgp_nufft_type1_rademacher": (_I, [_VP, C.c_uint64, _I64, _I, _PI64, _I, _VP, _VP]),
    "efgp_rademacher_fill": (_I, [_I, C.c_uint64, _I64, _I, _I64, _VP, _VP]),
    "efgp_nufft_type1_pair": (_I, [_VP, _VP, _PI64, _VP, _PI64, _VP, _VP]),
    "efgp_nufft_type2": (_I, [_VP, _VP, _I, _PI64, _I, _I, _VP, _I, _VP]),
    "efgp_nufft_type2_scaled": (_I, [_VP, _VP, _VP, _I, _PI64, _I, _I, _VP, _I, _VP]),
    "efgp_toeplitz_create": (_I, [C.POINTER(_VP), _I, _I, _PI64, _VP, _I, _VP]),
    "efgp_toeplitz_destroy": (_I, [_VP]),
    "efgp_toeplitz_apply": (_I, [_VP, _VP, _I, _VP, _VP]),
    "efgp_toeplitz_apply_scaled": (_I, [_VP, _VP, _I, _I, _VP, _VP, _VP, _VP]),
    "efgp_toeplitz_fft_shape": (_I, [_VP, _PI64]),
    "efgp_toeplitz_cg_shape": (_I, [_VP, _I, _PI64]),
    "efgp_toeplitz_single_launch_solves": (_I, [_VP]),
    "efgp_cg_solve": (_I, [_VP, _VP, _D, _I, _VP, _VP, _VP, _I, _D, _I, _I, _I, C.POINTER(_I), C.POINTER(_I), _VP]),
    "efgp_fft_c2c": (_I, [_I, _I, C.POINTER(C.c_longlong), C.c_longlong, _VP, _I, _I, _VP]),
    "efgp_cg_solve_hermitian": (_I, [_VP, _VP, _D, _I, _VP, _VP, _VP, _I, _D, _I, _I, _I, C.POINTER(_I), C.POINTER(_I), _VP]),
    "efgp_cg_solve_async": (_I, [_VP, _VP, _D, _I, _VP, _VP, _VP, _I, _D, _I, _I, _I, _VP, _VP]),
    "efgp_grid_bounds": (_I, [_I, _I, _D, _D, _D, _D, _D, _D, _D, _VP, _VP]),
    "efgp_spectral_weights_host": (_I, [_I, _I, _D, _D, _D, _D, _D, _I, _VP, _VP]),
    "efgp_spectral_weights": (_I, [_I, _I, _I, _D, _D, _D, _D, _D, _I, _VP, _VP, _VP]),
    "efgp_gradient_prepare": (_I, [_I, _I64, _VP, _VP, _VP, _D, _VP, _VP, _VP]),
    "efgp_gradient_assemble": (_I, [_I, _I64, _I, _I, _I, _I, C.POINTER(_I), _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _D, _D, _D, _D, _VP, _VP]),
    "efgp_gradient_step": (_I, [_VP, _I, _I, _I64, _VP, _VP, _D, _I, _I, _D, _D, _D, _D, _D, _D, _D, _D, _I, _I, C.c_uint64, C.c_uint64,
                           _I, _I, _I, _I, C.POINTER(_I), _VP, _D, _D, _VP, _VP, _VP, _VP, _VP]),
    "efgp_cg_solve_hermitian_async": (_I, [_VP, _VP, _D, _I, _VP, _VP, _VP, _I, _D, _I, _I, _I, _VP, _VP]),
    "efgp_cg_solve_from_zero_async": (_I, [_VP, _VP, _D, _I, _VP, _VP, _VP, _I, _D, _I, _I, _I, _I, _VP, _VP]),
    "efgp_cg_solve_mean_async": (_I, [_VP, _VP, _D, _VP, _VP, _VP, _D, _I, _I, _VP, _VP]),
    "efgp_cg_record_history": (_I, [_VP, _I]),
    "efgp_lanczos": (_I, [_VP, _VP, _D, _I, _VP, _I, _I, _VP, _VP, _VP, _VP, _VP]),
    "efgp_lag_sums": (_I, [_I, _I, _I64, _VP, _VP, _I, _VP, _VP]),
    "efgp_variance_rhs": (_I, [_I, _I, _I64, _D, _VP, _I64, _VP, _VP, _VP]),
    "efgp_variance_contract": (_I, [_I, _I, _I64, _D, _VP, _I64, _VP, _VP, _VP, _VP]),
    "efgp_comm_unique_id": (_I, [_VP]),
    "efgp_comm_init": (_I, [C.POINTER(_VP), _I, _I, _I, _VP]),
    "efgp_comm_allreduce_sum": (_I, [_VP, _VP, C.c_size_t, _VP]),
    "efgp_comm_allreduce_minmax": (_I, [_VP, _VP, C.c_size_t, _I, _VP]),
    "efgp_comm_broadcast": (_I, [_VP, _VP, C.c_size_t, _I, _VP]),
    "efgp_comm_destroy": (_I, [_VP]),
    "efgp_vdot_real": (_I, [_I, _VP, _I, _VP, _I, _I64, C.POINTER(_D), _VP]),
}


def lib():
    """The loaded library; RuntimeError if it was never built (no fallback exists)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(the EFGP hot path runs only as HIP kernels; there is no CPU fallback)")
    import torch  # noqa: F401  -- loads the HIP runtime / hipFFT this library is linked against
    handle = C.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(handle, name)
        fn.restype = res
        fn.argtypes = args
    _LIB = handle
    return handle


EFGP_EUNSUPPORTED = -4


def check(rc, what=""):
    if rc == 0:
        return
    msg = lib().efgp_last_error().decode("utf-8", "replace")
    if rc == EFGP_EINVAL:
        raise ValueError(f"{what}: {msg}" if what else msg)
    raise EfgpError(f"{what}: {msg} (code {rc})" if what else f"{msg} (code {rc})")
