"""In-tree build of libefgp_hip.so for gfx950 with hipcc.

The library is linked against the HIP runtime and hipFFT that ship inside the installed torch
wheel (torch/lib), so that it shares ONE HIP runtime with the torch tensors whose device
pointers it receives.  `-no-hip-rt` keeps hipcc from adding its own -L/opt/rocm/lib.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
SOURCES = ["es_kernel.cpp", "common.cpp", "nufft.hip", "toeplitz_cg.hip", "cg_persistent.hip"]
HEADERS = ["es_kernel.hpp", "common.hpp", "toeplitz_cg.hpp", os.path.join("..", "..", "include", "efgp_hip.h")]
TARGET = os.path.join(HERE, "libefgp_hip.so")


def _torch_lib_dir():
    import torch
    return os.path.join(os.path.dirname(torch.__file__), "lib")


def needs_build():
    if not os.path.exists(TARGET):
        return True
    t = os.path.getmtime(TARGET)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(p) > t for p in deps if os.path.exists(p))


def build(force=False, verbose=True, stamps=False):
    """stamps=True builds the DIAGNOSTIC library libefgp_hip_stamps.so (in-kernel cycle stamps in the
    persistent CG; never loaded by the product path)."""
    if stamps:
        return _build(TARGET.replace("libefgp_hip.so", "libefgp_hip_stamps.so"), ["-DEFGP_CG_STAMPS"], verbose)
    if not force and not needs_build():
        return TARGET
    return _build(TARGET, [], verbose)


def _build(target, extra, verbose):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tl = _torch_lib_dir()
    cmd = [hipcc, "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-munsafe-fp-atomics",
           "-no-hip-rt"] + extra + ["-x", "hip"] + [os.path.join(CSRC, s) for s in SOURCES] + [
           "-I/opt/rocm/include", "-L" + tl, "-lhipfft", "-lamdhip64", "-Wl,-rpath," + tl, "-o", target + ".tmp"]
    if verbose:
        print("[efgp_hip] " + " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(target + ".tmp", target)
    return target


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, stamps="--stamps" in sys.argv))
