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
SOURCES = ["es_kernel.cpp", "grid_host.cpp", "common.cpp", "nufft.hip", "spread_mfma.hip", "points_layout.hip", "small_dft.hip", "variance_ops.hip", "gradient_ops.hip", "comm.cpp", "toeplitz_cg.hip", "cg_persistent.hip"]
HEADERS = ["es_kernel.hpp", "common.hpp", "toeplitz_cg.hpp", "nufft_dev.hpp", "points_layout.hpp", "spread_mfma.hpp", "small_dft.hpp", os.path.join("..", "..", "include", "efgp_hip.h")]
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
    """Every source is compiled to its own object (in parallel, skipped when the object is newer than the source and
    all headers), then linked: a one-file edit costs one compile instead of the whole library."""
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tl = _torch_lib_dir()
    tag = "stamps" if extra else "obj"
    objdir = os.path.join(HERE, "build", tag)
    os.makedirs(objdir, exist_ok=True)
    hdr_time = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS if os.path.exists(os.path.join(CSRC, h)))
    flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-I/opt/rocm/include"] + extra

    def compile_one(src):
        sp = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src + ".o")
        if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(sp), hdr_time):
            return obj
        cmd = [hipcc] + flags + ["-x", "hip", "-c", sp, "-o", obj]
        if verbose:
            print("[efgp_hip] " + " ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), int(os.environ.get("EFGP_BUILD_JOBS", "6")))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-no-hip-rt"] + objs + [
        "-L" + tl, "-lhipfft", "-lamdhip64", "-lrccl", "-Wl,-rpath," + tl, "-o", target + ".tmp"]
    if verbose:
        print("[efgp_hip] " + " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(target + ".tmp", target)
    return target


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, stamps="--stamps" in sys.argv))
