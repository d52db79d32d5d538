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
SOURCES = ["es_kernel.cpp", "grid_host.cpp", "common.cpp", "nufft.hip", "spread_mfma.hip", "points_layout.hip", "small_dft.hip", "variance_ops.hip", "gradient_ops.hip", "gradient_step.cpp", "comm.cpp", "toeplitz_cg.hip", "cg_persistent.hip", "line_fft.hip"]
HEADERS = ["es_kernel.hpp", "common.hpp", "toeplitz_cg.hpp", "nufft_dev.hpp", "points_layout.hpp", "spread_mfma.hpp", "small_dft.hpp", "line_fft.hpp", os.path.join("..", "..", "include", "efgp_hip.h")]
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


def build(force=False, verbose=True, stamps=False, sanitize=False):
    """stamps=True builds the DIAGNOSTIC library libefgp_hip_stamps.so (in-kernel cycle stamps in the persistent CG, the phase
    switches of the MFMA spreader; never loaded by the product path).
    sanitize=True builds libefgp_hip_asan.so: the HOST side of every source (window design, grid bisections, plan / window / FFT
    plan caches, block pool, argument checks, the C ABI itself) instrumented with AddressSanitizer + UBSan, device code untouched
    (-fno-gpu-sanitize: GPU sanitizers are not available on the pool).  For the build container only: load it through
    EFGP_HIP_LIBRARY with the ASan runtime preloaded -- `python -m efgp_hip.build --sanitize-check` does both and runs the
    CPU-side suites against it."""
    if sanitize:
        return _build(TARGET.replace("libefgp_hip.so", "libefgp_hip_asan.so"),
                      ["-fsanitize=address,undefined", "-fno-gpu-sanitize", "-fno-omit-frame-pointer", "-g"], verbose,
                      tag="asan", opt="-O1", link_extra=["-fsanitize=address,undefined", "-shared-libsan"])
    if stamps:
        return _build(TARGET.replace("libefgp_hip.so", "libefgp_hip_stamps.so"), ["-DEFGP_CG_STAMPS", "-DEFGP_MFMA_DIAG"], verbose)
    if not force and not needs_build():
        return TARGET
    return _build(TARGET, [], verbose)


def asan_runtime():
    """Path of the shared AddressSanitizer runtime of the ROCm clang (to LD_PRELOAD into the Python process)."""
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang", "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True)
    path = out.stdout.strip()
    return path if os.path.isabs(path) and os.path.exists(path) else None


def sanitize_check(report=None):
    """Builds the sanitized library and runs tests/test_cabi.py + tests/test_host_logic.py against it (CPU only: symbol table,
    window design, grid bounds bit-identity sweep, spectral weights, error paths).  Returns the pytest exit code; the combined
    output (ASan / UBSan reports included) goes to `report` when given."""
    lib = build(sanitize=True)
    rt = asan_runtime()
    if rt is None:
        raise RuntimeError("no shared ASan runtime next to the ROCm clang")
    root = os.path.dirname(os.path.dirname(HERE))
    env = dict(os.environ, EFGP_HIP_LIBRARY=lib, LD_PRELOAD=rt,
               ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0:abort_on_error=0:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    cmd = [sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_cabi.py"), os.path.join(root, "tests", "test_host_logic.py"),
           "-q", "-m", "not gpu", "-p", "no:cacheprovider"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True)
    text = out.stdout + out.stderr
    if report:
        with open(report, "w") as fh:
            fh.write("# host-side sanitizer run (build container): " + " ".join(cmd) + "\n# library: " + lib + "\n# LD_PRELOAD=" + rt + "\n")
            fh.write(text)
    print(text[-3000:])
    return out.returncode


def _build(target, extra, verbose, tag=None, opt="-O3", link_extra=()):
    """Every source is compiled to its own object (in parallel, skipped when the object is newer than the source and
    all headers), then linked: a one-file edit costs one compile instead of the whole library."""
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tl = _torch_lib_dir()
    tag = tag or ("stamps" if extra else "obj")
    objdir = os.path.join(HERE, "build", tag)
    os.makedirs(objdir, exist_ok=True)
    hdr_time = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS if os.path.exists(os.path.join(CSRC, h)))
    flags = [opt, "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-I/opt/rocm/include"] + extra

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
    cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-no-hip-rt"] + list(link_extra) + objs + [
        "-L" + tl, "-lhipfft", "-lamdhip64", "-lrccl", "-Wl,-rpath," + tl, "-o", target + ".tmp"]
    if verbose:
        print("[efgp_hip] " + " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(target + ".tmp", target)
    return target


if __name__ == "__main__":
    if "--sanitize-check" in sys.argv:
        rep = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--report=")]
        sys.exit(sanitize_check(rep[0] if rep else None))
    print(build(force="--force" in sys.argv, stamps="--stamps" in sys.argv, sanitize="--sanitize" in sys.argv))
