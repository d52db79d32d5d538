"""Which random transforms exceed 2 x their requested tolerance?  Runs tools/fuzz_nufft.py's generator over many seeds and lists
every case whose worst error is above 1.5 x tol, with the plan's fine grid and window width."""
import os
import re
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (".", "tools", "gp-quadrature_amd"):
    sys.path.insert(0, os.path.join(R, p))
import fuzz_nufft  # noqa: E402
from efgp_hip.lib import lib  # noqa: E402

L = lib()
seeds = range(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 20)
import io, contextlib
hist = {}
for seed in seeds:
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        # verbose prints every 10th case only; patch: run with a wrapper that prints all
        fuzz_nufft.run.__globals__["_ALL"] = True
        worst, fails = fuzz_nufft.run(100, seed, verbose=True)
    for line in buf.getvalue().splitlines():
        m = re.match(r"case\s+(\d+) d=(\d) nm=(\d+) tol=(\S+) N=(\d+) .* cplx=(\w+) real_only=(\w+): type1 (\S+) type2 (\S+) adjoint (\S+)", line)
        if not m:
            continue
        d, nm, tol = int(m.group(2)), int(m.group(3)), float(m.group(4))
        e = [float(m.group(8)), float(m.group(9)), float(m.group(10))]
        ratio = max(e) / tol
        N = int(m.group(5))
        nf = L.efgp_fine_grid_size_nd(nm, tol, d, 1 if (d == 2 and N >= 4_000_000) else 0)
        w = L.efgp_window_width_nd(tol, nf / nm, d)
        key = (d, w)
        hist.setdefault(key, []).append(ratio)
        if ratio > float(os.environ.get("FUZZ_MIN_RATIO", "1.5")):
            print(f"seed {seed} {line.strip()}  | nf={nf} sigma={nf / nm:.2f} w={w} ratio={ratio:.2f}")
    print(f"seed {seed}: worst {worst:.2f}", flush=True)
print("\n(d, w): cases, max ratio, share above 2")
for key in sorted(hist):
    v = hist[key]
    print(key, len(v), f"{max(v):.2f}", f"{sum(r > 2 for r in v) / len(v):.2f}")
