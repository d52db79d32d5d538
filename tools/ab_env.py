#!/usr/bin/env python3
"""A/B of one environment switch on the bench's timed region, same box, alternating runs.
usage: ab_env.py VAR [repeats] [bench args...]   e.g.  ab_env.py EFGP_NO_GRID_TO_MODES 3 --steps 200 --warmup 20"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
var = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
extra = sys.argv[3:] or ["--steps", "200", "--warmup", "20"]
res = {"unset": [], "set": []}
for _ in range(reps):
    for label in ("unset", "set"):
        env = dict(os.environ)
        env.pop(var, None)
        if label == "set":
            env[var] = "1"
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--main-only"] + extra, env=env, capture_output=True, text=True).stdout
        line = [l for l in out.splitlines() if l.startswith("{")][-1]
        res[label].append(json.loads(line)["ms_per_step"])
for label, v in res.items():
    print(f"{var} {label:5s}: ms_per_step {['%.4f' % t for t in v]}  best {min(v):.4f}")
