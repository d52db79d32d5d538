"""The one stored output in the reference tree that was produced with the REAL third-party FINUFFT (not the exact-NUDFT stand-in
the other goldens use): experiments/cg_preconditioning_realdata.csv, written by benchmark_cg_preconditioning_realdata.py:295-330
on PRISM usa_temp (N = 4766) for two SE settings ("hard" l = 0.04, "very_hard" l = 0.03; variance 10, sigma^2 = 1e-4, eps = 1e-4,
nufft_eps = 1e-5, CG tol 1e-3).  Committed verbatim as a DATA fixture (tests/golden/data/); the helpers here rebuild the script's
`build_bundle` inputs (:67-80, 83-103) for the CPU (oracle) and GPU (product) twins of the comparison."""
import csv
import os

import torch

from _golden import GOLDEN, load_case

CSV = os.path.join(GOLDEN, "data", "cg_preconditioning_realdata.csv")
REGIMES = {"hard": (0.04, 10.0, 1e-4), "very_hard": (0.03, 10.0, 1e-4)}       # main(): (lengthscale, variance, sigmasq)
EPS, NUFFT_EPS, CG_TOL = 1e-4, 1e-5, 1e-3                                     # build_arg_parser() defaults
PRECS = {"none": None, "diag_ws2": 1.0, "diag_10ws2": 10.0, "diag_100ws2": 100.0, "diag_1000ws2": 1000.0, "diag_Nws2": "N"}
# bands: what another correct NUFFT does to the counts (the reference's code on the exact transform: 111/103/100/100 and
# 124/116/115/114 against the CSV's 111/103/100/99 and 125/113/116/116; the two ill-conditioned solves move by ~10 %)
TIGHT = {"diag_10ws2", "diag_100ws2", "diag_1000ws2", "diag_Nws2"}


def rows(regime, kind="mean"):
    with open(CSV, newline="") as fh:
        return {r["preconditioner"]: r for r in csv.DictReader(fh) if r["regime"] == regime and r["solve_kind"] == kind}


def usa_temp():
    """x min-max normalised per axis, y standardised (benchmark_cg_preconditioning_realdata.py:67-73): the inputs of golden c3."""
    _, x, y = load_case("c3_matern52_usatemp")
    assert x.shape == (4766, 2) and float(x.min()) == 0.0 and float(x.max()) == 1.0
    return x, y


def kernel(regime):
    """SquaredExponential(dimension=2) + set_hyper (:76-80): the hyper-parameters pass through the kernel's float32 log-space
    storage, which is what makes h = 0.8389676779198806 rather than the value for l = 0.04 exactly."""
    from kernels.squared_exponential import SquaredExponential
    ls, var, _ = REGIMES[regime]
    k = SquaredExponential(dimension=2)
    k.set_hyper("lengthscale", ls)
    k.set_hyper("variance", var)
    return k


def domain_length(x):
    return float((x.max(dim=0).values - x.min(dim=0).values).max().item())       # :98


def count_agrees(name, want, got, history, tol=CG_TOL):
    """Does a solve that stopped after `got` iterations (relative residuals `history`, possibly longer than `got`) agree with the
    CSV's `want`?  In these systems (condition ~1e9, tolerance 1e-3) the residual does not fall through the tolerance, it
    oscillates around it for dozens of iterations (very_hard / diag_Nws2, exact transform: below at 102-103, 115-118, 121, ...;
    the CSV stops at 116), so the FIRST crossing moves with the last bits of the transform.  Agreement = the same count +- 3, or
    -- inside a 15 % band -- the curve here is at the tolerance where the reference stopped (within 5 iterations, a factor 1.5).
    The two ill-conditioned solves (none, diag_ws2: 200-800 iterations) move by ~10 % under another correct NUFFT: band only."""
    if abs(got - want) <= 3:
        return True
    if abs(got - want) > 0.15 * want:
        return False
    if name not in TIGHT:
        return True
    # "at the tolerance": the oscillation has an amplitude of +-40 % here (0.7e-3 .. 1.4e-3 over the last dozen iterations), and the
    # last bits of the transform decide which dip crosses first -- 111 or 115 for hard / diag_10ws2 while the small-N pair pass
    # still summed with floating-point atomics (round 4 made it exact fixed point: the counts are now the same in every run,
    # 611 / 194 / 111 / 103 / 100 / 100 against the CSV's 629 / 194 / 111 / 103 / 100 / 99).  The curve must come within a factor 1.5
    # of the tolerance in the five iterations around the reference's stop.
    window = [float(r) for r in history[max(0, want - 5):want + 5]]
    return len(window) > 0 and min(window) <= 1.5 * tol
