// Host-side scalar work of the quadrature grid for the built-in kernels (no device code).
//
// The reference finds the two truncation bounds of `get_xis` by bisection on Python scalars (utils/kernels.py:28-69,
// 94-105) and evaluates the spectral density on the mode grid with a handful of torch CPU ops (efgpnd.py:766-780,
// kernels/*.py).  With the fit step at 0.3 ms of device time that host work (0.15 ms of interpreter and torch
// dispatch per fit) had become what the step waits for.  Here the same arithmetic runs in C: the bisection with the
// operations in the order of the Python expressions it replaces (same libm exp / pow, no contraction), so the bounds --
// and with them h and mtot -- are bit-identical (tests/test_host_logic.py sweeps kernels, dimensions and tolerances);
// the weights differ from torch's vectorised exp / pow by rounding only.
#pragma STDC FP_CONTRACT OFF
#include <cmath>
#include <cstdint>

#include "../../include/efgp_hip.h"
#include "common.hpp"

namespace {

// libm through pointers the optimiser cannot see through: Python's `x ** 2` IS pow(x, 2.0) and math.exp IS exp(); a compiler
// that turns pow(x, 2.0) into x * x (or picks a vector variant) would make the bounds differ in the last bit once in a while
double (*volatile p_pow)(double, double) = ::pow;
double (*volatile p_exp)(double) = ::exp;

// kernels/squared_exponential.py: _k_scalar / _S_scalar; kernels/matern.py: _k_scalar / _S_scalar.
// c0: SE (2 pi l^2)^(d/2) * var; Matern var * scaling(l) -- formed by the caller in Python (math.gamma is Python's own).
struct KernelFn {
    int kind;        // 0 squared exponential, 1 Matern
    int dim;
    double nu, ell, var, c0, s0;
    double k(double r) const {
        if (kind == 0) return var * p_exp(-0.5 * p_pow(r / ell, 2.0));
        const double s = std::fabs(r) / ell;
        if (nu == 0.5) return var * p_exp(-s);
        if (nu == 1.5) return var * (1 + std::sqrt(3.0) * s) * p_exp(-std::sqrt(3.0) * s);
        return var * (1 + std::sqrt(5.0) * s + 5 * p_pow(s, 2.0) / 3) * p_exp(-std::sqrt(5.0) * s);
    }
    double S(double r) const {
        if (kind == 0) {
            const double two_pi = 2.0 * M_PI;
            return c0 * p_exp(-p_pow(two_pi, 2.0) * p_pow(ell, 2.0) * (r * r) / 2);
        }
        return c0 * p_pow(2 * nu / p_pow(ell, 2.0) + (4 * p_pow(M_PI, 2.0)) * (r * r), -(nu + dim / 2.0));
    }
    double khat(double r) const { return std::fabs(p_pow(r, (double)(dim - 1))) * S(r) / s0; }
};

// GetTruncationBound.find_truncation_bound (utils/kernels.py:28-69): doubling from 1000, then bisection down to two
// adjacent doubles (the reference's fixed 200 passes leave the bracket unchanged from there on)
template <typename F>
double truncation_bound(double eps, F f) {
    double b = 1000.0;
    for (int i = 0; i < 10; ++i) {
        if (f(b) > eps) b *= 2;
        else break;
    }
    double a = 0.0, mid = (a + b) / 2;
    for (int i = 0; i < 200; ++i) {
        mid = (a + b) / 2;
        if (mid == a || mid == b) break;
        if (f(mid) > eps) a = mid;
        else b = mid;
    }
    return mid;
}

}  // namespace

extern "C" int efgp_grid_bounds(int kind, int dim, double nu, double lengthscale, double variance, double c0, double s0, double eps,
                                double trunc_eps, double* ltime_out, double* lfreq_out) {
    EFGP_REQUIRE(ltime_out && lfreq_out, "efgp_grid_bounds: null output");
    EFGP_REQUIRE(kind == 0 || (kind == 1 && (nu == 0.5 || nu == 1.5 || nu == 2.5)), "efgp_grid_bounds: kernel kind %d, nu %g not built in", kind, nu);
    EFGP_REQUIRE(dim >= 1 && dim <= 3 && lengthscale > 0.0 && variance > 0.0 && s0 > 0.0, "efgp_grid_bounds: bad kernel parameters");
    const KernelFn fn{kind, dim, nu, lengthscale, variance, c0, s0};
    *ltime_out = truncation_bound(eps, [&](double r) { return fn.k(r); });
    *lfreq_out = truncation_bound(trunc_eps, [&](double r) { return fn.khat(r); });
    return EFGP_OK;
}

// ws[k] = sqrt(S(|xi_k|) h^d) as complex (imaginary part 0) on the tensor grid xi = h * (-m..m)^d (efgpnd.py:766-780), and
// optionally dprime[k][0..1] = h^d * (dS/d lengthscale, dS/d variance) (kernels/*.py spectral_grad), complex as well.
extern "C" int efgp_spectral_weights_host(int kind, int dim, double nu, double lengthscale, double variance, double c0, double h, int mtot,
                                          double* ws_out, double* dprime_out) {
    EFGP_REQUIRE(ws_out, "efgp_spectral_weights_host: null output");
    EFGP_REQUIRE(kind == 0 || kind == 1, "efgp_spectral_weights_host: kernel kind %d not built in", kind);
    EFGP_REQUIRE(dim >= 1 && dim <= 3 && mtot >= 1 && (mtot & 1), "efgp_spectral_weights_host: bad grid");
    const int m = (mtot - 1) / 2;
    const double hd = std::pow(h, (double)dim);
    const double two_pi = 2.0 * M_PI, ell = lengthscale;
    int64_t M = 1;
    for (int a = 0; a < dim; ++a) M *= mtot;
    for (int64_t t = 0; t < M; ++t) {
        int64_t rem = t;
        double q = 0.0;
        double xs[3];
        for (int a = dim - 1; a >= 0; --a) {
            xs[a] = (double)((int)(rem % mtot) - m) * h;
            rem /= mtot;
        }
        for (int a = 0; a < dim; ++a) q += xs[a] * xs[a];
        double S, d_ell;
        if (kind == 0) {
            S = c0 * std::exp(-(two_pi * two_pi) * (ell * ell) * q / 2);
            d_ell = S * (dim / ell - (two_pi * two_pi) * ell * q);
        } else {
            const double den = 2 * nu / (ell * ell) + (4 * M_PI * M_PI) * q;
            S = c0 * std::pow(den, -(nu + dim / 2.0));
            d_ell = S * (-2 * nu / ell + (-(nu + dim / 2.0)) * (-4 * nu / (ell * ell * ell)) / den);
        }
        ws_out[2 * t] = std::sqrt(S * hd);
        ws_out[2 * t + 1] = 0.0;
        if (dprime_out) {
            dprime_out[4 * t] = hd * d_ell;
            dprime_out[4 * t + 1] = 0.0;
            dprime_out[4 * t + 2] = hd * (S / variance);
            dprime_out[4 * t + 3] = 0.0;
        }
    }
    return EFGP_OK;
}
