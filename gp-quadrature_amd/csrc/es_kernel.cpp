// Host-side window parameterisation (see es_kernel.hpp).  Pure C++, no device code.
#include "es_kernel.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace efgp {

double es_window(double z, double beta) {
    double q = 1.0 - z * z;
    if (q < 0.0) q = 0.0;
    return std::exp(beta * (std::sqrt(q) - 1.0));
}

static const double kSigmaCalibrated = 5.0;   // upper end of the upsampling ratios the width formula was fitted on

// Measured l2 error of this window against the exact sums, round 4: 3000 random transforms (tools/r4/fuzz_over2.py, seeds 0-29:
// d = 1..3, 3..90 modes, tolerances 1e-3..1e-11, 1..3e5 points), error = K e^1.5 exp(-pi w sqrt(1 - 1/sigma)) with
//   K median / max   sigma 2.0-2.2   2.5-2.8    2.8-3.2    3.2-4      (the shape parameter beta = 0.976 pi w (1 - 1/(2 sigma)) is
//   d = 1            1.25 / 3.1     1.6 / 4.2  1.8 / 3.6  2.0 / 5.4   the sigma = 2 optimum carried along: it loses on fine grids;
//   d = 2            1.74 / 3.5     2.2 / 4.1  2.6 / 5.8  3.0 / 6.0   the errors of the axes add up; the maximum over point positions
//   d = 3            2.26 / 4.2     2.8 / 5.1  3.0 / 6.6  3.7 / 9.7   is about twice the median)
// The round-3 model (C = e^1.5 up to sigma 2.6, no dimension) sized the window for the 1-D median: worst case 4.13 x the
// requested tolerance.  The constant below is HALF the measured maximum, i.e. every case of that set lands within 2 x its
// tolerance: ln C = 1.5 + {0.45, 0.70, 0.90}[d - 1] + log1p(0.35 (sigma - 2)), + 0.7 for mode boxes so small that their grid is
// the 32-cell minimum at a ratio beyond 5.  EFGP_WIDTH_MODEL_R3=1 restores the round-3 model (A/B measurements).
static double es_log_error_constant(double sigma, int dim, bool tiny_box) {
    static const bool r3 = std::getenv("EFGP_WIDTH_MODEL_R3") != nullptr;
    if (r3) return 1.5 + std::log1p(std::max(0.0, sigma - 2.6));
    static const double dim_term[3] = {0.45, 0.70, 0.90};
    return 1.5 + dim_term[std::min(3, std::max(1, dim)) - 1] + std::log1p(0.35 * std::max(0.0, sigma - 2.0)) + (tiny_box ? 0.7 : 0.0);
}

int es_width_for_tol(double tol, double sigma, int dim) {
    if (!(tol > 0.0)) tol = 1e-16;
    if (tol < 1e-16) tol = 1e-16;
    if (sigma < 1.1) sigma = 1.1;
    const bool tiny_box = sigma > kSigmaCalibrated;
    if (sigma > kSigmaCalibrated) sigma = kSigmaCalibrated;
    double rate = M_PI * std::sqrt(1.0 - 1.0 / sigma);
    int w = (int)std::ceil((std::log(1.0 / tol) + es_log_error_constant(sigma, dim, tiny_box)) / rate);
    return std::min(kMaxWidth, std::max(2, w));
}

namespace {

// window value at cell j as a function of s in [-1,1] (see EsParams::coef)
long double cell_value(int w, double beta, int j, long double s) {
    long double z = ((s + 1.0L) * 0.5L - 0.5L * w + j) * (2.0L / w);
    long double q = 1.0L - z * z;
    if (q < 0.0L) q = 0.0L;
    return expl((long double)beta * (sqrtl(q) - 1.0L));
}

// cos(pi k (i + 1/2) / n) and the Chebyshev nodes cos(pi (i + 1/2) / n) of one degree: the same for every cell of a window, so
// they are built once per degree (round 4: a training step with a new mode count spent 0.2-0.4 ms per window set in cosl / expl)
struct ChebTables {
    int n = 0;
    std::vector<long double> node, ck;        // node[i], ck[k * n + i]
    void build(int n_) {
        n = n_;
        const long double pi = acosl(-1.0L);
        node.resize(n);
        ck.resize((size_t)n * n);
        for (int i = 0; i < n; ++i) node[i] = cosl(pi * (i + 0.5L) / n);
        for (int k = 0; k < n; ++k)
            for (int i = 0; i < n; ++i) ck[(size_t)k * n + i] = cosl(pi * k * (i + 0.5L) / n);
    }
};

// Chebyshev interpolant of degree deg on [-1,1] -> monomial coefficients (same operations on the same operands as the original
// per-cell version: bit-identical coefficients)
void cheb_fit(int w, double beta, int j, int deg, const ChebTables& tab, long double* mono) {
    const int n = deg + 1;
    std::vector<long double> fv(n), a(n);
    for (int i = 0; i < n; ++i) fv[i] = cell_value(w, beta, j, tab.node[i]);
    for (int k = 0; k < n; ++k) {
        long double acc = 0;
        for (int i = 0; i < n; ++i) acc += fv[i] * tab.ck[(size_t)k * n + i];
        a[k] = acc * (k == 0 ? 1.0L : 2.0L) / n;
    }
    // sum_k a_k T_k(s) -> monomials, by the three-term recurrence on coefficient vectors
    std::vector<long double> Tkm1(n, 0.0L), Tk(n, 0.0L), Tkp1(n, 0.0L);
    for (int k = 0; k < n; ++k) mono[k] = 0.0L;
    Tkm1[0] = 1.0L;                       // T_0
    mono[0] += a[0];
    if (n > 1) {
        Tk[1] = 1.0L;                     // T_1
        for (int m = 0; m < n; ++m) mono[m] += a[1] * Tk[m];
    }
    for (int k = 2; k < n; ++k) {
        std::fill(Tkp1.begin(), Tkp1.end(), 0.0L);
        for (int m = 0; m + 1 < n; ++m) Tkp1[m + 1] += 2.0L * Tk[m];
        for (int m = 0; m < n; ++m) Tkp1[m] -= Tkm1[m];
        for (int m = 0; m < n; ++m) mono[m] += a[k] * Tkp1[m];
        Tkm1.swap(Tk);
        Tk.swap(Tkp1);
    }
}

// the window's values at the 201 check points of every cell: they do not depend on the degree under test.  The window is even
// (cell w - 1 - j at -s is cell j at s, and the check points are symmetric), so half the cells are evaluated and mirrored: the
// mirrored values differ from directly evaluated ones by a few 1e-20 (80-bit rounding of z), far below the 2e-15 floor of the
// fit budget they are compared against.
constexpr int kCheckPoints = 201;
void fit_reference(int w, double beta, std::vector<double>* ref) {
    ref->resize((size_t)w * kCheckPoints);
    for (int j = 0; j < (w + 1) / 2; ++j)
        for (int t = 0; t < kCheckPoints; ++t) {
            double s = -1.0 + 2.0 * t / 200.0;
            const double val = (double)cell_value(w, beta, j, s);
            (*ref)[(size_t)j * kCheckPoints + t] = val;
            (*ref)[(size_t)(w - 1 - j) * kCheckPoints + (kCheckPoints - 1 - t)] = val;
        }
}

double fit_error(const EsParams& p, const std::vector<double>& ref) {
    double worst = 0.0;
    const int stride = kMaxDegree + 1;
    for (int j = 0; j < p.w; ++j) {
        for (int t = 0; t < kCheckPoints; ++t) {
            double s = -1.0 + 2.0 * t / 200.0;
            double acc = p.coef[j * stride + p.degree];
            for (int k = p.degree - 1; k >= 0; --k) acc = acc * s + p.coef[j * stride + k];
            worst = std::max(worst, std::fabs(acc - ref[(size_t)j * kCheckPoints + t]));
        }
    }
    return worst;
}

}  // namespace

int es_make_params(double tol, double sigma, EsParams* p, int dim) {
    if (!p) return -1;
    std::memset(p, 0, sizeof(*p));
    // Tiny mode boxes sit on the 32-cell minimum grid (sigma up to ~10), outside the range the error model was
    // calibrated on: design the window as for sigma = 5 (a finer grid than assumed only moves the aliases further
    // out).  Found by tools/fuzz_nufft.py: 3-D, 4 modes per axis, tol 1e-7 gave 1.2e-6 with the sigma = 8 window.
    p->w = es_width_for_tol(tol, sigma, dim);
    if (sigma > kSigmaCalibrated) sigma = kSigmaCalibrated;
    p->beta = 0.976 * M_PI * p->w * (1.0 - 1.0 / (2.0 * sigma));
    const int stride = kMaxDegree + 1;
    // Fit error budget: a tenth of the tolerance.  (A twentieth made the search overshoot for the common W = 8,
    // tol = 6e-8 window: degree 9 reaches 4.0e-9 and the edge cells' sqrt singularity then holds the error above
    // 3e-9 until degree 14 -- 55 % more Horner work in every spread / gather kernel for nothing measurable.)
    const double target = std::max(0.1 * tol, 2e-15);
    long double mono[kMaxDegree + 1];
    std::vector<double> ref;
    fit_reference(p->w, p->beta, &ref);
    ChebTables tab;
    for (int deg = std::min(kMaxDegree, std::max(4, p->w + 1)); deg <= kMaxDegree; ++deg) {
        p->degree = deg;
        tab.build(deg + 1);
        for (int j = 0; j < p->w; ++j) {
            cheb_fit(p->w, p->beta, j, deg, tab, mono);
            for (int k = 0; k <= kMaxDegree; ++k) p->coef[j * stride + k] = (k <= deg) ? (double)mono[k] : 0.0;
        }
        p->fit_error = fit_error(*p, ref);
        if (p->fit_error <= target) break;
    }
    return 0;
}

namespace {
// Gauss-Legendre nodes and weights on [-1, 1] (Newton iteration on P_n), computed once per process
struct GaussLegendre96 {
    static constexpr int nq = 96;
    long double xs[nq], wt[nq];
    GaussLegendre96() {
        const long double pi = acosl(-1.0L);
        for (int i = 0; i < nq; ++i) {
            long double x = cosl(pi * (i + 0.75L) / (nq + 0.5L));
            long double dp = 1;
            for (int it = 0; it < 100; ++it) {
                long double p0 = 1.0L, p1 = x;
                for (int k = 2; k <= nq; ++k) {
                    long double pk = ((2 * k - 1) * x * p1 - (k - 1) * p0) / k;
                    p0 = p1; p1 = pk;
                }
                dp = nq * (x * p1 - p0) / (x * x - 1.0L);
                long double dx = p1 / dp;
                x -= dx;
                if (fabsl(dx) < 1e-19L) break;
            }
            {   // recompute derivative at the converged node
                long double p0 = 1.0L, p1 = x;
                for (int k = 2; k <= nq; ++k) {
                    long double pk = ((2 * k - 1) * x * p1 - (k - 1) * p0) / k;
                    p0 = p1; p1 = pk;
                }
                dp = nq * (x * p1 - p0) / (x * x - 1.0L);
            }
            xs[i] = x;
            wt[i] = 2.0L / ((1.0L - x * x) * dp * dp);
        }
    }
};
}  // namespace

void es_deconv_factors(const EsParams& p, int64_t nf, int64_t n_modes, std::vector<double>* out) {
    // Gauss-Legendre on [0,1] (integrand even).  Round 3: a training loop that changes the mode count at every step (1-D models)
    // spent 2.5-6.7 ms per window here -- 96 expl + sqrtl + cosl per mode.  The window values at the nodes do not depend on the
    // mode and the factor is even in it: same operations on the same operands as before (bit-identical factors), a third of the
    // transcendentals per mode and half the modes.
    static const GaussLegendre96 gl;
    constexpr int nq = GaussLegendre96::nq;
    const long double pi = acosl(-1.0L);
    long double zs[nq], tq[nq];
    for (int q = 0; q < nq; ++q) {
        // map node from [-1,1] to z in [0,1]:  z = (x+1)/2, weight/2; integrand even -> times 2
        const long double z = 0.5L * (gl.xs[q] + 1.0L);
        const long double ph = expl((long double)p.beta * (sqrtl(std::max(0.0L, 1.0L - z * z)) - 1.0L));
        zs[q] = z;
        tq[q] = gl.wt[q] * 0.5L * 2.0L * ph;
    }
    out->resize((size_t)n_modes);
    const int64_t kmin = -(n_modes / 2);
    // cos(k theta_q), theta_q = (w pi / nf) z_q, for k = 0 .. kmax at every node by Reinsch's form of the three-term recurrence
    // (d_{k+1} = d_k + delta c_k, c_{k+1} = c_k + d_{k+1}, delta = -4 sin^2(theta / 2): the error grows like k eps_80bit, far
    // below a double's rounding) instead of one cosl per (mode, node): round 4, 0.15-0.25 ms per window set -> microseconds;
    // the factors equal the cosl version's to the last bit or one ulp (tests/test_host_logic.py).
    const int64_t kmax = std::max<int64_t>(-kmin, n_modes - 1 + kmin);
    std::vector<long double> acc((size_t)kmax + 1, 0.0L);
    const long double base = (long double)p.w * pi / (long double)nf;
    for (int q = 0; q < nq; ++q) {
        const long double theta = base * zs[q];
        const long double sh = sinl(0.5L * theta);
        const long double delta = -4.0L * sh * sh;
        long double c = 1.0L, d = 0.5L * delta;          // c_0 = 1, d_1 = c_1 - c_0 = -2 sin^2(theta / 2)
        acc[0] += tq[q];
        for (int64_t k = 1; k <= kmax; ++k) {
            c += d;                                       // c_k
            acc[(size_t)k] += tq[q] * c;
            d += delta * c;                               // d_{k+1}
        }
    }
    for (int64_t i = 0; i < n_modes; ++i) {
        const int64_t kk = kmin + i;
        long double P = 0.5L * p.w * acc[(size_t)(kk < 0 ? -kk : kk)];
        (*out)[(size_t)i] = (double)(1.0L / P);
    }
}

int64_t next_smooth_even(int64_t n) {
    if (n < 2) n = 2;
    if (n & 1) ++n;
    for (;; n += 2) {
        int64_t m = n;
        while (m % 2 == 0) m /= 2;
        while (m % 3 == 0) m /= 3;
        while (m % 5 == 0) m /= 5;
        if (m == 1) return n;
    }
}

// Fine-grid size for n_modes modes.
// Small grids take the next size of the COARSE ladder 32, 48, 64, 96, 128, 192, ... (2^k and 3 * 2^k): a hyper-parameter
// optimisation changes mtot every few steps, every new FFT length costs a rocFFT runtime compilation (0.6-1.2 s per length,
// measured: tools/train_loop_steps.py -- a 2.3 s stall at each of mtot 23 -> 21 -> 19 -> 17 against 0.5 ms steps), and on the
// ladder all of mtot 17..23 share the lengths 96 (Toeplitz box) and 48 (probes).  The transform of so small a grid costs
// microseconds whatever its size, the spreaders' work per point does not depend on it, and a larger upsampling ratio only
// narrows the window.  Larger grids (where cells cost memory and FFT time: beyond 512 per axis, 96 in 3-D) keep the dense
// choice: among the 2^a3^b5^c even sizes in [2 n, 2.5 n] the one that needs the narrowest window (ties: the smallest grid).
int64_t es_fine_size(int64_t n_modes, double tol, int dim, bool dense) {
    const int64_t lo = std::max<int64_t>(32, next_smooth_even(2 * n_modes));
    const int64_t hi = std::max<int64_t>(lo, (5 * n_modes) / 2);
    int64_t best = lo;
    int best_w = es_width_for_tol(tol, (double)lo / (double)n_modes, dim);
    for (int64_t c = next_smooth_even(lo + 2); c <= hi; c = next_smooth_even(c + 2)) {
        const int w = es_width_for_tol(tol, (double)c / (double)n_modes, dim);
        if (w < best_w) {
            best_w = w;
            best = c;
        }
    }
    const int64_t coarse_limit = dim >= 3 ? 96 : 512;
    int64_t pick = best;
    if (2 * n_modes <= coarse_limit && std::getenv("EFGP_DENSE_FINE_SIZES") == nullptr) {
        // the first ladder size that needs no wider a window than the dense choice (the width falls with the ratio)
        // The width model is not monotone in the ratio beyond the calibrated range (its error constant grows with sigma), so a
        // ladder size with a window as narrow as the dense choice's need not exist (tol ~ 0.035, n = 40: W = 2 at 100 cells,
        // W = 3 on 96, 128, 192, ... and at every ratio >= 5): stop where es_make_params clamps the ratio -- the width no
        // longer changes beyond it -- and keep the dense choice.
        pick = 0;
        for (int64_t p2 = 32; !pick && p2 <= (int64_t)1 << 40; p2 *= 2) {
            for (int64_t c : {p2, p2 + p2 / 2})
                if (!pick && c >= 2 * n_modes && es_width_for_tol(tol, (double)c / (double)n_modes, dim) <= best_w) pick = c;
            if ((double)p2 > kSigmaCalibrated * (double)n_modes && p2 >= 2 * n_modes) break;
        }
        if (!pick) pick = best;
    }
    // Dense 2-D point sets (round 3): when millions of points share a few thousand fine cells the grid is nearly free (the MFMA
    // spreader flushes register tiles into an L2-resident accumulator, the small-grid transforms are dense DFT launches of
    // microseconds) while every point pays for the window width: W rows of operand stores and W + 1 tile columns in the spreader,
    // W^2 multiply-adds and W * ceil(W/2) 16-byte LDS reads in the gather.  Take the smallest 2^a3^b5^c grid (<= 256 cells, ratio
    // <= 5) whose window is one cell narrower.  Grids of <= 128 cells must keep the gather's two 16-byte-aligned copies inside the
    // 160 KB of LDS (interp_real2_pair_kernel).
    if (dense && dim == 2) {
        const int w0 = es_width_for_tol(tol, (double)pick / (double)n_modes, dim);
        const int64_t top = std::min<int64_t>(256, 5 * n_modes);
        for (int64_t c = next_smooth_even(pick + 2); c <= top; c = next_smooth_even(c + 2)) {
            const int w = es_width_for_tol(tol, (double)c / (double)n_modes, dim);
            if (w >= w0) continue;
            const int64_t pitch = (c + 2 * ((w + 1) / 2) + 1) & ~(int64_t)1;
            const bool gather_fits = 2 * (c + w - 1) * pitch * 8 <= 160 * 1024;
            if (c > 128 || gather_fits) pick = c;
            break;
        }
    }
    return pick;
}

}  // namespace efgp
