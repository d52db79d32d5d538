// Host-side parameterisation of the spreading window used by the HIP NUFFT.
//
// The reference reaches a third-party NUFFT (FINUFFT through pytorch_finufft,
// efgpnd.py:1496-1499, 1533-1549, 1679).  Its published algorithm is restated here
// from the method description (Barnett, Magland, af Klinteberg 2019): spread with the
// "exponential of semicircle" window phi(z) = exp(beta (sqrt(1-z^2) - 1)), |z|<=1, of
// width w fine-grid cells, FFT the fine grid, divide the kept modes by the window's
// Fourier transform.  Nothing here is copied from FINUFFT sources (absent from the tree).
#pragma once
#include <cstdint>
#include <vector>

namespace efgp {

constexpr int kMaxWidth = 16;
constexpr int kMaxDegree = 19;   // Horner degree <= kMaxDegree  (kMaxDegree+1 coefficients)

struct EsParams {
    int w = 0;            // window width in fine-grid cells
    double beta = 0.0;    // shape parameter
    int degree = 0;       // degree of the per-cell polynomials
    // coef[j*(kMaxDegree+1) + k]: coefficient of s^k of the polynomial giving the window value at
    // cell j (0..w-1) for a point whose first covered cell is i0, with
    //   s = 2*(i0 - X + w/2) - 1 in [-1,1),  X = point position in fine-grid units.
    double coef[kMaxWidth * (kMaxDegree + 1)];
    double fit_error = 0.0;   // max abs error of the polynomial fit (window peak = 1)
};

// window value, z in [-1,1]
double es_window(double z, double beta);

// width for a requested tolerance at upsampling ratio sigma = nf / n_modes in a transform of `dim` dimensions (the errors of the
// axes add up)
int es_width_for_tol(double tol, double sigma, int dim = 1);

// fill p (w, beta, polynomials) for tolerance/sigma; returns 0 or negative error
int es_make_params(double tol, double sigma, EsParams* p, int dim = 1);

// Fourier-side correction factors: out[i] = 1 / P(k_i), i = 0..n_modes-1, with k in CMCL order
// (k = -(n_modes/2) ... (n_modes-1)/2) where  P(k) = (w/2) * int_{-1}^{1} phi(z) cos(k w pi z / nf) dz.
void es_deconv_factors(const EsParams& p, int64_t nf, int64_t n_modes, std::vector<double>* out);

// smallest size >= n of the form 2^a 3^b 5^c (and even)
int64_t next_smooth_even(int64_t n);
// fine-grid size a plan takes for n_modes modes per axis at tolerance tol in `dim` dimensions (see es_kernel.cpp);
// dense: a 2-D plan over millions of points trades a finer grid for a window one cell narrower
int64_t es_fine_size(int64_t n_modes, double tol, int dim, bool dense = false);

}  // namespace efgp
