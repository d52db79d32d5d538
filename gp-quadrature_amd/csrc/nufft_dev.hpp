// Device-side pieces shared by the NUFFT translation units (nufft.hip, spread_mfma.hip): strength sources,
// fine-grid geometry, the symmetric Horner window evaluation.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "es_kernel.hpp"

namespace efgp {

constexpr int kSpreadThreads = 1024;
constexpr int kInterpThreads = 1024;        // LDS-resident fine grid
constexpr int kInterpThreadsGlobal = 256;   // fine grid read through L2
constexpr double kFixMagic = 6755399441055744.0;   // 1.5 * 2^52: adding it rounds to an integer in the mantissa

enum StrengthMode {
    STR_REAL = 0,            // one real row per fine grid
    STR_COMPLEX = 1,         // one complex row (re, im channels)
    STR_REAL_AND_ONES = 2,   // (y, 1): the fit-time pair
    STR_ONES = 3,
    STR_REAL_PAIR = 4,       // two real rows (2g, 2g+1) share one complex fine grid
    STR_RNG = 5,             // one Rademacher row generated in the kernel
    STR_RNG_PAIR = 6         // two Rademacher rows
};

// counter-based Rademacher probe: sign(seed, row, point index) -- splitmix64 finaliser
__host__ __device__ __forceinline__ unsigned long long efgp_mix64(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
constexpr unsigned long long kRademacherRowStride = 0xD1342543DE82EF95ull;
// probe (seed, row, n) = sign bit of a hash of row * kRademacherRowStride + n in WRAPPING 64-bit arithmetic: row r at index n is
// row 0 at index n + r * kRademacherRowStride (type1_real_rows uses that for a single odd row)
__host__ __device__ __forceinline__ double efgp_rademacher(unsigned long long seed, long long row, long long n) {
    const unsigned long long r = efgp_mix64(seed ^ efgp_mix64((unsigned long long)row * kRademacherRowStride +
                                                             (unsigned long long)n));
    return (r >> 63) ? 1.0 : -1.0;
}

struct StrengthSrc {
    const double* c;          // base pointer of all rows (layout by mode)
    int64_t npts;             // row length
    int mode;
    unsigned long long seed;
    int64_t index_offset;     // added to the point index for the RNG (global index of this shard's first point)
};

// strengths (c0, c1) of point `n` (original index) for fine grid `g`
__device__ __forceinline__ void fetch_strength(const StrengthSrc& s, int g, int64_t n, double& c0, double& c1) {
    c0 = 1.0;
    c1 = 1.0;
    switch (s.mode) {
        case STR_REAL: c0 = s.c[(int64_t)g * s.npts + n]; break;
        case STR_COMPLEX: {
            const double2 cc = reinterpret_cast<const double2*>(s.c)[(int64_t)g * s.npts + n];
            c0 = cc.x;
            c1 = cc.y;
        } break;
        case STR_REAL_AND_ONES: c0 = s.c[n]; break;
        case STR_REAL_PAIR:
            c0 = s.c[(int64_t)(2 * g) * s.npts + n];
            c1 = s.c[(int64_t)(2 * g + 1) * s.npts + n];
            break;
        case STR_RNG: c0 = efgp_rademacher(s.seed, g, (long long)((unsigned long long)n + (unsigned long long)s.index_offset)); break;
        case STR_RNG_PAIR:
            c0 = efgp_rademacher(s.seed, 2 * g, (long long)((unsigned long long)n + (unsigned long long)s.index_offset));
            c1 = efgp_rademacher(s.seed, 2 * g + 1, (long long)((unsigned long long)n + (unsigned long long)s.index_offset));
            break;
        default: break;
    }
}

struct GridGeom {
    int64_t nf[3];      // fine-grid size per dimension (unused dims = 1)
    double scale[3];    // X = scale * (x - xcen): h * nf
    double xcen[3];
    int64_t cells;      // prod nf
};

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
// The Horner coefficients are read through the CONSTANT address space: the table is never written while a
// kernel runs, and telling the compiler so lets it use scalar loads (s_load) for these wave-uniform values.
// With a plain global pointer the interpolation kernels (which also store to global memory) fell back to
// vector global loads inside the Horner loop -- a chain of ~2*degree dependent L2 round trips per point.
typedef const __attribute__((address_space(4))) double* const_coef_ptr;

// All dimensions of a point at once, with half the coefficient traffic: the window is even, so polynomial
// W-1-j at s equals polynomial j at -s.  Only the first RH = ceil(W/2) polynomials are read (table
// `sym` behind the plain one: [kMaxDegree+1][RHP] doubles, one scalar load per degree) and every loaded
// coefficient feeds 2*D FMAs as a scalar operand.
__host__ __device__ constexpr int sym_row(int W) { return (W + 1) / 2 <= 2 ? 2 : ((W + 1) / 2 <= 4 ? 4 : 8); }

template <int D, int W>
__device__ __forceinline__ void window_eval(const double* __restrict__ coef_generic, int degree, const double (&X)[3],
                                            int64_t nf0, int64_t nf1, int64_t nf2, int& f0, int& f1, int& f2,
                                            double (&v0)[W], double (&v1)[W], double (&v2)[W]) {
    constexpr int RH = (W + 1) / 2, RHP = sym_row(W);
    const_coef_ptr coef = (const_coef_ptr)(coef_generic + (kMaxDegree + 1) * W);
    const int64_t nfs[3] = {nf0, nf1, nf2};
    int fs[3] = {0, 0, 0};
    double s[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const double i0 = ceil(X[d] - 0.5 * W);
        s[d] = 2.0 * (i0 - X[d] + 0.5 * W) - 1.0;
        int f = (int)i0;
        if (f < 0) f += (int)nfs[d];
        fs[d] = f;
    }
    // p_j(s) = E_j(s^2) + s O_j(s^2) and its mirror p_{W-1-j}(s) = p_j(-s) = E_j - s O_j: the even and the odd half of
    // the polynomial are evaluated once for the pair (degree + 1 fused multiply-adds per pair instead of 2 degree)
    double vp[D][RH], vm[D][RH];
    double t[D];
#pragma unroll
    for (int d = 0; d < D; ++d) t[d] = s[d] * s[d];
    const int ke = degree & ~1, ko = (degree - 1) | 1;      // highest even / odd power (degree >= 1)
#pragma unroll
    for (int j = 0; j < RH; ++j) {
        const double ce = coef[ke * RHP + j], co = coef[ko * RHP + j];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            vp[d][j] = ce;
            vm[d][j] = co;
        }
    }
    for (int k = ke - 2; k >= 0; k -= 2) {
#pragma unroll
        for (int j = 0; j < RH; ++j) {
            const double c = coef[k * RHP + j];
#pragma unroll
            for (int d = 0; d < D; ++d) vp[d][j] = fma(vp[d][j], t[d], c);
        }
    }
    for (int k = ko - 2; k >= 1; k -= 2) {
#pragma unroll
        for (int j = 0; j < RH; ++j) {
            const double c = coef[k * RHP + j];
#pragma unroll
            for (int d = 0; d < D; ++d) vm[d][j] = fma(vm[d][j], t[d], c);
        }
    }
#pragma unroll
    for (int j = 0; j < RH; ++j) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const double e = vp[d][j], o = vm[d][j];
            vp[d][j] = fma(s[d], o, e);
            vm[d][j] = fma(-s[d], o, e);
        }
    }
    f0 = fs[0];
    f1 = fs[1];
    f2 = fs[2];
#pragma unroll
    for (int j = 0; j < RH; ++j) {
        v0[j] = vp[0][j];
        if (W - 1 - j != j) v0[W - 1 - j] = vm[0][j];
        if (D > 1) {
            v1[j] = vp[D > 1 ? 1 : 0][j];
            if (W - 1 - j != j) v1[W - 1 - j] = vm[D > 1 ? 1 : 0][j];
        }
        if (D > 2) {
            v2[j] = vp[D > 2 ? 2 : 0][j];
            if (W - 1 - j != j) v2[W - 1 - j] = vm[D > 2 ? 2 : 0][j];
        }
    }
}

__device__ __forceinline__ double fold(double X, double nf) {
    // reciprocal instead of an fp64 division per coordinate (1/nf is loop invariant); a quotient that lands one
    // unit off next to an integer leaves X within rounding of 0 or nf, which the guards below fold back
    X -= nf * floor(X * (1.0 / nf));
    // guard against X == nf after rounding
    if (X >= nf) X -= nf;
    if (X < 0.0) X = 0.0;
    return X;
}

__device__ __forceinline__ int wrap(int i, int nf) { return i >= nf ? i - nf : i; }

}  // namespace efgp
