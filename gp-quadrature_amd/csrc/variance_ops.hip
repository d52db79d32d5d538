// M-scale device operations of the predictive variance (SURVEY kernels K5 / K8) behind the C ABI:
//   efgp_lag_sums          Hutchinson lag sums  c[r] = mean_j sum_{k-l=r} gamma_j[k] eta_j[l]   (efgpnd.py:1660-1664)
//   efgp_variance_rhs      right-hand sides ws .* conj(f(x*)) of the 'regular' variance          (efgpnd.py:1805-1812)
//   efgp_variance_contract s^2(x*) = max(0, Re sum_k f_k(x*) ws_k gamma_k)                        (efgpnd.py:1817-1820)
// The reference does these with torch.fft / dense torch ops; here they are hipFFT transforms plus three small kernels,
// so a caller that binds only include/efgp_hip.h can compute a variance end to end.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "common.hpp"
#include "line_fft.hpp"
#include "toeplitz_cg.hpp"

namespace efgp {

struct LagGeom {
    int d;
    int n[3];       // mtot per dimension (unused = 1)
    int s[3];       // 2 n - 1: the lag box of the result
    int p[3];       // transform length per dimension: the next 2^k / 3 * 2^k size >= s (see lag_length)
    int64_t M, S, P;   // prod n, prod s, prod p
};

// Transform length of the zero-padded correlation: any length >= 2 n - 1 gives the same lags, so it comes from the short ladder
// 1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, ...  The exact lag-box size 2 mtot - 1 is an odd length that changes with every mtot, and
// every FFT length new to the process is a runtime compilation in rocFFT (0.5-2 s each, see LABNOTES.md 4.6c).
static int lag_length(int s) {
    for (int p2 = 1;; p2 *= 2) {
        if (p2 >= s) return p2;
        if (p2 >= 2 && p2 + p2 / 2 >= s) return p2 + p2 / 2;
    }
}

// zero-padded copies: pad[j][0][...] = gamma_j, pad[j][1][...] = eta_j  (two transforms per probe in one batch)
__global__ __launch_bounds__(256) void lag_pad_kernel(LagGeom g, const double2* __restrict__ gam, const double* __restrict__ eta,
                                                      double2* __restrict__ pad) {
    const int j = blockIdx.y;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < g.P; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i;
        const int i2 = (int)(r % g.p[2]);
        r /= g.p[2];
        const int i1 = (int)(r % g.p[1]);
        const int i0 = (int)(r / g.p[1]);
        const bool in = i0 < g.n[0] && i1 < g.n[1] && i2 < g.n[2];
        const int64_t src = ((int64_t)i0 * g.n[1] + i1) * g.n[2] + i2;
        double2* row = pad + (int64_t)2 * j * g.P;
        row[i] = in ? gam[(int64_t)j * g.M + src] : make_double2(0.0, 0.0);
        row[g.P + i] = in ? make_double2(eta[(int64_t)j * g.M + src], 0.0) : make_double2(0.0, 0.0);
    }
}

// acc[i] (+)= sum_j G_j[i] conj(E_j[i])   (the inverse transform is linear: one inverse FFT of the probe sum)
__global__ __launch_bounds__(256) void lag_mul_sum_kernel(int64_t S, int J, const double2* __restrict__ pad, double2* __restrict__ acc,
                                                          int add, int conj_out = 0) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < S; i += (int64_t)gridDim.x * blockDim.x) {
        double re = 0.0, im = 0.0;
        for (int j = 0; j < J; ++j) {
            const double2 a = pad[(int64_t)2 * j * S + i], b = pad[(int64_t)(2 * j + 1) * S + i];
            re += a.x * b.x + a.y * b.y;
            im += a.y * b.x - a.x * b.y;
        }
        if (add) {
            re += acc[i].x;
            im += conj_out ? -acc[i].y : acc[i].y;
        }
        acc[i] = make_double2(re, conj_out ? -im : im);       // conj_out: the accumulator holds conj(sum) throughout
    }
}

// out (lag box s, FFT order) = factor * the same lags of the padded correlation (length p per dimension, FFT order)
__global__ __launch_bounds__(256) void lag_scale_kernel(LagGeom g, double factor, const double2* __restrict__ in, double2* __restrict__ out,
                                                        int conj_in = 0) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < g.S; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i;
        int c[3];
        c[2] = (int)(r % g.s[2]);
        r /= g.s[2];
        c[1] = (int)(r % g.s[1]);
        c[0] = (int)(r / g.s[1]);
        int64_t src = 0;
        for (int a = 0; a < 3; ++a) {
            const int lag = c[a] < g.n[a] ? c[a] : c[a] - g.s[a];      // FFT order of the lag box: 0..n-1, -(n-1)..-1
            src = src * g.p[a] + (lag >= 0 ? lag : lag + g.p[a]);
        }
        out[i] = make_double2(in[src].x * factor, (conj_in ? -in[src].y : in[src].y) * factor);
    }
}

// phase 2 pi h k . x for mode index t of the (mtot,)*d box, k = i - (mtot-1)/2 per dimension, last dimension fastest
__device__ __forceinline__ double mode_phase(int d, int mtot, double h, const double* __restrict__ x, int64_t t) {
    const int m = (mtot - 1) / 2;
    double ph = 0.0;
    for (int a = d - 1; a >= 0; --a) {
        const int ia = (int)(t % mtot);
        t /= mtot;
        ph += (double)(ia - m) * x[a];
    }
    return 2.0 * M_PI * h * ph;
}

__global__ __launch_bounds__(256) void variance_rhs_kernel(int d, int mtot, int64_t M, double h, const double* __restrict__ x,
                                                           const double2* __restrict__ ws, double2* __restrict__ rhs) {
    const int b = blockIdx.y;
    const double* xb = x + (int64_t)b * d;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < M; t += (int64_t)gridDim.x * blockDim.x) {
        double sn, cs;
        sincos(mode_phase(d, mtot, h, xb, t), &sn, &cs);
        const double2 w = ws[t];
        rhs[(int64_t)b * M + t] = make_double2(w.x * cs + w.y * sn, w.y * cs - w.x * sn);      // ws * conj(f)
    }
}

__global__ __launch_bounds__(256) void variance_contract_kernel(int d, int mtot, int64_t M, double h, const double* __restrict__ x,
                                                                const double2* __restrict__ ws, const double2* __restrict__ gamma,
                                                                double* __restrict__ out) {
    __shared__ double part[4];
    const int b = blockIdx.x;
    const double* xb = x + (int64_t)b * d;
    double acc = 0.0;
    for (int64_t t = threadIdx.x; t < M; t += blockDim.x) {
        double sn, cs;
        sincos(mode_phase(d, mtot, h, xb, t), &sn, &cs);
        const double2 w = ws[t], g = gamma[(int64_t)b * M + t];
        const double2 wg = make_double2(w.x * g.x - w.y * g.y, w.x * g.y + w.y * g.x);
        acc += cs * wg.x - sn * wg.y;                                                          // Re (f * ws * gamma)
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += part[i];
        out[b] = s > 0.0 ? s : 0.0;                                                           // clamp_min(0), efgpnd.py:1820
    }
}

// ws[k] = sqrt(S(|xi_k|) h^d) (complex, imaginary part 0) and optionally h^d (dS/dl, dS/dvariance) on the tensor grid
// xi = h (-m..m)^d for the built-in kernels (efgpnd.py:766-780, kernels/*.py): one launch instead of computing the M
// weights on the host and staging them through a pinned buffer (host-side cost of every fit: the step was waiting for it).
__global__ __launch_bounds__(256) void spectral_weights_kernel(int kind, int dim, double nu, double ell, double var, double c0, double h,
                                                               int mtot, int64_t M, double2* __restrict__ ws, double2* __restrict__ dprime) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M) return;
    const int m = (mtot - 1) / 2;
    int64_t rem = t;
    double q = 0.0;
    for (int a = dim - 1; a >= 0; --a) {
        const double xa = (double)((int)(rem % mtot) - m) * h;
        rem /= mtot;
        q += xa * xa;
    }
    const double two_pi = 6.283185307179586476925286766559, pi = 3.14159265358979323846264338327950288;
    double hd = h;
    for (int a = 1; a < dim; ++a) hd *= h;
    double S, d_ell;
    if (kind == 0) {
        S = c0 * exp(-(two_pi * two_pi) * (ell * ell) * q / 2);
        d_ell = S * (dim / ell - (two_pi * two_pi) * ell * q);
    } else {
        const double den = 2 * nu / (ell * ell) + (4 * pi * pi) * q;
        S = c0 * pow(den, -(nu + dim / 2.0));
        d_ell = S * (-2 * nu / ell + (-(nu + dim / 2.0)) * (-4 * nu / (ell * ell * ell)) / den);
    }
    ws[t] = make_double2(sqrt(S * hd), 0.0);
    if (dprime) {
        dprime[2 * t] = make_double2(hd * d_ell, 0.0);
        dprime[2 * t + 1] = make_double2(hd * (S / var), 0.0);
    }
}

}  // namespace efgp

using namespace efgp;

extern "C" {

int efgp_lag_sums(int device, int dim, int64_t mtot, const void* gamma, const double* eta, int nprobes, void* out, void* stream_) {
    EFGP_REQUIRE(dim >= 1 && dim <= 3, "efgp_lag_sums: dim must be 1, 2 or 3");
    EFGP_REQUIRE(mtot >= 1 && nprobes >= 1, "efgp_lag_sums: mtot and nprobes must be >= 1");
    EFGP_REQUIRE(gamma && eta && out, "efgp_lag_sums: null argument");
    DeviceCtx* ctx = device_ctx(device);
    if (!ctx) return EFGP_EHIP;
    DeviceGuard guard(device, (hipStream_t)stream_);
    hipStream_t stream = (hipStream_t)stream_;
    LagGeom g;
    g.d = dim;
    g.M = 1;
    g.S = 1;
    g.P = 1;
    int64_t sizes[3] = {1, 1, 1};
    // slots are right-aligned so that the LAST real dimension is the fastest one (n[2] / s[2])
    for (int a = 0; a < 3; ++a) {
        const bool real = a >= 3 - dim;
        g.n[a] = real ? (int)mtot : 1;
        g.s[a] = real ? (int)(2 * mtot - 1) : 1;
        g.p[a] = real ? lag_length(g.s[a]) : 1;
        g.M *= g.n[a];
        g.S *= g.s[a];
        g.P *= g.p[a];
    }
    for (int a = 0; a < dim; ++a) sizes[a] = lag_length((int)(2 * mtot - 1));
    if (dim == 2 && 2 * mtot - 1 <= 64 && std::getenv("EFGP_NO_LAG64") == nullptr) {
        // 2-D lag boxes up to 64 x 64 (mtot <= 32): every transform on the library's own one-workgroup 64 x 64 kernel -- the first
        // variance of a process otherwise waits 1.8 s for rocFFT to compile a length-48 kernel.  The inverse transform is the forward
        // one on the conjugate: the accumulator is kept conjugated and the copy-out conjugates back.
        g.p[1] = g.p[2] = 64;
        g.P = 4096;
        const int64_t per64 = std::max<int64_t>(1, std::min<int64_t>(nprobes, ((int64_t)256 << 20) / (int64_t)(2 * g.P * sizeof(double2))));
        double2* pad64 = (double2*)scratch(ctx, SLOT_TOEP_PAD, (size_t)(2 * per64 + 2) * g.P * sizeof(double2));
        if (!pad64) return EFGP_ENOMEM;
        double2* acc64 = pad64 + (int64_t)2 * per64 * g.P;
        double2* inv64 = acc64 + g.P;
        for (int64_t j0 = 0; j0 < nprobes; j0 += per64) {
            const int J = (int)std::min<int64_t>(per64, nprobes - j0);
            int rc = fft2d64_batch_launch((const double2*)gamma + j0 * g.M, 0, g.M, (int)mtot, (int)mtot, pad64, 2 * g.P, J, stream);
            if (rc != EFGP_OK) return rc;
            rc = fft2d64_batch_launch(eta + j0 * g.M, 1, g.M, (int)mtot, (int)mtot, pad64 + g.P, 2 * g.P, J, stream);
            if (rc != EFGP_OK) return rc;
            hipLaunchKernelGGL(lag_mul_sum_kernel, dim3(16), dim3(256), 0, stream, g.P, J, (const double2*)pad64, acc64, j0 > 0 ? 1 : 0, 1);
            EFGP_HIP_CHECK(hipGetLastError());
        }
        int rc = fft2d64_batch_launch(acc64, 0, g.P, 64, 64, inv64, g.P, 1, stream);
        if (rc != EFGP_OK) return rc;
        const int ob = (int)std::max<int64_t>(1, std::min<int64_t>((g.S + 255) / 256, 1024));
        hipLaunchKernelGGL(lag_scale_kernel, dim3(ob), dim3(256), 0, stream, g, 1.0 / ((double)g.P * (double)nprobes), (const double2*)inv64,
                           (double2*)out, 1);
        EFGP_HIP_CHECK(hipGetLastError());
        return EFGP_OK;
    }
    // probes are processed in slabs so that the padded transforms stay within ~256 MB of scratch
    const int64_t per = std::max<int64_t>(1, std::min<int64_t>(nprobes, ((int64_t)256 << 20) / (int64_t)(2 * g.P * sizeof(double2))));
    double2* pad = (double2*)scratch(ctx, SLOT_TOEP_PAD, (size_t)(2 * per + 1) * g.P * sizeof(double2));
    if (!pad) return EFGP_ENOMEM;
    double2* acc = pad + (int64_t)2 * per * g.P;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((g.P + 255) / 256, 1024));
    for (int64_t j0 = 0; j0 < nprobes; j0 += per) {
        const int J = (int)std::min<int64_t>(per, nprobes - j0);
        hipLaunchKernelGGL(lag_pad_kernel, dim3(blocks, J), dim3(256), 0, stream, g, (const double2*)gamma + j0 * g.M, eta + j0 * g.M, pad);
        EFGP_HIP_CHECK(hipGetLastError());
        int rc = fft_c2c(ctx, dim, sizes, 2 * J, pad, true, stream);
        if (rc != EFGP_OK) return rc;
        hipLaunchKernelGGL(lag_mul_sum_kernel, dim3(blocks), dim3(256), 0, stream, g.P, J, (const double2*)pad, acc, j0 > 0 ? 1 : 0);
        EFGP_HIP_CHECK(hipGetLastError());
    }
    int rc = fft_c2c(ctx, dim, sizes, 1, acc, false, stream);
    if (rc != EFGP_OK) return rc;
    const int oblocks = (int)std::max<int64_t>(1, std::min<int64_t>((g.S + 255) / 256, 1024));
    hipLaunchKernelGGL(lag_scale_kernel, dim3(oblocks), dim3(256), 0, stream, g, 1.0 / ((double)g.P * (double)nprobes), (const double2*)acc,
                       (double2*)out);
    EFGP_HIP_CHECK(hipGetLastError());
    return EFGP_OK;
}

int efgp_variance_rhs(int device, int dim, int64_t mtot, double h, const double* x_new, int64_t npts, const void* ws, void* rhs,
                      void* stream_) {
    EFGP_REQUIRE(dim >= 1 && dim <= 3 && mtot >= 1 && npts >= 0, "efgp_variance_rhs: bad sizes");
    if (npts == 0) return EFGP_OK;
    EFGP_REQUIRE(x_new && ws && rhs, "efgp_variance_rhs: null argument");
    if (!device_ctx(device)) return EFGP_EHIP;
    DeviceGuard guard(device, (hipStream_t)stream_);
    int64_t M = 1;
    for (int a = 0; a < dim; ++a) M *= mtot;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((M + 255) / 256, 256));
    for (int64_t b0 = 0; b0 < npts; b0 += 65535) {
        const int nb = (int)std::min<int64_t>(65535, npts - b0);
        hipLaunchKernelGGL(variance_rhs_kernel, dim3(blocks, nb), dim3(256), 0, (hipStream_t)stream_, dim, (int)mtot, M, h, x_new + b0 * dim,
                           (const double2*)ws, (double2*)rhs + b0 * M);
    }
    EFGP_HIP_CHECK(hipGetLastError());
    return EFGP_OK;
}

int efgp_variance_contract(int device, int dim, int64_t mtot, double h, const double* x_new, int64_t npts, const void* ws,
                           const void* gamma, double* out, void* stream_) {
    EFGP_REQUIRE(dim >= 1 && dim <= 3 && mtot >= 1 && npts >= 0, "efgp_variance_contract: bad sizes");
    if (npts == 0) return EFGP_OK;
    EFGP_REQUIRE(x_new && ws && gamma && out, "efgp_variance_contract: null argument");
    if (!device_ctx(device)) return EFGP_EHIP;
    DeviceGuard guard(device, (hipStream_t)stream_);
    int64_t M = 1;
    for (int a = 0; a < dim; ++a) M *= mtot;
    hipLaunchKernelGGL(variance_contract_kernel, dim3((unsigned)npts), dim3(256), 0, (hipStream_t)stream_, dim, (int)mtot, M, h, x_new,
                       (const double2*)ws, (const double2*)gamma, out);
    EFGP_HIP_CHECK(hipGetLastError());
    return EFGP_OK;
}


int efgp_spectral_weights(int device, int kind, int dim, double nu, double lengthscale, double variance, double c0, double h, int mtot,
                          void* ws, void* dprime, void* stream_) {
    EFGP_REQUIRE(ws, "efgp_spectral_weights: null output");
    EFGP_REQUIRE(kind == 0 || kind == 1, "efgp_spectral_weights: kernel kind %d not built in", kind);
    EFGP_REQUIRE(dim >= 1 && dim <= 3 && mtot >= 1 && (mtot & 1), "efgp_spectral_weights: bad grid");
    if (!device_ctx(device)) return EFGP_EHIP;
    DeviceGuard guard(device, (hipStream_t)stream_);
    int64_t M = 1;
    for (int a = 0; a < dim; ++a) M *= mtot;
    hipLaunchKernelGGL(spectral_weights_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, kind, dim, nu, lengthscale,
                       variance, c0, h, mtot, M, (double2*)ws, (double2*)dprime);
    EFGP_HIP_CHECK(hipGetLastError());
    return EFGP_OK;
}

}  // extern "C"
