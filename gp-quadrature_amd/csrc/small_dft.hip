// Direct DFTs for the tiny grids of the 2-D EFGP step (fine grid <= 128 x 128, <= 64 modes per dimension).
//
// At N = 1e6 the fit + mean step spends ~45 us in nine dependent small launches around the two N-scale kernels, each
// 4.4-5 us, i.e. launch latency, for ~1 MFLOP of work.  On the type-2 side (precorrect + two rocFFT kernels, 14 us) a
// separable dense DFT from the 23 x 23 modes to the 48 x 48 real fine grid is ONE launch of 5.4 us (and sums in a fixed
// order in double: at least as accurate as the FFT route).  The same idea on the type-1 side (int64 grid 96 x 96 x 2 ->
// two mode boxes, replacing reduce + two rocFFT kernels + deconvolve = 19 us) measured 16.7-21 us in one launch: every
// workgroup has to pull the 147-KB grid through L2 and walk 96-term sums serially -- not kept.
// Reference operation replaced: the mode placement / correction + FFT inside finufft type 2 (efgpnd.py:1533-1536).
#include "small_dft.hpp"

#include <algorithm>

namespace efgp {

constexpr int kDftMaxNf = 128;
constexpr int kDftThreads = 256;

// ---- type 2 side -------------------------------------------------------------------------------------------------
struct M2GArgs {
    const double2* f;
    const double2* mul;
    int nm0, nm1, modeord, isign;
    const double *fac0, *fac1;
    int nf0, nf1;
    double2* fine;
};

// One workgroup per fine-grid row x0.  The corrected modes c = fac f mul go to LDS first (coalesced, CMCL order);
// t[k1] = sum_k0 c[k0][k1] e(k0 x0) with all threads (k1 x a slice of k0, combined through LDS); then
// fine[x0][x1] = Re sum_k1 t[k1] e(k1 x1).  Phases advance by addition.
__global__ __launch_bounds__(kDftThreads) void modes_to_grid_real_kernel(M2GArgs a) {
    __shared__ double2 tw0[kDftMaxNf], tw1[kDftMaxNf], t[64], part[4][64];
    extern __shared__ double2 cm[];                                   // [nm0][nm1] corrected modes, CMCL order
    const int nf0 = a.nf0, nf1 = a.nf1, nm0 = a.nm0, nm1 = a.nm1;
    const double sgn = a.isign < 0 ? -2.0 : 2.0;
    for (int i = threadIdx.x; i < nf0; i += kDftThreads) {
        double s, c;
        sincospi(sgn * (double)i / (double)nf0, &s, &c);
        tw0[i] = make_double2(c, s);
    }
    for (int i = threadIdx.x; i < nf1; i += kDftThreads) {
        double s, c;
        sincospi(sgn * (double)i / (double)nf1, &s, &c);
        tw1[i] = make_double2(c, s);
    }
    for (int i = threadIdx.x; i < nm0 * nm1; i += kDftThreads) {
        const int s0 = i / nm1, s1 = i - s0 * nm1;
        const int k0 = s0 - nm0 / 2, k1 = s1 - nm1 / 2;
        const int slot0 = a.modeord == 0 ? s0 : (k0 >= 0 ? k0 : k0 + nm0);
        const int slot1 = a.modeord == 0 ? s1 : (k1 >= 0 ? k1 : k1 + nm1);
        const int64_t idx = (int64_t)slot0 * nm1 + slot1;
        double2 v = a.f[idx];
        if (a.mul) {
            const double2 m = a.mul[idx];
            v = make_double2(v.x * m.x - v.y * m.y, v.x * m.y + v.y * m.x);
        }
        const double fc = a.fac0[s0] * a.fac1[s1];
        cm[i] = make_double2(v.x * fc, v.y * fc);
    }
    __syncthreads();
    const int x0 = blockIdx.x;
    {   // thread = (k1, quarter of the k0 range)
        const int s1 = threadIdx.x & 63, qtr = threadIdx.x >> 6;
        if (s1 < nm1) {
            const int lo = (nm0 * qtr) / 4, hi = (nm0 * (qtr + 1)) / 4;
            int ph = (int)(((int64_t)(lo - nm0 / 2) * x0) % nf0);
            ph = ph < 0 ? ph + nf0 : ph;
            double re = 0.0, im = 0.0;
            for (int s0 = lo; s0 < hi; ++s0) {
                const double2 v = cm[s0 * nm1 + s1], w = tw0[ph];
                re += v.x * w.x - v.y * w.y;
                im += v.x * w.y + v.y * w.x;
                ph += x0;
                ph = ph >= nf0 ? ph - nf0 : ph;
            }
            part[qtr][s1] = make_double2(re, im);
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < nm1) {
        const int s1 = threadIdx.x;
        t[s1] = make_double2(part[0][s1].x + part[1][s1].x + part[2][s1].x + part[3][s1].x,
                             part[0][s1].y + part[1][s1].y + part[2][s1].y + part[3][s1].y);
    }
    __syncthreads();
    for (int x1 = threadIdx.x; x1 < nf1; x1 += kDftThreads) {
        int ph = (int)(((int64_t)(-(nm1 / 2)) * x1) % nf1);
        ph = ph < 0 ? ph + nf1 : ph;
        double re = 0.0;
        for (int s1 = 0; s1 < nm1; ++s1) {
            re += t[s1].x * tw1[ph].x - t[s1].y * tw1[ph].y;
            ph += x1;
            ph = ph >= nf1 ? ph - nf1 : ph;
        }
        a.fine[(int64_t)x0 * nf1 + x1] = make_double2(re, 0.0);
    }
}

bool modes_to_grid_real_eligible(int nf0, int nf1, int nm0, int nm1) {
    return nf0 <= kDftMaxNf && nf1 <= kDftMaxNf && nm0 <= 64 && nm1 <= 64 && std::getenv("EFGP_NO_DIRECT_DFT") == nullptr;
}

int modes_to_grid_real_launch(DeviceCtx* ctx, const double2* f, const double2* mul, int nm0, int nm1, int modeord, int isign,
                              const double* fac0, const double* fac1, int nf0, int nf1, double2* fine, hipStream_t stream) {
    (void)ctx;
    M2GArgs a{f, mul, nm0, nm1, modeord, isign, fac0, fac1, nf0, nf1, fine};
    const size_t lds = (size_t)nm0 * nm1 * sizeof(double2);
    if (lds > 32 * 1024) EFGP_HIP_CHECK(hipFuncSetAttribute((const void*)modes_to_grid_real_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(modes_to_grid_real_kernel, dim3(nf0), dim3(kDftThreads), lds, stream, a);
    EFGP_HIP_CHECK(hipGetLastError());
    return EFGP_OK;
}

}  // namespace efgp
