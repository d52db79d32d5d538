// M-scale device operations of the hyper-parameter gradient (reference: efgpnd_gradient_batched, efgpnd.py:17-317) behind the C ABI:
//   efgp_gradient_prepare    Jacobi diagonal v[0] |ws|^2 + sigma^2 and right-hand side ws .* F*y          (efgpnd.py:128-141)
//   efgp_gradient_assemble   every inner product of terms 1 and 2 and the final (term1 - term2) / 2       (efgpnd.py:155-176, :238-262)
// The reference (and this package's literal mode) does these with ~60 torch operations on M-length vectors and 0-dim tensors;
// at M = 529 each of them is a 3 us kernel behind a 10 us launch, and the gradient step was bound by the host enqueueing
// them.  Here the whole tail is two launches and nothing is read back.
//
// Adjoint form (see efgpnd.py of this package, trace_mode="adjoint"): with g = ws .* beta, T g the Toeplitz product,
//   fa       = (F*y - T g) / sigma^2                              (= F* alpha)
//   term2[i] = Re <fa, D'_i fa>                                   kernel hypers other than the variance
//   y.z = Re <F*y, g>,  |z|^2 = Re <g, T g>,  |alpha|^2 = (yy - 2 y.z + |z|^2) / sigma^4,  y.alpha = (yy - y.z) / sigma^2
//   term1[i] = (1/T) sum_t Re <F*Z_t, D'_i F*Z_t - ws .* B_t> / sigma^2       (B = solves of the trace systems)
//   noise    = N / sigma^2 - (1/T) sum_t Re <V_t, Bn_t> / sigma^2
//   variance entries from the noise entries as the reference does (:170-176, :254-258).
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "common.hpp"

namespace efgp {

namespace grad {

constexpr int kMaxH = 4;                 // kernel hyper-parameters (the reference's kernels have 2: lengthscale, variance)
constexpr int kMaxK = 4;                 // of which need a trace estimate
constexpr int kQ = kMaxH + 2 + kMaxK + 1;
constexpr int kQPad = 16;
constexpr int kThreads = 256;
constexpr int kMaxBlocks = 256;

struct Args {
    int64_t M;
    int T, H, K;
    int trace_idx[kMaxK];
    const double2* fy;
    const double2* tg;
    const double2* ws;
    const double2* beta;
    const double2* dprime;     // (M, H)
    const double2* fz;         // (T, M)
    const double2* beta_k;     // (K*T, M)
    const double* v;           // (T, M)
    const double2* beta_n;     // (T, M)
    double sig;
    double* partial;           // [blocks][kQPad]
};

template <int THREADS>
__device__ __forceinline__ void partial_body(const Args& a) {
    double acc[kQ];
#pragma unroll
    for (int q = 0; q < kQ; ++q) acc[q] = 0.0;
    for (int64_t m = (int64_t)blockIdx.x * THREADS + threadIdx.x; m < a.M; m += (int64_t)gridDim.x * THREADS) {
        const double2 fy = a.fy[m], tg = a.tg[m], w = a.ws[m], be = a.beta[m];
        const double2 g = make_double2(w.x * be.x - w.y * be.y, w.x * be.y + w.y * be.x);
        const double far = (fy.x - tg.x) / a.sig, fai = (fy.y - tg.y) / a.sig;
        const double fa2 = far * far + fai * fai;
        double2 dp[kMaxH];
#pragma unroll
        for (int i = 0; i < kMaxH; ++i) {
            dp[i] = i < a.H ? a.dprime[m * a.H + i] : make_double2(0.0, 0.0);
            acc[i] += dp[i].x * fa2;
        }
        acc[kMaxH] += fy.x * g.x + fy.y * g.y;
        acc[kMaxH + 1] += g.x * tg.x + g.y * tg.y;
        for (int t = 0; t < a.T; ++t) {
            const int64_t o = (int64_t)t * a.M + m;
            if (a.K > 0) {
                const double2 z = a.fz[o];
#pragma unroll
                for (int s = 0; s < kMaxK; ++s) {
                    if (s < a.K) {
                        double2 d = make_double2(0.0, 0.0);
#pragma unroll
                        for (int i = 0; i < kMaxH; ++i)
                            if (i == a.trace_idx[s]) d = dp[i];
                        const double2 b = a.beta_k[((int64_t)s * a.T + t) * a.M + m];
                        const double dr = (d.x * z.x - d.y * z.y) - (w.x * b.x - w.y * b.y);
                        const double di = (d.x * z.y + d.y * z.x) - (w.x * b.y + w.y * b.x);
                        acc[kMaxH + 2 + s] += z.x * dr + z.y * di;
                    }
                }
            }
            acc[kMaxH + 2 + kMaxK] += a.v[o] * a.beta_n[o].x;
        }
    }
    __shared__ double red[THREADS / 64][kQPad];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < kQ; ++q) {
        double v = acc[q];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0) red[wave][q] = v;
    }
    __syncthreads();
    if (threadIdx.x < kQ) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < THREADS / 64; ++w) v += red[w][threadIdx.x];
        a.partial[(int64_t)blockIdx.x * kQPad + threadIdx.x] = v;
    }
}

struct FinishArgs {
    int blocks, T, H, K, variance_idx;
    int trace_idx[kMaxK];
    const double* partial;
    double sig, n_obs, yy, variance;
    double* out;               // grad[H+1] | term1[H+1] | term2[H+1] | y.alpha
};

__device__ __forceinline__ void finish_body(const FinishArgs& a) {
    __shared__ double sum[kQPad];
    if (threadIdx.x < kQ) {
        double v = 0.0;
        for (int b = 0; b < a.blocks; ++b) v += a.partial[(int64_t)b * kQPad + threadIdx.x];     // fixed order: reproducible
        sum[threadIdx.x] = v;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const int nh = a.H + 1;
    double* grad = a.out;
    double* term1 = a.out + nh;
    double* term2 = a.out + 2 * nh;
    const double y_z = sum[kMaxH], z_z = sum[kMaxH + 1];
    const double a_norm = (a.yy - 2.0 * y_z + z_z) / (a.sig * a.sig);
    const double y_alpha = (a.yy - y_z) / a.sig;
    for (int i = 0; i < a.H; ++i) {
        term2[i] = sum[i];
        term1[i] = 0.0;
    }
    term2[a.H] = a_norm;
    for (int s = 0; s < a.K; ++s) term1[a.trace_idx[s]] = sum[kMaxH + 2 + s] / a.sig / (double)a.T;
    const double t1_noise = a.n_obs / a.sig - sum[kMaxH + 2 + kMaxK] / a.sig / (double)a.T;
    term1[a.H] = t1_noise;
    if (a.variance_idx >= 0) {
        term2[a.variance_idx] = (y_alpha - a.sig * a_norm) / a.variance;
        term1[a.variance_idx] = (a.n_obs - a.sig * t1_noise) / a.variance;
    }
    for (int i = 0; i < nh; ++i) grad[i] = 0.5 * (term1[i] - term2[i]);
    a.out[3 * nh] = y_alpha;
}

__global__ __launch_bounds__(kThreads) void partial_kernel(Args a) { partial_body<kThreads>(a); }
__global__ __launch_bounds__(64) void finish_kernel(FinishArgs a) { finish_body(a); }
// Small mode grids (the partial sums fit one workgroup's loop): both steps in ONE launch -- a dependent launch costs ~5 us, the
// sums of M = 529 entries a fraction of that.  Same arithmetic; the order of the partial sums is that of a one-block launch.
constexpr int kSmallThreads = 1024;
__global__ __launch_bounds__(kSmallThreads) void assemble_small_kernel(Args a, FinishArgs f) {
    partial_body<kSmallThreads>(a);
    __threadfence_block();
    __syncthreads();
    finish_body(f);
}

__global__ __launch_bounds__(256) void prepare_kernel(int64_t M, const double2* __restrict__ ws, const double2* __restrict__ fy,
                                                      const double2* __restrict__ v_center, double sig, double* __restrict__ diag,
                                                      double2* __restrict__ rhs) {
    const double c = v_center ? v_center->x : 0.0;
    for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (int64_t)gridDim.x * blockDim.x) {
        const double2 w = ws[m];
        if (diag) diag[m] = __dadd_rn(__dmul_rn(c, __dmul_rn(w.x, w.x) + __dmul_rn(w.y, w.y)), sig);
        if (rhs) {
            const double2 f = fy[m];
            rhs[m] = make_double2(w.x * f.x - w.y * f.y, w.x * f.y + w.y * f.x);
        }
    }
}

}  // namespace grad
}  // namespace efgp

using namespace efgp;

extern "C" {

int efgp_gradient_prepare(int device, int64_t nmodes, const void* ws, const void* fy, const void* v_center, double sigmasq, double* diag,
                          void* rhs, void* stream_) {
    EFGP_REQUIRE(ws && nmodes >= 1, "efgp_gradient_prepare: null / empty argument");
    EFGP_REQUIRE(diag || rhs, "efgp_gradient_prepare: nothing to compute");
    EFGP_REQUIRE(!rhs || fy, "efgp_gradient_prepare: rhs wanted but fy is null");
    EFGP_REQUIRE(!diag || v_center, "efgp_gradient_prepare: diag wanted but v_center is null");
    if (!device_ctx(device)) return EFGP_EHIP;
    DeviceGuard guard(device, (hipStream_t)stream_);
    const unsigned blocks = (unsigned)std::min<int64_t>(1024, (nmodes + 255) / 256);
    hipLaunchKernelGGL(grad::prepare_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, nmodes, (const double2*)ws, (const double2*)fy,
                       (const double2*)v_center, sigmasq, diag, (double2*)rhs);
    EFGP_HIP_CHECK(hipGetLastError());
    return EFGP_OK;
}

int efgp_gradient_assemble(int device, int64_t nmodes, int nprobes, int n_kernel_hypers, int variance_idx, int n_trace,
                           const int* trace_idx, const void* fy, const void* tg, const void* ws, const void* beta, const void* dprime,
                           const void* fz, const double* v, const void* beta_all, double sigmasq, double n_obs, double yy, double variance,
                           double* out, void* stream_) {
    using namespace grad;
    EFGP_REQUIRE(fy && tg && ws && beta && out && nmodes >= 1, "efgp_gradient_assemble: null / empty argument");
    EFGP_REQUIRE(n_kernel_hypers >= 0 && n_kernel_hypers <= kMaxH, "efgp_gradient_assemble: %d kernel hyper-parameters (at most %d)",
                 n_kernel_hypers, kMaxH);
    EFGP_REQUIRE(n_trace >= 0 && n_trace <= kMaxK && n_trace <= n_kernel_hypers, "efgp_gradient_assemble: bad n_trace %d", n_trace);
    EFGP_REQUIRE(n_kernel_hypers == 0 || dprime, "efgp_gradient_assemble: dprime is null");
    EFGP_REQUIRE(nprobes >= 1 && v && beta_all, "efgp_gradient_assemble: the noise term needs nprobes >= 1 probes and their solves");
    EFGP_REQUIRE(n_trace == 0 || (fz && trace_idx), "efgp_gradient_assemble: trace probes missing");
    EFGP_REQUIRE(variance_idx >= -1 && variance_idx < n_kernel_hypers, "efgp_gradient_assemble: bad variance_idx %d", variance_idx);
    EFGP_REQUIRE(sigmasq > 0.0 && (variance_idx < 0 || variance != 0.0), "efgp_gradient_assemble: sigmasq / variance must be positive");
    for (int s = 0; s < n_trace; ++s)
        EFGP_REQUIRE(trace_idx[s] >= 0 && trace_idx[s] < n_kernel_hypers && trace_idx[s] != variance_idx,
                     "efgp_gradient_assemble: bad trace_idx[%d] = %d", s, trace_idx[s]);
    DeviceCtx* ctx = device_ctx(device);
    if (!ctx) return EFGP_EHIP;
    DeviceGuard guard(device, (hipStream_t)stream_);
    hipStream_t stream = (hipStream_t)stream_;
    const int blocks = (int)std::min<int64_t>(kMaxBlocks, (nmodes + kThreads - 1) / kThreads);
    double* partial = (double*)scratch(ctx, SLOT_MISC, (size_t)kMaxBlocks * kQPad * sizeof(double));
    if (!partial) return EFGP_ENOMEM;
    Args a;
    a.M = nmodes;
    a.T = nprobes;
    a.H = n_kernel_hypers;
    a.K = n_trace;
    for (int s = 0; s < kMaxK; ++s) a.trace_idx[s] = s < n_trace ? trace_idx[s] : -1;
    a.fy = (const double2*)fy;
    a.tg = (const double2*)tg;
    a.ws = (const double2*)ws;
    a.beta = (const double2*)beta;
    a.dprime = (const double2*)dprime;
    a.fz = (const double2*)fz;
    a.beta_k = (const double2*)beta_all;
    a.v = v;
    a.beta_n = (const double2*)beta_all + (int64_t)n_trace * nprobes * nmodes;
    a.sig = sigmasq;
    a.partial = partial;
    const bool small = nmodes <= 4096 && std::getenv("EFGP_ASSEMBLE_TWO_LAUNCHES") == nullptr;
    if (!small) {
        hipLaunchKernelGGL(partial_kernel, dim3(blocks), dim3(kThreads), 0, stream, a);
        EFGP_HIP_CHECK(hipGetLastError());
    }
    FinishArgs f;
    f.blocks = small ? 1 : blocks;
    f.T = nprobes;
    f.H = n_kernel_hypers;
    f.K = n_trace;
    f.variance_idx = variance_idx;
    for (int s = 0; s < kMaxK; ++s) f.trace_idx[s] = a.trace_idx[s];
    f.partial = partial;
    f.sig = sigmasq;
    f.n_obs = n_obs;
    f.yy = yy;
    f.variance = variance;
    f.out = out;
    if (small) hipLaunchKernelGGL(assemble_small_kernel, dim3(1), dim3(kSmallThreads), 0, stream, a, f);
    else hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(64), 0, stream, f);
    EFGP_HIP_CHECK(hipGetLastError());
    return EFGP_OK;
}

}  // extern "C"
