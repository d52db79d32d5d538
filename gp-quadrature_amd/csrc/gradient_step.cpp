// One hyper-gradient step's device work in ONE library call (round 4).
//
// The adjoint estimator of efgpnd_gradient_batched (reference: efgpnd.py:17-317; this package: efgpnd.py::_gradient_tail_native) is a
// fixed sequence of ~15 entry points of this library -- weights, fused (F*y, Toeplitz vector) pass, operator, prepare, mean solve,
// T g, probe transforms, three scaled Toeplitz products, probe fill, batched solve, assemble.  Driven from Python each of them
// costs 5-15 us of interpreter, ctypes and tensor bookkeeping, ~200 us per step against 276 us of kernels -- and the step ends in
// a read-back, so the device idles whenever the host falls behind.  Here the same calls are made back to back from C++ on
// buffers carved out of one pooled block; nothing else changes (same kernels, same order, same arithmetic: the results agree
// with the Python-driven sequence for the same seeds as well as that sequence agrees with itself from run to run, ~4e-9 in the
// gradient -- the spreader's floating-point atomics; tests/test_gpu_gradient_step.py).
//
// Scope: one GPU (no shards), built-in kernels (the weights come from efgp_spectral_weights), generated probes, grids whose solves
// are single asynchronous launches (the persistent kernels: circulant grid within one workgroup).  Anything else returns
// EFGP_EUNSUPPORTED before any work is enqueued and the caller runs the Python-driven sequence.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.hpp"
#include "toeplitz_cg.hpp"

using namespace efgp;

namespace {

struct Carver {          // 256-byte aligned sub-blocks of one device allocation
    char* base = nullptr;
    size_t off = 0;
    template <typename T>
    T* take(size_t count) {
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += (count * sizeof(T) + 255) & ~size_t(255);
        return p;
    }
};

}  // namespace

extern "C" int efgp_gradient_step(efgp_points_t* points, int device, int dim, int64_t npts, const double* x, const double* y,
                                  double h, int mtot, int kind, double nu, double lengthscale, double variance, double c0,
                                  double sigmasq, double tol_pair, double tol_probe, double cg_tol, int early_stop, int nprobes,
                                  uint64_t probe_seed, uint64_t v_seed, int use_mean_pc, int use_trace_pc, int variance_idx,
                                  int n_trace, const int* trace_idx, const void* beta0, double n_obs, double yy, void* beta_out,
                                  double* out_vec, int* mean_iters_dev, int* trace_rows_dev, void* stream_) {
    EFGP_REQUIRE(x && y && beta_out && out_vec && mean_iters_dev && trace_rows_dev, "efgp_gradient_step: null argument");
    EFGP_REQUIRE(dim >= 1 && dim <= 3 && npts >= 1 && mtot >= 1 && (mtot & 1) && nprobes >= 1, "efgp_gradient_step: bad sizes");
    EFGP_REQUIRE(n_trace >= 0 && n_trace <= 4 && (n_trace == 0 || trace_idx), "efgp_gradient_step: bad trace indices");
    constexpr int H = 2;                                  // (lengthscale, variance): the built-in kernels
    DeviceCtx* ctx = device_ctx(device);
    if (!ctx) return EFGP_EHIP;
    hipStream_t stream = (hipStream_t)stream_;
    DeviceGuard guard(device, stream);
    const int T = nprobes, K = n_trace, R = (K + 1) * T;
    const int m = (mtot - 1) / 2;
    int64_t M = 1, Lv = 1, cells = 1;
    int64_t shape_y[3] = {1, 1, 1}, shape_v[3] = {1, 1, 1};
    for (int a = 0; a < dim; ++a) {
        shape_y[a] = mtot;
        shape_v[a] = 4 * m + 1;
        M *= mtot;
        Lv *= 4 * m + 1;
        cells *= next_pow2(4 * m + 1);
    }
    // the solves must be single asynchronous launches (persistent kernels): everything else keeps the Python-driven sequence
    if (cells > 4096 || (dim == 1 && 4 * m + 1 > 512)) {
        set_error("efgp_gradient_step: circulant grid beyond the single-launch solvers");
        return EFGP_EUNSUPPORTED;
    }
    // ---- one pooled block for every temporary ----------------------------------------------------------------------------
    Carver cv;
    auto layout = [&](Carver& c, double2*& ws, double2*& dp, double2*& fy, double2*& v, double2*& tg,
                      double2*& fz, double2*& dcol, double*& V, double2*& ball, double2*& betaall) {
        ws = c.take<double2>(M);
        dp = c.take<double2>(2 * M);
        fy = c.take<double2>(M);
        v = c.take<double2>(Lv);
        tg = c.take<double2>(M);
        fz = c.take<double2>((size_t)T * M);
        dcol = c.take<double2>(M);
        V = c.take<double>((size_t)T * M);
        ball = c.take<double2>((size_t)R * M);
        betaall = c.take<double2>((size_t)R * M);
    };
    double2 *ws, *dp, *fy, *v, *tg, *fz, *dcol, *ball, *betaall;
    double* V;
    layout(cv, ws, dp, fy, v, tg, fz, dcol, V, ball, betaall);
    const size_t bytes = cv.off;
    char* block = (char*)pool_alloc(ctx, bytes);
    if (!block) return EFGP_ENOMEM;
    cv = Carver{block, 0};
    layout(cv, ws, dp, fy, v, tg, fz, dcol, V, ball, betaall);

    // EFGP_STEP_TRACE=1: host time of every entry of the sequence (where the call's enqueue time goes), printed per call
    static const bool trace = std::getenv("EFGP_STEP_TRACE") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (!trace) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[efgp_gradient_step] %6.1f us  %s\n", std::chrono::duration<double, std::micro>(now - t_prev).count(), what);
        t_prev = now;
    };
    mark("pool block");
    efgp_nufft_t* plan = nullptr;
    efgp_nufft_t* plan_p = nullptr;
    efgp_toeplitz_t* top = nullptr;
    int rc = EFGP_OK;
    auto done = [&](int code) {
        if (plan_p && plan_p != plan) (void)efgp_nufft_destroy(plan_p);
        if (plan) (void)efgp_nufft_destroy(plan);
        mark("destroy plans");
        if (top) (void)efgp_toeplitz_destroy(top);
        mark("destroy operator");
        pool_free(ctx, block, bytes);       // stream-ordered reuse: everything that read it was enqueued on `stream` before
        return code;
    };
#define STEP(call)          \
    do {                    \
        rc = (call);        \
        if (rc != EFGP_OK) return done(rc); \
        mark(#call);        \
    } while (0)

    // 1) weights and their hyper-derivatives (efgpnd.py:95-99)
    STEP(efgp_spectral_weights(device, kind, dim, nu, lengthscale, variance, c0, h, mtot, ws, dp, stream));
    // 2) plans: the Toeplitz vector at tol_pair (F*y rides in that pass), the probes at the caller's tolerance (:186-189)
    if (points) STEP(efgp_nufft_create_on(&plan, points, nullptr, h, tol_pair));
    else STEP(efgp_nufft_create(&plan, device, dim, npts, x, nullptr, h, tol_pair));
    plan_p = plan;
    if (tol_probe > tol_pair) {
        plan_p = nullptr;
        if (points) STEP(efgp_nufft_create_on(&plan_p, points, nullptr, h, tol_probe));
        else STEP(efgp_nufft_create(&plan_p, device, dim, npts, x, nullptr, h, tol_probe));
    }
    // 3) F*y and the Toeplitz vector in one pass, the operator (:118-124)
    STEP(efgp_nufft_type1_pair(plan, y, shape_y, fy, shape_v, v, stream));
    STEP(efgp_toeplitz_create(&top, device, dim, shape_v, v, 1, stream));
    if (efgp_toeplitz_single_launch_solves(top) != 1) {
        set_error("efgp_gradient_step: the grid's solves are not single launches");
        return done(EFGP_EUNSUPPORTED);
    }
    // 4) mean solve and T g (:128-153).  The Jacobi diagonal v[0] |ws|^2 + sigma^2 and the right-hand side D F*y are formed inside
    // the solve kernels from the centre of the Toeplitz vector (a device scalar) and F*y: no launch of their own.
    int64_t centre = 0;
    for (int a = 0; a < dim; ++a) centre = centre * (4 * m + 1) + 2 * m;
    const double* dscale = reinterpret_cast<const double*>(v + centre);          // Re v[0]
    const int max_iter = (int)std::min<int64_t>(2 * M, 2000000000);
    STEP(efgp_internal_cg_single_launch(top, ws, sigmasq, 0, nullptr, use_mean_pc ? dscale : nullptr, fy, /*b_times_ws*/ 1, beta_out,
                                        /*zero_x0*/ beta0 ? 0 : 1, 1, cg_tol, max_iter, early_stop, 0, mean_iters_dev, stream, /*hermitian*/ 1,
                                        /*x0: the warm start is read where it lies*/ beta0));
    STEP(efgp_toeplitz_apply_scaled(top, beta_out, 0, 1, ws, nullptr, tg, stream));
    // 6) probes and the right-hand sides of the trace systems (:179-203)
    if (K > 0) {
        STEP(efgp_nufft_type1_rademacher(plan_p, probe_seed, 0, T, shape_y, 0, fz, stream));
        for (int s = 0; s < K; ++s) {
            // D'_i rides in the zero-padding as a diagonal read at stride H: column trace_idx[s] of the (M, H) array
            rc = efgp_internal_apply_scaled(top, fz, 0, T, dp + trace_idx[s], H, ws, ball + (size_t)s * T * M, stream);
            if (rc == EFGP_EUNSUPPORTED) {          // grids without the single-launch product: a contiguous copy of the column
                if (hipMemcpy2DAsync(dcol, sizeof(double2), dp + trace_idx[s], (size_t)H * sizeof(double2), sizeof(double2), (size_t)M,
                                     hipMemcpyDeviceToDevice, stream) != hipSuccess)
                    return done(EFGP_EHIP);
                rc = efgp_toeplitz_apply_scaled(top, fz, 0, T, dcol, ws, ball + (size_t)s * T * M, stream);
            }
            if (rc != EFGP_OK) return done(rc);
            mark("apply_scaled(F*Z, pre = D'_i, post = ws)");
        }
    }
    STEP(efgp_rademacher_fill(device, v_seed, 0, T, M, V, stream));
    STEP(efgp_toeplitz_apply_scaled(top, V, 1, T, ws, ws, ball + (size_t)K * T * M, stream));
    // 7) batched CG from zero (:205-236): the kernel starts from x = 0 itself
    STEP(efgp_internal_cg_single_launch(top, ws, sigmasq, 0, nullptr, use_trace_pc ? dscale : nullptr, ball, 0, betaall, /*zero_x0*/ 1, R,
                                        cg_tol, max_iter, early_stop, 1, trace_rows_dev, stream, /*hermitian*/ 0, nullptr));
    // 7.5 / 8) inner products and the final algebra (:155-176, :238-262)
    STEP(efgp_gradient_assemble(device, M, T, H, variance_idx, K, trace_idx, fy, tg, ws, beta_out, dp, K > 0 ? fz : nullptr, V, betaall,
                                sigmasq, n_obs, yy, variance, out_vec, stream));
#undef STEP
    return done(EFGP_OK);
}
