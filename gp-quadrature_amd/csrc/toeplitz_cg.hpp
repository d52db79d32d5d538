// Shared declarations of the Toeplitz / CG translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace efgp {

struct ToepGeom {
    int d;
    int64_t n[3];     // block size per dimension (ns)
    int64_t F[3];     // FFT size per dimension
    int64_t M;        // prod n
    int64_t Ftot;     // prod F
};

// Lanczos mode of the persistent kernels (see pcg::Args)
struct LanczosOut {
    int steps;
    double* alpha;     // [rows][steps]
    double* beta;      // [rows][steps]
    double* norm2;     // [rows] |b|^2, may be null
};

// operands of cg_herm48_kernel: the operator's spectrum on the 48 x 48 circulant grid (FFT of the zero-padded Toeplitz vector
// / 2304, natural order) and exp(-2 pi i q / 48), q < 48
struct Herm48Operands {
    const double2* vhat;
    const double2* tw;
};

// single-launch CG with the FFT in LDS (cg_persistent.hip)
bool persistent_cg_eligible(const ToepGeom& g);
int persistent_cg_launch(const ToepGeom& g, const double2* const* twiddles, const double2* vhat, const double2* ws,
                         const double* diag, double sigmasq, int variant, double tol, int early_stop, int batched,
                         int max_iter, const double2* b, double2* x, int rows, int* d_iters, hipStream_t stream,
                         const double* diag_scale = nullptr, int b_times_ws = 0, int zero_x0 = 0, const LanczosOut* lz = nullptr,
                         int hermitian = 0 /* b, x0 and the Toeplitz vector are coefficient arrays of real functions */,
                         const Herm48Operands* h48 = nullptr /* 2-D blocks <= 23 x 23: Hermitian solves run on the 48 x 48 grid */,
                         const double2* x0 = nullptr /* start vectors when they are not in x (read before x is written) */);

// spectrum of the Toeplitz vector on the 64 x 64 circulant grid in one launch (cg_persistent.hip)
bool toeplitz_vhat_fused_eligible(const ToepGeom& g);
int toeplitz_vhat_fused_launch(const double2* v, int L0, int L1, double factor, double2* vhat, hipStream_t stream);
// the 64 x 64 spectrum (vhat64, may be null) and the 48 x 48 spectrum in one launch, each divided by its grid size
int toeplitz_vhat_pair_launch(const double2* v, int L0, int L1, double2* vhat64, double2* vhat48, hipStream_t stream);
// forward 64 x 64 transforms of nbatch zero-padded L0 x L1 arrays (complex or real), one workgroup each (cg_persistent.hip)
int fft2d64_batch_launch(const void* src, int src_is_real, int64_t src_stride, int L0, int L1, double2* dst, int64_t dst_stride,
                         int nbatch, hipStream_t stream);

// y[row] = post .* T(pre .* x[row]) on the 64 x 64 circulant grid in one launch (cg_persistent.hip)
bool toeplitz_apply_fused_eligible(const ToepGeom& g);
int toeplitz_apply_fused_launch(const ToepGeom& g, const double2* tw64, const double2* vhat, const double2* pre, const double2* post,
                                const void* x, int x_is_real, double2* y, int rows, hipStream_t stream, int pre_stride = 1);

}  // namespace efgp

// Internal entry points for callers inside the library (gradient_step.cpp): the exported solves / products with the optional
// operands of the persistent kernels that the C ABI does not expose.
struct efgp_toeplitz_s;
extern "C" {
// (C linkage like their exported siblings, hidden: not part of the ABI)
// efgp_toeplitz_apply_scaled with `pre` read at a stride (a column of a row-major (M, H) array: stride H).  Returns
// EFGP_EUNSUPPORTED -- nothing enqueued -- when pre_stride != 1 and the grid has no single-launch product.
__attribute__((visibility("hidden"))) int efgp_internal_apply_scaled(efgp_toeplitz_s* op, const void* x, int x_is_real, int nbatch, const void* pre, int pre_stride,
                               const void* post, void* y, hipStream_t stream);
// efgp_cg_solve_async / efgp_cg_solve_hermitian_async on a grid of the persistent kernels with: diag_scale (device scalar; Jacobi
// diagonal (*diag_scale) |ws|^2 + sigmasq formed in the kernel, used when diag is null), b_times_ws (right-hand side ws .* b),
// zero_x0 (x is output only), x0 (start vectors kept apart from x: no copy; NULL = in place).  EFGP_EUNSUPPORTED on every other grid.
__attribute__((visibility("hidden"))) int efgp_internal_cg_single_launch(efgp_toeplitz_s* op, const void* ws, double sigmasq, int variant, const double* diag,
                                   const double* diag_scale, const void* b, int b_times_ws, void* x, int zero_x0, int nbatch, double tol,
                                   int max_iter, int early_stop, int batched_semantics, int* row_iters_dev, hipStream_t stream,
                                   int hermitian, const void* x0);
}  // extern "C"
