/* efgp_hip.h -- C ABI of libefgp_hip.so, the MI355X (gfx950) back end of the EFGP solve path.
 *
 * The reference (danbider/gp-quadrature) has no FFI of its own: its only compiled code on this
 * path is reached through `pytorch_finufft.functional.finufft_type1/type2` (efgpnd.py:1496-1499,
 * 1533-1536, 1546-1549, 1679) and `torch.fft.fftn/ifftn` (efgpnd.py:1284, 1370, 1379, 1661-1663),
 * and its Krylov loop is interpreted Python (cg.py:86-244).  The entry points below are what a
 * binding for that path binds instead; each cites the reference interface it replaces.
 *
 * Conventions
 *   - every function returns 0 on success, a negative EFGP_E* code on failure;
 *     efgp_last_error() returns a human-readable message for the last failure on this thread.
 *   - all data pointers are DEVICE pointers owned by the caller (e.g. torch allocations) unless a
 *     parameter is documented "host"; the library owns only plans and workspaces and frees them in
 *     *_destroy / efgp_release_workspaces.
 *   - complex = interleaved double pairs (complex128); real = double.
 *   - all work is enqueued on the hipStream_t passed as `stream` (NULL = default stream) and is
 *     asynchronous w.r.t. the host, except where an `*_out` HOST pointer is written (documented).
 *   - the objects of a device share scratch and pooled blocks: entry points take a per-device lock (host threads are serialised
 *     per device; the reference is single-threaded Python), and a device's work is single-stream at any moment: an entry point
 *     called on another stream than the previous one is ordered behind everything queued on that one (event hand-over;
 *     serialised, never a race).  efgp_release_workspaces must not run concurrently with other calls.
 */
#ifndef EFGP_HIP_H_
#define EFGP_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EFGP_OK 0
#define EFGP_EINVAL (-1)    /* bad argument (shape, dimension, null pointer) -> Python ValueError */
#define EFGP_EHIP (-2)      /* a HIP / hipFFT runtime call failed            -> Python RuntimeError */
#define EFGP_ENOMEM (-3)
#define EFGP_EUNSUPPORTED (-4)

typedef struct efgp_nufft_s efgp_nufft_t;
typedef struct efgp_toeplitz_s efgp_toeplitz_t;
typedef struct efgp_points_s efgp_points_t;
typedef struct efgp_comm_s efgp_comm_t;

/* ---- library ---------------------------------------------------------------------------- */
int efgp_version(void);                      /* 1000*major + minor */
const char* efgp_last_error(void);
/* frees cached FFT plans and scratch buffers of `device` (-1: all devices) */
int efgp_release_workspaces(int device);

/* ---- per-kernel timing (HIP events recorded on the launch stream around selected kernels) -----
 * enable != 0 starts recording (and clears previous records).  Names: "spread", "interp",
 * "cg_iteration" (one pad+FFT+multiply+FFT+update group).  efgp_kernel_timing_read synchronises the
 * device and returns the summed duration (ms) and launch count of `name` since enabling. */
int efgp_kernel_timing(int enable);
/* Restrict the timers to the launches of one name (NULL or "": all names): two event records per timed launch sit in the
 * stream, so a benchmark that wants one kernel's duration inside its timed region should not pay for the others. */
int efgp_kernel_timing_only(const char* name);
int efgp_kernel_timing_read(const char* name, double* total_ms_out, int64_t* launches_out);

/* ---- spreading-window parameters (host only; no GPU needed) ------------------------------
 * Exposed so the window selection can be unit-tested on a CPU-only machine. */
/* width w chosen for tolerance `tol` at upsampling ratio sigma = nf/n_modes */
int efgp_window_width(double tol, double sigma);
/* the same for a transform of dim dimensions (the window errors of the axes add up: a plan of dimension dim uses this width) */
int efgp_window_width_nd(double tol, double sigma, int dim);
/* evaluates the w window values a point at fine-grid position X (grid units) contributes:
 * first_cell_out = ceil(X - w/2), vals_out[j] = window at cell first+j, both via the same
 * Horner polynomials the device kernels use.  vals_out holds >= 16 doubles (host). */
int efgp_window_eval(double tol, double sigma, double X, int64_t* first_cell_out, double* vals_out,
                     int* w_out, double* beta_out);
/* fine-grid size a 1-D / 2-D plan uses for n_modes modes per axis at tolerance tol.  Up to 512 cells per axis (96 in 3-D)
 * sizes come from the ladder 32, 48, 64, 96, 128, 192, ... (2^k, 3 * 2^k): the first one >= 2 n_modes whose window is no
 * wider than the best 2^a3^b5^c size in [2, 2.5] n_modes would need; larger grids take that dense choice.  (A training
 * loop changes n_modes every few steps and every new FFT length is a runtime compilation in rocFFT.) */
int64_t efgp_fine_grid_size(int64_t n_modes, double tol);
/* the same choice for a plan of dimension dim in {1, 2, 3}; dense != 0: the 2-D rule of plans with >= 4e6 points (a grid whose
 * window is one cell narrower).  Always returns (the ladder search is bounded; host only). */
int64_t efgp_fine_grid_size_nd(int64_t n_modes, double tol, int dim, int dense);
/* out[i] = Fourier-side correction for CMCL mode i (host array of n_modes doubles) */
int efgp_window_deconv(double tol, int64_t nf, int64_t n_modes, double* out);

/* ---- NUFFT: replaces efgpnd.py NUFFT.__init__ / type1 / type2 -------------------------------
 * Points enter as the reference stores them, x (N,d) row-major float64; the phase is
 * phi = 2 pi h (x - xcen) (efgpnd.py:1451).  d in {1,2,3}.  `x` must stay valid for the plan's
 * lifetime (not copied, as in the reference).  tol = requested relative accuracy (FINUFFT `eps`). */
int efgp_nufft_create(efgp_nufft_t** plan_out, int device, int dim, int64_t npts, const double* x,
                      const double* xcen_host /* d doubles or NULL (=0) */, double h, double tol);
int efgp_nufft_destroy(efgp_nufft_t* plan);

/* ---- per-model point layout: replaces the reference's habit of keeping x, y on the model and rebuilding a NUFFT
 * object over the SAME points at every hyper-parameter step (EFGPND.__init__, efgpnd.py:342,386-387; NUFFT.__init__,
 * efgpnd.py:1432-1451; one per fit / gradient evaluation at :786 and :97-101).  The layout is made once per model:
 * bounding box, and -- for d = 2, lazily, when a plan first needs it -- copies of the points sorted by a key that
 * does not depend on the grid spacing (band of x_1, then x_0), which the type-1 pass streams and accumulates in MFMA
 * register tiles (csrc/spread_mfma.hip).  `x` (npts, dim) row-major must stay valid and unchanged for the layout's
 * lifetime; the layout must outlive the plans created on it.  Synchronises `stream` (host-side statistics). */
int efgp_points_create(efgp_points_t** pts_out, int device, int dim, int64_t npts, const double* x, void* stream);
int efgp_points_destroy(efgp_points_t* pts);
/* bounding box of the points (dim HOST doubles each): what the reference computes with x.min / x.max for the
 * domain length L (efgpnd.py:752-759) */
int efgp_points_bounds(efgp_points_t* pts, double* lo_out, double* hi_out);
/* Declares y (npts doubles, the model's targets, efgpnd.py:387) as this layout's value array: transforms on plans of
 * this layout that are handed this very pointer as strengths (efgp_nufft_type1_pair, single-row efgp_nufft_type1)
 * read a sorted copy kept by the layout, and max|y| (needed for the fixed-point accumulation) is computed here once
 * instead of per transform.  y must stay unchanged while attached (attach again after changing it; NULL detaches). */
int efgp_points_attach_values(efgp_points_t* pts, const double* y, void* stream);
/* efgp_nufft_create for the points of a layout: same plan, same entry points; d = 2 type-1 transforms with window
 * width <= 8 (tol >= ~6e-8) and enough points per fine-grid cell run on the sorted copies. */
int efgp_nufft_create_on(efgp_nufft_t** plan_out, efgp_points_t* pts, const double* xcen_host /* d doubles or NULL */,
                         double h, double tol);

/* type 1 (points -> modes), replaces pff.finufft_type1(phi, vals, out_shape, eps, isign, modeord)
 * at efgpnd.py:1496-1499:
 *     out[b, k] = sum_n c[b, n] exp(isign * i * k . phi_n)
 * c: (nbatch, npts), complex if c_is_complex else real (the reference casts real y / ones / +-1
 * probes to complex, efgpnd.py:1467-1468; the real entry avoids that traffic).  Real rows are transformed two per
 * complex grid (rows 2g, 2g+1) and separated by Hermitian symmetry: rows of one pair should have comparable
 * magnitude -- the smaller row inherits a relative error of ~1e-16 x the magnitude ratio (probes and the batches of
 * the EFGP path are +-1 or same-scale; efgp_nufft_type1_pair normalises its y channel for exactly this reason).
 * out: (nbatch, prod n_modes) complex, modes per dimension in CMCL order
 * -(n/2)..(n-1)/2 when modeord == 0, FFT order 0..,-.. when modeord == 1; last dimension fastest.
 * n_modes: d host int64. */
int efgp_nufft_type1(efgp_nufft_t* plan, const void* c, int c_is_complex, int nbatch,
                     const int64_t* n_modes, int isign, int modeord, void* out, void* stream);

/* type 1 of Rademacher probes generated inside the spread kernel (the +-1 probes the reference draws at
 * efgpnd.py:179-182 never have to exist in memory):
 *     out[b, k] = sum_n Z[b, n] exp(-i k . phi_n),   Z[b, n] = +-1 from a counter-based hash of
 *     (seed, b, n + index_offset)   (index_offset = global index of this shard's first point).
 * efgp_rademacher_fill writes the very same Z[b, n] to memory (nbatch x npts doubles) for callers that need
 * the probes themselves (tests, the reference-order code path). */
int efgp_nufft_type1_rademacher(efgp_nufft_t* plan, uint64_t seed, int64_t index_offset, int nbatch,
                                const int64_t* n_modes, int modeord, void* out, void* stream);
int efgp_rademacher_fill(int device, uint64_t seed, int64_t index_offset, int nbatch, int64_t npts, double* out,
                         void* stream);

/* Fused fit-time pass over the same points (efgpnd.py:786 and :789-790 / :1395-1421):
 *     out_y[k]    = sum_n y_n exp(-i k . phi_n),  k in the n_modes_y   box (CMCL order)
 *     out_ones[k] = sum_n     exp(-i k . phi_n),  k in the n_modes_one box (CMCL order)
 * one read of x and y.  Either output may be NULL to skip it. */
int efgp_nufft_type1_pair(efgp_nufft_t* plan, const double* y, const int64_t* n_modes_y, void* out_y,
                          const int64_t* n_modes_one, void* out_ones, void* stream);

/* type 2 (modes -> points), replaces pff.finufft_type2(phi, fk, eps, isign, modeord) at
 * efgpnd.py:1533-1536, 1546-1549 and (modeord=1) :1679:
 *     out[b, n] = sum_k f[b, k] exp(isign * i * k . phi_n)
 * f: (nbatch, prod n_modes) complex; out: (nbatch, npts) complex, or real (double) holding only
 * the real part when real_only != 0 (the reference takes .real at efgpnd.py:922 and :1679). */
int efgp_nufft_type2(efgp_nufft_t* plan, const void* f, int nbatch, const int64_t* n_modes, int isign,
                     int modeord, void* out, int real_only, void* stream);

/* Same, with the modes multiplied on the fly by a mode-shaped complex array shared by all batch rows:
 *     out[b, n] = sum_k mode_scale[k] f[b, k] exp(isign * i * k . phi_n)
 * i.e. the reference's F (ws * beta) at efgpnd.py:919-922 without materialising ws * beta. */
int efgp_nufft_type2_scaled(efgp_nufft_t* plan, const void* f, const void* mode_scale, int nbatch, const int64_t* n_modes,
                            int isign, int modeord, void* out, int real_only, void* stream);

/* ---- Toeplitz operator: replaces efgpnd.py ToeplitzND (:1239-1393) --------------------------
 * v: (L_1,...,L_d) complex, T[j,l] = v[j - l + (n-1)], n_a = (L_a+1)/2.  FFT length per dimension
 * is next_pow2(L_a) when force_pow2 (efgpnd.py:1269) else the next 2^a3^b5^c size >= L_a. */
int efgp_toeplitz_create(efgp_toeplitz_t** op_out, int device, int dim, const int64_t* Ls,
                         const void* v, int force_pow2, void* stream);
int efgp_toeplitz_destroy(efgp_toeplitz_t* op);
/* y[b] = T x[b]; x, y (nbatch, prod n_a) complex, may alias.  (ToeplitzND.__call__, :1331-1393) */
int efgp_toeplitz_apply(efgp_toeplitz_t* op, const void* x, int nbatch, void* y, void* stream);
/* y[b] = post .* T(pre .* x[b]): the products the hyper-gradient forms around T outside a solve -- T(ws .* beta),
 * ws .* T(D' F*Z), ws .* T(ws .* V) (efgpnd.py:150-153, :186-189, :203) -- without materialising the scaled vectors.
 * pre, post: (prod n_a) complex diagonals or NULL; x: (nbatch, prod n_a) complex, or real doubles when x_is_real;
 * y complex, must not alias x.  On the 2-D 64 x 64 circulant grid (n_0 = n_1 <= 32) this is ONE launch (a workgroup per
 * row, transforms in LDS); elsewhere pad / FFT / multiply / FFT / crop with the diagonals folded into pad and crop. */
int efgp_toeplitz_apply_scaled(efgp_toeplitz_t* op, const void* x, int x_is_real, int nbatch, const void* pre, const void* post,
                               void* y, void* stream);
/* FFT grid shape chosen (d host int64), for inspection (ToeplitzND.fft_shape) */
int efgp_toeplitz_fft_shape(efgp_toeplitz_t* op, int64_t* shape_out);
/* circulant grid the fused CG solves of this operator run on (d host int64): any F >= 2 n - 1 embeds the Toeplitz product exactly
 * (the reference's own alternative to next_pow2: efgpnd.py:1269-1271, `_next_fast_fft_size`), so the solvers take the smallest grid
 * their transforms cover -- 2-D: 48 x 48 (hermitian != 0, blocks <= 23 x 23) or 64 x 64 in one workgroup, 96 / 128 / 192 / 256 / 384 /
 * 512 per axis in the cooperative launch; otherwise the reference's grid.  fft_shape and efgp_toeplitz_apply are unaffected. */
int efgp_toeplitz_cg_shape(efgp_toeplitz_t* op, int hermitian, int64_t* shape_out);
/* 1 when every fused solve on this operator is ONE asynchronous launch with no grid barrier (the persistent kernels: circulant
 * grid within one workgroup), 0 otherwise (cooperative / multi-launch iterations, whose callers read row counts back). */
int efgp_toeplitz_single_launch_solves(efgp_toeplitz_t* op);

/* ---- preconditioned CG on G = D T D: replaces cg.py ConjugateGradients.solve() for the operators
 * of efgpnd.py:1572-1631 --------------------------------------------------------------------------
 * variant 0 (A_mean, :1593-1600):  A u = ws * T(ws * u) + sigmasq * u
 * variant 1 (A_var,  :1602-1609):  A u = ws * T(ws * u) / sigmasq + u
 * ws: (M) complex (the reference keeps the real weights in complex dtype).
 * precond_diag: (M) real Jacobi diagonal (z = r / diag, :1619-1631) or NULL for none.
 * b: (nbatch, M) complex right-hand sides; x: (nbatch, M) complex, holds x0 on entry and the
 * solution on exit.  batched_semantics = 0 follows cg.py:86-153 (requires nbatch == 1; convergence
 * tested before the preconditioner), 1 follows cg.py:155-244 (per-row masks, test after the p
 * update, extra |r| < 1e-12 exit, iteration count includes the terminating pass).
 * max_iter <= 0 means 2*M (cg.py:59-65).  iters_out (HOST int) receives `iters_completed`;
 * row_iters_out (HOST, nbatch ints, may be NULL) the per-row iteration counts.  The call
 * synchronises the stream before returning. */
int efgp_cg_solve(efgp_toeplitz_t* op, const void* ws, double sigmasq, int variant,
                  const double* precond_diag, const void* b, void* x, int nbatch, double tol,
                  int max_iter, int early_stop, int batched_semantics, int* iters_out,
                  int* row_iters_out, void* stream);

/* Same solve without any host synchronisation: the per-row iteration counts are left in the DEVICE array
 * row_iters_dev (nbatch ints) and the call returns as soon as the work is enqueued, so the host can prepare
 * the next fit while this one runs.  Available when the circulant grid fits the single-launch persistent
 * kernel (power-of-two FFT sizes, <= 4608 padded elements) and on 2-D grids of 128^2..512^2 (cooperative
 * launches: G workgroups per system with grid barriers, all resident at once; a row whose barrier could not
 * complete because other work held the CUs reports -3 iterations and keeps x0 -- efgp_cg_solve retries such
 * rows itself); otherwise returns EFGP_EUNSUPPORTED and the caller uses efgp_cg_solve. */
int efgp_cg_solve_async(efgp_toeplitz_t* op, const void* ws, double sigmasq, int variant,
                        const double* precond_diag, const void* b, void* x, int nbatch, double tol,
                        int max_iter, int early_stop, int batched_semantics, int* row_iters_dev, void* stream);

/* Host-side scalar work of the quadrature grid for the built-in kernels (kind 0 squared exponential, 1 Matern with nu in
 * {1/2, 3/2, 5/2}); no device involved.  efgp_grid_bounds: the two bisections of get_xis(use_integral=True)
 * (utils/kernels.py:28-69, 94-105): *ltime_out with k(L) = eps, *lfreq_out with r^(d-1) S(r) / S(0) = trunc_eps, found with
 * the operations of the reference's Python expressions in the same order (bit-identical bounds, hence identical h and
 * mtot).  c0 = (2 pi l^2)^(d/2) * variance (SE) or variance * scaling(l) (Matern), s0 = S(0): formed by the caller.
 * efgp_spectral_weights_host: ws = sqrt(S(xi) h^d) on the (mtot)^d tensor grid xi = h * (-m..m)^d as complex numbers
 * (efgpnd.py:766-780) and, when dprime_out is not NULL, h^d * (dS/dlengthscale, dS/dvariance) (complex, (M, 2)). */
int efgp_grid_bounds(int kind, int dim, double nu, double lengthscale, double variance, double c0, double s0, double eps,
                     double trunc_eps, double* ltime_out, double* lfreq_out);
int efgp_spectral_weights_host(int kind, int dim, double nu, double lengthscale, double variance, double c0, double h, int mtot,
                               double* ws_out, double* dprime_out);
/* The same weights computed ON the device into ws (M complex128) and, when not NULL, dprime ((M, 2) complex128): one small launch
 * on `stream` instead of a host computation plus a staged upload. */
int efgp_spectral_weights(int device, int kind, int dim, double nu, double lengthscale, double variance, double c0, double h, int mtot,
                          void* ws, void* dprime, void* stream);

/* ---- M-scale tail of the hyper-parameter gradient: replaces the torch glue of efgpnd_gradient_batched ----------------
 * (efgpnd.py:128-141, :155-176, :238-262; the adjoint form of this package: every N-length inner product of the reference
 * evaluated in feature space).  All arrays live on `device`; nothing is read back.
 *
 * efgp_gradient_prepare: diag[k] = Re(*v_center) |ws[k]|^2 + sigmasq (the Jacobi diagonal, :128-133; v_center points at
 * the centre element of the Toeplitz vector ON THE DEVICE) and rhs[k] = ws[k] fy[k] (:141).  diag or rhs may be NULL. */
int efgp_gradient_prepare(int device, int64_t nmodes, const void* ws, const void* fy, const void* v_center, double sigmasq, double* diag,
                          void* rhs, void* stream);
/* efgp_gradient_assemble: with g = ws .* beta and tg = T g,
 *     fa = (fy - tg) / sigmasq;  term2[i] = Re<fa, dprime[:, i] fa>;  y.z = Re<fy, g>;  |z|^2 = Re<g, tg>;
 *     |alpha|^2 = (yy - 2 y.z + |z|^2) / sigmasq^2;  y.alpha = (yy - y.z) / sigmasq;                         (:155-176)
 *     term1[trace_idx[s]] = (1/T) sum_t Re<fz[t], dprime[:, trace_idx[s]] fz[t] - ws .* beta_all[s T + t]> / sigmasq;
 *     term1[noise] = n_obs / sigmasq - (1/T) sum_t Re<v[t], beta_all[n_trace T + t]> / sigmasq;               (:238-262)
 *     the variance entries from the noise entries as the reference does; out = grad | term1 | term2 | y.alpha with
 *     grad = (term1 - term2) / 2, each of length n_kernel_hypers + 1 (3 (n_kernel_hypers + 1) + 1 doubles, DEVICE).
 * fy, tg, ws, beta: (M) complex; dprime: (M, n_kernel_hypers) complex; fz: (T, M) complex transforms of the data-space
 * probes (may be NULL when n_trace = 0); v: (T, M) real feature-space probes; beta_all: ((n_trace + 1) T, M) complex
 * solves; trace_idx: HOST array of n_trace hyper indices; variance_idx: index of the variance hyper or -1.
 * n_kernel_hypers, n_trace <= 4.  Two launches: per-workgroup partial sums, then one workgroup adds them in a fixed
 * order (reproducible) and does the scalar algebra; ONE launch (a single workgroup does both) when nmodes <= 4096. */
int efgp_gradient_assemble(int device, int64_t nmodes, int nprobes, int n_kernel_hypers, int variance_idx, int n_trace,
                           const int* trace_idx, const void* fy, const void* tg, const void* ws, const void* beta, const void* dprime,
                           const void* fz, const double* v, const void* beta_all, double sigmasq, double n_obs, double yy, double variance,
                           double* out, void* stream);

/* One whole hyper-gradient step of the adjoint estimator in one call: replaces the body of efgpnd_gradient_batched
 * (efgpnd.py:95-262) for the built-in kernels on one GPU -- efgp_spectral_weights | plans | efgp_nufft_type1_pair |
 * efgp_toeplitz_create | efgp_gradient_prepare | efgp_cg_solve_hermitian_async | T g | efgp_nufft_type1_rademacher |
 * efgp_toeplitz_apply_scaled x (n_trace + 1) | efgp_rademacher_fill | efgp_cg_solve_async | efgp_gradient_assemble, enqueued
 * back to back on `stream` with temporaries from the device's block pool (same kernels, same order, same arithmetic as driving
 * those entry points one by one).  Nothing is read back.
 *   points: layout of the model's points (y attached) or NULL (then x (npts, dim) is used directly); y: npts DEVICE doubles;
 *   h, mtot: the quadrature grid (efgp_quadrature_grid); kind, nu, lengthscale, variance, c0: as efgp_spectral_weights;
 *   tol_pair / tol_probe: NUFFT tolerances of the (F*y, Toeplitz vector) pass and of the probe transforms;
 *   probe_seed / v_seed: counters of the data-space / feature-space +-1 probes; trace_idx: HOST array;
 *   beta0: warm start of the mean solve ((mtot^dim) complex, DEVICE) or NULL for zero;
 *   beta_out: (mtot^dim) complex mean coefficients; out_vec: 3 (2 + 1) + 1 doubles as efgp_gradient_assemble;
 *   mean_iters_dev: 1 int, trace_rows_dev: (n_trace + 1) nprobes ints (iteration counts, DEVICE).
 * EFGP_EUNSUPPORTED when the grid's solves are not single launches (see efgp_toeplitz_single_launch_solves): the caller
 * then drives the entry points itself. */
int efgp_gradient_step(efgp_points_t* points, int device, int dim, int64_t npts, const double* x, const double* y, double h, int mtot,
                       int kind, double nu, double lengthscale, double variance, double c0, double sigmasq, double tol_pair,
                       double tol_probe, double cg_tol, int early_stop, int nprobes, uint64_t probe_seed, uint64_t v_seed,
                       int use_mean_pc, int use_trace_pc, int variance_idx, int n_trace, const int* trace_idx, const void* beta0,
                       double n_obs, double yy, void* beta_out, double* out_vec, int* mean_iters_dev, int* trace_rows_dev,
                       void* stream);

/* efgp_cg_solve_async for systems whose vectors are Fourier coefficients of REAL functions on the symmetric mode grid:
 * every right-hand side and start vector satisfies u[-k] = conj u[k], ws is real and even, the Toeplitz vector comes
 * from real weights.  All CG systems of an EFGP model are of this kind (right-hand sides are D F* of real vectors:
 * y, Rademacher probes, kernel-derivative terms; efgpnd.py:186-189, 792, 1657).  On the 2-D 64 x 64 circulant grid
 * (odd mtot <= 31) the operator then runs on real transforms -- rows k0 >= 0 only, two real columns per complex
 * transform: 88 line transforms per application instead of 174 -- with the same recurrences and stopping rules; on every
 * other grid this is efgp_cg_solve_async.  A right-hand side that is not Hermitian to 1e-8 |b| is refused: its
 * row_iters_dev entry is -2 and its solution NaN.  precond_diag must be even as well (it is: a function of |ws|). */
int efgp_cg_solve_hermitian_async(efgp_toeplitz_t* op, const void* ws, double sigmasq, int variant,
                                  const double* precond_diag, const void* b, void* x, int nbatch, double tol,
                                  int max_iter, int early_stop, int batched_semantics, int* row_iters_dev, void* stream);

/* efgp_cg_solve_async / efgp_cg_solve_hermitian_async (hermitian != 0) from the start vector ZERO: x is output only -- no fill
 * by the caller -- and the initial operator application (A 0 = 0) is skipped: one launch and, on the cooperative 128^2..512^2
 * grids, one operator application less per solve.  The reference's solves of the variance and of the trace estimator all start
 * from zero (cg.py:86-101 with x0 = None; efgpnd.py:205-236). */
int efgp_cg_solve_from_zero_async(efgp_toeplitz_t* op, const void* ws, double sigmasq, int variant, const double* precond_diag,
                                  const void* b, void* x, int nbatch, double tol, int max_iter, int early_stop, int batched_semantics,
                                  int hermitian, int* row_iters_dev, void* stream);

/* efgp_cg_solve (synchronous, every grid) under the same promise as efgp_cg_solve_hermitian_async.  On 3-D circulant grids of
 * 64..256 per dimension (odd mtot) the multi-launch iteration then carries the planes k0 >= 0 only: half the lines in every
 * pass, two real columns per complex transform along dim 0, a REAL centred spectrum (cg3h_* kernels; BASELINE configs[4]:
 * 66 MB per operator application instead of 150 MB); same recurrences, stopping rules and iteration counts (cg.py:86-244).
 * Every other grid: efgp_cg_solve.  Data that break the promise (right-hand side or start vector not conjugate-even to
 * 1e-8, ws not real and even) are refused with EFGP_EINVAL before anything is written. */
int efgp_cg_solve_hermitian(efgp_toeplitz_t* op, const void* ws, double sigmasq, int variant,
                            const double* precond_diag, const void* b, void* x, int nbatch, double tol,
                            int max_iter, int early_stop, int batched_semantics, int* iters_out,
                            int* row_iters_out, void* stream);

/* The complex FFT behind the transforms above, on a caller's device array: `batch` contiguous row-major arrays of extents
 * n[0..rank), in place, double precision; forward != 0: exp(-i...), else the unnormalised inverse.  use_rocfft = 0: the
 * in-house line kernels (line_fft.hip: lengths 2^a 3^b 5^c <= 4096 per axis, no run-time compilation; EFGP_EUNSUPPORTED
 * otherwise), 1: hipFFT.  Replaces torch.fft.fftn / ifftn of ToeplitzND (efgpnd.py:1275-1290, 1331-1393) and the FFT stage of
 * finufft (efgpnd.py:1395-1421); exported for tests, benchmarks and integrators who want the same transform. */
int efgp_fft_c2c(int device, int rank, const long long* n, long long batch, void* data, int forward, int use_rocfft, void* stream);

/* The fit's mean system in one launch, straight from the transform outputs (efgpnd.py:792-803):
 *     (D T D + sigmasq I) beta = D fy,   D = diag(ws),   beta_0 = 0,
 * Jacobi diagonal (*diag_scale_dev) * |ws|^2 + sigmasq when diag_scale_dev is not NULL (device pointer to the
 * real centre value of the Toeplitz vector, efgpnd.py:795-799), single-system stopping rule of cg.py:116-150.
 * Replaces the reference's separate ws*Fy, |ws|^2 diagonal and zeros(x0) tensor ops plus efgp_cg_solve_async.
 * EFGP_EUNSUPPORTED when the grid does not fit the single-launch kernel (callers fall back to efgp_cg_solve).
 * CONTRACT: fy is the type-1 transform of REAL strengths (fy[-k] = conj fy[k]), ws is real and even and the operator's
 * Toeplitz vector comes from real weights -- true for every EFGP model (efgpnd.py:786-790).  On the 64 x 64 circulant
 * grid the solve then runs on real transforms (half the lines, see efgp_cg_solve_hermitian_async); an input that breaks
 * the contract is refused: iters_dev[0] = -2 and beta = NaN.  2-D grids of 128..512 per dimension: the cooperative launch of
 * efgp_cg_solve_hermitian_async with the same in-kernel right-hand side, diagonal and zero start (iters_dev[0] = -3 and NaN
 * when its grid barrier could not get the workgroups resident together: solve again through efgp_cg_solve). */
int efgp_cg_solve_mean_async(efgp_toeplitz_t* op, const void* ws, double sigmasq, const double* diag_scale_dev, const void* fy,
                             void* x, double tol, int max_iter, int early_stop, int* iters_dev, void* stream);

/* Lanczos three-term recurrence on A (variant as in efgp_cg_solve), the inner loop of the reference's stochastic
 * Lanczos quadrature log-determinant (logdet_slq, efgpnd.py:1716-1738):
 *     q_0 = z / |z|;  v = A q_k - beta_{k-1} q_{k-1};  alpha_k = Re<q_k, v>;  v -= alpha_k q_k;  beta_k = |v|;
 *     stop after `steps` steps or when beta_k < 1e-12;  q_{k+1} = v / beta_k
 * for nprobes start vectors z (nprobes, M) complex, one workgroup per probe, all steps inside ONE launch (the reference
 * loops in Python).  alpha_dev, beta_dev: (nprobes, steps) doubles; norm2_dev: (nprobes) |z|^2 or NULL;
 * steps_taken_dev: (nprobes) ints -- all DEVICE arrays; no host synchronisation.  EFGP_EUNSUPPORTED when the circulant
 * grid does not fit the single-launch kernel (callers then loop over efgp_toeplitz_apply). */
int efgp_lanczos(efgp_toeplitz_t* op, const void* ws, double sigmasq, int variant, const void* z, int nprobes, int steps,
                 double* alpha_dev, double* beta_dev, double* norm2_dev, int* steps_taken_dev, void* stream);

/* Diagnostic hook for iterate-level parity with cg.py:132 (`norm(r) / (norm(b) + div_eps) < tol`): while a device
 * buffer is registered, every solve enqueued by efgp_cg_solve* writes row 0's relative residual of iteration i
 * (1-based, the value the stopping rule tests) to history_dev[i - 1], i <= capacity.  NULL (or capacity 0) switches
 * it off.  Process-wide; not meant for production use. */
int efgp_cg_record_history(double* history_dev, int capacity);

/* ---- N-length reductions of the hyper-gradient (efgpnd.py:163, 170, 239) ----------------------
 * out_host[0] = Re sum_n conj(a_n) b_n over n < count; each operand is complex (interleaved) when
 * its *_is_complex flag is set, else real.  Wavefront-shuffle + LDS reduction, deterministic order.
 * Synchronises the stream (the result is a HOST double). */
int efgp_vdot_real(int device, const void* a, int a_is_complex, const void* b, int b_is_complex,
                   int64_t count, double* out_host, void* stream);

/* ---- M-scale pieces of the predictive variance (efgpnd.py:1634-1679, 1805-1820) -------------------------------------
 * Lag sums of the stochastic variance (diag_sums_nd, :1660-1664):
 *     out[r] = (1/nprobes) sum_j sum_{k - l = r} gamma[j, k] eta[j, l],   r in [-(mtot-1), mtot-1]^d,
 * stored in FFT order per dimension (index r mod (2 mtot - 1)), exactly the tensor the reference builds with
 * fftn/ifftn of size 2 mtot - 1 and feeds to the FFT-ordered type-2 transform (:1679 -> efgp_nufft_type2, modeord 1).
 * gamma: (nprobes, mtot^d) complex, eta: (nprobes, mtot^d) real (+-1 probes), out: ((2 mtot - 1)^d) complex. */
int efgp_lag_sums(int device, int dim, int64_t mtot, const void* gamma, const double* eta, int nprobes, void* out, void* stream);
/* 'regular' variance (compute_prediction_variance, :1805-1820): explicit feature rows f_k(x*) = exp(2 pi i h k . x*) on the
 * (mtot,)^d mode box (k = -(mtot-1)/2 .. (mtot-1)/2 per dimension, last dimension fastest).
 *   efgp_variance_rhs:      rhs[b, k] = ws[k] * conj(f_k(x*_b))                       (right-hand sides of the A_var solves)
 *   efgp_variance_contract: out[b]    = max(0, Re sum_k f_k(x*_b) ws[k] gamma[b, k])  (gamma = the solutions)
 * x_new: (npts, dim) doubles; ws: (mtot^d) complex; rhs / gamma: (npts, mtot^d) complex; out: (npts) doubles. */
int efgp_variance_rhs(int device, int dim, int64_t mtot, double h, const double* x_new, int64_t npts, const void* ws, void* rhs,
                      void* stream);
int efgp_variance_contract(int device, int dim, int64_t mtot, double h, const double* x_new, int64_t npts, const void* ws,
                           const void* gamma, double* out, void* stream);

/* ---- collectives of the point-sharded fit (one process per GPU, RCCL over xGMI) ------------------------------------
 * The reference has no distributed code; sharding the N observation points needs exactly these sums between the spread
 * pass and the replicated solve: the gridded partials F*y and v (efgpnd.py:118-124 / 786-790), the batched F*Z of the
 * gradient (:186-189) and its N-length scalars (:163, 170, 239), plus MIN/MAX of the coordinates for the domain length
 * (:752-759) and a broadcast for state that is drawn at random (probe seeds, feature-space probes).
 * efgp_comm_unique_id: HOST buffer of 128 bytes, filled on rank 0 and carried to the other ranks by the caller.
 * All buffers are DEVICE pointers; calls are stream-ordered (no host synchronisation). */
int efgp_comm_unique_id(void* id_out_128_bytes);
int efgp_comm_init(efgp_comm_t** comm_out, int device, int rank, int world_size, const void* unique_id_128_bytes);
int efgp_comm_allreduce_sum(efgp_comm_t* comm, double* buf, size_t n_doubles, void* stream);
int efgp_comm_allreduce_minmax(efgp_comm_t* comm, double* buf, size_t n_doubles, int take_max, void* stream);
int efgp_comm_broadcast(efgp_comm_t* comm, void* buf, size_t nbytes, int root, void* stream);
int efgp_comm_destroy(efgp_comm_t* comm);

#ifdef __cplusplus
}
#endif
#endif /* EFGP_HIP_H_ */
