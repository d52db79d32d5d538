// d-dimensional Toeplitz mat-vec by circulant embedding (hipFFT/rocFFT) and the Jacobi-preconditioned
// CG solver of the EFGP normal equations, for gfx950.
//
// Replaces ToeplitzND (efgpnd.py:1239-1393), create_Gv/A_mean/A_var/jacobi (efgpnd.py:1572-1631) and
// ConjugateGradients.solve (cg.py:86-244).
//
// Data layout in HBM
//   vhat   [prod F]   complex: FFT of the zero-padded Toeplitz vector, pre-divided by prod F
//   pad    [rows][prod F] complex: zero-padded operand / FFT work buffer (library scratch)
//   x,r,p,Ap [rows][M] complex, M = prod n; per-row scalars (rz, |b|, flags) in a small scratch array
// The whole iteration state lives on the device; the host only launches kernels and polls a status
// word every few iterations.  Per iteration and row: one pad+scale kernel, two batched FFTs, one
// spectral multiply, one fused update kernel (crop, A p, <p,Ap>, x/r update, norms, p update).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>
#include <utility>
#include <vector>

#include <cstdlib>

#include "common.hpp"
#include "es_kernel.hpp"
#include "line_fft.hpp"
#include "toeplitz_cg.hpp"

namespace efgp {

// EFGP_CG_TRACE=1: report host-side spans of the multi-kernel solve that take longer than 2 ms (stderr)
struct TraceSpan {
    const char* what;
    std::chrono::steady_clock::time_point t0;
    bool on;
    explicit TraceSpan(const char* w) : what(w), on(std::getenv("EFGP_CG_TRACE") != nullptr) {
        if (on) t0 = std::chrono::steady_clock::now();
    }
    ~TraceSpan() {
        if (!on) return;
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms > 2.0) std::fprintf(stderr, "[efgp cg trace] %s: %.1f ms\n", what, ms);
    }
};


constexpr int kVecThreads = 256;
constexpr int kCgThreads = 512;
constexpr double kDivEps = 1e-16;   // cg.py:57

__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// index of block element `flat` (row-major over n) inside the padded FFT grid, shifted by `off` per dim
__device__ __forceinline__ int64_t pad_index(const ToepGeom& g, int64_t flat, int64_t off_mul) {
    int64_t idx = 0, stride = 1;
    for (int a = g.d - 1; a >= 0; --a) {
        int64_t ia = flat % g.n[a];
        flat /= g.n[a];
        idx += (ia + off_mul * (g.n[a] - 1)) * stride;
        stride *= g.F[a];
    }
    return idx;
}

// the same in 32-bit arithmetic (grids of the CG path have far fewer than 2^31 cells): a third of the ALU work
__device__ __forceinline__ int pad_index32(const ToepGeom& g, int flat, int off_mul) {
    int idx = 0, stride = 1;
    for (int a = g.d - 1; a >= 0; --a) {
        const int na = (int)g.n[a];
        const int q = flat / na;
        const int ia = flat - q * na;
        flat = q;
        idx += (ia + off_mul * (na - 1)) * stride;
        stride *= (int)g.F[a];
    }
    return idx;
}

// pad[row][:] = 0 except the leading n-box which receives scale .* src[rows[row]]
// (scale may be null), times the real `factor`.  One launch covers all rows: grid = (blocks, nrows).
__global__ void pad_scale_kernel(ToepGeom g, const double2* __restrict__ src, int64_t src_stride,
                                 const double2* __restrict__ scale, const int* __restrict__ rows,
                                 const int* __restrict__ row_active, double2* __restrict__ pad, double factor) {
    const int slot = blockIdx.y;
    const int row = rows ? rows[slot] : slot;
    if (row < 0) return;
    if (row_active && !row_active[row]) return;
    double2* P = pad + (int64_t)slot * g.Ftot;
    const double2* S = src + (int64_t)row * src_stride;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < g.Ftot; t += (int64_t)gridDim.x * blockDim.x) {
        int64_t rem = t, flat = 0;
        bool inside = true;
        // decompose t over F (row-major) and test against n
        int64_t coords[3];
        for (int a = g.d - 1; a >= 0; --a) {
            coords[a] = rem % g.F[a];
            rem /= g.F[a];
            inside = inside && coords[a] < g.n[a];
        }
        double2 v = make_double2(0.0, 0.0);
        if (inside) {
            for (int a = 0; a < g.d; ++a) flat = flat * g.n[a] + coords[a];
            v = S[flat];
            if (scale) v = cmul(v, scale[flat]);
            v.x *= factor;
            v.y *= factor;
        }
        P[t] = v;
    }
}

__global__ void spectral_mul_kernel(int64_t Ftot, const double2* __restrict__ vhat, double2* __restrict__ pad) {
    double2* P = pad + (int64_t)blockIdx.y * Ftot;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < Ftot; t += (int64_t)gridDim.x * blockDim.x)
        P[t] = cmul(P[t], vhat[t]);
}

// y[row][flat] = pad[row][flat index shifted by n-1]
__global__ void crop_kernel(ToepGeom g, const double2* __restrict__ pad, double2* __restrict__ y) {
    const int row = blockIdx.y;
    const double2* P = pad + (int64_t)row * g.Ftot;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < g.M; t += (int64_t)gridDim.x * blockDim.x)
        y[(int64_t)row * g.M + t] = P[pad_index(g, t, 1)];
}

// efgp_toeplitz_apply_scaled on grids without the single-launch kernel: the diagonals ride in the pad and crop passes
// pad[row] = 0 except the leading n-box = pre .* x[row] (x real or complex)
template <bool REAL>
__global__ void pad_pre_kernel(ToepGeom g, const void* __restrict__ src, const double2* __restrict__ pre, double2* __restrict__ pad) {
    const int row = blockIdx.y;
    double2* P = pad + (int64_t)row * g.Ftot;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < g.Ftot; t += (int64_t)gridDim.x * blockDim.x) {
        int64_t rem = t, flat = 0, coords[3];
        bool inside = true;
        for (int a = g.d - 1; a >= 0; --a) {
            coords[a] = rem % g.F[a];
            rem /= g.F[a];
            inside = inside && coords[a] < g.n[a];
        }
        double2 v = make_double2(0.0, 0.0);
        if (inside) {
            for (int a = 0; a < g.d; ++a) flat = flat * g.n[a] + coords[a];
            if (REAL) v = make_double2(((const double*)src)[(int64_t)row * g.M + flat], 0.0);
            else v = ((const double2*)src)[(int64_t)row * g.M + flat];
            if (pre) v = cmul(v, pre[flat]);
        }
        P[t] = v;
    }
}
// y[row] = post .* window of pad[row]
__global__ void crop_post_kernel(ToepGeom g, const double2* __restrict__ pad, const double2* __restrict__ post, double2* __restrict__ y) {
    const int row = blockIdx.y;
    const double2* P = pad + (int64_t)row * g.Ftot;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < g.M; t += (int64_t)gridDim.x * blockDim.x) {
        double2 v = P[pad_index(g, t, 1)];
        if (post) v = cmul(v, post[t]);
        y[(int64_t)row * g.M + t] = v;
    }
}

// ------------------------------------------------------------------------------------------
// CG state
// ------------------------------------------------------------------------------------------
struct CgRowScalars {   // one per row
    double rz;
    double den;        // |b| or 1
    double pAp;        // <p, A p> + div_eps of the current iteration (written by the last block of cg_dot_kernel)
    double beta;       // p <- z + beta p is applied by the NEXT iteration's cg_pad_kernel when do_p is set
    int active;
    int iters;         // iterations in which this row was updated
    int do_p;
    int pad_;
};

constexpr int kCgBlocksMax = 256;    // workgroups per system in the multi-kernel update (partials per reduction; <= block size)

struct CgArgs {
    ToepGeom g;
    const double2* ws;
    const double* diag;       // may be null
    double sigmasq;
    int variant;              // 0 A_mean, 1 A_var
    double tol;
    int early_stop;
    int batched;              // convergence-test placement (cg.py single vs batched)
    const double2* b;
    double2* x;
    double2* r;
    double2* p;
    double2* ap;              // A p of the current iteration (written by cg_dot_kernel, read by cg_axpy_kernel)
    const double2* pad;       // inverse-FFT output, row slot = blockIdx.y
    double2* pad_out;         // the same buffer, written by cg_pad_kernel
    const int* rows;          // slot -> row (null: identity)
    CgRowScalars* sc;
    int* status;              // [0] = number of active rows (recomputed by host), [1] = any-change flag
    double* partial;          // [rows][3][kCgBlocksMax] per-workgroup partial sums
    int* counter;             // [rows][2] arrival counters of the two reductions
    int nblk;                 // workgroups per system
    double* hist;             // diagnostic: row 0's |r_i| / |b| per iteration (efgp_cg_record_history) or null
    int hist_cap;
    // the part of every vector the update kernels walk: [v_off, v_off + v_len) of the M entries; entries below v_off + v_w1
    // count once in the dot products, the others twice.  General systems: (0, M, M).  Hermitian 3-D systems (round 3): the
    // planes k0 >= 0 -- (h0 n1 n2, (h0 + 1) n1 n2, n1 n2): plane k0 = 0 once, every other plane for itself and its mirror image.
    int64_t v_off, v_len, v_w1;
};

__device__ __forceinline__ double block_sum(double v, double* red) {
    // wave reduce (64 lanes) then across waves through LDS
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    double t = 0.0;
    const int nw = blockDim.x >> 6;
    for (int i = 0; i < nw; ++i) t += red[i];     // same order in every thread: deterministic
    return t;
}

// A u for block element t given the cropped Toeplitz product Tu = T(ws*u)[t]
__device__ __forceinline__ double2 apply_A(const CgArgs& a, double2 wst, double2 Tu, double2 u) {
    double2 g = cmul(wst, Tu);
    if (a.variant == 0) return make_double2(g.x + a.sigmasq * u.x, g.y + a.sigmasq * u.y);
    return make_double2(g.x / a.sigmasq + u.x, g.y / a.sigmasq + u.y);
}

// ---- one CG iteration (cg.py:116-150 / :193-241), spread over nblk workgroups per system ----------------
// A system with M up to ~2e5 unknowns is far too long for one workgroup (measured: 60 us per iteration at
// M = 12167), so the iteration is three launches over (nblk, slots) grids; dot products are reduced in two
// deterministic levels: fixed-order sums inside a workgroup, then the LAST workgroup to arrive (device-scope
// arrival counter) adds the nblk partials in index order and publishes the scalars.  The direction update
// p <- z + beta p is deferred into the next iteration's pad kernel, which needs ws .* p anyway.
__device__ __forceinline__ double load_agent(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // bypasses this CU's L1
}
__device__ __forceinline__ void store_agent(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);         // write-through, leaves no dirty L2 line behind
}
// Arrival at a reduction: returns true in every thread of the LAST of `nblk` workgroups to arrive at `counter`.
// Thread 0 has stored this workgroup's partial sums with store_agent; it drains them (s_waitcnt vmcnt(0)) before its arrival,
// and the last arriver reads all partials with load_agent (L1-bypassing): the hand-off form R1 of MI355X_MICROARCH.md, no
// fence.  Round 3: the __threadfence() pair that stood here cost EVERY workgroup an L2 write-back, serialised per XCD --
// 15-18 of cg3_inv2_kernel's 23 us at 207 workgroups, ~9 of cg_axpy_kernel's 14 us at 64.  Everything else these kernels write
// is read by later launches only (kernel boundary).
__device__ __forceinline__ bool arrive_count(int* counter, int nblk, int* flag) {
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int prev = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = prev == nblk - 1;
        if (*flag) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // all arrivals of this launch are in
    }
    __syncthreads();
    return *flag != 0;
}
__device__ __forceinline__ bool arrive_last(const CgArgs& a, int row, int which, int* flag) {
    return arrive_count(&a.counter[2 * row + which], a.nblk, flag);
}
__device__ __forceinline__ void block_range(const CgArgs& a, int64_t& lo, int64_t& hi) {
    const int64_t per = (a.v_len + a.nblk - 1) / a.nblk, end = a.v_off + a.v_len;
    lo = a.v_off + (int64_t)blockIdx.x * per;
    hi = lo + per < end ? lo + per : end;
}

// r = b - A x0, z = r/diag, p = z, rz = <r,z>, den = |b| (cg.py:94-111 / :164-186).  grid = (nblk, rows): round 3 -- with ONE
// workgroup per system (as before) a 3-D system of 185 k unknowns (BASELINE configs[4] at eps 1e-3) spent 630 us here, 11 % of
// a hyper-gradient step; the two sums are reduced like the iteration's (fixed order inside a workgroup, partials added in index
// order by the last workgroup to arrive).
__global__ __launch_bounds__(kVecThreads) void cg_init_kernel(CgArgs a) {
    __shared__ double red[kVecThreads / 64];
    __shared__ int flag;
    const int row = blockIdx.y;
    const double2* P = a.pad + (int64_t)row * a.g.Ftot;
    const int64_t base = (int64_t)row * a.g.M;
    int64_t lo, hi;
    block_range(a, lo, hi);
    double rz = 0.0, bb = 0.0;
    for (int64_t t = lo + threadIdx.x; t < hi; t += kVecThreads) {
        const int64_t o = base + t;
        double2 x0 = a.x[o];
        double2 Ax = apply_A(a, a.ws[t], P[pad_index(a.g, t, 1)], x0);
        double2 bv = a.b[o];
        double2 rv = make_double2(bv.x - Ax.x, bv.y - Ax.y);
        double2 zv = rv;
        if (a.diag) {
            zv.x = rv.x / a.diag[t];
            zv.y = rv.y / a.diag[t];
        }
        a.r[o] = rv;
        a.p[o] = zv;
        rz += rv.x * zv.x + rv.y * zv.y;
        bb += bv.x * bv.x + bv.y * bv.y;
    }
    rz = block_sum(rz, red);
    bb = block_sum(bb, red);
    double* part = a.partial + (int64_t)row * 3 * kCgBlocksMax;
    if (threadIdx.x == 0) {
        store_agent(&part[kCgBlocksMax + blockIdx.x], rz);
        store_agent(&part[2 * kCgBlocksMax + blockIdx.x], bb);
    }
    if (!arrive_last(a, row, 1, &flag)) return;
    __shared__ double fin[2 * kCgBlocksMax];
    if (threadIdx.x < a.nblk) {
        fin[threadIdx.x] = load_agent(&part[kCgBlocksMax + threadIdx.x]);
        fin[kCgBlocksMax + threadIdx.x] = load_agent(&part[2 * kCgBlocksMax + threadIdx.x]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double srz = 0.0, sbb = 0.0;
        for (int i = 0; i < a.nblk; ++i) {
            srz += fin[i];
            sbb += fin[kCgBlocksMax + i];
        }
        const double bn = sqrt(sbb);
        a.sc[row].rz = srz;
        a.sc[row].den = bn > 0.0 ? bn : 1.0;
        a.sc[row].active = 1;
        a.sc[row].iters = 0;
        a.sc[row].pAp = 1.0;
        a.sc[row].beta = 0.0;
        a.sc[row].do_p = 0;
        a.sc[row].pad_ = 0;
    }
}

// pad[slot] = zero-padded ws .* p, after the deferred direction update p <- r/diag + beta p
__global__ __launch_bounds__(kVecThreads) void cg_pad_kernel(CgArgs a) {
    const int slot = blockIdx.y;
    const int row = a.rows ? a.rows[slot] : slot;
    if (row < 0) return;
    const CgRowScalars sc = a.sc[row];
    if (!sc.active) return;
    const ToepGeom& g = a.g;
    double2* P = a.pad_out + (int64_t)slot * g.Ftot;
    const int64_t base = (int64_t)row * g.M;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < g.Ftot; t += (int64_t)gridDim.x * blockDim.x) {
        int64_t rem = t, flat = 0;
        bool inside = true;
        int64_t coords[3];
        for (int q = g.d - 1; q >= 0; --q) {
            coords[q] = rem % g.F[q];
            rem /= g.F[q];
            inside = inside && coords[q] < g.n[q];
        }
        double2 v = make_double2(0.0, 0.0);
        if (inside) {
            for (int q = 0; q < g.d; ++q) flat = flat * g.n[q] + coords[q];
            double2 pv = a.p[base + flat];
            if (sc.do_p) {
                double2 zv = a.r[base + flat];
                if (a.diag) {
                    zv.x /= a.diag[flat];
                    zv.y /= a.diag[flat];
                }
                pv = make_double2(zv.x + sc.beta * pv.x, zv.y + sc.beta * pv.y);
                a.p[base + flat] = pv;
            }
            v = cmul(pv, a.ws[flat]);
        }
        P[t] = v;
    }
}

// A p and <p, A p>
__global__ __launch_bounds__(kVecThreads) void cg_dot_kernel(CgArgs a) {
    __shared__ double red[kVecThreads / 64];
    __shared__ int flag;
    const int slot = blockIdx.y;
    const int row = a.rows ? a.rows[slot] : slot;
    if (row < 0) return;
    if (!a.sc[row].active) return;
    const double2* P = a.pad + (int64_t)slot * a.g.Ftot;
    const int64_t base = (int64_t)row * a.g.M;
    int64_t lo, hi;
    block_range(a, lo, hi);
    double pAp = 0.0;
    for (int64_t t = lo + threadIdx.x; t < hi; t += kVecThreads) {
        const double2 pv = a.p[base + t];
        const double2 Ap = apply_A(a, a.ws[t], P[pad_index32(a.g, (int)t, 1)], pv);
        a.ap[base + t] = Ap;
        pAp += pv.x * Ap.x + pv.y * Ap.y;
    }
    pAp = block_sum(pAp, red);
    double* part = a.partial + (int64_t)row * 3 * kCgBlocksMax;
    if (threadIdx.x == 0) store_agent(&part[blockIdx.x], pAp);
    if (arrive_last(a, row, 0, &flag)) {
        __shared__ double fin[kCgBlocksMax];
        if (threadIdx.x < a.nblk) fin[threadIdx.x] = load_agent(&part[threadIdx.x]);     // loads in parallel
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int i = 0; i < a.nblk; ++i) t += fin[i];                                    // summed in index order
            a.sc[row].pAp = t + kDivEps;
        }
    }
}

// x += alpha p, r -= alpha A p, |r|^2, <r, r/diag>; the last workgroup decides convergence and beta
__global__ __launch_bounds__(kVecThreads) void cg_axpy_kernel(CgArgs a) {
    __shared__ double red[kVecThreads / 64];
    __shared__ int flag;
    const int slot = blockIdx.y;
    const int row = a.rows ? a.rows[slot] : slot;
    if (row < 0) return;
    const CgRowScalars sc = a.sc[row];
    if (!sc.active) return;
    const int64_t base = (int64_t)row * a.g.M;
    int64_t lo, hi;
    block_range(a, lo, hi);
    const double alpha = sc.rz / sc.pAp;
    double rr = 0.0, rz_new = 0.0;
    for (int64_t t = lo + threadIdx.x; t < hi; t += kVecThreads) {
        const double2 pv = a.p[base + t];
        const double2 Ap = a.ap[base + t];
        double2 xv = a.x[base + t];
        double2 rv = a.r[base + t];
        xv.x += alpha * pv.x;
        xv.y += alpha * pv.y;
        rv.x -= alpha * Ap.x;
        rv.y -= alpha * Ap.y;
        a.x[base + t] = xv;
        a.r[base + t] = rv;
        const double q = (t - a.v_off < a.v_w1 ? 1.0 : 2.0) * (rv.x * rv.x + rv.y * rv.y);
        rr += q;
        rz_new += a.diag ? q / a.diag[t] : q;
    }
    rr = block_sum(rr, red);
    rz_new = block_sum(rz_new, red);
    double* part = a.partial + (int64_t)row * 3 * kCgBlocksMax;
    if (threadIdx.x == 0) {
        store_agent(&part[kCgBlocksMax + blockIdx.x], rr);
        store_agent(&part[2 * kCgBlocksMax + blockIdx.x], rz_new);
    }
    if (!arrive_last(a, row, 1, &flag)) return;
    __shared__ double fin[2 * kCgBlocksMax];
    if (threadIdx.x < a.nblk) {
        fin[threadIdx.x] = load_agent(&part[kCgBlocksMax + threadIdx.x]);
        fin[kCgBlocksMax + threadIdx.x] = load_agent(&part[2 * kCgBlocksMax + threadIdx.x]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double srr = 0.0, srz = 0.0;
        for (int i = 0; i < a.nblk; ++i) {
            srr += fin[i];
            srz += fin[kCgBlocksMax + i];
        }
        const double rnorm = sqrt(srr);
        if (a.hist && row == 0 && sc.iters < a.hist_cap) a.hist[sc.iters] = rnorm / (sc.den + kDivEps);
        const bool conv = a.early_stop && ((rnorm / (sc.den + kDivEps) < a.tol) || (a.batched && rnorm < 1e-12));
        CgRowScalars out = sc;
        out.iters = sc.iters + 1;
        if (conv) {                          // single: stop before the p update (cg.py:132); batched: after it
            out.active = 0;                  // (cg.py:229-241) -- p is never used again either way
            out.do_p = 0;
            atomicAdd(&a.status[1], 1);
        } else {
            out.beta = srz / (sc.rz + kDivEps);
            out.do_p = 1;
        }
        if (!(conv && !a.batched)) out.rz = srz;
        a.sc[row] = out;
    }
}

// ==========================================================================================================
// Fused line-FFT iteration for 2-D grids beyond one CU's LDS (F = 128..512 per dimension).
//   The generic iteration above spends 11 launches (pad, 2x2 rocFFT passes, multiply, update kernels) and moves
//   the whole padded F x F grid through HBM five times although only n of its F rows are non-zero on the way in
//   and only n rows / columns are kept on the way out.  Here a matvec is three launches of batched in-LDS line
//   transforms with the pruning built in:
//     cg_rows_fwd_kernel : (deferred p update) ws .* p, zero-padded, forward FFT along dim 1 of the n rows -> B1[n][F]
//     cg_cols_mid_kernel : per group of columns: forward FFT along dim 0 (n non-zero inputs), .* vhat, inverse FFT,
//                          rows of the crop window -> B2[n][F]
//     cg_rows_inv_kernel : inverse FFT along dim 1 of the n window rows, crop, A p, <p, A p> (two-level reduction)
//   followed by cg_axpy_kernel.  Lines are transformed with a Stockham autosort FFT (radix 4, one radix-2 stage
//   for odd log2 F) in two LDS ping-pong buffers; twiddles come from the per-length table exp(-2 pi i q / F).
// ==========================================================================================================
constexpr int kLineThreads = 256;

__device__ __forceinline__ int ilog2(int v) { return 31 - __clz(v); }

// forward FFT of `nl` lines of length F (a power of two) held in A (line l at A + l*ld); result in the returned
// buffer (A or B).  `tw` = exp(-2 pi i q / F), q < F, in LDS.
__device__ __forceinline__ double2* line_fft(double2* A, double2* B, int F, int ld, int nl, const double2* tw) {
    double2* x = A;
    double2* y = B;
    int Ns = 1;
    while (Ns < F) {
        // radix 8 for the LATER stages where the remaining length allows it (a first radix-8 stage writes with a stride of 8
        // elements: 8-way bank conflicts): a stage is one LDS round trip and one barrier, and a radix-8 butterfly gives
        // its thread 8 independent loads (radix-4 stages are bound by the LDS latency of dependent work items)
        const int rem = F / Ns;
        const int R = (F >= 256 && (rem == 512 || rem == 64 || rem == 8)) ? 8 : ((rem & 3) == 0 ? 4 : 2);   // 256 = 4 8 8, 512 = 8 8 8
        // (measured, cooperative solve: 512^2 49.8 -> 40.4 us per iteration, 256^2 neutral, 128^2 = 4 4 8 slower than 4 4 4 2)
        const int per = F / R;                       // butterflies per line (a power of two)
        const int lper = ilog2(per);
        const int mult = F / (Ns * R);               // twiddle index step: angle = -2 pi r k / (Ns R)
        for (int w = threadIdx.x; w < (nl << lper); w += kLineThreads) {
            const int l = w >> lper, j = w & (per - 1);
            const int k = j & (Ns - 1);
            const double2* xl = x + l * ld;
            double2* yl = y + l * ld;
            const int j0 = (j - k) * R + k;
            if (R == 8) {
                double2 v[8];
#pragma unroll
                for (int m = 0; m < 8; ++m) v[m] = xl[j + m * per];
                if (Ns > 1) {
                    const int q = k * mult;          // 7 q < F: no wrap
#pragma unroll
                    for (int m = 1; m < 8; ++m) v[m] = cmul(v[m], tw[m * q]);
                }
                // DFT-8: two DFT-4 (even / odd inputs), odd outputs times w8^m, combine
                const double2 e0 = make_double2(v[0].x + v[4].x, v[0].y + v[4].y), e1 = make_double2(v[0].x - v[4].x, v[0].y - v[4].y);
                const double2 e2 = make_double2(v[2].x + v[6].x, v[2].y + v[6].y), e3 = make_double2(v[2].y - v[6].y, v[6].x - v[2].x);
                const double2 E0 = make_double2(e0.x + e2.x, e0.y + e2.y), E1 = make_double2(e1.x + e3.x, e1.y + e3.y);
                const double2 E2 = make_double2(e0.x - e2.x, e0.y - e2.y), E3 = make_double2(e1.x - e3.x, e1.y - e3.y);
                const double2 o0 = make_double2(v[1].x + v[5].x, v[1].y + v[5].y), o1 = make_double2(v[1].x - v[5].x, v[1].y - v[5].y);
                const double2 o2 = make_double2(v[3].x + v[7].x, v[3].y + v[7].y), o3 = make_double2(v[3].y - v[7].y, v[7].x - v[3].x);
                const double2 O0 = make_double2(o0.x + o2.x, o0.y + o2.y), O1r = make_double2(o1.x + o3.x, o1.y + o3.y);
                const double2 O2r = make_double2(o0.x - o2.x, o0.y - o2.y), O3r = make_double2(o1.x - o3.x, o1.y - o3.y);
                const double hh = 0.70710678118654752440;
                const double2 O1 = make_double2(hh * (O1r.x + O1r.y), hh * (O1r.y - O1r.x));
                const double2 O2 = make_double2(O2r.y, -O2r.x);
                const double2 O3 = make_double2(hh * (O3r.y - O3r.x), -hh * (O3r.x + O3r.y));
                yl[j0] = make_double2(E0.x + O0.x, E0.y + O0.y);
                yl[j0 + Ns] = make_double2(E1.x + O1.x, E1.y + O1.y);
                yl[j0 + 2 * Ns] = make_double2(E2.x + O2.x, E2.y + O2.y);
                yl[j0 + 3 * Ns] = make_double2(E3.x + O3.x, E3.y + O3.y);
                yl[j0 + 4 * Ns] = make_double2(E0.x - O0.x, E0.y - O0.y);
                yl[j0 + 5 * Ns] = make_double2(E1.x - O1.x, E1.y - O1.y);
                yl[j0 + 6 * Ns] = make_double2(E2.x - O2.x, E2.y - O2.y);
                yl[j0 + 7 * Ns] = make_double2(E3.x - O3.x, E3.y - O3.y);
            } else if (R == 4) {
                double2 v0 = xl[j], v1 = xl[j + per], v2 = xl[j + 2 * per], v3 = xl[j + 3 * per];
                if (Ns > 1) {
                    const int q = k * mult;          // 3 q < F: no wrap
                    v1 = cmul(v1, tw[q]);
                    v2 = cmul(v2, tw[2 * q]);
                    v3 = cmul(v3, tw[3 * q]);
                }
                const double2 t0 = make_double2(v0.x + v2.x, v0.y + v2.y), t1 = make_double2(v0.x - v2.x, v0.y - v2.y);
                const double2 t2 = make_double2(v1.x + v3.x, v1.y + v3.y);
                const double2 t3 = make_double2(v1.y - v3.y, v3.x - v1.x);          // -i (v1 - v3)
                yl[j0] = make_double2(t0.x + t2.x, t0.y + t2.y);
                yl[j0 + Ns] = make_double2(t1.x + t3.x, t1.y + t3.y);
                yl[j0 + 2 * Ns] = make_double2(t0.x - t2.x, t0.y - t2.y);
                yl[j0 + 3 * Ns] = make_double2(t1.x - t3.x, t1.y - t3.y);
            } else {
                double2 v0 = xl[j], v1 = xl[j + per];
                if (Ns > 1) v1 = cmul(v1, tw[k * mult]);
                yl[j0] = make_double2(v0.x + v1.x, v0.y + v1.y);
                yl[j0 + Ns] = make_double2(v0.x - v1.x, v0.y - v1.y);
            }
        }
        __syncthreads();
        double2* t = x;
        x = y;
        y = t;
        Ns *= R;
    }
    return x;
}

// ---- in-wave line transform (lengths 128, 256, 512) ---------------------------------------------------------------------
// A line of F = 64 R points (R = 2, 4, 8) is transformed by 8 R lanes of ONE wave, 8 points per lane, as R interleaved
// 64-point transforms (x[R m + r], r < R: radix 8 x 8 inside 8 adjacent lanes, as in cg_persistent.hip) followed by one
// radix-R combine over r:   X[k1 + 64 k2] = sum_r w_F^(r k1) w_R^(r k2) Y_r[k1].
// Three LDS exchanges, each written and read by the line's own lanes (wave fences, no workgroup barrier), all with
// conflict-free patterns: exchange 1 (inside the 64-point transforms) XOR-swizzled in the destination line, exchange 2
// in [r][k1] order in the source line (its contents are in registers by then), the result in natural order in the
// destination line.  ~230 fp64 instructions and 48 LDS accesses per lane for 8 points, no index arithmetic in loops.
// The generic Stockham stages above cost 1.5-2.5 k cycles per work item (measured in the cooperative solve).
__device__ __forceinline__ void dft8_inplace(double2 (&v)[8]) {
    const double2 e0 = make_double2(v[0].x + v[4].x, v[0].y + v[4].y), e1 = make_double2(v[0].x - v[4].x, v[0].y - v[4].y);
    const double2 e2 = make_double2(v[2].x + v[6].x, v[2].y + v[6].y), e3 = make_double2(v[2].y - v[6].y, v[6].x - v[2].x);
    const double2 E0 = make_double2(e0.x + e2.x, e0.y + e2.y), E1 = make_double2(e1.x + e3.x, e1.y + e3.y);
    const double2 E2 = make_double2(e0.x - e2.x, e0.y - e2.y), E3 = make_double2(e1.x - e3.x, e1.y - e3.y);
    const double2 o0 = make_double2(v[1].x + v[5].x, v[1].y + v[5].y), o1 = make_double2(v[1].x - v[5].x, v[1].y - v[5].y);
    const double2 o2 = make_double2(v[3].x + v[7].x, v[3].y + v[7].y), o3 = make_double2(v[3].y - v[7].y, v[7].x - v[3].x);
    const double2 O0 = make_double2(o0.x + o2.x, o0.y + o2.y), O1r = make_double2(o1.x + o3.x, o1.y + o3.y);
    const double2 O2r = make_double2(o0.x - o2.x, o0.y - o2.y), O3r = make_double2(o1.x - o3.x, o1.y - o3.y);
    const double hh = 0.70710678118654752440;
    const double2 O1 = make_double2(hh * (O1r.x + O1r.y), hh * (O1r.y - O1r.x));
    const double2 O2 = make_double2(O2r.y, -O2r.x);
    const double2 O3 = make_double2(hh * (O3r.y - O3r.x), -hh * (O3r.x + O3r.y));
    v[0] = make_double2(E0.x + O0.x, E0.y + O0.y);
    v[1] = make_double2(E1.x + O1.x, E1.y + O1.y);
    v[2] = make_double2(E2.x + O2.x, E2.y + O2.y);
    v[3] = make_double2(E3.x + O3.x, E3.y + O3.y);
    v[4] = make_double2(E0.x - O0.x, E0.y - O0.y);
    v[5] = make_double2(E1.x - O1.x, E1.y - O1.y);
    v[6] = make_double2(E2.x - O2.x, E2.y - O2.y);
    v[7] = make_double2(E3.x - O3.x, E3.y - O3.y);
}
__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// forward transform of nl lines (line l at src + l * ld, natural order) into dst + l * ld (natural order); src is destroyed.
// tw = exp(-2 pi i q / F), q < F, in LDS.  Ends with a workgroup barrier.
// MUL = 1: the outputs are multiplied by mul[k * mul_stride + l] (entry k of line l) and conjugated on the way out -- the
// spectrum multiply and the conjugation in front of the inverse transform, done on the registers that hold the result;
// the return value is the thread's share of sum Re(mul) |X|^2 (with the centred spectrum: <w, T w> by Parseval).
// MUL = 2 (round 3, Hermitian cooperative solve): line l carries TWO real columns as real and imaginary part; they are
// multiplied by the REAL spectra Re mul[k * mul_stride + l] and Re mul[k * mul_stride + l + mul_pair], halved (the unpacking
// behind the inverse transform adds two terms) and conjugated; the return value is sum S_a re^2 + S_b im^2.
// MUL = 3: the same with `mul` pointing at an array of REAL spectrum values (3-D Hermitian iteration: half the bytes).
// MUL = 5: the thread's eight spectrum pairs are handed in (`mul` = its register array, filled by spectrum_prefetch<R> before the
// data of the lines was even loaded: 3-D mid0 kernel, one pass of <= 256 / (8 R) lines).
// MUL = 4: the same with the two real spectra of every line already in LDS as double2 pairs, `mul`[k * mul_stride + l] (the
// cooperative Hermitian solve keeps its workgroup's slice there: no global round trip inside the transform).
// B = 48 (round 4): lines of 48 R points (96, 192, 384 = 3 * 2^k) -- the smallest smooth circulant grids between the powers of two.
// The R interleaved sub-transforms are 48-point lines as 6 x 8 in eight lanes (cg_persistent.hip, cg_herm48_kernel: radix 6 on
// eight lanes, writer-side twiddles, radix 8 on six lanes), the sub-blocks of the exchanges are 48 entries long, and the combine
// over r hands 48 / (8 R) = 3, 1.5, 0.75 butterflies to a lane (predicated beyond 48).
__device__ __forceinline__ void dft3_inplace(double2 b0, double2 b1, double2 b2, double2& x0, double2& x1, double2& x2) {
    const double s = 0.86602540378443864676;
    const double2 sm = make_double2(b1.x + b2.x, b1.y + b2.y), df = make_double2(b1.x - b2.x, b1.y - b2.y);
    x0 = make_double2(b0.x + sm.x, b0.y + sm.y);
    const double2 m = make_double2(fma(-0.5, sm.x, b0.x), fma(-0.5, sm.y, b0.y));
    x1 = make_double2(fma(s, df.y, m.x), fma(-s, df.x, m.y));
    x2 = make_double2(fma(-s, df.y, m.x), fma(s, df.x, m.y));
}
__device__ __forceinline__ void dft6_inplace(double2 (&v)[6]) {
    const double s = 0.86602540378443864676;
    double2 e0, e1, e2, o0, o1, o2;
    dft3_inplace(v[0], v[2], v[4], e0, e1, e2);
    dft3_inplace(v[1], v[3], v[5], o0, o1, o2);
    const double2 w1 = make_double2(fma(s, o1.y, 0.5 * o1.x), fma(-s, o1.x, 0.5 * o1.y));
    const double2 w2 = make_double2(fma(s, o2.y, -0.5 * o2.x), fma(-s, o2.x, -0.5 * o2.y));
    v[0] = make_double2(e0.x + o0.x, e0.y + o0.y);
    v[3] = make_double2(e0.x - o0.x, e0.y - o0.y);
    v[1] = make_double2(e1.x + w1.x, e1.y + w1.y);
    v[4] = make_double2(e1.x - w1.x, e1.y - w1.y);
    v[2] = make_double2(e2.x + w2.x, e2.y + w2.y);
    v[5] = make_double2(e2.x - w2.x, e2.y - w2.y);
}

template <int R, int MUL = 0, int B = 64>
__device__ __forceinline__ double line_fft_inwave(double2* src, double2* dst, int ld, int nl, const double2* tw,
                                                  const double2* __restrict__ mul = nullptr, int64_t mul_stride = 0,
                                                  int64_t mul_pair = 0) {
    static_assert(B == 64 || (B == 48 && R >= 2), "sub-transforms of 64 points, or of 48 points under a combine");
    constexpr int LPL = 8 * R, LINES = kLineThreads / LPL;
    constexpr int U = B == 64 ? (R > 1 ? 8 / R : 1) : (48 + LPL - 1) / LPL;       // k1 values per lane in the combine
    constexpr bool kPartial = B == 48 && U * LPL != 48;                          // the last k1 of a lane may lie beyond the block
    double psum = 0.0;          // MUL: this thread's share of sum_k Re(mul_k) |X_k|^2 over its lines (Parseval: <w, T w>)
    const int li = threadIdx.x & (LPL - 1), lsub = threadIdx.x / LPL;
    const int r = li >> 3, j = li & 7;
    for (int l0 = 0; l0 < nl; l0 += LINES) {
        const bool act = l0 + lsub < nl;
        double2* s = src + (act ? l0 + lsub : l0) * ld;
        double2* d = dst + (act ? l0 + lsub : l0) * ld;
        double2 v[8];
        if (B == 64) {
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = s[R * (j + 8 * t) + r];
            dft8_inplace(v);
            wave_sync_lds();
            if (act) {
#pragma unroll
                for (int m = 0; m < 8; ++m) d[r * 64 + 8 * j + (m ^ j)] = v[m];
            }
            wave_sync_lds();
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = d[r * 64 + 8 * t + (j ^ t)];
#pragma unroll
            for (int t = 1; t < 8; ++t) v[t] = cmul(v[t], tw[R * j * t]);            // w_64^(j t)
            dft8_inplace(v);                                                        // v[t] = Y_r[j + 8 t]
        } else {
            double2 u6[6];
#pragma unroll
            for (int t = 0; t < 6; ++t) u6[t] = s[R * (j + 8 * t) + r];
            dft6_inplace(u6);
#pragma unroll
            for (int k = 1; k < 6; ++k) u6[k] = cmul(u6[k], tw[R * j * k]);          // w_48^(j k), writer side (35 R < 48 R: no wrap)
            wave_sync_lds();
            if (act) {
#pragma unroll
                for (int k = 0; k < 6; ++k) d[r * 48 + 6 * j + k] = u6[k];
            }
            wave_sync_lds();
            const int jr = j < 6 ? j : 0;                                           // lanes 6, 7 of a sub-transform idle from here on
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = d[r * 48 + 6 * t + jr];
            dft8_inplace(v);                                                        // v[t] = Y_r[j + 6 t], j < 6
        }
        if (R == 1) {                                                           // a plain 64-point line: done
            if (MUL) {
                double2 mv[8];
                const int lq = act ? l0 + lsub : l0;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    if (MUL == 5) {
                        mv[t] = mul[t];
                    } else if (MUL == 4) {
                        mv[t] = mul[(j + 8 * t) * (int)mul_stride + lq];
                    } else if (MUL == 3) {
                        const double* rs = reinterpret_cast<const double*>(mul);
                        mv[t] = make_double2(rs[(int64_t)(j + 8 * t) * mul_stride + lq], rs[(int64_t)(j + 8 * t) * mul_stride + lq + mul_pair]);
                    } else {
                        mv[t] = mul[(int64_t)(j + 8 * t) * mul_stride + lq];
                        if (MUL == 2) mv[t].y = mul[(int64_t)(j + 8 * t) * mul_stride + lq + mul_pair].x;
                    }
                }
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    if (MUL >= 2) {
                        if (act) psum += mv[t].x * v[t].x * v[t].x + mv[t].y * v[t].y * v[t].y;
                        v[t] = make_double2(0.5 * mv[t].x * v[t].x, -0.5 * mv[t].y * v[t].y);
                    } else {
                        if (act) psum += mv[t].x * (v[t].x * v[t].x + v[t].y * v[t].y);
                        const double2 m = cmul(v[t], mv[t]);
                        v[t] = make_double2(m.x, -m.y);
                    }
                }
            }
            wave_sync_lds();
            if (act) {
#pragma unroll
                for (int t = 0; t < 8; ++t) d[j + 8 * t] = v[t];
            }
            continue;
        }
        constexpr int PS = B == 64 ? 8 : 6;                                     // stride of a lane's outputs inside its sub-transform
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = cmul(v[t], tw[r * ((B == 64 ? j : (j < 6 ? j : 0)) + PS * t)]);      // w_F^(r k1)
        wave_sync_lds();
        if (act && (B == 64 || j < 6)) {
#pragma unroll
            for (int t = 0; t < 8; ++t) s[r * B + j + PS * t] = v[t];
        }
        wave_sync_lds();
        // combine over r: lane li takes k1 = li + LPL u, u < U (B = 48: while k1 < 48)
        double2 mv[MUL ? 8 : 1];
        if (MUL) {                                                              // requested now, used after the butterflies
            const int lq = act ? l0 + lsub : l0;
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int q = 0; q < R; ++q) {
                    const int k1 = (kPartial && li + LPL * u >= B) ? 0 : li + LPL * u;
                    if (MUL == 5) {
                        mv[u * R + q] = mul[u * R + q];
                    } else if (MUL == 4) {
                        mv[u * R + q] = mul[(k1 + B * q) * (int)mul_stride + lq];
                    } else if (MUL == 3) {
                        const double* rs = reinterpret_cast<const double*>(mul);
                        mv[u * R + q] = make_double2(rs[(int64_t)(k1 + B * q) * mul_stride + lq],
                                                     rs[(int64_t)(k1 + B * q) * mul_stride + lq + mul_pair]);
                    } else {
                        mv[u * R + q] = mul[(int64_t)(k1 + B * q) * mul_stride + lq];
                        if (MUL == 2) mv[u * R + q].y = mul[(int64_t)(k1 + B * q) * mul_stride + lq + mul_pair].x;
                    }
                }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int q = 0; q < R; ++q) v[u * R + q] = s[q * B + ((kPartial && li + LPL * u >= B) ? 0 : li + LPL * u)];
        if (R == 8) {
            dft8_inplace(v);
        } else if (R == 4) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const double2 a0 = v[4 * u], a1 = v[4 * u + 1], a2 = v[4 * u + 2], a3 = v[4 * u + 3];
                const double2 t0 = make_double2(a0.x + a2.x, a0.y + a2.y), t1 = make_double2(a0.x - a2.x, a0.y - a2.y);
                const double2 t2 = make_double2(a1.x + a3.x, a1.y + a3.y), t3 = make_double2(a1.y - a3.y, a3.x - a1.x);
                v[4 * u] = make_double2(t0.x + t2.x, t0.y + t2.y);
                v[4 * u + 1] = make_double2(t1.x + t3.x, t1.y + t3.y);
                v[4 * u + 2] = make_double2(t0.x - t2.x, t0.y - t2.y);
                v[4 * u + 3] = make_double2(t1.x - t3.x, t1.y - t3.y);
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double2 a0 = v[2 * u], a1 = v[2 * u + 1];
                v[2 * u] = make_double2(a0.x + a1.x, a0.y + a1.y);
                v[2 * u + 1] = make_double2(a0.x - a1.x, a0.y - a1.y);
            }
        }
        if (MUL) {
#pragma unroll
            for (int i = 0; i < U * R; ++i) {
                const bool live = act && !(kPartial && li + LPL * (i / R) >= B);
                if (MUL >= 2) {
                    if (live) psum += mv[i].x * v[i].x * v[i].x + mv[i].y * v[i].y * v[i].y;
                    v[i] = make_double2(0.5 * mv[i].x * v[i].x, -0.5 * mv[i].y * v[i].y);
                } else {
                    if (live) psum += mv[i].x * (v[i].x * v[i].x + v[i].y * v[i].y);
                    const double2 m = cmul(v[i], mv[i]);
                    v[i] = make_double2(m.x, -m.y);
                }
            }
        }
        wave_sync_lds();
        if (act) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int q = 0; q < R; ++q)
                    if (!(kPartial && li + LPL * u >= B)) d[li + LPL * u + B * q] = v[u * R + q];
        }
    }
    __syncthreads();
    return psum;
}

// One out-of-line copy per (R, MUL) shared by every call site of a kernel: inlined four times per operator application the
// unrolled transforms put the cooperative kernel's loop body beyond the instruction cache (its speed then moved by 10-40 %
// with every unrelated code change).  Operands are passed as offsets into the kernel's dynamic LDS so that the accesses
// stay LDS instructions.
extern __shared__ double2 efgp_line_lds[];
template <int R, int MUL, int B = 64>
__device__ __noinline__ double line_fft_inwave_call(int src_off, int dst_off, int ld, int nl, int tw_off, const double2* mul,
                                                    int64_t mul_stride, int64_t mul_pair = 0) {
    return line_fft_inwave<R, MUL, B>(efgp_line_lds + src_off, efgp_line_lds + dst_off, ld, nl, efgp_line_lds + tw_off, mul, mul_stride,
                                      mul_pair);
}
// lengths the in-wave transforms cover: 64 R and (round 4) 48 R
__device__ __host__ __forceinline__ bool line_fft_fast_len(int F) {
    return F == 64 || F == 128 || F == 256 || F == 512 || F == 96 || F == 192 || F == 384;
}
// one dispatch over the line length for every multiply mode (MUL as in line_fft_inwave)
template <int MUL>
__device__ __forceinline__ double line_fft_dispatch(int F, int so, int dn, int ld, int nl, int to, const double2* mul, int64_t mul_stride,
                                                    int64_t mul_pair) {
    switch (F) {
        case 128: return line_fft_inwave_call<2, MUL>(so, dn, ld, nl, to, mul, mul_stride, mul_pair);
        case 256: return line_fft_inwave_call<4, MUL>(so, dn, ld, nl, to, mul, mul_stride, mul_pair);
        case 512: return line_fft_inwave_call<8, MUL>(so, dn, ld, nl, to, mul, mul_stride, mul_pair);
        case 96: return line_fft_inwave_call<2, MUL, 48>(so, dn, ld, nl, to, mul, mul_stride, mul_pair);
        case 192: return line_fft_inwave_call<4, MUL, 48>(so, dn, ld, nl, to, mul, mul_stride, mul_pair);
        default: return line_fft_inwave_call<8, MUL, 48>(so, dn, ld, nl, to, mul, mul_stride, mul_pair);      // 384
    }
}
// result buffer is always `dst`
__device__ __forceinline__ double2* line_fft_fast(double2* src, double2* dst, int F, int ld, int nl, const double2* tw) {
    const int so = (int)(src - efgp_line_lds), dn = (int)(dst - efgp_line_lds), to = (int)(tw - efgp_line_lds);
    if (F == 64) line_fft_inwave_call<1, 0>(so, dn, ld, nl, to, nullptr, 0);
    else line_fft_dispatch<0>(F, so, dn, ld, nl, to, nullptr, 0, 0);
    return dst;
}
// forward transform, spectrum multiply (mul[k * mul_stride + l]) and conjugation in one pass (F = 96 .. 512)
__device__ __forceinline__ double2* line_fft_fast_mul(double2* src, double2* dst, int F, int ld, int nl, const double2* tw,
                                                      const double2* mul, int64_t mul_stride, double& psum) {
    const int so = (int)(src - efgp_line_lds), dn = (int)(dst - efgp_line_lds), to = (int)(tw - efgp_line_lds);
    psum += line_fft_dispatch<1>(F, so, dn, ld, nl, to, mul, mul_stride, 0);
    return dst;
}
// the same for lines that carry two real columns (MUL = 2)
__device__ __forceinline__ double2* line_fft_fast_mul2(double2* src, double2* dst, int F, int ld, int nl, const double2* tw,
                                                       const double2* mul, int64_t mul_stride, int64_t mul_pair, double& psum) {
    const int so = (int)(src - efgp_line_lds), dn = (int)(dst - efgp_line_lds), to = (int)(tw - efgp_line_lds);
    psum += line_fft_dispatch<2>(F, so, dn, ld, nl, to, mul, mul_stride, mul_pair);
    return dst;
}
// two real columns per line, spectra resident in LDS at offset spec_off (MUL = 4)
__device__ __forceinline__ double2* line_fft_fast_mul4(double2* src, double2* dst, int F, int ld, int nl, const double2* tw,
                                                       const double2* spec_lds, int stride, double& psum) {
    const int so = (int)(src - efgp_line_lds), dn = (int)(dst - efgp_line_lds), to = (int)(tw - efgp_line_lds);
    const double2* mul = efgp_line_lds + (int)(spec_lds - efgp_line_lds);
    psum += line_fft_dispatch<4>(F, so, dn, ld, nl, to, mul, stride, 0);
    return dst;
}
// two real columns per line, spectra given as a REAL array (MUL = 3)
__device__ __forceinline__ double2* line_fft_fast_mul3(double2* src, double2* dst, int F, int ld, int nl, const double2* tw,
                                                       const double* spec, int64_t stride, int64_t pair) {
    const int so = (int)(src - efgp_line_lds), dn = (int)(dst - efgp_line_lds), to = (int)(tw - efgp_line_lds);
    const double2* mul = reinterpret_cast<const double2*>(spec);
    if (F == 64) line_fft_inwave_call<1, 3>(so, dn, ld, nl, to, mul, stride, pair);
    else if (F == 128) line_fft_inwave_call<2, 3>(so, dn, ld, nl, to, mul, stride, pair);
    else if (F == 256) line_fft_inwave_call<4, 3>(so, dn, ld, nl, to, mul, stride, pair);
    else line_fft_inwave_call<8, 3>(so, dn, ld, nl, to, mul, stride, pair);
    return dst;
}
// the (S_a, S_b) pairs line_fft_inwave<R, 2..4> would read for this thread in its single pass over nl <= 256 / (8 R) lines
template <int R>
__device__ __forceinline__ void spectrum_prefetch(const double* __restrict__ spec, int64_t stride, int64_t pair, int nl, double2 (&pre)[8]) {
    constexpr int LPL = 8 * R, U = R > 1 ? 8 / R : 1;
    const int li = threadIdx.x & (LPL - 1), lsub = threadIdx.x / LPL;
    const int lq = lsub < nl ? lsub : 0;
    if (R == 1) {
        const int j = li & 7;
#pragma unroll
        for (int t = 0; t < 8; ++t) pre[t] = make_double2(spec[(int64_t)(j + 8 * t) * stride + lq], spec[(int64_t)(j + 8 * t) * stride + lq + pair]);
    } else {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int q = 0; q < R; ++q)
                pre[u * R + q] = make_double2(spec[(int64_t)(li + LPL * u + 64 * q) * stride + lq], spec[(int64_t)(li + LPL * u + 64 * q) * stride + lq + pair]);
    }
}
// in-wave transform for the lengths it covers, the generic Stockham stages otherwise; the result buffer is returned
__device__ __forceinline__ double2* line_fft_any(double2* A, double2* B, int F, int ld, int nl, const double2* tw) {
    if (line_fft_fast_len(F)) return line_fft_fast(A, B, F, ld, nl, tw);
    return line_fft(A, B, F, ld, nl, tw);
}

// cooperative copy of a twiddle table into LDS (visible after the caller's next barrier)
__device__ __forceinline__ void load_twiddles(double2* dst, const double2* __restrict__ src, int F) {
    for (int i = threadIdx.x; i < F; i += kLineThreads) dst[i] = src[i];
}

struct LineArgs {
    CgArgs c;
    const double2* vhat;      // [F0][F1], already divided by F0*F1
    const double2* tw0;       // exp(-2 pi i q / F0)
    const double2* tw1;
    double2* b1;              // [slots][n0][F1]
    double2* b2;              // [slots][n0][F1]
    int nblk_rows;            // row blocks per system = ceil(n0 / lpb): partial sums of <p, A p>
    int lpb;                  // lines per workgroup (a power of two; larger for batched solves: fewer reduction arrivals)
};

__global__ __launch_bounds__(kLineThreads) void cg_rows_fwd_kernel(LineArgs a) {
    extern __shared__ double2 lsm[];
    const CgArgs& c = a.c;
    const int slot = blockIdx.y;
    const int row = c.rows ? c.rows[slot] : slot;
    if (row < 0) return;
    const CgRowScalars sc = c.sc[row];
    if (!sc.active) return;
    const int n0 = (int)c.g.n[0], n1 = (int)c.g.n[1], F1 = (int)c.g.F[1], ld = F1 + 1;
    const int r0 = blockIdx.x * a.lpb;
    const int nl = min(a.lpb, n0 - r0);
    double2* A = lsm;
    double2* B = lsm + a.lpb * ld;
    double2* tws = B + a.lpb * ld;
    load_twiddles(tws, a.tw1, F1);
    const int64_t base = (int64_t)row * c.g.M;
    for (int w = threadIdx.x; w < nl * F1; w += kLineThreads) {
        const int l = w / F1, i1 = w - l * F1;
        double2 v = make_double2(0.0, 0.0);
        if (i1 < n1) {
            const int t = (r0 + l) * n1 + i1;
            double2 pv = c.p[base + t];
            if (sc.do_p) {                                       // deferred p <- r/diag + beta p
                double2 zv = c.r[base + t];
                if (c.diag) {
                    zv.x /= c.diag[t];
                    zv.y /= c.diag[t];
                }
                pv = make_double2(zv.x + sc.beta * pv.x, zv.y + sc.beta * pv.y);
                c.p[base + t] = pv;
            }
            v = cmul(pv, c.ws[t]);
        }
        A[l * ld + i1] = v;
    }
    __syncthreads();
    const double2* X = line_fft_any(A, B, F1, ld, nl, tws);
    double2* out = a.b1 + ((int64_t)slot * n0 + r0) * F1;
    for (int w = threadIdx.x; w < nl * F1; w += kLineThreads) {
        const int l = w / F1, i1 = w - l * F1;
        out[(int64_t)l * F1 + i1] = X[l * ld + i1];
    }
}

__global__ __launch_bounds__(kLineThreads) void cg_cols_mid_kernel(LineArgs a) {
    extern __shared__ double2 lsm[];
    const CgArgs& c = a.c;
    const int slot = blockIdx.y;
    const int row = c.rows ? c.rows[slot] : slot;
    if (row < 0) return;
    if (!c.sc[row].active) return;
    const int n0 = (int)c.g.n[0], F0 = (int)c.g.F[0], F1 = (int)c.g.F[1], ld = F0 + 1;
    const int c0 = blockIdx.x * a.lpb;                  // first column of this block (F1 % 8 == 0)
    double2* A = lsm;
    double2* B = lsm + a.lpb * ld;
    double2* tws = B + a.lpb * ld;
    load_twiddles(tws, a.tw0, F0);
    const double2* in = a.b1 + (int64_t)slot * n0 * F1;
    // line l = column c0 + l; consecutive threads read consecutive columns of one row (128-B segments)
    for (int w = threadIdx.x; w < a.lpb * F0; w += kLineThreads) {
        const int i0 = w / a.lpb, l = w - i0 * a.lpb;
        A[l * ld + i0] = i0 < n0 ? in[(int64_t)i0 * F1 + c0 + l] : make_double2(0.0, 0.0);
    }
    __syncthreads();
    double2* X = line_fft_any(A, B, F0, ld, a.lpb, tws);
    double2* Y = X == A ? B : A;
    // .* vhat, conjugate: the inverse transform is conj(FFT(conj(.)))
    for (int w = threadIdx.x; w < a.lpb * F0; w += kLineThreads) {
        const int i0 = w / a.lpb, l = w - i0 * a.lpb;
        const double2 m = cmul(X[l * ld + i0], a.vhat[(int64_t)i0 * F1 + c0 + l]);
        X[l * ld + i0] = make_double2(m.x, -m.y);
    }
    __syncthreads();
    const double2* Z = line_fft_any(X, Y, F0, ld, a.lpb, tws);
    double2* out = a.b2 + (int64_t)slot * n0 * F1;
    for (int w = threadIdx.x; w < a.lpb * n0; w += kLineThreads) {
        const int j = w / a.lpb, l = w - j * a.lpb;
        const double2 z = Z[l * ld + (n0 - 1) + j];               // crop window rows [n0-1, 2 n0-1)
        out[(int64_t)j * F1 + c0 + l] = make_double2(z.x, -z.y);
    }
}

__global__ __launch_bounds__(kLineThreads) void cg_rows_inv_kernel(LineArgs a) {
    extern __shared__ double2 lsm[];
    __shared__ double red[kLineThreads / 64];
    __shared__ int flag;
    const CgArgs& c = a.c;
    const int slot = blockIdx.y;
    const int row = c.rows ? c.rows[slot] : slot;
    if (row < 0) return;
    if (!c.sc[row].active) return;
    const int n0 = (int)c.g.n[0], n1 = (int)c.g.n[1], F1 = (int)c.g.F[1], ld = F1 + 1;
    const int r0 = blockIdx.x * a.lpb;
    const int nl = min(a.lpb, n0 - r0);
    double2* A = lsm;
    double2* B = lsm + a.lpb * ld;
    double2* tws = B + a.lpb * ld;
    load_twiddles(tws, a.tw1, F1);
    const double2* in = a.b2 + ((int64_t)slot * n0 + r0) * F1;
    for (int w = threadIdx.x; w < nl * F1; w += kLineThreads) {
        const int l = w / F1, i1 = w - l * F1;
        const double2 v = in[(int64_t)l * F1 + i1];
        A[l * ld + i1] = make_double2(v.x, -v.y);
    }
    __syncthreads();
    const double2* X = line_fft_any(A, B, F1, ld, nl, tws);
    const int64_t base = (int64_t)row * c.g.M;
    double pAp = 0.0;
    for (int w = threadIdx.x; w < nl * n1; w += kLineThreads) {
        const int l = w / n1, i1 = w - l * n1;
        const int t = (r0 + l) * n1 + i1;
        const double2 z = X[l * ld + (n1 - 1) + i1];
        const double2 pv = c.p[base + t];
        const double2 Ap = apply_A(c, c.ws[t], make_double2(z.x, -z.y), pv);
        c.ap[base + t] = Ap;
        pAp += pv.x * Ap.x + pv.y * Ap.y;
    }
    pAp = block_sum(pAp, red);
    double* part = c.partial + (int64_t)row * 3 * kCgBlocksMax;
    if (threadIdx.x == 0) store_agent(&part[blockIdx.x], pAp);
    if (!arrive_count(&c.counter[2 * row], a.nblk_rows, &flag)) return;      // arrival counter 0 with this kernel's own block count
    __shared__ double fin[kCgBlocksMax];
    if (threadIdx.x < a.nblk_rows) fin[threadIdx.x] = load_agent(&part[threadIdx.x]);
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < a.nblk_rows; ++i) t += fin[i];
        c.sc[row].pAp = t + kDivEps;
    }
}

// ==========================================================================================================
// Cooperative single-launch solve for the same 2-D grids: the WHOLE CG loop in one kernel spread over G workgroups
// (one per CU) per system, with grid barriers between the phases instead of kernel boundaries -- no launch latency
// (3-4 us x 4 launches per iteration above), no host poll, no hipGraph.
//   phase R  : own rows  -- ws .* p (from registers), zero-padded, forward FFT along dim 1           -> B1 (write-through)
//   barrier
//   phase C  : own column block -- forward FFT along dim 0 of the n non-zero inputs, .* vhat, inverse, crop rows -> B2
//   barrier
//   phase Ri : own rows  -- inverse FFT along dim 1, crop, A p (registers), partial <p, A p>
//   barrier  : every workgroup adds the G partials in index order (same bits everywhere: uniform control flow)
//   update   : own rows  -- x, r, z in registers, partial <r,r>, <r,z>
//   barrier  : sums, stopping rule, beta, p
// Ownership is fixed for the whole solve: workgroup g owns rows [g L, (g+1) L) of the mode block in the row phases and
// the vector elements of those rows (x, r, p, A p, ws, diag stay in ITS registers: 4 per thread), and columns
// [g c, (g+1) c) in the column phase.  Only B1, B2 and the partial sums cross workgroups; they are written and read
// with agent-scope accesses (write-through stores / cache-bypassing loads: the XCDs' L2 caches are not coherent with
// each other), so a barrier needs no cache write-back or invalidate: the stores are waited for (s_waitcnt) before the
// workgroup's arrival is counted.  A barrier is one agent-scope atomic add on a monotone counter and a bounded poll;
// a poll that runs out (a co-resident workgroup never arrived) sets a status flag on which every workgroup leaves and
// the host falls back to the multi-launch iteration above.  All G x systems workgroups must be resident at once:
// the host launches at most one workgroup per CU.
// ==========================================================================================================
constexpr int kCoopSlots = 4;                 // vector elements per thread (own rows x n1 <= 4 x 256)
constexpr unsigned kCoopPollLimit = 1u << 22; // ~ seconds of polling before a barrier is declared dead
constexpr int kCoopMaxG = 64;
constexpr int kCoopLoads = 16;                // grid elements a thread brings in per phase (lpbc F0 / 256 and lines F1 / 256 at most)

struct CoopArgs {
    ToepGeom g;
    const double2* ws;
    const double* diag;
    const double* diag_scale;  // when diag is null and this is not: Jacobi diagonal (*diag_scale) |ws|^2 + sigmasq (device scalar)
    int b_times_ws;           // right-hand side is ws .* b (the fit's D F*y, efgpnd.py:792)
    int zero_x0;              // x0 = 0: x is output only, the initial operator application is skipped (A 0 = 0)
    double sigmasq;
    int variant;
    double tol;
    int early_stop;
    int batched;
    int max_iter;
    const double2* b;         // [systems][M]
    double2* x;               // in: x0 (unless zero_x0), out: solution
    const double2* vhat;
    const double2* tw0;
    const double2* tw1;
    double2* b1;              // [systems][n0][F1]
    double2* b2;              // [systems][n0][F1]
    double2* b3;              // [systems][n0][F1]  (Hermitian kernel: row transforms of the current direction)
    int spec_lds;             // Hermitian kernel: the workgroup's slice of the spectrum lives in LDS (cols_wg == lpbc)
    double* partial;          // [systems][3][kCoopMaxG]
    unsigned* bar;            // [systems] arrival counters, 64 bytes apart, zero at launch
    int* iters;               // [systems]
    int* status;              // [0] != 0: a barrier timed out
    int nan_on_dead;          // asynchronous entries: a system whose barrier died gets NaN in x (nobody may mistake x0 for a solution)
    double* hist;
    int hist_cap;
    int G;                    // workgroups per system
    int bar_need;             // arrivals a barrier waits for: G (G + 1 under the test hook EFGP_COOP_TEST_DEAD: every barrier dies)
    int rows_wg;              // rows of the mode block owned per workgroup (ceil(n0 / G))
    int cols_wg;              // columns owned per workgroup (F1 / G)
    int lines;                // rows per LDS pass of the row phases
    int lpbc;                 // columns per LDS pass of the column phase (a power of two)
    int dbg;                  // EFGP_COOP_DBG=2: cycle counters per phase (workgroup 0 of system 0)
    double* stamps;
};

// w / F for w < 2^16 and the line lengths of these kernels (96 .. 512) through one multiply-high: magic = floor(2^32 / F) + 1 is
// exact on that range (F w < 2^32 / F); the loops that scatter rows into LDS images and back did a full integer division per
// element (~40 instructions each, 16 per thread and phase)
__device__ __forceinline__ int coop_div(int w, unsigned magic) { return (int)__umulhi((unsigned)w, magic); }

// Jacobi diagonal entry t of the cooperative kernels: explicit array, or scale |ws_t|^2 + sigmasq with the two roundings of the
// reference's torch expression (efgpnd.py:795-799; pcg::jacobi_entry), or 1
__device__ __forceinline__ double coop_jacobi(const CoopArgs& a, double2 w, int64_t t) {
    if (a.diag) return a.diag[t];
    if (a.diag_scale) return __dadd_rn(__dmul_rn(*a.diag_scale, __dadd_rn(__dmul_rn(w.x, w.x), __dmul_rn(w.y, w.y))), a.sigmasq);
    return 1.0;
}

// SOLO (G == 1): the intermediate grids are private to the workgroup -- ordinary cached accesses, no grid barrier
template <bool SOLO>
__device__ __forceinline__ void store_x2(double2* p, double2 v) {
    if (SOLO) {
        *p = v;
    } else {
        __hip_atomic_store(&p->x, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&p->y, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
template <bool SOLO>
__device__ __forceinline__ double2 load_x2(const double2* p) {
    if (SOLO) return *p;
    return make_double2(__hip_atomic_load(&p->x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                        __hip_atomic_load(&p->y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// returns false (in every thread of the workgroup) when the barrier is dead
__device__ __forceinline__ bool coop_barrier(unsigned* bar, unsigned& epoch, int G, int* status, int* sflag) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);                       // this thread's write-through stores have been acknowledged
    __syncthreads();
    ++epoch;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = epoch * (unsigned)G;
        int ok = 1;
        unsigned polls = 0;
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            ++polls;
            if (polls > kCoopPollLimit || ((polls & 255u) == 0u && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *sflag = ok;
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return *sflag != 0;
}

// KS: vector elements per thread (owned rows x n1 <= KS x 256).  Two launch shapes (chosen by the host):
//   latency  (few systems)  : G = F1 / lpbc workgroups per system, one row pass and one column pass per phase, KS = 4
//   throughput (many systems): as few workgroups per system as the registers allow (G = 1: no grid barrier at all, the
//                              intermediate grids stay in this CU's caches), several LDS passes per phase, KS = 8
template <int KS, bool SOLO>
__global__ __launch_bounds__(kLineThreads) void cg_coop2d_kernel(CoopArgs a) {
    extern __shared__ double2 lsm[];
    __shared__ double red[kLineThreads / 64];
    __shared__ double fin[3 * kCoopMaxG];
    __shared__ int sflag;
    const int tid = threadIdx.x, wg = blockIdx.x, sys = blockIdx.y, G = a.G;
    const int n0 = (int)a.g.n[0], n1 = (int)a.g.n[1], F0 = (int)a.g.F[0], F1 = (int)a.g.F[1];
    const int ldr = F1 + 1, ldc = F0 + 1;
    const int lgC = ilog2(a.lpbc);
    const unsigned magicF1 = 0xFFFFFFFFu / (unsigned)F1 + 1u, magicZ = 0xFFFFFFFFu / (unsigned)max(1, F1 - n1) + 1u;
    const int bufsz = max(a.lines * ldr, a.lpbc * ldc);
    double2* A = lsm;
    double2* B = lsm + bufsz;
    double2* tw1s = B + bufsz;
    double2* tw0s = F0 == F1 ? tw1s : tw1s + F1;
    load_twiddles(tw1s, a.tw1, F1);
    if (F0 != F1) load_twiddles(tw0s, a.tw0, F0);
    __syncthreads();
    const int64_t M = a.g.M;
    const int r0 = wg * a.rows_wg;
    const int nrows = max(0, min(a.rows_wg, n0 - r0));      // rows owned (0 for trailing workgroups)
    const int cnt = nrows * n1;
    const int64_t base = (int64_t)sys * M + (int64_t)r0 * n1;    // the owned elements are contiguous in the flat vector
    const int c_lo = wg * a.cols_wg;
    double2* b1 = a.b1 + (int64_t)sys * n0 * F1;
    double2* b2 = a.b2 + (int64_t)sys * n0 * F1;
    double* part = a.partial + (int64_t)sys * 3 * kCoopMaxG;
    unsigned* bar = a.bar + (int64_t)sys * 16;
    unsigned epoch = 0;
    long long st_prev = (long long)__builtin_readcyclecounter();
#define COOP_STAMP(slot_)                                                                     \
    do {                                                                                      \
        if (a.dbg == 2 && wg == 0 && sys == 0 && tid == 0) {                                  \
            const long long now_ = (long long)__builtin_readcyclecounter();                   \
            a.stamps[slot_] += (double)(now_ - st_prev);                                      \
            st_prev = now_;                                                                   \
        }                                                                                     \
    } while (0)

    double2 xv[KS], rv[KS], pv[KS], wsv[KS];
    double dg[KS];
    int lrow[KS], lcol[KS];                              // owned element s: row r0 + lrow[s], column lcol[s] of the mode block
    bool ok[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int e = tid + s * kLineThreads;
        ok[s] = e < cnt;
        lrow[s] = ok[s] ? e / n1 : 0;
        lcol[s] = ok[s] ? e - lrow[s] * n1 : 0;
        if (ok[s]) {
            const int t = r0 * n1 + e;
            xv[s] = a.zero_x0 ? make_double2(0.0, 0.0) : a.x[base + e];
            wsv[s] = a.ws[t];
            dg[s] = coop_jacobi(a, wsv[s], t);
        } else {
            xv[s] = wsv[s] = make_double2(0.0, 0.0);
            dg[s] = 1.0;
        }
        rv[s] = pv[s] = make_double2(0.0, 0.0);
    }
    auto sync_grid = [&]() __attribute__((always_inline)) -> bool {
        if (SOLO) {
            __syncthreads();
            return true;
        }
        return coop_barrier(bar, epoch, a.bar_need, a.status, &sflag);
    };

    // Au = A u for the owned elements; false when a barrier died
    // sum of K (<= 2) per-workgroup partials over the G (<= 64) workgroups, the same bits in every workgroup: both values go
    // through ONE workgroup reduction (fixed shuffle tree, then the four wave sums in order), and every wave then loads the G
    // partials into its lanes and reduces them with the same fixed butterfly -- no LDS staging, no serial chain of G adds.
    // `slot0`: first of the three partial arrays used -- consecutive sums must use DIFFERENT arrays (a fast workgroup writes
    // its next partial while a slow one still reads the previous sum's: no barrier sits between a sum's reads and the next
    // sum's writes)
    auto all_sum = [&](double (&v)[3], int K, int slot0) __attribute__((always_inline)) -> bool {
        double a0 = v[0], a1 = K > 1 ? v[1] : 0.0;
        for (int off = 32; off > 0; off >>= 1) {
            a0 += __shfl_down(a0, off, 64);
            a1 += __shfl_down(a1, off, 64);
        }
        const int lane = tid & 63, wid = tid >> 6;
        __syncthreads();
        if (lane == 0) {
            red[wid] = a0;
            fin[wid] = a1;
        }
        __syncthreads();
        a0 = ((red[0] + red[1]) + red[2]) + red[3];
        a1 = ((fin[0] + fin[1]) + fin[2]) + fin[3];
        if (SOLO) {
            v[0] = a0;
            v[1] = a1;
            __syncthreads();
            return true;
        }
        if (tid == 0) {
            __hip_atomic_store(&part[slot0 * kCoopMaxG + wg], a0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (K > 1) __hip_atomic_store(&part[(slot0 + 1) * kCoopMaxG + wg], a1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!coop_barrier(bar, epoch, a.bar_need, a.status, &sflag)) return false;
        double b0 = lane < G ? __hip_atomic_load(&part[slot0 * kCoopMaxG + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        double b1 = (K > 1 && lane < G) ? __hip_atomic_load(&part[(slot0 + 1) * kCoopMaxG + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        for (int off = 32; off > 0; off >>= 1) {
            b0 += __shfl_xor(b0, off, 64);
            b1 += __shfl_xor(b1, off, 64);
        }
        v[0] = b0;
        v[1] = b1;
        return true;
    };
    // <u, A u> comes out of the operator application itself: sigma^2 |u|^2 (own rows, phase R) + sum_f Re(vhat_c) |FFT(ws u)|^2 (own
    // columns, phase C; Parseval with the CENTRED spectrum vhat_c = vhat . e^(2 pi i f (n-1)/F), whose circular convolution has its
    // crop window at [0, n)), reduced over the workgroups ON the barrier between phases C and Ri -- the separate <p, A p> barrier
    // and all-reduce of an iteration are gone (3 grid barriers instead of 4)
    auto apply = [&](const double2 (&u)[KS], double2 (&Au)[KS], double& uAu) __attribute__((always_inline)) -> bool {
        double pp = 0.0, cs = 0.0;
#pragma unroll
        for (int s = 0; s < KS; ++s) pp += u[s].x * u[s].x + u[s].y * u[s].y;
        // R: owned rows in passes of a.lines
        for (int p0 = 0; p0 < nrows; p0 += a.lines) {
            const int nl = min(a.lines, nrows - p0);
            // zero padding behind the n1 inputs of each row, all rows of the pass as one index range (a loop over the rows left
            // all but F1 - n1 threads idle in each of its trips: 8.4 k of the 100 k ticks of a 96 x 96 iteration)
            const int zw = F1 - n1;
            for (int e = tid; e < nl * zw; e += kLineThreads) {
                const int l = coop_div(e, magicZ);
                A[l * ldr + n1 + (e - l * zw)] = make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int s = 0; s < KS; ++s)
                if (ok[s] && lrow[s] >= p0 && lrow[s] < p0 + nl) A[(lrow[s] - p0) * ldr + lcol[s]] = cmul(u[s], wsv[s]);
            __syncthreads();
            COOP_STAMP(12);
            const double2* X = line_fft_fast(A, B, F1, ldr, nl, tw1s);
            COOP_STAMP(13);
            for (int w = tid; w < nl * F1; w += kLineThreads) {
                const int l = coop_div(w, magicF1), i1 = w - l * F1;
                store_x2<SOLO>(b1 + (int64_t)(r0 + p0 + l) * F1 + i1, X[l * ldr + i1]);
            }
            __syncthreads();                                                  // the next pass refills the buffers
        }
        COOP_STAMP(0);
        if (!sync_grid()) return false;
        COOP_STAMP(1);
        // C: owned columns in passes of a.lpbc
        for (int c0 = c_lo; c0 < c_lo + a.cols_wg; c0 += a.lpbc) {
            // every thread's loads are issued before the first is consumed (one at a time costs a memory round trip each)
            double2 tmp[kCoopLoads];
#pragma unroll
            for (int q = 0; q < kCoopLoads; ++q) {
                const int w = tid + q * kLineThreads, i0 = w >> lgC;
                tmp[q] = (w < (F0 << lgC) && i0 < n0) ? load_x2<SOLO>(b1 + (int64_t)i0 * F1 + c0 + (w & (a.lpbc - 1))) : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int q = 0; q < kCoopLoads; ++q) {
                const int w = tid + q * kLineThreads;
                if (w < (F0 << lgC)) A[(w & (a.lpbc - 1)) * ldc + (w >> lgC)] = tmp[q];
            }
            __syncthreads();
            COOP_STAMP(8);
            // forward transform, .* vhat and the conjugation in front of the inverse transform on the result registers
            double2* X = line_fft_fast_mul(A, B, F0, ldc, a.lpbc, tw0s, a.vhat + c0, F1, cs);
            double2* Y = X == A ? B : A;
            COOP_STAMP(9);
            COOP_STAMP(10);
            const double2* Z = line_fft_fast(X, Y, F0, ldc, a.lpbc, tw0s);
            COOP_STAMP(11);
            for (int w = tid; w < (n0 << lgC); w += kLineThreads) {
                const int j = w >> lgC, l = w & (a.lpbc - 1);
                const double2 z = Z[l * ldc + j];
                store_x2<SOLO>(b2 + (int64_t)j * F1 + c0 + l, make_double2(z.x, -z.y));
            }
            __syncthreads();
        }
        COOP_STAMP(2);
        {
            double t3[3] = {a.variant == 0 ? a.sigmasq * pp + cs : pp + cs / a.sigmasq, 0.0, 0.0};
            if (!all_sum(t3, 1, 2)) return false;
            uAu = t3[0];
        }
        COOP_STAMP(3);
        // Ri: owned rows in passes
#pragma unroll
        for (int s = 0; s < KS; ++s) Au[s] = make_double2(0.0, 0.0);
        for (int p0 = 0; p0 < nrows; p0 += a.lines) {
            const int nl = min(a.lines, nrows - p0);
            double2 tmp[kCoopLoads];
#pragma unroll
            for (int q = 0; q < kCoopLoads; ++q) {
                const int w = tid + q * kLineThreads;
                tmp[q] = w < nl * F1 ? load_x2<SOLO>(b2 + (int64_t)(r0 + p0) * F1 + w) : make_double2(0.0, 0.0);   // rows are contiguous
            }
#pragma unroll
            for (int q = 0; q < kCoopLoads; ++q) {
                const int w = tid + q * kLineThreads;
                if (w < nl * F1) A[(w / F1) * ldr + (w % F1)] = make_double2(tmp[q].x, -tmp[q].y);
            }
            __syncthreads();
            const double2* X = line_fft_fast(A, B, F1, ldr, nl, tw1s);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (ok[s] && lrow[s] >= p0 && lrow[s] < p0 + nl) {
                    const double2 z = X[(lrow[s] - p0) * ldr + lcol[s]];
                    const double2 gq = cmul(wsv[s], make_double2(z.x, -z.y));
                    if (a.variant == 0) Au[s] = make_double2(gq.x + a.sigmasq * u[s].x, gq.y + a.sigmasq * u[s].y);
                    else Au[s] = make_double2(gq.x / a.sigmasq + u[s].x, gq.y / a.sigmasq + u[s].y);
                }
            }
            __syncthreads();                                      // the next pass / phase overwrites the buffers
        }
        COOP_STAMP(4);
        return true;
    };
    auto dead = [&]() {
        if (wg == 0 && tid == 0) a.iters[sys] = -3;
        // The asynchronous entries have no host that reads the iteration counts before the result is used (EFGPND keeps them as
        // lazy device values): the unsolved system must show in the DATA, as the Hermitian kernel's refusal does.  The synchronous
        // entry keeps x0 in place and re-solves the system through the multi-launch iteration.
        if (a.nan_on_dead) {
#pragma unroll
            for (int s = 0; s < KS; ++s)
                if (ok[s]) a.x[base + tid + s * kLineThreads] = make_double2(__builtin_nan(""), __builtin_nan(""));
        }
    };

    const bool precond = a.diag != nullptr || a.diag_scale != nullptr;
    double2 Ap[KS];
    double uAu = 0.0;
    if (a.zero_x0) {
#pragma unroll
        for (int s = 0; s < KS; ++s) Ap[s] = make_double2(0.0, 0.0);
    } else if (!apply(xv, Ap, uAu)) {
        return dead();
    }
    double acc[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        if (ok[s]) {
            double2 bv = a.b[base + tid + s * kLineThreads];
            if (a.b_times_ws) bv = cmul(wsv[s], bv);
            rv[s] = make_double2(bv.x - Ap[s].x, bv.y - Ap[s].y);
            pv[s] = precond ? make_double2(rv[s].x / dg[s], rv[s].y / dg[s]) : rv[s];
            acc[0] += rv[s].x * pv[s].x + rv[s].y * pv[s].y;
            acc[1] += bv.x * bv.x + bv.y * bv.y;
        }
    }
    if (!all_sum(acc, 2, 0)) return dead();
    double rz = acc[0];
    const double bn = sqrt(acc[1]);
    const double den = bn > 0.0 ? bn : 1.0;
    int it = 0;
    for (; it < a.max_iter;) {
        if (!apply(pv, Ap, uAu)) return dead();
        COOP_STAMP(5);
        const double alpha = rz / (uAu + kDivEps);
        acc[0] = acc[1] = 0.0;
        double2 zv[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            xv[s].x += alpha * pv[s].x;
            xv[s].y += alpha * pv[s].y;
            rv[s].x -= alpha * Ap[s].x;
            rv[s].y -= alpha * Ap[s].y;
            zv[s] = precond ? make_double2(rv[s].x / dg[s], rv[s].y / dg[s]) : rv[s];
            acc[0] += rv[s].x * rv[s].x + rv[s].y * rv[s].y;
            acc[1] += rv[s].x * zv[s].x + rv[s].y * zv[s].y;
        }
        COOP_STAMP(6);
        if (!all_sum(acc, 2, 0)) return dead();
        COOP_STAMP(7);
        ++it;
        const double rnorm = sqrt(acc[0]), rzn = acc[1];
        if (a.hist && sys == 0 && wg == 0 && tid == 0 && it <= a.hist_cap) a.hist[it - 1] = rnorm / (den + kDivEps);
        const bool conv = a.early_stop && ((rnorm / (den + kDivEps) < a.tol) || (a.batched && rnorm < 1e-12));
        if (!a.batched && conv) break;                            // cg.py:132
        const double beta = rzn / (rz + kDivEps);
#pragma unroll
        for (int s = 0; s < KS; ++s) pv[s] = make_double2(zv[s].x + beta * pv[s].x, zv[s].y + beta * pv[s].y);
        rz = rzn;
        if (conv) break;                                          // cg.py:229-241
    }
#pragma unroll
    for (int s = 0; s < KS; ++s)
        if (ok[s]) a.x[base + tid + s * kLineThreads] = xv[s];
    if (wg == 0 && tid == 0) a.iters[sys] = it;
#undef COOP_STAMP
}

// ==========================================================================================================
// Hermitian specialisation of the cooperative solve (round 3; the mid-size counterpart of cg_herm64_kernel).
// Every CG system of an EFGP model has vectors that are coefficient arrays of REAL functions (u[-k] = conj u[k] on the
// centred modes k in [-h, h]^2), a real even ws and a real centred spectrum S = vhat_c.  With the modes placed CENTRED
// on the torus (mode k at position k mod F) the operator is
//     coefficients -> real function r(f) -> S r -> coefficients,
// and only half of every phase is needed:
//   R : rows k0 = 0..h0 of ws .* u, forward transform along dim 1              -> B1[k0][f1]      (h0 + 1 of n0 rows)
//   C : the column of a real function is the transform of a Hermitian sequence z[k0] = G[k0][c], z[-k0] = conj G[k0][c]:
//       REAL, so two columns c = q, q + F1/2 ride through one complex transform as real and imaginary part (F1/2 column
//       lines); .* the two real spectra (halved), conjugate, transform again; unpack T_q[k0] = T[k0] + conj T[-k0],
//       T_q'[k0] = (T[k0] - conj T[-k0]) / i for k0 >= 0                       -> B2[k0][f1]
//   Ri: rows k0 >= 0, inverse transform along dim 1, crop to |k1| <= h1, A u = ws .* (.) + sigma^2 u
// Row k0 = 0 stores BOTH +k1 and -k1; its line G[0][.] is real only up to the anti-Hermitian rounding noise of the iterates,
// so phase C keeps its real part only (the projection the 64 x 64 kernel needed, see cg_persistent.hip).  Dot products weigh
// the rows k0 > 0 twice.  Same recurrences, stopping rule, barrier / all-reduce machinery as cg_coop2d_kernel; a right-hand
// side that is not conjugate-even (or a ws that is not real and even) is refused: -2 iterations, NaN in x.
// Launch geometry in CoopArgs: rows_wg = rows k0 per workgroup (of h0 + 1), cols_wg = column PAIRS per workgroup (of F1 / 2),
// lpbc = pairs per LDS pass.
// ==========================================================================================================
template <int KS, bool SOLO>
__global__ __launch_bounds__(kLineThreads) void cg_coop2d_herm_kernel(CoopArgs a) {
    long long st_prev = (long long)__builtin_readcyclecounter();   // EFGP_COOP_DBG=2: shader clock ticks per phase (workgroup 0, system 0)
#define COOP_STAMP(slot_)                                                                     \
    do {                                                                                      \
        if (a.dbg == 2 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {           \
            const long long now_ = (long long)__builtin_readcyclecounter();                   \
            a.stamps[slot_] += (double)(now_ - st_prev);                                      \
            st_prev = now_;                                                                   \
        }                                                                                     \
    } while (0)
    extern __shared__ double2 lsm[];
    __shared__ double red[kLineThreads / 64];
    __shared__ double fin[3 * kCoopMaxG];
    __shared__ int sflag;
    const int tid = threadIdx.x, wg = blockIdx.x, sys = blockIdx.y, G = a.G;
    const int n0 = (int)a.g.n[0], n1 = (int)a.g.n[1], F0 = (int)a.g.F[0], F1 = (int)a.g.F[1];
    const int h0 = (n0 - 1) / 2, h1 = (n1 - 1) / 2, nh = h0 + 1, halfF1 = F1 >> 1;
    const int ldr = F1 + 1, ldc = F0 + 1;
    const int lgC = ilog2(a.lpbc);
    const unsigned magicF1 = 0xFFFFFFFFu / (unsigned)F1 + 1u;
    const int bufsz = max(a.lines * ldr, a.lpbc * ldc);
    double2* A = lsm;
    double2* B = lsm + bufsz;
    double2* tw1s = B + bufsz;
    double2* tw0s = F0 == F1 ? tw1s : tw1s + F1;
    load_twiddles(tw1s, a.tw1, F1);
    if (F0 != F1) load_twiddles(tw0s, a.tw0, F0);
    // the real spectra of this workgroup's column pairs, [t0][pair] as (S_a, S_b): read inside every column phase
    double2* specL = tw0s + F0;
    if (a.spec_lds) {
        const int c_first = wg * a.cols_wg;
        for (int w = tid; w < F0 * a.lpbc; w += kLineThreads) {
            const int k = w / a.lpbc, l = w - k * a.lpbc;
            specL[w] = make_double2(a.vhat[(int64_t)k * F1 + c_first + l].x, a.vhat[(int64_t)k * F1 + c_first + l + (F1 >> 1)].x);
        }
    }
    __syncthreads();
    const int64_t M = a.g.M;
    const int r0 = wg * a.rows_wg;                            // first owned row k0
    const int nrows = max(0, min(a.rows_wg, nh - r0));
    const int cnt = nrows * n1;
    const int64_t base = (int64_t)sys * M;
    const int p_lo = wg * a.cols_wg;                          // first owned column pair
    double2* b1 = a.b1 + (int64_t)sys * n0 * F1;              // row transforms of the vector just handed to phase R
    double2* b2 = a.b2 + (int64_t)sys * n0 * F1;              // column phase output
    double2* b3 = a.b3 + (int64_t)sys * n0 * F1;              // row transforms of ws .* p, carried from iteration to iteration
    double* part = a.partial + (int64_t)sys * 3 * kCoopMaxG;
    unsigned* bar = a.bar + (int64_t)sys * 16;
    unsigned epoch = 0;

    double2 xv[KS], rv[KS], pv[KS];
    double wsr[KS], dg[KS], wgt[KS];
    int lrow[KS], pos1[KS], tix[KS];                          // owned element s: row k0 = r0 + lrow, torus position of k1, flat index
    bool ok[KS];
    double ws_bad = 0.0;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int e = tid + s * kLineThreads;
        ok[s] = e < cnt;
        lrow[s] = ok[s] ? e / n1 : 0;
        const int lcol = ok[s] ? e - lrow[s] * n1 : 0;
        const int k0 = r0 + lrow[s];
        pos1[s] = (lcol - h1 + F1) % F1;
        tix[s] = (h0 + k0) * n1 + lcol;
        wgt[s] = k0 == 0 ? 1.0 : 2.0;
        if (ok[s]) {
            xv[s] = a.zero_x0 ? make_double2(0.0, 0.0) : a.x[base + tix[s]];
            const double2 w = a.ws[tix[s]], wm = a.ws[M - 1 - tix[s]];
            wsr[s] = w.x;
            ws_bad += w.y * w.y + (w.x - wm.x) * (w.x - wm.x) + wm.y * wm.y;
            dg[s] = coop_jacobi(a, w, tix[s]);
        } else {
            xv[s] = make_double2(0.0, 0.0);
            wsr[s] = 0.0;
            dg[s] = 1.0;
            wgt[s] = 0.0;
        }
        rv[s] = pv[s] = make_double2(0.0, 0.0);
    }
    auto sync_grid = [&]() __attribute__((always_inline)) -> bool {
        if (SOLO) {
            __syncthreads();
            return true;
        }
        return coop_barrier(bar, epoch, a.bar_need, a.status, &sflag);
    };
    auto all_sum = [&](double (&v)[3], int K, int slot0) __attribute__((always_inline)) -> bool {
        double a0 = v[0], a1 = K > 1 ? v[1] : 0.0;
        for (int off = 32; off > 0; off >>= 1) {
            a0 += __shfl_down(a0, off, 64);
            a1 += __shfl_down(a1, off, 64);
        }
        const int lane = tid & 63, wid = tid >> 6;
        __syncthreads();
        if (lane == 0) {
            red[wid] = a0;
            fin[wid] = a1;
        }
        __syncthreads();
        a0 = ((red[0] + red[1]) + red[2]) + red[3];
        a1 = ((fin[0] + fin[1]) + fin[2]) + fin[3];
        if (SOLO) {
            v[0] = a0;
            v[1] = a1;
            __syncthreads();
            return true;
        }
        if (tid == 0) {
            __hip_atomic_store(&part[slot0 * kCoopMaxG + wg], a0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (K > 1) __hip_atomic_store(&part[(slot0 + 1) * kCoopMaxG + wg], a1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!coop_barrier(bar, epoch, a.bar_need, a.status, &sflag)) return false;
        double b0 = lane < G ? __hip_atomic_load(&part[slot0 * kCoopMaxG + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        double b1v = (K > 1 && lane < G) ? __hip_atomic_load(&part[(slot0 + 1) * kCoopMaxG + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        for (int off = 32; off > 0; off >>= 1) {
            b0 += __shfl_xor(b0, off, 64);
            b1v += __shfl_xor(b1v, off, 64);
        }
        v[0] = b0;
        v[1] = b1v;
        return true;
    };
    // R: row transforms of ws .* u for the owned rows k0 (mode k1 at position k1 mod F1) -> b1
    auto phase_rows = [&](const double2 (&u)[KS]) __attribute__((always_inline)) {
        for (int p0 = 0; p0 < nrows; p0 += a.lines) {
            const int nl = min(a.lines, nrows - p0);
            for (int l = 0; l < nl; ++l)                                      // zeros between the two ends of the mode range
                for (int i1 = h1 + 1 + tid; i1 < F1 - h1; i1 += kLineThreads) A[l * ldr + i1] = make_double2(0.0, 0.0);
#pragma unroll
            for (int s = 0; s < KS; ++s)
                if (ok[s] && lrow[s] >= p0 && lrow[s] < p0 + nl) A[(lrow[s] - p0) * ldr + pos1[s]] = make_double2(wsr[s] * u[s].x, wsr[s] * u[s].y);
            __syncthreads();
            const double2* X = line_fft_fast(A, B, F1, ldr, nl, tw1s);
            for (int w = tid; w < nl * F1; w += kLineThreads) {
                const int l = coop_div(w, magicF1), i1 = w - l * F1;
                store_x2<SOLO>(b1 + (int64_t)(r0 + p0 + l) * F1 + i1, X[l * ldr + i1]);
            }
            __syncthreads();
        }
    };
    // C: owned column pairs (q, q + F1/2) in passes of a.lpbc lines.  mode 0: the columns of b1 as they are (operator on a given
    // vector); 1: the same and b3 <- b1 (first direction); 2: b3 <- b1 + beta b3 first -- the row transform of ws .* p follows the
    // direction's own recurrence p = z + beta p, so phase R could run on z BEFORE beta was known (see the iteration below).
    auto phase_cols = [&](int mode, double beta, double& cs) __attribute__((always_inline)) {
        for (int c0 = p_lo; c0 < p_lo + a.cols_wg; c0 += a.lpbc) {
            double2 ta[kCoopLoads / 2], tb[kCoopLoads / 2], tc[kCoopLoads / 2], td[kCoopLoads / 2];
#pragma unroll
            for (int q = 0; q < kCoopLoads / 2; ++q) {                        // every load is issued before the first is used
                const int w = tid + q * kLineThreads, l = w & (a.lpbc - 1), k0 = w >> lgC;
                const bool in = k0 < nh;
                const int64_t o = (int64_t)k0 * F1 + c0 + l;
                ta[q] = in ? load_x2<SOLO>(b1 + o) : make_double2(0.0, 0.0);
                tb[q] = in ? load_x2<SOLO>(b1 + o + halfF1) : make_double2(0.0, 0.0);
                tc[q] = (in && mode == 2) ? load_x2<SOLO>(b3 + o) : make_double2(0.0, 0.0);
                td[q] = (in && mode == 2) ? load_x2<SOLO>(b3 + o + halfF1) : make_double2(0.0, 0.0);
            }
            for (int w = tid; w < a.lpbc * (F0 - 2 * h0 - 1); w += kLineThreads) {          // zeros between the two ends
                const int l = w & (a.lpbc - 1), i0 = h0 + 1 + (w >> lgC);
                A[l * ldc + i0] = make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int q = 0; q < kCoopLoads / 2; ++q) {
                const int w = tid + q * kLineThreads, l = w & (a.lpbc - 1), k0 = w >> lgC;
                if (k0 < nh) {
                    double2 ga = ta[q], gb = tb[q];
                    if (mode == 2) {
                        ga = make_double2(ga.x + beta * tc[q].x, ga.y + beta * tc[q].y);
                        gb = make_double2(gb.x + beta * td[q].x, gb.y + beta * td[q].y);
                    }
                    if (mode != 0) {
                        const int64_t o = (int64_t)k0 * F1 + c0 + l;
                        store_x2<SOLO>(b3 + o, ga);
                        store_x2<SOLO>(b3 + o + halfF1, gb);
                    }
                    if (k0 == 0) {
                        A[l * ldc] = make_double2(ga.x, gb.x);                                    // real part of line k0 = 0
                    } else {
                        A[l * ldc + k0] = make_double2(ga.x - gb.y, ga.y + gb.x);                 // G_a + i G_b
                        A[l * ldc + F0 - k0] = make_double2(ga.x + gb.y, gb.x - ga.y);            // conj G_a + i conj G_b
                    }
                }
            }
            __syncthreads();
            COOP_STAMP(4);
            double2* X = a.spec_lds ? line_fft_fast_mul4(A, B, F0, ldc, a.lpbc, tw0s, specL, a.lpbc, cs)
                                    : line_fft_fast_mul2(A, B, F0, ldc, a.lpbc, tw0s, a.vhat + c0, F1, halfF1, cs);
            double2* Y = X == A ? B : A;
            const double2* Z = line_fft_fast(X, Y, F0, ldc, a.lpbc, tw0s);    // Z = conj T (T = packed, halved result)
            COOP_STAMP(5);
            for (int w = tid; w < (nh << lgC); w += kLineThreads) {
                const int l = w & (a.lpbc - 1), k0 = w >> lgC;
                const double2 zp = Z[l * ldc + k0], zm = Z[l * ldc + (k0 == 0 ? 0 : F0 - k0)];
                const double px = zp.x, py = -zp.y, mx = zm.x, my = -zm.y;       // T[k0], T[-k0]
                store_x2<SOLO>(b2 + (int64_t)k0 * F1 + c0 + l, make_double2(px + mx, py - my));            // T + conj T(-)
                store_x2<SOLO>(b2 + (int64_t)k0 * F1 + c0 + l + halfF1, make_double2(py + my, mx - px));   // (T - conj T(-)) / i
            }
            __syncthreads();
            COOP_STAMP(6);
        }
    };
    // Ri: inverse row transforms of b2 for the owned rows, crop, A u = ws .* (.) + sigma^2 u (or / sigma^2 + u)
    auto phase_rows_inv = [&](const double2 (&u)[KS], double2 (&Au)[KS]) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < KS; ++s) Au[s] = make_double2(0.0, 0.0);
        for (int p0 = 0; p0 < nrows; p0 += a.lines) {
            const int nl = min(a.lines, nrows - p0);
            double2 tmp[kCoopLoads];
#pragma unroll
            for (int q = 0; q < kCoopLoads; ++q) {
                const int w = tid + q * kLineThreads;
                tmp[q] = w < nl * F1 ? load_x2<SOLO>(b2 + (int64_t)(r0 + p0) * F1 + w) : make_double2(0.0, 0.0);   // rows are contiguous
            }
#pragma unroll
            for (int q = 0; q < kCoopLoads; ++q) {
                const int w = tid + q * kLineThreads;
                if (w < nl * F1) A[(w / F1) * ldr + (w % F1)] = make_double2(tmp[q].x, -tmp[q].y);
            }
            __syncthreads();
            COOP_STAMP(8);
            const double2* X = line_fft_fast(A, B, F1, ldr, nl, tw1s);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (ok[s] && lrow[s] >= p0 && lrow[s] < p0 + nl) {
                    const double2 z = X[(lrow[s] - p0) * ldr + pos1[s]];
                    const double2 gq = make_double2(wsr[s] * z.x, -wsr[s] * z.y);
                    if (a.variant == 0) Au[s] = make_double2(gq.x + a.sigmasq * u[s].x, gq.y + a.sigmasq * u[s].y);
                    else Au[s] = make_double2(gq.x / a.sigmasq + u[s].x, gq.y / a.sigmasq + u[s].y);
                }
            }
            __syncthreads();
        }
    };
    auto fill_nan = [&]() {
#pragma unroll
        for (int s = 0; s < KS; ++s)
            if (ok[s]) {
                a.x[base + tix[s]] = make_double2(__builtin_nan(""), __builtin_nan(""));
                a.x[base + M - 1 - tix[s]] = make_double2(__builtin_nan(""), __builtin_nan(""));
            }
    };
    auto dead = [&]() {
        if (wg == 0 && tid == 0) a.iters[sys] = -3;
        if (a.nan_on_dead) fill_nan();
    };

    // the contract: b conjugate-even, ws real and even (rounding leaves ~1e-32 |b|^2)
    double2 bv[KS];
    double acc[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        bv[s] = make_double2(0.0, 0.0);
        if (ok[s]) {
            bv[s] = a.b[base + tix[s]];
            double2 bm = a.b[base + M - 1 - tix[s]];
            if (a.b_times_ws) {                     // ws is real and even here (anything else trips ws_bad above)
                bv[s] = make_double2(wsr[s] * bv[s].x, wsr[s] * bv[s].y);
                bm = make_double2(wsr[s] * bm.x, wsr[s] * bm.y);
            }
            acc[0] += (bv[s].x - bm.x) * (bv[s].x - bm.x) + (bv[s].y + bm.y) * (bv[s].y + bm.y);
            acc[1] += wgt[s] * (bv[s].x * bv[s].x + bv[s].y * bv[s].y);
        }
    }
    acc[0] += ws_bad > 0.0 ? 1e300 : 0.0;
    if (!all_sum(acc, 2, 0)) return dead();
    const double bb = acc[1];
    if (!(acc[0] <= 1e-16 * bb)) {
        fill_nan();
        if (wg == 0 && tid == 0) a.iters[sys] = -2;
        return;
    }
    // r_0 = b - A x_0: one plain operator application (R | barrier | C | barrier | Ri)
    const bool precond = a.diag != nullptr || a.diag_scale != nullptr;
    double2 Ap[KS];
    if (a.zero_x0) {
#pragma unroll
        for (int s = 0; s < KS; ++s) Ap[s] = make_double2(0.0, 0.0);
    } else {
        double cs = 0.0;
        phase_rows(xv);
        if (!sync_grid()) return dead();
        phase_cols(0, 0.0, cs);
        if (!sync_grid()) return dead();
        phase_rows_inv(xv, Ap);
    }
    double2 zv[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        rv[s] = ok[s] ? make_double2(bv[s].x - Ap[s].x, bv[s].y - Ap[s].y) : make_double2(0.0, 0.0);
        zv[s] = precond ? make_double2(rv[s].x / dg[s], rv[s].y / dg[s]) : rv[s];
    }
    const double bn = sqrt(bb);
    const double den = bn > 0.0 ? bn : 1.0;
    // Iteration with TWO grid barriers (the complex kernel needs three): the all-reduce of <r,r>, <r,z> rides on the barrier
    // between the row and the column phase of the NEXT operator application.  That application needs p = z + beta p, and beta
    // needs <r,z> -- but the row transform is linear: phase R transforms ws .* z while beta is unknown, and the column phase
    // forms  rows(ws p) = rows(ws z) + beta rows(ws p_prev)  from the copy b3 it keeps of the previous direction's row
    // transforms.  Same recurrences as cg.py:116-150, the direction's transform updated by the direction's own recurrence.
    //   top of trip i: r_i, z_i local; p_{i-1}, <r,z>_{i-1} known; b3 = rows(ws p_{i-1})
    double rz = 0.0;
    int it = 0;
    st_prev = (long long)__builtin_readcyclecounter();
    for (;;) {
        acc[0] = acc[1] = 0.0;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            acc[0] += wgt[s] * (rv[s].x * rv[s].x + rv[s].y * rv[s].y);
            acc[1] += wgt[s] * (rv[s].x * zv[s].x + rv[s].y * zv[s].y);
        }
        COOP_STAMP(0);
        phase_rows(zv);
        COOP_STAMP(1);
        if (!all_sum(acc, 2, 0)) return dead();                               // barrier 1: <r,r>, <r,z> of r_i
        COOP_STAMP(2);
        const double rnorm = sqrt(acc[0]), rzn = acc[1];
        bool conv = false;
        if (it > 0) {
            if (a.hist && sys == 0 && wg == 0 && tid == 0 && it <= a.hist_cap) a.hist[it - 1] = rnorm / (den + kDivEps);
            conv = a.early_stop && ((rnorm / (den + kDivEps) < a.tol) || (a.batched && rnorm < 1e-12));
            if (conv) break;                                                  // cg.py:132 / :229-241 (p is not an output)
        }
        if (it >= a.max_iter) break;
        const double beta = it > 0 ? rzn / (rz + kDivEps) : 0.0;
        double pp = 0.0, cs = 0.0;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            pv[s] = make_double2(zv[s].x + beta * pv[s].x, zv[s].y + beta * pv[s].y);
            pp += wgt[s] * (pv[s].x * pv[s].x + pv[s].y * pv[s].y);
        }
        rz = rzn;
        COOP_STAMP(3);
        phase_cols(it > 0 ? 2 : 1, beta, cs);
        double t3[3] = {a.variant == 0 ? a.sigmasq * pp + cs : pp + cs / a.sigmasq, 0.0, 0.0};
        if (!all_sum(t3, 1, 2)) return dead();                                // barrier 2: <p, A p>
        COOP_STAMP(7);
        phase_rows_inv(pv, Ap);
        COOP_STAMP(9);
        const double alpha = rz / (t3[0] + kDivEps);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            xv[s].x += alpha * pv[s].x;
            xv[s].y += alpha * pv[s].y;
            rv[s].x -= alpha * Ap[s].x;
            rv[s].y -= alpha * Ap[s].y;
            zv[s] = precond ? make_double2(rv[s].x / dg[s], rv[s].y / dg[s]) : rv[s];
        }
        ++it;
    }
#pragma unroll
    for (int s = 0; s < KS; ++s)
        if (ok[s]) {
            a.x[base + tix[s]] = xv[s];
            if (r0 + lrow[s] > 0) a.x[base + M - 1 - tix[s]] = make_double2(xv[s].x, -xv[s].y);        // mode -k
        }
    if (wg == 0 && tid == 0) a.iters[sys] = it;
#undef COOP_STAMP
}

// ==========================================================================================================
// The same for 3-D grids (F = 64..256 per dimension): five launches of pruned batched line transforms.
//   fwd2 : (deferred p update) ws .* p, FFT along dim 2 of the n0*n1 non-zero lines            -> B1[n0][n1][F2]
//   fwd1 : per (i0, group of t2): FFT along dim 1 of n1 non-zero inputs                        -> B2[n0][F1][F2]
//   mid0 : per (t1, group of t2): FFT along dim 0 of n0 non-zero inputs, .* vhat, inverse FFT,
//          crop-window rows, in place                                                           -> B2[n0][F1][F2]
//   inv1 : per (j0, group of t2): inverse FFT along dim 1, crop-window rows                    -> B1[n0][n1][F2]
//   inv2 : inverse FFT along dim 2 of the n0*n1 window lines, crop, A p, <p, A p>
//   The generic iteration moves the full F^3 grid (4 MB at 64^3) through ~12 passes per matvec; these kernels touch
//   n0 F1 F2 elements at most (1.5 MB) four times.  `lpb` lines per workgroup (a power of two chosen by the host).
// ==========================================================================================================
struct Line3Args {
    CgArgs c;
    const double2* vhat;      // [F0][F1][F2], already divided by F0*F1*F2
    const double2* tw[3];     // exp(-2 pi i q / F[a])
    double2* b1;              // [slots][n0][n1][F2]
    double2* b2;              // [slots][n0][F1][F2]
    int lpb_c;                // lines per workgroup, contiguous kernels (fwd2, inv2)
    int lpb_s;                // adjacent t2 columns per workgroup, strided kernels (fwd1, mid0, inv1)
    int nblk_lines;           // workgroups of inv2 per system: partial sums of <p, A p>
};

__global__ __launch_bounds__(kLineThreads) void cg3_fwd2_kernel(Line3Args a) {
    extern __shared__ double2 lsm[];
    const CgArgs& c = a.c;
    const int slot = blockIdx.y;
    const int row = c.rows ? c.rows[slot] : slot;
    if (row < 0) return;
    const CgRowScalars sc = c.sc[row];
    if (!sc.active) return;
    const int n0 = (int)c.g.n[0], n1 = (int)c.g.n[1], n2 = (int)c.g.n[2], F2 = (int)c.g.F[2], ld = F2 + 1;
    const int nlines = n0 * n1;
    const int l0 = blockIdx.x * a.lpb_c;
    const int nl = min(a.lpb_c, nlines - l0);
    double2* A = lsm;
    double2* B = lsm + a.lpb_c * ld;
    double2* tws = B + a.lpb_c * ld;
    load_twiddles(tws, a.tw[2], F2);
    const int64_t base = (int64_t)row * c.g.M;
    for (int w = threadIdx.x; w < nl * F2; w += kLineThreads) {
        const int l = w / F2, i2 = w - l * F2;
        double2 v = make_double2(0.0, 0.0);
        if (i2 < n2) {
            const int t = (l0 + l) * n2 + i2;                     // flat block index: line = i0*n1 + i1
            double2 pv = c.p[base + t];
            if (sc.do_p) {                                       // deferred p <- r/diag + beta p
                double2 zv = c.r[base + t];
                if (c.diag) {
                    zv.x /= c.diag[t];
                    zv.y /= c.diag[t];
                }
                pv = make_double2(zv.x + sc.beta * pv.x, zv.y + sc.beta * pv.y);
                c.p[base + t] = pv;
            }
            v = cmul(pv, c.ws[t]);
        }
        A[l * ld + i2] = v;
    }
    __syncthreads();
    const double2* X = line_fft_any(A, B, F2, ld, nl, tws);
    double2* out = a.b1 + ((int64_t)slot * nlines + l0) * F2;
    for (int w = threadIdx.x; w < nl * F2; w += kLineThreads) {
        const int l = w / F2, i2 = w - l * F2;
        out[(int64_t)l * F2 + i2] = X[l * ld + i2];
    }
}

// MODE 0: forward along dim 1 (B1 -> B2);  MODE 1: inverse along dim 1 with crop (B2 -> B1)
template <int MODE>
__global__ __launch_bounds__(kLineThreads) void cg3_dim1_kernel(Line3Args a) {
    extern __shared__ double2 lsm[];
    const CgArgs& c = a.c;
    const int slot = blockIdx.y;
    const int row = c.rows ? c.rows[slot] : slot;
    if (row < 0) return;
    if (!c.sc[row].active) return;
    const int n0 = (int)c.g.n[0], n1 = (int)c.g.n[1], F1 = (int)c.g.F[1], F2 = (int)c.g.F[2], ld = F1 + 1;
    const int L = a.lpb_s;
    const int groups = F2 / L;
    const int i0 = blockIdx.x / groups, c0 = (blockIdx.x - i0 * groups) * L;      // plane i0 (or j0), first t2 column
    double2* A = lsm;
    double2* B = lsm + L * ld;
    double2* tws = B + L * ld;
    load_twiddles(tws, a.tw[1], F1);
    const double2* b1 = a.b1 + (int64_t)slot * n0 * n1 * F2;
    double2* b2 = a.b2 + (int64_t)slot * n0 * F1 * F2;
    if (MODE == 0) {
        for (int w = threadIdx.x; w < L * F1; w += kLineThreads) {
            const int i1 = w / L, l = w - i1 * L;
            A[l * ld + i1] = i1 < n1 ? b1[((int64_t)i0 * n1 + i1) * F2 + c0 + l] : make_double2(0.0, 0.0);
        }
    } else {
        for (int w = threadIdx.x; w < L * F1; w += kLineThreads) {
            const int t1 = w / L, l = w - t1 * L;
            const double2 v = b2[((int64_t)i0 * F1 + t1) * F2 + c0 + l];
            A[l * ld + t1] = make_double2(v.x, -v.y);
        }
    }
    __syncthreads();
    const double2* X = line_fft_any(A, B, F1, ld, L, tws);
    if (MODE == 0) {
        for (int w = threadIdx.x; w < L * F1; w += kLineThreads) {
            const int t1 = w / L, l = w - t1 * L;
            b2[((int64_t)i0 * F1 + t1) * F2 + c0 + l] = X[l * ld + t1];
        }
    } else {
        double2* out = a.b1 + (int64_t)slot * n0 * n1 * F2;
        for (int w = threadIdx.x; w < L * n1; w += kLineThreads) {
            const int j1 = w / L, l = w - j1 * L;
            const double2 z = X[l * ld + (n1 - 1) + j1];
            out[((int64_t)i0 * n1 + j1) * F2 + c0 + l] = make_double2(z.x, -z.y);
        }
    }
}

__global__ __launch_bounds__(kLineThreads) void cg3_mid0_kernel(Line3Args a) {
    extern __shared__ double2 lsm[];
    const CgArgs& c = a.c;
    const int slot = blockIdx.y;
    const int row = c.rows ? c.rows[slot] : slot;
    if (row < 0) return;
    if (!c.sc[row].active) return;
    const int n0 = (int)c.g.n[0], F0 = (int)c.g.F[0], F1 = (int)c.g.F[1], F2 = (int)c.g.F[2], ld = F0 + 1;
    const int L = a.lpb_s;
    const int groups = F2 / L;
    const int t1 = blockIdx.x / groups, c0 = (blockIdx.x - t1 * groups) * L;
    double2* A = lsm;
    double2* B = lsm + L * ld;
    double2* tws = B + L * ld;
    load_twiddles(tws, a.tw[0], F0);
    double2* b2 = a.b2 + (int64_t)slot * n0 * F1 * F2;
    for (int w = threadIdx.x; w < L * F0; w += kLineThreads) {
        const int i0 = w / L, l = w - i0 * L;
        A[l * ld + i0] = i0 < n0 ? b2[((int64_t)i0 * F1 + t1) * F2 + c0 + l] : make_double2(0.0, 0.0);
    }
    __syncthreads();
    double2* X = line_fft_any(A, B, F0, ld, L, tws);
    double2* Y = X == A ? B : A;
    for (int w = threadIdx.x; w < L * F0; w += kLineThreads) {
        const int t0 = w / L, l = w - t0 * L;
        const double2 m = cmul(X[l * ld + t0], a.vhat[((int64_t)t0 * F1 + t1) * F2 + c0 + l]);
        X[l * ld + t0] = make_double2(m.x, -m.y);
    }
    __syncthreads();
    const double2* Z = line_fft_any(X, Y, F0, ld, L, tws);
    for (int w = threadIdx.x; w < L * n0; w += kLineThreads) {
        const int j0 = w / L, l = w - j0 * L;
        const double2 z = Z[l * ld + (n0 - 1) + j0];
        b2[((int64_t)j0 * F1 + t1) * F2 + c0 + l] = make_double2(z.x, -z.y);
    }
}

__global__ __launch_bounds__(kLineThreads) void cg3_inv2_kernel(Line3Args a) {
    extern __shared__ double2 lsm[];
    __shared__ double red[kLineThreads / 64];
    __shared__ int flag;
    const CgArgs& c = a.c;
    const int slot = blockIdx.y;
    const int row = c.rows ? c.rows[slot] : slot;
    if (row < 0) return;
    if (!c.sc[row].active) return;
    const int n0 = (int)c.g.n[0], n1 = (int)c.g.n[1], n2 = (int)c.g.n[2], F2 = (int)c.g.F[2], ld = F2 + 1;
    const int nlines = n0 * n1;
    const int l0 = blockIdx.x * a.lpb_c;
    const int nl = min(a.lpb_c, nlines - l0);
    double2* A = lsm;
    double2* B = lsm + a.lpb_c * ld;
    double2* tws = B + a.lpb_c * ld;
    load_twiddles(tws, a.tw[2], F2);
    const double2* in = a.b1 + ((int64_t)slot * nlines + l0) * F2;
    for (int w = threadIdx.x; w < nl * F2; w += kLineThreads) {
        const int l = w / F2, i2 = w - l * F2;
        const double2 v = in[(int64_t)l * F2 + i2];
        A[l * ld + i2] = make_double2(v.x, -v.y);
    }
    __syncthreads();
    const double2* X = line_fft_any(A, B, F2, ld, nl, tws);
    const int64_t base = (int64_t)row * c.g.M;
    double pAp = 0.0;
    for (int w = threadIdx.x; w < nl * n2; w += kLineThreads) {
        const int l = w / n2, i2 = w - l * n2;
        const int t = (l0 + l) * n2 + i2;
        const double2 z = X[l * ld + (n2 - 1) + i2];
        const double2 pv = c.p[base + t];
        const double2 Ap = apply_A(c, c.ws[t], make_double2(z.x, -z.y), pv);
        c.ap[base + t] = Ap;
        pAp += pv.x * Ap.x + pv.y * Ap.y;
    }
    pAp = block_sum(pAp, red);
    double* part = c.partial + (int64_t)row * 3 * kCgBlocksMax;
    if (threadIdx.x == 0) store_agent(&part[blockIdx.x], pAp);
    if (!arrive_count(&c.counter[2 * row], a.nblk_lines, &flag)) return;
    __shared__ double fin[kCgBlocksMax];
    if (threadIdx.x < a.nblk_lines) fin[threadIdx.x] = load_agent(&part[threadIdx.x]);
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < a.nblk_lines; ++i) t += fin[i];
        c.sc[row].pAp = t + kDivEps;
    }
}

// ==========================================================================================================
// Round 3: the 3-D line iteration for HERMITIAN systems (vectors = Fourier coefficients of real functions on the symmetric mode
// box, ws real and even, Toeplitz vector Hermitian -- every system EFGP solves: efgpnd.py:119-141, 192-206, 926-941).
//   u[-k] = conj u[k]: only the planes k0 = 0..h0 are carried ((h0 + 1) n1 n2 of the M entries, the upper half of the flat
//   vector).  Modes sit CENTRED on the torus (k at k mod F), which makes the spectrum of the (Hermitian) Toeplitz vector real:
//   vc = Re(vhat e^(2 pi i sum f_a (n_a - 1) / F_a)), stored as doubles.
//     fwd2 / dim1<0> : as above on the (h0 + 1) planes: half the lines
//     mid0           : the k0-sequence of a column (t1, t2) is Hermitian, its transform real: TWO columns (t2, t2 + F2/2) ride
//                      through one complex transform as real and imaginary part (line_fft_inwave<R, 3>), are multiplied by
//                      their real spectra and come back through the second transform; unpacked as T + conj T(-) and
//                      (T - conj T(-)) / i.  Plane k0 = 0 keeps its real part only (projection on the Hermitian subspace:
//                      without it rounding noise in that plane is fed back, cf. cg_herm64_kernel).
//     dim1<1> / inv2 : back on the (h0 + 1) planes; <p, A p> counts plane 0 once and the others twice
//   Per operator application at mtot = 57 (F = 128): 66 MB through HBM/L2 instead of 150 MB (the spectrum alone 33.5 -> 16.8 MB).
// ==========================================================================================================
struct Line3HArgs {
    CgArgs c;
    const double* vc;         // [F0][F1][F2] real centred spectrum, already divided by F0*F1*F2
    const double2* tw[3];
    double2* b1;              // [slots][(h0+1) n1][F2]
    double2* b2;              // [slots][h0+1][F1][F2]
    int lpb_c;                // lines per workgroup, contiguous kernels (fwd2, inv2)
    int lpb_s;                // adjacent t2 columns per workgroup, dim-1 kernels
    int lpb_m;                // column pairs per workgroup, mid0
    int nblk_lines;           // workgroups of inv2 per system
};
// torus position -> index into the centred mode range of n = 2h+1 (or -1)
__device__ __forceinline__ int centred_index(int pos, int h, int F) { return pos <= h ? pos + h : (pos >= F - h ? pos - (F - h) : -1); }

__global__ __launch_bounds__(kLineThreads) void cg3h_fwd2_kernel(Line3HArgs a) {
    extern __shared__ double2 lsm[];
    const CgArgs& c = a.c;
    const int slot = blockIdx.y;
    const int row = c.rows ? c.rows[slot] : slot;
    if (row < 0) return;
    const CgRowScalars sc = c.sc[row];
    if (!sc.active) return;
    const int n0 = (int)c.g.n[0], n1 = (int)c.g.n[1], n2 = (int)c.g.n[2], F2 = (int)c.g.F[2], ld = F2 + 1;
    const int h2 = (n2 - 1) / 2, lgF2 = ilog2(F2);
    const int nlines = ((n0 + 1) / 2) * n1;
    const int l0 = blockIdx.x * a.lpb_c;
    const int nl = min(a.lpb_c, nlines - l0);
    double2* A = lsm;
    double2* B = lsm + a.lpb_c * ld;
    double2* tws = B + a.lpb_c * ld;
    load_twiddles(tws, a.tw[2], F2);
    const int64_t base = (int64_t)row * c.g.M;
    for (int w = threadIdx.x; w < (nl << lgF2); w += kLineThreads) {
        const int l = w >> lgF2, pos = w & (F2 - 1);
        const int i2 = centred_index(pos, h2, F2);
        double2 v = make_double2(0.0, 0.0);
        if (i2 >= 0) {
            const int64_t t = c.v_off + (int64_t)(l0 + l) * n2 + i2;
            double2 pv = c.p[base + t];
            if (sc.do_p) {                                       // deferred p <- r/diag + beta p
                double2 zv = c.r[base + t];
                if (c.diag) {
                    zv.x /= c.diag[t];
                    zv.y /= c.diag[t];
                }
                pv = make_double2(zv.x + sc.beta * pv.x, zv.y + sc.beta * pv.y);
                c.p[base + t] = pv;
            }
            const double wr = c.ws[t].x;
            v = make_double2(wr * pv.x, wr * pv.y);
        }
        A[l * ld + pos] = v;
    }
    __syncthreads();
    const double2* X = line_fft_any(A, B, F2, ld, nl, tws);
    double2* out = a.b1 + ((int64_t)slot * nlines + l0) * F2;
    for (int w = threadIdx.x; w < (nl << lgF2); w += kLineThreads) {
        const int l = w >> lgF2, i2 = w & (F2 - 1);
        out[(int64_t)l * F2 + i2] = X[l * ld + i2];
    }
}

// MODE 0: forward along dim 1 (b1 -> b2);  MODE 1: inverse along dim 1 with crop (b2 -> b1); planes k0 = 0..h0
template <int MODE>
__global__ __launch_bounds__(kLineThreads) void cg3h_dim1_kernel(Line3HArgs a) {
    extern __shared__ double2 lsm[];
    const CgArgs& c = a.c;
    const int slot = blockIdx.y;
    const int row = c.rows ? c.rows[slot] : slot;
    if (row < 0) return;
    if (!c.sc[row].active) return;
    const int n0 = (int)c.g.n[0], n1 = (int)c.g.n[1], F1 = (int)c.g.F[1], F2 = (int)c.g.F[2], ld = F1 + 1;
    const int nh = (n0 + 1) / 2, h1 = (n1 - 1) / 2;
    const int L = a.lpb_s, lgL = ilog2(L);
    const int groups = F2 / L;
    const int k0 = blockIdx.x / groups, c0 = (blockIdx.x - k0 * groups) * L;
    double2* A = lsm;
    double2* B = lsm + L * ld;
    double2* tws = B + L * ld;
    load_twiddles(tws, a.tw[1], F1);
    double2* b1 = a.b1 + (int64_t)slot * nh * n1 * F2;
    double2* b2 = a.b2 + (int64_t)slot * nh * F1 * F2;
    if (MODE == 0) {
        for (int w = threadIdx.x; w < (F1 << lgL); w += kLineThreads) {
            const int pos = w >> lgL, l = w & (L - 1);
            const int i1 = centred_index(pos, h1, F1);
            A[l * ld + pos] = i1 >= 0 ? b1[((int64_t)k0 * n1 + i1) * F2 + c0 + l] : make_double2(0.0, 0.0);
        }
    } else {
        for (int w = threadIdx.x; w < (F1 << lgL); w += kLineThreads) {
            const int t1 = w >> lgL, l = w & (L - 1);
            const double2 v = b2[((int64_t)k0 * F1 + t1) * F2 + c0 + l];
            A[l * ld + t1] = make_double2(v.x, -v.y);
        }
    }
    __syncthreads();
    const double2* X = line_fft_any(A, B, F1, ld, L, tws);
    if (MODE == 0) {
        for (int w = threadIdx.x; w < (F1 << lgL); w += kLineThreads) {
            const int t1 = w >> lgL, l = w & (L - 1);
            b2[((int64_t)k0 * F1 + t1) * F2 + c0 + l] = X[l * ld + t1];
        }
    } else {
        for (int w = threadIdx.x; w < (n1 << lgL); w += kLineThreads) {
            const int i1 = w >> lgL, l = w & (L - 1);
            const double2 z = X[l * ld + ((i1 - h1) & (F1 - 1))];
            b1[((int64_t)k0 * n1 + i1) * F2 + c0 + l] = make_double2(z.x, -z.y);
        }
    }
}

// per (t1, group of L column pairs (c, c + F2/2)): packed transform along dim 0, real spectra, back, in place in b2
__global__ __launch_bounds__(kLineThreads) void cg3h_mid0_kernel(Line3HArgs a) {
    extern __shared__ double2 lsm[];
    const CgArgs& c = a.c;
    const int slot = blockIdx.y;
    const int row = c.rows ? c.rows[slot] : slot;
    if (row < 0) return;
    if (!c.sc[row].active) return;
    const int n0 = (int)c.g.n[0], F0 = (int)c.g.F[0], F1 = (int)c.g.F[1], F2 = (int)c.g.F[2], ld = F0 + 1;
    const int nh = (n0 + 1) / 2, h0 = nh - 1, halfF2 = F2 >> 1;
    const int L = a.lpb_m, lgL = ilog2(L);
    const int groups = halfF2 / L;
    const int t1 = blockIdx.x / groups, c0 = (blockIdx.x - t1 * groups) * L;
    double2* A = lsm;
    double2* B = lsm + L * ld;
    double2* tws = B + L * ld;
    // the spectrum values this thread multiplies by behind the first transform: requested before anything else, so that their
    // round trip runs beside the load of the lines and the transform instead of inside it
    const double* spec = a.vc + (int64_t)t1 * F2 + c0;
    const bool pre_ok = (F0 == 64 && L <= 32) || (F0 == 128 && L <= 16) || (F0 == 256 && L <= 8);
    double2 pre[8];
    if (pre_ok) {
        if (F0 == 64) spectrum_prefetch<1>(spec, (int64_t)F1 * F2, halfF2, L, pre);
        else if (F0 == 128) spectrum_prefetch<2>(spec, (int64_t)F1 * F2, halfF2, L, pre);
        else spectrum_prefetch<4>(spec, (int64_t)F1 * F2, halfF2, L, pre);
    }
    load_twiddles(tws, a.tw[0], F0);
    double2* b2 = a.b2 + (int64_t)slot * nh * F1 * F2;
    for (int w = threadIdx.x; w < ((F0 - 2 * h0 - 1) << lgL); w += kLineThreads) {           // zeros between the two ends
        const int l = w & (L - 1), i0 = h0 + 1 + (w >> lgL);
        A[l * ld + i0] = make_double2(0.0, 0.0);
    }
    for (int w = threadIdx.x; w < (nh << lgL); w += kLineThreads) {
        const int k0 = w >> lgL, l = w & (L - 1);
        const int64_t o = ((int64_t)k0 * F1 + t1) * F2 + c0 + l;
        const double2 ga = b2[o], gb = b2[o + halfF2];
        if (k0 == 0) {
            A[l * ld] = make_double2(ga.x, gb.x);                                       // real part of plane k0 = 0
        } else {
            A[l * ld + k0] = make_double2(ga.x - gb.y, ga.y + gb.x);                    // G_a + i G_b
            A[l * ld + F0 - k0] = make_double2(ga.x + gb.y, gb.x - ga.y);               // conj G_a + i conj G_b
        }
    }
    __syncthreads();
    double2* X;
    if (pre_ok) {
        if (F0 == 64) line_fft_inwave<1, 5>(A, B, ld, L, tws, pre);
        else if (F0 == 128) line_fft_inwave<2, 5>(A, B, ld, L, tws, pre);
        else line_fft_inwave<4, 5>(A, B, ld, L, tws, pre);
        X = B;
    } else if (F0 == 64 || F0 == 128 || F0 == 256 || F0 == 512) {
        X = line_fft_fast_mul3(A, B, F0, ld, L, tws, spec, (int64_t)F1 * F2, halfF2);
    } else {
        X = line_fft(A, B, F0, ld, L, tws);
        for (int w = threadIdx.x; w < (F0 << lgL); w += kLineThreads) {
            const int t0 = w >> lgL, l = w & (L - 1);
            const int64_t o = ((int64_t)t0 * F1 + t1) * F2 + c0 + l;
            const double2 v = X[l * ld + t0];
            X[l * ld + t0] = make_double2(0.5 * a.vc[o] * v.x, -0.5 * a.vc[o + halfF2] * v.y);
        }
        __syncthreads();
    }
    double2* Y = X == A ? B : A;
    const double2* Z = line_fft_any(X, Y, F0, ld, L, tws);                              // Z = conj T (T = packed, halved result)
    for (int w = threadIdx.x; w < (nh << lgL); w += kLineThreads) {
        const int k0 = w >> lgL, l = w & (L - 1);
        const double2 zp = Z[l * ld + k0], zm = Z[l * ld + ((F0 - k0) & (F0 - 1))];
        const double px = zp.x, py = -zp.y, mx = zm.x, my = -zm.y;                      // T[k0], T[-k0]
        const int64_t o = ((int64_t)k0 * F1 + t1) * F2 + c0 + l;
        b2[o] = make_double2(px + mx, py - my);                                         // T + conj T(-)
        b2[o + halfF2] = make_double2(py + my, mx - px);                                // (T - conj T(-)) / i
    }
}

__global__ __launch_bounds__(kLineThreads) void cg3h_inv2_kernel(Line3HArgs a) {
    extern __shared__ double2 lsm[];
    __shared__ double red[kLineThreads / 64];
    __shared__ int flag;
    const CgArgs& c = a.c;
    const int slot = blockIdx.y;
    const int row = c.rows ? c.rows[slot] : slot;
    if (row < 0) return;
    if (!c.sc[row].active) return;
    const int n0 = (int)c.g.n[0], n1 = (int)c.g.n[1], n2 = (int)c.g.n[2], F2 = (int)c.g.F[2], ld = F2 + 1;
    const int h2 = (n2 - 1) / 2, lgF2 = ilog2(F2);
    const int nlines = ((n0 + 1) / 2) * n1;
    const int l0 = blockIdx.x * a.lpb_c;
    const int nl = min(a.lpb_c, nlines - l0);
    double2* A = lsm;
    double2* B = lsm + a.lpb_c * ld;
    double2* tws = B + a.lpb_c * ld;
    load_twiddles(tws, a.tw[2], F2);
    const double2* in = a.b1 + ((int64_t)slot * nlines + l0) * F2;
    for (int w = threadIdx.x; w < (nl << lgF2); w += kLineThreads) {
        const int l = w >> lgF2, i2 = w & (F2 - 1);
        const double2 v = in[(int64_t)l * F2 + i2];
        A[l * ld + i2] = make_double2(v.x, -v.y);
    }
    __syncthreads();
    const double2* X = line_fft_any(A, B, F2, ld, nl, tws);
    const int64_t base = (int64_t)row * c.g.M;
    double pAp = 0.0;
    for (int w = threadIdx.x; w < nl * n2; w += kLineThreads) {
        const int l = w / n2, i2 = w - l * n2;
        const int64_t t = c.v_off + (int64_t)(l0 + l) * n2 + i2;
        const double2 z = X[l * ld + ((i2 - h2) & (F2 - 1))];
        const double2 pv = c.p[base + t];
        const double2 Ap = apply_A(c, make_double2(c.ws[t].x, 0.0), make_double2(z.x, -z.y), pv);
        c.ap[base + t] = Ap;
        pAp += (l0 + l < n1 ? 1.0 : 2.0) * (pv.x * Ap.x + pv.y * Ap.y);                 // plane k0 = 0 once, the others twice
    }
    pAp = block_sum(pAp, red);
    double* part = c.partial + (int64_t)row * 3 * kCgBlocksMax;
    if (threadIdx.x == 0) store_agent(&part[blockIdx.x], pAp);
    if (!arrive_count(&c.counter[2 * row], a.nblk_lines, &flag)) return;
    __shared__ double fin[kCgBlocksMax];
    if (threadIdx.x < a.nblk_lines) fin[threadIdx.x] = load_agent(&part[threadIdx.x]);
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < a.nblk_lines; ++i) t += fin[i];
        c.sc[row].pAp = t + kDivEps;
    }
}

// x[-k] = conj x[k] for the planes k0 > 0 (the iteration carried k0 >= 0 only)
__global__ __launch_bounds__(kVecThreads) void cg3h_mirror_kernel(CgArgs a) {
    const int row = blockIdx.y;
    double2* x = a.x + (int64_t)row * a.g.M;
    for (int64_t e = a.v_w1 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < a.v_len; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = a.v_off + e;
        const double2 v = x[t];
        x[a.g.M - 1 - t] = make_double2(v.x, -v.y);
    }
}

// the contract of the Hermitian iteration, checked on the data: out[0] = sum |b[t] - conj b[M-1-t]|^2 + |x0[t] - conj x0[M-1-t]|^2,
// out[1] = sum |b|^2 + |x0|^2 over all rows, out[2] = sum over t of (Im ws)^2 + (ws[t] - ws[M-1-t])^2
__global__ __launch_bounds__(kVecThreads) void cg_herm_check_kernel(const double2* __restrict__ b, const double2* __restrict__ x0,
                                                                    const double2* __restrict__ ws, int64_t M, int rows, double* out) {
    __shared__ double red[kVecThreads / 64];
    double viol = 0.0, bb = 0.0, wbad = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)rows * M; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / M, t = i - r * M;
        const double2 u = b[i], um = b[r * M + M - 1 - t], y = x0[i], ym = x0[r * M + M - 1 - t];
        viol += (u.x - um.x) * (u.x - um.x) + (u.y + um.y) * (u.y + um.y) + (y.x - ym.x) * (y.x - ym.x) + (y.y + ym.y) * (y.y + ym.y);
        bb += u.x * u.x + u.y * u.y + y.x * y.x + y.y * y.y;
        if (r == 0) {
            const double2 w = ws[t], wm = ws[M - 1 - t];
            wbad += w.y * w.y + (w.x - wm.x) * (w.x - wm.x);
        }
    }
    viol = block_sum(viol, red);
    bb = block_sum(bb, red);
    wbad = block_sum(wbad, red);
    if (threadIdx.x == 0) {
        atomicAdd(&out[0], viol);
        atomicAdd(&out[1], bb);
        atomicAdd(&out[2], wbad);
    }
}

// vc[f] = Re(vhat[f] e^(2 pi i sum_a f_a (n_a - 1) / F_a)): the real spectrum of the Hermitian Toeplitz vector with its lags centred
__global__ __launch_bounds__(256) void center_spectrum3_real_kernel(const double2* __restrict__ vhat, const double2* __restrict__ tw0,
                                                                    const double2* __restrict__ tw1, const double2* __restrict__ tw2,
                                                                    int n0, int n1, int n2, int F0, int F1, int F2, double* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)F0 * F1 * F2) return;
    const int f2 = (int)(t % F2), f1 = (int)((t / F2) % F1), f0 = (int)(t / ((int64_t)F1 * F2));
    const double2 a = tw0[((int64_t)(n0 - 1) * f0) & (F0 - 1)], b = tw1[((int64_t)(n1 - 1) * f1) & (F1 - 1)], cc = tw2[((int64_t)(n2 - 1) * f2) & (F2 - 1)];
    const double2 ab = cmul(a, b), abc = cmul(ab, cc);
    out[t] = cmul(vhat[t], make_double2(abc.x, -abc.y)).x;
}

// vhat_c[f] = vhat[f] e^(2 pi i (f0 (n0-1)/F0 + f1 (n1-1)/F1)): the spectrum of the Toeplitz vector circularly shifted so that the
// crop window of the product sits at [0, n) -- real for the Hermitian vectors of EFGP; the cooperative solve multiplies by it
// and reads <w, T w> = sum_f Re(vhat_c) |w^|^2 off the forward transform (Parseval)
__global__ __launch_bounds__(256) void center_spectrum_kernel(const double2* __restrict__ vhat, const double2* __restrict__ tw0,
                                                              const double2* __restrict__ tw1, int n0, int n1, int F0, int F1,
                                                              double2* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)F0 * F1) return;
    const int f0 = (int)(t / F1), f1 = (int)(t - (int64_t)f0 * F1);
    const double2 a = tw0[((int64_t)(n0 - 1) * f0) % F0], b = tw1[((int64_t)(n1 - 1) * f1) % F1];
    const double2 ph = make_double2(a.x * b.x - a.y * b.y, -(a.x * b.y + a.y * b.x));      // conj(a b)
    out[t] = cmul(vhat[t], ph);
}

template <bool AC, bool BC>
__global__ void vdot_real_kernel(const double* __restrict__ a, const double* __restrict__ b, int64_t n,
                                 double* __restrict__ partial) {
    __shared__ double red[kVecThreads / 64];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double ar, ai = 0.0, br, bi = 0.0;
        if (AC) {
            double2 u = reinterpret_cast<const double2*>(a)[i];
            ar = u.x;
            ai = u.y;
        } else {
            ar = a[i];
        }
        if (BC) {
            double2 v = reinterpret_cast<const double2*>(b)[i];
            br = v.x;
            bi = v.y;
        } else {
            br = b[i];
        }
        acc += ar * br + ai * bi;
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

}  // namespace efgp

using namespace efgp;

struct efgp_toeplitz_s {
    int device = 0;
    DeviceCtx* ctx = nullptr;
    ToepGeom g;
    int64_t Ls[3] = {1, 1, 1};
    double2* vhat = nullptr;
    double2* tw[3] = {nullptr, nullptr, nullptr};   // exp(-2 pi i q / F[a]) tables for the persistent CG
    bool persistent_ok = false;
    bool lines_ok = false;       // 2-D, power-of-two F in [128, 512]: fused line-FFT CG iteration
    bool lines3_ok = false;      // 3-D, power-of-two F in [64, 256]: the same with five pruned line passes
    // 2-D blocks of up to 16 x 16 modes (circulant grids 8^2 .. 32^2): the CG solves run on a 64 x 64 embedding instead, through
    // the specialised 64 x 64 kernels (any F >= 2 n - 1 embeds the Toeplitz product exactly; measured 3.1 us per iteration
    // against 9-11 us of the generic kernel on the 32 x 32 grid).  fft_shape / efgp_toeplitz_apply keep the reference's grid.
    double2* vhat_c = nullptr;   // lines_ok grids: centred spectrum for the cooperative solve (center_spectrum_kernel)
    // Round 4: the cooperative solve runs on the smallest grid of the in-wave transforms (64 R or 48 R: 96, 128, 192, 256, 384, 512)
    // that holds 2 n - 1 -- 141 -> 192 instead of 256, 81 -> 96 instead of 128 (0.56 x the grid).  fft_shape, efgp_toeplitz_apply
    // and the multi-launch iteration keep the reference's power-of-two grid.
    bool coop_small = false;
    ToepGeom g_co;
    double2* vhat_co = nullptr;  // centred spectrum on g_co
    // ... and the spectrum on the reference's grid (`vhat`) is then made on first use (ensure_reference_spectrum) from a copy of
    // the Toeplitz vector: a fit only runs the cooperative solve, which never reads it (pad | two transform passes per operator)
    bool vhat_ready = true;
    double2* v_keep = nullptr;
    size_t v_keep_bytes = 0;
    double2* tw_co[2] = {nullptr, nullptr};
    double* vc3 = nullptr;       // lines3_ok grids: REAL centred spectrum of the Hermitian 3-D iteration, built on first use
    ToepGeom g_cg;
    double2* vhat_cg = nullptr;
    double2* tw_cg[3] = {nullptr, nullptr, nullptr};
    bool cg64 = false;
    // 2-D blocks of up to 23 x 23 modes (2 n - 1 <= 48): the HERMITIAN solves (the model's mean systems) run on the smallest
    // circulant grid, 48 x 48 (cg_herm48_kernel, round 4), from a third spectrum made in the same launch as the 64 x 64 one
    double2* vhat48 = nullptr;
    Herm48Operands h48 = {nullptr, nullptr};
};

// exp(-2 pi i q / n), q < n, on the device (cached per context)
static double2* twiddle_table_for(DeviceCtx* ctx, int64_t n, hipStream_t stream) {
    auto it = ctx->twiddles.find(n);
    if (it != ctx->twiddles.end()) return (double2*)it->second;
    std::vector<double2> tw((size_t)n);
    const long double two_pi = 2.0L * acosl(-1.0L);
    for (int64_t q = 0; q < n; ++q) {
        long double ang = -two_pi * (long double)q / (long double)n;
        tw[(size_t)q] = make_double2((double)cosl(ang), (double)sinl(ang));
    }
    double2* dtw = nullptr;
    if (hipMalloc((void**)&dtw, (size_t)n * sizeof(double2)) != hipSuccess ||
        hipMemcpyAsync(dtw, tw.data(), (size_t)n * sizeof(double2), hipMemcpyHostToDevice, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess) {
        if (dtw) (void)hipFree(dtw);
        (void)hipGetLastError();
        return nullptr;
    }
    ctx->twiddles[n] = dtw;
    return dtw;
}

// geometry, twiddles and spectrum the single-launch CG kernels use for this operator
static void cg_operands(const efgp_toeplitz_s* op, const ToepGeom** g, const double2* const** tw, const double2** vhat) {
    if (op->cg64) {
        *g = &op->g_cg;
        *tw = (const double2* const*)op->tw_cg;
        *vhat = op->vhat_cg;
    } else {
        *g = &op->g;
        *tw = (const double2* const*)op->tw;
        *vhat = op->vhat;
    }
}

namespace efgp {

static dim3 grid_for(int64_t work, int rows, int threads, int cap = 1024) {
    int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((work + threads - 1) / threads, cap));
    return dim3(blocks, rows);
}

// vhat = FFT(zero-padded v) / Ftot on the reference's grid, made on first use when the operator was created with a smaller
// cooperative grid (efgp_toeplitz_create)
static int ensure_reference_spectrum(efgp_toeplitz_s* op, hipStream_t stream) {
    if (op->vhat_ready) return EFGP_OK;
    ToepGeom gv = op->g;
    gv.M = 1;
    for (int a = 0; a < 3; ++a) {
        gv.n[a] = op->Ls[a];
        gv.M *= gv.n[a];
    }
    hipLaunchKernelGGL(pad_scale_kernel, grid_for(op->g.Ftot, 1, kVecThreads), dim3(kVecThreads), 0, stream, gv, (const double2*)op->v_keep,
                       gv.M, (const double2*)nullptr, (const int*)nullptr, (const int*)nullptr, op->vhat, 1.0 / (double)op->g.Ftot);
    EFGP_HIP_CHECK(hipGetLastError());
    int rc = fft_c2c(op->ctx, op->g.d, op->g.F, 1, op->vhat, true, stream);
    if (rc != EFGP_OK) return rc;
    op->vhat_ready = true;
    pool_free(op->ctx, op->v_keep, op->v_keep_bytes);        // stream-ordered reuse: the pad launch above read it on this stream
    op->v_keep = nullptr;
    return EFGP_OK;
}

// pad = FFT^-1( FFT(pad) .* vhat ) for `slots` rows
static int circulant(efgp_toeplitz_s* op, double2* pad, int slots, hipStream_t stream) {
    // in-house transforms know the padding: the input is non-zero on [0, n) per axis, the product is read on [n - 1, 2 n - 1)
    const bool own = own_fft_supported(op->g.d, op->g.F) && std::getenv("EFGP_NO_PRUNED_FFT") == nullptr;
    int64_t lo_in[3] = {0, 0, 0}, lo_out[3], cnt[3];
    for (int a = 0; a < 3; ++a) {
        cnt[a] = a < op->g.d ? op->g.n[a] : 1;
        lo_out[a] = a < op->g.d ? std::min(op->g.n[a] - 1, op->g.F[a] - op->g.n[a]) : 0;
    }
    int rc = ensure_reference_spectrum(op, stream);
    if (rc != EFGP_OK) return rc;
    rc = own ? own_fft_exec_windowed(op->ctx, op->g.d, op->g.F, slots, pad, true, lo_in, cnt, false, stream)
             : fft_c2c(op->ctx, op->g.d, op->g.F, slots, pad, true, stream);
    if (rc != EFGP_OK) return rc;
    hipLaunchKernelGGL(spectral_mul_kernel, grid_for(op->g.Ftot, slots, kVecThreads), dim3(kVecThreads), 0, stream,
                       op->g.Ftot, (const double2*)op->vhat, pad);
    EFGP_HIP_CHECK(hipGetLastError());
    return own ? own_fft_exec_windowed(op->ctx, op->g.d, op->g.F, slots, pad, false, lo_out, cnt, true, stream)
               : fft_c2c(op->ctx, op->g.d, op->g.F, slots, pad, false, stream);
}

// vhat_c = vhat rotated so that the circular convolution has its crop window at [0, n) (cooperative solve on the reference's grid)
static bool ensure_centred_spectrum(efgp_toeplitz_s* op, hipStream_t stream) {
    if (op->vhat_c) return true;
    if (ensure_reference_spectrum(op, stream) != EFGP_OK) return false;
    op->vhat_c = (double2*)pool_alloc(op->ctx, (size_t)op->g.Ftot * sizeof(double2));
    if (!op->vhat_c) return false;
    hipLaunchKernelGGL(center_spectrum_kernel, dim3((unsigned)((op->g.Ftot + 255) / 256)), dim3(256), 0, stream, op->vhat, op->tw[0],
                       op->tw[1], (int)op->g.n[0], (int)op->g.n[1], (int)op->g.F[0], (int)op->g.F[1], op->vhat_c);
    if (hipGetLastError() != hipSuccess) {
        pool_free(op->ctx, op->vhat_c, (size_t)op->g.Ftot * sizeof(double2));
        op->vhat_c = nullptr;
        return false;
    }
    return true;
}

}  // namespace efgp

extern "C" {

int efgp_toeplitz_create(efgp_toeplitz_t** op_out, int device, int dim, const int64_t* Ls, const void* v,
                         int force_pow2, void* stream_) {
    EFGP_REQUIRE(op_out && Ls && v, "efgp_toeplitz_create: null argument");
    EFGP_REQUIRE(dim >= 1 && dim <= 3, "efgp_toeplitz_create: dim must be 1, 2 or 3 (got %d)", dim);
    for (int a = 0; a < dim; ++a) EFGP_REQUIRE(Ls[a] >= 1, "efgp_toeplitz_create: Ls[%d] < 1", a);
    DeviceCtx* ctx = device_ctx(device);
    if (!ctx) return EFGP_EHIP;
    hipStream_t stream = (hipStream_t)stream_;
    DeviceGuard guard(device, (hipStream_t)stream_);
    auto* op = new efgp_toeplitz_s();
    op->device = device;
    op->ctx = ctx;
    op->g.d = dim;
    op->g.M = 1;
    op->g.Ftot = 1;
    for (int a = 0; a < 3; ++a) {
        op->Ls[a] = a < dim ? Ls[a] : 1;
        op->g.n[a] = a < dim ? (Ls[a] + 1) / 2 : 1;                       // efgpnd.py:1259
        op->g.F[a] = a < dim ? (force_pow2 ? next_pow2(Ls[a]) : next_smooth_even(Ls[a])) : 1;   // :1269
        op->g.M *= op->g.n[a];
        op->g.Ftot *= op->g.F[a];
    }
    op->vhat = (double2*)pool_alloc(ctx, (size_t)op->g.Ftot * sizeof(double2));
    if (!op->vhat) {
        delete op;
        return EFGP_ENOMEM;
    }
    // vhat = FFT(zero-padded v) / Ftot.  Reuse the pad kernel with n := L, i.e. a geometry whose block is L.
    ToepGeom gv = op->g;
    gv.M = 1;
    for (int a = 0; a < 3; ++a) {
        gv.n[a] = op->Ls[a];
        gv.M *= gv.n[a];
    }
    int rc = EFGP_OK;
    // Hermitian solves of blocks up to 23 x 23 run on the 48 x 48 circulant grid: its spectrum rides in the launch that makes the
    // 64 x 64 one (this grid's, or the embedding's below)
    const bool want48 = dim == 2 && op->g.n[0] == op->g.n[1] && (op->g.n[0] & 1) && op->g.n[0] <= 23 && op->g.F[0] <= 64 &&
                        op->g.F[0] == op->g.F[1] && persistent_cg_eligible(op->g) && std::getenv("EFGP_NO_CG48") == nullptr &&
                        std::getenv("EFGP_NO_CG64") == nullptr && std::getenv("EFGP_NO_CG_HERM") == nullptr;
    if (want48) {
        const double2* tw48 = twiddle_table_for(ctx, 48, stream);
        op->vhat48 = tw48 ? (double2*)pool_alloc(ctx, (size_t)2304 * sizeof(double2)) : nullptr;
        if (op->vhat48) op->h48.tw = tw48;
    }
    bool made48 = false;
    // 2-D grids of the cooperative solve (128..512 per axis): when a smaller cooperative grid exists (made below) nothing on the
    // fit path reads the reference grid's spectrum -- keep a copy of v and make it on first use
    bool defer_ref = dim == 2 && std::getenv("EFGP_NO_COOP_SMALL") == nullptr && std::getenv("EFGP_EAGER_REF_SPECTRUM") == nullptr;
    for (int a = 0; a < dim && defer_ref; ++a) {
        const int64_t F = op->g.F[a];
        defer_ref = F >= 128 && F <= 512 && (F & (F - 1)) == 0;
    }
    if (defer_ref) {
        op->v_keep_bytes = (size_t)gv.M * sizeof(double2);
        op->v_keep = (double2*)pool_alloc(ctx, op->v_keep_bytes);
        defer_ref = op->v_keep != nullptr &&
                    hipMemcpyAsync(op->v_keep, v, op->v_keep_bytes, hipMemcpyDeviceToDevice, stream) == hipSuccess;
        if (!defer_ref && op->v_keep) {
            pool_free(ctx, op->v_keep, op->v_keep_bytes);
            op->v_keep = nullptr;
        }
        op->vhat_ready = !defer_ref;
    }
    if (defer_ref) {
        // nothing now
    } else if (toeplitz_vhat_fused_eligible(op->g)) {
        if (op->vhat48) {
            rc = toeplitz_vhat_pair_launch((const double2*)v, (int)op->Ls[0], (int)op->Ls[1], op->vhat, op->vhat48, stream);
            made48 = rc == EFGP_OK;
        } else {
            rc = toeplitz_vhat_fused_launch((const double2*)v, (int)op->Ls[0], (int)op->Ls[1], 1.0 / (double)op->g.Ftot, op->vhat,
                                            stream);
        }
    } else {
        hipLaunchKernelGGL(pad_scale_kernel, grid_for(op->g.Ftot, 1, kVecThreads), dim3(kVecThreads), 0, stream, gv,
                           (const double2*)v, gv.M, (const double2*)nullptr, (const int*)nullptr, (const int*)nullptr,
                           op->vhat, 1.0 / (double)op->g.Ftot);   // the inverse transform's 1/Ftot, folded in before the FFT
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            pool_free(ctx, op->vhat, (size_t)op->g.Ftot * sizeof(double2));
            delete op;
            set_error("efgp_toeplitz_create: pad launch failed: %s", hipGetErrorString(e));
            return EFGP_EHIP;
        }
        rc = fft_c2c(ctx, dim, op->g.F, 1, op->vhat, true, stream);
    }
    if (rc != EFGP_OK) {
        pool_free(ctx, op->vhat, (size_t)op->g.Ftot * sizeof(double2));
        delete op;
        return rc;
    }
    op->persistent_ok = persistent_cg_eligible(op->g);
    op->lines_ok = dim == 2;
    for (int a = 0; a < dim && op->lines_ok; ++a) {
        const int64_t F = op->g.F[a];
        op->lines_ok = F >= 128 && F <= 512 && (F & (F - 1)) == 0;
    }
    op->lines3_ok = dim == 3;
    for (int a = 0; a < dim && op->lines3_ok; ++a) {
        const int64_t F = op->g.F[a];
        op->lines3_ok = F >= 64 && F <= 256 && (F & (F - 1)) == 0;
    }
    if (op->persistent_ok || op->lines_ok || op->lines3_ok) {
        for (int a = 0; a < dim; ++a) {
            const int64_t n = op->g.F[a];
            auto it = ctx->twiddles.find(n);
            if (it != ctx->twiddles.end()) {
                op->tw[a] = (double2*)it->second;
                continue;
            }
            std::vector<double2> tw((size_t)n);
            const long double two_pi = 2.0L * acosl(-1.0L);
            for (int64_t q = 0; q < n; ++q) {
                long double ang = -two_pi * (long double)q / (long double)n;
                tw[(size_t)q] = make_double2((double)cosl(ang), (double)sinl(ang));
            }
            double2* dtw = nullptr;
            if (hipMalloc((void**)&dtw, (size_t)n * sizeof(double2)) != hipSuccess ||
                hipMemcpyAsync(dtw, tw.data(), (size_t)n * sizeof(double2), hipMemcpyHostToDevice, stream) != hipSuccess ||
                hipStreamSynchronize(stream) != hipSuccess) {
                if (dtw) (void)hipFree(dtw);
                op->persistent_ok = false;
                op->lines_ok = false;
                op->lines3_ok = false;
                break;
            }
            ctx->twiddles[n] = dtw;
            op->tw[a] = dtw;
        }
    }
    if (op->lines_ok && std::getenv("EFGP_NO_COOP_SMALL") == nullptr) {
        static const int64_t ladder[] = {96, 128, 192, 256, 384, 512};
        op->g_co = op->g;
        op->g_co.Ftot = 1;
        bool smaller = false;
        for (int a = 0; a < 2; ++a) {
            for (int64_t c : ladder)
                if (c >= op->Ls[a]) {
                    op->g_co.F[a] = c;
                    break;
                }
            smaller = smaller || op->g_co.F[a] < op->g.F[a];
            op->g_co.Ftot *= op->g_co.F[a];
        }
        if (smaller) {
            op->tw_co[0] = twiddle_table_for(ctx, op->g_co.F[0], stream);
            op->tw_co[1] = twiddle_table_for(ctx, op->g_co.F[1], stream);
            op->vhat_co = (op->tw_co[0] && op->tw_co[1]) ? (double2*)pool_alloc(ctx, (size_t)op->g_co.Ftot * sizeof(double2)) : nullptr;
            if (op->vhat_co) {
                // spectrum on the small grid: FFT(zero-padded v) / Ftot, then the centring rotation in place
                ToepGeom gp = op->g_co;
                gp.M = 1;
                for (int a = 0; a < 3; ++a) {
                    gp.n[a] = op->Ls[a];
                    gp.M *= gp.n[a];
                }
                hipLaunchKernelGGL(pad_scale_kernel, grid_for(gp.Ftot, 1, kVecThreads), dim3(kVecThreads), 0, stream, gp, (const double2*)v, gp.M,
                                   (const double2*)nullptr, (const int*)nullptr, (const int*)nullptr, op->vhat_co, 1.0 / (double)gp.Ftot);
                bool ok = hipGetLastError() == hipSuccess && fft_c2c(ctx, 2, op->g_co.F, 1, op->vhat_co, true, stream) == EFGP_OK;
                if (ok) {
                    hipLaunchKernelGGL(center_spectrum_kernel, dim3((unsigned)((op->g_co.Ftot + 255) / 256)), dim3(256), 0, stream, op->vhat_co,
                                       op->tw_co[0], op->tw_co[1], (int)op->g.n[0], (int)op->g.n[1], (int)op->g_co.F[0], (int)op->g_co.F[1],
                                       op->vhat_co);
                    ok = hipGetLastError() == hipSuccess;
                }
                if (ok) {
                    op->coop_small = true;
                } else {
                    (void)hipGetLastError();
                    pool_free(ctx, op->vhat_co, (size_t)op->g_co.Ftot * sizeof(double2));
                    op->vhat_co = nullptr;
                }
            }
        }
    }
    // the centred spectrum on the reference's grid: needed by the cooperative solve only when it runs there (no smaller grid, or
    // EFGP_NO_COOP_SMALL later on: built on first use then)
    if (!op->vhat_ready && !op->coop_small) {
        rc = ensure_reference_spectrum(op, stream);
        if (rc != EFGP_OK) {
            efgp_toeplitz_destroy(op);
            return rc;
        }
    }
    if (op->lines_ok && !op->coop_small) (void)ensure_centred_spectrum(op, stream);
    if (dim == 2 && op->persistent_ok && op->g.n[0] == op->g.n[1] && op->g.F[0] == op->g.F[1] && op->g.F[0] < 64 &&
        op->Ls[0] <= 63 && std::getenv("EFGP_NO_CG64_EMBED") == nullptr && std::getenv("EFGP_NO_CG64") == nullptr) {
        op->g_cg = op->g;
        op->g_cg.F[0] = op->g_cg.F[1] = 64;
        op->g_cg.Ftot = 64 * 64;
        op->vhat_cg = (double2*)pool_alloc(ctx, (size_t)4096 * sizeof(double2));
        bool ok = op->vhat_cg != nullptr;
        if (ok) {
            auto it = ctx->twiddles.find(64);
            if (it == ctx->twiddles.end()) {
                std::vector<double2> tw(64);
                const long double two_pi = 2.0L * acosl(-1.0L);
                for (int q = 0; q < 64; ++q) tw[(size_t)q] = make_double2((double)cosl(-two_pi * q / 64.0L), (double)sinl(-two_pi * q / 64.0L));
                double2* dtw = nullptr;
                ok = hipMalloc((void**)&dtw, 64 * sizeof(double2)) == hipSuccess &&
                     hipMemcpyAsync(dtw, tw.data(), 64 * sizeof(double2), hipMemcpyHostToDevice, stream) == hipSuccess &&
                     hipStreamSynchronize(stream) == hipSuccess;
                if (ok) ctx->twiddles[64] = dtw;
                else if (dtw) (void)hipFree(dtw);
            }
            if (ok) op->tw_cg[0] = op->tw_cg[1] = (double2*)ctx->twiddles[64];
        }
        if (ok) {
            if (op->vhat48 && !made48) {
                ok = toeplitz_vhat_pair_launch((const double2*)v, (int)op->Ls[0], (int)op->Ls[1], op->vhat_cg, op->vhat48, stream) == EFGP_OK;
                made48 = ok;
            } else {
                ok = toeplitz_vhat_fused_launch((const double2*)v, (int)op->Ls[0], (int)op->Ls[1], 1.0 / 4096.0, op->vhat_cg, stream) == EFGP_OK;
            }
        }
        if (ok) {
            op->cg64 = true;
        } else {
            (void)hipGetLastError();
            if (op->vhat_cg) pool_free(ctx, op->vhat_cg, (size_t)4096 * sizeof(double2));
            op->vhat_cg = nullptr;
        }
    }
    if (op->vhat48 && made48) {
        op->h48.vhat = op->vhat48;
    } else if (op->vhat48) {          // no 64 x 64 launch carried it (grid not on the fused path): the Hermitian solves keep 64 x 64
        pool_free(ctx, op->vhat48, (size_t)2304 * sizeof(double2));
        op->vhat48 = nullptr;
    }
    *op_out = op;
    return EFGP_OK;
}

int efgp_toeplitz_destroy(efgp_toeplitz_t* op) {
    if (!op) return EFGP_OK;
    DeviceGuard guard(op->device);
    // the spectrum block goes back to the pool; work already enqueued on the caller's stream that reads it
    // finishes before any later enqueue on that stream can overwrite a recycled block (single stream)
    if (op->vhat) pool_free(op->ctx, op->vhat, (size_t)op->g.Ftot * sizeof(double2));
    if (op->vhat_cg) pool_free(op->ctx, op->vhat_cg, (size_t)4096 * sizeof(double2));
    if (op->vhat48) pool_free(op->ctx, op->vhat48, (size_t)2304 * sizeof(double2));
    if (op->vhat_c) pool_free(op->ctx, op->vhat_c, (size_t)op->g.Ftot * sizeof(double2));
    if (op->vhat_co) pool_free(op->ctx, op->vhat_co, (size_t)op->g_co.Ftot * sizeof(double2));
    if (op->v_keep) pool_free(op->ctx, op->v_keep, op->v_keep_bytes);
    if (op->vc3) pool_free(op->ctx, op->vc3, (size_t)op->g.Ftot * sizeof(double));
    delete op;
    return EFGP_OK;
}

int efgp_toeplitz_fft_shape(efgp_toeplitz_t* op, int64_t* shape_out) {
    EFGP_REQUIRE(op && shape_out, "efgp_toeplitz_fft_shape: null argument");
    for (int a = 0; a < op->g.d; ++a) shape_out[a] = op->g.F[a];
    return EFGP_OK;
}

int efgp_toeplitz_single_launch_solves(efgp_toeplitz_t* op) {
    EFGP_REQUIRE(op, "efgp_toeplitz_single_launch_solves: null argument");
    return (op->persistent_ok && std::getenv("EFGP_NO_PERSISTENT_CG") == nullptr) ? 1 : 0;
}

int efgp_toeplitz_cg_shape(efgp_toeplitz_t* op, int hermitian, int64_t* shape_out) {
    EFGP_REQUIRE(op && shape_out, "efgp_toeplitz_cg_shape: null argument");
    for (int a = 0; a < op->g.d; ++a) shape_out[a] = op->g.F[a];
    if (op->g.d == 2 && op->persistent_ok) {
        if (hermitian && op->h48.vhat && std::getenv("EFGP_NO_CG48") == nullptr) shape_out[0] = shape_out[1] = 48;
        else if (op->cg64) shape_out[0] = shape_out[1] = 64;
    } else if (op->coop_small && std::getenv("EFGP_NO_COOP_SMALL") == nullptr && std::getenv("EFGP_NO_CG_COOP") == nullptr) {
        for (int a = 0; a < 2; ++a) shape_out[a] = op->g_co.F[a];
    }
    return EFGP_OK;
}

int efgp_toeplitz_apply(efgp_toeplitz_t* op, const void* x, int nbatch, void* y, void* stream_) {
    EFGP_REQUIRE(op && x && y, "efgp_toeplitz_apply: null argument");
    EFGP_REQUIRE(nbatch >= 1, "efgp_toeplitz_apply: nbatch must be >= 1");
    hipStream_t stream = (hipStream_t)stream_;
    DeviceGuard guard(op->device, (hipStream_t)stream_);
    // bound the scratch: process rows in chunks of at most ~256 MB of padded grid
    const int64_t max_rows = std::max<int64_t>(1, (int64_t)(256ll << 20) / (op->g.Ftot * (int64_t)sizeof(double2)));
    for (int64_t r0 = 0; r0 < nbatch; r0 += max_rows) {
        const int rows = (int)std::min<int64_t>(max_rows, nbatch - r0);
        double2* pad = (double2*)scratch(op->ctx, SLOT_TOEP_PAD, (size_t)rows * (size_t)op->g.Ftot * sizeof(double2));
        if (!pad) return EFGP_ENOMEM;
        hipLaunchKernelGGL(pad_scale_kernel, grid_for(op->g.Ftot, rows, kVecThreads), dim3(kVecThreads), 0, stream, op->g,
                           (const double2*)x + r0 * op->g.M, op->g.M, (const double2*)nullptr, (const int*)nullptr,
                           (const int*)nullptr, pad, 1.0);
        EFGP_HIP_CHECK(hipGetLastError());
        int rc = circulant(op, pad, rows, stream);
        if (rc != EFGP_OK) return rc;
        hipLaunchKernelGGL(crop_kernel, grid_for(op->g.M, rows, kVecThreads), dim3(kVecThreads), 0, stream, op->g,
                           (const double2*)pad, (double2*)y + r0 * op->g.M);
        EFGP_HIP_CHECK(hipGetLastError());
    }
    return EFGP_OK;
}

int efgp_toeplitz_apply_scaled(efgp_toeplitz_t* op, const void* x, int x_is_real, int nbatch, const void* pre, const void* post,
                               void* y, void* stream_) {
    EFGP_REQUIRE(op && x && y, "efgp_toeplitz_apply_scaled: null argument");
    EFGP_REQUIRE(nbatch >= 1, "efgp_toeplitz_apply_scaled: nbatch must be >= 1");
    EFGP_REQUIRE(x != y, "efgp_toeplitz_apply_scaled: x and y must not alias");
    hipStream_t stream = (hipStream_t)stream_;
    DeviceGuard guard(op->device, (hipStream_t)stream_);
    const ToepGeom* gq;
    const double2* const* twq;
    const double2* vq;
    cg_operands(op, &gq, &twq, &vq);
    if (toeplitz_apply_fused_eligible(*gq))
        return toeplitz_apply_fused_launch(*gq, twq[0], vq, (const double2*)pre, (const double2*)post, x, x_is_real, (double2*)y, nbatch,
                                           stream);
    const int64_t max_rows = std::max<int64_t>(1, (int64_t)(256ll << 20) / (op->g.Ftot * (int64_t)sizeof(double2)));
    for (int64_t r0 = 0; r0 < nbatch; r0 += max_rows) {
        const int rows = (int)std::min<int64_t>(max_rows, nbatch - r0);
        double2* pad = (double2*)scratch(op->ctx, SLOT_TOEP_PAD, (size_t)rows * (size_t)op->g.Ftot * sizeof(double2));
        if (!pad) return EFGP_ENOMEM;
        if (x_is_real)
            hipLaunchKernelGGL(pad_pre_kernel<true>, grid_for(op->g.Ftot, rows, kVecThreads), dim3(kVecThreads), 0, stream, op->g,
                               (const void*)((const double*)x + r0 * op->g.M), (const double2*)pre, pad);
        else
            hipLaunchKernelGGL(pad_pre_kernel<false>, grid_for(op->g.Ftot, rows, kVecThreads), dim3(kVecThreads), 0, stream, op->g,
                               (const void*)((const double2*)x + r0 * op->g.M), (const double2*)pre, pad);
        EFGP_HIP_CHECK(hipGetLastError());
        int rc = circulant(op, pad, rows, stream);
        if (rc != EFGP_OK) return rc;
        hipLaunchKernelGGL(crop_post_kernel, grid_for(op->g.M, rows, kVecThreads), dim3(kVecThreads), 0, stream, op->g,
                           (const double2*)pad, (const double2*)post, (double2*)y + r0 * op->g.M);
        EFGP_HIP_CHECK(hipGetLastError());
    }
    return EFGP_OK;
}

int efgp_internal_apply_scaled(efgp_toeplitz_s* op, const void* x, int x_is_real, int nbatch, const void* pre, int pre_stride,
                               const void* post, void* y, hipStream_t stream) {
    if (pre == nullptr || pre_stride == 1) return efgp_toeplitz_apply_scaled(op, x, x_is_real, nbatch, pre, post, y, stream);
    EFGP_REQUIRE(op && x && y && nbatch >= 1 && x != y && pre_stride >= 1, "efgp_internal_apply_scaled: bad argument");
    DeviceGuard guard(op->device, stream);
    const ToepGeom* gq;
    const double2* const* twq;
    const double2* vq;
    cg_operands(op, &gq, &twq, &vq);
    if (!toeplitz_apply_fused_eligible(*gq)) return EFGP_EUNSUPPORTED;
    return toeplitz_apply_fused_launch(*gq, twq[0], vq, (const double2*)pre, (const double2*)post, x, x_is_real, (double2*)y, nbatch, stream,
                                       pre_stride);
}

int efgp_internal_cg_single_launch(efgp_toeplitz_s* op, const void* ws, double sigmasq, int variant, const double* diag,
                                   const double* diag_scale, const void* b, int b_times_ws, void* x, int zero_x0, int nbatch, double tol,
                                   int max_iter, int early_stop, int batched_semantics, int* row_iters_dev, hipStream_t stream,
                                   int hermitian, const void* x0) {
    EFGP_REQUIRE(op && ws && b && x && row_iters_dev && nbatch >= 1, "efgp_internal_cg_single_launch: bad argument");
    EFGP_REQUIRE(batched_semantics || nbatch == 1, "efgp_internal_cg_single_launch: single-system semantics need nbatch == 1");
    if (!op->persistent_ok || std::getenv("EFGP_NO_PERSISTENT_CG") != nullptr) return EFGP_EUNSUPPORTED;
    DeviceGuard guard(op->device, stream);
    if (max_iter <= 0) max_iter = (int)std::min<int64_t>(2 * op->g.M, 2000000000);
    KernelTimer timer("cg_persistent", stream);
    const ToepGeom* gq;
    const double2* const* twq;
    const double2* vq;
    cg_operands(op, &gq, &twq, &vq);
    return persistent_cg_launch(*gq, twq, vq, (const double2*)ws, diag, sigmasq, variant, tol, early_stop, batched_semantics, max_iter,
                                (const double2*)b, (double2*)x, nbatch, row_iters_dev, stream, diag ? nullptr : diag_scale, b_times_ws, zero_x0,
                                nullptr, hermitian, op->h48.vhat ? &op->h48 : nullptr, (const double2*)x0);
}

// Enqueues the cooperative solve of `nbatch` systems on a 2-D 128^2..512^2 grid (cg_coop2d_kernel).  Few systems: G = 32-64
// workgroups per system (latency); many systems (variance / trace probes): as few workgroups per system as the registers
// allow, G = 1 when the mode block has <= 2048 entries -- no grid barrier, one system per CU (throughput).  Iteration counts
// go to d_iters (device, nbatch ints; -3 where a grid barrier died), *d_status (device int) is non-zero when one did.
// EFGP_EUNSUPPORTED when no launch shape fits.
struct CoopInfo {
    int* d_status = nullptr;
    int G = 0, rows_wg = 0, lines = 0, cols_wg = 0, per = 0;
    double* stamps = nullptr;
    int dbg = 0;
    bool herm = false;
};
static int coop_enqueue(efgp_toeplitz_s* op, const void* ws, double sigmasq, int variant, const double* precond_diag, const void* b,
                        void* x, int nbatch, double tol, int max_iter, int early_stop, int batched_semantics, int* d_iters,
                        hipStream_t stream, CoopInfo* info, int nan_on_dead, int hermitian = 0, const double* diag_scale = nullptr,
                        int b_times_ws = 0, int zero_x0 = 0) {
    DeviceCtx* ctx = op->ctx;
    const bool small = op->coop_small && std::getenv("EFGP_NO_COOP_SMALL") == nullptr;
    const ToepGeom g = small ? op->g_co : op->g;
    const int F0 = (int)g.F[0], F1 = (int)g.F[1], n0 = (int)g.n[0], n1 = (int)g.n[1];
    // Hermitian systems (the caller's promise, checked by the kernel): rows k0 >= 0 only, column pairs (cg_coop2d_herm_kernel)
    const bool herm = hermitian && (n0 & 1) && (n1 & 1) && n0 >= 3 && std::getenv("EFGP_NO_CG_COOP_HERM") == nullptr;
    const int nrow = herm ? (n0 + 1) / 2 : n0;            // rows of the mode block the workgroups share out
    const int ncol = herm ? F1 / 2 : F1;                  // column lines (pairs) they share out
    // workgroups per system: as many as the latency shape uses (16 / 32 / 64) while the whole batch stays resident (one
    // workgroup per CU), never fewer than the registers need (8 vector entries per thread)
    int G_lat = F1 / 8;      // 16 / 32 / 64 (measured at 128^2: 19.9 us per iteration with 16 workgroups, 22.7 with 32, 20.6 with 8)
    if (herm && F1 <= 256) G_lat = F1 / 16;   // half the work per system: 8 / 16 workgroups measured best at 128^2 / 256^2, 64 at 512^2
    if (const char* ed = std::getenv("EFGP_COOP_GDIV")) G_lat = std::max(1, F1 / std::max(1, std::atoi(ed)));  // experiments: G = F1 / div
    if (const char* eg = std::getenv("EFGP_COOP_G")) G_lat = std::max(1, std::min(G_lat, std::atoi(eg)));   // experiments
    // the workgroup counts a grid offers: G_lat halved while it stays whole (16 8 4 2 1; 12 6 3 1 on the 48 R grids)
    auto halve = [](int Gv) { return Gv > 1 ? ((Gv & 1) ? 1 : Gv / 2) : 1; };
    int G_min = G_lat;       // the smallest count whose rows still fit a workgroup's registers (8 vector entries per thread)
    while (G_min > 1 && ((nrow + halve(G_min) - 1) / halve(G_min)) * n1 <= 8 * kLineThreads) G_min = halve(G_min);
    if (const char* eg = std::getenv("EFGP_COOP_GMIN")) {                                                       // experiments
        int Gv = G_lat;
        while (Gv > G_min && halve(Gv) >= std::atoi(eg)) Gv = halve(Gv);
        G_min = std::max(G_min, Gv);
    }
    int G = G_lat;
    while (G > G_min && (int64_t)G * nbatch > ctx->num_cu) G = halve(G);
    const int ks = ((nrow + G - 1) / G) * n1 <= 4 * kLineThreads ? 4 : 8;
    // columns per LDS pass: as many as the workgroup owns, the per-thread load registers (16) and the LDS allow -- a pass of
    // 8 columns leaves one work item per thread and stage (latency bound: 134 us per iteration of a 128^2 system on one CU
    // with 8, 4 items with 32).  Hermitian: a line is a column PAIR and a thread loads two values per (k0, line) slot.
    const int load_cap = herm ? (kCoopLoads / 2) * kLineThreads / nrow : kCoopLoads * kLineThreads / F0;
    int lpbc = 1;
    while (lpbc * 2 <= std::min(ncol / G, load_cap)) lpbc <<= 1;
    // LDS of a column pass: two images of lpbc lines (+ the Hermitian kernel's slice of the spectrum).  (Until late in round 4 the
    // bound was four images: half the columns per pass -- 96^2, one system per workgroup: 50 -> 42 us per iteration; 384^2
    // general: 30.7 -> 26.9.)
    const size_t lds_factor = std::getenv("EFGP_COOP_LDSF") ? (size_t)std::atoi(std::getenv("EFGP_COOP_LDSF")) : (herm ? 3 : 2);
    while (lpbc > 4 && (lds_factor * lpbc * (F0 + 1) + (size_t)F0 + (size_t)F1) * sizeof(double2) + 2048 > (size_t)ctx->max_lds) lpbc >>= 1;
    while (lpbc > 1 && (ncol / G) % lpbc) lpbc >>= 1;           // a pass count per workgroup must be whole (48 R grids: 3 * 2^k lines)
    bool shape_ok = G <= kCoopMaxG && ((nrow + G - 1) / G) * n1 <= ks * kLineThreads && ncol % (G * lpbc) == 0;
    const int rows_wg = (nrow + G - 1) / G, cols_wg = ncol / G;
    int lines = std::min(rows_wg, kCoopLoads * kLineThreads / F1);
    auto lds_for = [&](int ln) {
        const size_t bufsz = (size_t)std::max(ln * (F1 + 1), lpbc * (F0 + 1));
        return (2 * bufsz + (size_t)F1 + (F0 == F1 ? 0 : (size_t)F0)) * sizeof(double2);
    };
    while (lines > 1 && lds_for(lines) + 2048 > (size_t)ctx->max_lds) --lines;
    size_t lds = lds_for(lines);
    // Hermitian, one column pass per workgroup: its slice of the spectrum stays in LDS (16-32 KB)
    const size_t spec_bytes = (size_t)F0 * lpbc * sizeof(double2);
    const bool spec_lds = herm && cols_wg == lpbc && lds + spec_bytes + 2048 <= (size_t)ctx->max_lds &&
                          std::getenv("EFGP_NO_COOP_SPEC_LDS") == nullptr;
    if (spec_lds) lds += spec_bytes;
    shape_ok = shape_ok && lds + 2048 <= (size_t)ctx->max_lds && G <= ctx->num_cu &&
               (herm ? lpbc * nrow <= (kCoopLoads / 2) * kLineThreads : lpbc * F0 <= kCoopLoads * kLineThreads);
    if (!shape_ok) return EFGP_EUNSUPPORTED;
    const int cap = std::max(1, ctx->num_cu / G);                      // systems resident at once (one workgroup per CU)
    const int per = std::min(cap, nbatch);
    const size_t grid_elems = (size_t)per * (size_t)n0 * (size_t)F1;
    double2* pad = (double2*)scratch(ctx, SLOT_TOEP_PAD, 3 * grid_elems * sizeof(double2));
    // partial sums | arrival counters (64 B apart) | status
    const size_t off_bar = (size_t)per * 3 * kCoopMaxG * sizeof(double);
    const size_t off_status = off_bar + (size_t)per * 64;
    char* scb = (char*)scratch(ctx, SLOT_MISC, off_status + 64 + 128);
    if (!pad || !scb) return EFGP_ENOMEM;
    CoopArgs ca;
    ca.g = g;
    ca.ws = (const double2*)ws;
    ca.diag = precond_diag;
    ca.diag_scale = precond_diag ? nullptr : diag_scale;
    ca.b_times_ws = b_times_ws;
    ca.zero_x0 = zero_x0;
    ca.sigmasq = sigmasq;
    ca.variant = variant;
    ca.tol = tol;
    ca.early_stop = early_stop;
    ca.batched = batched_semantics;
    ca.max_iter = max_iter;
    if (!small && !ensure_centred_spectrum(op, stream)) return EFGP_EUNSUPPORTED;
    ca.vhat = small ? op->vhat_co : op->vhat_c;
    ca.tw0 = small ? op->tw_co[0] : op->tw[0];
    ca.tw1 = small ? op->tw_co[1] : op->tw[1];
    ca.b1 = pad;
    ca.b2 = pad + grid_elems;
    ca.b3 = pad + 2 * grid_elems;
    ca.spec_lds = spec_lds ? 1 : 0;
    ca.partial = (double*)scb;
    ca.bar = (unsigned*)(scb + off_bar);
    ca.status = (int*)(scb + off_status);
    ca.hist_cap = cg_history().capacity;
    ca.nan_on_dead = nan_on_dead;
    ca.G = G;
    // test hook (tests/test_gpu_variance_ops.py): every grid barrier of this launch dies at once -- one arrival more than there
    // are workgroups is awaited and the status word is preset, so the first status check (256 polls) ends the wait
    const bool test_dead = G > 1 && std::getenv("EFGP_COOP_TEST_DEAD") != nullptr;
    ca.bar_need = test_dead ? G + 1 : G;
    ca.rows_wg = rows_wg;
    ca.cols_wg = cols_wg;
    ca.lines = lines;
    ca.lpbc = lpbc;
    ca.dbg = std::getenv("EFGP_COOP_DBG") ? std::atoi(std::getenv("EFGP_COOP_DBG")) : 0;
    ca.stamps = (double*)(scb + off_status + 64);
    if (ca.dbg == 2) EFGP_HIP_CHECK(hipMemsetAsync(ca.stamps, 0, 128, stream));
    auto launch = [&](auto kern, int nsys) -> hipError_t {
        // attribute and occupancy query once per (device, kernel, LDS size): both are host round trips into the runtime.  The
        // attribute is a per-kernel maximum: it is only ever raised.
        static std::mutex mu;
        static std::map<std::pair<int, const void*>, size_t> attr_set;
        static std::map<std::tuple<int, const void*, size_t>, bool> fits;
        bool raise_attr, known;
        {
            std::lock_guard<std::mutex> lk(mu);
            auto it = attr_set.find(std::make_pair(op->device, (const void*)kern));
            raise_attr = it == attr_set.end() || it->second < lds;
            known = fits.count(std::make_tuple(op->device, (const void*)kern, lds)) != 0;
        }
        if (raise_attr) {
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            std::lock_guard<std::mutex> lk(mu);
            attr_set[std::make_pair(op->device, (const void*)kern)] = lds;
        }
        if (!known) {
            // the hand-rolled grid barrier needs every workgroup of a launch resident: at most one workgroup per CU is asked for
            // (G * nsys <= num_cu above), so it is enough that ONE fits a CU with these registers and this much LDS
            int fit = 0;
            hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, kern, kLineThreads, lds);
            if (e != hipSuccess) return e;
            if (fit < 1) return hipErrorLaunchOutOfResources;
            std::lock_guard<std::mutex> lk(mu);
            fits[std::make_tuple(op->device, (const void*)kern, lds)] = true;
        }
        hipLaunchKernelGGL(kern, dim3(G, nsys), dim3(kLineThreads), lds, stream, ca);
        return hipGetLastError();
    };
    {
        KernelTimer timer("cg_coop", stream);
        for (int s0 = 0; s0 < nbatch; s0 += per) {
            const int nsys = std::min(per, nbatch - s0);
            ca.b = (const double2*)b + (int64_t)s0 * g.M;
            ca.x = (double2*)x + (int64_t)s0 * g.M;
            ca.iters = d_iters + s0;
            ca.hist = s0 == 0 ? cg_history().buf : nullptr;
            // the status word is cleared before EVERY launch: a dead barrier in one slab of systems must not make the later
            // slabs give up at their first poll (the per-system iteration counts carry the -3 of the slab that died).  It sits
            // right behind the arrival counters: one fill for both
            if (!test_dead) {
                EFGP_HIP_CHECK(hipMemsetAsync(scb + off_bar, 0, (size_t)per * 64 + 64, stream));
            } else {
                EFGP_HIP_CHECK(hipMemsetAsync(scb + off_bar, 0, (size_t)per * 64, stream));
                EFGP_HIP_CHECK(hipMemsetAsync(scb + off_status, 1, 64, stream));
            }
            hipError_t e;
            if (herm) {
                if (G == 1 && ks == 8) e = launch(cg_coop2d_herm_kernel<8, true>, nsys);
                else if (G == 1) e = launch(cg_coop2d_herm_kernel<4, true>, nsys);
                else if (ks == 8) e = launch(cg_coop2d_herm_kernel<8, false>, nsys);
                else e = launch(cg_coop2d_herm_kernel<4, false>, nsys);
            } else if (G == 1 && ks == 8) e = launch(cg_coop2d_kernel<8, true>, nsys);
            else if (G == 1) e = launch(cg_coop2d_kernel<4, true>, nsys);
            else if (ks == 8) e = launch(cg_coop2d_kernel<8, false>, nsys);
            else e = launch(cg_coop2d_kernel<4, false>, nsys);
            EFGP_HIP_CHECK(e);
        }
    }
    if (info) {
        info->d_status = ca.status;
        info->G = G;
        info->rows_wg = rows_wg;
        info->lines = lines;
        info->cols_wg = cols_wg;
        info->per = per;
        info->stamps = ca.stamps;
        info->herm = herm;
        info->dbg = ca.dbg;
    }
    if (herm && ca.dbg == 2) {          // diagnostic run (EFGP_COOP_DBG=2): wait and print the phase shares of workgroup 0, system 0
        double hs[12];
        int it0 = 0;
        EFGP_HIP_CHECK(stream_wait(stream));
        EFGP_HIP_CHECK(hipMemcpy(hs, ca.stamps, sizeof(hs), hipMemcpyDeviceToHost));
        EFGP_HIP_CHECK(hipMemcpy(&it0, d_iters, sizeof(int), hipMemcpyDeviceToHost));
        const char* nm[10] = {"partial <r,r>, <r,z>", "R: rows of ws z -> b1", "barrier 1 + all-reduce", "beta, p update", "C: loads b1/b3, assemble",
                              "C: transform, spectrum, transform", "C: unpack, stores to b2", "barrier 2 + all-reduce", "Ri: loads of b2", "Ri: transform, A p, x r z"};
        double tot = 0;
        for (int q = 0; q < 10; ++q) tot += hs[q];
        std::fprintf(stderr, "[coop-herm] G = %d, rows/wg %d, lines/pass %d, column pairs/wg %d, systems/launch %d, %d iterations\n", G, rows_wg, lines, cols_wg, per, it0);
        for (int q = 0; q < 10; ++q) std::fprintf(stderr, "[coop-herm] %-36s %9.0f ticks/iter %5.1f%%\n", nm[q], hs[q] / std::max(1, it0), 100.0 * hs[q] / tot);
    }
    return EFGP_OK;
}

static thread_local bool t_no_coop = false;    // set while the cooperative solve hands systems to the multi-launch path

static int cg_solve_impl(efgp_toeplitz_t* op, const void* ws, double sigmasq, int variant, const double* precond_diag,
                         const void* b, void* x, int nbatch, double tol, int max_iter, int early_stop,
                         int batched_semantics, int* iters_out, int* row_iters_out, void* stream_, int hermitian);

int efgp_cg_solve(efgp_toeplitz_t* op, const void* ws, double sigmasq, int variant, const double* precond_diag,
                  const void* b, void* x, int nbatch, double tol, int max_iter, int early_stop,
                  int batched_semantics, int* iters_out, int* row_iters_out, void* stream_) {
    return cg_solve_impl(op, ws, sigmasq, variant, precond_diag, b, x, nbatch, tol, max_iter, early_stop, batched_semantics, iters_out,
                         row_iters_out, stream_, 0);
}

int efgp_cg_solve_hermitian(efgp_toeplitz_t* op, const void* ws, double sigmasq, int variant, const double* precond_diag,
                            const void* b, void* x, int nbatch, double tol, int max_iter, int early_stop,
                            int batched_semantics, int* iters_out, int* row_iters_out, void* stream_) {
    return cg_solve_impl(op, ws, sigmasq, variant, precond_diag, b, x, nbatch, tol, max_iter, early_stop, batched_semantics, iters_out,
                         row_iters_out, stream_, 1);
}

static int cg_solve_impl(efgp_toeplitz_t* op, const void* ws, double sigmasq, int variant, const double* precond_diag,
                         const void* b, void* x, int nbatch, double tol, int max_iter, int early_stop,
                         int batched_semantics, int* iters_out, int* row_iters_out, void* stream_, int hermitian) {
    EFGP_REQUIRE(op && ws && b && x, "efgp_cg_solve: null argument");
    EFGP_REQUIRE(nbatch >= 1, "efgp_cg_solve: nbatch must be >= 1");
    EFGP_REQUIRE(variant == 0 || variant == 1, "efgp_cg_solve: variant must be 0 or 1");
    EFGP_REQUIRE(batched_semantics || nbatch == 1, "efgp_cg_solve: single-system semantics need nbatch == 1");
    EFGP_REQUIRE(sigmasq > 0.0 || variant == 0, "efgp_cg_solve: sigmasq must be positive for A_var");
    hipStream_t stream = (hipStream_t)stream_;
    DeviceGuard guard(op->device, (hipStream_t)stream_);
    DeviceCtx* ctx = op->ctx;
    const ToepGeom g = op->g;
    if (max_iter <= 0) max_iter = (int)std::min<int64_t>(2 * g.M, 2000000000);

    // small circulant grids: the whole solve in one persistent kernel, one workgroup per system
    const bool no_persistent = std::getenv("EFGP_NO_PERSISTENT_CG") != nullptr;   // test hook
    if (op->persistent_ok && !no_persistent) {
        int* d_iters = (int*)scratch(ctx, SLOT_CG_SCALARS, (size_t)nbatch * sizeof(int) + 64);
        int* host = pinned_host(ctx, (size_t)nbatch * sizeof(int) + 64);
        if (!d_iters || !host) return EFGP_ENOMEM;
        int rc;
        {
            KernelTimer timer("cg_persistent", stream);
            const ToepGeom* gq;
            const double2* const* twq;
            const double2* vq;
            cg_operands(op, &gq, &twq, &vq);
            rc = persistent_cg_launch(*gq, twq, vq, (const double2*)ws, precond_diag, sigmasq,
                                      variant, tol, early_stop, batched_semantics, max_iter, (const double2*)b, (double2*)x,
                                      nbatch, d_iters, stream);
        }
        if (rc != EFGP_OK) return rc;
        EFGP_HIP_CHECK(hipMemcpyAsync(host, d_iters, (size_t)nbatch * sizeof(int), hipMemcpyDeviceToHost, stream));
        EFGP_HIP_CHECK(stream_wait(stream));
        int mx = 0;
        for (int i = 0; i < nbatch; ++i) {
            mx = std::max(mx, host[i]);
            if (row_iters_out) row_iters_out[i] = host[i];
        }
        int total = mx;
        if (batched_semantics && mx < max_iter) total = mx + 1;      // the terminating pass, cg.py:193-199,243
        if (iters_out) *iters_out = total;
        return EFGP_OK;
    }

    // 2-D grids of 128..512 per dimension: the whole solve in cooperative launches (coop_enqueue)
    if (op->lines_ok && !t_no_coop && std::getenv("EFGP_NO_CG_COOP") == nullptr && std::getenv("EFGP_NO_CG_LINES") == nullptr) {
        int* d_iters = (int*)scratch(ctx, SLOT_CG_SCALARS, (size_t)nbatch * sizeof(int) + 64);
        int* host = pinned_host(ctx, (size_t)nbatch * sizeof(int) + 128);
        if (!d_iters || !host) return EFGP_ENOMEM;
        CoopInfo ci;
        const int rcq = coop_enqueue(op, ws, sigmasq, variant, precond_diag, b, x, nbatch, tol, max_iter, early_stop, batched_semantics,
                                     d_iters, stream, &ci, /*nan_on_dead: this entry re-solves dead systems from x0*/ 0);
        if (rcq != EFGP_OK && rcq != EFGP_EUNSUPPORTED) return rcq;
        if (rcq == EFGP_OK) {
            EFGP_HIP_CHECK(hipMemcpyAsync(host, ci.d_status, sizeof(int), hipMemcpyDeviceToHost, stream));
            EFGP_HIP_CHECK(hipMemcpyAsync(host + 16, d_iters, (size_t)nbatch * sizeof(int), hipMemcpyDeviceToHost, stream));
            EFGP_HIP_CHECK(stream_wait(stream));
            const int dead = host[0];
            const int* hit = host + 16;
            if (ci.dbg == 2) {
                double hs[14];
                EFGP_HIP_CHECK(hipMemcpy(hs, ci.stamps, sizeof(hs), hipMemcpyDeviceToHost));
                const char* nm[14] = {"R store to b1", "barrier 1", "C store", "barrier 2", "Ri load+fft", "pAp sum (incl. barrier)", "update", "rr/rz sum (incl. barrier)",
                                      "C load", "C transform 1 (+ multiply)", "-", "C transform 2", "R zero fill, ws u", "R transform"};
                double tot = 0;
                for (int q = 0; q < 14; ++q) tot += hs[q];
                std::fprintf(stderr, "[coop] G = %d, rows/wg %d, lines/pass %d, columns/wg %d, systems/launch %d\n", ci.G, ci.rows_wg, ci.lines, ci.cols_wg, ci.per);
                for (int q = 0; q < 14; ++q) std::fprintf(stderr, "[coop] %-28s %9.0f cycles/iter %5.1f%%\n", nm[q], hs[q] / std::max(1, hit[0]), 100.0 * hs[q] / tot);
            }
            bool any_dead = dead != 0;
            for (int i = 0; i < nbatch; ++i) any_dead = any_dead || hit[i] < 0;
            if (!any_dead) {
                int mx = 0;
                for (int i = 0; i < nbatch; ++i) {
                    mx = std::max(mx, hit[i]);
                    if (row_iters_out) row_iters_out[i] = hit[i];
                }
                if (iters_out) *iters_out = (batched_semantics && mx < max_iter) ? mx + 1 : mx;
                return EFGP_OK;
            }
            // a grid barrier ran out of polls (the workgroups were not co-resident): the systems it hit still hold x0 in x (a
            // solution is written only by a system that finished) and go through the multi-launch path one by one
            std::fprintf(stderr, "[efgp_hip] cooperative CG: grid barrier timed out, falling back to the multi-launch iteration\n");
            int mx = 0;
            std::vector<int> its(hit, hit + nbatch);
            for (int i = 0; i < nbatch; ++i) {
                if (its[i] < 0) {
                    int one = 0;
                    t_no_coop = true;
                    const int rc1 = efgp_cg_solve(op, ws, sigmasq, variant, precond_diag, (const double2*)b + (int64_t)i * g.M,
                                                  (double2*)x + (int64_t)i * g.M, 1, tol, max_iter, early_stop, batched_semantics, nullptr, &one,
                                                  stream_);
                    t_no_coop = false;
                    if (rc1 != EFGP_OK) return rc1;
                    its[i] = one;
                }
                mx = std::max(mx, its[i]);
                if (row_iters_out) row_iters_out[i] = its[i];
            }
            if (iters_out) *iters_out = (batched_semantics && mx < max_iter) ? mx + 1 : mx;
            return EFGP_OK;
        }
    }

    // The iteration bursts below are replayed as hipGraphs, which cannot be captured on the legacy default stream:
    // run the multi-kernel solve on a side stream ordered after the caller's stream (the final poll of every group
    // synchronises the host with it, so later work on the caller's stream is ordered after the solve).
    if (!timing_enabled() && std::getenv("EFGP_NO_CG_GRAPH") == nullptr) {
        if (!ctx->aux_stream) {
            if (hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&ctx->aux_event, hipEventDisableTiming) != hipSuccess) {
                (void)hipGetLastError();
                ctx->aux_stream = nullptr;
            }
        }
        if (ctx->aux_stream) {
            EFGP_HIP_CHECK(hipEventRecord(ctx->aux_event, stream));
            EFGP_HIP_CHECK(hipStreamWaitEvent(ctx->aux_stream, ctx->aux_event, 0));
            stream = ctx->aux_stream;
        }
    }
    // rows are processed in groups whose padded grids fit ~512 MB of scratch
    const int64_t group_cap = std::max<int64_t>(1, (int64_t)(512ll << 20) / (g.Ftot * (int64_t)sizeof(double2)));
    int global_iters = 0;
    for (int64_t r0 = 0; r0 < nbatch; r0 += group_cap) {
        const int rows = (int)std::min<int64_t>(group_cap, nbatch - r0);
        double2* pad = (double2*)scratch(ctx, SLOT_TOEP_PAD, (size_t)rows * (size_t)g.Ftot * sizeof(double2));
        double2* vec = (double2*)scratch(ctx, SLOT_CG_VEC, (size_t)3 * rows * (size_t)g.M * sizeof(double2));
        // scalars | status (64 B) + slot->row map | arrival counters | per-workgroup partial sums
        const size_t off_status = ((size_t)rows * sizeof(CgRowScalars) + 63) & ~size_t(63);
        const size_t off_counter = off_status + 64 + (((size_t)rows * sizeof(int) + 63) & ~size_t(63));
        const size_t off_partial = off_counter + (((size_t)2 * rows * sizeof(int) + 63) & ~size_t(63));
        const size_t sc_bytes = off_partial + (size_t)rows * 3 * kCgBlocksMax * sizeof(double);
        char* scb = (char*)scratch(ctx, SLOT_CG_SCALARS, sc_bytes);
        int* host = pinned_host(ctx, (size_t)rows * sizeof(CgRowScalars) + 64);
        if (!pad || !vec || !scb || !host) return EFGP_ENOMEM;
        CgArgs a;
        a.g = g;
        a.ws = (const double2*)ws;
        a.diag = precond_diag;
        a.sigmasq = sigmasq;
        a.variant = variant;
        a.tol = tol;
        a.early_stop = early_stop;
        a.batched = batched_semantics;
        a.b = (const double2*)b + r0 * g.M;
        a.x = (double2*)x + r0 * g.M;
        a.r = vec;
        a.p = vec + (int64_t)rows * g.M;
        a.ap = vec + (int64_t)2 * rows * g.M;
        a.pad = pad;
        a.pad_out = pad;
        a.rows = nullptr;
        a.hist = r0 == 0 ? cg_history().buf : nullptr;
        a.hist_cap = cg_history().capacity;
        a.sc = (CgRowScalars*)scb;
        a.status = (int*)(scb + off_status);
        int* d_rows = a.status + 16;
        a.counter = (int*)(scb + off_counter);
        a.partial = (double*)(scb + off_partial);
        a.v_off = 0;
        a.v_len = g.M;
        a.v_w1 = g.M;
        a.nblk = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(64 /* update kernels: measured optimum */, (g.M + kVecThreads - 1) / kVecThreads),
                                                            std::max<int64_t>(1, 1024 / rows)));
        EFGP_HIP_CHECK(hipMemsetAsync(a.status, 0, off_partial - off_status, stream));      // status, map, counters

        // r0 = b - A x0
        hipLaunchKernelGGL(pad_scale_kernel, grid_for(g.Ftot, rows, kVecThreads), dim3(kVecThreads), 0, stream, g,
                           (const double2*)a.x, g.M, a.ws, (const int*)nullptr, (const int*)nullptr, pad, 1.0);
        EFGP_HIP_CHECK(hipGetLastError());
        int rc = circulant(op, pad, rows, stream);
        if (rc != EFGP_OK) return rc;
        hipLaunchKernelGGL(cg_init_kernel, dim3(a.nblk, rows), dim3(kVecThreads), 0, stream, a);
        EFGP_HIP_CHECK(hipGetLastError());

        // iteration loop; active rows are compacted on the host whenever the status is polled
        std::vector<int> active_rows(rows);
        for (int i = 0; i < rows; ++i) active_rows[i] = i;
        int n_active = rows;
        bool compacted = false;
        int it = 0;
        int last_active_it = 0;       // number of iterations in which at least one row was active
        const int poll_every = 8;
        std::vector<CgRowScalars> hsc(rows);
        // 2-D mid-size grids: three launches of pruned in-LDS line transforms instead of pad + rocFFT + multiply
        const bool use_lines = op->lines_ok && std::getenv("EFGP_NO_CG_LINES") == nullptr;
        LineArgs la;
        size_t lds_rows = 0, lds_cols = 0;
        if (use_lines) {
            rc = ensure_reference_spectrum(op, stream);
            if (rc != EFGP_OK) return rc;
            la.vhat = op->vhat;
            la.tw0 = op->tw[0];
            la.tw1 = op->tw[1];
            la.b1 = pad;                                                     // [rows][n0][F1] fits: Ftot >= 2 n0 F1
            la.b2 = pad + (int64_t)rows * g.n[0] * g.F[1];
            // every workgroup of the reducing kernels pays a device-scope fence (an L2 write-back, ~2 us each, serialised
            // per XCD): few systems -> many small workgroups for parallelism, many systems -> few large ones
            const int64_t Fmax = std::max(g.F[0], g.F[1]);
            int lpb = 4;
            if (rows > 8) {
                const int64_t fit = ((int64_t)ctx->max_lds / (int64_t)sizeof(double2) - Fmax) / (2 * (Fmax + 1));
                while (lpb * 2 <= fit && lpb < 32) lpb <<= 1;
            }
            la.lpb = lpb;
            la.nblk_rows = (int)((g.n[0] + lpb - 1) / lpb);
            lds_rows = ((size_t)2 * lpb * (size_t)(g.F[1] + 1) + (size_t)g.F[1]) * sizeof(double2);   // + twiddles
            lds_cols = ((size_t)2 * lpb * (size_t)(g.F[0] + 1) + (size_t)g.F[0]) * sizeof(double2);
            EFGP_HIP_CHECK(hipFuncSetAttribute((const void*)cg_rows_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_rows));
            EFGP_HIP_CHECK(hipFuncSetAttribute((const void*)cg_rows_inv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_rows));
            EFGP_HIP_CHECK(hipFuncSetAttribute((const void*)cg_cols_mid_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cols));
        }
        bool use_lines3 = op->lines3_ok && std::getenv("EFGP_NO_CG_LINES") == nullptr;
        if (use_lines3) {
            // the line kernels keep 2 x lpb lines of F2 + 1 elements in LDS and one partial sum per workgroup: both limits must
            // hold (found at full size: mtot = 57, F = 128 asked for 64 lines per block = 266 KB of LDS)
            const int64_t fit = ((int64_t)ctx->max_lds / (int64_t)sizeof(double2) - g.F[2]) / (2 * (g.F[2] + 1));
            int64_t lmax = 16;
            while (lmax * 2 <= fit) lmax <<= 1;
            if (fit < 16 || (g.n[0] * g.n[1] + lmax - 1) / lmax > kCgBlocksMax) use_lines3 = false;
        }
        Line3Args l3;
        size_t lds3_c = 0, lds3_s[3] = {0, 0, 0};
        if (use_lines3) {
            l3.vhat = op->vhat;
            for (int q = 0; q < 3; ++q) l3.tw[q] = op->tw[q];
            l3.b1 = pad;
            l3.b2 = pad + (int64_t)rows * g.n[0] * g.n[1] * g.F[2];
            const int64_t nlines = g.n[0] * g.n[1];
            const int64_t fit = ((int64_t)ctx->max_lds / (int64_t)sizeof(double2) - g.F[2]) / (2 * (g.F[2] + 1));
            int lpb = 16;
            while ((nlines + lpb - 1) / lpb > kCgBlocksMax && lpb * 2 <= fit) lpb <<= 1;
            if (rows > 8) {                               // batched: fewer, larger workgroups (fewer reduction fences)
                while (lpb * 2 <= fit && lpb < 64) lpb <<= 1;
            }
            l3.lpb_c = lpb;
            l3.lpb_s = 16;
            l3.nblk_lines = (int)((nlines + lpb - 1) / lpb);
            lds3_c = ((size_t)2 * lpb * (size_t)(g.F[2] + 1) + (size_t)g.F[2]) * sizeof(double2);
            for (int q = 0; q < 2; ++q) lds3_s[q] = ((size_t)2 * l3.lpb_s * (size_t)(g.F[q] + 1) + (size_t)g.F[q]) * sizeof(double2);
            EFGP_HIP_CHECK(hipFuncSetAttribute((const void*)cg3_fwd2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3_c));
            EFGP_HIP_CHECK(hipFuncSetAttribute((const void*)cg3_inv2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3_c));
            EFGP_HIP_CHECK(hipFuncSetAttribute((const void*)cg3_dim1_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3_s[1]));
            EFGP_HIP_CHECK(hipFuncSetAttribute((const void*)cg3_dim1_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3_s[1]));
            EFGP_HIP_CHECK(hipFuncSetAttribute((const void*)cg3_mid0_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3_s[0]));
        }
        // Hermitian 3-D systems: the planes k0 >= 0 only (cg3h_* kernels).  The data are checked once per group of rows.
        bool use_lines3h = use_lines3 && hermitian && (g.n[0] & 1) && (g.n[1] & 1) && (g.n[2] & 1) && g.n[0] >= 3 &&
                           std::getenv("EFGP_NO_CG_HERM3") == nullptr;
        Line3HArgs l3h;
        size_t lds3h_c = 0, lds3h_s[2] = {0, 0};
        if (use_lines3h) {
            if (!op->vc3) {
                op->vc3 = (double*)pool_alloc(ctx, (size_t)g.Ftot * sizeof(double));
                if (!op->vc3) return EFGP_ENOMEM;
                hipLaunchKernelGGL(center_spectrum3_real_kernel, dim3((unsigned)((g.Ftot + 255) / 256)), dim3(256), 0, stream, op->vhat, op->tw[0],
                                   op->tw[1], op->tw[2], (int)g.n[0], (int)g.n[1], (int)g.n[2], (int)g.F[0], (int)g.F[1], (int)g.F[2], op->vc3);
                EFGP_HIP_CHECK(hipGetLastError());
            }
            double* chk_buf = (double*)scratch(ctx, SLOT_MISC, 64);
            if (!chk_buf) return EFGP_ENOMEM;
            EFGP_HIP_CHECK(hipMemsetAsync(chk_buf, 0, 64, stream));
            hipLaunchKernelGGL(cg_herm_check_kernel, dim3(256), dim3(kVecThreads), 0, stream, a.b, (const double2*)a.x, a.ws, g.M, rows, chk_buf);
            EFGP_HIP_CHECK(hipGetLastError());
            EFGP_HIP_CHECK(hipMemcpyAsync(host, chk_buf, 3 * sizeof(double), hipMemcpyDeviceToHost, stream));
            EFGP_HIP_CHECK(stream_wait(stream));
            double hv[3];
            std::memcpy(hv, host, sizeof(hv));
            if (!(hv[0] <= 1e-16 * hv[1]) || hv[2] > 0.0) {
                set_error("efgp_cg_solve_hermitian: right-hand side / start vector not conjugate-even or ws not real and even");
                return EFGP_EINVAL;
            }
            const int64_t nh = (g.n[0] + 1) / 2;
            a.v_off = (nh - 1) * g.n[1] * g.n[2];
            a.v_len = nh * g.n[1] * g.n[2];
            a.v_w1 = g.n[1] * g.n[2];
            l3h.vc = op->vc3;
            for (int q = 0; q < 3; ++q) l3h.tw[q] = op->tw[q];
            l3h.b1 = pad;
            l3h.b2 = pad + (int64_t)rows * nh * g.n[1] * g.F[2];
            const int64_t nlines = nh * g.n[1];
            const int64_t fit = ((int64_t)ctx->max_lds / (int64_t)sizeof(double2) - g.F[2]) / (2 * (g.F[2] + 1));
            int lpb = 8;
            while ((nlines + lpb - 1) / lpb > kCgBlocksMax && lpb * 2 <= fit) lpb <<= 1;
            if (rows >= 3 && lpb < 16 && 32 <= fit) lpb = 16;          // measured at mtot 57: 125 vs 133 us per iteration of 3 systems
            if (rows > 8) {
                while (lpb * 2 <= fit && lpb < 64) lpb <<= 1;
            }
            auto knob = [](const char* name, int dflt) {
                const char* e = std::getenv(name);
                int v = e ? std::atoi(e) : dflt;
                int p2 = 1;
                while (p2 * 2 <= v) p2 <<= 1;
                return std::max(1, p2);
            };
            if (std::getenv("EFGP_CG3_LC")) {
                lpb = knob("EFGP_CG3_LC", lpb);
                while ((nlines + lpb - 1) / lpb > kCgBlocksMax) lpb <<= 1;
            }
            l3h.lpb_c = lpb;
            l3h.lpb_s = (int)std::min<int64_t>(knob("EFGP_CG3_LS", 16), g.F[2] / 2);
            l3h.lpb_m = (int)std::min<int64_t>(knob("EFGP_CG3_LM", rows >= 3 ? 32 : 16), g.F[2] / 2);      // 3 systems: 124.6 vs 132.6 us
            l3h.nblk_lines = (int)((nlines + lpb - 1) / lpb);
            lds3h_c = ((size_t)2 * lpb * (size_t)(g.F[2] + 1) + (size_t)g.F[2]) * sizeof(double2);
            lds3h_s[1] = ((size_t)2 * l3h.lpb_s * (size_t)(g.F[1] + 1) + (size_t)g.F[1]) * sizeof(double2);
            lds3h_s[0] = ((size_t)2 * l3h.lpb_m * (size_t)(g.F[0] + 1) + (size_t)g.F[0]) * sizeof(double2);
            EFGP_HIP_CHECK(hipFuncSetAttribute((const void*)cg3h_fwd2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3h_c));
            EFGP_HIP_CHECK(hipFuncSetAttribute((const void*)cg3h_inv2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3h_c));
            EFGP_HIP_CHECK(hipFuncSetAttribute((const void*)cg3h_dim1_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3h_s[1]));
            EFGP_HIP_CHECK(hipFuncSetAttribute((const void*)cg3h_dim1_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3h_s[1]));
            EFGP_HIP_CHECK(hipFuncSetAttribute((const void*)cg3h_mid0_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3h_s[0]));
            a.nblk = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(std::min(kCgBlocksMax, knob("EFGP_CG_NBLK", 128)), (a.v_len + kVecThreads - 1) / kVecThreads),
                                                                std::max<int64_t>(1, 1024 / rows)));
        }
        bool use_graph = !timing_enabled() && std::getenv("EFGP_NO_CG_GRAPH") == nullptr;
        hipGraphExec_t graph_exec = nullptr;
        int graph_slots = -1;
        const int* graph_rows = nullptr;
        while (it < max_iter && n_active > 0) {
            const int burst = std::min(poll_every, max_iter - it);
            // FFT batch = number of slots (bucketed to a power of two to bound the number of plans)
            int slots = n_active;
            if (compacted) {
                int p2 = 1;
                while (p2 < n_active) p2 <<= 1;
                slots = std::min(p2, rows);
            }
            // One iteration = 5 launches of ours + two rocFFT executions (~11 kernels): enqueueing them one by one
            // costs the host ~300 us per iteration (measured, 3-D 64^3), far more than the GPU needs.  A full burst is
            // therefore captured ONCE into a hipGraph per (slots, row map) state and replayed with one launch.
            auto enqueue_iteration = [&]() -> int {
                if (use_lines3h) {
                    l3h.c = a;
                    const unsigned nhp = (unsigned)((g.n[0] + 1) / 2);
                    const unsigned gs = (unsigned)(g.F[2] / l3h.lpb_s), gp = (unsigned)(g.F[2] / 2 / l3h.lpb_m);
                    hipLaunchKernelGGL(cg3h_fwd2_kernel, dim3(l3h.nblk_lines, slots), dim3(kLineThreads), lds3h_c, stream, l3h);
                    hipLaunchKernelGGL((cg3h_dim1_kernel<0>), dim3(nhp * gs, slots), dim3(kLineThreads), lds3h_s[1], stream, l3h);
                    hipLaunchKernelGGL(cg3h_mid0_kernel, dim3((unsigned)g.F[1] * gp, slots), dim3(kLineThreads), lds3h_s[0], stream, l3h);
                    hipLaunchKernelGGL((cg3h_dim1_kernel<1>), dim3(nhp * gs, slots), dim3(kLineThreads), lds3h_s[1], stream, l3h);
                    hipLaunchKernelGGL(cg3h_inv2_kernel, dim3(l3h.nblk_lines, slots), dim3(kLineThreads), lds3h_c, stream, l3h);
                    hipLaunchKernelGGL(cg_axpy_kernel, dim3(a.nblk, slots), dim3(kVecThreads), 0, stream, a);
                    EFGP_HIP_CHECK(hipGetLastError());
                    return EFGP_OK;
                }
                if (use_lines3) {
                    l3.c = a;
                    const unsigned gs = (unsigned)(g.F[2] / l3.lpb_s);
                    hipLaunchKernelGGL(cg3_fwd2_kernel, dim3(l3.nblk_lines, slots), dim3(kLineThreads), lds3_c, stream, l3);
                    hipLaunchKernelGGL((cg3_dim1_kernel<0>), dim3((unsigned)g.n[0] * gs, slots), dim3(kLineThreads), lds3_s[1], stream, l3);
                    hipLaunchKernelGGL(cg3_mid0_kernel, dim3((unsigned)g.F[1] * gs, slots), dim3(kLineThreads), lds3_s[0], stream, l3);
                    hipLaunchKernelGGL((cg3_dim1_kernel<1>), dim3((unsigned)g.n[0] * gs, slots), dim3(kLineThreads), lds3_s[1], stream, l3);
                    hipLaunchKernelGGL(cg3_inv2_kernel, dim3(l3.nblk_lines, slots), dim3(kLineThreads), lds3_c, stream, l3);
                    hipLaunchKernelGGL(cg_axpy_kernel, dim3(a.nblk, slots), dim3(kVecThreads), 0, stream, a);
                    EFGP_HIP_CHECK(hipGetLastError());
                    return EFGP_OK;
                }
                if (use_lines) {
                    la.c = a;
                    hipLaunchKernelGGL(cg_rows_fwd_kernel, dim3(la.nblk_rows, slots), dim3(kLineThreads), lds_rows, stream, la);
                    hipLaunchKernelGGL(cg_cols_mid_kernel, dim3((unsigned)(g.F[1] / la.lpb), slots), dim3(kLineThreads),
                                       lds_cols, stream, la);
                    hipLaunchKernelGGL(cg_rows_inv_kernel, dim3(la.nblk_rows, slots), dim3(kLineThreads), lds_rows, stream, la);
                    hipLaunchKernelGGL(cg_axpy_kernel, dim3(a.nblk, slots), dim3(kVecThreads), 0, stream, a);
                    EFGP_HIP_CHECK(hipGetLastError());
                    return EFGP_OK;
                }
                hipLaunchKernelGGL(cg_pad_kernel, grid_for(g.Ftot, slots, kVecThreads), dim3(kVecThreads), 0, stream, a);
                EFGP_HIP_CHECK(hipGetLastError());
                int rcc = circulant(op, pad, slots, stream);
                if (rcc != EFGP_OK) return rcc;
                hipLaunchKernelGGL(cg_dot_kernel, dim3(a.nblk, slots), dim3(kVecThreads), 0, stream, a);
                hipLaunchKernelGGL(cg_axpy_kernel, dim3(a.nblk, slots), dim3(kVecThreads), 0, stream, a);
                EFGP_HIP_CHECK(hipGetLastError());
                return EFGP_OK;
            };
            bool launched = false;
            if (use_graph && burst == poll_every) {
                if (graph_exec && (graph_slots != slots || graph_rows != a.rows)) {
                    (void)hipGraphExecDestroy(graph_exec);
                    graph_exec = nullptr;
                }
                if (!graph_exec) {
                    TraceSpan span_build("graph build (plan + capture + instantiate)");
                    hipGraph_t graph = nullptr;
                    // make sure the FFT plan exists and is bound to the stream before capturing
                    if (own_fft_supported(g.d, g.F)) {
                        rc = own_fft_prepare(ctx, g.d, g.F, stream);
                    } else {
                        hipfftHandle fh_unused;
                        rc = fft_plan(ctx, g.d, g.F, slots, stream, &fh_unused);
                    }
                    if (rc != EFGP_OK) return rc;
                    bool ok = hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
                    if (ok) {
                        int rcap = EFGP_OK;
                        for (int k = 0; k < burst && rcap == EFGP_OK; ++k) rcap = enqueue_iteration();
                        const hipError_t ee = hipStreamEndCapture(stream, &graph);
                        ok = rcap == EFGP_OK && ee == hipSuccess && graph != nullptr;
                    }
                    if (ok) ok = hipGraphInstantiate(&graph_exec, graph, nullptr, nullptr, 0) == hipSuccess;
                    if (graph) (void)hipGraphDestroy(graph);
                    if (!ok) {
                        (void)hipGetLastError();
                        graph_exec = nullptr;
                        use_graph = false;            // this runtime cannot capture the sequence: enqueue directly
                    } else {
                        graph_slots = slots;
                        graph_rows = a.rows;
                    }
                }
                if (graph_exec) {
                    TraceSpan span_launch("hipGraphLaunch");
                    EFGP_HIP_CHECK(hipGraphLaunch(graph_exec, stream));
                    launched = true;
                }
            }
            if (!launched) {
                for (int k = 0; k < burst; ++k) {
                    KernelTimer timer("cg_iteration", stream);
                    rc = enqueue_iteration();
                    if (rc != EFGP_OK) return rc;
                }
            }
            it += burst;
            {
                TraceSpan span_poll("poll (D2H scalars + stream synchronize)");
                // into the context's PINNED buffer (a pageable destination is staged by the runtime on every call)
                EFGP_HIP_CHECK(hipMemcpyAsync(host, a.sc, (size_t)rows * sizeof(CgRowScalars), hipMemcpyDeviceToHost, stream));
                EFGP_HIP_CHECK(stream_wait(stream));
                std::memcpy(hsc.data(), host, (size_t)rows * sizeof(CgRowScalars));
            }
            int new_active = 0;
            for (int i = 0; i < rows; ++i) {
                if (hsc[i].active) active_rows[new_active++] = i;
                last_active_it = std::max(last_active_it, hsc[i].iters);
            }
            if (new_active != n_active || !compacted) {
                if (new_active > 0 && new_active < rows) {
                    int p2 = 1;
                    while (p2 < new_active) p2 <<= 1;
                    const int nslots = std::min(p2, rows);
                    std::vector<int> map(nslots, -1);
                    for (int i = 0; i < new_active; ++i) map[i] = active_rows[i];
                    std::memcpy(host, map.data(), nslots * sizeof(int));
                    EFGP_HIP_CHECK(hipMemcpyAsync(d_rows, host, nslots * sizeof(int), hipMemcpyHostToDevice, stream));
                    EFGP_HIP_CHECK(stream_wait(stream));
                    a.rows = d_rows;
                    compacted = true;
                }
            }
            n_active = new_active;
        }
        if (graph_exec) {
            TraceSpan span_destroy("hipGraphExecDestroy");
            (void)hipGraphExecDestroy(graph_exec);
        }
        if (use_lines3h) {          // the planes k0 < 0 of the solutions
            CgArgs am = a;
            am.rows = nullptr;
            hipLaunchKernelGGL(cg3h_mirror_kernel, dim3(64, rows), dim3(kVecThreads), 0, stream, am);
            EFGP_HIP_CHECK(hipGetLastError());
            EFGP_HIP_CHECK(stream_wait(stream));
        }
        // iteration counts (cg.py:152 single; cg.py:193-199,243 batched: +1 for the terminating pass)
        int group_iters;
        if (!batched_semantics) {
            group_iters = hsc.empty() ? 0 : hsc[0].iters;
            if (it == 0) {   // max_iter == 0 or nothing ran
                EFGP_HIP_CHECK(stream_wait(stream));
            }
        } else {
            group_iters = last_active_it;
            if (n_active == 0 && last_active_it < max_iter) group_iters = last_active_it + 1;
        }
        if (row_iters_out)
            for (int i = 0; i < rows; ++i) row_iters_out[r0 + i] = hsc[i].iters;
        global_iters = std::max(global_iters, group_iters);
    }
    if (iters_out) *iters_out = global_iters;
    return EFGP_OK;
}

static int cg_solve_async_impl(efgp_toeplitz_t* op, const void* ws, double sigmasq, int variant, const double* precond_diag,
                               const void* b, void* x, int nbatch, double tol, int max_iter, int early_stop,
                               int batched_semantics, int* row_iters_dev, void* stream_, int hermitian, int zero_x0);

int efgp_cg_solve_async(efgp_toeplitz_t* op, const void* ws, double sigmasq, int variant, const double* precond_diag,
                        const void* b, void* x, int nbatch, double tol, int max_iter, int early_stop,
                        int batched_semantics, int* row_iters_dev, void* stream_) {
    return cg_solve_async_impl(op, ws, sigmasq, variant, precond_diag, b, x, nbatch, tol, max_iter, early_stop, batched_semantics,
                               row_iters_dev, stream_, 0, 0);
}

int efgp_cg_solve_hermitian_async(efgp_toeplitz_t* op, const void* ws, double sigmasq, int variant, const double* precond_diag,
                                  const void* b, void* x, int nbatch, double tol, int max_iter, int early_stop,
                                  int batched_semantics, int* row_iters_dev, void* stream_) {
    return cg_solve_async_impl(op, ws, sigmasq, variant, precond_diag, b, x, nbatch, tol, max_iter, early_stop, batched_semantics,
                               row_iters_dev, stream_, 1, 0);
}

int efgp_cg_solve_from_zero_async(efgp_toeplitz_t* op, const void* ws, double sigmasq, int variant, const double* precond_diag,
                                  const void* b, void* x, int nbatch, double tol, int max_iter, int early_stop, int batched_semantics,
                                  int hermitian, int* row_iters_dev, void* stream_) {
    return cg_solve_async_impl(op, ws, sigmasq, variant, precond_diag, b, x, nbatch, tol, max_iter, early_stop, batched_semantics,
                               row_iters_dev, stream_, hermitian ? 1 : 0, 1);
}

static int cg_solve_async_impl(efgp_toeplitz_t* op, const void* ws, double sigmasq, int variant, const double* precond_diag,
                               const void* b, void* x, int nbatch, double tol, int max_iter, int early_stop,
                               int batched_semantics, int* row_iters_dev, void* stream_, int hermitian, int zero_x0) {
    EFGP_REQUIRE(op && ws && b && x && row_iters_dev, "efgp_cg_solve_async: null argument");
    EFGP_REQUIRE(nbatch >= 1, "efgp_cg_solve_async: nbatch must be >= 1");
    EFGP_REQUIRE(variant == 0 || variant == 1, "efgp_cg_solve_async: variant must be 0 or 1");
    EFGP_REQUIRE(batched_semantics || nbatch == 1, "efgp_cg_solve_async: single-system semantics need nbatch == 1");
    EFGP_REQUIRE(sigmasq > 0.0 || variant == 0, "efgp_cg_solve_async: sigmasq must be positive for A_var");
    if (!op->persistent_ok || std::getenv("EFGP_NO_PERSISTENT_CG") != nullptr) {
        // 2-D 128^2..512^2: the cooperative launches need no host either (a row whose grid barrier died -- workgroups not
        // co-resident -- reports -3 iterations and keeps x0; the synchronous entry retries those through the multi-launch path)
        if (op->lines_ok && std::getenv("EFGP_NO_CG_COOP") == nullptr && std::getenv("EFGP_NO_CG_LINES") == nullptr) {
            hipStream_t stream_c = (hipStream_t)stream_;
            DeviceGuard guard_c(op->device, (hipStream_t)stream_);
            if (max_iter <= 0) max_iter = (int)std::min<int64_t>(2 * op->g.M, 2000000000);
            return coop_enqueue(op, ws, sigmasq, variant, precond_diag, b, x, nbatch, tol, max_iter, early_stop, batched_semantics,
                                row_iters_dev, stream_c, nullptr, /*nan_on_dead*/ 1, hermitian, nullptr, 0, zero_x0);
        }
        set_error("efgp_cg_solve_async: grid does not fit the persistent kernel");
        return EFGP_EUNSUPPORTED;
    }
    hipStream_t stream = (hipStream_t)stream_;
    DeviceGuard guard(op->device, (hipStream_t)stream_);
    if (max_iter <= 0) max_iter = (int)std::min<int64_t>(2 * op->g.M, 2000000000);
    KernelTimer timer("cg_persistent", stream);
    const ToepGeom* gq;
    const double2* const* twq;
    const double2* vq;
    cg_operands(op, &gq, &twq, &vq);
    return persistent_cg_launch(*gq, twq, vq, (const double2*)ws, precond_diag, sigmasq,
                                variant, tol, early_stop, batched_semantics, max_iter, (const double2*)b, (double2*)x, nbatch,
                                row_iters_dev, stream, nullptr, 0, zero_x0, nullptr, hermitian, op->h48.vhat ? &op->h48 : nullptr);
}

int efgp_lanczos(efgp_toeplitz_t* op, const void* ws, double sigmasq, int variant, const void* z, int nprobes, int steps,
                 double* alpha_dev, double* beta_dev, double* norm2_dev, int* steps_taken_dev, void* stream_) {
    EFGP_REQUIRE(op && ws && z && alpha_dev && beta_dev && steps_taken_dev, "efgp_lanczos: null argument");
    EFGP_REQUIRE(nprobes >= 1 && steps >= 1, "efgp_lanczos: nprobes and steps must be >= 1");
    EFGP_REQUIRE(variant == 0 || variant == 1, "efgp_lanczos: variant must be 0 or 1");
    EFGP_REQUIRE(sigmasq > 0.0 || variant == 0, "efgp_lanczos: sigmasq must be positive for A_var");
    if (!op->persistent_ok || std::getenv("EFGP_NO_PERSISTENT_CG") != nullptr) {
        set_error("efgp_lanczos: grid does not fit the persistent kernel");
        return EFGP_EUNSUPPORTED;
    }
    hipStream_t stream = (hipStream_t)stream_;
    DeviceGuard guard(op->device, (hipStream_t)stream_);
    const LanczosOut lz{steps, alpha_dev, beta_dev, norm2_dev};
    const ToepGeom* gq;
    const double2* const* twq;
    const double2* vq;
    cg_operands(op, &gq, &twq, &vq);
    return persistent_cg_launch(*gq, twq, vq, (const double2*)ws, nullptr, sigmasq, variant, 0.0, 0,
                                1, steps, (const double2*)z, nullptr, nprobes, steps_taken_dev, stream, nullptr, 0, 1, &lz);
}

int efgp_cg_solve_mean_async(efgp_toeplitz_t* op, const void* ws, double sigmasq, const double* diag_scale_dev, const void* fy,
                             void* x, double tol, int max_iter, int early_stop, int* iters_dev, void* stream_) {
    EFGP_REQUIRE(op && ws && fy && x && iters_dev, "efgp_cg_solve_mean_async: null argument");
    if (!op->persistent_ok || std::getenv("EFGP_NO_PERSISTENT_CG") != nullptr) {
        // 2-D 128^2..512^2: the cooperative launch forms the right-hand side, the diagonal and the zero start itself as well (no
        // prepare launch, no fill, no initial operator application); a dead grid barrier leaves iters = -3 and NaN
        if (op->lines_ok && std::getenv("EFGP_NO_CG_COOP") == nullptr && std::getenv("EFGP_NO_CG_LINES") == nullptr) {
            DeviceGuard guard_c(op->device, (hipStream_t)stream_);
            if (max_iter <= 0) max_iter = (int)std::min<int64_t>(2 * op->g.M, 2000000000);
            return coop_enqueue(op, ws, sigmasq, 0, nullptr, fy, x, 1, tol, max_iter, early_stop, 0, iters_dev, (hipStream_t)stream_, nullptr,
                                /*nan_on_dead*/ 1, /*hermitian*/ 1, diag_scale_dev, /*b_times_ws*/ 1, /*zero_x0*/ 1);
        }
        set_error("efgp_cg_solve_mean_async: grid does not fit the persistent kernel");
        return EFGP_EUNSUPPORTED;
    }
    hipStream_t stream = (hipStream_t)stream_;
    DeviceGuard guard(op->device, (hipStream_t)stream_);
    if (max_iter <= 0) max_iter = (int)std::min<int64_t>(2 * op->g.M, 2000000000);
    KernelTimer timer("cg_persistent", stream);
    const ToepGeom* gq;
    const double2* const* twq;
    const double2* vq;
    cg_operands(op, &gq, &twq, &vq);
    return persistent_cg_launch(*gq, twq, vq, (const double2*)ws, nullptr, sigmasq, 0, tol,
                                early_stop, 0, max_iter, (const double2*)fy, (double2*)x, 1, iters_dev, stream, diag_scale_dev,
                                1, 1, nullptr, /*hermitian: F*y of a real y, Toeplitz vector of real weights*/ 1,
                                op->h48.vhat ? &op->h48 : nullptr);
}

int efgp_vdot_real(int device, const void* a, int a_is_complex, const void* b, int b_is_complex, int64_t count,
                   double* out_host, void* stream_) {
    EFGP_REQUIRE(out_host, "efgp_vdot_real: null out");
    EFGP_REQUIRE(count >= 0, "efgp_vdot_real: negative count");
    *out_host = 0.0;
    if (count == 0) return EFGP_OK;
    EFGP_REQUIRE(a && b, "efgp_vdot_real: null input");
    DeviceCtx* ctx = device_ctx(device);
    if (!ctx) return EFGP_EHIP;
    hipStream_t stream = (hipStream_t)stream_;
    DeviceGuard guard(device, (hipStream_t)stream_);
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((count + kVecThreads - 1) / kVecThreads, 2048));
    double* partial = (double*)scratch(ctx, SLOT_MISC, (size_t)blocks * sizeof(double));
    double* host = (double*)pinned_host(ctx, (size_t)blocks * sizeof(double));
    if (!partial || !host) return EFGP_ENOMEM;
    const double* pa = (const double*)a;
    const double* pb = (const double*)b;
    if (a_is_complex && b_is_complex)
        hipLaunchKernelGGL((vdot_real_kernel<true, true>), dim3(blocks), dim3(kVecThreads), 0, stream, pa, pb, count, partial);
    else if (a_is_complex)
        hipLaunchKernelGGL((vdot_real_kernel<true, false>), dim3(blocks), dim3(kVecThreads), 0, stream, pa, pb, count, partial);
    else if (b_is_complex)
        hipLaunchKernelGGL((vdot_real_kernel<false, true>), dim3(blocks), dim3(kVecThreads), 0, stream, pa, pb, count, partial);
    else
        hipLaunchKernelGGL((vdot_real_kernel<false, false>), dim3(blocks), dim3(kVecThreads), 0, stream, pa, pb, count, partial);
    EFGP_HIP_CHECK(hipGetLastError());
    EFGP_HIP_CHECK(hipMemcpyAsync(host, partial, (size_t)blocks * sizeof(double), hipMemcpyDeviceToHost, stream));
    EFGP_HIP_CHECK(stream_wait(stream));
    double acc = 0.0;
    for (int i = 0; i < blocks; ++i) acc += host[i];
    *out_host = acc;
    return EFGP_OK;
}

}  // extern "C"
