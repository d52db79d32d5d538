// Persistent fused CG: one workgroup solves one system completely inside ONE kernel launch.
//
// For the headline configuration (d=2, mtot=23 -> Toeplitz FFT 64x64) the whole circulant grid is
// 64 KB, so the entire CG iteration (cg.py:116-150 / :193-241) -- diagonal scaling, zero padding,
// forward FFT, spectral multiply, inverse FFT, crop, <p,Ap>, x/r updates, norms, preconditioner,
// p update, convergence test -- runs out of LDS and registers with no launch, no HBM round trip and
// no host involvement per iteration.  Batched solves map one system per workgroup (grid = rows).
//
// LDS layout: two ping-pong buffers of the padded grid (Stockham autosort FFT, radix 8/4/2 stages);
// the fastest dimension is padded by one complex element so that lines of the other dimensions are
// conflict-free across lanes.  FFT passes are pruned: only lines that can be non-zero are
// transformed forward (the operand occupies the leading n-box), and the inverse passes only produce
// the [n-1, 2n-1) window that the Toeplitz product keeps (efgpnd.py:1289-1290).
//
// Eligibility (checked on the host): every FFT length a power of two, padded grid <= kMaxGrid
// complex elements, M <= kSlots * kThreads.
#include <cmath>
#include <cstdlib>

#include "common.hpp"
#include "toeplitz_cg.hpp"

namespace efgp {

namespace pcg {

constexpr int kThreads = 512;
constexpr int kSlots = 4;            // vector elements per thread (M <= kSlots * kThreads)
constexpr int kMaxGrid = 4608;       // complex elements per ping-pong buffer (2 x 72 KB = 144 KB LDS)
constexpr int kRedWaves = kThreads / 64;

struct Pass {            // one 1-D FFT pass over dimension `dim`
    int dim;
    int n;               // FFT length
    int nstages;
    int radix[6];
    int w1_log2;         // lines are enumerated on a power-of-two padded box (no integer division):
    int lines_log2;      //   line = item & (2^lines_log2 - 1); c1 = line & (2^w1_log2 - 1); c0 = line >> w1_log2
    // line ranges of the two other dimensions [lo, hi)
    int other[2];
    int lo[2], hi[2];
    int in_limit;        // first stage: positions >= in_limit are zero (forward), n otherwise
    int out_lo, out_hi;  // last stage: only positions in [out_lo, out_hi) are stored
    int vs_pos, vs_c0, vs_c1;   // strides of (position, other[0], other[1]) in the UNPADDED spectrum array
    int pstride, s0, s1;        // strides of the same three coordinates in the padded LDS grid
    int tw_lds;                 // offset of this pass's twiddle table in LDS (double2 units) or -1 (use tw_glob)
    const double2* tw_glob;
};

struct Geom {
    int d;
    int n[3];            // block sizes
    int F[3];            // FFT sizes
    int ld[3];           // element strides of the padded grid per dimension
    int M;
    int padded;          // elements per buffer
    int npass;           // forward passes (== d), inverse passes are derived
    Pass fwd[3];
    Pass inv[3];
    const double2* tw[3];    // twiddle table per dimension: tw[a][q] = exp(-2 pi i q / F[a])  (global memory)
    int tw_lds_off[3];       // offset (in double2) of the LDS copy of tw[a] behind the two buffers, or -1
    int tw_lds_total;        // double2 elements of LDS used by twiddle copies
    int fuse_mid;            // 1: last forward stage, spectral multiply and first inverse stage run fused in registers
};

struct Args {
    Geom g;
    const double2* ws;
    const double* diag;      // Jacobi diagonal [M] or null
    const double* diag_scale;   // when diag is null and this is not: diagonal = (*diag_scale) * |ws|^2 + sigmasq (device scalar)
    int b_times_ws;          // right-hand side is ws .* b (the fit's D F*y, efgpnd.py:792)
    int zero_x0;             // x0 = 0: x is output only and the initial operator application is skipped (A 0 = 0)
    const double2* x0;       // start vectors (read once, before x is written; normally x itself: in place)
    const double2* vhat;     // [prod F] (unpadded row-major), already divided by prod F
    double sigmasq;
    int variant;
    double tol;
    int early_stop;
    int batched;
    int max_iter;
    const double2* b;
    double2* x;              // in: x0, out: solution
    int* iters;              // per row
    double* hist;            // diagnostic: row 0's |r_i| / |b| per iteration (efgp_cg_record_history) or null
    int hist_cap;
    // Lanczos mode (efgp_lanczos; reference: logdet_slq, efgpnd.py:1716-1738): lz_steps > 0 runs that many three-term
    // recurrence steps on A (variant as given) from q_0 = b / |b| instead of the CG loop; row r's alpha_k, beta_k go to
    // lz_alpha / lz_beta [r * lz_steps + k], |b|^2 to lz_norm2[r], the number of steps taken to iters[r]
    int lz_steps;
    double* lz_alpha;
    double* lz_beta;
    double* lz_norm2;
#ifdef EFGP_CG_STAMPS
    long long* stamps;       // diagnostic build only: cycles per phase class, accumulated by block 0
#endif
};

#ifdef EFGP_CG_STAMPS
#define EFGP_STAMP(slot)                                                     \
    do {                                                                     \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                           \
            const long long now__ = (long long)__builtin_readcyclecounter(); \
            a.stamps[slot] += now__ - stamp_prev;                            \
            stamp_prev = now__;                                              \
        }                                                                    \
    } while (0)
#else
#define EFGP_STAMP(slot) \
    do {                 \
    } while (0)
#endif

__device__ __forceinline__ double2 cmulp(double2 a, double2 b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
// multiply by -i (forward) : (x, y) -> (y, -x)
__device__ __forceinline__ double2 mul_mi(double2 a) { return make_double2(a.y, -a.x); }

template <int R>
__device__ __forceinline__ void dft_fwd(double2 (&v)[R]);

template <>
__device__ __forceinline__ void dft_fwd<2>(double2 (&v)[2]) {
    double2 a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
}

template <>
__device__ __forceinline__ void dft_fwd<4>(double2 (&v)[4]) {
    double2 t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]);
    double2 t2 = cadd(v[1], v[3]), t3 = mul_mi(csub(v[1], v[3]));
    v[0] = cadd(t0, t2);
    v[1] = cadd(t1, t3);
    v[2] = csub(t0, t2);
    v[3] = csub(t1, t3);
}

template <>
__device__ __forceinline__ void dft_fwd<8>(double2 (&v)[8]) {
    // two interleaved DFT4 (even / odd), then combine with w8^m
    double2 e[4] = {v[0], v[2], v[4], v[6]};
    double2 o[4] = {v[1], v[3], v[5], v[7]};
    dft_fwd<4>(e);
    dft_fwd<4>(o);
    const double h = 0.70710678118654752440;
    // o[m] *= exp(-2 pi i m / 8)
    double2 o1 = make_double2(h * (o[1].x + o[1].y), h * (o[1].y - o[1].x));
    double2 o2 = mul_mi(o[2]);
    double2 o3 = make_double2(h * (o[3].y - o[3].x), -h * (o[3].x + o[3].y));
    v[0] = cadd(e[0], o[0]);
    v[4] = csub(e[0], o[0]);
    v[1] = cadd(e[1], o1);
    v[5] = csub(e[1], o1);
    v[2] = cadd(e[2], o2);
    v[6] = csub(e[2], o2);
    v[3] = cadd(e[3], o3);
    v[7] = csub(e[3], o3);
}

// eight named registers with a compile-time accessor (a plain array here ends up in scratch memory)
struct V8 {
    double2 a0, a1, a2, a3, a4, a5, a6, a7;
};
template <int T>
__device__ __forceinline__ double2& v8(V8& v) {
    if constexpr (T == 0) return v.a0;
    else if constexpr (T == 1) return v.a1;
    else if constexpr (T == 2) return v.a2;
    else if constexpr (T == 3) return v.a3;
    else if constexpr (T == 4) return v.a4;
    else if constexpr (T == 5) return v.a5;
    else if constexpr (T == 6) return v.a6;
    else return v.a7;
}
template <int T, int R>
__device__ __forceinline__ void spectral_mul_regs(double2 (&v)[R], V8& sp) {
    if constexpr (T < R) {
        double2 val = cmulp(v[T], v8<T>(sp));
        val.y = -val.y;
        v[T] = val;
        spectral_mul_regs<T + 1, R>(v, sp);
    }
}
template <int T, int R>
__device__ __forceinline__ void spectral_load_regs(V8& sp, const double2* __restrict__ vhat, int idx0, int step) {
    if constexpr (T < R) {
        v8<T>(sp) = vhat[idx0 + T * step];
        spectral_load_regs<T + 1, R>(sp, vhat, idx0, step);
    }
}

// One Stockham stage of radix R over all lines of a pass.  INV: inverse transform (conjugate trick).
// `first`/`last` select the pruning predicates.  MID (forward only): this is the last stage of the last
// forward pass; its outputs are multiplied by the spectrum `vreg` (held in registers, one butterfly per
// thread) and immediately pushed through the FIRST inverse stage of the same dimension (same radix,
// Ns = 1, no twiddles: the outputs k + t*n/R of the forward stage are exactly the inputs of that
// inverse stage), saving one LDS round trip per iteration.
template <int R, bool INV, bool MID>
__device__ __forceinline__ void stage(const Geom& g, const Pass& p, int Ns, int ns_log2, bool first, bool last,
                                      const double2* __restrict__ src, double2* __restrict__ dst,
                                      const double2* __restrict__ tw, const double2* __restrict__ vhat, V8& sp,
                                      bool use_sp) {
    const int n = p.n;
    const int nb = n / R;                                  // butterflies per line
    const int w0 = p.hi[0] - p.lo[0], w1 = p.hi[1] - p.lo[1];
    const int pstride = p.pstride;
    const int s0 = p.s0, s1 = p.s1;
    const int tw_step = n / (Ns * R);
    const int items = nb << p.lines_log2;
    const int lmask = (1 << p.lines_log2) - 1, w1mask = (1 << p.w1_log2) - 1;
    for (int item = threadIdx.x; item < items; item += kThreads) {
        const int line = item & lmask;                     // lines vary fastest across lanes
        const int j = item >> p.lines_log2;
        const int c1r = line & w1mask, c0r = line >> p.w1_log2;
        if (c1r >= w1 || c0r >= w0) continue;
        const int c1 = p.lo[1] + c1r, c0 = p.lo[0] + c0r;
        const int base = c0 * s0 + c1 * s1;
        const int k = j & (Ns - 1);
        double2 v[R];
        const int a_in = base + j * pstride, in_step = nb * pstride, tw_k = k * tw_step;
#pragma unroll
        for (int t = 0; t < R; ++t) {
            double2 val = make_double2(0.0, 0.0);
            if (!first || j + t * nb < p.in_limit) val = src[a_in + t * in_step];
            if (INV) val.y = -val.y;
            if (t > 0 && Ns > 1) val = cmulp(val, tw[t * tw_k]);
            v[t] = val;
        }
        dft_fwd<R>(v);
        if (MID) {
            // spectral multiply, then the first inverse stage (Ns = 1) on the same R values
            if (use_sp) {
                spectral_mul_regs<0, R>(v, sp);
            } else {
                const int i0 = ((j - k) * R + k) * p.vs_pos + c0 * p.vs_c0 + c1 * p.vs_c1, istep = Ns * p.vs_pos;
#pragma unroll
                for (int t = 0; t < R; ++t) {
                    double2 val = cmulp(v[t], vhat[i0 + t * istep]);
                    val.y = -val.y;                        // conj in
                    v[t] = val;
                }
            }
            dft_fwd<R>(v);
            // inverse stage 0 writes position j*R + t; if it is also the last inverse stage of the pass
            // (single-stage FFT) the crop window applies
            const bool only = p.nstages == 1;
            const int a_out = base + j * R * pstride;
#pragma unroll
            for (int t = 0; t < R; ++t) {
                const int pos = j * R + t;
                if (only && (pos < p.in_limit - 1 || pos >= 2 * p.in_limit - 1)) continue;   // in_limit == n[dim] (forward pass)
                double2 val = v[t];
                val.y = -val.y;                            // conj out
                dst[a_out + t * pstride] = val;
            }
            continue;
        }
        const int ob = (j - k) * R + k;
        const int a_out = base + ob * pstride, out_step = Ns * pstride;
#pragma unroll
        for (int t = 0; t < R; ++t) {
            const int pos = ob + t * Ns;
            if (last && (pos < p.out_lo || pos >= p.out_hi)) continue;
            double2 val = v[t];
            if (INV) val.y = -val.y;
            if (vhat) val = cmulp(val, vhat[pos * p.vs_pos + c0 * p.vs_c0 + c1 * p.vs_c1]);
            dst[a_out + t * out_step] = val;
        }
    }
}

// runs the stages [s_begin, s_end) of a pass
template <bool INV>
__device__ __forceinline__ void run_pass(const Geom& g, const Pass& p, int s_begin, int s_end, double2*& cur,
                                         double2*& alt, const double2* tw, const double2* vhat_last, bool mid_last,
                                         V8& sp, bool use_sp) {
    int Ns = 1, lg = 0;
    for (int s = 0; s < s_begin; ++s) {
        Ns *= p.radix[s];
        lg += p.radix[s] == 8 ? 3 : (p.radix[s] == 4 ? 2 : 1);
    }
    for (int s = s_begin; s < s_end; ++s) {
        const bool first = s == 0, last = s == p.nstages - 1;
        const double2* vh = (last && !mid_last ? vhat_last : nullptr);
        const bool mid = last && mid_last && !INV;
        if (mid) {
            switch (p.radix[s]) {
                case 8: stage<8, false, true>(g, p, Ns, lg, first, last, cur, alt, tw, vhat_last, sp, use_sp); break;
                case 4: stage<4, false, true>(g, p, Ns, lg, first, last, cur, alt, tw, vhat_last, sp, use_sp); break;
                default: stage<2, false, true>(g, p, Ns, lg, first, last, cur, alt, tw, vhat_last, sp, use_sp); break;
            }
        } else {
            switch (p.radix[s]) {
                case 8: stage<8, INV, false>(g, p, Ns, lg, first, last, cur, alt, tw, vh, sp, false); break;
                case 4: stage<4, INV, false>(g, p, Ns, lg, first, last, cur, alt, tw, vh, sp, false); break;
                default: stage<2, INV, false>(g, p, Ns, lg, first, last, cur, alt, tw, vh, sp, false); break;
            }
        }
        Ns *= p.radix[s];
        lg += p.radix[s] == 8 ? 3 : (p.radix[s] == 4 ? 2 : 1);
        __syncthreads();
        double2* t = cur;
        cur = alt;
        alt = t;
    }
}

// wave-level sum with DPP moves (VALU only; the LDS pipe stays free for the FFT stages).  gfx9 row_shr /
// row_bcast controls; the value ends up in lane 63 and is broadcast through an SGPR.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, true);
    return __hiloint2double(hi, lo);
}
#define EFGP_DPP_ADD(v, ctrl, rmask) v += dpp_move<ctrl, rmask>(v)
__device__ __forceinline__ double wave_sum(double v) {
    EFGP_DPP_ADD(v, 0x111, 0xF);   // row_shr:1
    EFGP_DPP_ADD(v, 0x112, 0xF);   // row_shr:2
    EFGP_DPP_ADD(v, 0x114, 0xF);   // row_shr:4
    EFGP_DPP_ADD(v, 0x118, 0xF);   // row_shr:8   -> lane 15 of every 16-lane row holds the row sum
    EFGP_DPP_ADD(v, 0x142, 0xA);   // row_bcast:15 into rows 1 and 3
    EFGP_DPP_ADD(v, 0x143, 0xC);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave sum
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// Jacobi diagonal entry t: explicit array, or scale * |ws_t|^2 + sigmasq with the two roundings of the
// reference's torch expression (efgpnd.py:795-799), or 1 (no preconditioner)
__device__ __forceinline__ double jacobi_entry(const Args& a, double2 w, int t) {
    if (a.diag) return a.diag[t];
    if (a.diag_scale) return __dadd_rn(__dmul_rn(*a.diag_scale, __dadd_rn(__dmul_rn(w.x, w.x), __dmul_rn(w.y, w.y))), a.sigmasq);
    return 1.0;
}

// Block reductions.  The single sum and the pair use DISJOINT scratch (red[2*kRedWaves ..] vs red[0 ..]), and a
// scratch region is rewritten only after other barriers have passed since its last read (the other reduction
// and the FFT phases sit in between), so no barrier is needed in front of the writes.
__device__ __forceinline__ double block_sum2(double v, double* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    double* r1 = red + 2 * kRedWaves;
    if (lane == 0) r1[wid] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < kRedWaves; ++i) t += r1[i];
    return t;
}

// two sums with one barrier
__device__ __forceinline__ void block_sum_pair(double& u, double& v, double* red) {
    u = wave_sum(u);
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) {
        red[wid] = u;
        red[kRedWaves + wid] = v;
    }
    __syncthreads();
    double t = 0.0, w = 0.0;
#pragma unroll
    for (int i = 0; i < kRedWaves; ++i) {
        t += red[i];
        w += red[kRedWaves + i];
    }
    u = t;
    v = w;
}

// flat block index -> offset in the padded grid, optionally shifted by (n-1) per dimension
__device__ __forceinline__ int grid_offset(const Geom& g, int flat, int shift) {
    int off = 0;
    for (int a = g.d - 1; a >= 0; --a) {
        const int ia = flat % g.n[a];
        flat /= g.n[a];
        off += (ia + shift * (g.n[a] - 1)) * g.ld[a];
    }
    return off;
}

// Lanczos three-term recurrence inside the persistent kernels (shared by both): no launch and no host round trip per step
// (the host loop it replaces made two device->host reads per step: 100 probes x 25 steps x 2 at the defaults).
#define EFGP_LANCZOS_BRANCH(NS_, TID_)                                                                          \
    if (a.lz_steps > 0) {                                                                                       \
        double2 q[NS_], qp[NS_], vv[NS_];                                                                       \
        double nb = 0.0;                                                                                        \
        _Pragma("unroll") for (int s = 0; s < NS_; ++s) {                                                       \
            const int t = TID_ + s * kThreads;                                                                  \
            q[s] = t < M ? a.b[base + t] : make_double2(0.0, 0.0);                                              \
            qp[s] = make_double2(0.0, 0.0);                                                                     \
            nb += q[s].x * q[s].x + q[s].y * q[s].y;                                                            \
        }                                                                                                       \
        nb = block_sum2(nb, red);                                                                               \
        const double inv_nb = nb > 0.0 ? 1.0 / sqrt(nb) : 0.0;                                                  \
        _Pragma("unroll") for (int s = 0; s < NS_; ++s) q[s] = make_double2(q[s].x * inv_nb, q[s].y * inv_nb);  \
        double beta_prev = 0.0;                                                                                 \
        int k = 0;                                                                                              \
        while (k < a.lz_steps) {                                                                                \
            apply_A(q, vv);                                                                                     \
            double al = 0.0;                                                                                    \
            _Pragma("unroll") for (int s = 0; s < NS_; ++s) {                                                   \
                vv[s].x -= beta_prev * qp[s].x;                                                                 \
                vv[s].y -= beta_prev * qp[s].y;                                                                 \
                al += q[s].x * vv[s].x + q[s].y * vv[s].y;                                                      \
            }                                                                                                   \
            al = block_sum2(al, red);                                                                           \
            double bt = 0.0, unused = 0.0;                                                                      \
            _Pragma("unroll") for (int s = 0; s < NS_; ++s) {                                                   \
                vv[s].x -= al * q[s].x;                                                                         \
                vv[s].y -= al * q[s].y;                                                                         \
                bt += vv[s].x * vv[s].x + vv[s].y * vv[s].y;                                                    \
            }                                                                                                   \
            block_sum_pair(bt, unused, red);                                                                    \
            bt = sqrt(bt);                                                                                      \
            if (TID_ == 0) {                                                                                    \
                a.lz_alpha[(int64_t)row * a.lz_steps + k] = al;                                                 \
                a.lz_beta[(int64_t)row * a.lz_steps + k] = bt;                                                  \
            }                                                                                                   \
            ++k;                                                                                                \
            if (bt < 1e-12) break;                                                                              \
            const double ib = 1.0 / bt;                                                                         \
            _Pragma("unroll") for (int s = 0; s < NS_; ++s) {                                                   \
                qp[s] = q[s];                                                                                   \
                q[s] = make_double2(vv[s].x * ib, vv[s].y * ib);                                                \
            }                                                                                                   \
            beta_prev = bt;                                                                                     \
        }                                                                                                       \
        if (TID_ == 0) {                                                                                        \
            a.iters[row] = k;                                                                                   \
            if (a.lz_norm2) a.lz_norm2[row] = nb;                                                               \
        }                                                                                                       \
        return;                                                                                                 \
    }

__global__ __launch_bounds__(kThreads) void cg_persistent_kernel(Args a) {
    extern __shared__ double2 lds2[];
    __shared__ double red[3 * kRedWaves];
    const Geom& g = a.g;
    double2* bufA = lds2;
    double2* bufB = lds2 + g.padded;
    // twiddle tables: LDS copies (when they fit behind the two buffers) are filled once
    double2* tw_lds_base = lds2 + 2 * g.padded;
    for (int a_ = 0; a_ < g.d; ++a_) {
        if (g.tw_lds_off[a_] < 0) continue;
        bool dup = false;
        for (int b_ = 0; b_ < a_; ++b_) dup = dup || (g.tw_lds_off[b_] == g.tw_lds_off[a_]);
        if (!dup)
            for (int q = threadIdx.x; q < g.F[a_]; q += kThreads) tw_lds_base[g.tw_lds_off[a_] + q] = g.tw[a_][q];
    }
    const int row = blockIdx.x;
    const int M = g.M;
    const int64_t base = (int64_t)row * M;

    // per-thread slices of the vectors (element t lives in thread t % kThreads, slot t / kThreads)
    double2 xv[kSlots], rv[kSlots], pv[kSlots], wsv[kSlots];
    double dg[kSlots];
    const bool precond = a.diag != nullptr || a.diag_scale != nullptr;
    int off_in[kSlots], off_out[kSlots];
#pragma unroll
    for (int s = 0; s < kSlots; ++s) {
        const int t = threadIdx.x + s * kThreads;
        if (t < M) {
            xv[s] = a.zero_x0 ? make_double2(0.0, 0.0) : a.x0[base + t];
            wsv[s] = a.ws[t];
            dg[s] = jacobi_entry(a, wsv[s], t);
            off_in[s] = grid_offset(g, t, 0);
            off_out[s] = grid_offset(g, t, 1);
        } else {
            xv[s] = wsv[s] = make_double2(0.0, 0.0);
            dg[s] = 1.0;
            off_in[s] = off_out[s] = 0;
        }
        rv[s] = pv[s] = make_double2(0.0, 0.0);
    }

    // spectrum values of this thread's butterfly in the fused middle stage (when it has one item per thread)
    V8 sp;
    sp.a0 = sp.a1 = sp.a2 = sp.a3 = sp.a4 = sp.a5 = sp.a6 = sp.a7 = make_double2(0.0, 0.0);
    bool use_sp = false;
    {
        const Pass& pm = g.fwd[g.npass - 1];
        const int R = pm.radix[pm.nstages - 1];
        const int nb = pm.n / R;
        const int items = nb << pm.lines_log2;
        use_sp = g.fuse_mid && items <= kThreads;
        if (use_sp && (int)threadIdx.x < items) {
            const int line = threadIdx.x & ((1 << pm.lines_log2) - 1);
            const int j = threadIdx.x >> pm.lines_log2;
            const int c1r = line & ((1 << pm.w1_log2) - 1), c0r = line >> pm.w1_log2;
            if (c1r < pm.hi[1] - pm.lo[1] && c0r < pm.hi[0] - pm.lo[0]) {
                const int k = j & (nb - 1);          // last stage: Ns = n / R = nb
                const int i0 = ((j - k) * R + k) * pm.vs_pos + (pm.lo[0] + c0r) * pm.vs_c0 + (pm.lo[1] + c1r) * pm.vs_c1;
                const int istep = nb * pm.vs_pos;
                if (R == 8) spectral_load_regs<0, 8>(sp, a.vhat, i0, istep);
                else if (R == 4) spectral_load_regs<0, 4>(sp, a.vhat, i0, istep);
                else spectral_load_regs<0, 2>(sp, a.vhat, i0, istep);
            }
        }
    }
    __syncthreads();      // twiddle copies visible

    // A u for this thread's slots: u -> LDS -> pruned FFT -> .* vhat -> pruned inverse FFT -> crop
#ifdef EFGP_CG_STAMPS
    long long stamp_prev = (long long)__builtin_readcyclecounter();
#endif
    auto apply_A = [&](const double2 (&u)[kSlots], double2 (&Au)[kSlots]) __attribute__((always_inline)) {
        EFGP_STAMP(7);
#pragma unroll
        for (int s = 0; s < kSlots; ++s) {
            const int t = threadIdx.x + s * kThreads;
            if (t < M) bufA[off_in[s]] = cmulp(wsv[s], u[s]);
        }
        __syncthreads();
        EFGP_STAMP(0);
        double2* cur = bufA;
        double2* alt = bufB;
        for (int q = 0; q < g.npass; ++q) {
            const bool lastp = q == g.npass - 1;
            const Pass& pf = g.fwd[q];
            run_pass<false>(g, pf, 0, pf.nstages, cur, alt, pf.tw_lds >= 0 ? tw_lds_base + pf.tw_lds : pf.tw_glob,
                            lastp ? a.vhat : nullptr, lastp && g.fuse_mid, sp, use_sp);
            EFGP_STAMP(1 + q);
        }
        for (int q = 0; q < g.npass; ++q) {   // with the fused middle stage, inverse pass 0 starts at its stage 1
            const Pass& pi = g.inv[q];
            run_pass<true>(g, pi, (q == 0 && g.fuse_mid) ? 1 : 0, pi.nstages, cur, alt,
                           pi.tw_lds >= 0 ? tw_lds_base + pi.tw_lds : pi.tw_glob, nullptr, false, sp, false);
            EFGP_STAMP(4 + q);
        }
#pragma unroll
        for (int s = 0; s < kSlots; ++s) {
            const int t = threadIdx.x + s * kThreads;
            if (t < M) {
                double2 Tu = cur[off_out[s]];
                double2 gg = cmulp(wsv[s], Tu);
                if (a.variant == 0) Au[s] = make_double2(gg.x + a.sigmasq * u[s].x, gg.y + a.sigmasq * u[s].y);
                else Au[s] = make_double2(gg.x / a.sigmasq + u[s].x, gg.y / a.sigmasq + u[s].y);
            } else {
                Au[s] = make_double2(0.0, 0.0);
            }
        }
        __syncthreads();     // the buffers are reused by the next application
        EFGP_STAMP(8);
    };

    EFGP_LANCZOS_BRANCH(kSlots, (int)threadIdx.x)

    // r = b - A x0, z = r / diag, p = z
    double2 Ap[kSlots];
    if (a.zero_x0) {
#pragma unroll
        for (int s = 0; s < (int)(sizeof(Ap) / sizeof(Ap[0])); ++s) Ap[s] = make_double2(0.0, 0.0);
    } else {
        apply_A(xv, Ap);
    }
    double rz = 0.0, bb = 0.0;
#pragma unroll
    for (int s = 0; s < kSlots; ++s) {
        const int t = threadIdx.x + s * kThreads;
        if (t < M) {
            double2 bv = a.b[base + t];
            if (a.b_times_ws) bv = cmulp(wsv[s], bv);
            rv[s] = csub(bv, Ap[s]);
            pv[s] = precond ? make_double2(rv[s].x / dg[s], rv[s].y / dg[s]) : rv[s];
            rz += rv[s].x * pv[s].x + rv[s].y * pv[s].y;
            bb += bv.x * bv.x + bv.y * bv.y;
        }
    }
    block_sum_pair(rz, bb, red);
    const double bn = sqrt(bb);
    const double den = bn > 0.0 ? bn : 1.0;

    int it = 0;
    for (; it < a.max_iter;) {
        apply_A(pv, Ap);
        double pAp = 0.0;
#pragma unroll
        for (int s = 0; s < kSlots; ++s) pAp += pv[s].x * Ap[s].x + pv[s].y * Ap[s].y;
        pAp = block_sum2(pAp, red) + 1e-16;
        const double alpha = rz / pAp;
        double rr = 0.0, rzn = 0.0;
        double2 zv[kSlots];
#pragma unroll
        for (int s = 0; s < kSlots; ++s) {
            xv[s].x += alpha * pv[s].x;
            xv[s].y += alpha * pv[s].y;
            rv[s].x -= alpha * Ap[s].x;
            rv[s].y -= alpha * Ap[s].y;
            zv[s] = precond ? make_double2(rv[s].x / dg[s], rv[s].y / dg[s]) : rv[s];
            rr += rv[s].x * rv[s].x + rv[s].y * rv[s].y;
            rzn += rv[s].x * zv[s].x + rv[s].y * zv[s].y;
        }
        block_sum_pair(rr, rzn, red);
        ++it;
        const double rnorm = sqrt(rr);
        if (a.hist && row == 0 && threadIdx.x == 0 && it <= a.hist_cap) a.hist[it - 1] = rnorm / (den + 1e-16);
        const bool conv = a.early_stop && ((rnorm / (den + 1e-16) < a.tol) || (a.batched && rnorm < 1e-12));
        if (!a.batched && conv) break;                  // cg.py:132 (before the preconditioner / p update)
        const double beta = rzn / (rz + 1e-16);
#pragma unroll
        for (int s = 0; s < kSlots; ++s) pv[s] = make_double2(zv[s].x + beta * pv[s].x, zv[s].y + beta * pv[s].y);
        rz = rzn;
        if (conv) break;                                // cg.py:229-241 (after the p update)
    }
#pragma unroll
    for (int s = 0; s < kSlots; ++s) {
        const int t = threadIdx.x + s * kThreads;
        if (t < M) a.x[base + t] = xv[s];
    }
    if (threadIdx.x == 0) a.iters[row] = it;
}

// ------------------------------------------------------------------------------------------------
// Specialised kernel for the headline shape: d = 2, circulant grid 64 x 64 (mtot <= 32), radix-8
// Stockham stages with compile-time strides.  All addresses, twiddles and the thread's slice of the
// spectrum are computed once, outside the CG loop; a stage is 8 ds_read_b128 + ~100 fp64 ops +
// 8 ds_write_b128 with immediate offsets.  Same arithmetic as the generic kernel above.
// ------------------------------------------------------------------------------------------------
namespace s64 {
constexpr int F = 64, LD = 65, BUF = F * LD;

__device__ __forceinline__ double2 conjd(double2 a) { return make_double2(a.x, -a.y); }

// v[t] = src[t * STRIDE] (t < 8), entries with t >= nvalid are zero
template <int STRIDE>
__device__ __forceinline__ void load8(const double2* __restrict__ src, int nvalid, double2 (&v)[8]) {
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = t < nvalid ? src[t * STRIDE] : make_double2(0.0, 0.0);
}
template <int STRIDE>
__device__ __forceinline__ void load8_all(const double2* __restrict__ src, double2 (&v)[8]) {
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = src[t * STRIDE];
}
template <int STRIDE>
__device__ __forceinline__ void store8_all(double2* __restrict__ dst, const double2 (&v)[8]) {
#pragma unroll
    for (int t = 0; t < 8; ++t) dst[t * STRIDE] = v[t];
}
__device__ __forceinline__ void twiddle8(double2 (&v)[8], const double2 (&w)[7]) {
#pragma unroll
    for (int t = 1; t < 8; ++t) v[t] = cmulp(v[t], w[t - 1]);
}
__device__ __forceinline__ void conj8(double2 (&v)[8]) {
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t].y = -v[t].y;
}
}  // namespace s64

__global__ __launch_bounds__(kThreads) void cg_persistent_2d64_kernel(Args a) {
    using namespace s64;
    // Row pitch 72 (not s64::LD = 65): the eight butterflies of a row sit in EIGHT ADJACENT LANES of one wave (lane = row * 8 + j),
    // so that both radix-8 stages of a row transform run inside that wave with only a wave-level fence between them (no
    // workgroup barrier: 2 of the 10 barriers per iteration gone, and the three row waves no longer wait for each other).
    // With that lane map a 16-lane LDS group covers two rows x eight j: pitch = 8 mod 16 (in 16-byte elements) puts them on
    // 16 distinct slots.  Column passes read 64 consecutive elements per wave: conflict free at any pitch.
    constexpr int LD = 72, BUF = F * LD;
    constexpr int KS = 2;                   // M = n*n <= 1024 = KS * kThreads
    extern __shared__ double2 lds2[];
    __shared__ double red[3 * kRedWaves];
    double2* const bufA = lds2;
    double2* const bufB = lds2 + BUF;
    const int n = a.g.n[0];                 // block size per dimension (n0 == n1 for this kernel)
    const int M = a.g.M;
    const int row = blockIdx.x;
    const int64_t base = (int64_t)row * M;
    const int tid = threadIdx.x;

    // ---- iteration-invariant per-thread data -------------------------------------------------
    // row passes: lane = row * 8 + butterfly (rows 8w..8w+7 in wave w); column passes: 64 columns x 8 (all threads)
    const int j_row = tid & 7, r_row = tid >> 3;
    const bool row_act = r_row < n;
    const int c_col = tid & 63, j_col = tid >> 6;
    // valid leading inputs of the pruned first stages: positions j + 8t < n
    const int nv_row = j_row < n ? (n - 1 - j_row) / 8 + 1 : 0;
    const int nv_col = j_col < n ? (n - 1 - j_col) / 8 + 1 : 0;
    // crop window [n-1, 2n-1) masks of the pruned last inverse stages: position j + 8t
    unsigned keep_col = 0, keep_row = 0;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int pc = j_col + 8 * t, pr = j_row + 8 * t;
        if (pc >= n - 1 && pc < 2 * n - 1) keep_col |= 1u << t;
        if (pr >= n - 1 && pr < 2 * n - 1) keep_row |= 1u << t;
    }
    double2 twr[7], twc[7], spec[8];
#pragma unroll
    for (int t = 1; t < 8; ++t) {
        twr[t - 1] = a.g.tw[0][(j_row & 7) * t];
        twc[t - 1] = a.g.tw[0][j_col * t];
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) spec[t] = a.vhat[(j_col + 8 * t) * F + c_col];

    double2 xv[KS], rv[KS], pv[KS], wsv[KS];
    double dg[KS];
    const bool precond = a.diag != nullptr || a.diag_scale != nullptr;
    int off_in[KS], off_out[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int t = tid + s * kThreads;
        if (t < M) {
            xv[s] = a.zero_x0 ? make_double2(0.0, 0.0) : a.x0[base + t];
            wsv[s] = a.ws[t];
            dg[s] = jacobi_entry(a, wsv[s], t);
            const int i0 = t / n, i1 = t - i0 * n;
            off_in[s] = i0 * LD + i1;
            off_out[s] = (i0 + n - 1) * LD + (i1 + n - 1);
        } else {
            xv[s] = wsv[s] = make_double2(0.0, 0.0);
            dg[s] = 1.0;
            off_in[s] = off_out[s] = 0;
        }
        rv[s] = pv[s] = make_double2(0.0, 0.0);
    }
#ifdef EFGP_CG_STAMPS
    long long stamp_prev = (long long)__builtin_readcyclecounter();
#endif

    auto apply_A = [&](const double2 (&u)[KS], double2 (&Au)[KS]) __attribute__((always_inline)) {
        EFGP_STAMP(7);
#pragma unroll
        for (int s = 0; s < KS; ++s)
            if (tid + s * kThreads < M) bufA[off_in[s]] = cmulp(wsv[s], u[s]);
        __syncthreads();
        EFGP_STAMP(0);
        double2 v[8];
        // P1/P2: forward FFT along dim 1 of the n non-zero rows (A -> B -> A); a row lives in one wave: wave-level fence only
        if (row_act) {
            load8<8>(bufA + r_row * LD + j_row, nv_row, v);
            dft_fwd<8>(v);
            store8_all<1>(bufB + r_row * LD + j_row * 9, v);      // butterfly j at [9 j, 9 j + 8): both sides conflict free
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (row_act) {
            load8_all<9>(bufB + r_row * LD + j_row, v);
            twiddle8(v, twr);
            dft_fwd<8>(v);
            store8_all<8>(bufA + r_row * LD + j_row, v);
        }
        __syncthreads();
        EFGP_STAMP(1);
        // P3: forward FFT along dim 0, stage 1 (A -> B), rows >= n are zero
        load8<8 * LD>(bufA + j_col * LD + c_col, nv_col, v);
        dft_fwd<8>(v);
        store8_all<LD>(bufB + j_col * 8 * LD + c_col, v);
        __syncthreads();
        // P4: stage 2 of the forward column FFT, spectral multiply, stage 1 of the inverse column FFT (B -> A)
        load8_all<8 * LD>(bufB + j_col * LD + c_col, v);
        twiddle8(v, twc);
        dft_fwd<8>(v);
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = conjd(cmulp(v[t], spec[t]));
        dft_fwd<8>(v);
        conj8(v);
        store8_all<LD>(bufA + j_col * 8 * LD + c_col, v);
        __syncthreads();
        EFGP_STAMP(2);
        // P5: inverse column FFT stage 2 (A -> B), only rows of the crop window are stored
        load8_all<8 * LD>(bufA + j_col * LD + c_col, v);
        conj8(v);
        twiddle8(v, twc);
        dft_fwd<8>(v);
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (keep_col & (1u << t)) bufB[(j_col + 8 * t) * LD + c_col] = conjd(v[t]);
        __syncthreads();
        EFGP_STAMP(4);
        // P6/P7: inverse FFT along dim 1 of the n window rows (B -> A -> B), cropped columns at the end
        const int wrow = n - 1 + r_row;
        if (row_act) {
            load8_all<8>(bufB + wrow * LD + j_row, v);
            conj8(v);
            dft_fwd<8>(v);
            conj8(v);
            store8_all<1>(bufA + wrow * LD + j_row * 9, v);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (row_act) {
            load8_all<9>(bufA + wrow * LD + j_row, v);
            conj8(v);
            twiddle8(v, twr);
            dft_fwd<8>(v);
#pragma unroll
            for (int t = 0; t < 8; ++t)
                if (keep_row & (1u << t)) bufB[wrow * LD + j_row + 8 * t] = conjd(v[t]);
        }
        __syncthreads();
        EFGP_STAMP(5);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (tid + s * kThreads < M) {
                const double2 gg = cmulp(wsv[s], bufB[off_out[s]]);
                if (a.variant == 0) Au[s] = make_double2(gg.x + a.sigmasq * u[s].x, gg.y + a.sigmasq * u[s].y);
                else Au[s] = make_double2(gg.x / a.sigmasq + u[s].x, gg.y / a.sigmasq + u[s].y);
            } else {
                Au[s] = make_double2(0.0, 0.0);
            }
        }
        // no barrier here: the next LDS writes are reduction scratch (disjoint) and, behind that reduction's
        // barrier, bufA -- whose last readers (P7) are already behind the barrier above
        EFGP_STAMP(8);
    };

    EFGP_LANCZOS_BRANCH(KS, tid)

    double2 Ap[KS];
    if (a.zero_x0) {
#pragma unroll
        for (int s = 0; s < (int)(sizeof(Ap) / sizeof(Ap[0])); ++s) Ap[s] = make_double2(0.0, 0.0);
    } else {
        apply_A(xv, Ap);
    }
    double rz = 0.0, bb = 0.0;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        if (tid + s * kThreads < M) {
            double2 bv = a.b[base + tid + s * kThreads];
            if (a.b_times_ws) bv = cmulp(wsv[s], bv);
            rv[s] = csub(bv, Ap[s]);
            pv[s] = precond ? make_double2(rv[s].x / dg[s], rv[s].y / dg[s]) : rv[s];
            rz += rv[s].x * pv[s].x + rv[s].y * pv[s].y;
            bb += bv.x * bv.x + bv.y * bv.y;
        }
    }
    block_sum_pair(rz, bb, red);
    const double bn = sqrt(bb);
    const double den = bn > 0.0 ? bn : 1.0;
    int it = 0;
    for (; it < a.max_iter;) {
        apply_A(pv, Ap);
        double pAp = 0.0;
#pragma unroll
        for (int s = 0; s < KS; ++s) pAp += pv[s].x * Ap[s].x + pv[s].y * Ap[s].y;
        pAp = block_sum2(pAp, red) + 1e-16;
        const double alpha = rz / pAp;
        double rr = 0.0, rzn = 0.0;
        double2 zv[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            xv[s].x += alpha * pv[s].x;
            xv[s].y += alpha * pv[s].y;
            rv[s].x -= alpha * Ap[s].x;
            rv[s].y -= alpha * Ap[s].y;
            zv[s] = precond ? make_double2(rv[s].x / dg[s], rv[s].y / dg[s]) : rv[s];
            rr += rv[s].x * rv[s].x + rv[s].y * rv[s].y;
            rzn += rv[s].x * zv[s].x + rv[s].y * zv[s].y;
        }
        block_sum_pair(rr, rzn, red);
        ++it;
        const double rnorm = sqrt(rr);
        if (a.hist && row == 0 && threadIdx.x == 0 && it <= a.hist_cap) a.hist[it - 1] = rnorm / (den + 1e-16);
        const bool conv = a.early_stop && ((rnorm / (den + 1e-16) < a.tol) || (a.batched && rnorm < 1e-12));
        if (!a.batched && conv) break;
        const double beta = rzn / (rz + 1e-16);
#pragma unroll
        for (int s = 0; s < KS; ++s) pv[s] = make_double2(zv[s].x + beta * pv[s].x, zv[s].y + beta * pv[s].y);
        rz = rzn;
        if (conv) break;
    }
#pragma unroll
    for (int s = 0; s < KS; ++s)
        if (tid + s * kThreads < M) a.x[base + tid + s * kThreads] = xv[s];
    if (tid == 0) a.iters[row] = it;
}

// ------------------------------------------------------------------------------------------------
// Hermitian specialisation of the 64 x 64 solve (the fit's mean system, efgpnd.py:786-814).
//
// The right-hand side D F*y of a REAL y, the Toeplitz vector of real weights and the symmetric real ws make every CG
// vector the coefficient array of a real function: u[-k] = conj u[k] on the centred modes k in [-h, h]^2, h = (n-1)/2.
// The operator then is   coefficients -> real function on the 64 x 64 torus -> times a REAL spectrum -> coefficients,
// and real transforms cost half:
//   * only the rows k0 = 0..h of a vector are stored and transformed (A: h + 1 row lines instead of n),
//   * two real columns f1 = q, q + 32 ride through ONE complex column transform as real and imaginary part
//     (B/C: 32 column lines instead of 64; the spectrum multiply is two real multiplies),
//   * the rows of the result for k0 = 0..h come from the packed columns at +k0 and -k0 (D: h + 1 row lines).
// 88 line transforms per operator instead of 174.  Every line transform runs inside ONE wave (8 adjacent lanes x 8
// points, radix 8 x 8, one exchange through wave-private LDS), so an iteration has 4 workgroup barriers: A | B C | D |
// <p,Ap> | <r,r>,<r,z>.  The CG vectors live in the registers of the row lanes in exactly the positions the first
// radix-8 stage of A consumes and the last stage of D produces (position j + 8t of lane j, t in {0,1,6,7}: modes k1 and
// k1 - 64), so neither the load of ws.*p nor the crop touches LDS.  Dot products weigh the rows k0 > 0 twice.
// Same recurrences, stopping rule and preconditioner as the kernels above; the arithmetic differs from them (and from
// the reference's complex FFTs) by rounding only: the spectrum's rotated imaginary part (|.| ~ 1e-16 relative) and the
// anti-Hermitian rounding noise of the inputs are dropped.
// ------------------------------------------------------------------------------------------------
namespace h64 {
constexpr int kThreadsH = 256, kWavesH = kThreadsH / 64;
constexpr int LDR = 72;                  // pitch of the G rows and of the row lanes' exchange scratch (8 mod 16: conflict free)
constexpr int LT = 40;                   // pitch of the T rows (32 packed columns)
constexpr int LQ = 82;                   // column lanes' exchange scratch per line (2 mod 16)
constexpr int G_ELEMS = 16 * LDR, T_ELEMS = 32 * LT, Q_ELEMS = kWavesH * 8 * LQ;
constexpr int kLdsElems = G_ELEMS + T_ELEMS + Q_ELEMS;     // 80.9 KB

// DFT-8 of (a0, a1, 0, 0, 0, 0, a6, a7): the zero-padded first stage (modes 0..15 and -16..-1 of a 64-point line)
__device__ __forceinline__ void dft8_in4(double2 a0, double2 a1, double2 a6, double2 a7, double2 (&v)[8]) {
    const double2 ia6 = make_double2(-a6.y, a6.x), ia7 = make_double2(-a7.y, a7.x);
    const double2 e0 = cadd(a0, a6), e1 = cadd(a0, ia6), e2 = csub(a0, a6), e3 = csub(a0, ia6);
    const double2 o0 = cadd(a1, a7), o1r = cadd(a1, ia7), o2r = csub(a1, a7), o3r = csub(a1, ia7);
    const double hh = 0.70710678118654752440;
    const double2 o1 = make_double2(hh * (o1r.x + o1r.y), hh * (o1r.y - o1r.x));
    const double2 o2 = mul_mi(o2r);
    const double2 o3 = make_double2(hh * (o3r.y - o3r.x), -hh * (o3r.x + o3r.y));
    v[0] = cadd(e0, o0);
    v[4] = csub(e0, o0);
    v[1] = cadd(e1, o1);
    v[5] = csub(e1, o1);
    v[2] = cadd(e2, o2);
    v[6] = csub(e2, o2);
    v[3] = cadd(e3, o3);
    v[7] = csub(e3, o3);
}

// DFT-8 of which only the outputs 0, 1, 6, 7 are wanted (positions 0..15 and 48..63 of a cropped 64-point line)
__device__ __forceinline__ void dft8_out4(double2 (&v)[8]) {
    double2 e[4] = {v[0], v[2], v[4], v[6]};
    double2 o[4] = {v[1], v[3], v[5], v[7]};
    dft_fwd<4>(e);
    dft_fwd<4>(o);
    const double hh = 0.70710678118654752440;
    const double2 o1 = make_double2(hh * (o[1].x + o[1].y), hh * (o[1].y - o[1].x));
    const double2 o2 = mul_mi(o[2]);
    const double2 o3 = make_double2(hh * (o[3].y - o[3].x), -hh * (o[3].x + o[3].y));
    v[0] = cadd(e[0], o[0]);
    v[1] = cadd(e[1], o1);
    v[6] = csub(e[2], o2);
    v[7] = csub(e[3], o3);
}

// a / b from the correctly rounded reciprocal rb = 1 / b: one residual step (Markstein) -- the quotient the hardware division
// sequence returns for operands in the normal range, in 3 instructions instead of ~30
__device__ __forceinline__ double div_rcp(double a_, double b_, double rb) {
    const double q0 = a_ * rb;
    return fma(fma(-q0, b_, a_), rb, q0);
}

// second half of a 64-point line transform inside a wave: the 8 lanes of a line swap their first-stage outputs through
// `wr` (lane's own 8 slots, stride 1) / `rd` (stride 9), twiddle, second radix-8 stage.  v[t] = X[j + 8 t] on return.
template <bool OUT4 = false>
__device__ __forceinline__ void exchange_stage2(double2 (&v)[8], double2* __restrict__ wr, const double2* __restrict__ rd,
                                                const double2 (&tw)[7]) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    s64::store8_all<1>(wr, v);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    s64::load8_all<9>(rd, v);
    s64::twiddle8(v, tw);
    if (OUT4) dft8_out4(v);
    else dft_fwd<8>(v);
}

__device__ __forceinline__ double block_sum_h(double v, double* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    double* r1 = red + 2 * kWavesH;
    if (lane == 0) r1[wid] = v;
    __syncthreads();
    return (r1[0] + r1[1]) + (r1[2] + r1[3]);
}
__device__ __forceinline__ void block_sum_pair_h(double& u, double& v, double* red) {
    u = wave_sum(u);
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) {
        red[wid] = u;
        red[kWavesH + wid] = v;
    }
    __syncthreads();
    u = (red[0] + red[1]) + (red[2] + red[3]);
    v = (red[kWavesH] + red[kWavesH + 1]) + (red[kWavesH + 2] + red[kWavesH + 3]);
}
}  // namespace h64

template <int VARIANT>
__global__ __launch_bounds__(h64::kThreadsH) void cg_herm64_kernel(Args a) {
    using namespace h64;
    constexpr int F = 64, KS = 4;
    extern __shared__ double2 lds2[];
    __shared__ double red[4 * kWavesH];
    __shared__ double s_rcp;             // 1 / (<r,z> + 1e-16) for the next beta, computed by an idle wave during A
    __shared__ int s_stop;               // convergence decision of the last completed iteration, taken by an idle wave during A
    double2* const Gb = lds2;                    // G[k0][f1], k0 = 0..15 (rows beyond h are zero)
    double2* const Tb = lds2 + G_ELEMS;          // conj T[row(p)][q]: rows p = 0..15 and 48..63 (stored at p - 32)
    double2* const Qb = Tb + T_ELEMS;            // exchange scratch: column lines (B/C), aliased by the row lines (A, D)
    const int n = a.g.n[0], h = (n - 1) / 2, M = a.g.M;
    const int row = blockIdx.x;
    const int64_t base = (int64_t)row * M;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // row role (waves 0, 1): lane = k0 * 8 + j; column role (all waves): lane = jc * 8 + ql, packed column q = 8 wave + ql
    const int k0 = tid >> 3, j = tid & 7;
    const bool row_role = tid < 128;
    const int jc = lane >> 3, ql = lane & 7, q = wave * 8 + ql;
    double2* const xw = Qb + (k0 & 15) * LDR + 9 * j;
    const double2* const xr = Qb + (k0 & 15) * LDR + j;
    double2* const qw = Qb + (wave * 8 + ql) * LQ + 9 * jc;
    const double2* const qr = Qb + (wave * 8 + ql) * LQ + jc;

    double2 twr[7], twc[7];
#pragma unroll
    for (int t = 1; t < 8; ++t) {
        twr[t - 1] = a.g.tw[0][j * t];
        twc[t - 1] = a.g.tw[0][jc * t];
    }
    // real spectrum of the centred lags, halved (the unpacking of D averages two terms): vhat = w^((n-1)(f0+f1)) S
    double sa[8], sb[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int f0 = jc + 8 * t;
        const double2 va = a.vhat[f0 * F + q], vb = a.vhat[f0 * F + q + 32];
        const double2 wa = a.g.tw[0][((n - 1) * (f0 + q)) & 63], wb = a.g.tw[0][((n - 1) * (f0 + q + 32)) & 63];
        sa[t] = 0.5 * (va.x * wa.x + va.y * wa.y);
        sb[t] = -0.5 * (vb.x * wb.x + vb.y * wb.y);      // sign: conjugation in front of the inverse transform
    }
    const bool z6_ok = jc > 0;                            // row 16 - jc exists

    // vector slots of a row lane: positions j + 8 t, t in {0, 1, 6, 7} <-> k1 = j, j + 8, j - 16, j - 8
    double2 xv[KS], rv[KS], pv[KS];
    double wsr[KS], dg[KS], rdg[KS];     // ws is real here (checked below): ws * u costs two multiplies
    bool ok[KS];
    int idx[KS];
    const bool precond = a.diag != nullptr || a.diag_scale != nullptr;
    const double wgt = k0 == 0 ? 1.0 : 2.0;
    double ws_bad = 0.0;                 // > 0: ws is not real and even
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int k1 = s == 0 ? j : (s == 1 ? j + 8 : (s == 2 ? j - 16 : j - 8));
        ok[s] = row_role && k0 <= h && k1 <= h && -k1 <= h;
        idx[s] = ok[s] ? (k0 + h) * n + (k1 + h) : 0;
        double ws_im = 0.0;
        if (ok[s]) {
            xv[s] = a.zero_x0 ? make_double2(0.0, 0.0) : a.x0[base + idx[s]];
            const double2 w = a.ws[idx[s]], wm = a.ws[M - 1 - idx[s]];
            wsr[s] = w.x;
            ws_im = w.y * w.y + (w.x - wm.x) * (w.x - wm.x) + wm.y * wm.y;
            dg[s] = jacobi_entry(a, w, idx[s]);
        } else {
            xv[s] = make_double2(0.0, 0.0);
            wsr[s] = 0.0;
            dg[s] = 1.0;
        }
        ws_bad += ws_im;
        rdg[s] = 1.0 / dg[s];
        rv[s] = pv[s] = make_double2(0.0, 0.0);
    }

#ifdef EFGP_CG_STAMPS
    long long stamp_prev = (long long)__builtin_readcyclecounter();
#endif
    int stop = 0;
    double rcp_rz = 0.0;
    auto apply_A = [&](const double2 (&u)[KS], double2 (&Au)[KS]) __attribute__((always_inline)) {
        double2 v[8];
        EFGP_STAMP(7);
        // A: rows k0 >= 0, modes k1 wrapped to positions k1 mod 64 -> G[k0][f1]
        if (row_role) {
            dft8_in4(make_double2(wsr[0] * u[0].x, wsr[0] * u[0].y), make_double2(wsr[1] * u[1].x, wsr[1] * u[1].y),
                     make_double2(wsr[2] * u[2].x, wsr[2] * u[2].y), make_double2(wsr[3] * u[3].x, wsr[3] * u[3].y), v);
            exchange_stage2(v, xw, xr, twr);
            s64::store8_all<8>(Gb + k0 * LDR + j, v);
        }
        __syncthreads();
        stop = s_stop;
        rcp_rz = s_rcp;
        if (stop) return;                                 // uniform: the iteration that just started is abandoned
        EFGP_STAMP(0);
        // B: packed columns z[k0] = G[k0][q] + i G[k0][q + 32] (k0 >= 0), conj G[-k0][q] + i conj G[-k0][q + 32] (k0 < 0)
        {
            const double2* g0 = Gb + q;
            const double2 a0 = g0[jc * LDR], b0 = g0[jc * LDR + 32];
            const double2 a1 = g0[(jc + 8) * LDR], b1 = g0[(jc + 8) * LDR + 32];
            const double2 a6 = g0[((16 - jc) & 15) * LDR], b6 = g0[((16 - jc) & 15) * LDR + 32];
            const double2 a7 = g0[(8 - jc) * LDR], b7 = g0[(8 - jc) * LDR + 32];
            // jc = 0 reads row k0 = 0, whose modes +k1 and -k1 are BOTH stored: its line G[0][.] is real only up to the rounding
            // noise the iterates have collected in that row's anti-Hermitian part, and the packing would hand that noise to the
            // other column of the pair as if it were signal -- a non-symmetric perturbation of the operator that held the TRUE
            // residual of a cg_tol = 1e-12 solve at 2e-6 (round 3, c2 golden: beta 1.3e-6 off; the reference reaches 7e-10).
            // Dropping the imaginary parts is the orthogonal projection onto coefficient arrays of real functions.
            const double2 z0 = jc == 0 ? make_double2(a0.x, b0.x) : make_double2(a0.x - b0.y, a0.y + b0.x);
            const double2 z1 = make_double2(a1.x - b1.y, a1.y + b1.x);
            double2 z6 = make_double2(a6.x + b6.y, b6.x - a6.y);
            const double2 z7 = make_double2(a7.x + b7.y, b7.x - a7.y);
            if (!z6_ok) z6 = make_double2(0.0, 0.0);
            dft8_in4(z0, z1, z6, z7, v);
            exchange_stage2(v, qw, qr, twc);              // v[t] = R_q[f0] + i R_{q+32}[f0], f0 = jc + 8 t
            // C: times the real spectrum, conjugate, forward transform = conj of the inverse transform
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = make_double2(v[t].x * sa[t], v[t].y * sb[t]);
            dft_fwd<8>(v);
            exchange_stage2<true>(v, qw, qr, twc);        // v[t] = conj T[jc + 8 t], t in {0, 1, 6, 7}
            double2* t0 = Tb + jc * LT + q;
            t0[0] = v[0];
            t0[8 * LT] = v[1];
            t0[16 * LT] = v[6];
            t0[24 * LT] = v[7];
        }
        __syncthreads();
        EFGP_STAMP(1);
        // D: row k0 >= 0 of the result from the packed columns at +k0 and -k0; conjugated inputs, forward transform
        if (row_role) {
            const double2* tp = Tb + (k0 & 15) * LT + j;
            const double2* tm = Tb + ((k0 & 15) == 0 ? 0 : 32 - (k0 & 15)) * LT + j;
#pragma unroll
            for (int u4 = 0; u4 < 4; ++u4) {
                const double2 P = tp[8 * u4], Mv = tm[8 * u4];
                v[u4] = make_double2(P.x + Mv.x, P.y - Mv.y);
                v[u4 + 4] = make_double2(-P.y - Mv.y, P.x - Mv.x);
            }
            dft_fwd<8>(v);
            exchange_stage2<true>(v, xw, xr, twr);        // v[t] = conj Y[k0][j + 8 t], t in {0, 1, 6, 7}
            const double2 y[KS] = {s64::conjd(v[0]), s64::conjd(v[1]), s64::conjd(v[6]), s64::conjd(v[7])};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const double2 gg = make_double2(wsr[s] * y[s].x, wsr[s] * y[s].y);
                if (VARIANT == 0) Au[s] = make_double2(gg.x + a.sigmasq * u[s].x, gg.y + a.sigmasq * u[s].y);
                else Au[s] = make_double2(gg.x / a.sigmasq + u[s].x, gg.y / a.sigmasq + u[s].y);
            }
        } else {
#pragma unroll
            for (int s = 0; s < KS; ++s) Au[s] = make_double2(0.0, 0.0);
        }
        EFGP_STAMP(2);
    };

    if (tid == 0) {
        s_stop = 0;
        s_rcp = 0.0;
    }
    double2 Ap[KS];
    if (a.zero_x0) {
#pragma unroll
        for (int s = 0; s < KS; ++s) Ap[s] = make_double2(0.0, 0.0);
    } else {
        apply_A(xv, Ap);
    }
    double rz = 0.0, bb = 0.0, asym = 0.0;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        if (ok[s]) {
            double2 bv = a.b[base + idx[s]];
            double2 bm = a.b[base + (M - 1 - idx[s])];                 // mode -k: must be the conjugate
            if (a.b_times_ws) {
                bv = make_double2(wsr[s] * bv.x, wsr[s] * bv.y);
                bm = make_double2(wsr[s] * bm.x, wsr[s] * bm.y);
            }
            asym += (bv.x - bm.x) * (bv.x - bm.x) + (bv.y + bm.y) * (bv.y + bm.y);
            rv[s] = csub(bv, Ap[s]);
            pv[s] = precond ? make_double2(div_rcp(rv[s].x, dg[s], rdg[s]), div_rcp(rv[s].y, dg[s], rdg[s])) : rv[s];
            rz += rv[s].x * pv[s].x + rv[s].y * pv[s].y;
            bb += bv.x * bv.x + bv.y * bv.y;
        }
    }
    rz *= wgt;
    bb *= wgt;
    block_sum_pair_h(rz, bb, red);
    asym = block_sum_h(asym, red);
    ws_bad = block_sum_h(ws_bad, red + kWavesH);      // (disjoint scratch: no barrier between the two sums' reads and writes)
    if (!(asym <= 1e-16 * bb) || ws_bad != 0.0) {
        // not the coefficients of a real function (rounding leaves ~1e-32 |b|^2): refuse loudly instead of solving another system
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (ok[s]) {
                a.x[base + idx[s]] = make_double2(__builtin_nan(""), __builtin_nan(""));
                a.x[base + (M - 1 - idx[s])] = make_double2(__builtin_nan(""), __builtin_nan(""));
            }
        }
        if (tid == 0) a.iters[row] = -2;
        return;
    }
    const double bn = sqrt(bb);
    const double den = bn > 0.0 ? bn : 1.0;
    const double den_eps = den + 1e-16, rcp_den = 1.0 / den_eps;
    // Per iteration the row waves (0, 1) carry the dependent chain  A -> B/C -> D -> <p,Ap> -> alpha -> r -> <r,r>,<r,z> ->
    // beta -> p -> A;  everything that is not on it is done by wave 3, idle during A: the norm test of iteration i (sqrt and a
    // division) and the reciprocal for the next beta.  Its decision is read behind the first barrier of the next operator
    // application, whose A phase has then run speculatively (LDS only; x is final, p is not an output).
    if (wave == 3 && lane == 0) s_rcp = 1.0 / (rz + 1e-16);
    int it = 0;
    for (; it < a.max_iter;) {
        apply_A(pv, Ap);
        if (stop) break;
        double pAp = 0.0;
#pragma unroll
        for (int s = 0; s < KS; ++s) pAp += pv[s].x * Ap[s].x + pv[s].y * Ap[s].y;
        pAp = block_sum_h(pAp * wgt, red) + 1e-16;
        EFGP_STAMP(3);
        const double alpha = rz / pAp;
        double rr = 0.0, rzn = 0.0;
        double2 zv[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            xv[s].x += alpha * pv[s].x;
            xv[s].y += alpha * pv[s].y;
            rv[s].x -= alpha * Ap[s].x;
            rv[s].y -= alpha * Ap[s].y;
            zv[s] = precond ? make_double2(div_rcp(rv[s].x, dg[s], rdg[s]), div_rcp(rv[s].y, dg[s], rdg[s])) : rv[s];
            rr += rv[s].x * rv[s].x + rv[s].y * rv[s].y;
            rzn += rv[s].x * zv[s].x + rv[s].y * zv[s].y;
        }
        rr *= wgt;
        rzn *= wgt;
        EFGP_STAMP(4);
        block_sum_pair_h(rr, rzn, red);
        EFGP_STAMP(5);
        ++it;
        if (wave == 3) {
            const double ratio = div_rcp(sqrt(rr), den_eps, rcp_den);
            const bool conv = a.early_stop && ((ratio < a.tol) || (a.batched && sqrt(rr) < 1e-12));
            if (lane == 0) {
                if (a.hist && row == 0 && it <= a.hist_cap) a.hist[it - 1] = ratio;
                s_stop = conv ? 1 : 0;
                s_rcp = 1.0 / (rzn + 1e-16);
            }
        }
        // cg.py:132 / 229: the test sits before (single) or after (batched) this update; p is not an output
        const double beta = div_rcp(rzn, rz + 1e-16, rcp_rz);
#pragma unroll
        for (int s = 0; s < KS; ++s) pv[s] = make_double2(zv[s].x + beta * pv[s].x, zv[s].y + beta * pv[s].y);
        rz = rzn;
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        if (ok[s]) {
            a.x[base + idx[s]] = xv[s];
            if (k0 > 0) a.x[base + (M - 1 - idx[s])] = s64::conjd(xv[s]);     // mode -k
        }
    }
    if (tid == 0) a.iters[row] = it;
}

// ------------------------------------------------------------------------------------------------
// Hermitian solve on the SMALLEST circulant grid (round 4): 48 x 48 for blocks of up to 23 x 23 modes.
// The Toeplitz product is exact on any circulant grid F >= 2 n - 1 (wrap-around lands outside the crop window,
// efgpnd.py:1266-1271: the reference's own `_next_fast_fft_size` branch); next_pow2 is a choice.  The headline block
// (mtot 23, 2 n - 1 = 45) fits 48 = 3 * 16: 0.56 x the grid of cg_herm64_kernel -- 24 packed column lines instead of 32, 48-point
// lines instead of 64-point ones.  Same structure as cg_herm64_kernel (rows k0 >= 0 only, two real columns per complex column
// transform, CG vectors in the row lanes' registers, the norm test on an idle wave, refusal of non-conforming input), same
// recurrences and stopping rules (cg.py:86-153 / 155-244); results differ from the 64 x 64 embedding by rounding only.
// fft_shape / efgp_toeplitz_apply keep the reference's grid.
//
// A 48-point line lives in 8 adjacent lanes as 48 = 6 x 8, in two mirrored factorizations so that every pruned stage is a
// radix-8 butterfly with four live legs (dft8_in4 / dft8_out4 of the 64-point kernel) on SIX lanes holding positions j + 6 t,
// and every full stage a radix-6 butterfly on EIGHT lanes holding positions j + 8 t:
//   "6x8" (phases A, B: zero-padded input):  lanes j < 6: DFT-8 over t of x[j + 6 t] (t in {0, 1, 6, 7} live) -> exchange ->
//                                            lanes k < 8: twiddle w48^(k t), DFT-6 over t -> X[k + 8 k2], k2 < 6
//   "8x6" (phases C, D: cropped output):     lanes j < 8: DFT-6 over t of x[j + 8 t], twiddle w48^(j k) -> exchange ->
//                                            lanes k < 6: DFT-8 over the 8 writers -> X[k + 6 k2], k2 in {0, 1, 6, 7} kept
// The output layout of one is the input layout of the other: the CG vectors stay in the registers of the lanes j < 6 of a row
// (slots k1 = j, j + 6, j - 12, j - 6) between D and the next A, and the spectrum multiply between B and C works on registers.
// ------------------------------------------------------------------------------------------------
namespace h48 {
constexpr int kThreadsH = 256, kWavesH = kThreadsH / 64;
constexpr int F = 48, HF = 24, NR = 12;  // grid, packed column pairs, stored rows k0 = 0..11
constexpr int LDR = 56;                  // pitch of the G rows (8 mod 16)
constexpr int LT = 28;                   // pitch of the T rows (24 packed columns)
constexpr int LX = 72;                   // row lanes' exchange scratch per line: [writer][value], writer pitch 9
constexpr int LQ = 82;                   // column lanes' exchange scratch per line (2 mod 16)
constexpr int G_ELEMS = NR * LDR, T_ELEMS = 2 * NR * LT, Q_ELEMS = HF * LQ;
static_assert(NR * LX <= Q_ELEMS, "the row lines' scratch aliases the column lines' scratch");
constexpr int kLdsElems = G_ELEMS + T_ELEMS + Q_ELEMS;     // 53 KB

// forward DFT-6 in natural order: even / odd split into two DFT-3
__device__ __forceinline__ void dft3(double2 b0, double2 b1, double2 b2, double2& x0, double2& x1, double2& x2) {
    const double s = 0.86602540378443864676;
    const double2 sm = cadd(b1, b2), df = csub(b1, b2);
    x0 = cadd(b0, sm);
    const double2 m = make_double2(fma(-0.5, sm.x, b0.x), fma(-0.5, sm.y, b0.y));
    x1 = make_double2(fma(s, df.y, m.x), fma(-s, df.x, m.y));      // m - i s (b1 - b2)
    x2 = make_double2(fma(-s, df.y, m.x), fma(s, df.x, m.y));
}
__device__ __forceinline__ void dft6(double2 (&v)[6]) {
    const double s = 0.86602540378443864676;
    double2 e0, e1, e2, o0, o1, o2;
    dft3(v[0], v[2], v[4], e0, e1, e2);
    dft3(v[1], v[3], v[5], o0, o1, o2);
    const double2 w1 = make_double2(fma(s, o1.y, 0.5 * o1.x), fma(-s, o1.x, 0.5 * o1.y));        // w6   o1, w6 = (1/2, -s)
    const double2 w2 = make_double2(fma(s, o2.y, -0.5 * o2.x), fma(-s, o2.x, -0.5 * o2.y));      // w6^2 o2
    v[0] = cadd(e0, o0);
    v[3] = csub(e0, o0);
    v[1] = cadd(e1, w1);
    v[4] = csub(e1, w1);
    v[2] = cadd(e2, w2);
    v[5] = csub(e2, w2);
}
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// "6x8", second half: the lanes j < 6 of a line hand their eight first-stage outputs over (scratch [writer][value], pitch 9);
// every lane k < 8 takes value k of the six writers, twiddles with tw[t - 1] = w48^(k t) and runs the radix-6 stage:
// u[k2] = X[k + 8 k2].
__device__ __forceinline__ void exchange_8to6(const double2 (&v)[8], double2 (&u)[6], bool writer, double2* __restrict__ wr,
                                              const double2* __restrict__ rd, const double2 (&tw)[5]) {
    wave_sync();
    if (writer) s64::store8_all<1>(wr, v);
    wave_sync();
#pragma unroll
    for (int t = 0; t < 6; ++t) u[t] = rd[9 * t];
#pragma unroll
    for (int t = 1; t < 6; ++t) u[t] = cmulp(u[t], tw[t - 1]);
    dft6(u);
}
// "8x6", second half: every lane j < 8 has run the radix-6 stage on u and twiddles its outputs with tw[k - 1] = w48^(j k); the
// lanes k < 6 take value k of the eight writers and run the radix-8 stage of which the outputs 0, 1, 6, 7 are wanted:
// v[k2] = X[k + 6 k2].
__device__ __forceinline__ void exchange_6to8(double2 (&u)[6], double2 (&v)[8], double2* __restrict__ wr, const double2* __restrict__ rd,
                                              const double2 (&tw)[5]) {
#pragma unroll
    for (int t = 1; t < 6; ++t) u[t] = cmulp(u[t], tw[t - 1]);
    wave_sync();
#pragma unroll
    for (int t = 0; t < 6; ++t) wr[t] = u[t];
    wave_sync();
    s64::load8_all<9>(rd, v);
    h64::dft8_out4(v);
}
}  // namespace h48

template <int VARIANT>
__global__ __launch_bounds__(h48::kThreadsH) void cg_herm48_kernel(Args a) {
    using namespace h48;
    using h64::block_sum_h;
    using h64::block_sum_pair_h;
    using h64::dft8_in4;
    using h64::div_rcp;
    constexpr int KS = 4;
    static_assert(h64::kWavesH == kWavesH, "the block reductions are shared with the 64 x 64 kernel");
    extern __shared__ double2 lds2[];
    __shared__ double red[4 * kWavesH];
    __shared__ double s_rcp;             // 1 / (<r,z> + 1e-16) for the next beta, computed by an idle wave during A
    __shared__ int s_stop;               // convergence decision of the last completed iteration, taken by an idle wave during A
    double2* const Gb = lds2;                    // G[k0][f1], k0 = 0..11 (rows beyond h are zero), f1 = 0..47
    double2* const Tb = lds2 + G_ELEMS;          // conj T[row(p)][q]: rows p = 0..11 and 36..47 (stored at p - 24)
    double2* const Qb = Tb + T_ELEMS;            // exchange scratch: column lines (B/C), aliased by the row lines (A, D)
    const int n = a.g.n[0], h = (n - 1) / 2, M = a.g.M;
    const int row = blockIdx.x;
    const int64_t base = (int64_t)row * M;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // row role (12 lines x 8 lanes: wave 0 and half of wave 1): lane = k0 * 8 + j, the vector lives in the lanes j < 6;
    // column role (waves 0..2): lane = jc * 8 + ql, packed column q = 8 wave + ql < 24
    const int k0 = tid >> 3, j = tid & 7;
    const bool row_role = tid < 8 * NR, vec_role = row_role && j < 6;
    const int jc = lane >> 3, ql = lane & 7, q = wave * 8 + ql;
    const bool col_role = wave < 3, col6 = jc < 6;
    const int k0s = row_role ? k0 : 0;
    double2* const xw = Qb + k0s * LX + 9 * j;
    const double2* const xr = Qb + k0s * LX + j;
    const int qs = col_role ? q : 0;
    double2* const qw = Qb + qs * LQ + 9 * jc;
    const double2* const qr = Qb + qs * LQ + jc;

    double2 twr[5], twc[5];              // w48^(lane-in-line * t), t = 1..5 (35 at most: no wrap)
#pragma unroll
    for (int t = 1; t < 6; ++t) {
        twr[t - 1] = a.g.tw[0][j * t];
        twc[t - 1] = a.g.tw[0][jc * t];
    }
    // real spectrum of the centred lags, halved (the unpacking of D averages two terms): vhat = w^((n-1)(f0+f1)) S
    double sa[6], sb[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const int f0 = jc + 8 * t;
        const double2 va = a.vhat[f0 * F + qs], vb = a.vhat[f0 * F + qs + HF];
        const double2 wa = a.g.tw[0][((n - 1) * (f0 + qs)) % F], wb = a.g.tw[0][((n - 1) * (f0 + qs + HF)) % F];
        sa[t] = 0.5 * (va.x * wa.x + va.y * wa.y);
        sb[t] = -0.5 * (vb.x * wb.x + vb.y * wb.y);      // sign: conjugation in front of the inverse transform
    }
    const bool z6_ok = jc > 0;                            // row 12 - jc exists
    const int jr = col6 ? jc : 0;                         // rows this lane packs in B (the lanes jc >= 6 only take part in the radix-6 stages)

    // vector slots of a row lane j < 6: positions j + 6 t, t in {0, 1, 6, 7} <-> k1 = j, j + 6, j - 12, j - 6
    double2 xv[KS], rv[KS], pv[KS];
    double wsr[KS], dg[KS], rdg[KS];     // ws is real here (checked below): ws * u costs two multiplies
    bool ok[KS];
    int idx[KS];
    const bool precond = a.diag != nullptr || a.diag_scale != nullptr;
    const double wgt = k0 == 0 ? 1.0 : 2.0;
    double ws_bad = 0.0;                 // > 0: ws is not real and even
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int k1 = s == 0 ? j : (s == 1 ? j + 6 : (s == 2 ? j - 12 : j - 6));
        ok[s] = vec_role && k0 <= h && k1 <= h && -k1 <= h;
        idx[s] = ok[s] ? (k0 + h) * n + (k1 + h) : 0;
        double ws_im = 0.0;
        if (ok[s]) {
            xv[s] = a.zero_x0 ? make_double2(0.0, 0.0) : a.x0[base + idx[s]];
            const double2 w = a.ws[idx[s]], wm = a.ws[M - 1 - idx[s]];
            wsr[s] = w.x;
            ws_im = w.y * w.y + (w.x - wm.x) * (w.x - wm.x) + wm.y * wm.y;
            dg[s] = jacobi_entry(a, w, idx[s]);
        } else {
            xv[s] = make_double2(0.0, 0.0);
            wsr[s] = 0.0;
            dg[s] = 1.0;
        }
        ws_bad += ws_im;
        rdg[s] = 1.0 / dg[s];
        rv[s] = pv[s] = make_double2(0.0, 0.0);
    }

#ifdef EFGP_CG_STAMPS
    long long stamp_prev = (long long)__builtin_readcyclecounter();
#endif
    int stop = 0;
    double rcp_rz = 0.0;
    auto apply_A = [&](const double2 (&u)[KS], double2 (&Au)[KS]) __attribute__((always_inline)) {
        double2 v[8], u6[6];
        EFGP_STAMP(7);
        // A ("6x8"): rows k0 >= 0, modes k1 wrapped to positions k1 mod 48 -> G[k0][f1]
        if (row_role) {
            dft8_in4(make_double2(wsr[0] * u[0].x, wsr[0] * u[0].y), make_double2(wsr[1] * u[1].x, wsr[1] * u[1].y),
                     make_double2(wsr[2] * u[2].x, wsr[2] * u[2].y), make_double2(wsr[3] * u[3].x, wsr[3] * u[3].y), v);
            exchange_8to6(v, u6, j < 6, xw, xr, twr);     // u6[k2] = G[k0][j + 8 k2]
            double2* g0 = Gb + k0 * LDR + j;
#pragma unroll
            for (int t = 0; t < 6; ++t) g0[8 * t] = u6[t];
        }
        __syncthreads();
        stop = s_stop;
        rcp_rz = s_rcp;
        if (stop) return;                                 // uniform: the iteration that just started is abandoned
        EFGP_STAMP(0);
        // B ("6x8"): packed columns z[k0] = G[k0][q] + i G[k0][q + 24] (k0 >= 0), conj G[-k0][q] + i conj G[-k0][q + 24] (k0 < 0),
        // rows of lane jc < 6: k0 = jc, jc + 6, jc - 12, jc - 6
        if (col_role) {
            const double2* g0 = Gb + q;
            const double2 a0 = g0[jr * LDR], b0 = g0[jr * LDR + HF];
            const double2 a1 = g0[(jr + 6) * LDR], b1 = g0[(jr + 6) * LDR + HF];
            const double2 a6 = g0[((12 - jr) % 12) * LDR], b6 = g0[((12 - jr) % 12) * LDR + HF];
            const double2 a7 = g0[(6 - jr) * LDR], b7 = g0[(6 - jr) * LDR + HF];
            // jc = 0 reads row k0 = 0, whose modes +k1 and -k1 are BOTH stored: drop the imaginary parts (the projection onto
            // coefficient arrays of real functions; see cg_herm64_kernel)
            const double2 z0 = jc == 0 ? make_double2(a0.x, b0.x) : make_double2(a0.x - b0.y, a0.y + b0.x);
            const double2 z1 = make_double2(a1.x - b1.y, a1.y + b1.x);
            double2 z6 = make_double2(a6.x + b6.y, b6.x - a6.y);
            const double2 z7 = make_double2(a7.x + b7.y, b7.x - a7.y);
            if (!z6_ok) z6 = make_double2(0.0, 0.0);
            dft8_in4(z0, z1, z6, z7, v);
            exchange_8to6(v, u6, col6, qw, qr, twc);      // u6[t] = R_q[f0] + i R_{q+24}[f0], f0 = jc + 8 t
            // C ("8x6"): times the real spectrum, conjugate, forward transform = conj of the inverse transform
#pragma unroll
            for (int t = 0; t < 6; ++t) u6[t] = make_double2(u6[t].x * sa[t], u6[t].y * sb[t]);
            dft6(u6);
            exchange_6to8(u6, v, qw, qr, twc);            // lanes jc < 6: v[k2] = conj T[jc + 6 k2], k2 in {0, 1, 6, 7}
            if (col6) {
                double2* t0 = Tb + jc * LT + q;
                t0[0] = v[0];
                t0[6 * LT] = v[1];
                t0[12 * LT] = v[6];                       // p = jc + 36, stored at p - 24
                t0[18 * LT] = v[7];
            }
        }
        __syncthreads();
        EFGP_STAMP(1);
        // D ("8x6"): row k0 >= 0 of the result from the packed columns at +k0 and -k0; conjugated inputs, forward transform
        if (row_role) {
            const double2* tp = Tb + k0 * LT + j;
            const double2* tm = Tb + (k0 == 0 ? 0 : 2 * NR - k0) * LT + j;
#pragma unroll
            for (int u3 = 0; u3 < 3; ++u3) {
                const double2 P = tp[8 * u3], Mv = tm[8 * u3];
                u6[u3] = make_double2(P.x + Mv.x, P.y - Mv.y);
                u6[u3 + 3] = make_double2(-P.y - Mv.y, P.x - Mv.x);
            }
            dft6(u6);
            exchange_6to8(u6, v, xw, xr, twr);            // lanes j < 6: v[k2] = conj Y[k0][j + 6 k2], k2 in {0, 1, 6, 7}
            const double2 y[KS] = {s64::conjd(v[0]), s64::conjd(v[1]), s64::conjd(v[6]), s64::conjd(v[7])};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const double2 gg = make_double2(wsr[s] * y[s].x, wsr[s] * y[s].y);
                double2 val;
                if (VARIANT == 0) val = make_double2(gg.x + a.sigmasq * u[s].x, gg.y + a.sigmasq * u[s].y);
                else val = make_double2(gg.x / a.sigmasq + u[s].x, gg.y / a.sigmasq + u[s].y);
                // the lanes j >= 6 of a line only lend their radix-6 stages: what they read in the radix-8 stage is stale scratch
                Au[s] = ok[s] ? val : make_double2(0.0, 0.0);
            }
        } else {
#pragma unroll
            for (int s = 0; s < KS; ++s) Au[s] = make_double2(0.0, 0.0);
        }
        EFGP_STAMP(2);
    };

    if (tid == 0) {
        s_stop = 0;
        s_rcp = 0.0;
    }
    double2 Ap[KS];
    if (a.zero_x0) {
#pragma unroll
        for (int s = 0; s < KS; ++s) Ap[s] = make_double2(0.0, 0.0);
    } else {
        apply_A(xv, Ap);
    }
    double rz = 0.0, bb = 0.0, asym = 0.0;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        if (ok[s]) {
            double2 bv = a.b[base + idx[s]];
            double2 bm = a.b[base + (M - 1 - idx[s])];                 // mode -k: must be the conjugate
            if (a.b_times_ws) {
                bv = make_double2(wsr[s] * bv.x, wsr[s] * bv.y);
                bm = make_double2(wsr[s] * bm.x, wsr[s] * bm.y);
            }
            asym += (bv.x - bm.x) * (bv.x - bm.x) + (bv.y + bm.y) * (bv.y + bm.y);
            rv[s] = csub(bv, Ap[s]);
            pv[s] = precond ? make_double2(div_rcp(rv[s].x, dg[s], rdg[s]), div_rcp(rv[s].y, dg[s], rdg[s])) : rv[s];
            rz += rv[s].x * pv[s].x + rv[s].y * pv[s].y;
            bb += bv.x * bv.x + bv.y * bv.y;
        }
    }
    rz *= wgt;
    bb *= wgt;
    block_sum_pair_h(rz, bb, red);
    asym = block_sum_h(asym, red);
    ws_bad = block_sum_h(ws_bad, red + kWavesH);      // (disjoint scratch: no barrier between the two sums' reads and writes)
    if (!(asym <= 1e-16 * bb) || ws_bad != 0.0) {
        // not the coefficients of a real function (rounding leaves ~1e-32 |b|^2): refuse loudly instead of solving another system
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (ok[s]) {
                a.x[base + idx[s]] = make_double2(__builtin_nan(""), __builtin_nan(""));
                a.x[base + (M - 1 - idx[s])] = make_double2(__builtin_nan(""), __builtin_nan(""));
            }
        }
        if (tid == 0) a.iters[row] = -2;
        return;
    }
    const double bn = sqrt(bb);
    const double den = bn > 0.0 ? bn : 1.0;
    const double den_eps = den + 1e-16, rcp_den = 1.0 / den_eps;
    // as in cg_herm64_kernel: the row waves carry the dependent chain, wave 3 (idle during A) takes the norm test of iteration i
    // and the reciprocal for the next beta; its decision is read behind the first barrier of the next operator application
    if (wave == 3 && lane == 0) s_rcp = 1.0 / (rz + 1e-16);
    int it = 0;
    for (; it < a.max_iter;) {
        apply_A(pv, Ap);
        if (stop) break;
        double pAp = 0.0;
#pragma unroll
        for (int s = 0; s < KS; ++s) pAp += pv[s].x * Ap[s].x + pv[s].y * Ap[s].y;
        pAp = block_sum_h(pAp * wgt, red) + 1e-16;
        EFGP_STAMP(3);
        const double alpha = rz / pAp;
        double rr = 0.0, rzn = 0.0;
        double2 zv[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            xv[s].x += alpha * pv[s].x;
            xv[s].y += alpha * pv[s].y;
            rv[s].x -= alpha * Ap[s].x;
            rv[s].y -= alpha * Ap[s].y;
            zv[s] = precond ? make_double2(div_rcp(rv[s].x, dg[s], rdg[s]), div_rcp(rv[s].y, dg[s], rdg[s])) : rv[s];
            rr += rv[s].x * rv[s].x + rv[s].y * rv[s].y;
            rzn += rv[s].x * zv[s].x + rv[s].y * zv[s].y;
        }
        rr *= wgt;
        rzn *= wgt;
        EFGP_STAMP(4);
        block_sum_pair_h(rr, rzn, red);
        EFGP_STAMP(5);
        ++it;
        if (wave == 3) {
            const double ratio = div_rcp(sqrt(rr), den_eps, rcp_den);
            const bool conv = a.early_stop && ((ratio < a.tol) || (a.batched && sqrt(rr) < 1e-12));
            if (lane == 0) {
                if (a.hist && row == 0 && it <= a.hist_cap) a.hist[it - 1] = ratio;
                s_stop = conv ? 1 : 0;
                s_rcp = 1.0 / (rzn + 1e-16);
            }
        }
        // cg.py:132 / 229: the test sits before (single) or after (batched) this update; p is not an output
        const double beta = div_rcp(rzn, rz + 1e-16, rcp_rz);
#pragma unroll
        for (int s = 0; s < KS; ++s) pv[s] = make_double2(zv[s].x + beta * pv[s].x, zv[s].y + beta * pv[s].y);
        rz = rzn;
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        if (ok[s]) {
            a.x[base + idx[s]] = xv[s];
            if (k0 > 0) a.x[base + (M - 1 - idx[s])] = s64::conjd(xv[s]);     // mode -k
        }
    }
    if (tid == 0) a.iters[row] = it;
}

// ------------------------------------------------------------------------------------------------
// 1-D systems (round 3): the whole solve in ONE WAVE per system.  cg_persistent_kernel above spends ~8 us per iteration on a
// 1-D block of 63 unknowns (BASELINE configs[0] is such a model: mtot 35, F = 128): its passes are built for grids -- 1024
// threads, a workgroup barrier per radix stage, two-level block reductions -- and a single 128-point line leaves all but a few
// lanes idle at every barrier.  Here a system is one wave: the vectors in registers (<= 4 entries per lane: n <= 255, F <= 512),
// the line transform a radix-4 / radix-2 Stockham ping-pong in LDS with wave-level synchronisation only, the spectrum and the
// twiddles resident in LDS, dot products by wave reductions.  Same operator (zero-padded input at [0, n), product read at
// [n - 1, 2 n - 1)), recurrences and stopping rules as cg_persistent_kernel (cg.py:86-244).
// ------------------------------------------------------------------------------------------------
namespace l1d {
constexpr int KS = 4;
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// forward transform of the F-point line in `x` (F = 2^lg), scratch `y`; twiddles tw[q] = exp(-2 pi i q / F); returns the buffer
// that holds the result (natural order).  All 64 lanes of the wave take part.
__device__ __forceinline__ double2* wave_fft(double2* x, double2* y, int F, int lg, const double2* __restrict__ tw) {
    const int lane = threadIdx.x;
    int Ns = 1, rem = lg;
    while (rem >= 2) {
        const int nb = F >> 2, tstep = F / (Ns * 4);
        for (int j = lane; j < nb; j += 64) {
            const int k = j & (Ns - 1), j0 = ((j - k) << 2) + k;
            double2 v0 = x[j], v1 = x[j + nb], v2 = x[j + 2 * nb], v3 = x[j + 3 * nb];
            if (k != 0) {
                v1 = cmulp(v1, tw[k * tstep]);
                v2 = cmulp(v2, tw[2 * k * tstep]);
                v3 = cmulp(v3, tw[3 * k * tstep]);
            }
            const double2 t0 = cadd(v0, v2), t1 = csub(v0, v2), t2 = cadd(v1, v3), d13 = csub(v1, v3);
            const double2 t3 = make_double2(d13.y, -d13.x);              // (v1 - v3) * (-i)
            y[j0] = cadd(t0, t2);
            y[j0 + Ns] = cadd(t1, t3);
            y[j0 + 2 * Ns] = csub(t0, t2);
            y[j0 + 3 * Ns] = csub(t1, t3);
        }
        wave_sync();
        double2* t = x;
        x = y;
        y = t;
        Ns <<= 2;
        rem -= 2;
    }
    if (rem == 1) {
        const int nb = F >> 1;                                           // Ns = F / 2: tstep = 1
        for (int j = lane; j < nb; j += 64) {
            const int k = j & (Ns - 1), j0 = ((j - k) << 1) + k;
            const double2 v0 = x[j];
            double2 v1 = x[j + nb];
            if (k != 0) v1 = cmulp(v1, tw[k]);
            y[j0] = cadd(v0, v1);
            y[j0 + Ns] = csub(v0, v1);
        }
        wave_sync();
        double2* t = x;
        x = y;
        y = t;
    }
    return x;
}
}  // namespace l1d

__global__ __launch_bounds__(64) void cg_line1d_kernel(Args a) {
    using namespace l1d;
    extern __shared__ double2 lds2[];
    const int F = a.g.F[0], n = a.g.n[0], M = a.g.M;
    const int lg = 31 - __clz(F);
    double2* const X = lds2;
    double2* const Y = lds2 + F;
    double2* const TW = Y + F;
    double2* const VH = TW + F;
    const int lane = threadIdx.x, row = blockIdx.x;
    const int64_t base = (int64_t)row * M;
    for (int q = lane; q < F; q += 64) {
        TW[q] = a.g.tw[0][q];
        VH[q] = a.vhat[q];
    }
    double2 xv[KS], rv[KS], pv[KS], wsv[KS];
    double dg[KS];
    const bool precond = a.diag != nullptr || a.diag_scale != nullptr;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int t = lane + 64 * s;
        if (t < M) {
            xv[s] = a.zero_x0 ? make_double2(0.0, 0.0) : a.x0[base + t];
            wsv[s] = a.ws[t];
            dg[s] = jacobi_entry(a, wsv[s], t);
        } else {
            xv[s] = wsv[s] = make_double2(0.0, 0.0);
            dg[s] = 1.0;
        }
        rv[s] = pv[s] = make_double2(0.0, 0.0);
    }
    wave_sync();
    auto apply_A = [&](const double2 (&u)[KS], double2 (&Au)[KS]) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int t = lane + 64 * s;
            if (t < M) X[t] = cmulp(wsv[s], u[s]);
        }
        for (int q = n + lane; q < F; q += 64) X[q] = make_double2(0.0, 0.0);
        wave_sync();
        double2* R = wave_fft(X, Y, F, lg, TW);
        double2* O = R == X ? Y : X;
        for (int q = lane; q < F; q += 64) {
            const double2 m = cmulp(R[q], VH[q]);
            R[q] = make_double2(m.x, -m.y);                               // conjugate: the inverse transform by a forward one
        }
        wave_sync();
        const double2* Z = wave_fft(R, O, F, lg, TW);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int t = lane + 64 * s;
            if (t < M) {
                const double2 z = Z[n - 1 + t];
                const double2 gg = cmulp(wsv[s], make_double2(z.x, -z.y));
                if (a.variant == 0) Au[s] = make_double2(gg.x + a.sigmasq * u[s].x, gg.y + a.sigmasq * u[s].y);
                else Au[s] = make_double2(gg.x / a.sigmasq + u[s].x, gg.y / a.sigmasq + u[s].y);
            } else {
                Au[s] = make_double2(0.0, 0.0);
            }
        }
        wave_sync();                                                      // the buffers are rewritten by the next application
    };
    double2 Ap[KS];
    if (a.zero_x0) {
#pragma unroll
        for (int s = 0; s < KS; ++s) Ap[s] = make_double2(0.0, 0.0);
    } else {
        apply_A(xv, Ap);
    }
    double rz = 0.0, bb = 0.0;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int t = lane + 64 * s;
        if (t < M) {
            double2 bv = a.b[base + t];
            if (a.b_times_ws) bv = cmulp(wsv[s], bv);
            rv[s] = csub(bv, Ap[s]);
            pv[s] = precond ? make_double2(rv[s].x / dg[s], rv[s].y / dg[s]) : rv[s];
            rz += rv[s].x * pv[s].x + rv[s].y * pv[s].y;
            bb += bv.x * bv.x + bv.y * bv.y;
        }
    }
    rz = wave_sum(rz);
    bb = wave_sum(bb);
    const double bn = sqrt(bb);
    const double den = bn > 0.0 ? bn : 1.0;
    int it = 0;
    for (; it < a.max_iter;) {
        apply_A(pv, Ap);
        double pAp = 0.0;
#pragma unroll
        for (int s = 0; s < KS; ++s) pAp += pv[s].x * Ap[s].x + pv[s].y * Ap[s].y;
        pAp = wave_sum(pAp) + 1e-16;
        const double alpha = rz / pAp;
        double rr = 0.0, rzn = 0.0;
        double2 zv[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            xv[s].x += alpha * pv[s].x;
            xv[s].y += alpha * pv[s].y;
            rv[s].x -= alpha * Ap[s].x;
            rv[s].y -= alpha * Ap[s].y;
            zv[s] = precond ? make_double2(rv[s].x / dg[s], rv[s].y / dg[s]) : rv[s];
            rr += rv[s].x * rv[s].x + rv[s].y * rv[s].y;
            rzn += rv[s].x * zv[s].x + rv[s].y * zv[s].y;
        }
        rr = wave_sum(rr);
        rzn = wave_sum(rzn);
        ++it;
        const double rnorm = sqrt(rr);
        if (a.hist && row == 0 && lane == 0 && it <= a.hist_cap) a.hist[it - 1] = rnorm / (den + 1e-16);
        const bool conv = a.early_stop && ((rnorm / (den + 1e-16) < a.tol) || (a.batched && rnorm < 1e-12));
        if (!a.batched && conv) break;                  // cg.py:132 (before the preconditioner / p update)
        const double beta = rzn / (rz + 1e-16);
#pragma unroll
        for (int s = 0; s < KS; ++s) pv[s] = make_double2(zv[s].x + beta * pv[s].x, zv[s].y + beta * pv[s].y);
        rz = rzn;
        if (conv) break;                                // cg.py:229-241 (after the p update)
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int t = lane + 64 * s;
        if (t < M) a.x[base + t] = xv[s];
    }
    if (lane == 0) a.iters[row] = it;
}

// Spectrum of the Toeplitz vector for the 64 x 64 circulant grid in ONE launch (efgpnd.py:1283-1290: pad to the FFT
// box, forward fftn): `factor * v` zero-padded into LDS, four radix-8 Stockham stages as in the solver above, result
// in natural order.  Replaces pad_scale_kernel + two rocFFT launches (~14 us of dependent 4-5 us launches per fit).
__device__ __forceinline__ void toeplitz_vhat_2d64_body(const double2* __restrict__ v, int L0, int L1, double factor,
                                                        double2* __restrict__ vhat, double2* lds2) {
    using namespace s64;
    double2* const bufA = lds2;
    double2* const bufB = lds2 + BUF;
    const int tid = threadIdx.x;
    for (int t = tid; t < F * F; t += kThreads) {
        const int i0 = t >> 6, i1 = t & 63;
        double2 x = make_double2(0.0, 0.0);
        if (i0 < L0 && i1 < L1) {
            x = v[i0 * L1 + i1];
            x.x *= factor;
            x.y *= factor;
        }
        bufA[i0 * LD + i1] = x;
    }
    const int c = tid & 63, j = tid >> 6;      // line (row, then column) and butterfly within the line
    double2 tw[7];
#pragma unroll
    for (int t = 1; t < 8; ++t) {
        double sn, cs;
        sincospi(-(double)(j * t) / 32.0, &sn, &cs);      // exp(-2 pi i j t / 64)
        tw[t - 1] = make_double2(cs, sn);
    }
    __syncthreads();
    double2 x[8];
    // dimension 1 (contiguous), all 64 rows: lane = row
    load8_all<8>(bufA + c * LD + j, x);
    dft_fwd<8>(x);
    store8_all<1>(bufB + c * LD + j * 8, x);
    __syncthreads();
    load8_all<8>(bufB + c * LD + j, x);
    twiddle8(x, tw);
    dft_fwd<8>(x);
    store8_all<8>(bufA + c * LD + j, x);
    __syncthreads();
    // dimension 0: lane = column
    load8_all<8 * LD>(bufA + j * LD + c, x);
    dft_fwd<8>(x);
    store8_all<LD>(bufB + j * 8 * LD + c, x);
    __syncthreads();
    load8_all<8 * LD>(bufB + j * LD + c, x);
    twiddle8(x, tw);
    dft_fwd<8>(x);
#pragma unroll
    for (int t = 0; t < 8; ++t) vhat[(j + 8 * t) * F + c] = x[t];
}
__global__ __launch_bounds__(kThreads) void toeplitz_vhat_2d64_kernel(const double2* __restrict__ v, int L0, int L1,
                                                                      double factor, double2* __restrict__ vhat) {
    extern __shared__ double2 lds2[];
    toeplitz_vhat_2d64_body(v, L0, L1, factor, vhat, lds2);
}

// The same for the 48 x 48 circulant grid of cg_herm48_kernel (round 4): vhat[f0][f1] = factor * sum_l v[l] w48^(f0 l0 + f1 l1).
// Lines as in that kernel: 8 adjacent lanes, "8x6" (radix 6 on eight lanes, radix 8 on six), rows then columns, 64 line slots.
__device__ __forceinline__ void toeplitz_vhat_2d48_body(const double2* __restrict__ v, int L0, int L1, double factor,
                                                        double2* __restrict__ vhat, double2* lds2) {
    constexpr int F = 48, LD = 49, LX = 72;
    double2* const buf = lds2;                       // [48][LD]
    double2* const scr = lds2 + F * LD;              // [64 lines][writer 9 j + value]
    const int tid = threadIdx.x;
    for (int t = tid; t < F * F; t += kThreads) {
        const int i0 = t / F, i1 = t - i0 * F;
        double2 x = make_double2(0.0, 0.0);
        if (i0 < L0 && i1 < L1) {
            x = v[i0 * L1 + i1];
            x.x *= factor;
            x.y *= factor;
        }
        buf[i0 * LD + i1] = x;
    }
    const int line = tid >> 3, j = tid & 7;
    const bool act = line < F;
    double2 tw[5];
#pragma unroll
    for (int t = 1; t < 6; ++t) {
        double sn, cs;
        sincospi(-(double)(j * t) / 24.0, &sn, &cs);      // exp(-2 pi i j t / 48)
        tw[t - 1] = make_double2(cs, sn);
    }
    double2* const wr = scr + line * LX + 9 * j;
    const double2* const rd = scr + line * LX + j;
    double2 u6[6], x8[8];
    __syncthreads();
    // dimension 1 (contiguous): line = row
#pragma unroll
    for (int t = 0; t < 6; ++t) u6[t] = act ? buf[line * LD + j + 8 * t] : make_double2(0.0, 0.0);
    h48::dft6(u6);
#pragma unroll
    for (int t = 1; t < 6; ++t) u6[t] = cmulp(u6[t], tw[t - 1]);
    h48::wave_sync();
#pragma unroll
    for (int t = 0; t < 6; ++t) wr[t] = u6[t];
    h48::wave_sync();
    s64::load8_all<9>(rd, x8);
    dft_fwd<8>(x8);
    if (act && j < 6) {
#pragma unroll
        for (int t = 0; t < 8; ++t) buf[line * LD + j + 6 * t] = x8[t];
    }
    __syncthreads();
    // dimension 0: line = column
#pragma unroll
    for (int t = 0; t < 6; ++t) u6[t] = act ? buf[(j + 8 * t) * LD + line] : make_double2(0.0, 0.0);
    h48::dft6(u6);
#pragma unroll
    for (int t = 1; t < 6; ++t) u6[t] = cmulp(u6[t], tw[t - 1]);
    h48::wave_sync();
#pragma unroll
    for (int t = 0; t < 6; ++t) wr[t] = u6[t];
    h48::wave_sync();
    s64::load8_all<9>(rd, x8);
    dft_fwd<8>(x8);
    if (act && j < 6) {
#pragma unroll
        for (int t = 0; t < 8; ++t) vhat[(j + 6 * t) * F + line] = x8[t];
    }
}
// both spectra of an operator in ONE launch (two workgroups on two CUs): a dependent launch costs more than either transform
__global__ __launch_bounds__(kThreads) void toeplitz_vhat_pair_kernel(const double2* __restrict__ v, int L0, int L1, double factor64,
                                                                      double2* __restrict__ vhat64, double factor48,
                                                                      double2* __restrict__ vhat48) {
    extern __shared__ double2 lds2[];
    if (vhat64 != nullptr && blockIdx.x == 0) toeplitz_vhat_2d64_body(v, L0, L1, factor64, vhat64, lds2);
    else toeplitz_vhat_2d48_body(v, L0, L1, factor48, vhat48, lds2);
}

// Batched variant for the lag-sum correlation of the stochastic variance (variance_ops.hip): transform b reads the
// L0 x L1 array src + b * src_stride (complex, or real doubles when REAL), zero-pads it to 64 x 64 and writes the forward
// transform in natural order to dst + b * dst_stride.  One workgroup per transform, no rocFFT (no run-time compilation).
template <bool REAL>
__global__ __launch_bounds__(kThreads) void fft2d64_batch_kernel(const void* __restrict__ src, int64_t src_stride, int L0, int L1,
                                                                 double2* __restrict__ dst, int64_t dst_stride) {
    using namespace s64;
    extern __shared__ double2 lds2[];
    double2* const bufA = lds2;
    double2* const bufB = lds2 + BUF;
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.x;
    for (int t = tid; t < F * F; t += kThreads) {
        const int i0 = t >> 6, i1 = t & 63;
        double2 x = make_double2(0.0, 0.0);
        if (i0 < L0 && i1 < L1) {
            if (REAL) x.x = ((const double*)src)[b * src_stride + i0 * L1 + i1];
            else x = ((const double2*)src)[b * src_stride + i0 * L1 + i1];
        }
        bufA[i0 * LD + i1] = x;
    }
    const int c = tid & 63, j = tid >> 6;
    double2 tw[7];
#pragma unroll
    for (int t = 1; t < 8; ++t) {
        double sn, cs;
        sincospi(-(double)(j * t) / 32.0, &sn, &cs);
        tw[t - 1] = make_double2(cs, sn);
    }
    __syncthreads();
    double2 x[8];
    load8_all<8>(bufA + c * LD + j, x);
    dft_fwd<8>(x);
    store8_all<1>(bufB + c * LD + j * 8, x);
    __syncthreads();
    load8_all<8>(bufB + c * LD + j, x);
    twiddle8(x, tw);
    dft_fwd<8>(x);
    store8_all<8>(bufA + c * LD + j, x);
    __syncthreads();
    load8_all<8 * LD>(bufA + j * LD + c, x);
    dft_fwd<8>(x);
    store8_all<LD>(bufB + j * 8 * LD + c, x);
    __syncthreads();
    load8_all<8 * LD>(bufB + j * LD + c, x);
    twiddle8(x, tw);
    dft_fwd<8>(x);
    double2* out = dst + b * dst_stride;
#pragma unroll
    for (int t = 0; t < 8; ++t) out[(j + 8 * t) * F + c] = x[t];
}

}  // namespace pcg

int fft2d64_batch_launch(const void* src, int src_is_real, int64_t src_stride, int L0, int L1, double2* dst, int64_t dst_stride,
                         int nbatch, hipStream_t stream) {
    using namespace pcg;
    bool& attr = per_device_flag("fft2d64_batch");
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)fft2d64_batch_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)fft2d64_batch_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        if (e != hipSuccess) {
            set_error("64 x 64 batched transform: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return EFGP_EHIP;
        }
        attr = true;
    }
    const size_t lds = (size_t)2 * s64::BUF * sizeof(double2);
    if (src_is_real) hipLaunchKernelGGL(fft2d64_batch_kernel<true>, dim3(nbatch), dim3(kThreads), lds, stream, src, src_stride, L0, L1, dst, dst_stride);
    else hipLaunchKernelGGL(fft2d64_batch_kernel<false>, dim3(nbatch), dim3(kThreads), lds, stream, src, src_stride, L0, L1, dst, dst_stride);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("64 x 64 batched transform launch failed: %s", hipGetErrorString(e));
        return EFGP_EHIP;
    }
    return EFGP_OK;
}

namespace pcg {

}  // namespace pcg

bool toeplitz_vhat_fused_eligible(const ToepGeom& g) {
    return g.d == 2 && g.F[0] == 64 && g.F[1] == 64 && std::getenv("EFGP_NO_VHAT64") == nullptr;
}

int toeplitz_vhat_fused_launch(const double2* v, int L0, int L1, double factor, double2* vhat, hipStream_t stream) {
    using namespace pcg;
    bool& attr = per_device_flag("vhat_2d64");
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)toeplitz_vhat_2d64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024 - 256);
        if (e != hipSuccess) {
            set_error("Toeplitz spectrum (64x64): hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return EFGP_EHIP;
        }
        attr = true;
    }
    hipLaunchKernelGGL(toeplitz_vhat_2d64_kernel, dim3(1), dim3(kThreads), (size_t)2 * s64::BUF * sizeof(double2), stream,
                       v, L0, L1, factor, vhat);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("Toeplitz spectrum (64x64) launch failed: %s", hipGetErrorString(e));
        return EFGP_EHIP;
    }
    return EFGP_OK;
}

// vhat64 (may be null) and vhat48 in one launch; both are FFT(zero-padded v) / (their grid size), natural order
int toeplitz_vhat_pair_launch(const double2* v, int L0, int L1, double2* vhat64, double2* vhat48, hipStream_t stream) {
    using namespace pcg;
    bool& attr = per_device_flag("vhat_pair");
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)toeplitz_vhat_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024 - 256);
        if (e != hipSuccess) {
            set_error("Toeplitz spectra (64x64 + 48x48): hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return EFGP_EHIP;
        }
        attr = true;
    }
    const size_t lds = vhat64 ? (size_t)2 * s64::BUF * sizeof(double2) : (size_t)(48 * 49 + 64 * 72) * sizeof(double2);
    hipLaunchKernelGGL(toeplitz_vhat_pair_kernel, dim3(vhat64 ? 2 : 1), dim3(kThreads), lds, stream, v, L0, L1, 1.0 / 4096.0, vhat64,
                       1.0 / 2304.0, vhat48);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("Toeplitz spectra (64x64 + 48x48) launch failed: %s", hipGetErrorString(e));
        return EFGP_EHIP;
    }
    return EFGP_OK;
}

namespace pcg {

// y[row] = post .* T(pre .* x[row]) on the 64 x 64 circulant grid in ONE launch, one workgroup per row: the operator part of
// cg_persistent_2d64_kernel's iteration (same lane maps, same pruned passes, same arithmetic) without the recurrences.
// The hyper-gradient applies T three times per step outside a solve (T g, T D'F*Z, T D V: efgpnd.py:150-153, :186-189, :203);
// through pad / two rocFFT plans / multiply / two rocFFT plans / crop that is 7 dependent launches each.
// pre / post: M-length complex diagonals or null; x may be real (x_is_real: M doubles per row).
struct ApplyArgs {
    int n, M;
    const double2* tw;       // exp(-2 pi i q / 64)
    const double2* vhat;     // [64][64], already divided by 4096
    const double2* pre;
    int pre_stride;          // entries between consecutive elements of pre (a column of an (M, H) array: H)
    const double2* post;
    const void* x;
    int x_is_real;
    double2* y;
};

__global__ __launch_bounds__(kThreads) void toeplitz_apply_2d64_kernel(ApplyArgs a) {
    using namespace s64;
    constexpr int LD = 72, BUF = F * LD;    // row pitch as in cg_persistent_2d64_kernel (8 mod 16)
    constexpr int KS = 2;
    extern __shared__ double2 lds2[];
    double2* const bufA = lds2;
    double2* const bufB = lds2 + BUF;
    const int n = a.n, M = a.M;
    const int64_t base = (int64_t)blockIdx.x * M;
    const int tid = threadIdx.x;
    const int j_row = tid & 7, r_row = tid >> 3;
    const bool row_act = r_row < n;
    const int c_col = tid & 63, j_col = tid >> 6;
    const int nv_row = j_row < n ? (n - 1 - j_row) / 8 + 1 : 0;
    const int nv_col = j_col < n ? (n - 1 - j_col) / 8 + 1 : 0;
    unsigned keep_col = 0, keep_row = 0;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int pc = j_col + 8 * t, pr = j_row + 8 * t;
        if (pc >= n - 1 && pc < 2 * n - 1) keep_col |= 1u << t;
        if (pr >= n - 1 && pr < 2 * n - 1) keep_row |= 1u << t;
    }
    double2 twr[7], twc[7];
#pragma unroll
    for (int t = 1; t < 8; ++t) {
        twr[t - 1] = a.tw[j_row * t];
        twc[t - 1] = a.tw[j_col * t];
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int t = tid + s * kThreads;
        if (t < M) {
            double2 u = a.x_is_real ? make_double2(((const double*)a.x)[base + t], 0.0) : ((const double2*)a.x)[base + t];
            if (a.pre) u = cmulp(a.pre[(int64_t)t * a.pre_stride], u);
            const int i0 = t / n, i1 = t - i0 * n;
            bufA[i0 * LD + i1] = u;
        }
    }
    __syncthreads();
    double2 v[8];
    // forward along dim 1 of the n non-zero rows (A -> B -> A), a row lives in one wave
    if (row_act) {
        load8<8>(bufA + r_row * LD + j_row, nv_row, v);
        dft_fwd<8>(v);
        store8_all<1>(bufB + r_row * LD + j_row * 9, v);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (row_act) {
        load8_all<9>(bufB + r_row * LD + j_row, v);
        twiddle8(v, twr);
        dft_fwd<8>(v);
        store8_all<8>(bufA + r_row * LD + j_row, v);
    }
    __syncthreads();
    // forward along dim 0 (rows >= n are zero), spectrum, inverse along dim 0 keeping the crop window's rows
    load8<8 * LD>(bufA + j_col * LD + c_col, nv_col, v);
    dft_fwd<8>(v);
    store8_all<LD>(bufB + j_col * 8 * LD + c_col, v);
    __syncthreads();
    load8_all<8 * LD>(bufB + j_col * LD + c_col, v);
    twiddle8(v, twc);
    dft_fwd<8>(v);
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = conjd(cmulp(v[t], a.vhat[(j_col + 8 * t) * F + c_col]));
    dft_fwd<8>(v);
    conj8(v);
    store8_all<LD>(bufA + j_col * 8 * LD + c_col, v);
    __syncthreads();
    load8_all<8 * LD>(bufA + j_col * LD + c_col, v);
    conj8(v);
    twiddle8(v, twc);
    dft_fwd<8>(v);
#pragma unroll
    for (int t = 0; t < 8; ++t)
        if (keep_col & (1u << t)) bufB[(j_col + 8 * t) * LD + c_col] = conjd(v[t]);
    __syncthreads();
    // inverse along dim 1 of the n window rows (B -> A -> B), cropped columns
    const int wrow = n - 1 + r_row;
    if (row_act) {
        load8_all<8>(bufB + wrow * LD + j_row, v);
        conj8(v);
        dft_fwd<8>(v);
        conj8(v);
        store8_all<1>(bufA + wrow * LD + j_row * 9, v);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (row_act) {
        load8_all<9>(bufA + wrow * LD + j_row, v);
        conj8(v);
        twiddle8(v, twr);
        dft_fwd<8>(v);
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (keep_row & (1u << t)) bufB[wrow * LD + j_row + 8 * t] = conjd(v[t]);
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int t = tid + s * kThreads;
        if (t < M) {
            const int i0 = t / n, i1 = t - i0 * n;
            double2 g = bufB[(i0 + n - 1) * LD + (i1 + n - 1)];
            if (a.post) g = cmulp(a.post[t], g);
            a.y[base + t] = g;
        }
    }
}

}  // namespace pcg

bool toeplitz_apply_fused_eligible(const ToepGeom& g) {
    return g.d == 2 && g.F[0] == 64 && g.F[1] == 64 && g.n[0] == g.n[1] && g.n[0] <= 32 && g.M <= 2 * pcg::kThreads &&
           std::getenv("EFGP_NO_APPLY64") == nullptr;
}

int toeplitz_apply_fused_launch(const ToepGeom& g, const double2* tw64, const double2* vhat, const double2* pre, const double2* post,
                                const void* x, int x_is_real, double2* y, int rows, hipStream_t stream, int pre_stride) {
    using namespace pcg;
    bool& attr = per_device_flag("apply_2d64");
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)toeplitz_apply_2d64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024 - 256);
        if (e != hipSuccess) {
            set_error("Toeplitz apply (64x64): hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return EFGP_EHIP;
        }
        attr = true;
    }
    ApplyArgs a;
    a.n = (int)g.n[0];
    a.M = (int)g.M;
    a.tw = tw64;
    a.vhat = vhat;
    a.pre = pre;
    a.pre_stride = pre_stride;
    a.post = post;
    a.x = x;
    a.x_is_real = x_is_real;
    a.y = y;
    hipLaunchKernelGGL(toeplitz_apply_2d64_kernel, dim3(rows), dim3(kThreads), (size_t)2 * 64 * 72 * sizeof(double2), stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("Toeplitz apply (64x64) launch failed: %s", hipGetErrorString(e));
        return EFGP_EHIP;
    }
    return EFGP_OK;
}

namespace pcg {

#ifdef EFGP_CG_STAMPS
static long long* g_last_stamps = nullptr;
#endif

static void radices_for(int n, Pass* p) {
    int k = 0;
    while ((1 << k) < n) ++k;
    p->nstages = 0;
    while (k >= 3 && (k != 4)) {
        p->radix[p->nstages++] = 8;
        k -= 3;
    }
    if (k == 4) {
        p->radix[p->nstages++] = 4;
        p->radix[p->nstages++] = 4;
        k = 0;
    }
    if (k == 2) p->radix[p->nstages++] = 4;
    if (k == 1) p->radix[p->nstages++] = 2;
}

}  // namespace pcg

#ifdef EFGP_CG_STAMPS
extern "C" int efgp_debug_cg_stamps(long long* out16) {
    (void)hipDeviceSynchronize();
    if (!pcg::g_last_stamps) return -1;
    return hipMemcpy(out16, pcg::g_last_stamps, 16 * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}
#endif

bool persistent_cg_eligible(const ToepGeom& tg) {
    int64_t padded = 1;
    for (int a = 0; a < tg.d; ++a) {
        if (tg.F[a] & (tg.F[a] - 1)) return false;
        if (tg.F[a] > 4096) return false;
    }
    // padded leading dimension on the fastest axis when d > 1
    for (int a = 0; a < tg.d; ++a) padded *= (a == tg.d - 1 && tg.d > 1) ? tg.F[a] + 1 : tg.F[a];
    if (padded > pcg::kMaxGrid) return false;
    if (tg.M > (int64_t)pcg::kSlots * pcg::kThreads) return false;
    return true;
}

int persistent_cg_launch(const ToepGeom& tg, const double2* const* twiddles, const double2* vhat, const double2* ws,
                         const double* diag, double sigmasq, int variant, double tol, int early_stop, int batched,
                         int max_iter, const double2* b, double2* x, int rows, int* d_iters, hipStream_t stream,
                         const double* diag_scale, int b_times_ws, int zero_x0, const LanczosOut* lz, int hermitian,
                         const Herm48Operands* h48, const double2* x0) {
    using namespace pcg;
    Args a;
    Geom& g = a.g;
    g.d = tg.d;
    for (int i = 0; i < 3; ++i) {
        g.n[i] = i < tg.d ? (int)tg.n[i] : 1;
        g.F[i] = i < tg.d ? (int)tg.F[i] : 1;
        g.tw[i] = i < tg.d ? twiddles[i] : nullptr;
    }
    // shift geometry so that the LAST real dimension sits in slot 2 for the vhat flat index (F[1], F[2] used)
    // -> we keep dims in slots 0..d-1 and set missing trailing F to 1, flat = (i0*F1 + i1)*F2 + i2 holds.
    // strides of the padded grid
    int stride = 1;
    for (int i = 2; i >= 0; --i) {
        if (i >= tg.d) {
            g.ld[i] = 0;
            continue;
        }
        g.ld[i] = stride;
        stride *= (i == tg.d - 1 && tg.d > 1) ? g.F[i] + 1 : g.F[i];
    }
    g.padded = stride;
    g.M = (int)tg.M;
    g.npass = tg.d;
    // forward: last dimension first; lines restricted to the leading n-box of not-yet-transformed dims
    for (int q = 0; q < tg.d; ++q) {
        const int dim = tg.d - 1 - q;
        Pass& p = g.fwd[q];
        p.dim = dim;
        p.n = g.F[dim];
        radices_for(p.n, &p);
        int oi = 0;
        for (int o = 0; o < 3; ++o) {
            if (o == dim) continue;
            p.other[oi] = o;
            if (o >= tg.d) {
                p.lo[oi] = 0;
                p.hi[oi] = 1;
            } else if (o < dim) {       // not transformed yet: only the n-box is non-zero
                p.lo[oi] = 0;
                p.hi[oi] = g.n[o];
            } else {                    // already transformed: full range
                p.lo[oi] = 0;
                p.hi[oi] = g.F[o];
            }
            ++oi;
        }
        p.in_limit = g.n[dim];
        p.out_lo = 0;
        p.out_hi = p.n;
    }
    // inverse: first dimension first; outputs cropped to [n-1, 2n-1); later passes only visit cropped lines
    for (int q = 0; q < tg.d; ++q) {
        const int dim = q;
        Pass& p = g.inv[q];
        p.dim = dim;
        p.n = g.F[dim];
        radices_for(p.n, &p);
        int oi = 0;
        for (int o = 0; o < 3; ++o) {
            if (o == dim) continue;
            p.other[oi] = o;
            if (o >= tg.d) {
                p.lo[oi] = 0;
                p.hi[oi] = 1;
            } else if (o < dim) {       // already inverse-transformed and cropped
                p.lo[oi] = g.n[o] - 1;
                p.hi[oi] = 2 * g.n[o] - 1;
            } else {
                p.lo[oi] = 0;
                p.hi[oi] = g.F[o];
            }
            ++oi;
        }
        p.in_limit = p.n;
        p.out_lo = g.n[dim] - 1;
        p.out_hi = 2 * g.n[dim] - 1;
    }
    // inverse passes use the forward radices in reverse order (so that the fused middle stage lines up)
    for (int q = 0; q < tg.d; ++q) {
        Pass& p = g.inv[q];
        for (int i = 0; i < p.nstages / 2; ++i) std::swap(p.radix[i], p.radix[p.nstages - 1 - i]);
    }
    // `other[1]` must be the faster-varying (smaller stride) of the two so that lanes walk it first
    for (int q = 0; q < tg.d; ++q) {
        for (Pass* p : {&g.fwd[q], &g.inv[q]}) {
            const int s0 = p->other[0] < tg.d ? g.ld[p->other[0]] : 0;
            const int s1 = p->other[1] < tg.d ? g.ld[p->other[1]] : 0;
            const bool swap = (p->other[1] >= tg.d) ? (p->other[0] < tg.d) : (p->other[0] < tg.d && s0 < s1 && s0 > 0);
            if (swap) {
                std::swap(p->other[0], p->other[1]);
                std::swap(p->lo[0], p->lo[1]);
                std::swap(p->hi[0], p->hi[1]);
            }
        }
    }
    auto ilog2 = [](int v) {
        int l = 0;
        while ((1 << l) < v) ++l;
        return l;
    };
    for (int q = 0; q < tg.d; ++q) {
        for (Pass* p : {&g.fwd[q], &g.inv[q]}) {
            p->w1_log2 = ilog2(p->hi[1] - p->lo[1]);
            p->lines_log2 = p->w1_log2 + ilog2(p->hi[0] - p->lo[0]);
        }
    }
    // fused middle stage: forward last pass (dim 0) and inverse first pass (dim 0) share dimension and the
    // last forward radix equals the first inverse radix by construction; both visit the same lines (all of
    // the other dimensions' FFT range), so the fusion is always structurally valid when d >= 1
    g.fuse_mid = 1;
    // LDS twiddle copies behind the two ping-pong buffers while they fit in the 160 KB budget
    g.tw_lds_total = 0;
    const int lds_budget = 160 * 1024 - 256;
    for (int i = 0; i < 3; ++i) g.tw_lds_off[i] = -1;
    for (int i = 0; i < tg.d; ++i) {
        int shared = -1;
        for (int b_ = 0; b_ < i; ++b_)
            if (g.F[b_] == g.F[i] && g.tw_lds_off[b_] >= 0) shared = g.tw_lds_off[b_];
        if (shared >= 0) {
            g.tw_lds_off[i] = shared;
            continue;
        }
        const size_t need = ((size_t)2 * g.padded + g.tw_lds_total + g.F[i]) * sizeof(double2);
        if (need <= (size_t)lds_budget) {
            g.tw_lds_off[i] = g.tw_lds_total;
            g.tw_lds_total += g.F[i];
        }
    }
    int vstride[3] = {0, 0, 0};
    {
        int acc = 1;
        for (int i = tg.d - 1; i >= 0; --i) {
            vstride[i] = acc;
            acc *= g.F[i];
        }
    }
    for (int q = 0; q < tg.d; ++q) {
        for (Pass* p : {&g.fwd[q], &g.inv[q]}) {
            p->vs_pos = vstride[p->dim];
            p->vs_c0 = p->other[0] < tg.d ? vstride[p->other[0]] : 0;
            p->vs_c1 = p->other[1] < tg.d ? vstride[p->other[1]] : 0;
            p->pstride = g.ld[p->dim];
            p->s0 = p->other[0] < tg.d ? g.ld[p->other[0]] : 0;
            p->s1 = p->other[1] < tg.d ? g.ld[p->other[1]] : 0;
            p->tw_lds = g.tw_lds_off[p->dim];
            p->tw_glob = g.tw[p->dim];
        }
    }
    a.ws = ws;
    a.diag = diag;
    a.diag_scale = diag_scale;
    a.b_times_ws = b_times_ws;
    a.zero_x0 = zero_x0;
    a.x0 = x0 ? x0 : x;
    a.vhat = vhat;
    a.sigmasq = sigmasq;
    a.variant = variant;
    a.tol = tol;
    a.early_stop = early_stop;
    a.batched = batched;
    a.max_iter = max_iter;
    a.b = b;
    a.x = x;
    a.iters = d_iters;
    a.hist = cg_history().buf;
    a.hist_cap = cg_history().capacity;
    a.lz_steps = lz ? lz->steps : 0;
    a.lz_alpha = lz ? lz->alpha : nullptr;
    a.lz_beta = lz ? lz->beta : nullptr;
    a.lz_norm2 = lz ? lz->norm2 : nullptr;
#ifdef EFGP_CG_STAMPS
    {
        static long long* d_stamps = nullptr;
        if (!d_stamps) (void)hipMalloc((void**)&d_stamps, 16 * sizeof(long long));
        (void)hipMemsetAsync(d_stamps, 0, 16 * sizeof(long long), stream);
        a.stamps = d_stamps;
        g_last_stamps = d_stamps;
    }
#endif
    const size_t lds = ((size_t)2 * g.padded + g.tw_lds_total) * sizeof(double2);
    bool& attr_set = per_device_flag("cg_persistent");
    if (!attr_set && lds > 65536) {
        hipError_t e = hipFuncSetAttribute((const void*)cg_persistent_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024 - 256);
        if (e != hipSuccess) {
            set_error("persistent CG: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return EFGP_EHIP;
        }
        attr_set = true;
    }
    const bool fast64 = tg.d == 2 && g.F[0] == 64 && g.F[1] == 64 && g.n[0] == g.n[1] && g.n[0] <= 32 &&
                        std::getenv("EFGP_NO_CG64") == nullptr;
    const bool herm64 = fast64 && hermitian && !lz && (g.n[0] & 1) && g.n[0] <= 31 && std::getenv("EFGP_NO_CG_HERM") == nullptr;
    // blocks of up to 23 x 23 modes: the smallest circulant grid, 48 x 48 (the operator holds a second spectrum for it)
    const bool herm48 = herm64 && h48 != nullptr && h48->vhat != nullptr && g.n[0] <= 23 && std::getenv("EFGP_NO_CG48") == nullptr;
    if (herm48) {
        bool& attr_h = per_device_flag("cg_herm48");
        const size_t lds_h = (size_t)h48::kLdsElems * sizeof(double2);
        if (!attr_h) {
            hipError_t e2 = hipFuncSetAttribute((const void*)cg_herm48_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h);
            if (e2 == hipSuccess) e2 = hipFuncSetAttribute((const void*)cg_herm48_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h);
            if (e2 != hipSuccess) {
                set_error("persistent CG (48x48, Hermitian): hipFuncSetAttribute failed: %s", hipGetErrorString(e2));
                return EFGP_EHIP;
            }
            attr_h = true;
        }
        a.vhat = h48->vhat;
        g.F[0] = g.F[1] = 48;
        g.tw[0] = g.tw[1] = h48->tw;
        KernelTimer timer("cg_solve", stream);
        if (variant == 0) hipLaunchKernelGGL(cg_herm48_kernel<0>, dim3(rows), dim3(h48::kThreadsH), lds_h, stream, a);
        else hipLaunchKernelGGL(cg_herm48_kernel<1>, dim3(rows), dim3(h48::kThreadsH), lds_h, stream, a);
    } else if (herm64) {
        bool& attr_h = per_device_flag("cg_herm64");
        const size_t lds_h = (size_t)h64::kLdsElems * sizeof(double2);
        if (!attr_h) {
            hipError_t e2 = hipFuncSetAttribute((const void*)cg_herm64_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h);
            if (e2 == hipSuccess) e2 = hipFuncSetAttribute((const void*)cg_herm64_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h);
            if (e2 != hipSuccess) {
                set_error("persistent CG (64x64, Hermitian): hipFuncSetAttribute failed: %s", hipGetErrorString(e2));
                return EFGP_EHIP;
            }
            attr_h = true;
        }
        KernelTimer timer("cg_solve", stream);
        if (variant == 0) hipLaunchKernelGGL(cg_herm64_kernel<0>, dim3(rows), dim3(h64::kThreadsH), lds_h, stream, a);
        else hipLaunchKernelGGL(cg_herm64_kernel<1>, dim3(rows), dim3(h64::kThreadsH), lds_h, stream, a);
    } else if (tg.d == 1 && !lz && g.n[0] <= 64 * l1d::KS - 1 && g.F[0] >= 8 && g.F[0] <= 512 && (g.F[0] & (g.F[0] - 1)) == 0 &&
               std::getenv("EFGP_NO_CG_LINE1D") == nullptr) {
        KernelTimer timer("cg_solve", stream);             // 1-D: one wave per system
        hipLaunchKernelGGL(cg_line1d_kernel, dim3(rows), dim3(64), (size_t)4 * g.F[0] * sizeof(double2), stream, a);
    } else if (fast64) {
        bool& attr64 = per_device_flag("cg_2d64");
        if (!attr64) {
            hipError_t e2 = hipFuncSetAttribute((const void*)cg_persistent_2d64_kernel,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
            if (e2 != hipSuccess) {
                set_error("persistent CG (64x64): hipFuncSetAttribute failed: %s", hipGetErrorString(e2));
                return EFGP_EHIP;
            }
            attr64 = true;
        }
        KernelTimer timer("cg_solve", stream);
        hipLaunchKernelGGL(cg_persistent_2d64_kernel, dim3(rows), dim3(kThreads), (size_t)2 * 64 * 72 * sizeof(double2),
                           stream, a);
    } else {
        KernelTimer timer("cg_solve", stream);
        hipLaunchKernelGGL(cg_persistent_kernel, dim3(rows), dim3(kThreads), lds, stream, a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("persistent CG launch failed: %s", hipGetErrorString(e));
        return EFGP_EHIP;
    }
    return EFGP_OK;
}

}  // namespace efgp
