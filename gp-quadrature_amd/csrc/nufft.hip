// NUFFT type-1 / type-2 for gfx950: hand-written spread / interpolate kernels around hipFFT
// (rocFFT) transforms of a small fine grid.
//
// Replaces the FINUFFT calls of the reference (efgpnd.py:1496-1499, 1533-1549, 1679).
//
// Data layout in HBM
//   x        (N,d) row-major doubles, caller-owned, read once per transform (coalesced 8d B/point)
//   strengths real (N) / complex (N) per batch row, read once
//   slabs    [batch][workgroup][channel][prod nf] doubles  -- per-workgroup partial fine grids that
//            were accumulated in LDS (each workgroup streams a contiguous chunk of the points)
//   fine     [batch][prod nf] complex -- reduced fine grid, transformed in place by hipFFT
//   out      [batch][prod n_modes] complex
// The fine grid is tiny for this workload (n_modes <= ~150 per dimension), so whenever it fits in
// the 160 KB LDS of a CU the points are streamed UNSORTED and accumulated with LDS atomics; larger
// grids fall back to global (L2) atomics.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.hpp"
#include "es_kernel.hpp"
#include "nufft_dev.hpp"
#include "points_layout.hpp"
#include "spread_mfma.hpp"
#include "small_dft.hpp"
#include "line_fft.hpp"

namespace efgp {


// ------------------------------------------------------------------------------------------
// type-1 spreader.  One thread per point; grid = (workgroups, nbatch).
//   USE_LDS: accumulate into an LDS-resident copy of the fine grid, flush to the workgroup's slab.
//   else   : atomics straight into slab 0 (global memory, pre-zeroed).
// ------------------------------------------------------------------------------------------
struct SpreadArgs {
    const double* x;
    StrengthSrc src;
    int64_t npts;
    GridGeom g;
    const double* coef;       // [degree+1][W]
    int degree;
    int channels;             // 1 or 2
    double* slabs;            // [batch][nslab][channels][cells]  (int64 fixed point when USE_LDS)
    int nslab;
    const double* scale;      // [0] = power-of-two fixed-point scale S, [1] = 1/S   (USE_LDS only)
    const int* order;         // spread_pad_kernel: per-plan bank-balanced processing order (null: balance per chunk)
};

// LDS accumulation is done in exact 64-bit fixed point: a contribution v is added as round(v*S) with
// S a power of two chosen from max|c| and the points per workgroup so that no sum can overflow.
// Integer LDS atomics are ~2x faster than ds_add_f64 under the random bank conflicts of unsorted
// points (tools/lds_atomic_bench.hip) and make the result independent of the accumulation order.
__device__ __forceinline__ void lds_add_fixed(double* cell, double a_scaled, double b) {
    const double t = fma(a_scaled, b, kFixMagic);
    const long long m = __double_as_longlong(t) - __double_as_longlong(kFixMagic);
    __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(cell), (unsigned long long)m, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
}
// the same sum in a GLOBAL int64 grid (device-scope atomic): the spreader of grids beyond LDS with too few points for the tiled
// path -- exact and order independent like the LDS form (floating-point atomics made the small-N pair pass differ from run to
// run in the last bits, which an ill-conditioned CG amplifies into a different iteration count)
__device__ __forceinline__ void global_add_fixed(double* cell, double a_scaled, double b) {
    const double t = fma(a_scaled, b, kFixMagic);
    const long long m = __double_as_longlong(t) - __double_as_longlong(kFixMagic);
    __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(cell), (unsigned long long)m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// "raw" variant: adds the bit pattern of (v*S + 1.5*2^52) itself.  bits(magic) = 0x4338 << 48 has zero low 48
// bits, so after n additions the low 48 bits of the cell hold (sum of the rounded integers) mod 2^48, which is
// the exact sum whenever it is below 2^47 in magnitude (guaranteed by the choice of S); the flush
// sign-extends bit 47.  Saves the 64-bit subtraction (two VALU instructions) per atomic.
__device__ __forceinline__ void lds_add_raw(double* cell, double a_scaled, double b) {
    const double t = fma(a_scaled, b, kFixMagic);
    __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(cell), (unsigned long long)__double_as_longlong(t),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ long long sext48(long long v) { return (v << 16) >> 16; }

template <int D, int W, bool USE_LDS>
__global__ __launch_bounds__(kSpreadThreads) void spread_kernel(SpreadArgs a) {
    extern __shared__ double lds[];
    const int batch = blockIdx.y;
    const int64_t cells = a.g.cells;
    const int C = a.channels;
    double* slab = a.slabs + ((int64_t)batch * a.nslab + (USE_LDS ? blockIdx.x : 0)) * C * cells;
    if (USE_LDS) {
        for (int64_t i = threadIdx.x; i < C * cells; i += kSpreadThreads) lds[i] = 0.0;
        __syncthreads();
    }
    double* acc = USE_LDS ? lds : slab;
    const double S = a.scale[0], S1 = a.scale[2];        // both forms accumulate in fixed point

    // contiguous chunk of points per workgroup (streaming, coalesced)
    const int64_t per = (a.npts + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per;
    const int64_t hi = lo + per < a.npts ? lo + per : a.npts;
    const int nf0 = (int)a.g.nf[0], nf1 = (int)a.g.nf[1], nf2 = (int)a.g.nf[2];

    for (int64_t n = lo + threadIdx.x; n < hi; n += kSpreadThreads) {
        double c0, c1;
        fetch_strength(a.src, batch, n, c0, c1);
        c0 *= S;
        c1 *= S1;
        double v0[W], v1[W], v2[W];
        int f0 = 0, f1 = 0, f2 = 0;
        {
            double Xw[3] = {0.0, 0.0, 0.0};
            Xw[0] = fold(a.g.scale[0] * (a.x[n * D + 0] - a.g.xcen[0]), (double)nf0);
            if (D > 1) Xw[1] = fold(a.g.scale[1] * (a.x[n * D + 1] - a.g.xcen[1]), (double)nf1);
            if (D > 2) Xw[2] = fold(a.g.scale[2] * (a.x[n * D + 2] - a.g.xcen[2]), (double)nf2);
            window_eval<D, W>(a.coef, a.degree, Xw, nf0, nf1, nf2, f0, f1, f2, v0, v1, v2);
        }
        if (D == 1) {
#pragma unroll
            for (int j = 0; j < W; ++j) {
                int i = wrap(f0 + j, nf0);
                double w = v0[j];
                if (USE_LDS) {
                    lds_add_fixed(&acc[i], c0, w);
                    if (C == 2) lds_add_fixed(&acc[cells + i], c1, w);
                } else {
                    global_add_fixed(&acc[i], c0, w);
                    if (C == 2) global_add_fixed(&acc[cells + i], c1, w);
                }
            }
        } else if (D == 2) {
#pragma unroll
            for (int j0 = 0; j0 < W; ++j0) {
                const int row = wrap(f0 + j0, nf0) * nf1;
                const double w0 = v0[j0];
                const double a0 = w0 * c0, a1 = w0 * c1;
#pragma unroll
                for (int j1 = 0; j1 < W; ++j1) {
                    int i = row + wrap(f1 + j1, nf1);
                    if (USE_LDS) {
                        lds_add_fixed(&acc[i], a0, v1[j1]);
                        if (C == 2) lds_add_fixed(&acc[cells + i], a1, v1[j1]);
                    } else {
                        global_add_fixed(&acc[i], a0, v1[j1]);
                        if (C == 2) global_add_fixed(&acc[cells + i], a1, v1[j1]);
                    }
                }
            }
        } else {
            for (int j0 = 0; j0 < W; ++j0) {
                const int64_t p0 = (int64_t)wrap(f0 + j0, nf0) * nf1;
                for (int j1 = 0; j1 < W; ++j1) {
                    const int64_t p1 = (p0 + wrap(f1 + j1, nf1)) * nf2;
                    const double w01 = v0[j0] * v1[j1];
                    const double a0 = w01 * c0, a1 = w01 * c1;
#pragma unroll
                    for (int j2 = 0; j2 < W; ++j2) {
                        int64_t i = p1 + wrap(f2 + j2, nf2);
                        if (USE_LDS) {
                            lds_add_fixed(&acc[i], a0, v2[j2]);
                            if (C == 2) lds_add_fixed(&acc[cells + i], a1, v2[j2]);
                        } else {
                            global_add_fixed(&acc[i], a0, v2[j2]);
                            if (C == 2) global_add_fixed(&acc[cells + i], a1, v2[j2]);
                        }
                    }
                }
            }
        }
    }
    if (USE_LDS) {
        __syncthreads();
        for (int64_t i = threadIdx.x; i < C * cells; i += kSpreadThreads) slab[i] = lds[i];
    }
}

// LDS-resident spreader with the LAST dimension padded by W-1 halo cells: the innermost stencil loop then
// needs no wrap arithmetic and its atomics use immediate offsets from one address per stencil row (PMC on the
// plain kernel: 1365 VALU instructions per point, most of them address / wrap / fixed-point bookkeeping).
// RAW48: see lds_add_raw.  The halo columns are folded back when the tile is flushed to the slab.
// Per-plan processing order for spread_pad_kernel.  An LDS atomic wave-instruction is served in groups of 16
// consecutive lanes and is conflict free when their 8-byte cells differ mod 16 (tools/lds_atomic_bench.hip,
// patterns 6/7: 22 instead of 12 lane-atomics per ns per CU, whatever the rows).  All lanes walk the same stencil
// offsets, so this is a property of the points' first cells: class = (padded linear index of the first cell)
// mod 16.  Within windows of kOrderWindow points of a workgroup's range the points are arranged round-robin
// over the classes (r-th point of class c at sum_c' min(cnt[c'], r) + #{c' < c: cnt[c'] > r}): 16 consecutive
// positions hold 16 different classes until the rarest class runs out (~86 % of a 4096-point window).
// The window keeps the gather of x and the strengths inside ~100 KB.  Depends only on (x, fine grid, W, launch
// geometry): built once per plan and reused by every pass.  (32 classes over 32-lane groups measured no better.)
constexpr int kOrderWindow = 4096;

template <int D, int NC>
__global__ __launch_bounds__(kSpreadThreads) void class_order_kernel(GridGeom g, int W, const double* __restrict__ x, int64_t npts,
                                                                    int64_t per, int* __restrict__ order) {
    __shared__ int cnt[NC], rank_next[NC];
    const int64_t lo = (int64_t)blockIdx.x * per;
    const int64_t hi = lo + per < npts ? lo + per : npts;
    const int nl = (int)g.nf[D - 1];
    const int pl = nl + W - 1;
    constexpr int kPer = kOrderWindow / kSpreadThreads;
    for (int64_t wbase = lo; wbase < hi; wbase += kOrderWindow) {
        if (threadIdx.x < NC) {
            cnt[threadIdx.x] = 0;
            rank_next[threadIdx.x] = 0;
        }
        __syncthreads();
        int cls[kPer];
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const int64_t n = wbase + threadIdx.x + (int64_t)u * kSpreadThreads;
            cls[u] = -1;
            if (n < hi) {
                int lin = 0;
#pragma unroll
                for (int q = 0; q < D; ++q) {
                    const int nfq = (int)g.nf[q];
                    const double Xq = fold(g.scale[q] * (x[n * D + q] - g.xcen[q]), (double)nfq);
                    int fq = (int)ceil(Xq - 0.5 * W);
                    if (fq < 0) fq += nfq;
                    lin = lin * (q == D - 1 ? pl : nfq) + fq;
                }
                cls[u] = lin & (NC - 1);
                atomicAdd(&cnt[cls[u]], 1);
            }
        }
        __syncthreads();
        int c16[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) c16[c] = cnt[c];
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            if (cls[u] < 0) continue;
            const int64_t n = wbase + threadIdx.x + (int64_t)u * kSpreadThreads;
            const int c = cls[u];
            const int r = atomicAdd(&rank_next[c], 1);
            int pos = 0;
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                pos += min(c16[q], r);
                if (q < c && c16[q] > r) ++pos;
            }
            order[wbase + pos] = (int)n;
        }
        __syncthreads();
    }
}

template <int D, int W, bool RAW48>
__global__ __launch_bounds__(kSpreadThreads) void spread_pad_kernel(SpreadArgs a) {
    extern __shared__ double lds[];
    const int batch = blockIdx.y;
    const int64_t cells = a.g.cells;
    const int C = a.channels;
    const int nf0 = (int)a.g.nf[0], nf1 = (int)a.g.nf[1], nf2 = (int)a.g.nf[2];
    const int nl = D == 1 ? nf0 : (D == 2 ? nf1 : nf2);            // size of the last (padded) dimension
    const int pl = nl + W - 1;                                     // padded row length
    const int rows = (int)(cells / nl);
    const int plane = rows * pl;                                    // padded cells per channel
    double* slab = a.slabs + ((int64_t)batch * a.nslab + blockIdx.x) * C * cells;
    for (int i = threadIdx.x; i < C * plane; i += kSpreadThreads) lds[i] = 0.0;
    __syncthreads();
    const double S = a.scale[0], S1 = a.scale[2];
    const int64_t per = (a.npts + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per;
    const int64_t hi = lo + per < a.npts ? lo + per : a.npts;
    // Bank-balanced lane assignment -- fallback when the plan carries no precomputed order (a.order == nullptr: small
    // N, or EFGP_NO_CLASS_ORDER).  All lanes walk the same stencil offsets, so an atomic wave-instruction is conflict
    // free when the first cells of the 16 lanes of an LDS lane group differ mod 16 (see class_order_kernel).  Each
    // chunk of 1024 points is counting-sorted in LDS by (first cell mod 32) -- a refinement of that class; lane L of
    // group G takes the G-th point of class L, surplus points of over-full classes fill the lanes of under-full ones
    // (one pass, every lane busy).  PMC before: 67 % of LDS-active cycles were conflicts, with it 44 %.
    __shared__ int cls_cnt[32];
    __shared__ int cls_free[33];
    __shared__ int left_cnt;
    __shared__ unsigned short cls_list[32][32];
    __shared__ unsigned short left_list[kSpreadThreads];
    constexpr bool kBalance = D >= 2;          // 1-D has only W atomics per point: the bookkeeping does not pay
    for (int64_t cbase = lo; cbase < hi; cbase += kSpreadThreads) {
      int src = -1;
      int64_t n_ord = -1;
      if (a.order) {
        if (cbase + threadIdx.x < hi) {
            n_ord = a.order[cbase + threadIdx.x];
            src = 0;
        }
      } else if (!kBalance) {
        if (cbase + threadIdx.x < hi) src = threadIdx.x;
      } else {
        if (threadIdx.x < 32) cls_cnt[threadIdx.x] = 0;
        if (threadIdx.x == 0) left_cnt = 0;
        __syncthreads();
        {
            const int64_t nn = cbase + threadIdx.x;
            if (nn < hi) {
                int lin = 0;
#pragma unroll
                for (int q = 0; q < D; ++q) {
                    const int nfq = (int)a.g.nf[q];
                    const double Xq = fold(a.g.scale[q] * (a.x[nn * D + q] - a.g.xcen[q]), (double)nfq);
                    int fq = (int)ceil(Xq - 0.5 * W);
                    if (fq < 0) fq += nfq;
                    lin = lin * (q == D - 1 ? pl : nfq) + fq;
                }
                const int key = lin & 31;
                const int r = atomicAdd(&cls_cnt[key], 1);
                if (r < 32) cls_list[key][r] = (unsigned short)threadIdx.x;
                else left_list[atomicAdd(&left_cnt, 1)] = (unsigned short)threadIdx.x;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int run = 0;
            for (int q = 0; q < 32; ++q) {
                cls_free[q] = run;
                run += 32 - min(cls_cnt[q], 32);
            }
            cls_free[32] = run;
        }
        __syncthreads();
        {
            const int L = threadIdx.x & 31, G = threadIdx.x >> 5;
            const int have = min(cls_cnt[L], 32);
            if (G < have) {
                src = cls_list[L][G];
            } else {
                const int fi = cls_free[L] + (G - have);
                if (fi < left_cnt) src = left_list[fi];
            }
        }
        __syncthreads();                       // lists are rebuilt by the next chunk
      }
        if (src < 0) continue;
        const int64_t n = n_ord >= 0 ? n_ord : cbase + src;
        double c0, c1;
        fetch_strength(a.src, batch, n, c0, c1);
        c0 *= S;
        c1 *= S1;
        double v0[W], v1[W], v2[W];
        int f0 = 0, f1 = 0, f2 = 0;
        {
            double Xw[3] = {0.0, 0.0, 0.0};
            Xw[0] = fold(a.g.scale[0] * (a.x[n * D + 0] - a.g.xcen[0]), (double)nf0);
            if (D > 1) Xw[1] = fold(a.g.scale[1] * (a.x[n * D + 1] - a.g.xcen[1]), (double)nf1);
            if (D > 2) Xw[2] = fold(a.g.scale[2] * (a.x[n * D + 2] - a.g.xcen[2]), (double)nf2);
            window_eval<D, W>(a.coef, a.degree, Xw, nf0, nf1, nf2, f0, f1, f2, v0, v1, v2);
        }
        if (D == 1) {
            double* p = lds + f0;
#pragma unroll
            for (int j = 0; j < W; ++j) {
                if (RAW48) lds_add_raw(p + j, c0, v0[j]); else lds_add_fixed(p + j, c0, v0[j]);
                if (C == 2) { if (RAW48) lds_add_raw(p + plane + j, c1, v0[j]); else lds_add_fixed(p + plane + j, c1, v0[j]); }
            }
        } else if (D == 2) {
#pragma unroll
            for (int j0 = 0; j0 < W; ++j0) {
                double* p = lds + wrap(f0 + j0, nf0) * pl + f1;
                double* q = p + plane;
                const double a0 = v0[j0] * c0, a1 = v0[j0] * c1;
#pragma unroll
                for (int j1 = 0; j1 < W; ++j1) {
                    if (RAW48) lds_add_raw(p + j1, a0, v1[j1]); else lds_add_fixed(p + j1, a0, v1[j1]);
                    if (C == 2) { if (RAW48) lds_add_raw(q + j1, a1, v1[j1]); else lds_add_fixed(q + j1, a1, v1[j1]); }
                }
            }
        } else {
            for (int j0 = 0; j0 < W; ++j0) {
                const int r0 = wrap(f0 + j0, nf0) * nf1;
                for (int j1 = 0; j1 < W; ++j1) {
                    double* p = lds + (r0 + wrap(f1 + j1, nf1)) * pl + f2;
                    double* q = p + plane;
                    const double w01 = v0[j0] * v1[j1];
                    const double a0 = w01 * c0, a1 = w01 * c1;
#pragma unroll
                    for (int j2 = 0; j2 < W; ++j2) {
                        if (RAW48) lds_add_raw(p + j2, a0, v2[j2]); else lds_add_fixed(p + j2, a0, v2[j2]);
                        if (C == 2) { if (RAW48) lds_add_raw(q + j2, a1, v2[j2]); else lds_add_fixed(q + j2, a1, v2[j2]); }
                    }
                }
            }
        }
    }
    __syncthreads();
    // flush: fold the halo columns, undo the raw encoding, write the int64 slab
    const long long* li = reinterpret_cast<const long long*>(lds);
    long long* so = reinterpret_cast<long long*>(slab);
    for (int64_t i = threadIdx.x; i < C * cells; i += kSpreadThreads) {
        const int ch = (int)(i / cells);
        const int64_t cell = i - (int64_t)ch * cells;
        const int r = (int)(cell / nl), col = (int)(cell - (int64_t)r * nl);
        const long long* row = li + (int64_t)ch * plane + (int64_t)r * pl;
        long long v = RAW48 ? sext48(row[col]) : row[col];
        if (col < W - 1) v += RAW48 ? sext48(row[nl + col]) : row[nl + col];
        so[i] = v;
    }
}

// out[b][n] = the +-1 probe the spread kernels generate for (seed, row b, point n + index_offset)
__global__ void rademacher_fill_kernel(unsigned long long seed, int64_t npts, int64_t index_offset, double* __restrict__ out) {
    const int row = blockIdx.y;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < npts; n += (int64_t)gridDim.x * blockDim.x)
        out[(int64_t)row * npts + n] = efgp_rademacher(seed, row, (long long)((unsigned long long)n + (unsigned long long)index_offset));
}

// Fixed-point scales, one per channel: scale[0] = S0 (channel 0), [1] = 1/S0, [2] = S1 (channel 1), [3] = 1/S1,
// [4] = the factor the channel-0 result still has to be multiplied by (1 unless the channel is carried normalised).
//   S = largest power of two with  max|c| * S <= 2^min(50,sum_bits)  and  points_per_wg * max|c| * S <= 2^sum_bits.
//   Channel 0 follows the data maximum.  Channel 1 follows it too unless it is the implicit all-ones channel of the
//   fused fit pass (ones_channel != 0): that one always has magnitude 1 -- sharing the scale of a large |y| would
//   quantise the Toeplitz vector at max|y| * 2^-46 (y ~ 1e6 already costs its 6e-8 accuracy; y ~ 1e60 zeroes it).
// Consumes the max|c| accumulator and resets it to zero for the next transform (no memset launch per call).
__device__ __forceinline__ double fixed_scale_for(double cmax, int64_t per, int sum_bits) {
    if (!(cmax > 0.0) || !isfinite(cmax)) cmax = 1.0;
    // every value below 2^50 (or 2^(sum_bits)) and every per-workgroup sum below 2^sum_bits
    const double lim = fmin(ldexp(1.0, sum_bits < 50 ? sum_bits : 50), ldexp(1.0, sum_bits) / (double)(per > 1 ? per : 1));
    int e = 0;
    frexp(lim / cmax, &e);                  // lim/cmax = f * 2^e, f in [0.5, 1)
    return ldexp(1.0, e - 1);               // largest power of two <= lim/cmax
}
struct ScaleJob {            // what fixed_scale_write needs besides max|c|
    double floor_bound;
    int ones_channel;
    int64_t per;
    double* scale;
    int sum_bits;
};
__device__ __forceinline__ void fixed_scale_write(double cmax, double floor_bound, int ones_channel, int64_t per,
                                                  double* __restrict__ scale, int sum_bits) {
    if (!isfinite(cmax)) {
        // NaN / Inf among the strengths: the fixed-point sums are meaningless; a NaN inverse scale turns every output of the
        // transform into NaN, which is what the reference's floating-point sums give (FINUFFT propagates them)
        const double qnan = __longlong_as_double(0x7FF8000000000000ll);
        scale[0] = 1.0;
        scale[1] = qnan;
        scale[2] = 1.0;
        scale[3] = ones_channel ? 1.0 : qnan;
        scale[4] = 1.0;
        return;
    }
    if (!ones_channel) {
        const double S0 = fixed_scale_for(fmax(cmax, floor_bound), per, sum_bits);
        scale[0] = S0;
        scale[1] = 1.0 / S0;
        scale[2] = S0;
        scale[3] = 1.0 / S0;
        scale[4] = 1.0;
        return;
    }
    // Fused (y, ones) pass: the two real channels share ONE complex FFT and are separated afterwards by Hermitian
    // symmetry, so the transform's rounding error of the larger channel leaks into the smaller one (measured: |y| ~
    // 1e60 destroyed the Toeplitz vector, |y| ~ 1e-60 would destroy F*y).  Channel 0 is therefore carried
    // NORMALISED by norm0 = the power of two >= max|y| (exact), and deconvolve_pair_kernel multiplies F*y back.
    int e0 = 0;
    double norm0 = 1.0;
    if (cmax > 0.0 && isfinite(cmax)) {
        frexp(cmax, &e0);                   // cmax = f * 2^e0, f in [0.5, 1)
        norm0 = ldexp(1.0, e0);             // >= cmax
    }
    const double Sn = fixed_scale_for(1.0, per, sum_bits);      // both normalised channels have magnitude <= 1
    scale[0] = Sn / norm0;
    scale[1] = 1.0 / Sn;
    scale[2] = Sn;
    scale[3] = 1.0 / Sn;
    scale[4] = norm0;
}
// stand-alone form for strengths that need no max|c| pass (implicit ones, generated +-1 probes)
__global__ void fixed_scale_kernel(double floor_bound, int ones_channel, int64_t per, double* __restrict__ scale,
                                   int sum_bits) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    fixed_scale_write(0.0, floor_bound, ones_channel, per, scale, sum_bits);
}

// same, with max|c| read from a device word that holds it as an ordered bit pattern (the per-model cache of
// points_layout: y never changes between fits, so its N-length maximum pass runs once per model)
__global__ void fixed_scale_cached_kernel(const unsigned long long* __restrict__ cmax_bits, ScaleJob job) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    fixed_scale_write(__longlong_as_double((long long)*cmax_bits), job.floor_bound, job.ones_channel, job.per, job.scale, job.sum_bits);
}

// max |c| over n doubles as an ordered bit pattern (non-negative doubles compare like integers).
// One atomic per workgroup (block-level reduction first): thousands of same-address atomics serialise.
// The last workgroup to arrive (device-scope ticket) consumes the maximum, resets the accumulator and the ticket
// and writes the fixed-point scales: one launch instead of two dependent ones (~4.5 us each on this machine).
__global__ __launch_bounds__(1024) void maxabs_kernel(const double* __restrict__ c, int64_t n,
                                                       unsigned long long* __restrict__ out,
                                                       unsigned int* __restrict__ ticket, ScaleJob job) {
    __shared__ double part[16];
    double m = 0.0;
    const int64_t n2 = n >> 1;
    const double2* c2 = reinterpret_cast<const double2*>(c);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x) {
        const double2 v = c2[i];
        // fmax drops NaN: a non-finite strength must poison the transform (the reference returns NaN), not vanish
        m = fmax(m, fmax(isfinite(v.x) ? fabs(v.x) : INFINITY, isfinite(v.y) ? fabs(v.y) : INFINITY));
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) m = fmax(m, isfinite(c[n - 1]) ? fabs(c[n - 1]) : INFINITY);
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_down(m, off, 64));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t = fmax(t, part[i]);
        if (t > 0.0) atomicMax(out, (unsigned long long)__double_as_longlong(t));
        __threadfence();
        if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
            __threadfence();
            const double cmax = __longlong_as_double((long long)atomicExch(out, 0ull));
            *ticket = 0u;
            fixed_scale_write(cmax, job.floor_bound, job.ones_channel, job.per, job.scale, job.sum_bits);
        }
    }
}

// sum the slabs of one batch row into the complex fine grid: fine = ch0 + i*ch1 (ch1 = 0 if absent).
// FIXED: slabs hold int64 fixed-point sums (exact, order independent), rescaled by inv_scale.
// Block = 64 cells x 8 slab groups; every slab row segment read is 512 contiguous bytes.
template <bool FIXED>
__global__ __launch_bounds__(1024) void reduce_slabs_kernel(const double* __restrict__ slabs, int nslab, int channels,
                                                            int64_t cells, const double* __restrict__ scale,
                                                            double2* __restrict__ fine, int reset_source = 0) {
    // 8 (512 threads) or 16 (1024 threads) slab groups per 64 cells: the many-slab integer reduction is a chain of
    // latency-bound loads, so it takes the wider block (245 slabs of 96 x 96 x 2: 12.9 -> 8.4 us)
    __shared__ double part[2][8][64];
    const int batch = blockIdx.y;
    const int lane_cell = threadIdx.x & 63, grp = threadIdx.x >> 6, ngrp = blockDim.x >> 6;
    const int64_t cell = (int64_t)blockIdx.x * 64 + lane_cell;
    const double* base = slabs + (int64_t)batch * nslab * channels * cells;
    double re = 0.0, im = 0.0;
    long long ire = 0, iim = 0;
    if (cell < cells) {
#pragma unroll 4
        for (int s = grp; s < nslab; s += ngrp) {
            const double* p = base + (int64_t)s * channels * cells;
            if (FIXED) {
                ire += reinterpret_cast<const long long*>(p)[cell];
                if (channels == 2) iim += reinterpret_cast<const long long*>(p)[cells + cell];
            } else {
                re += p[cell];
                if (channels == 2) im += p[cells + cell];
            }
        }
    }
    if (FIXED) {
        // exact integer partial sums; combine groups in integer too
        __shared__ long long ipart[2][16][64];
        ipart[0][grp][lane_cell] = ire;
        ipart[1][grp][lane_cell] = iim;
        __syncthreads();
        if (grp == 0 && cell < cells) {
            long long a = 0, b = 0;
            for (int g = 0; g < ngrp; ++g) {
                a += ipart[0][g][lane_cell];
                b += ipart[1][g][lane_cell];
            }
            fine[(int64_t)batch * cells + cell] = make_double2((double)a * scale[1], (double)b * scale[3]);
            if (reset_source) {       // single int64 grid (MFMA spreader): leave it zeroed for the next pass
                long long* src = reinterpret_cast<long long*>(const_cast<double*>(base));
                src[cell] = 0;
                if (channels == 2) src[cells + cell] = 0;
            }
        }
    } else {
        part[0][grp][lane_cell] = re;
        part[1][grp][lane_cell] = im;
        __syncthreads();
        if (grp == 0 && cell < cells) {
            double a = 0.0, b = 0.0;
            for (int g = 0; g < 8; ++g) {
                a += part[0][g][lane_cell];
                b += part[1][g][lane_cell];
            }
            fine[(int64_t)batch * cells + cell] = make_double2(a, b);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Tiled spreading for fine grids that do not fit LDS.
//   The fine grid is cut into tiles of T[a] cells per dimension; a point belongs to the tile of its first
//   covered cell.  Points are counting-sorted by tile ONCE PER PLAN (x is fixed; LDS-ranked scatter, one
//   global atomic per (workgroup, tile)); `xs` holds the coordinates in tile order, `order` the original
//   indices (strengths are gathered through it).  A workgroup then takes a contiguous chunk of sorted
//   points, and for every tile segment inside it accumulates a (T+W-1)^d LDS tile with the same fixed-point
//   atomics as the LDS-resident spreader and adds the tile to the int64 global grid.
// ------------------------------------------------------------------------------------------
struct TileGeom {
    int d;
    int nf[3];
    int T[3];        // tile size in cells
    int nt[3];       // tiles per dimension
    int ext[3];      // T + W - 1: cells an LDS tile spans per dimension (1 for unused dims)
    int nbins;
    int W;
    double scale[3];
    double xcen[3];
};

template <int D>
__device__ __forceinline__ int tile_of_point(const TileGeom& t, const double* __restrict__ x, int64_t n) {
    int bin = 0;
#pragma unroll
    for (int a = 0; a < D; ++a) {
        const double X = fold(t.scale[a] * (x[n * D + a] - t.xcen[a]), (double)t.nf[a]);
        int f = (int)ceil(X - 0.5 * t.W);
        if (f < 0) f += t.nf[a];
        bin = bin * t.nt[a] + f / t.T[a];
    }
    return bin;
}

// per-workgroup LDS histogram -> global histogram
template <int D>
__global__ __launch_bounds__(1024) void bin_hist_kernel(TileGeom t, const double* __restrict__ x, int64_t npts,
                                                         int* __restrict__ hist) {
    extern __shared__ int lhist[];
    for (int i = threadIdx.x; i < t.nbins; i += blockDim.x) lhist[i] = 0;
    __syncthreads();
    const int64_t per = (npts + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < npts ? lo + per : npts;
    for (int64_t n = lo + threadIdx.x; n < hi; n += blockDim.x) atomicAdd(&lhist[tile_of_point<D>(t, x, n)], 1);
    __syncthreads();
    for (int i = threadIdx.x; i < t.nbins; i += blockDim.x)
        if (lhist[i]) atomicAdd(&hist[i], lhist[i]);
}

// exclusive scan of the histogram (one workgroup); start[nbins] = total; cursor = copy of start
__global__ __launch_bounds__(1024) void bin_scan_kernel(const int* __restrict__ hist, int nbins, int* __restrict__ start,
                                                         int* __restrict__ cursor) {
    __shared__ int part[1024];
    const int per = (nbins + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = min(lo + per, nbins);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += hist[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < 1024; ++i) {
            const int v = part[i];
            part[i] = run;
            run += v;
        }
        start[nbins] = run;
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int i = lo; i < hi; ++i) {
        start[i] = run;
        cursor[i] = run;
        run += hist[i];
    }
}

// scatter the points of this workgroup's chunk to their tile ranges
template <int D>
__global__ __launch_bounds__(1024) void bin_scatter_kernel(TileGeom t, const double* __restrict__ x, int64_t npts,
                                                            int* __restrict__ cursor, double* __restrict__ xs,
                                                            int* __restrict__ order) {
    extern __shared__ int lmem[];
    int* lcount = lmem;                 // points of this chunk per tile
    int* lbase = lmem + t.nbins;        // global offset reserved for them
    for (int i = threadIdx.x; i < t.nbins; i += blockDim.x) lcount[i] = 0;
    __syncthreads();
    const int64_t per = (npts + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < npts ? lo + per : npts;
    // pass 1: local ranks (kept in registers; a thread handles up to kMaxPer points of the chunk)
    constexpr int kMaxPer = 32;
    int bins[kMaxPer], ranks[kMaxPer];
    int cnt = 0;
    for (int64_t n = lo + threadIdx.x; n < hi && cnt < kMaxPer; n += blockDim.x, ++cnt) {
        bins[cnt] = tile_of_point<D>(t, x, n);
        ranks[cnt] = atomicAdd(&lcount[bins[cnt]], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < t.nbins; i += blockDim.x) lbase[i] = lcount[i] ? atomicAdd(&cursor[i], lcount[i]) : 0;
    __syncthreads();
    cnt = 0;
    for (int64_t n = lo + threadIdx.x; n < hi && cnt < kMaxPer; n += blockDim.x, ++cnt) {
        const int64_t dst = (int64_t)lbase[bins[cnt]] + ranks[cnt];
#pragma unroll
        for (int a = 0; a < D; ++a) xs[dst * D + a] = x[n * D + a];
        order[dst] = (int)n;
    }
}

// Bank-balanced order inside every tile (the tiled counterpart of class_order_kernel): a tile's points are in
// arbitrary order after the counting sort, so they may be permuted freely.  Class = (linear cell index of the first
// covered cell inside the tile's LDS image) mod 16; within windows of kOrderWindow sorted positions of one tile the
// points are laid out round-robin over the classes, so that 16 consecutive positions -- the 16 lanes of an LDS
// atomic group in spread_tile_kernel -- hit 16 different 8-byte columns.  Out-of-place.
template <int D>
__global__ __launch_bounds__(kSpreadThreads) void tile_class_order_kernel(TileGeom t, const double* __restrict__ xs_in,
                                                                         const int* __restrict__ order_in,
                                                                         const int* __restrict__ start, int64_t npts,
                                                                         double* __restrict__ xs_out, int* __restrict__ order_out) {
    __shared__ int cnt[16], rank_next[16];
    __shared__ int s_bin;
    // one workgroup per window of kOrderWindow sorted positions; a window that spans several tiles lays out each
    // tile's segment separately
    const int64_t wlo = (int64_t)blockIdx.x * kOrderWindow;
    const int64_t whi = wlo + kOrderWindow < npts ? wlo + kOrderWindow : npts;
    if (threadIdx.x == 0) {
        int l = 0, r = t.nbins;            // invariant: start[l] <= wlo < start[r]
        while (r - l > 1) {
            const int m = (l + r) >> 1;
            if ((int64_t)start[m] <= wlo) l = m;
            else r = m;
        }
        s_bin = l;
    }
    __syncthreads();
    int bin = s_bin;
    constexpr int kPer = kOrderWindow / kSpreadThreads;
    int64_t cur = wlo;
    while (cur < whi) {
        while ((int64_t)start[bin + 1] <= cur) ++bin;            // skip empty tiles
        const int64_t lo = cur;
        const int64_t hi = (int64_t)start[bin + 1] < whi ? (int64_t)start[bin + 1] : whi;
        int rem = bin, o[3] = {0, 0, 0};
        for (int q = D - 1; q >= 0; --q) {
            o[q] = (rem % t.nt[q]) * t.T[q];
            rem /= t.nt[q];
        }
        if (threadIdx.x < 16) {
            cnt[threadIdx.x] = 0;
            rank_next[threadIdx.x] = 0;
        }
        __syncthreads();
        int cls[kPer];
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const int64_t n = lo + threadIdx.x + (int64_t)u * kSpreadThreads;
            cls[u] = -1;
            if (n < hi) {
                int lin = 0;
#pragma unroll
                for (int q = 0; q < D; ++q) {
                    const double Xq = fold(t.scale[q] * (xs_in[n * D + q] - t.xcen[q]), (double)t.nf[q]);
                    int fq = (int)ceil(Xq - 0.5 * t.W);
                    if (fq < 0) fq += t.nf[q];
                    lin = lin * t.ext[q] + (fq - o[q]);
                }
                cls[u] = lin & 15;
                atomicAdd(&cnt[cls[u]], 1);
            }
        }
        __syncthreads();
        int c16[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) c16[c] = cnt[c];
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            if (cls[u] < 0) continue;
            const int64_t n = lo + threadIdx.x + (int64_t)u * kSpreadThreads;
            const int c = cls[u];
            const int r = atomicAdd(&rank_next[c], 1);
            int pos = 0;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                pos += min(c16[q], r);
                if (q < c && c16[q] > r) ++pos;
            }
            const int64_t dst = lo + pos;
#pragma unroll
            for (int q = 0; q < D; ++q) xs_out[dst * D + q] = xs_in[n * D + q];
            order_out[dst] = order_in[n];
        }
        __syncthreads();
        cur = hi;
    }
}

// Extent of SLOT_SLABS known to be zero after a pass over its first acc_bytes (which the converting kernel cleared again): a
// pass SMALLER than what was known to be zero leaves the tail zero too -- the pair pass (2 channels on the 96 / 128 grid) and the
// probe pass (T channels on the 48 / 64 grid) of a gradient step alternate, and neither needs a memset launch.
static inline size_t zero_extent_after(size_t acc_bytes, size_t known_zero_before, bool same_buffer) {
    return std::max(acc_bytes, same_buffer ? known_zero_before : (size_t)0);
}

struct TileSpreadArgs {
    TileGeom t;
    const double* xs;         // tile-sorted coordinates
    const int* order;         // sorted position -> original index
    const int* start;         // [nbins + 1]
    StrengthSrc src;
    int64_t npts;
    int64_t chunk;            // sorted points per workgroup
    const double* coef;
    int degree;
    int channels;
    long long* gacc;          // [batch][channels][cells] int64 fixed-point global grid (pre-zeroed)
    int64_t cells;
    const double* scale;
};

template <int D, int W>
__global__ __launch_bounds__(kSpreadThreads) void spread_tile_kernel(TileSpreadArgs a) {
    extern __shared__ double lds[];
    __shared__ int s_bin;
    const TileGeom& t = a.t;
    const int batch = blockIdx.y;
    const int C = a.channels;
    const int e0 = t.ext[0], e1 = t.ext[1], e2 = t.ext[2];
    const int tcells = e0 * e1 * e2;
    const int64_t lo = (int64_t)blockIdx.x * a.chunk;
    const int64_t hi = lo + a.chunk < a.npts ? lo + a.chunk : a.npts;
    if (lo >= hi) return;
    const double S = a.scale[0], S1 = a.scale[2];
    long long* gacc = a.gacc + (int64_t)batch * C * a.cells;
    // first tile that contains sorted position `lo` (binary search by one lane)
    if (threadIdx.x == 0) {
        int l = 0, r = t.nbins;            // invariant: start[l] <= lo < start[r]
        while (r - l > 1) {
            const int m = (l + r) >> 1;
            if ((int64_t)a.start[m] <= lo) l = m;
            else r = m;
        }
        s_bin = l;
    }
    __syncthreads();
    int bin = s_bin;
    int64_t cur = lo;
    while (cur < hi) {
        while ((int64_t)a.start[bin + 1] <= cur) ++bin;            // skip empty tiles
        const int64_t seg_hi = (int64_t)a.start[bin + 1] < hi ? (int64_t)a.start[bin + 1] : hi;
        // tile origin
        int rem = bin, o[3] = {0, 0, 0};
        for (int q = D - 1; q >= 0; --q) {
            o[q] = (rem % t.nt[q]) * t.T[q];
            rem /= t.nt[q];
        }
        for (int i = threadIdx.x; i < C * tcells; i += kSpreadThreads) lds[i] = 0.0;
        __syncthreads();
        for (int64_t n = cur + threadIdx.x; n < seg_hi; n += kSpreadThreads) {
            double c0, c1;
            fetch_strength(a.src, batch, a.src.mode == STR_ONES ? 0 : (int64_t)a.order[n], c0, c1);
            c0 *= S;
            c1 *= S1;
            double v0[W], v1[W], v2[W];
            int f0 = 0, f1 = 0, f2 = 0;
            {
            double Xw[3] = {0.0, 0.0, 0.0};
            Xw[0] = fold(t.scale[0] * (a.xs[n * D + 0] - t.xcen[0]), (double)t.nf[0]);
            if (D > 1) Xw[1] = fold(t.scale[1] * (a.xs[n * D + 1] - t.xcen[1]), (double)t.nf[1]);
            if (D > 2) Xw[2] = fold(t.scale[2] * (a.xs[n * D + 2] - t.xcen[2]), (double)t.nf[2]);
            window_eval<D, W>(a.coef, a.degree, Xw, t.nf[0], t.nf[1], t.nf[2], f0, f1, f2, v0, v1, v2);
            f0 -= o[0];
            if (D > 1) f1 -= o[1];
            if (D > 2) f2 -= o[2];
        }
            if (D == 1) {
#pragma unroll
                for (int j = 0; j < W; ++j) {
                    lds_add_fixed(&lds[f0 + j], c0, v0[j]);
                    if (C == 2) lds_add_fixed(&lds[tcells + f0 + j], c1, v0[j]);
                }
            } else if (D == 2) {
#pragma unroll
                for (int j0 = 0; j0 < W; ++j0) {
                    const int row = (f0 + j0) * e1 + f1;
                    const double a0 = v0[j0] * c0, a1 = v0[j0] * c1;
#pragma unroll
                    for (int j1 = 0; j1 < W; ++j1) {
                        lds_add_fixed(&lds[row + j1], a0, v1[j1]);
                        if (C == 2) lds_add_fixed(&lds[tcells + row + j1], a1, v1[j1]);
                    }
                }
            } else {
                for (int j0 = 0; j0 < W; ++j0) {
                    for (int j1 = 0; j1 < W; ++j1) {
                        const int row = ((f0 + j0) * e1 + (f1 + j1)) * e2 + f2;
                        const double w01 = v0[j0] * v1[j1];
                        const double a0 = w01 * c0, a1 = w01 * c1;
#pragma unroll
                        for (int j2 = 0; j2 < W; ++j2) {
                            lds_add_fixed(&lds[row + j2], a0, v2[j2]);
                            if (C == 2) lds_add_fixed(&lds[tcells + row + j2], a1, v2[j2]);
                        }
                    }
                }
            }
        }
        __syncthreads();
        // add the tile (with wrap-around) to the global int64 grid
        for (int i = threadIdx.x; i < tcells; i += kSpreadThreads) {
            int l2 = i % e2, l1 = (i / e2) % e1, l0 = i / (e2 * e1);
            int g0 = o[0] + l0, g1 = o[1] + l1, g2 = o[2] + l2;
            if (g0 >= t.nf[0]) g0 -= t.nf[0];
            if (D > 1 && g1 >= t.nf[1]) g1 -= t.nf[1];
            if (D > 2 && g2 >= t.nf[2]) g2 -= t.nf[2];
            const int64_t gi = D == 1 ? g0 : (D == 2 ? (int64_t)g0 * t.nf[1] + g1 : ((int64_t)g0 * t.nf[1] + g1) * t.nf[2] + g2);
            const long long m0 = reinterpret_cast<const long long*>(lds)[i];
            if (m0) atomicAdd(reinterpret_cast<unsigned long long*>(&gacc[gi]), (unsigned long long)m0);
            if (C == 2) {
                const long long m1 = reinterpret_cast<const long long*>(lds)[tcells + i];
                if (m1) atomicAdd(reinterpret_cast<unsigned long long*>(&gacc[a.cells + gi]), (unsigned long long)m1);
            }
        }
        __syncthreads();
        cur = seg_hi;
    }
}

// ------------------------------------------------------------------------------------------
// Register-accumulating spreader over BASE-CELL-SORTED points (2-D).
//   With the points counting-sorted by their first covered cell (tile size 1: the binning above), all
//   points of a bin share the same W x W stencil, so their contributions can be summed in REGISTERS with
//   plain FMAs and flushed once per run -- no LDS atomics in the inner loop (the LDS-resident spreader is
//   bound by ~128 LDS atomics per point, see LABNOTES.md section 4.1).
//   A wavefront takes a contiguous chunk of sorted points and walks the cell runs inside it, 16 points per
//   iteration; all control flow is wave-uniform.  The 4 lanes of a point (one per DPP row: role = lane/16)
//   each own one quadrant of the stencil (rows [0,RH) or the mirrored rows, columns likewise).  The window
//   is even, so polynomial W-1-j at s equals polynomial j at -s: every lane evaluates the SAME first RH
//   Horner polynomials (at +-s), whose coefficients stay resident in VGPRs -- no per-iteration coefficient
//   traffic and no cross-lane exchange in the inner loop.  Coordinates and strengths are staged through a
//   small LDS ring two windows of 64 points ahead (the strength gather is a dependent load), so the inner
//   loop does not wait on global memory.  At the end of a run the 16 partial quadrants of each role are
//   summed over the DPP row and added to the double-precision global grid.
//   Sums are floating point in run order: deterministic for a fixed plan and launch geometry, but not the
//   exact fixed-point arithmetic of the LDS spreaders.
// ------------------------------------------------------------------------------------------
struct CellSpreadArgs {
    TileGeom t;               // T = 1: bins are cells
    const double* xs;
    const int* order;
    const int* start;
    StrengthSrc src;
    const double* coef;       // [kMaxDegree + 1][W], rows above the fitted degree are zero
    int channels;
    double* gacc;             // [batch][channels][cells] doubles, pre-zeroed
    int64_t cells;
    int npts;
    int chunk;                // sorted points per wavefront
};

// 64-bit DPP move: lanes of the banks in BANK_MASK read `v` through the DPP pattern, the others keep `old`
template <int CTRL, int BANK_MASK>
__device__ __forceinline__ double dpp_f64(double old, double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(__double2loint(old), lo, CTRL, 0xF, BANK_MASK, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(old), hi, CTRL, 0xF, BANK_MASK, false);
    return __hiloint2double(hi, lo);
}
// Reduce-scatter over the 16 lanes of a DPP row: every lane holds 16 partial values v[0..15]; on return lane p
// of the row holds the row-wide sum of v[p].  Each step halves the list: a lane keeps the half selected by one
// bit of its lane index and adds the partner's copy of that half (partners: lane^8, mirror within 8, lane^2,
// lane^1).  For the two upper bits the keep/send selection rides on the DPP bank mask (banks = 4 lanes).
__device__ __forceinline__ double reduce_scatter16(double (&v)[16], int pt) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {                     // row_ror:8; lanes 0-7 keep v[i], lanes 8-15 keep v[i+8]
        const double x = dpp_f64<0x128, 0x3>(v[i + 8], v[i]);
        const double y = dpp_f64<0x128, 0xC>(v[i], v[i + 8]);
        v[i] = x + y;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {                     // row_half_mirror; lanes with bit 2 clear keep v[i]
        const double x = dpp_f64<0x141, 0x5>(v[i + 4], v[i]);
        const double y = dpp_f64<0x141, 0xA>(v[i], v[i + 4]);
        v[i] = x + y;
    }
    const bool b1 = (pt & 2) != 0, b0 = (pt & 1) != 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {                     // quad_perm [2,3,0,1]
        const double keep = b1 ? v[i + 2] : v[i], send = b1 ? v[i] : v[i + 2];
        v[i] = keep + dpp_f64<0x4E, 0xF>(send, send);
    }
    const double keep = b0 ? v[1] : v[0], send = b0 ? v[0] : v[1];
    return keep + dpp_f64<0xB1, 0xF>(send, send);      // quad_perm [1,0,3,2]
}

constexpr int kCellWin = 64;                                            // staged points per window (one per lane)
constexpr size_t kCellLds = 4 * 2 * kCellWin * 2 * sizeof(double2);     // 4 waves x 2 windows x (xy, c01)
typedef const __attribute__((address_space(4))) int* const_int_ptr;

template <int W, int C, int DEG>
__global__ __launch_bounds__(256) void spread_cell_kernel(CellSpreadArgs a) {
    constexpr int RH = (W + 1) / 2;                  // polynomials / stencil rows / columns per quadrant
    constexpr bool kOdd = (W & 1) != 0;
    constexpr int kWin = kCellWin, kRing = 2 * kWin;
    const TileGeom& t = a.t;
    const int batch = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int pt = lane & 15, role = lane >> 4;
    const bool ra = (role >> 1) != 0, rb = (role & 1) != 0;     // mirrored rows / mirrored columns
    const int nf0 = t.nf[0], nf1 = t.nf[1];
    double* gacc = a.gacc + (int64_t)batch * C * a.cells;
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    const int lo = wave * a.chunk;
    if (lo >= a.npts) return;
    const int hi = min(lo + a.chunk, a.npts);
    const bool by_index = a.src.mode != STR_ONES;
    const_int_ptr start = (const_int_ptr)a.start;     // written by the binning kernels of an earlier launch
    // Horner coefficients of polynomials 0..RH-1, pinned in VGPRs (the asm keeps the compiler from treating
    // them as wave-uniform scalars, which it would spill and read back lane by lane)
    double cf[DEG + 1][RH];
    {
        const_coef_ptr coef = (const_coef_ptr)a.coef;
#pragma unroll
        for (int k = 0; k <= DEG; ++k)
#pragma unroll
            for (int j = 0; j < RH; ++j) {
                double v = coef[k * W + j];
                asm volatile("" : "+v"(v));
                cf[k][j] = v;
            }
    }
    // staging ring: lane l stages sorted position (window base + l)
    extern __shared__ double2 stage_lds[];
    double2* sXY = stage_lds + (threadIdx.x >> 6) * (2 * kRing);
    double2* sC = sXY + kRing;
    const int nwin = (hi - lo + kWin - 1) / kWin;
    int ordN = 0;
    double x0N = 0.0, x1N = 0.0, c0N = 0.0, c1N = 0.0;
    auto load_ord = [&](int k) __attribute__((always_inline)) {
        const int idx = min(lo + k * kWin + lane, hi - 1);
        ordN = by_index ? a.order[idx] : 0;
    };
    auto load_xc = [&](int k) __attribute__((always_inline)) {
        const int idx = min(lo + k * kWin + lane, hi - 1);
        const double2 xy = reinterpret_cast<const double2*>(a.xs)[idx];
        x0N = xy.x;
        x1N = xy.y;
        fetch_strength(a.src, batch, (int64_t)ordN, c0N, c1N);
    };
    auto store_win = [&](int k) __attribute__((always_inline)) {     // first use of the loaded registers: waits here
        const bool ok = lo + k * kWin + lane < hi;
        const int slot = (k & 1) * kWin + lane;
        sXY[slot] = make_double2(x0N, x1N);
        sC[slot] = make_double2(ok ? c0N : 0.0, ok ? c1N : 0.0);
    };
    // prologue: windows 0 and 1 go to LDS, window 2 stays in flight in registers, indices of window 3 are
    // requested -- issued as two batches of independent loads (indices, then coordinates + strengths)
    {
        int ordA, ordB = 0;
        load_ord(0);
        ordA = ordN;
        if (nwin > 1) {
            load_ord(1);
            ordB = ordN;
        }
        if (nwin > 2) load_ord(2);
        const int ordC = ordN;
        double xa0, xa1, ca0, ca1;
        ordN = ordA;
        load_xc(0);
        xa0 = x0N; xa1 = x1N; ca0 = c0N; ca1 = c1N;
        double xb0 = 0.0, xb1 = 0.0, cb0 = 0.0, cb1 = 0.0;
        if (nwin > 1) {
            ordN = ordB;
            load_xc(1);
            xb0 = x0N; xb1 = x1N; cb0 = c0N; cb1 = c1N;
        }
        if (nwin > 2) {
            ordN = ordC;
            load_xc(2);
        }
        const double xc0 = x0N, xc1 = x1N, cc0 = c0N, cc1 = c1N;
        if (nwin > 3) load_ord(3);
        x0N = xa0; x1N = xa1; c0N = ca0; c1N = ca1;
        store_win(0);
        if (nwin > 1) {
            x0N = xb0; x1N = xb1; c0N = cb0; c1N = cb1;
            store_win(1);
        }
        x0N = xc0; x1N = xc1; c0N = cc0; c1N = cc1;
    }
    int kw = 0;                                        // window being consumed; windows kw, kw+1 are in LDS
    int win_end = min(lo + kWin, hi);
    // cell run that contains sorted position `lo`: the bin of that point, corrected against the run table
    int bin = __builtin_amdgcn_readfirstlane(tile_of_point<2>(t, a.xs, (int64_t)lo));
    bin = min(max(bin, 0), t.nbins - 1);
    while (bin > 0 && start[bin] > lo) --bin;
    const double scale0 = t.scale[0], scale1 = t.scale[1], xcen0 = t.xcen[0], xcen1 = t.xcen[1];
    const double dnf0 = (double)nf0, dnf1 = (double)nf1, inv0 = 1.0 / dnf0, inv1 = 1.0 / dnf1;
    int pos = lo;
    while (pos < hi) {
        while (start[bin + 1] <= pos) ++bin;          // skip empty cells
        const int seg_hi = min(start[bin + 1], hi);
        const int b0 = bin / t.nt[1], b1 = bin - b0 * t.nt[1];          // T = 1: tile index = first covered cell
        const double base0 = (double)b0 + 0.5 * W, base1 = (double)b1 + 0.5 * W;
        double acc0[RH][RH], acc1[RH][RH];
#pragma unroll
        for (int r = 0; r < RH; ++r)
#pragma unroll
            for (int c = 0; c < RH; ++c) {
                acc0[r][c] = 0.0;
                acc1[r][c] = 0.0;
            }
        while (pos < seg_hi) {
            if (pos >= win_end) {                      // advance the staging ring by one window
                ++kw;
                win_end = min(lo + (kw + 1) * kWin, hi);
                if (kw + 1 < nwin) store_win(kw + 1);
                if (kw + 2 < nwin) {
                    load_xc(kw + 2);
                    if (kw + 3 < nwin) load_ord(kw + 3);
                }
            }
            const int take = min(16, seg_hi - pos);
            const bool valid = pt < take;
            const int slot = (pos - lo + (valid ? pt : 0)) & (kRing - 1);
            const double2 xy = sXY[slot];
            double2 cc = sC[slot];
            if (!valid) cc = make_double2(0.0, 0.0);
            pos += take;
            // Horner variables.  u = (first cell - X + W/2) mod nf lies in [0,1); the first covered cell is the
            // run's cell (known from the bin), not re-derived from X, so a rounding difference against the
            // binning pass cannot shift the stencil.
            double t0, t1;
            {
                double u = base0 - scale0 * (xy.x - xcen0);
                u -= dnf0 * rint((u - 0.5) * inv0);
                const double sv = 2.0 * u - 1.0;
                t0 = ra ? -sv : sv;
            }
            {
                double u = base1 - scale1 * (xy.y - xcen1);
                u -= dnf1 * rint((u - 0.5) * inv1);
                const double sv = 2.0 * u - 1.0;
                t1 = rb ? -sv : sv;
            }
            double R[RH], Cw[RH];
#pragma unroll
            for (int j = 0; j < RH; ++j) {
                R[j] = cf[DEG][j];
                Cw[j] = cf[DEG][j];
            }
#pragma unroll
            for (int k = DEG - 1; k >= 0; --k) {
#pragma unroll
                for (int j = 0; j < RH; ++j) {
                    R[j] = fma(R[j], t0, cf[k][j]);
                    Cw[j] = fma(Cw[j], t1, cf[k][j]);
                }
            }
            if (kOdd) {                                // the middle row / column belongs to the unmirrored quadrant
                if (ra) R[RH - 1] = 0.0;
                if (rb) Cw[RH - 1] = 0.0;
            }
#pragma unroll
            for (int r = 0; r < RH; ++r) {
                const double f0 = R[r] * cc.x;
                const double f1 = C == 2 ? R[r] * cc.y : 0.0;
#pragma unroll
                for (int c = 0; c < RH; ++c) {
                    acc0[r][c] = fma(f0, Cw[c], acc0[r][c]);
                    if (C == 2) acc1[r][c] = fma(f1, Cw[c], acc1[r][c]);
                }
            }
        }
        // reduce-scatter the 16 point-lanes of each role: lane p of a role ends with quadrant element p, so that
        // ONE atomic wave-instruction per channel adds the whole W x W stencil (W-element contiguous row pieces)
        {
            static_assert(RH * RH <= 16, "quadrant must fit one DPP row");
            const int er = pt / RH, ec = pt - er * RH;
            bool mine = pt < RH * RH;
            if (kOdd && ((ra && er == RH - 1) || (rb && ec == RH - 1))) mine = false;
            const int g0 = wrap(b0 + (ra ? W - 1 - er : er), nf0), g1 = wrap(b1 + (rb ? W - 1 - ec : ec), nf1);
            const int64_t gi = (int64_t)g0 * nf1 + g1;
            double v[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = e < RH * RH ? acc0[e / RH][e % RH] : 0.0;
            const double s0 = reduce_scatter16(v, pt);
            if (mine) unsafeAtomicAdd(&gacc[gi], s0);
            if (C == 2) {
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = e < RH * RH ? acc1[e / RH][e % RH] : 0.0;
                const double s1 = reduce_scatter16(v, pt);
                if (mine) unsafeAtomicAdd(&gacc[a.cells + gi], s1);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// mode <-> fine-grid index bookkeeping shared by the deconvolve / precorrect kernels
// ------------------------------------------------------------------------------------------
struct ModeGeom {
    int d;
    int64_t nm[3];      // modes per dimension
    int64_t nf[3];
    int64_t total;      // prod nm
    int modeord;        // 0 CMCL, 1 FFT order
    const double* fac[3];   // per-dimension correction factors, indexed by (k - kmin), kmin = -(nm/2)
};

__device__ __forceinline__ int64_t mode_of_slot(int64_t slot, int64_t nm, int modeord) {
    if (modeord == 0) return slot - nm / 2;
    return slot <= (nm - 1) / 2 ? slot : slot - nm;
}

// type 1: out[b][slot] = fac * FFT(fine)[k mod nf]; optional Hermitian split for the (y, ones) pair
//   part = 0: plain;  part = 1: (H[k] + conj(H[-k]))/2;  part = 2: (H[k] - conj(H[-k]))/(2i)
//   part = 3: fine grid g holds real rows (2g, 2g+1): both parts are written, to out rows 2g and 2g+1
__device__ __forceinline__ void deconvolve_body(const double2* __restrict__ fine, int64_t cells, const ModeGeom& m, int part,
                                                double2* __restrict__ out, int batch, int rows_limit = 1 << 30) {
    const double2* F = fine + (int64_t)batch * cells;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < m.total; t += (int64_t)gridDim.x * blockDim.x) {
        int64_t rem = t;
        int64_t idx = 0, idxn = 0;
        double f = 1.0;
        int64_t slots[3];
        for (int a = m.d - 1; a >= 0; --a) {
            slots[a] = rem % m.nm[a];
            rem /= m.nm[a];
        }
        for (int a = 0; a < m.d; ++a) {
            int64_t k = mode_of_slot(slots[a], m.nm[a], m.modeord);
            f *= m.fac[a][k + m.nm[a] / 2];
            int64_t p = k < 0 ? k + m.nf[a] : k;
            int64_t q = k > 0 ? m.nf[a] - k : -k;       // index of -k
            idx = idx * m.nf[a] + p;
            idxn = idxn * m.nf[a] + q;
        }
        double2 H = F[idx];
        double2 r;
        if (part == 0) {
            r = make_double2(H.x * f, H.y * f);
        } else {
            double2 G = F[idxn];        // H[-k]
            const double2 r1 = make_double2(0.5 * (H.x + G.x) * f, 0.5 * (H.y - G.y) * f);
            const double2 r2 = make_double2(0.5 * (H.y + G.y) * f, 0.5 * (G.x - H.x) * f);
            if (part == 3) {
                out[(int64_t)(2 * batch) * m.total + t] = r1;
                if (2 * batch + 1 < rows_limit) out[(int64_t)(2 * batch + 1) * m.total + t] = r2;    // padded odd row count
                continue;
            }
            r = part == 1 ? r1 : r2;
        }
        out[(int64_t)batch * m.total + t] = r;
    }
}
__global__ void deconvolve_kernel(const double2* __restrict__ fine, int64_t cells, ModeGeom m, int part,
                                  double2* __restrict__ out, int rows_limit) {
    deconvolve_body(fine, cells, m, part, out, blockIdx.y, rows_limit);
}
// the fit-time pair in one launch: blockIdx.y = 0 -> part 1 into out_a on box ma, 1 -> part 2 into out_b on box mb
__global__ void deconvolve_pair_kernel(const double2* __restrict__ fine, int64_t cells, ModeGeom ma, double2* __restrict__ out_a,
                                       ModeGeom mb, double2* __restrict__ out_b, const double* __restrict__ scale) {
    if (blockIdx.y == 0) {
        deconvolve_body(fine, cells, ma, 1, out_a, 0);
        const double norm0 = scale ? scale[4] : 1.0;              // channel 0 was carried normalised (fixed_scale_kernel)
        if (norm0 != 1.0) {
            // every thread rescales exactly the elements it wrote (same loop shape as deconvolve_body)
            for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < ma.total; t += (int64_t)gridDim.x * blockDim.x) {
                out_a[t].x *= norm0;
                out_a[t].y *= norm0;
            }
        }
    } else {
        deconvolve_body(fine, cells, mb, 2, out_b, 0);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Small 2-D fine grids straight from the MFMA spreader's int64 accumulator to the modes, ONE launch: fixed-point
// conversion, pruned dense DFT (only the mode box is computed: for n modes out of nf >= 2 n cells that is cheaper than the
// full transform), Hermitian split and correction factors.  Replaces reduce_slabs | FFT rows | FFT columns | deconvolve
// (four dependent 4-5 us launches, 18 us per type-1 transform) and keeps rocFFT -- whose kernels are compiled at run time,
// 0.5-2 s per new length -- off the fit / gradient path of 2-D models with mtot <= 64.
//   H[k0][k1] = sum_x0 w0^(k0 x0) sum_x1 w1^(k1 x1) G[x0][x1],  G = ch0 + i ch1,  w_a = exp(sign 2 pi i / nf_a)
// A workgroup owns the four rows k0 = +-j, +-(j+1) (the split needs H[k] and H[-k] together):
//   stage 1: B[r][x1] = sum_x0 G[x0][x1] w0^(k_r x0): lane = x1 (coalesced int64 loads), eight x0 classes per workgroup;
//   stage 2: H[r][k1] = sum_x1 B[r][x1] w1^(k1 x1) for k1 in [-h1, h1] out of LDS;
//   epilogue: parts as in deconvolve_body (0 plain, 3 real row pairs, 4 = the fit's (F*y, Toeplitz vector) pair on two boxes).
// The accumulator must be left zeroed for the next pass: the LAST workgroup to have finished reading it (arrival counter)
// clears it.  Sums of <= 256 terms with exact table twiddles: the result differs from the FFT's by rounding only.
// Round 3: on the finer grids of dense point sets (144..256 cells per axis) stage 1 -- 45 rows x0 per thread group, each value
// an int64 -> double conversion and eight multiply-adds on 12 of the chip's CUs -- is split over 4 workgroups per row tile (every
// 4th row x0 each); their partial B rows meet in global memory behind a per-tile arrival counter and the last arriver does
// stage 2 and the epilogue.  Small grids keep one workgroup per tile (the exchange costs more than it saves there).
// LOG2NF = 7: grids up to 128 x 128 (lane = x1, eight x0 classes); LOG2NF = 8 (round 3: the finer grids dense point sets take,
// 144..256 cells per axis): 256 lanes per row, four x0 classes, the rows of a class in rounds of 16 loads.
constexpr int kG2MMaxNf = 256;
constexpr int kG2MMaxH = 32;                 // modes k in [-32, 32] per axis
constexpr int kG2MThreads = 1024;

struct G2MArgs {
    long long* gacc;          // [nbatch][channels][nf0 * nf1]
    const double2* fine;      // FINE variant: [nbatch][nf0 * nf1] complex grid already reduced (other spreaders); nothing is cleared
    const double* scale;      // fixed-point block: [1] = 1/S0, [3] = 1/S1, [4] = channel-0 norm
    int channels, nf0, nf1, h0, h1;
    int part;                 // 0, 3, or 4 (pair: part 1 on box ma -> out_a, part 2 on box mb -> out_b)
    int rows_limit;
    int sign;                 // -1 forward, +1 backward
    ModeGeom ma, mb;
    double2* out_a;
    double2* out_b;
    unsigned int* ticket;     // zero on entry, zero again on exit
    unsigned int total_wgs;
    long long acc_words;      // words to clear
    // stage 1 split over gridDim.z workgroups per (tile, batch) (round 3): each takes every gridDim.z-th row x0, writes its
    // partial B rows to `partial` and arrives at the tile's counter; the last one adds the partials up and goes on alone
    double2* partial;         // [nbatch][tiles][split][4][nf1]
    unsigned int* tile_ticket;   // [nbatch][tiles], zero on entry, zero again on exit
};

__device__ __forceinline__ bool g2m_slot(const ModeGeom& m, int k0, int k1, int64_t* t, double* f) {
    const int lo0 = -(int)(m.nm[0] / 2), hi0 = (int)((m.nm[0] - 1) / 2), lo1 = -(int)(m.nm[1] / 2), hi1 = (int)((m.nm[1] - 1) / 2);
    if (k0 < lo0 || k0 > hi0 || k1 < lo1 || k1 > hi1) return false;
    const int64_t s0 = m.modeord == 0 ? k0 - lo0 : (k0 >= 0 ? k0 : k0 + m.nm[0]);
    const int64_t s1 = m.modeord == 0 ? k1 - lo1 : (k1 >= 0 ? k1 : k1 + m.nm[1]);
    *t = s0 * m.nm[1] + s1;
    *f = m.fac[0][k0 - lo0] * m.fac[1][k1 - lo1];
    return true;
}

template <bool FINE, int LOG2NF>
__global__ __launch_bounds__(kG2MThreads) void grid_to_modes_kernel(G2MArgs a) {
    constexpr int kNf = 1 << LOG2NF, kGroups = kG2MThreads / kNf;       // lanes per row, x0 classes of stage 1
    __shared__ double2 T0[kNf], T1[kNf];
    __shared__ double2 Bp[kGroups][4][kNf];                   // 64 KB
    __shared__ double2 Hs[4][2 * kG2MMaxH + 2];
    __shared__ int s_last;
    const int tid = threadIdx.x, b = blockIdx.y;
    const int j0 = 2 * (int)blockIdx.x;
    const int split = (int)blockIdx.z, nsplit = (int)gridDim.z;
    for (int q = tid; q < a.nf0; q += kG2MThreads) {
        double sn, cs;
        sincospi((double)a.sign * 2.0 * (double)q / (double)a.nf0, &sn, &cs);
        T0[q] = make_double2(cs, sn);
    }
    for (int q = tid; q < a.nf1; q += kG2MThreads) {
        double sn, cs;
        sincospi((double)a.sign * 2.0 * (double)q / (double)a.nf1, &sn, &cs);
        T1[q] = make_double2(cs, sn);
    }
    const int64_t cells = (int64_t)a.nf0 * a.nf1;
    const long long* g0 = FINE ? nullptr : a.gacc + (int64_t)b * a.channels * cells;
    const double2* f0 = FINE ? a.fine + (int64_t)b * cells : nullptr;
    const double s0 = FINE ? 1.0 : a.scale[1], s1 = FINE ? 1.0 : a.scale[3];
    // stage 1: lane = x1, group = x0 mod kGroups.  ALL loads of a round (16 rows x 2 channels per thread) are issued before the
    // first is used: the accumulator was written by device-scope atomics and comes from memory, and with two groups and one load
    // per iteration this stage was 48 dependent round trips long (33 us for the 96 x 96 pair grid).
    {
        const int x1 = tid & (kNf - 1), grp = tid >> LOG2NF;
        constexpr int U = 16;
        const bool lane_on = x1 < a.nf1;
        // rows +j and -j share their products: with w0^(j x0) = c + i s and G = re + i im,
        //   B(+j) = (P - S) + i (Q + R),  B(-j) = (P + S) + i (R - Q),  P = sum re c, Q = sum re s, R = sum im c, S = sum im s
        double2 pq[2], rs[2];
        int idx[2], step[2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int k = j0 + jj;
            pq[jj] = rs[jj] = make_double2(0.0, 0.0);
            step[jj] = (int)(((long long)nsplit * kGroups * k) % a.nf0);
            idx[jj] = (int)(((long long)k * (split + nsplit * grp)) % a.nf0);
        }
        for (int r0 = 0; split + nsplit * kGroups * r0 < a.nf0; r0 += U) {
            long long ire[U], iim[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int x0 = split + nsplit * (grp + kGroups * (r0 + u));
                const bool in = lane_on && x0 < a.nf0;
                if (FINE) {
                    const double2 v = in ? f0[(int64_t)x0 * a.nf1 + x1] : make_double2(0.0, 0.0);
                    ire[u] = __double_as_longlong(v.x);
                    iim[u] = __double_as_longlong(v.y);
                } else {
                    ire[u] = in ? g0[(int64_t)x0 * a.nf1 + x1] : 0;
                    iim[u] = (in && a.channels == 2) ? g0[cells + (int64_t)x0 * a.nf1 + x1] : 0;
                }
            }
            if (r0 == 0) __syncthreads();                          // twiddle tables
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double re = FINE ? __longlong_as_double(ire[u]) : (double)ire[u] * s0;
                const double im = FINE ? __longlong_as_double(iim[u]) : (double)iim[u] * s1;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const double2 tw = T0[idx[jj]];
                    pq[jj].x += re * tw.x;
                    pq[jj].y += re * tw.y;
                    rs[jj].x += im * tw.x;
                    rs[jj].y += im * tw.y;
                    idx[jj] += step[jj];
                    if (idx[jj] >= a.nf0) idx[jj] -= a.nf0;
                }
            }
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            Bp[grp][2 * jj][x1] = pq[jj];
            Bp[grp][2 * jj + 1][x1] = rs[jj];
        }
    }
    __syncthreads();
    // every load of the accumulator by this workgroup has been consumed: arrive (the answer is looked at after stage 2);
    // the last workgroup to arrive clears the accumulator
    unsigned int arrived = 0u;
    if (!FINE && tid == 0) arrived = atomicAdd(a.ticket, 1u);          // no fence: the loads were consumed (barrier above), nothing was stored
    for (int o = tid; o < 2 * a.nf1; o += kG2MThreads) {
        const int jj = o / a.nf1, x = o - jj * a.nf1;
        double2 u = Bp[0][2 * jj][x], v = Bp[0][2 * jj + 1][x];
#pragma unroll
        for (int g = 1; g < kGroups; ++g) {
            u.x += Bp[g][2 * jj][x].x;
            u.y += Bp[g][2 * jj][x].y;
            v.x += Bp[g][2 * jj + 1][x].x;
            v.y += Bp[g][2 * jj + 1][x].y;
        }
        Bp[0][2 * jj][x] = make_double2(u.x - v.y, u.y + v.x);          // row +j
        Bp[0][2 * jj + 1][x] = make_double2(u.x + v.y, v.x - u.y);      // row -j
    }
    __syncthreads();
    if (nsplit > 1) {
        // this workgroup's share of the four B rows goes to global memory; the last of the tile's workgroups to arrive adds all
        // shares up (in split order: the sum does not depend on who arrives last) and carries on, the others are done
        double2* const mine = a.partial + ((((int64_t)b * gridDim.x + blockIdx.x) * nsplit + split) * 4) * a.nf1;
        for (int o = tid; o < 4 * a.nf1; o += kG2MThreads) mine[o] = Bp[0][o / a.nf1][o % a.nf1];
        __threadfence();                                               // release: the shares before the arrival
        __syncthreads();
        if (tid == 0) {
            const unsigned int t = atomicAdd(a.tile_ticket + (int64_t)b * gridDim.x + blockIdx.x, 1u);
            s_last = (t == (unsigned)nsplit - 1u) ? 1 : 0;
        }
        __syncthreads();
        const bool tile_last = s_last != 0;
        __syncthreads();                                               // s_last is reused below
        if (!tile_last) {
            // not the tile's finisher; still possibly the last reader of the accumulator
            if (tid == 0) s_last = (!FINE && arrived == a.total_wgs - 1u) ? 1 : 0;
            __syncthreads();
            if (s_last) {
                for (long long i = tid; i < a.acc_words; i += kG2MThreads) a.gacc[i] = 0;
                if (tid == 0) *a.ticket = 0u;
            }
            return;
        }
        __threadfence();                                               // acquire: the other workgroups' shares
        const double2* const all = a.partial + (((int64_t)b * gridDim.x + blockIdx.x) * nsplit * 4) * a.nf1;
        for (int o = tid; o < 4 * a.nf1; o += kG2MThreads) {
            double2 u = all[o];
            for (int sp = 1; sp < nsplit; ++sp) {
                const double2 v = all[(int64_t)sp * 4 * a.nf1 + o];
                u.x += v.x;
                u.y += v.y;
            }
            Bp[0][o / a.nf1][o % a.nf1] = u;
        }
        if (tid == 0) a.tile_ticket[(int64_t)b * gridDim.x + blockIdx.x] = 0u;
        __syncthreads();
    }
    // stage 2: four quarter ranges of x1 per output, combined through LDS
    const int nk1 = 2 * a.h1 + 1;
    double2* const Hq = &Bp[1][0][0];                          // [4 quarters][4 rows][nk1 <= 65]: Bp[1..] is free now
    for (int o = tid; o < 16 * nk1; o += kG2MThreads) {
        const int qtr = o / (4 * nk1), rc = o - qtr * 4 * nk1;
        const int r = rc / nk1, c = rc - r * nk1;
        int st = (c - a.h1) % a.nf1;
        if (st < 0) st += a.nf1;
        const int xa = (a.nf1 * qtr) / 4, xe = (a.nf1 * (qtr + 1)) / 4;
        int idx = (int)(((long long)st * xa) % a.nf1);
        double sx = 0.0, sy = 0.0;
        for (int x = xa; x < xe; ++x) {
            const double2 bv = Bp[0][r][x], tw = T1[idx];
            sx += bv.x * tw.x - bv.y * tw.y;
            sy += bv.x * tw.y + bv.y * tw.x;
            idx += st;
            if (idx >= a.nf1) idx -= a.nf1;
        }
        Hq[(qtr * 4 + r) * (2 * kG2MMaxH + 2) + c] = make_double2(sx, sy);
    }
    if (tid == 0) s_last = (!FINE && arrived == a.total_wgs - 1u) ? 1 : 0;        // the counter's answer has had stage 2 to come back
    __syncthreads();
    if (s_last) {                                                      // these stores drain behind the epilogue
        for (long long i = tid; i < a.acc_words; i += kG2MThreads) a.gacc[i] = 0;
        if (tid == 0) *a.ticket = 0u;
    }
    for (int o = tid; o < 4 * nk1; o += kG2MThreads) {
        const int r = o / nk1, c = o - r * nk1;
        double2 u = Hq[r * (2 * kG2MMaxH + 2) + c];
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            const double2 v = Hq[(q * 4 + r) * (2 * kG2MMaxH + 2) + c];
            u.x += v.x;
            u.y += v.y;
        }
        Hs[r][c] = u;
    }
    __syncthreads();
    // epilogue
    for (int o = tid; o < 4 * nk1; o += kG2MThreads) {
        const int r = o / nk1, c = o - r * nk1;
        const int j = j0 + (r >> 1);
        if (j > a.h0 || (j == 0 && (r & 1))) continue;            // beyond the box; -0 duplicates +0
        const int k0 = (r & 1) ? -j : j, k1 = c - a.h1;
        const double2 H = Hs[r][c], G = Hs[r ^ 1][nk1 - 1 - c];    // modes k and -k
        int64_t t;
        double f;
        if (a.part == 0) {
            if (g2m_slot(a.ma, k0, k1, &t, &f)) a.out_a[(int64_t)b * a.ma.total + t] = make_double2(H.x * f, H.y * f);
            continue;
        }
        if (g2m_slot(a.ma, k0, k1, &t, &f)) {
            double2 r1 = make_double2(0.5 * (H.x + G.x) * f, 0.5 * (H.y - G.y) * f);
            if (a.part == 3) {
                a.out_a[(int64_t)(2 * b) * a.ma.total + t] = r1;
                if (2 * b + 1 < a.rows_limit)
                    a.out_a[(int64_t)(2 * b + 1) * a.ma.total + t] = make_double2(0.5 * (H.y + G.y) * f, 0.5 * (G.x - H.x) * f);
            } else {
                const double norm0 = a.scale ? a.scale[4] : 1.0;  // channel 0 was carried normalised (fixed_scale_kernel)
                if (norm0 != 1.0) {
                    r1.x *= norm0;
                    r1.y *= norm0;
                }
                a.out_a[t] = r1;
            }
        }
        if (a.part == 4 && g2m_slot(a.mb, k0, k1, &t, &f)) a.out_b[t] = make_double2(0.5 * (H.y + G.y) * f, 0.5 * (G.x - H.x) * f);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Round 3: the same accumulator -> modes step as TWO launches that use the whole chip.
// grid_to_modes_kernel gives every workgroup four mode rows k0 and makes it pull the WHOLE accumulator through one CU
// (12 workgroups for the 45-mode box): 20 us on the 96 x 96 pair grid, 47-60 us on the 180 x 180 grid of a dense point set.
// A one-launch variant with the x1 transform spread over the rows and the whole x0 transform done by the last workgroup to
// arrive measured 30 / 51 us: 45 x 45 x nf0 complex multiply-adds are 5-10 us of ONE CU's fp64 pipe, and data freshly written
// by other XCDs costs a microsecond per dependent round trip.  So:
//   grid_rows_kernel    (workgroup = kG2RRows rows x0 of the accumulator): C[x0][k1] = sum_x1 G[x0][x1] w1^(k1 x1) for
//                       k1 in [-h1, h1]; only its own rows are loaded (converted from the int64 fixed point, and cleared for
//                       the next pass: no arrival counter); the sums for +k1 and -k1 share their four real products;
//   rows_modes_kernel   (workgroup = the column pair +-k1): H[k0][+-k1] = sum_x0 C[x0][+-k1] w0^(k0 x0) for all k0 out of LDS
//                       (two columns of C = 2 nf0 values, every load in flight at once), then the epilogue of
//                       grid_to_modes_kernel (parts 0 | 3 | 4): the modes k and -k it pairs are both in this workgroup.
// Sums of <= 256 terms with exact table twiddles, fixed order: equal to the FFT sequence to rounding, reproducible.
constexpr int kG2RRows = 4;
constexpr int kG2RThreads = 1024;
constexpr int kR2MThreads = 256;

struct G2RArgs {
    long long* gacc;          // [nbatch][channels][nf0 * nf1]   (null: `fine`)
    const double2* fine;      // [nbatch][nf0 * nf1] complex grid already reduced (other spreaders); nothing is cleared
    const double* scale;
    int channels, nf0, nf1, h0, h1;
    int part, rows_limit, sign;
    ModeGeom ma, mb;
    double2* out_a;
    double2* out_b;
    double2* cbuf;            // [nbatch][nf0][2 h1 + 1]
};

template <bool FINE>
__global__ __launch_bounds__(kG2RThreads) void grid_rows_kernel(G2RArgs a) {
    extern __shared__ double2 g2r_lds[];
    const int tid = threadIdx.x, b = blockIdx.y;
    const int nf0 = a.nf0, nf1 = a.nf1, h1 = a.h1;
    const int nk1 = 2 * h1 + 1;
    const int64_t cells = (int64_t)nf0 * nf1;
    double2* const T = g2r_lds;                         // w1^q, q < nf1 <= 256
    double2* const Row = T + 256;                       // [kG2RRows][nf1] (re, im) of the owned rows
    for (int q = tid; q < nf1; q += kG2RThreads) {
        double sn, cs;
        sincospi((double)a.sign * 2.0 * (double)q / (double)nf1, &sn, &cs);
        T[q] = make_double2(cs, sn);
    }
    const int x0_lo = (int)blockIdx.x * kG2RRows;
    {
        const double s0 = FINE ? 1.0 : a.scale[1], s1 = FINE ? 1.0 : a.scale[3];
        for (int o = tid; o < kG2RRows * nf1; o += kG2RThreads) {
            const int r = o / nf1, x1 = o - r * nf1, x0 = x0_lo + r;
            double2 v = make_double2(0.0, 0.0);
            if (x0 < nf0) {
                if (FINE) {
                    v = a.fine[(int64_t)b * cells + (int64_t)x0 * nf1 + x1];
                } else {
                    long long* g = a.gacc + (int64_t)b * a.channels * cells + (int64_t)x0 * nf1 + x1;
                    const long long ire = g[0];
                    g[0] = 0;                                       // the next pass finds a zeroed accumulator
                    long long iim = 0;
                    if (a.channels == 2) {
                        iim = g[cells];
                        g[cells] = 0;
                    }
                    v = make_double2((double)ire * s0, (double)iim * s1);
                }
            }
            Row[o] = v;
        }
    }
    __syncthreads();
    // task = (row r, k1 >= 0, segment of x1): eight segments per (r, k1) in neighbouring lanes, combined by shuffles
    constexpr int SEG = 8;
    const int seg = tid & (SEG - 1), pair = tid >> 3;            // pair < 128
    const int npairs = kG2RRows * (h1 + 1);
    for (int pr = pair; pr < ((npairs + 127) / 128) * 128; pr += 128) {
        const bool on = pr < npairs;
        const int r = on ? pr / (h1 + 1) : 0, k1 = on ? pr - r * (h1 + 1) : 0;
        const int xa = (nf1 * seg) / SEG, xe = (nf1 * (seg + 1)) / SEG;
        int idx = (int)(((long long)k1 * xa) % nf1);
        double P = 0.0, Q = 0.0, R = 0.0, S = 0.0;
        if (on) {
            const double2* row = Row + r * nf1;
            for (int x = xa; x < xe; ++x) {
                const double2 g = row[x], tw = T[idx];
                P = fma(g.x, tw.x, P);
                Q = fma(g.x, tw.y, Q);
                R = fma(g.y, tw.x, R);
                S = fma(g.y, tw.y, S);
                idx += k1;
                if (idx >= nf1) idx -= nf1;
            }
        }
#pragma unroll
        for (int m = 1; m < SEG; m <<= 1) {
            P += __shfl_xor(P, m, 64);
            Q += __shfl_xor(Q, m, 64);
            R += __shfl_xor(R, m, 64);
            S += __shfl_xor(S, m, 64);
        }
        const int x0 = x0_lo + r;
        if (on && seg == 0 && x0 < nf0) {
            double2* c = a.cbuf + ((int64_t)b * nf0 + x0) * nk1;
            c[h1 + k1] = make_double2(P - S, Q + R);                       // sum (re + i im)(c + i s)
            if (k1 > 0) c[h1 - k1] = make_double2(P + S, R - Q);           // ... (c - i s)
        }
    }
}

__global__ __launch_bounds__(kR2MThreads) void rows_modes_kernel(G2RArgs a) {
    __shared__ double2 T[256], Cc[2][256], Hs[2][2 * kG2MMaxH + 2], part[2][4][2 * kG2MMaxH + 2];
    const int tid = threadIdx.x, b = blockIdx.y, k1 = (int)blockIdx.x;             // column pair +-k1, k1 = 0..h1
    const int nf0 = a.nf0, h0 = a.h0, h1 = a.h1;
    const int nk0 = 2 * h0 + 1, nk1 = 2 * h1 + 1;
    const double2* C = a.cbuf + (int64_t)b * nf0 * nk1;
    for (int o = tid; o < 2 * nf0; o += kR2MThreads) {                           // every load issued before any is used
        const int col = o >= nf0 ? 1 : 0, x0 = o - col * nf0;
        Cc[col][x0] = C[(int64_t)x0 * nk1 + (col ? h1 - k1 : h1 + k1)];
    }
    for (int q = tid; q < nf0; q += kR2MThreads) {
        double sn, cs;
        sincospi((double)a.sign * 2.0 * (double)q / (double)nf0, &sn, &cs);
        T[q] = make_double2(cs, sn);
    }
    __syncthreads();
    // H[k0][col] = sum_x0 Cc[col][x0] w0^(k0 x0): task = (col, k0, quarter of x0)
    for (int o = tid; o < 8 * nk0; o += kR2MThreads) {
        const int qtr = o / (2 * nk0), rc = o - qtr * 2 * nk0;
        const int col = rc / nk0, r = rc - col * nk0;
        int st = (r - h0) % nf0;
        if (st < 0) st += nf0;
        const int xa = (nf0 * qtr) / 4, xe = (nf0 * (qtr + 1)) / 4;
        int idx = (int)(((long long)st * xa) % nf0);
        double sx = 0.0, sy = 0.0;
        for (int x = xa; x < xe; ++x) {
            const double2 cv = Cc[col][x], tw = T[idx];
            sx = fma(cv.x, tw.x, fma(-cv.y, tw.y, sx));
            sy = fma(cv.x, tw.y, fma(cv.y, tw.x, sy));
            idx += st;
            if (idx >= nf0) idx -= nf0;
        }
        part[col][qtr][r] = make_double2(sx, sy);
    }
    __syncthreads();
    for (int o = tid; o < 2 * nk0; o += kR2MThreads) {
        const int col = o / nk0, r = o - col * nk0;
        Hs[col][r] = make_double2((part[col][0][r].x + part[col][1][r].x) + (part[col][2][r].x + part[col][3][r].x),
                                  (part[col][0][r].y + part[col][1][r].y) + (part[col][2][r].y + part[col][3][r].y));
    }
    __syncthreads();
    // epilogue (as grid_to_modes_kernel): mode (k0, +-k1) with its partner (-k0, -+k1) in the other column
    for (int o = tid; o < 2 * nk0; o += kR2MThreads) {
        const int col = o / nk0, r = o - col * nk0;
        if (col == 1 && k1 == 0) continue;                                       // -0 duplicates +0
        const int k0 = r - h0, kk1 = col ? -k1 : k1;
        const double2 H = Hs[col][r], G = Hs[k1 == 0 ? 0 : 1 - col][nk0 - 1 - r];
        int64_t t;
        double f;
        if (a.part == 0) {
            if (g2m_slot(a.ma, k0, kk1, &t, &f)) a.out_a[(int64_t)b * a.ma.total + t] = make_double2(H.x * f, H.y * f);
            continue;
        }
        if (g2m_slot(a.ma, k0, kk1, &t, &f)) {
            double2 r1 = make_double2(0.5 * (H.x + G.x) * f, 0.5 * (H.y - G.y) * f);
            if (a.part == 3) {
                a.out_a[(int64_t)(2 * b) * a.ma.total + t] = r1;
                if (2 * b + 1 < a.rows_limit)
                    a.out_a[(int64_t)(2 * b + 1) * a.ma.total + t] = make_double2(0.5 * (H.y + G.y) * f, 0.5 * (G.x - H.x) * f);
            } else {
                const double norm0 = a.scale ? a.scale[4] : 1.0;  // channel 0 was carried normalised (fixed_scale_kernel)
                if (norm0 != 1.0) {
                    r1.x *= norm0;
                    r1.y *= norm0;
                }
                a.out_a[t] = r1;
            }
        }
        if (a.part == 4 && g2m_slot(a.mb, k0, kk1, &t, &f)) a.out_b[t] = make_double2(0.5 * (H.y + G.y) * f, 0.5 * (G.x - H.x) * f);
    }
}

// type 2: fine[b][k mod nf] = fac * f[b][slot] (* mul[slot] when given), zero outside the mode box: every
// fine cell is written, so the grid needs no memset.  herm != 0 stores the Hermitian part
// (f[k] + conj f[-k])/2, whose transform is the real part of the full one (real_only outputs).
__global__ void precorrect_kernel(const double2* __restrict__ fin, const double2* __restrict__ mul, ModeGeom m, int herm,
                                  int64_t cells, double2* __restrict__ fine) {
    const int batch = blockIdx.y;
    const double2* fb = fin + (int64_t)batch * m.total;
    double2* F = fine + (int64_t)batch * cells;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < cells; c += (int64_t)gridDim.x * blockDim.x) {
        int64_t rem = c;
        int64_t ks[3] = {0, 0, 0};
        bool inside = true;
        for (int a = m.d - 1; a >= 0; --a) {
            const int64_t i = rem % m.nf[a];
            rem /= m.nf[a];
            const int64_t k = i <= (m.nf[a] - 1) / 2 ? i : i - m.nf[a];
            ks[a] = k;
            inside = inside && k >= -(m.nm[a] / 2) && k <= (m.nm[a] - 1) / 2;
        }
        double2 r = make_double2(0.0, 0.0);
        if (inside) {
            int64_t t = 0, tn = 0;
            bool has_neg = true;
            double f = 1.0;
            for (int a = 0; a < m.d; ++a) {
                const int64_t k = ks[a], kmin = -(m.nm[a] / 2), kmax = (m.nm[a] - 1) / 2;
                f *= m.fac[a][k - kmin];
                t = t * m.nm[a] + (m.modeord == 0 ? k - kmin : (k >= 0 ? k : k + m.nm[a]));
                const int64_t kn = -k;                       // slot of -k inside the mode box (absent for even sizes)
                if (kn < kmin || kn > kmax) has_neg = false;
                tn = tn * m.nm[a] + (m.modeord == 0 ? kn - kmin : (kn >= 0 ? kn : kn + m.nm[a]));
            }
            double2 v = fb[t];
            if (mul) v = make_double2(v.x * mul[t].x - v.y * mul[t].y, v.x * mul[t].y + v.y * mul[t].x);
            // a mode without partner (k = -nm/2 of an even box) stays as it is: the gather reads only the real part of
            // the transformed grid, and Re(f e^{ikx}) is that mode's whole contribution to the real part
            if (herm && has_neg) {
                double2 u = fb[tn];
                if (mul) u = make_double2(u.x * mul[tn].x - u.y * mul[tn].y, u.x * mul[tn].y + u.y * mul[tn].x);
                v = make_double2(0.5 * (v.x + u.x), 0.5 * (v.y - u.y));
            }
            r = make_double2(v.x * f, v.y * f);
        }
        F[c] = r;
    }
}

// ------------------------------------------------------------------------------------------
// type-2 interpolation.  One thread per point; grid = (workgroups, nbatch).
//   CPLX: fine grid values are complex (else only the real part is used / produced).
//   USE_LDS: the batch row's fine grid is first copied to LDS.
// ------------------------------------------------------------------------------------------
struct InterpArgs {
    const double* x;
    int64_t npts;
    GridGeom g;
    const double* coef;
    int degree;
    const double2* fine;     // [batch][cells]
    void* out;               // [batch][npts] complex or real
    const int* order;        // interp_real_halo_kernel: bank-balanced processing order over global 4096-point windows (or null)
};

template <int D, int W, bool CPLX, bool USE_LDS>
__global__ __launch_bounds__(USE_LDS ? kInterpThreads : kInterpThreadsGlobal) void interp_kernel(InterpArgs a) {
    constexpr int kThr = USE_LDS ? kInterpThreads : kInterpThreadsGlobal;
    extern __shared__ double lds[];
    const int batch = blockIdx.y;
    const int64_t cells = a.g.cells;
    const double2* F = a.fine + (int64_t)batch * cells;
    if (USE_LDS) {
        if (CPLX) {
            double2* l2 = reinterpret_cast<double2*>(lds);
            for (int64_t i = threadIdx.x; i < cells; i += kThr) l2[i] = F[i];
        } else {
            for (int64_t i = threadIdx.x; i < cells; i += kThr) lds[i] = F[i].x;
        }
        __syncthreads();
    }
    const int nf0 = (int)a.g.nf[0], nf1 = (int)a.g.nf[1], nf2 = (int)a.g.nf[2];
    auto load = [&](int64_t i, double& re, double& im) {
        if (USE_LDS) {
            if (CPLX) {
                double2 v = reinterpret_cast<const double2*>(lds)[i];
                re = v.x;
                im = v.y;
            } else {
                re = lds[i];
                im = 0.0;
            }
        } else {
            if (CPLX) {
                double2 v = F[i];
                re = v.x;
                im = v.y;
            } else {
                re = F[i].x;
                im = 0.0;
            }
        }
    };
    for (int64_t n = (int64_t)blockIdx.x * kThr + threadIdx.x; n < a.npts;
         n += (int64_t)gridDim.x * kThr) {
        double v0[W], v1[W], v2[W];
        int f0 = 0, f1 = 0, f2 = 0;
        {
            double Xw[3] = {0.0, 0.0, 0.0};
            Xw[0] = fold(a.g.scale[0] * (a.x[n * D + 0] - a.g.xcen[0]), (double)nf0);
            if (D > 1) Xw[1] = fold(a.g.scale[1] * (a.x[n * D + 1] - a.g.xcen[1]), (double)nf1);
            if (D > 2) Xw[2] = fold(a.g.scale[2] * (a.x[n * D + 2] - a.g.xcen[2]), (double)nf2);
            window_eval<D, W>(a.coef, a.degree, Xw, nf0, nf1, nf2, f0, f1, f2, v0, v1, v2);
        }
        double sre = 0.0, sim = 0.0;
        if (D == 1) {
#pragma unroll
            for (int j = 0; j < W; ++j) {
                double re, im;
                load(wrap(f0 + j, nf0), re, im);
                sre = fma(v0[j], re, sre);
                if (CPLX) sim = fma(v0[j], im, sim);
            }
        } else if (D == 2) {
#pragma unroll
            for (int j0 = 0; j0 < W; ++j0) {
                const int row = wrap(f0 + j0, nf0) * nf1;
                double rre = 0.0, rim = 0.0;
#pragma unroll
                for (int j1 = 0; j1 < W; ++j1) {
                    double re, im;
                    load(row + wrap(f1 + j1, nf1), re, im);
                    rre = fma(v1[j1], re, rre);
                    if (CPLX) rim = fma(v1[j1], im, rim);
                }
                sre = fma(v0[j0], rre, sre);
                if (CPLX) sim = fma(v0[j0], rim, sim);
            }
        } else {
            for (int j0 = 0; j0 < W; ++j0) {
                const int64_t p0 = (int64_t)wrap(f0 + j0, nf0) * nf1;
                double are = 0.0, aim = 0.0;
                for (int j1 = 0; j1 < W; ++j1) {
                    const int64_t p1 = (p0 + wrap(f1 + j1, nf1)) * nf2;
                    double rre = 0.0, rim = 0.0;
#pragma unroll
                    for (int j2 = 0; j2 < W; ++j2) {
                        double re, im;
                        load(p1 + wrap(f2 + j2, nf2), re, im);
                        rre = fma(v2[j2], re, rre);
                        if (CPLX) rim = fma(v2[j2], im, rim);
                    }
                    are = fma(v1[j1], rre, are);
                    if (CPLX) aim = fma(v1[j1], rim, aim);
                }
                sre = fma(v0[j0], are, sre);
                if (CPLX) sim = fma(v0[j0], aim, sim);
            }
        }
        if (CPLX) reinterpret_cast<double2*>(a.out)[(int64_t)batch * a.npts + n] = make_double2(sre, sim);
        else reinterpret_cast<double*>(a.out)[(int64_t)batch * a.npts + n] = sre;
    }
}

// Real-output interpolation with a HALO-PADDED LDS copy of the real fine grid: W-1 wrap-around rows /
// columns are duplicated behind the grid, so a point's W^d cells are `base + constant offsets` -- no wrap
// arithmetic and all W loads of a row are issued back to back before they are consumed.
template <int D, int W>
__global__ __launch_bounds__(kInterpThreads) void interp_real_halo_kernel(InterpArgs a) {
    extern __shared__ double lds[];
    const int batch = blockIdx.y;
    const double2* F = a.fine + (int64_t)batch * a.g.cells;
    const int nf0 = (int)a.g.nf[0], nf1 = (int)a.g.nf[1], nf2 = (int)a.g.nf[2];
    const int p0 = nf0 + W - 1, p1 = D > 1 ? nf1 + W - 1 : 1, p2 = D > 2 ? nf2 + W - 1 : 1;
    const int total = p0 * p1 * p2;
    for (int i = threadIdx.x; i < total; i += kInterpThreads) {
        int i2 = i % p2, i1 = (i / p2) % p1, i0 = i / (p2 * p1);
        if (i0 >= nf0) i0 -= nf0;
        if (D > 1 && i1 >= nf1) i1 -= nf1;
        if (D > 2 && i2 >= nf2) i2 -= nf2;
        const int64_t gi = D == 1 ? i0 : (D == 2 ? (int64_t)i0 * nf1 + i1 : ((int64_t)i0 * nf1 + i1) * nf2 + i2);
        lds[i] = F[gi].x;
    }
    __syncthreads();
    double* out = reinterpret_cast<double*>(a.out) + (int64_t)batch * a.npts;
    // The gather is bound by LDS read bank conflicts (PMC at N=1e7: the LDS array busy 88 % of the time, 67 % of those
    // cycles conflicts): with a per-plan order that puts 16 different column classes into every 16 consecutive
    // positions (class_order_kernel, global windows) the lanes of a read group hit different banks.
    for (int64_t pos = (int64_t)blockIdx.x * kInterpThreads + threadIdx.x; pos < a.npts;
         pos += (int64_t)gridDim.x * kInterpThreads) {
        const int64_t n = a.order ? (int64_t)a.order[pos] : pos;
        double v0[W], v1[W], v2[W];
        int f0 = 0, f1 = 0, f2 = 0;
        {
            double Xw[3] = {0.0, 0.0, 0.0};
            Xw[0] = fold(a.g.scale[0] * (a.x[n * D + 0] - a.g.xcen[0]), (double)nf0);
            if (D > 1) Xw[1] = fold(a.g.scale[1] * (a.x[n * D + 1] - a.g.xcen[1]), (double)nf1);
            if (D > 2) Xw[2] = fold(a.g.scale[2] * (a.x[n * D + 2] - a.g.xcen[2]), (double)nf2);
            window_eval<D, W>(a.coef, a.degree, Xw, nf0, nf1, nf2, f0, f1, f2, v0, v1, v2);
        }
        double acc = 0.0;
        if (D == 1) {
            const double* p = lds + f0;
            double c[W];
#pragma unroll
            for (int j = 0; j < W; ++j) c[j] = p[j];
#pragma unroll
            for (int j = 0; j < W; ++j) acc = fma(v0[j], c[j], acc);
        } else if (D == 2) {
            const double* p = lds + f0 * p1 + f1;
#pragma unroll
            for (int j0 = 0; j0 < W; ++j0) {
                double c[W];
#pragma unroll
                for (int j1 = 0; j1 < W; ++j1) c[j1] = p[j0 * p1 + j1];
                double r = 0.0;
#pragma unroll
                for (int j1 = 0; j1 < W; ++j1) r = fma(v1[j1], c[j1], r);
                acc = fma(v0[j0], r, acc);
            }
        } else {
            for (int j0 = 0; j0 < W; ++j0) {
                double r0 = 0.0;
                for (int j1 = 0; j1 < W; ++j1) {
                    const double* p = lds + ((f0 + j0) * p1 + (f1 + j1)) * p2 + f2;
                    double c[W];
#pragma unroll
                    for (int j2 = 0; j2 < W; ++j2) c[j2] = p[j2];
                    double r = 0.0;
#pragma unroll
                    for (int j2 = 0; j2 < W; ++j2) r = fma(v2[j2], c[j2], r);
                    r0 = fma(v1[j1], r, r0);
                }
                acc = fma(v0[j0], r0, acc);
            }
        }
        out[n] = acc;
    }
}

// 2-D real-output gather from TWO halo-padded LDS copies of the real fine grid: copy 1 is copy 0 shifted left by one
// column, so a stencil row starting at an odd column is read from copy 1 at the even column below it and every row
// read is a 16-byte-aligned ds_read_b128 (two cells per lane-instruction).  The single-copy kernel has to use
// ds_read2_b64 (8-byte alignment): twice the LDS cycles per byte (MI355X_MICROARCH section LDS), and the gather is
// LDS-bound (PMC, round 1).  No processing order is needed: points are streamed as the caller holds them, so x
// loads and result stores stay coalesced and no per-plan ordering pass exists.
template <int W>
__global__ __launch_bounds__(kInterpThreads) void interp_real2_pair_kernel(InterpArgs a) {
    extern __shared__ double lds[];
    constexpr int WP = (W + 1) / 2;                  // 16-byte pairs per stencil row
    const int batch = blockIdx.y;
    const double2* F = a.fine + (int64_t)batch * a.g.cells;
    const int nf0 = (int)a.g.nf[0], nf1 = (int)a.g.nf[1];
    const int p0 = nf0 + W - 1, p1 = (nf1 + 2 * WP + 1) & ~1;            // even row pitch, room for the pair overhang
    const int plane = p0 * p1;
    for (int i = threadIdx.x; i < 2 * plane; i += kInterpThreads) {
        const int cp = i >= plane ? 1 : 0;
        const int r = (i - cp * plane) / p1, c = (i - cp * plane) - r * p1 + cp;      // copy 1 holds column c + 1
        int i0 = r >= nf0 ? r - nf0 : r;
        int i1 = c % nf1;
        lds[i] = F[(int64_t)i0 * nf1 + i1].x;
    }
    __syncthreads();
    double* out = reinterpret_cast<double*>(a.out) + (int64_t)batch * a.npts;
    for (int64_t n = (int64_t)blockIdx.x * kInterpThreads + threadIdx.x; n < a.npts; n += (int64_t)gridDim.x * kInterpThreads) {
        double v0[W], v1[W], v2[W];
        int f0 = 0, f1 = 0, f2 = 0;
        {
            const double2 xy = reinterpret_cast<const double2*>(a.x)[n];
            double Xw[3] = {0.0, 0.0, 0.0};
            Xw[0] = fold(a.g.scale[0] * (xy.x - a.g.xcen[0]), (double)nf0);
            Xw[1] = fold(a.g.scale[1] * (xy.y - a.g.xcen[1]), (double)nf1);
            window_eval<2, W>(a.coef, a.degree, Xw, nf0, nf1, 1, f0, f1, f2, v0, v1, v2);
        }
        const double2* p = reinterpret_cast<const double2*>(lds + (f1 & 1) * plane + f0 * p1 + (f1 & ~1));
        double acc = 0.0;
#pragma unroll
        for (int j0 = 0; j0 < W; ++j0) {
            double2 c[WP];
#pragma unroll
            for (int k = 0; k < WP; ++k) c[k] = p[j0 * (p1 / 2) + k];
            double r = 0.0;
#pragma unroll
            for (int k = 0; k < WP; ++k) {
                r = fma(v1[2 * k], c[k].x, r);
                if (2 * k + 1 < W) r = fma(v1[2 * k + 1], c[k].y, r);
            }
            acc = fma(v0[j0], r, acc);
        }
        out[n] = acc;
    }
}

// (Round 3, measured and dropped: ONE halo-padded copy read with 8-byte ds_read_b64 -- same bytes per LDS cycle as ds_read_b128, no
// alignment copy, so two workgroups per CU where the two copies of a 72 x 72 grid leave room for one.  N = 1e7, W = 7: 178 us
// against 130 us for the two-copy kernel below: 49 read instructions per point instead of 28 fill the LDS command queue, and
// 32 lanes spread over 32 bank pairs collide more often than 16 lanes over 16 quads.)
// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct WindowSet {          // device copies of the window data for one (tolerance, sigma, nf, n_modes) setting
    double tol = 0.0;
    int dim = 0;
    bool dense = false;     // es_fine_size's rule for dense 2-D point sets
    EsParams p;
    double* d_block = nullptr;      // ONE pooled device block behind d_coef and d_fac (round 4: one upload, no hipMalloc per set)
    size_t block_bytes = 0;
    DeviceCtx* ctx = nullptr;
    hipStream_t up_stream = nullptr;   // the upload is ordered on this stream; other streams wait for `ready`
    hipEvent_t ready = nullptr;
    double* d_coef = nullptr;       // [kMaxDegree+1][W], rows above `degree` are zero
    double* d_fac[3] = {nullptr, nullptr, nullptr};
    int64_t nm[3] = {0, 0, 0};
    int64_t nf[3] = {0, 0, 0};
};

constexpr int64_t kDensePoints = 4000000;

struct ClassOrder {         // see class_order_kernel
    int64_t nf[3];
    int W;
    int nwg;
    int* order = nullptr;
    size_t bytes = 0;
};

struct BinSet {             // points counting-sorted by fine-grid tile (see spread_tile_kernel)
    TileGeom t;
    int channels = 0;
    double* xs = nullptr;
    int* order = nullptr;
    int* start = nullptr;       // nbins + 1
    size_t xs_bytes = 0, order_bytes = 0, start_bytes = 0;
    int uses = 0;               // passes served so far
    bool balanced = false;      // tile_class_order_kernel applied
};

}  // namespace efgp

using namespace efgp;

struct efgp_nufft_s {
    int device = 0;
    int dim = 0;
    int64_t npts = 0;
    const double* x = nullptr;
    double xcen[3] = {0, 0, 0};
    double h = 0.0;
    double tol = 1e-6;
    DeviceCtx* ctx = nullptr;
    std::vector<efgp::BinSet*> bins;     // tile-sorted copies of the points, per (fine grid, W, tile) geometry
    std::vector<efgp::ClassOrder*> orders;   // bank-balanced processing orders of spread_pad_kernel, per (fine grid, W, launch)
    efgp_points_s* points = nullptr;     // per-model sorted layout (efgp_nufft_create_on), not owned
};

namespace efgp {

static void free_window(WindowSet* w) {
    if (!w) return;
    if (w->d_block && w->ctx) pool_free(w->ctx, w->d_block, w->block_bytes);
    if (w->ready) (void)hipEventDestroy(w->ready);
    delete w;
}

// window data for a mode box: fine grid = next 2^a3^b5^c size >= 2*n_modes (>= 32).  The data depend
// only on (tolerance, dimension, mode box), not on the points, so they are cached per device.
static int get_window(efgp_nufft_s* plan, const int64_t* n_modes, hipStream_t stream, WindowSet** out) {
    const int d = plan->dim;
    // from 4e6 points on a 2-D plan takes the finer grid / narrower window (N = 1e7, mtot = 23: spread + gather 316 -> 276 us on
    // the same box).  The accumulator -> modes step behind them grows with the grid: as ONE launch it went from 20 to 47-60 us
    // (180 x 180 cells) and ate the gain -- the two-launch form (grid_rows_kernel + rows_modes_kernel) takes 12 / 17 us.
    const char* dense_env = std::getenv("EFGP_DENSE_POINTS");           // experiments: another threshold for the dense-sigma rule
    const int64_t dense_from = dense_env ? std::max<int64_t>(1, std::atoll(dense_env)) : kDensePoints;
    const bool dense = d == 2 && plan->npts >= dense_from && std::getenv("EFGP_NO_DENSE_SIGMA") == nullptr;
    for (void* vp : plan->ctx->window_cache) {
        WindowSet* w = (WindowSet*)vp;
        bool same = w->dim == d && w->tol == plan->tol && w->dense == dense;
        for (int a = 0; a < d && same; ++a) same = w->nm[a] == n_modes[a];
        if (same) {
            if (w->ready && stream != w->up_stream) EFGP_HIP_CHECK(hipStreamWaitEvent(stream, w->ready, 0));
            *out = w;
            return EFGP_OK;
        }
    }
    auto* w = new WindowSet();
    w->tol = plan->tol;
    w->dim = d;
    w->dense = dense;
    double sigma_min = 1e30;
    for (int a = 0; a < 3; ++a) {
        w->nm[a] = a < d ? n_modes[a] : 1;
        w->nf[a] = a < d ? es_fine_size(n_modes[a], plan->tol, d, dense) : 1;
        if (a < d) sigma_min = std::min(sigma_min, (double)w->nf[a] / (double)w->nm[a]);
    }
    es_make_params(plan->tol, sigma_min, &w->p, d);
    const int W = w->p.w, deg = w->p.degree;
    // two tables: [kMaxDegree+1][W] (rows above `deg` stay zero: padded Horner), then the first ceil(W/2)
    // polynomials again as [kMaxDegree+1][sym_row(W)] for window_eval
    const int RHP = sym_row(W);
    std::vector<double> coef((size_t)(kMaxDegree + 1) * (W + RHP), 0.0);
    for (int k = 0; k <= deg; ++k) {
        for (int j = 0; j < W; ++j) coef[(size_t)k * W + j] = w->p.coef[j * (kMaxDegree + 1) + k];
        for (int j = 0; j < (W + 1) / 2; ++j)
            coef[(size_t)(kMaxDegree + 1) * W + (size_t)k * RHP + j] = w->p.coef[j * (kMaxDegree + 1) + k];
    }
    // coefficient tables and the correction factors of every axis in ONE pooled block and ONE stream-ordered upload from a pinned
    // staging slot: a training step with a new mode count used to pay 1 + d hipMalloc calls and as many copies, each behind a
    // hipStreamSynchronize that made the host wait for everything the step had queued
    std::vector<double> host(coef);
    size_t off[3] = {0, 0, 0};
    for (int a = 0; a < d; ++a) {
        std::vector<double> fac;
        es_deconv_factors(w->p, w->nf[a], w->nm[a], &fac);
        while (host.size() % 32) host.push_back(0.0);            // 256-byte alignment of every table
        off[a] = host.size();
        host.insert(host.end(), fac.begin(), fac.end());
    }
    w->ctx = plan->ctx;
    w->block_bytes = host.size() * sizeof(double);
    w->d_block = (double*)pool_alloc(plan->ctx, w->block_bytes);
    if (!w->d_block) {
        free_window(w);
        return EFGP_ENOMEM;
    }
    const int urc = upload_small(plan->ctx, host.data(), w->block_bytes, w->d_block, stream);
    if (urc != EFGP_OK) {
        free_window(w);
        return urc;
    }
    w->up_stream = stream;
    if (hipEventCreateWithFlags(&w->ready, hipEventDisableTiming) != hipSuccess || hipEventRecord(w->ready, stream) != hipSuccess) {
        (void)hipGetLastError();
        EFGP_HIP_CHECK(hipStreamSynchronize(stream));          // no event: make the tables visible to every stream the old way
        if (w->ready) (void)hipEventDestroy(w->ready);
        w->ready = nullptr;
    }
    w->d_coef = w->d_block;
    for (int a = 0; a < d; ++a) w->d_fac[a] = w->d_block + off[a];
    if (plan->ctx->window_cache.size() >= 64) {       // bound the cache: drop the oldest entry
        (void)hipDeviceSynchronize();
        free_window((WindowSet*)plan->ctx->window_cache.front());
        plan->ctx->window_cache.erase(plan->ctx->window_cache.begin());
    }
    plan->ctx->window_cache.push_back(w);
    *out = w;
    return EFGP_OK;
}

static GridGeom make_geom(const efgp_nufft_s* plan, const WindowSet* w) {
    GridGeom g;
    g.cells = 1;
    for (int a = 0; a < 3; ++a) {
        g.nf[a] = w->nf[a];
        g.scale[a] = plan->h * (double)w->nf[a];
        g.xcen[a] = plan->xcen[a];
        g.cells *= w->nf[a];
    }
    return g;
}

static ModeGeom make_modes(const efgp_nufft_s* plan, const WindowSet* w, const int64_t* nm, int modeord) {
    ModeGeom m;
    m.d = plan->dim;
    m.total = 1;
    for (int a = 0; a < 3; ++a) {
        m.nm[a] = a < plan->dim ? nm[a] : 1;
        m.nf[a] = w->nf[a];
        m.fac[a] = w->d_fac[a];
        m.total *= m.nm[a];
    }
    m.modeord = modeord;
    return m;
}

template <int D, int W>
static hipError_t launch_spread_dw(bool use_lds, dim3 grid, size_t lds_bytes, hipStream_t s, const SpreadArgs& a) {
    if (use_lds) {
        auto k = spread_kernel<D, W, true>;
        if (lds_bytes > 65536) {
            hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(k, grid, dim3(kSpreadThreads), lds_bytes, s, a);
    } else {
        hipLaunchKernelGGL((spread_kernel<D, W, false>), grid, dim3(kSpreadThreads), 0, s, a);
    }
    return hipGetLastError();
}

template <int D, int W, bool RAW48>
static hipError_t launch_spread_pad_dwr(dim3 grid, size_t lds_bytes, hipStream_t s, const SpreadArgs& a) {
    auto k = spread_pad_kernel<D, W, RAW48>;
    if (lds_bytes > 65536) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k, grid, dim3(kSpreadThreads), lds_bytes, s, a);
    return hipGetLastError();
}

template <int D>
static hipError_t launch_spread_pad_d(int W, bool raw48, dim3 grid, size_t lds_bytes, hipStream_t s, const SpreadArgs& a) {
    switch (W) {
#define EFGP_CASE(w_)                                                                          \
    case w_:                                                                                   \
        return raw48 ? launch_spread_pad_dwr<D, w_, true>(grid, lds_bytes, s, a)               \
                     : launch_spread_pad_dwr<D, w_, false>(grid, lds_bytes, s, a);
        EFGP_CASE(2) EFGP_CASE(3) EFGP_CASE(4) EFGP_CASE(5) EFGP_CASE(6) EFGP_CASE(7) EFGP_CASE(8) EFGP_CASE(9)
        EFGP_CASE(10) EFGP_CASE(11) EFGP_CASE(12) EFGP_CASE(13) EFGP_CASE(14) EFGP_CASE(15) EFGP_CASE(16)
#undef EFGP_CASE
    }
    return hipErrorInvalidValue;
}

template <int D>
static hipError_t launch_spread_d(int W, bool use_lds, dim3 grid, size_t lds_bytes, hipStream_t s, const SpreadArgs& a) {
    switch (W) {
#define EFGP_CASE(w_) case w_: return launch_spread_dw<D, w_>(use_lds, grid, lds_bytes, s, a);
        EFGP_CASE(2) EFGP_CASE(3) EFGP_CASE(4) EFGP_CASE(5) EFGP_CASE(6) EFGP_CASE(7) EFGP_CASE(8) EFGP_CASE(9)
        EFGP_CASE(10) EFGP_CASE(11) EFGP_CASE(12) EFGP_CASE(13) EFGP_CASE(14) EFGP_CASE(15) EFGP_CASE(16)
#undef EFGP_CASE
    }
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------
// Tiled type-2 gather for fine grids beyond LDS (all 3-D cases, large 2-D): the counterpart of
// spread_tile_kernel.  Points are counting-sorted by tile once per plan (same binning kernels); a workgroup
// takes a contiguous chunk of sorted points and, per tile segment, copies the tile's (T+W-1)^d neighbourhood of
// the fine grid (real part only for real outputs) into LDS with the wrap-around resolved, then every thread
// gathers its points from LDS with plain strided reads and writes the result to the point's original slot.
// Before: W^d reads per point through L2 (15.4 ms at N=1e7, d=3).
// ------------------------------------------------------------------------------------------
struct TileInterpArgs {
    TileGeom t;
    const double* xs;         // tile-sorted coordinates
    const int* order;         // sorted position -> original index
    const int* start;         // [nbins + 1]
    int64_t npts;
    int64_t chunk;            // sorted points per workgroup
    const double* coef;
    int degree;
    const double2* fine;      // [batch][cells]
    int64_t cells;
    void* out;                // [batch][npts] complex or real
};

template <int D, int W, bool CPLX>
__global__ __launch_bounds__(kInterpThreads) void interp_tile_kernel(TileInterpArgs a) {
    extern __shared__ double lds[];
    __shared__ int s_bin;
    const TileGeom& t = a.t;
    const int batch = blockIdx.y;
    const int e0 = t.ext[0], e1 = t.ext[1], e2 = t.ext[2];
    const int tcells = e0 * e1 * e2;
    const int64_t lo = (int64_t)blockIdx.x * a.chunk;
    const int64_t hi = lo + a.chunk < a.npts ? lo + a.chunk : a.npts;
    if (lo >= hi) return;
    const double2* F = a.fine + (int64_t)batch * a.cells;
    double2* lc = reinterpret_cast<double2*>(lds);
    if (threadIdx.x == 0) {
        int l = 0, r = t.nbins;            // invariant: start[l] <= lo < start[r]
        while (r - l > 1) {
            const int m = (l + r) >> 1;
            if ((int64_t)a.start[m] <= lo) l = m;
            else r = m;
        }
        s_bin = l;
    }
    __syncthreads();
    int bin = s_bin;
    int64_t cur = lo;
    while (cur < hi) {
        while ((int64_t)a.start[bin + 1] <= cur) ++bin;            // skip empty tiles
        const int64_t seg_hi = (int64_t)a.start[bin + 1] < hi ? (int64_t)a.start[bin + 1] : hi;
        int rem = bin, o[3] = {0, 0, 0};
        for (int q = D - 1; q >= 0; --q) {
            o[q] = (rem % t.nt[q]) * t.T[q];
            rem /= t.nt[q];
        }
        for (int i = threadIdx.x; i < tcells; i += kInterpThreads) {
            const int l2 = i % e2, l1 = (i / e2) % e1, l0 = i / (e2 * e1);
            int g0 = o[0] + l0, g1 = o[1] + l1, g2 = o[2] + l2;
            if (g0 >= t.nf[0]) g0 -= t.nf[0];
            if (D > 1 && g1 >= t.nf[1]) g1 -= t.nf[1];
            if (D > 2 && g2 >= t.nf[2]) g2 -= t.nf[2];
            const int64_t gi = D == 1 ? g0 : (D == 2 ? (int64_t)g0 * t.nf[1] + g1 : ((int64_t)g0 * t.nf[1] + g1) * t.nf[2] + g2);
            if (CPLX) lc[i] = F[gi];
            else lds[i] = F[gi].x;
        }
        __syncthreads();
        for (int64_t n = cur + threadIdx.x; n < seg_hi; n += kInterpThreads) {
            double v0[W], v1[W], v2[W];
            int f0 = 0, f1 = 0, f2 = 0;
            {
                double Xw[3] = {0.0, 0.0, 0.0};
                Xw[0] = fold(t.scale[0] * (a.xs[n * D + 0] - t.xcen[0]), (double)t.nf[0]);
                if (D > 1) Xw[1] = fold(t.scale[1] * (a.xs[n * D + 1] - t.xcen[1]), (double)t.nf[1]);
                if (D > 2) Xw[2] = fold(t.scale[2] * (a.xs[n * D + 2] - t.xcen[2]), (double)t.nf[2]);
                window_eval<D, W>(a.coef, a.degree, Xw, t.nf[0], t.nf[1], t.nf[2], f0, f1, f2, v0, v1, v2);
                f0 -= o[0];
                if (D > 1) f1 -= o[1];
                if (D > 2) f2 -= o[2];
            }
            double ar = 0.0, ai = 0.0;
            if (D == 1) {
#pragma unroll
                for (int j = 0; j < W; ++j) {
                    if (CPLX) {
                        const double2 c = lc[f0 + j];
                        ar = fma(v0[j], c.x, ar);
                        ai = fma(v0[j], c.y, ai);
                    } else {
                        ar = fma(v0[j], lds[f0 + j], ar);
                    }
                }
            } else if (D == 2) {
#pragma unroll
                for (int j0 = 0; j0 < W; ++j0) {
                    const int rowi = (f0 + j0) * e1 + f1;
                    double rr = 0.0, ri = 0.0;
#pragma unroll
                    for (int j1 = 0; j1 < W; ++j1) {
                        if (CPLX) {
                            const double2 c = lc[rowi + j1];
                            rr = fma(v1[j1], c.x, rr);
                            ri = fma(v1[j1], c.y, ri);
                        } else {
                            rr = fma(v1[j1], lds[rowi + j1], rr);
                        }
                    }
                    ar = fma(v0[j0], rr, ar);
                    if (CPLX) ai = fma(v0[j0], ri, ai);
                }
            } else {
                for (int j0 = 0; j0 < W; ++j0) {
                    double r0r = 0.0, r0i = 0.0;
                    for (int j1 = 0; j1 < W; ++j1) {
                        const int rowi = ((f0 + j0) * e1 + (f1 + j1)) * e2 + f2;
                        double rr = 0.0, ri = 0.0;
#pragma unroll
                        for (int j2 = 0; j2 < W; ++j2) {
                            if (CPLX) {
                                const double2 c = lc[rowi + j2];
                                rr = fma(v2[j2], c.x, rr);
                                ri = fma(v2[j2], c.y, ri);
                            } else {
                                rr = fma(v2[j2], lds[rowi + j2], rr);
                            }
                        }
                        r0r = fma(v1[j1], rr, r0r);
                        if (CPLX) r0i = fma(v1[j1], ri, r0i);
                    }
                    ar = fma(v0[j0], r0r, ar);
                    if (CPLX) ai = fma(v0[j0], r0i, ai);
                }
            }
            const int64_t dst = (int64_t)batch * a.npts + a.order[n];
            if (CPLX) reinterpret_cast<double2*>(a.out)[dst] = make_double2(ar, ai);
            else reinterpret_cast<double*>(a.out)[dst] = ar;
        }
        __syncthreads();
        cur = seg_hi;
    }
}

template <int D>
static hipError_t launch_interp_tile_d(int W, bool cplx, dim3 grid, size_t lds_bytes, hipStream_t s, const TileInterpArgs& a) {
    switch (W) {
#define EFGP_CASE(w_)                                                                                               \
    case w_: {                                                                                                      \
        auto kc = interp_tile_kernel<D, w_, true>;                                                                  \
        auto kr = interp_tile_kernel<D, w_, false>;                                                                 \
        if (lds_bytes > 65536) {                                                                                    \
            hipError_t e = hipFuncSetAttribute(cplx ? (const void*)kc : (const void*)kr,                            \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);         \
            if (e != hipSuccess) return e;                                                                          \
        }                                                                                                           \
        if (cplx) hipLaunchKernelGGL(kc, grid, dim3(kInterpThreads), lds_bytes, s, a);                              \
        else hipLaunchKernelGGL(kr, grid, dim3(kInterpThreads), lds_bytes, s, a);                                   \
        return hipGetLastError();                                                                                   \
    }
        EFGP_CASE(2) EFGP_CASE(3) EFGP_CASE(4) EFGP_CASE(5) EFGP_CASE(6) EFGP_CASE(7) EFGP_CASE(8) EFGP_CASE(9)
        EFGP_CASE(10) EFGP_CASE(11) EFGP_CASE(12) EFGP_CASE(13) EFGP_CASE(14) EFGP_CASE(15) EFGP_CASE(16)
#undef EFGP_CASE
    }
    return hipErrorInvalidValue;
}

template <int D, int W, bool CPLX>
static hipError_t launch_interp_dwc(bool use_lds, dim3 grid, size_t lds_bytes, hipStream_t s, const InterpArgs& a) {
    if (use_lds) {
        auto k = interp_kernel<D, W, CPLX, true>;
        if (lds_bytes > 65536) {
            hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(k, grid, dim3(kInterpThreads), lds_bytes, s, a);
    } else {
        hipLaunchKernelGGL((interp_kernel<D, W, CPLX, false>), grid, dim3(kInterpThreadsGlobal), 0, s, a);
    }
    return hipGetLastError();
}

template <int D>
static hipError_t launch_interp_halo_d(int W, dim3 grid, size_t lds_bytes, hipStream_t s, const InterpArgs& a) {
    switch (W) {
#define EFGP_CASE(w_)                                                                                               \
    case w_: {                                                                                                      \
        auto k = interp_real_halo_kernel<D, w_>;                                                                    \
        if (lds_bytes > 65536) {                                                                                    \
            hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
            if (e != hipSuccess) return e;                                                                          \
        }                                                                                                           \
        hipLaunchKernelGGL(k, grid, dim3(kInterpThreads), lds_bytes, s, a);                                         \
        return hipGetLastError();                                                                                   \
    }
        EFGP_CASE(2) EFGP_CASE(3) EFGP_CASE(4) EFGP_CASE(5) EFGP_CASE(6) EFGP_CASE(7) EFGP_CASE(8) EFGP_CASE(9)
        EFGP_CASE(10) EFGP_CASE(11) EFGP_CASE(12) EFGP_CASE(13) EFGP_CASE(14) EFGP_CASE(15) EFGP_CASE(16)
#undef EFGP_CASE
    }
    return hipErrorInvalidValue;
}

static size_t interp_pair_lds_bytes(int nf0, int nf1, int W) {
    const int wp = (W + 1) / 2;
    return 2 * (size_t)(nf0 + W - 1) * (size_t)((nf1 + 2 * wp + 1) & ~1) * sizeof(double);
}

static hipError_t launch_interp_pair(int W, dim3 grid, size_t lds_bytes, hipStream_t s, const InterpArgs& a) {
    switch (W) {
#define EFGP_CASE(w_)                                                                                               \
    case w_: {                                                                                                      \
        auto k = interp_real2_pair_kernel<w_>;                                                                      \
        if (lds_bytes > 65536) {                                                                                    \
            hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
            if (e != hipSuccess) return e;                                                                          \
        }                                                                                                           \
        hipLaunchKernelGGL(k, grid, dim3(kInterpThreads), lds_bytes, s, a);                                         \
        return hipGetLastError();                                                                                   \
    }
        EFGP_CASE(2) EFGP_CASE(3) EFGP_CASE(4) EFGP_CASE(5) EFGP_CASE(6) EFGP_CASE(7) EFGP_CASE(8) EFGP_CASE(9)
        EFGP_CASE(10) EFGP_CASE(11) EFGP_CASE(12) EFGP_CASE(13) EFGP_CASE(14) EFGP_CASE(15) EFGP_CASE(16)
#undef EFGP_CASE
    }
    return hipErrorInvalidValue;
}

template <int D>
static hipError_t launch_interp_d(int W, bool cplx, bool use_lds, dim3 grid, size_t lds_bytes, hipStream_t s,
                                  const InterpArgs& a) {
    switch (W) {
#define EFGP_CASE(w_)                                                                   \
    case w_:                                                                            \
        return cplx ? launch_interp_dwc<D, w_, true>(use_lds, grid, lds_bytes, s, a)   \
                    : launch_interp_dwc<D, w_, false>(use_lds, grid, lds_bytes, s, a);
        EFGP_CASE(2) EFGP_CASE(3) EFGP_CASE(4) EFGP_CASE(5) EFGP_CASE(6) EFGP_CASE(7) EFGP_CASE(8) EFGP_CASE(9)
        EFGP_CASE(10) EFGP_CASE(11) EFGP_CASE(12) EFGP_CASE(13) EFGP_CASE(14) EFGP_CASE(15) EFGP_CASE(16)
#undef EFGP_CASE
    }
    return hipErrorInvalidValue;
}

// ---- tiled path: geometry, binning, launch -------------------------------------------------------
// force_tile > 0: use that tile size (1 = bins are single cells, for the cell-sorted spreader)
static bool make_tile_geom(const efgp_nufft_s* plan, const WindowSet* w, int channels, size_t lds_budget, TileGeom* out,
                           int force_tile = 0) {
    TileGeom t;
    const int d = plan->dim, W = w->p.w;
    t.d = d;
    t.W = W;
    // largest cubic-ish tile whose (T+W-1)^d * channels * 8 B fits the LDS budget
    const double cells_max = (double)lds_budget / (8.0 * channels);
    int ext = (int)std::floor(std::pow(cells_max, 1.0 / d));
    while (ext > W && std::pow((double)ext, d) > cells_max) --ext;
    int Tmax = ext - (W - 1);
    if (force_tile > 0) Tmax = force_tile;
    if (Tmax < 1 || (force_tile == 0 && Tmax < 2)) return false;
    t.nbins = 1;
    for (int a = 0; a < 3; ++a) {
        if (a < d) {
            t.nf[a] = (int)w->nf[a];
            t.nt[a] = (t.nf[a] + Tmax - 1) / Tmax;
            t.T[a] = (t.nf[a] + t.nt[a] - 1) / t.nt[a];
            t.ext[a] = t.T[a] + W - 1;
            t.scale[a] = plan->h * (double)w->nf[a];
            t.xcen[a] = plan->xcen[a];
            t.nbins *= t.nt[a];
        } else {
            t.nf[a] = 1;
            t.nt[a] = 1;
            t.T[a] = 1;
            t.ext[a] = 1;
            t.scale[a] = 0.0;
            t.xcen[a] = 0.0;
        }
    }
    if (t.nbins > 16384) return false;          // LDS histograms of the binning kernels (2 x 64 KB)
    *out = t;
    return true;
}

static void free_binset(DeviceCtx* ctx, BinSet* b) {
    if (!b) return;
    pool_free(ctx, b->xs, b->xs_bytes);
    pool_free(ctx, b->order, b->order_bytes);
    pool_free(ctx, b->start, b->start_bytes);
    delete b;
}

// Bank-balanced order inside the tiles (tile_class_order_kernel); tiles proper only: the single-cell bins of the
// cell-sorted spreader have one class.  Costs one out-of-place pass over the sorted points.
static int balance_tiles(efgp_nufft_s* plan, BinSet* b, hipStream_t stream) {
    KernelTimer order_timer("order", stream);
    DeviceCtx* ctx = plan->ctx;
    const TileGeom& t = b->t;
    if (b->balanced) return EFGP_OK;
    b->balanced = true;
    if (!(t.T[0] > 1 && plan->dim >= 2 && plan->npts >= (int64_t)kOrderWindow * 16 && std::getenv("EFGP_NO_CLASS_ORDER") == nullptr))
        return EFGP_OK;
    double* xs2 = (double*)pool_alloc(ctx, b->xs_bytes);
    int* order2 = (int*)pool_alloc(ctx, b->order_bytes);
    if (xs2 && order2) {
        const unsigned nwin = (unsigned)((plan->npts + kOrderWindow - 1) / kOrderWindow);
        if (plan->dim == 2)
            hipLaunchKernelGGL((tile_class_order_kernel<2>), dim3(nwin), dim3(kSpreadThreads), 0, stream, t, (const double*)b->xs,
                               (const int*)b->order, (const int*)b->start, plan->npts, xs2, order2);
        else
            hipLaunchKernelGGL((tile_class_order_kernel<3>), dim3(nwin), dim3(kSpreadThreads), 0, stream, t, (const double*)b->xs,
                               (const int*)b->order, (const int*)b->start, plan->npts, xs2, order2);
        EFGP_HIP_CHECK(hipGetLastError());
        // the old copies may still be read by the kernel just queued: park them in the pool (stream-ordered reuse)
        pool_free(ctx, b->xs, b->xs_bytes);
        pool_free(ctx, b->order, b->order_bytes);
        b->xs = xs2;
        b->order = order2;
    } else {
        if (xs2) pool_free(ctx, xs2, b->xs_bytes);
        if (order2) pool_free(ctx, order2, b->order_bytes);
    }
    return EFGP_OK;
}

static int get_bins(efgp_nufft_s* plan, const TileGeom& t, int channels, hipStream_t stream, BinSet** out) {
    for (BinSet* b : plan->bins) {
        bool same = b->t.W == t.W && b->t.nbins == t.nbins;
        for (int a = 0; a < plan->dim && same; ++a) same = b->t.nf[a] == t.nf[a] && b->t.T[a] == t.T[a];
        if (same) {
            if (++b->uses >= 1 && !b->balanced) {          // second pass over this binning: the reordering now pays
                int rcb = balance_tiles(plan, b, stream);
                if (rcb != EFGP_OK) return rcb;
            }
            *out = b;
            return EFGP_OK;
        }
    }
    DeviceCtx* ctx = plan->ctx;
    auto* b = new BinSet();
    b->t = t;
    b->channels = channels;
    b->xs_bytes = (size_t)plan->npts * plan->dim * sizeof(double);
    b->order_bytes = (size_t)plan->npts * sizeof(int);
    b->start_bytes = (size_t)(t.nbins + 1) * sizeof(int);
    b->xs = (double*)pool_alloc(ctx, b->xs_bytes);
    b->order = (int*)pool_alloc(ctx, b->order_bytes);
    b->start = (int*)pool_alloc(ctx, b->start_bytes);
    int* tmp = (int*)scratch(ctx, SLOT_MISC, (size_t)2 * (t.nbins + 1) * sizeof(int));     // hist, cursor
    if (!b->xs || !b->order || !b->start || !tmp) {
        free_binset(ctx, b);
        return EFGP_ENOMEM;
    }
    int* hist = tmp;
    int* cursor = tmp + t.nbins + 1;
    KernelTimer* order_timer = new KernelTimer("order", stream);      // closed after the scatter below
    EFGP_HIP_CHECK(hipMemsetAsync(hist, 0, (size_t)(t.nbins + 1) * sizeof(int), stream));
    // chunks of <= 32 * 1024 points per workgroup (bin_scatter_kernel keeps 32 ranks per thread)
    const int nwg = (int)std::max<int64_t>(1, (plan->npts + 32 * 1024 - 1) / (32 * 1024));
    const size_t lds_h = (size_t)t.nbins * sizeof(int), lds_s = (size_t)2 * t.nbins * sizeof(int);
#define EFGP_BIN_LAUNCH(D_)                                                                                              \
    if (lds_s > 65536) {                                                                                                  \
        EFGP_HIP_CHECK(hipFuncSetAttribute((const void*)bin_scatter_kernel<D_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                           (int)lds_s));                                                                  \
    }                                                                                                                     \
    if (lds_h > 65536) {                                                                                                  \
        EFGP_HIP_CHECK(hipFuncSetAttribute((const void*)bin_hist_kernel<D_>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                           (int)lds_h));                                                                  \
    }                                                                                                                     \
    hipLaunchKernelGGL((bin_hist_kernel<D_>), dim3(nwg), dim3(1024), lds_h, stream, t, plan->x, plan->npts, hist);        \
    hipLaunchKernelGGL(bin_scan_kernel, dim3(1), dim3(1024), 0, stream, (const int*)hist, t.nbins, b->start, cursor);     \
    hipLaunchKernelGGL((bin_scatter_kernel<D_>), dim3(nwg), dim3(1024), lds_s, stream, t, plan->x, plan->npts, cursor,    \
                       b->xs, b->order);
    if (plan->dim == 1) { EFGP_BIN_LAUNCH(1) }
    else if (plan->dim == 2) { EFGP_BIN_LAUNCH(2) }
    else { EFGP_BIN_LAUNCH(3) }
#undef EFGP_BIN_LAUNCH
    delete order_timer;
    EFGP_HIP_CHECK(hipGetLastError());
    // 3-D stencils (W^3 LDS atomics per point) repay the reordering pass at once; 2-D tiles on the second pass
    if (plan->dim == 3) {
        int rcb = balance_tiles(plan, b, stream);
        if (rcb != EFGP_OK) return rcb;
    }
    plan->bins.push_back(b);
    *out = b;
    return EFGP_OK;
}

template <int D>
static hipError_t launch_tile_d(int W, dim3 grid, size_t lds_bytes, hipStream_t s, const TileSpreadArgs& a) {
    switch (W) {
#define EFGP_CASE(w_)                                                                                               \
    case w_: {                                                                                                      \
        auto k = spread_tile_kernel<D, w_>;                                                                         \
        if (lds_bytes > 65536) {                                                                                    \
            hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
            if (e != hipSuccess) return e;                                                                          \
        }                                                                                                           \
        hipLaunchKernelGGL(k, grid, dim3(kSpreadThreads), lds_bytes, s, a);                                         \
        return hipGetLastError();                                                                                   \
    }
        EFGP_CASE(2) EFGP_CASE(3) EFGP_CASE(4) EFGP_CASE(5) EFGP_CASE(6) EFGP_CASE(7) EFGP_CASE(8) EFGP_CASE(9)
        EFGP_CASE(10) EFGP_CASE(11) EFGP_CASE(12) EFGP_CASE(13) EFGP_CASE(14) EFGP_CASE(15) EFGP_CASE(16)
#undef EFGP_CASE
    }
    return hipErrorInvalidValue;
}

// the cell kernel unrolls its Horner loop: degrees are padded up to W + 2 or W + 4 (zero rows in the table)
constexpr int kCellMaxW = 8;       // register budget: RH^2 accumulators per channel + RH (DEG+1) coefficients
template <int W>
static hipError_t launch_cell_w(int channels, int degree, dim3 grid, hipStream_t s, const CellSpreadArgs& a) {
    if (degree <= W + 2) {
        if (channels == 2) hipLaunchKernelGGL((spread_cell_kernel<W, 2, W + 2>), grid, dim3(256), kCellLds, s, a);
        else hipLaunchKernelGGL((spread_cell_kernel<W, 1, W + 2>), grid, dim3(256), kCellLds, s, a);
    } else {
        if (channels == 2) hipLaunchKernelGGL((spread_cell_kernel<W, 2, W + 4>), grid, dim3(256), kCellLds, s, a);
        else hipLaunchKernelGGL((spread_cell_kernel<W, 1, W + 4>), grid, dim3(256), kCellLds, s, a);
    }
    return hipGetLastError();
}
static hipError_t launch_cell(int W, int channels, int degree, dim3 grid, hipStream_t s, const CellSpreadArgs& a) {
    switch (W) {
#define EFGP_CASE(w_) case w_: return launch_cell_w<w_>(channels, degree, grid, s, a);
        EFGP_CASE(2) EFGP_CASE(3) EFGP_CASE(4) EFGP_CASE(5) EFGP_CASE(6) EFGP_CASE(7) EFGP_CASE(8)
#undef EFGP_CASE
    }
    return hipErrorInvalidValue;
}

// 64-byte block of doubles: [0..3] S0, 1/S0, S1, 1/S1, [4] channel-0 norm, [6] arrival ticket of maxabs_kernel, [7] max|c| bit pattern.
// Zeroed when first allocated; afterwards maxabs_kernel's last workgroup leaves ticket and accumulator at zero.
static char* scale_slot(DeviceCtx* ctx, hipStream_t stream) {
    char* misc = (char*)scratch(ctx, SLOT_SCALE, 64);
    if (misc && !ctx->scale_slot_ready) {
        if (hipMemsetAsync(misc, 0, 64, stream) != hipSuccess) return nullptr;
        ctx->scale_slot_ready = true;
    }
    return misc;
}

// Level of the plan's point layout the MFMA spreader can use for this fine grid, or nullptr (no layout, not 2-D,
// window wider than the register tile, too few points per run to amortise the tile flushes, EFGP_NO_MFMA_SPREAD).
// *band_cells = bound on the height of the level's bands in fine cells: 1 for dense point sets (>= 192 points per run at
// one-cell bands: the spreader then needs W + 1 tile columns, fits three waves per SIMD and stores seven zeros less per point),
// else 8 (the tile's 16 columns hold the 8-cell stencil at offsets 0..8).  EFGP_MFMA_BAND_CELLS = 1 | 8 forces one.
static SortedLevel* pick_level(efgp_nufft_s* plan, const WindowSet* w, const GridGeom& g, hipStream_t stream, int* band_cells) {
    efgp_points_s* pts = plan->points;
    if (!pts || plan->dim != 2 || w->p.w > kMfmaMaxW || plan->npts < 32768 || std::getenv("EFGP_NO_MFMA_SPREAD")) return nullptr;
    double span[2], far = 0.0;
    for (int a = 0; a < 2; ++a) {
        span[a] = (pts->hi[a] - pts->lo[a]) * std::fabs(g.scale[a]);
        far = std::max(far, std::max(std::fabs(pts->hi[a] - plan->xcen[a]), std::fabs(pts->lo[a] - plan->xcen[a])) * std::fabs(g.scale[a]));
    }
    if (!(far < 1e9) || !(g.scale[0] > 0.0) || !(g.scale[1] > 0.0)) return nullptr;      // cell indices are kept in 32-bit ints
    const char* force = std::getenv("EFGP_MFMA_BAND_CELLS");
    const int forced = force ? std::atoi(force) : 0;
    int nb = 0, cells = 0;
    for (int hb : {1, 8}) {
        if (forced && forced != hb) continue;
        int n = 1;
        while (span[1] / n > (double)hb - 1e-6 && n <= kMaxBands) n *= 2;
        if (n > kMaxBands) continue;
        const double per_run = (double)plan->npts / ((double)n * std::max(1.0, span[0]));
        if (per_run < (forced ? 1.0 : (hb == 1 ? 192.0 : 24.0))) continue;
        nb = n;
        cells = hb;
        if (hb == 8) {
            // A level costs a sort of the N points (2 ms at N = 1e6) and 28 B per point.  A gradient step makes two plans on one
            // model -- the Toeplitz box (4 m + 1 modes) and the probe box (mtot modes) on a grid half as fine -- whose coarsest
            // levels differ by a factor two: the probe plan takes the pair plan's level when it exists (bands 2-4 cells high instead
            // of 4-8: a few more run ends per band, no second sort; round 4: the layout-building step 4.3 -> 2.4 ms).
            bool have = false;
            for (const SortedLevel* l : pts->levels) have = have || l->nbands == n;
            for (int f = 2; !have && f <= 4; f *= 2) {
                const double per_run_f = (double)plan->npts / ((double)(n * f) * std::max(1.0, span[0]));
                if (n * f > kMaxBands || per_run_f < 24.0) break;
                for (const SortedLevel* l : pts->levels)
                    if (l->nbands == n * f) {
                        nb = n * f;
                        have = true;
                    }
            }
        }
        break;
    }
    if (!nb) return nullptr;
    SortedLevel* lvl = nullptr;
    if (points_level(pts, nb, stream, &lvl) != EFGP_OK) return nullptr;
    *band_cells = cells;
    return lvl;
}

// What the caller wants from the transformed grid; when the pass runs on the MFMA spreader's single int64 grid and the grid
// is small, spread_and_fft hands the accumulator straight to grid_to_modes_kernel and sets `done` (no fine grid is produced).
struct G2MRequest {
    int part = 0;              // 0 | 3 | 4 (see G2MArgs)
    int rows_limit = 1 << 30;
    ModeGeom ma, mb;
    void* out_a = nullptr;
    void* out_b = nullptr;
    bool done = false;
    // set by transform_fine when the in-house pruned transform ran: the spectrum handed back holds only the crop_nf[a] lowest-|k|
    // bins per axis (FFT order on that smaller torus) -- the mode extraction indexes it with these extents instead of nf
    bool cropped = false;
    int64_t crop_nf[3] = {1, 1, 1};
};
static void apply_crop(const G2MRequest* req, ModeGeom& m, int64_t& cells) {
    if (!req || !req->cropped) return;
    cells = 1;
    for (int a = 0; a < 3; ++a) {
        if (a < m.d) m.nf[a] = req->crop_nf[a];
        cells *= a < m.d ? req->crop_nf[a] : 1;
    }
}

static bool g2m_eligible(const efgp_nufft_s* plan, const GridGeom& g, const G2MRequest* req) {
    if (!req || plan->dim != 2 || std::getenv("EFGP_NO_GRID_TO_MODES")) return false;
    if (g.nf[0] > kG2MMaxNf || g.nf[1] > kG2MMaxNf) return false;
    for (int a = 0; a < 2; ++a) {
        if (req->ma.nm[a] / 2 > kG2MMaxH) return false;
        if (req->part == 4 && req->mb.nm[a] / 2 > kG2MMaxH) return false;
    }
    return true;
}

// gacc != null: from the MFMA spreader's int64 accumulator (converted, cleared by the kernel); else from the reduced complex grid
static int g2m_launch(DeviceCtx* ctx, const GridGeom& g, G2MRequest* req, long long* gacc, const double2* fine, const double* scale,
                      int channels, int nbatch, int isign, unsigned int* ticket, long long acc_words, hipStream_t stream) {
    G2MArgs ga;
    ga.gacc = gacc;
    ga.fine = fine;
    ga.scale = scale;
    ga.channels = channels;
    ga.nf0 = (int)g.nf[0];
    ga.nf1 = (int)g.nf[1];
    ga.h0 = (int)std::max(req->ma.nm[0] / 2, req->part == 4 ? req->mb.nm[0] / 2 : (int64_t)0);
    ga.h1 = (int)std::max(req->ma.nm[1] / 2, req->part == 4 ? req->mb.nm[1] / 2 : (int64_t)0);
    ga.part = req->part;
    ga.rows_limit = req->rows_limit;
    ga.sign = isign < 0 ? -1 : 1;
    ga.ma = req->ma;
    ga.mb = req->part == 4 ? req->mb : req->ma;
    ga.out_a = (double2*)req->out_a;
    ga.out_b = (double2*)req->out_b;
    ga.ticket = ticket;
    if (std::getenv("EFGP_G2M_V1") == nullptr) {
        // round 3: two launches -- rows of the accumulator over the chip, then one workgroup per column pair of the mode box
        G2RArgs ra;
        ra.gacc = gacc;
        ra.fine = fine;
        ra.scale = scale;
        ra.channels = channels;
        ra.nf0 = ga.nf0;
        ra.nf1 = ga.nf1;
        ra.h0 = ga.h0;
        ra.h1 = ga.h1;
        ra.part = ga.part;
        ra.rows_limit = ga.rows_limit;
        ra.sign = ga.sign;
        ra.ma = ga.ma;
        ra.mb = ga.mb;
        ra.out_a = ga.out_a;
        ra.out_b = ga.out_b;
        const int nk1 = 2 * ga.h1 + 1;
        ra.cbuf = (double2*)scratch(ctx, SLOT_G2M, (size_t)nbatch * ga.nf0 * nk1 * sizeof(double2));
        if (!ra.cbuf) return EFGP_ENOMEM;
        ctx->g2m_zeroed_for = nullptr;                   // (the one-launch kernel's counters share this slot)
        const size_t lds = (256 + (size_t)kG2RRows * ga.nf1) * sizeof(double2);
        const dim3 grid((ga.nf0 + kG2RRows - 1) / kG2RRows, nbatch);
        if (gacc) hipLaunchKernelGGL((grid_rows_kernel<false>), grid, dim3(kG2RThreads), lds, stream, ra);
        else hipLaunchKernelGGL((grid_rows_kernel<true>), grid, dim3(kG2RThreads), lds, stream, ra);
        EFGP_HIP_CHECK(hipGetLastError());
        hipLaunchKernelGGL(rows_modes_kernel, dim3(ga.h1 + 1, nbatch), dim3(kR2MThreads), 0, stream, ra);
        EFGP_HIP_CHECK(hipGetLastError());
        req->done = true;
        return EFGP_OK;
    }
    const unsigned tiles = (unsigned)(ga.h0 / 2 + 1);
    const bool small = ga.nf0 <= 128 && ga.nf1 <= 128;
    // stage-1 split, measured (rocprofv3, launch average): 96 x 96 grid 20.0 / 27.7 / 32.4 / 45.7 us at 1 / 2 / 4 / 8 workgroups
    // per tile -- the shares' round trip through memory (two device-scope fences, an atomic) costs more than a quarter of stage 1
    // saves; 180 x 180 grid 59.8 / 51.2 / 47.5 / 56.2 us.
    int split = small ? 1 : 4;
    if (const char* e = std::getenv("EFGP_G2M_SPLIT")) split = std::max(1, std::min(16, std::atoi(e)));
    while (split > 1 && (int64_t)tiles * nbatch * split > 4096) split /= 2;
    constexpr size_t kTicketBytes = 65536;
    ga.partial = nullptr;
    ga.tile_ticket = nullptr;
    if (split > 1) {
        if ((size_t)tiles * nbatch * sizeof(unsigned int) > kTicketBytes) split = 1;
    }
    if (split > 1) {
        const size_t bytes = kTicketBytes + (size_t)nbatch * tiles * split * 4 * ga.nf1 * sizeof(double2);
        char* base = (char*)scratch(ctx, SLOT_G2M, bytes);
        if (!base) return EFGP_ENOMEM;
        if (ctx->g2m_zeroed_for != (const void*)base) {
            EFGP_HIP_CHECK(hipMemsetAsync(base, 0, kTicketBytes, stream));
            ctx->g2m_zeroed_for = base;
        }
        ga.tile_ticket = (unsigned int*)base;
        ga.partial = (double2*)(base + kTicketBytes);
    }
    ga.total_wgs = tiles * (unsigned)nbatch * (unsigned)split;
    ga.acc_words = acc_words;
    const dim3 grid(tiles, nbatch, split);
    if (gacc && small) hipLaunchKernelGGL((grid_to_modes_kernel<false, 7>), grid, dim3(kG2MThreads), 0, stream, ga);
    else if (gacc) hipLaunchKernelGGL((grid_to_modes_kernel<false, 8>), grid, dim3(kG2MThreads), 0, stream, ga);
    else if (small) hipLaunchKernelGGL((grid_to_modes_kernel<true, 7>), grid, dim3(kG2MThreads), 0, stream, ga);
    else hipLaunchKernelGGL((grid_to_modes_kernel<true, 8>), grid, dim3(kG2MThreads), 0, stream, ga);
    EFGP_HIP_CHECK(hipGetLastError());
    req->done = true;
    return EFGP_OK;
}

// the reduced fine grids of the other spreaders: one-launch pruned DFT when the grid is small, else the FFT in place
// `acc` != null: the caller's spread left its sums in the int64 accumulator and has NOT converted them to `fine`; the pruned
// transform's first pass reads (and, if asked, clears) the accumulator itself -- one pass over the grid less per type-1
// transform (reduce_slabs_kernel: 197 us of a 3-D pair transform at 240^3); every other route converts first.
static int transform_fine(efgp_nufft_s* plan, const GridGeom& g, double2* fine, int nbatch, int isign, hipStream_t stream,
                          G2MRequest* req, const double* scale, double2** fine_out, const FftAccSource* acc = nullptr) {
    auto convert = [&]() -> int {
        if (!acc) return EFGP_OK;
        const int rb = (int)((g.cells + 63) / 64);
        hipLaunchKernelGGL((reduce_slabs_kernel<true>), dim3(rb, nbatch), dim3(512), 0, stream, (const double*)acc->acc, 1, acc->channels,
                           g.cells, acc->scale, fine, acc->reset);
        EFGP_HIP_CHECK(hipGetLastError());
        return EFGP_OK;
    };
    if (g2m_eligible(plan, g, req)) {
        const int rcc = convert();
        if (rcc != EFGP_OK) return rcc;
        *fine_out = nullptr;
        return g2m_launch(plan->ctx, g, req, nullptr, fine, scale, 2, nbatch, isign, nullptr, 0, stream);
    }
    // In-house pruned transform when the caller said which modes it wants: every pass keeps only the bins of the (larger) mode
    // box, so the strided passes and the extraction behind them run on a fraction of the grid (line_fft.hip).
    if (req && own_fft_supported(plan->dim, g.nf) && std::getenv("EFGP_NO_PRUNED_FFT") == nullptr) {
        int64_t nc[3] = {1, 1, 1}, wk = nbatch;
        bool smaller = false;
        for (int a = 0; a < plan->dim; ++a) {
            int64_t hmax = req->ma.nm[a] / 2;
            if (req->part == 4) hmax = std::max(hmax, req->mb.nm[a] / 2);
            nc[a] = std::min<int64_t>(2 * hmax + 1, g.nf[a]);
            smaller = smaller || nc[a] < g.nf[a];
            wk *= a == plan->dim - 1 ? nc[a] : g.nf[a];
        }
        if (smaller) {
            double2* work = (double2*)scratch(plan->ctx, SLOT_FFT_WORK, (size_t)wk * sizeof(double2));
            if (!work) return EFGP_ENOMEM;
            double2* res = nullptr;
            const bool fuse = acc && g.nf[plan->dim - 1] > 1 && std::getenv("EFGP_NO_FFT_FROM_ACC") == nullptr;
            if (!fuse) {
                const int rcc = convert();
                if (rcc != EFGP_OK) return rcc;
            }
            const int rcp = own_fft_pruned_forward(plan->ctx, plan->dim, g.nf, nc, nbatch, fine, work, isign < 0, &res, stream, fuse ? acc : nullptr);
            if (rcp != EFGP_OK) return rcp;
            req->cropped = true;
            for (int a = 0; a < 3; ++a) req->crop_nf[a] = nc[a];
            *fine_out = res;
            return EFGP_OK;
        }
    }
    const int rcc = convert();
    if (rcc != EFGP_OK) return rcc;
    const int rc = fft_c2c(plan->ctx, plan->dim, g.nf, nbatch, fine, isign < 0, stream);      // in-house line kernels, hipFFT beyond their sizes
    if (rc != EFGP_OK) return rc;
    *fine_out = fine;
    return EFGP_OK;
}

// spread + reduce + FFT; leaves the transformed fine grids in SLOT_FINE
static int spread_and_fft(efgp_nufft_s* plan, WindowSet* w, const double* c, int mode, int nbatch, int isign,
                          hipStream_t stream, double2** fine_out, unsigned long long seed = 0, int64_t index_offset = 0,
                          const double** scale_out = nullptr, G2MRequest* req = nullptr) {
    DeviceCtx* ctx = plan->ctx;
    const GridGeom g = make_geom(plan, w);
    if (scale_out) *scale_out = nullptr;
    const int channels = (mode == STR_COMPLEX || mode == STR_REAL_AND_ONES || mode == STR_REAL_PAIR || mode == STR_RNG_PAIR) ? 2 : 1;
    // strengths read from memory need a max|c| pass for the fixed-point scale; generated / implicit ones are +-1
    const bool need_max = (mode == STR_REAL || mode == STR_COMPLEX || mode == STR_REAL_AND_ONES || mode == STR_REAL_PAIR);
    const double floor_bound = (mode == STR_REAL_AND_ONES || mode == STR_ONES || mode == STR_RNG || mode == STR_RNG_PAIR) ? 1.0 : 0.0;
    const int64_t nvals = (mode == STR_REAL_AND_ONES) ? plan->npts
                          : (int64_t)nbatch * plan->npts * ((mode == STR_COMPLEX || mode == STR_REAL_PAIR) ? 2 : 1);
    StrengthSrc src;
    src.c = c;
    src.npts = plan->npts;
    src.mode = mode;
    src.seed = seed;
    src.index_offset = index_offset;
    const size_t lds_bytes = (size_t)channels * (size_t)g.cells * sizeof(double);
    const bool use_lds = lds_bytes <= (size_t)ctx->max_lds && plan->npts > 0;
    // 2-D plans made on a per-model point layout (efgp_nufft_create_on): MFMA register accumulation over
    // (band, x_0)-sorted points, no per-plan sorting (spread_mfma.hip)
    int band_cells = 8;
    if (SortedLevel* lvl = pick_level(plan, w, g, stream, &band_cells)) {
        const double* ys = nullptr;
        if (c && c == plan->points->values && (mode == STR_REAL_AND_ONES || (mode == STR_REAL && nbatch == 1))) {
            int rc = points_level_values(plan->points, lvl, stream);
            if (rc != EFGP_OK) return rc;
            ys = lvl->ys;
        }
        const size_t acc_bytes = (size_t)nbatch * channels * (size_t)g.cells * sizeof(long long);
        const size_t known_zero = ctx->slabs_zero_bytes;
        const void* slabs_before = ctx->buf[SLOT_SLABS];
        unsigned long long* gacc = (unsigned long long*)scratch(ctx, SLOT_SLABS, acc_bytes);
        double2* fine = (double2*)scratch(ctx, SLOT_FINE, (size_t)nbatch * (size_t)g.cells * sizeof(double2));
        char* misc = scale_slot(ctx, stream);
        if (!gacc || !fine || !misc) return EFGP_ENOMEM;
        double* d_scale = (double*)misc;
        unsigned long long* d_cmax = (unsigned long long*)(misc + 56);
        if (scale_out) *scale_out = d_scale;
        // the converting kernel below zeroes what it reads: back-to-back passes of this path need no memset launch
        if (known_zero < acc_bytes || slabs_before != (const void*)gacc) EFGP_HIP_CHECK(hipMemsetAsync(gacc, 0, acc_bytes, stream));
        // the global int64 grid sums over ALL points: the scale is bounded with N
        ScaleJob job{floor_bound, mode == STR_REAL_AND_ONES ? 1 : 0, plan->npts, d_scale, 61};
        if (need_max && ys && plan->points->d_values_max && mode == STR_REAL_AND_ONES) {
            // the fit-time pair on the attached targets: max|y|, N and the bit budget are fixed per model, so the scale block is
            // computed once per attach and kept with the layout (one launch less per fit)
            efgp_points_s* pts = plan->points;
            d_scale = pts->d_pair_scale;
            job.scale = d_scale;
            if (scale_out) *scale_out = d_scale;
            if (!pts->pair_scale_ready) {
                hipLaunchKernelGGL(fixed_scale_cached_kernel, dim3(1), dim3(64), 0, stream, (const unsigned long long*)pts->d_values_max, job);
                pts->pair_scale_ready = true;
            }
        } else if (need_max && ys && plan->points->d_values_max) {
            hipLaunchKernelGGL(fixed_scale_cached_kernel, dim3(1), dim3(64), 0, stream,
                               (const unsigned long long*)plan->points->d_values_max, job);
        } else if (need_max) {
            const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((nvals + 8191) / 8192, 256));
            hipLaunchKernelGGL(maxabs_kernel, dim3(blocks), dim3(1024), 0, stream, c, nvals, d_cmax,
                               (unsigned int*)(misc + 48), job);
        } else {
            // generated +-1 probes / implicit ones: the block depends on (floor, N, bit budget) only -- constant for the layout
            efgp_points_s* pts = plan->points;
            if (floor_bound == 1.0 && job.ones_channel == 0) {
                if (!pts->d_fixed_scale) EFGP_HIP_CHECK(hipMalloc((void**)&pts->d_fixed_scale, 64));
                d_scale = pts->d_fixed_scale;
                job.scale = d_scale;
                if (scale_out) *scale_out = d_scale;
            }
            if (job.scale != pts->d_fixed_scale || !pts->fixed_scale_ready) {
                hipLaunchKernelGGL(fixed_scale_kernel, dim3(1), dim3(64), 0, stream, job.floor_bound, job.ones_channel, job.per,
                                   job.scale, job.sum_bits);
                if (job.scale == pts->d_fixed_scale) pts->fixed_scale_ready = true;
            }
        }
        EFGP_HIP_CHECK(hipGetLastError());
        int rc = spread_mfma_launch(ctx, lvl, band_cells, ys, src, g, w->p.w, w->d_coef, w->p.degree, channels, nbatch, gacc, d_scale, stream);
        if (rc != EFGP_OK) return rc;
        if (g2m_eligible(plan, g, req)) {
            rc = g2m_launch(ctx, g, req, (long long*)gacc, nullptr, d_scale, channels, nbatch, isign, (unsigned int*)(misc + 40),
                            (long long)(acc_bytes / sizeof(long long)), stream);
            if (rc != EFGP_OK) return rc;
            ctx->slabs_zero_bytes = zero_extent_after(acc_bytes, known_zero, slabs_before == (const void*)gacc);
            *fine_out = nullptr;
            return EFGP_OK;
        }
        const FftAccSource accsrc{(long long*)gacc, channels, g.cells, (const double*)d_scale, 1};     // converted (and cleared) by whoever reads it
        ctx->slabs_zero_bytes = zero_extent_after(acc_bytes, known_zero, slabs_before == (const void*)gacc);
        return transform_fine(plan, g, fine, nbatch, isign, stream, req, scale_out ? *scale_out : nullptr, fine_out, &accsrc);
    }
    // 2-D with many points per fine-grid cell: register accumulation over base-cell-sorted points
    {
        // Opt-in (EFGP_CELLSORT=1): the kernel itself is 1.2-1.7x faster than the LDS-atomic spreader at >= 250 points
        // per cell, but the per-plan counting sort it needs (0.15 ms at N=1e6, 0.6 ms at N=1e7) only pays off after
        // several passes over the same plan; see LABNOTES.md section 4.1.
        const char* force = std::getenv("EFGP_CELLSORT");
        const bool use_cells = force && force[0] == '1' && plan->dim == 2 && w->p.w <= kCellMaxW &&
                               w->p.degree <= w->p.w + 4 && g.cells <= 16384 && plan->npts > 0;
        TileGeom cg;
        if (use_cells && make_tile_geom(plan, w, channels, (size_t)ctx->max_lds - 4096, &cg, 1)) {
            BinSet* bins = nullptr;
            int rc = get_bins(plan, cg, channels, stream, &bins);
            if (rc != EFGP_OK) return rc;
            const size_t acc_bytes = (size_t)nbatch * channels * (size_t)g.cells * sizeof(double);
            double* gacc = (double*)scratch(ctx, SLOT_SLABS, acc_bytes);
            double2* fine = (double2*)scratch(ctx, SLOT_FINE, (size_t)nbatch * (size_t)g.cells * sizeof(double2));
            if (!gacc || !fine) return EFGP_ENOMEM;
            EFGP_HIP_CHECK(hipMemsetAsync(gacc, 0, acc_bytes, stream));
            CellSpreadArgs ca;
            ca.t = cg;
            ca.xs = bins->xs;
            ca.order = bins->order;
            ca.start = bins->start;
            ca.src = src;
            ca.coef = w->d_coef;
            ca.channels = channels;
            ca.gacc = gacc;
            ca.cells = g.cells;
            // enough wavefronts for a few rounds of the chip's resident waves, chunks long enough to amortise the flushes
            // one chunk per resident wave slot (2 per SIMD) when N is large: balanced, one prologue per wave
            int64_t chunk = (plan->npts + 8 * (int64_t)ctx->num_cu - 1) / (8 * (int64_t)ctx->num_cu);
            chunk = std::max<int64_t>(256, (chunk + 63) / 64 * 64);
            if (const char* ce = std::getenv("EFGP_CELL_CHUNK")) chunk = std::max(64, std::atoi(ce) / 64 * 64);   // diagnostics
            ca.npts = (int)plan->npts;
            ca.chunk = (int)chunk;
            const int64_t nwaves = (plan->npts + chunk - 1) / chunk;
            const int blocks = (int)((nwaves + 3) / 4);
            hipError_t e;
            {
                KernelTimer timer("spread", stream);
                e = launch_cell(w->p.w, channels, w->p.degree, dim3(blocks, nbatch), stream, ca);
            }
            if (e != hipSuccess) {
                set_error("cell-sorted spread kernel launch failed: %s", hipGetErrorString(e));
                return EFGP_EHIP;
            }
            const int rb = (int)((g.cells + 63) / 64);
            hipLaunchKernelGGL((reduce_slabs_kernel<false>), dim3(rb, nbatch), dim3(512), 0, stream, (const double*)gacc, 1,
                               channels, g.cells, (const double*)nullptr, fine);
            EFGP_HIP_CHECK(hipGetLastError());
            return transform_fine(plan, g, fine, nbatch, isign, stream, req, scale_out ? *scale_out : nullptr, fine_out);
        }
    }
    // grids beyond LDS: tile-sorted points + LDS tiles (large N), else global atomics (small N)
    TileGeom tg;
    const bool use_tiles = !use_lds && plan->npts >= 32768 && std::getenv("EFGP_NO_TILES") == nullptr &&
                           make_tile_geom(plan, w, channels, (size_t)ctx->max_lds - 4096, &tg);
    if (use_tiles) {
        BinSet* bins = nullptr;
        int rc = get_bins(plan, tg, channels, stream, &bins);
        if (rc != EFGP_OK) return rc;
        const size_t acc_bytes = (size_t)nbatch * channels * (size_t)g.cells * sizeof(long long);
        const size_t known_zero = ctx->slabs_zero_bytes;
        const void* slabs_before = ctx->buf[SLOT_SLABS];
        long long* gacc = (long long*)scratch(ctx, SLOT_SLABS, acc_bytes);
        double2* fine = (double2*)scratch(ctx, SLOT_FINE, (size_t)nbatch * (size_t)g.cells * sizeof(double2));
        char* misc = scale_slot(ctx, stream);
        if (!gacc || !fine || !misc) return EFGP_ENOMEM;
        double* d_scale = (double*)misc;
        unsigned long long* d_cmax = (unsigned long long*)(misc + 56);
        if (scale_out) *scale_out = d_scale;
        // the kernel that converts the accumulator clears what it reads: back-to-back passes need no memset launch
        if (known_zero < acc_bytes || slabs_before != (const void*)gacc) EFGP_HIP_CHECK(hipMemsetAsync(gacc, 0, acc_bytes, stream));
        // the global int64 grid sums over ALL points: bound the scale with N instead of points per workgroup
        const ScaleJob job{floor_bound, mode == STR_REAL_AND_ONES ? 1 : 0, plan->npts, d_scale, 61};
        if (need_max) {
            const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((nvals + 8191) / 8192, 256));
            hipLaunchKernelGGL(maxabs_kernel, dim3(blocks), dim3(1024), 0, stream, c, nvals, d_cmax,
                               (unsigned int*)(misc + 48), job);
        } else {
            hipLaunchKernelGGL(fixed_scale_kernel, dim3(1), dim3(64), 0, stream, job.floor_bound, job.ones_channel, job.per,
                               job.scale, job.sum_bits);
        }
        EFGP_HIP_CHECK(hipGetLastError());
        TileSpreadArgs ta;
        ta.t = tg;
        ta.xs = bins->xs;
        ta.order = bins->order;
        ta.start = bins->start;
        ta.src = src;
        ta.npts = plan->npts;
        // enough chunks to fill the chip, but long enough to amortise the tile flushes
        int64_t chunk = std::max<int64_t>(4096, (plan->npts + 4 * ctx->num_cu - 1) / (4 * (int64_t)ctx->num_cu));
        ta.chunk = chunk;
        ta.coef = w->d_coef;
        ta.degree = w->p.degree;
        ta.channels = channels;
        ta.gacc = gacc;
        ta.cells = g.cells;
        ta.scale = d_scale;
        const size_t tile_lds = (size_t)channels * tg.ext[0] * tg.ext[1] * tg.ext[2] * sizeof(double);
        dim3 grid((unsigned)((plan->npts + chunk - 1) / chunk), nbatch);
        hipError_t e;
        {
            KernelTimer timer("spread", stream);
            if (plan->dim == 1) e = launch_tile_d<1>(w->p.w, grid, tile_lds, stream, ta);
            else if (plan->dim == 2) e = launch_tile_d<2>(w->p.w, grid, tile_lds, stream, ta);
            else e = launch_tile_d<3>(w->p.w, grid, tile_lds, stream, ta);
        }
        if (e != hipSuccess) {
            set_error("tiled spread kernel launch failed: %s", hipGetErrorString(e));
            return EFGP_EHIP;
        }
        const FftAccSource accsrc{gacc, channels, g.cells, (const double*)d_scale, 1};     // converted (and cleared) by whoever reads it
        ctx->slabs_zero_bytes = zero_extent_after(acc_bytes, known_zero, slabs_before == (const void*)gacc);
        return transform_fine(plan, g, fine, nbatch, isign, stream, req, scale_out ? *scale_out : nullptr, fine_out, &accsrc);
    }
    int nwg = 1;
    if (use_lds) {
        int per_cu = std::max(1, std::min(2, (int)((size_t)ctx->max_lds / std::max<size_t>(lds_bytes, 1))));
        // LDS-atomic bound: spread the points over as many CUs as there are full 1024-point chunks (each workgroup also
        // pays a 147-KB zero + flush, ~5 us, so not below one point per thread)
        int64_t want = (plan->npts + kSpreadThreads - 1) / kSpreadThreads;
        nwg = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)ctx->num_cu * per_cu, want));
    } else {
        int64_t want = (plan->npts + kSpreadThreads - 1) / kSpreadThreads;
        nwg = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)ctx->num_cu * 8, want));
    }
    const int nslab = use_lds ? nwg : 1;
    const size_t slab_bytes = (size_t)nbatch * nslab * channels * (size_t)g.cells * sizeof(double);
    double* slabs = (double*)scratch(ctx, SLOT_SLABS, slab_bytes);
    double2* fine = (double2*)scratch(ctx, SLOT_FINE, (size_t)nbatch * (size_t)g.cells * sizeof(double2));
    char* misc = scale_slot(ctx, stream);
    if (!slabs || !fine || !misc) return EFGP_ENOMEM;
    double* d_scale = (double*)misc;                                  // [0] S0, [1] 1/S0, [2] S1, [3] 1/S1
    unsigned long long* d_cmax = (unsigned long long*)(misc + 56);
    if (!use_lds) EFGP_HIP_CHECK(hipMemsetAsync(slabs, 0, slab_bytes, stream));

    const int64_t per = (plan->npts + nwg - 1) / std::max(nwg, 1);
    // padded-row variant when the padded grid still fits LDS; 48-bit raw accumulation when its rounding
    // floor (points per workgroup * 2^-47 relative to max|c|) stays two orders below the requested tolerance
    const int64_t nlast = g.nf[plan->dim - 1];
    const size_t pad_bytes = (size_t)channels * (size_t)(g.cells / nlast) * (size_t)(nlast + w->p.w - 1) * sizeof(double);
    const bool use_pad = use_lds && pad_bytes + 4608 <= (size_t)ctx->max_lds && std::getenv("EFGP_NO_PAD") == nullptr;   // + class lists
    const bool raw48 = use_pad && (double)per * std::ldexp(1.0, -47) <= 0.01 * plan->tol && std::getenv("EFGP_NO_RAW48") == nullptr;
    if (plan->npts > 0) {      // (both forms: LDS tiles per workgroup, or one global int64 grid)
        if (scale_out) *scale_out = d_scale;
        // fixed-point scale from max |c| (device side, no host round trip)
        // raw48: every workgroup's sums stay below 2^46 and reduce_slabs_kernel adds at most 512 of them in int64.
        // Otherwise the bound must hold for the SUM OVER ALL SLABS (reduce_slabs_kernel adds them in plain int64):
        // bounding only one workgroup's share let same-sign strengths (the all-ones channel, clustered points) wrap
        // the total, e.g. 1-D, N = 1e6, tol <= 1e-9: 512 slabs x 2^61 / 1954 points each.
        const ScaleJob job{floor_bound, mode == STR_REAL_AND_ONES ? 1 : 0, raw48 ? per : plan->npts, d_scale, raw48 ? 46 : 61};
        if (need_max) {
            const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((nvals + 8191) / 8192, 256));
            hipLaunchKernelGGL(maxabs_kernel, dim3(blocks), dim3(1024), 0, stream, c, nvals, d_cmax,
                               (unsigned int*)(misc + 48), job);
        } else {
            hipLaunchKernelGGL(fixed_scale_kernel, dim3(1), dim3(64), 0, stream, job.floor_bound, job.ones_channel, job.per,
                               job.scale, job.sum_bits);
        }
        EFGP_HIP_CHECK(hipGetLastError());
    }

    SpreadArgs a;
    a.x = plan->x;
    a.src = src;
    a.npts = plan->npts;
    a.g = g;
    a.coef = w->d_coef;
    a.degree = w->p.degree;
    a.channels = channels;
    a.slabs = slabs;
    a.nslab = nslab;
    a.scale = d_scale;
    a.order = nullptr;
    // per-plan bank-balanced order (d >= 2, enough points for the one-off pass to pay)
    if (use_pad && plan->dim >= 2 && plan->npts >= (int64_t)kOrderWindow * 64 && std::getenv("EFGP_NO_CLASS_ORDER") == nullptr) {
        ClassOrder* co = nullptr;
        for (ClassOrder* o : plan->orders)
            if (o->W == w->p.w && o->nwg == nwg && o->nf[0] == g.nf[0] && o->nf[1] == g.nf[1] && o->nf[2] == g.nf[2]) co = o;
        if (!co) {
            co = new ClassOrder();
            for (int q = 0; q < 3; ++q) co->nf[q] = g.nf[q];
            co->W = w->p.w;
            co->nwg = nwg;
            co->bytes = (size_t)plan->npts * sizeof(int);
            co->order = (int*)pool_alloc(ctx, co->bytes);
            if (!co->order) {
                delete co;
                return EFGP_ENOMEM;
            }
            KernelTimer order_timer("order", stream);
            if (plan->dim == 2)
                hipLaunchKernelGGL((class_order_kernel<2, 16>), dim3(nwg), dim3(kSpreadThreads), 0, stream, g, w->p.w, plan->x, plan->npts, per,
                                   co->order);
            else
                hipLaunchKernelGGL((class_order_kernel<3, 16>), dim3(nwg), dim3(kSpreadThreads), 0, stream, g, w->p.w, plan->x, plan->npts, per,
                                   co->order);
            EFGP_HIP_CHECK(hipGetLastError());
            plan->orders.push_back(co);
        }
        a.order = co->order;
    }
    dim3 grid(nwg, nbatch);
    hipError_t e = hipSuccess;
    if (plan->npts > 0 && use_pad) {
        KernelTimer timer("spread", stream);
        if (plan->dim == 1) e = launch_spread_pad_d<1>(w->p.w, raw48, grid, pad_bytes, stream, a);
        else if (plan->dim == 2) e = launch_spread_pad_d<2>(w->p.w, raw48, grid, pad_bytes, stream, a);
        else e = launch_spread_pad_d<3>(w->p.w, raw48, grid, pad_bytes, stream, a);
    } else if (plan->npts > 0) {
        KernelTimer timer("spread", stream);
        if (plan->dim == 1) e = launch_spread_d<1>(w->p.w, use_lds, grid, use_lds ? lds_bytes : 0, stream, a);
        else if (plan->dim == 2) e = launch_spread_d<2>(w->p.w, use_lds, grid, use_lds ? lds_bytes : 0, stream, a);
        else e = launch_spread_d<3>(w->p.w, use_lds, grid, use_lds ? lds_bytes : 0, stream, a);
    } else {
        EFGP_HIP_CHECK(hipMemsetAsync(slabs, 0, slab_bytes, stream));
    }
    if (e != hipSuccess) {
        set_error("spread kernel launch failed: %s", hipGetErrorString(e));
        return EFGP_EHIP;
    }
    {
        const int blocks = (int)((g.cells + 63) / 64);
        if (plan->npts > 0)
            hipLaunchKernelGGL((reduce_slabs_kernel<true>), dim3(blocks, nbatch), dim3(nslab >= 64 ? 1024 : 512), 0, stream,
                               slabs, nslab, channels, g.cells, (const double*)d_scale, fine);
        else
            hipLaunchKernelGGL((reduce_slabs_kernel<false>), dim3(blocks, nbatch), dim3(512), 0, stream, slabs, nslab,
                               channels, g.cells, (const double*)d_scale, fine);
        EFGP_HIP_CHECK(hipGetLastError());
    }
    return transform_fine(plan, g, fine, nbatch, isign, stream, req, scale_out ? *scale_out : nullptr, fine_out);
}

static int run_deconvolve(efgp_nufft_s* plan, WindowSet* w, const double2* fine, const int64_t* nm, int modeord,
                          int part, int nbatch, void* out, hipStream_t stream, int rows_limit = 1 << 30, const G2MRequest* req = nullptr) {
    ModeGeom m = make_modes(plan, w, nm, modeord);
    int64_t cells = 1;
    for (int a = 0; a < 3; ++a) cells *= w->nf[a];
    apply_crop(req, m, cells);
    int threads = 256;
    int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((m.total + threads - 1) / threads, 2048));
    hipLaunchKernelGGL(deconvolve_kernel, dim3(blocks, nbatch), dim3(threads), 0, stream, fine, cells, m, part,
                       (double2*)out, rows_limit);
    EFGP_HIP_CHECK(hipGetLastError());
    return EFGP_OK;
}

}  // namespace efgp

extern "C" {

int efgp_nufft_create(efgp_nufft_t** plan_out, int device, int dim, int64_t npts, const double* x,
                      const double* xcen_host, double h, double tol) {
    EFGP_REQUIRE(plan_out, "efgp_nufft_create: null plan_out");
    EFGP_REQUIRE(dim >= 1 && dim <= 3, "efgp_nufft_create: dim must be 1, 2 or 3 (got %d)", dim);
    EFGP_REQUIRE(npts >= 0, "efgp_nufft_create: negative point count");
    EFGP_REQUIRE(npts == 0 || x, "efgp_nufft_create: null x");
    EFGP_REQUIRE(std::isfinite(h), "efgp_nufft_create: h must be finite");
    DeviceCtx* ctx = device_ctx(device);
    if (!ctx) return EFGP_EHIP;
    auto* p = new efgp_nufft_s();
    p->device = device;
    p->dim = dim;
    p->npts = npts;
    p->x = x;
    p->h = h;
    p->tol = tol;
    p->ctx = ctx;
    for (int a = 0; a < dim; ++a) p->xcen[a] = xcen_host ? xcen_host[a] : 0.0;
    *plan_out = p;
    return EFGP_OK;
}

int efgp_nufft_create_on(efgp_nufft_t** plan_out, efgp_points_t* pts, const double* xcen_host, double h, double tol) {
    EFGP_REQUIRE(pts, "efgp_nufft_create_on: null point layout");
    int rc = efgp_nufft_create(plan_out, pts->device, pts->dim, pts->npts, pts->x, xcen_host, h, tol);
    if (rc != EFGP_OK) return rc;
    (*plan_out)->points = pts;
    return EFGP_OK;
}

int efgp_nufft_destroy(efgp_nufft_t* plan) {
    if (!plan) return EFGP_OK;
    DeviceGuard guard(plan->device);
    for (BinSet* b : plan->bins) free_binset(plan->ctx, b);
    for (efgp::ClassOrder* o : plan->orders) {
        pool_free(plan->ctx, o->order, o->bytes);
        delete o;
    }
    delete plan;
    return EFGP_OK;
}

// shared by the strengths-from-memory and the generated-probe entry points: real rows are processed two
// per fine grid (re/im channels, separated after the FFT by Hermitian symmetry), an odd last row alone
static int type1_real_rows(efgp_nufft_s* plan, WindowSet* w, const double* c, bool rng, unsigned long long seed,
                           int64_t index_offset, int nbatch, const int64_t* n_modes, int isign, int modeord, void* out,
                           hipStream_t stream) {
    int64_t total = 1;
    for (int a = 0; a < plan->dim; ++a) total *= n_modes[a];
    // Generated probes, odd count, 2-D: the last row rides as the real part of one more pair grid whose imaginary row (index nbatch)
    // is generated and dropped -- one pass and one transform chain instead of two (6 launches; the 2-D spread costs the same per
    // grid with one channel or two).  Rows read from memory have no row `nbatch` to read, and the 1-D / 3-D passes pay per channel.
    const bool pad_odd = rng && (nbatch & 1) && nbatch > 1 && plan->dim == 2;
    const int npair = pad_odd ? (nbatch + 1) / 2 : nbatch / 2;
    double2* fine = nullptr;
    if (npair > 0) {
        G2MRequest req;
        req.part = 3;
        req.rows_limit = nbatch;
        req.ma = make_modes(plan, w, n_modes, modeord);
        req.out_a = out;
        int rc = spread_and_fft(plan, w, c, rng ? STR_RNG_PAIR : STR_REAL_PAIR, npair, isign, stream, &fine, seed, index_offset, nullptr,
                                &req);
        if (rc != EFGP_OK) return rc;
        // for isign = +1 the roles of k and -k swap in the Hermitian split; conjugating H handles both signs:
        // the split below assumes the forward (isign = -1) transform, which is what the reference uses for type 1
        if (!req.done) rc = run_deconvolve(plan, w, fine, n_modes, modeord, 3, npair, out, stream, nbatch, &req);
        if (rc != EFGP_OK) return rc;
    }
    if ((nbatch & 1) && !pad_odd) {
        const int last = nbatch - 1;
        int rc;
        G2MRequest req;
        req.part = 0;
        req.ma = make_modes(plan, w, n_modes, modeord);
        req.out_a = (double2*)out + (int64_t)last * total;
        if (rng) {
            // STR_RNG takes the fine-grid index (0 here) as the row: row `last` of the same seed is row 0 at the point index shifted
            // by last * kRowStride (efgp_rademacher hashes row * kRowStride + index in wrapping 64-bit arithmetic)
            rc = spread_and_fft(plan, w, nullptr, STR_RNG, 1, isign, stream, &fine, seed,
                                (int64_t)((unsigned long long)index_offset + (unsigned long long)last * kRademacherRowStride), nullptr,
                                &req);
        } else {
            rc = spread_and_fft(plan, w, c + (int64_t)last * plan->npts, STR_REAL, 1, isign, stream, &fine, 0, 0, nullptr, &req);
        }
        if (rc != EFGP_OK) return rc;
        if (!req.done) rc = run_deconvolve(plan, w, fine, n_modes, modeord, 0, 1, (double2*)out + (int64_t)last * total, stream, 1 << 30, &req);
        if (rc != EFGP_OK) return rc;
    }
    return EFGP_OK;
}

int efgp_nufft_type1(efgp_nufft_t* plan, const void* c, int c_is_complex, int nbatch, const int64_t* n_modes,
                     int isign, int modeord, void* out, void* stream_) {
    EFGP_REQUIRE(plan && n_modes && out, "efgp_nufft_type1: null argument");
    EFGP_REQUIRE(nbatch >= 1, "efgp_nufft_type1: nbatch must be >= 1");
    EFGP_REQUIRE(plan->npts == 0 || c, "efgp_nufft_type1: null strengths");
    for (int a = 0; a < plan->dim; ++a) EFGP_REQUIRE(n_modes[a] >= 1, "efgp_nufft_type1: n_modes[%d] < 1", a);
    EFGP_REQUIRE(isign == 1 || isign == -1, "efgp_nufft_type1: isign must be +-1");
    hipStream_t stream = (hipStream_t)stream_;
    DeviceGuard guard(plan->device, (hipStream_t)stream_);
    WindowSet* w = nullptr;
    int rc = get_window(plan, n_modes, stream, &w);
    if (rc != EFGP_OK) return rc;
    if (!c_is_complex && isign == -1 && plan->npts > 0)
        return type1_real_rows(plan, w, (const double*)c, false, 0, 0, nbatch, n_modes, isign, modeord, out, stream);
    double2* fine = nullptr;
    G2MRequest req;
    req.part = 0;
    req.ma = make_modes(plan, w, n_modes, modeord);
    req.out_a = out;
    rc = spread_and_fft(plan, w, (const double*)c, c_is_complex ? STR_COMPLEX : STR_REAL, nbatch, isign, stream, &fine, 0, 0, nullptr, &req);
    if (rc != EFGP_OK) return rc;
    if (req.done) return EFGP_OK;
    return run_deconvolve(plan, w, fine, n_modes, modeord, 0, nbatch, out, stream, 1 << 30, &req);
}

int efgp_nufft_type1_rademacher(efgp_nufft_t* plan, uint64_t seed, int64_t index_offset, int nbatch,
                                const int64_t* n_modes, int modeord, void* out, void* stream_) {
    EFGP_REQUIRE(plan && n_modes && out, "efgp_nufft_type1_rademacher: null argument");
    EFGP_REQUIRE(nbatch >= 1, "efgp_nufft_type1_rademacher: nbatch must be >= 1");
    for (int a = 0; a < plan->dim; ++a) EFGP_REQUIRE(n_modes[a] >= 1, "efgp_nufft_type1_rademacher: n_modes[%d] < 1", a);
    hipStream_t stream = (hipStream_t)stream_;
    DeviceGuard guard(plan->device, (hipStream_t)stream_);
    WindowSet* w = nullptr;
    int rc = get_window(plan, n_modes, stream, &w);
    if (rc != EFGP_OK) return rc;
    if (plan->npts == 0) {
        int64_t total = 1;
        for (int a = 0; a < plan->dim; ++a) total *= n_modes[a];
        EFGP_HIP_CHECK(hipMemsetAsync(out, 0, (size_t)nbatch * total * sizeof(double2), stream));
        return EFGP_OK;
    }
    return type1_real_rows(plan, w, nullptr, true, (unsigned long long)seed, index_offset, nbatch, n_modes, -1, modeord, out,
                           stream);
}

int efgp_rademacher_fill(int device, uint64_t seed, int64_t index_offset, int nbatch, int64_t npts, double* out,
                         void* stream_) {
    EFGP_REQUIRE(out || npts == 0, "efgp_rademacher_fill: null out");
    EFGP_REQUIRE(nbatch >= 1 && npts >= 0, "efgp_rademacher_fill: bad sizes");
    if (npts == 0) return EFGP_OK;
    if (!device_ctx(device)) return EFGP_EHIP;
    DeviceGuard guard(device, (hipStream_t)stream_);
    hipStream_t stream = (hipStream_t)stream_;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((npts + 255) / 256, 4096));
    hipLaunchKernelGGL(rademacher_fill_kernel, dim3(blocks, nbatch), dim3(256), 0, stream, (unsigned long long)seed, npts, index_offset,
                       out);
    EFGP_HIP_CHECK(hipGetLastError());
    return EFGP_OK;
}

int efgp_nufft_type1_pair(efgp_nufft_t* plan, const double* y, const int64_t* n_modes_y, void* out_y,
                          const int64_t* n_modes_one, void* out_ones, void* stream_) {
    EFGP_REQUIRE(plan, "efgp_nufft_type1_pair: null plan");
    EFGP_REQUIRE(out_y || out_ones, "efgp_nufft_type1_pair: nothing to compute");
    EFGP_REQUIRE(!out_y || (y && n_modes_y), "efgp_nufft_type1_pair: y / n_modes_y missing");
    EFGP_REQUIRE(!out_ones || n_modes_one, "efgp_nufft_type1_pair: n_modes_one missing");
    hipStream_t stream = (hipStream_t)stream_;
    DeviceGuard guard(plan->device, (hipStream_t)stream_);
    // one fine grid sized for the larger box serves both
    int64_t box[3] = {1, 1, 1};
    for (int a = 0; a < plan->dim; ++a) {
        int64_t m = 1;
        if (out_y) m = std::max(m, n_modes_y[a]);
        if (out_ones) m = std::max(m, n_modes_one[a]);
        EFGP_REQUIRE(m >= 1, "efgp_nufft_type1_pair: bad mode count");
        box[a] = m;
    }
    WindowSet* w = nullptr;
    int rc = get_window(plan, box, stream, &w);
    if (rc != EFGP_OK) return rc;
    int mode = (out_y && out_ones) ? STR_REAL_AND_ONES : (out_y ? STR_REAL : STR_ONES);
    double2* fine = nullptr;
    const double* pair_scale = nullptr;
    G2MRequest req;
    G2MRequest* reqp = nullptr;
    if (mode == STR_REAL_AND_ONES) {
        req.part = 4;
        req.ma = make_modes(plan, w, n_modes_y, 0);
        req.mb = make_modes(plan, w, n_modes_one, 0);
        for (int a = 0; a < plan->dim; ++a) {
            req.ma.fac[a] = w->d_fac[a] + (box[a] / 2 - n_modes_y[a] / 2);
            req.mb.fac[a] = w->d_fac[a] + (box[a] / 2 - n_modes_one[a] / 2);
        }
        req.out_a = out_y;
        req.out_b = out_ones;
        reqp = &req;
    }
    rc = spread_and_fft(plan, w, y, mode, 1, -1, stream, &fine, 0, 0, &pair_scale, reqp);
    if (rc != EFGP_OK) return rc;
    if (req.done) return EFGP_OK;
    // correction factors were built for `box`; a smaller centred box indexes them with an offset
    auto sub = [&](const int64_t* nm, int part, void* out) -> int {
        ModeGeom m = make_modes(plan, w, nm, 0);
        for (int a = 0; a < plan->dim; ++a) m.fac[a] = w->d_fac[a] + (box[a] / 2 - nm[a] / 2);
        int64_t cells = 1;
        for (int a = 0; a < 3; ++a) cells *= w->nf[a];
        int threads = 256;
        int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((m.total + threads - 1) / threads, 2048));
        hipLaunchKernelGGL(deconvolve_kernel, dim3(blocks, 1), dim3(threads), 0, stream, (const double2*)fine, cells, m,
                           part, (double2*)out, 1 << 30);
        EFGP_HIP_CHECK(hipGetLastError());
        return EFGP_OK;
    };
    if (mode == STR_REAL_AND_ONES) {
        ModeGeom ma = make_modes(plan, w, n_modes_y, 0), mb = make_modes(plan, w, n_modes_one, 0);
        for (int a = 0; a < plan->dim; ++a) {
            ma.fac[a] = w->d_fac[a] + (box[a] / 2 - n_modes_y[a] / 2);
            mb.fac[a] = w->d_fac[a] + (box[a] / 2 - n_modes_one[a] / 2);
        }
        int64_t cells = 1;
        for (int a = 0; a < 3; ++a) cells *= w->nf[a];
        apply_crop(reqp, ma, cells);
        apply_crop(reqp, mb, cells);
        const int64_t most = std::max(ma.total, mb.total);
        const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((most + 255) / 256, 2048));
        hipLaunchKernelGGL(deconvolve_pair_kernel, dim3(blocks, 2), dim3(256), 0, stream, (const double2*)fine, cells, ma,
                           (double2*)out_y, mb, (double2*)out_ones, pair_scale);
        EFGP_HIP_CHECK(hipGetLastError());
        return EFGP_OK;
    }
    return out_y ? sub(n_modes_y, 0, out_y) : sub(n_modes_one, 0, out_ones);
}

static int type2_impl(efgp_nufft_t* plan, const void* f, const void* mode_scale, int nbatch, const int64_t* n_modes, int isign,
                      int modeord, void* out, int real_only, void* stream_);

int efgp_nufft_type2(efgp_nufft_t* plan, const void* f, int nbatch, const int64_t* n_modes, int isign, int modeord,
                     void* out, int real_only, void* stream_) {
    return type2_impl(plan, f, nullptr, nbatch, n_modes, isign, modeord, out, real_only, stream_);
}

int efgp_nufft_type2_scaled(efgp_nufft_t* plan, const void* f, const void* mode_scale, int nbatch, const int64_t* n_modes,
                            int isign, int modeord, void* out, int real_only, void* stream_) {
    EFGP_REQUIRE(mode_scale, "efgp_nufft_type2_scaled: null mode_scale");
    return type2_impl(plan, f, mode_scale, nbatch, n_modes, isign, modeord, out, real_only, stream_);
}

static int type2_impl(efgp_nufft_t* plan, const void* f, const void* mode_scale, int nbatch, const int64_t* n_modes, int isign,
                      int modeord, void* out, int real_only, void* stream_) {
    EFGP_REQUIRE(plan && f && n_modes, "efgp_nufft_type2: null argument");
    EFGP_REQUIRE(nbatch >= 1, "efgp_nufft_type2: nbatch must be >= 1");
    EFGP_REQUIRE(plan->npts == 0 || out, "efgp_nufft_type2: null out");
    for (int a = 0; a < plan->dim; ++a) EFGP_REQUIRE(n_modes[a] >= 1, "efgp_nufft_type2: n_modes[%d] < 1", a);
    EFGP_REQUIRE(isign == 1 || isign == -1, "efgp_nufft_type2: isign must be +-1");
    if (plan->npts == 0) return EFGP_OK;
    hipStream_t stream = (hipStream_t)stream_;
    DeviceGuard guard(plan->device, (hipStream_t)stream_);
    DeviceCtx* ctx = plan->ctx;
    WindowSet* w = nullptr;
    int rc = get_window(plan, n_modes, stream, &w);
    if (rc != EFGP_OK) return rc;
    const GridGeom g = make_geom(plan, w);
    double2* fine = (double2*)scratch(ctx, SLOT_FINE, (size_t)nbatch * (size_t)g.cells * sizeof(double2));
    if (!fine) return EFGP_ENOMEM;
    ModeGeom m = make_modes(plan, w, n_modes, modeord);
    // small 2-D real-output transforms: the real fine grid by ONE dense-DFT launch instead of precorrect + two rocFFT kernels
    const bool direct_grid = real_only && nbatch == 1 && plan->dim == 2 &&
                             modes_to_grid_real_eligible((int)g.nf[0], (int)g.nf[1], (int)n_modes[0], (int)n_modes[1]);
    if (direct_grid) {
        rc = modes_to_grid_real_launch(ctx, (const double2*)f, (const double2*)mode_scale, (int)n_modes[0], (int)n_modes[1], modeord, isign,
                                       w->d_fac[0], w->d_fac[1], (int)g.nf[0], (int)g.nf[1], fine, stream);
        if (rc != EFGP_OK) return rc;
    } else {
        int threads = 256;
        // In-house pruned transform: the corrected modes go to a compact array (the 2 (nm/2) + 1 lowest bins per axis), the passes
        // expand it axis by axis, slowest first -- only the last, contiguous pass writes the full grid (line_fft.hip).
        int64_t nc[3] = {1, 1, 1}, ccells = 1, region = nbatch;
        bool smaller = false;
        for (int a = 0; a < plan->dim; ++a) {
            nc[a] = std::min<int64_t>(2 * (n_modes[a] / 2) + 1, g.nf[a]);
            smaller = smaller || nc[a] < g.nf[a];
            ccells *= nc[a];
            region *= a == plan->dim - 1 ? nc[a] : g.nf[a];
        }
        if (smaller && own_fft_supported(plan->dim, g.nf) && std::getenv("EFGP_NO_PRUNED_FFT") == nullptr) {
            double2* work = (double2*)scratch(ctx, SLOT_FFT_WORK, (size_t)2 * (size_t)region * sizeof(double2));
            if (!work) return EFGP_ENOMEM;
            ModeGeom mc = m;
            for (int a = 0; a < plan->dim; ++a) mc.nf[a] = nc[a];
            int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((ccells + threads - 1) / threads, 2048));
            hipLaunchKernelGGL(precorrect_kernel, dim3(blocks, nbatch), dim3(threads), 0, stream, (const double2*)f,
                               (const double2*)mode_scale, mc, real_only ? 1 : 0, ccells, work);
            EFGP_HIP_CHECK(hipGetLastError());
            rc = own_fft_pruned_backward(ctx, plan->dim, nc, g.nf, nbatch, work, fine, work, region, isign < 0, stream);
            if (rc != EFGP_OK) return rc;
        } else {
            int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((g.cells + threads - 1) / threads, 2048));
            hipLaunchKernelGGL(precorrect_kernel, dim3(blocks, nbatch), dim3(threads), 0, stream, (const double2*)f,
                               (const double2*)mode_scale, m, real_only ? 1 : 0, g.cells, fine);
            EFGP_HIP_CHECK(hipGetLastError());
            rc = fft_c2c(ctx, plan->dim, g.nf, nbatch, fine, isign < 0, stream);
            if (rc != EFGP_OK) return rc;
        }
    }
    const bool cplx = !real_only;
    size_t lds_bytes = (size_t)g.cells * (cplx ? sizeof(double2) : sizeof(double));
    // real outputs: halo-padded LDS copy when it fits (no wrap arithmetic in the gather)
    size_t halo_cells = 1;
    for (int a_ = 0; a_ < plan->dim; ++a_) halo_cells *= (size_t)(g.nf[a_] + w->p.w - 1);
    const bool use_halo = !cplx && halo_cells * sizeof(double) <= (size_t)ctx->max_lds && std::getenv("EFGP_NO_HALO") == nullptr;
    if (use_halo) lds_bytes = halo_cells * sizeof(double);
    // 2-D real outputs whose two parity copies fit LDS: aligned 16-byte reads, no processing order
    const size_t pair_bytes = plan->dim == 2 ? interp_pair_lds_bytes((int)g.nf[0], (int)g.nf[1], w->p.w) : 0;
    const bool use_pair = use_halo && plan->dim == 2 && pair_bytes <= (size_t)ctx->max_lds && std::getenv("EFGP_NO_PAIR_GATHER") == nullptr;
    if (use_pair) lds_bytes = pair_bytes;
    const bool use_lds = lds_bytes <= (size_t)ctx->max_lds;
    // grids beyond LDS: tile-sorted points + LDS tiles (the binning is shared with the tiled spreader when the
    // tile geometry coincides, and cached in the plan otherwise)
    TileGeom tg;
    if (!use_lds && plan->npts >= 32768 && std::getenv("EFGP_NO_TILES") == nullptr &&
        make_tile_geom(plan, w, cplx ? 2 : 1, (size_t)ctx->max_lds - 4096, &tg)) {
        BinSet* bins = nullptr;
        rc = get_bins(plan, tg, cplx ? 2 : 1, stream, &bins);
        if (rc != EFGP_OK) return rc;
        TileInterpArgs ta;
        ta.t = tg;
        ta.xs = bins->xs;
        ta.order = bins->order;
        ta.start = bins->start;
        ta.npts = plan->npts;
        ta.chunk = std::max<int64_t>(4096, (plan->npts + 4 * ctx->num_cu - 1) / (4 * (int64_t)ctx->num_cu));
        ta.coef = w->d_coef;
        ta.degree = w->p.degree;
        ta.fine = fine;
        ta.cells = g.cells;
        ta.out = out;
        const size_t tile_lds = (size_t)(cplx ? 2 : 1) * tg.ext[0] * tg.ext[1] * tg.ext[2] * sizeof(double);
        dim3 tgrid((unsigned)((plan->npts + ta.chunk - 1) / ta.chunk), nbatch);
        hipError_t te;
        {
            KernelTimer timer("interp", stream);
            if (plan->dim == 1) te = launch_interp_tile_d<1>(w->p.w, cplx, tgrid, tile_lds, stream, ta);
            else if (plan->dim == 2) te = launch_interp_tile_d<2>(w->p.w, cplx, tgrid, tile_lds, stream, ta);
            else te = launch_interp_tile_d<3>(w->p.w, cplx, tgrid, tile_lds, stream, ta);
        }
        if (te != hipSuccess) {
            set_error("tiled interp kernel launch failed: %s", hipGetErrorString(te));
            return EFGP_EHIP;
        }
        return EFGP_OK;
    }
    InterpArgs a;
    a.x = plan->x;
    a.npts = plan->npts;
    a.g = g;
    a.coef = w->d_coef;
    a.degree = w->p.degree;
    a.fine = fine;
    a.out = out;
    const int thr = use_lds ? kInterpThreads : kInterpThreadsGlobal;
    int64_t want = (plan->npts + thr - 1) / thr;
    int nwg;
    if (use_lds) {
        int per_cu = std::max(1, std::min(4, (int)((size_t)ctx->max_lds / std::max<size_t>(lds_bytes, 1))));
        // each workgroup pays one fine-grid copy into LDS: keep >= 2 points per thread
        int64_t cap = std::max<int64_t>(1, plan->npts / (2 * kInterpThreads));
        nwg = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>((int64_t)ctx->num_cu * per_cu, want), cap));
    } else {
        nwg = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)ctx->num_cu * 8, want));
    }
    a.order = nullptr;
    if (use_halo && !use_pair && plan->dim >= 2 && plan->npts >= (int64_t)kOrderWindow * 64 && std::getenv("EFGP_NO_CLASS_ORDER") == nullptr) {
        // same classes as the padded spreader (row pitch nf + W - 1), over global windows: nwg = 0 marks that layout
        ClassOrder* co = nullptr;
        for (ClassOrder* o : plan->orders)
            if (o->W == w->p.w && o->nwg == 0 && o->nf[0] == g.nf[0] && o->nf[1] == g.nf[1] && o->nf[2] == g.nf[2]) co = o;
        if (!co) {
            co = new ClassOrder();
            for (int q = 0; q < 3; ++q) co->nf[q] = g.nf[q];
            co->W = w->p.w;
            co->nwg = 0;
            co->bytes = (size_t)plan->npts * sizeof(int);
            co->order = (int*)pool_alloc(ctx, co->bytes);
            if (!co->order) {
                delete co;
                return EFGP_ENOMEM;
            }
            const unsigned nwin = (unsigned)((plan->npts + kOrderWindow - 1) / kOrderWindow);
            KernelTimer order_timer("order", stream);
            if (plan->dim == 2)
                hipLaunchKernelGGL((class_order_kernel<2, 16>), dim3(nwin), dim3(kSpreadThreads), 0, stream, g, w->p.w, plan->x, plan->npts,
                                   (int64_t)kOrderWindow, co->order);
            else
                hipLaunchKernelGGL((class_order_kernel<3, 16>), dim3(nwin), dim3(kSpreadThreads), 0, stream, g, w->p.w, plan->x, plan->npts,
                                   (int64_t)kOrderWindow, co->order);
            EFGP_HIP_CHECK(hipGetLastError());
            plan->orders.push_back(co);
        }
        a.order = co->order;
    }
    dim3 grid(nwg, nbatch);
    hipError_t e;
    KernelTimer timer("interp", stream);
    if (use_pair) {
        e = launch_interp_pair(w->p.w, grid, lds_bytes, stream, a);
    } else if (use_halo) {
        if (plan->dim == 1) e = launch_interp_halo_d<1>(w->p.w, grid, lds_bytes, stream, a);
        else if (plan->dim == 2) e = launch_interp_halo_d<2>(w->p.w, grid, lds_bytes, stream, a);
        else e = launch_interp_halo_d<3>(w->p.w, grid, lds_bytes, stream, a);
    } else if (plan->dim == 1) e = launch_interp_d<1>(w->p.w, cplx, use_lds, grid, use_lds ? lds_bytes : 0, stream, a);
    else if (plan->dim == 2) e = launch_interp_d<2>(w->p.w, cplx, use_lds, grid, use_lds ? lds_bytes : 0, stream, a);
    else e = launch_interp_d<3>(w->p.w, cplx, use_lds, grid, use_lds ? lds_bytes : 0, stream, a);
    if (e != hipSuccess) {
        set_error("interp kernel launch failed: %s", hipGetErrorString(e));
        return EFGP_EHIP;
    }
    return EFGP_OK;
}

}  // extern "C"
