// Type-1 spreading with MFMA register accumulation over (band, x_0)-sorted points (2-D, window width <= 8).
//
// Replaces the FINUFFT type-1 call of the reference (efgpnd.py:1496-1499) for the N-scale pass of a fit
// (F*y and the Toeplitz vector in one pass, efgpnd.py:786, 789-790) and for the probe transforms of the
// hyper-gradient (efgpnd.py:186-189).
//
// Why a matrix instruction: the contribution of a point to the fine grid is the outer product
// (c wx) (x) wy of two 8-vectors, and the sum over the points that share a stencil is a rank-K update
// D += A B with K = points -- exactly what v_mfma_f64_16x16x4_f64 computes (4 points per instruction), with the
// reduction over points done inside the matrix unit instead of by LDS atomics (the LDS-atomic spreader spends
// 2 W^2 = 128 ds_add_u64 per point and is bound by LDS bank conflicts, LABNOTES.md section 4.1).
//   rows    (16) = (channel, x stencil cell): A[(ch, i)][k] = c_ch,k * wx_k[i]
//   columns (16) = y stencil cell relative to the band's first one: B[k][j + off_k] = wy_k[j], 0 <= off_k <= 8
// The tile belongs to one x-cell run of one band of the per-model point layout (points_layout.hpp); it is added
// to the global int64 fine grid (exact fixed point, order independent) when the run ends: one flush of 256 values
// per ~1000 points instead of 128 atomics per point.  On gfx950 the fp64 matrix and vector pipes are the same
// unit (tools/mfma_f64_bench.hip: an independent v_fma_f64 next to the MFMA costs its full issue time), so the
// budget per point is  16 Horner polynomials (lane = point, no redundancy: 2.5 wave-FMAs)  +  1/4 MFMA.
//
// Per wave, no workgroup barrier anywhere: a wave owns chunks of sorted points (grid-stride) and a private LDS
// region; per batch of 64 points, phase 1 (lane = point) evaluates the windows and writes the operand rows to
// LDS [component][point]; phase 2 (lane = (k, r)) reads its A / B element of 4 points per step and issues the
// MFMA.  LDS row stride 66 doubles: the 16 rows x 2 points read by one 32-lane group hit 32 distinct bank pairs.
#include "spread_mfma.hpp"

#include <algorithm>
#include <cstdlib>

namespace efgp {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int kMfmaRow = 66;                  // doubles per LDS row: 64 points + 2 (bank spread)
constexpr int kMfmaMaxWaves = 4;              // waves per workgroup (independent; 3 or 4, chosen at launch)
// LDS rows of a wave: A = (channel, x cell) 16 rows, then B = y cell + offset, W + HB rows, where HB bounds the height of a band
// in fine cells (the stencils of a band start at offsets 0..HB).  HB = 8: any band level, B fills the tile's 16 columns; HB = 1
// (dense point sets, round 3): 9 rows at W = 8 -- 13.2 KB per wave instead of 16.9, seven zero-fill stores less per point, and
// with 168 VGPRs THREE waves per SIMD instead of two.  The x cell of every point (64 ints) lives in the rows' padding doubles.
__host__ __device__ constexpr int mfma_rows(int W, int HB) { return 16 + W + HB; }
static size_t mfma_lds_bytes(int waves, int W, int HB) { return (size_t)waves * mfma_rows(W, HB) * kMfmaRow * sizeof(double); }

struct MfmaSpreadArgs {
    const double* xs;
    const double* ys;          // strengths in level order (null: fetch through src + perm)
    const int* perm;
    const int* chunks;
    int nchunks;
    int total_waves;
    const double* band_lo;
    StrengthSrc src;
    double scale0, scale1, xcen0, xcen1;      // X = scale * (x - xcen) in fine-grid cells
    int nf0, nf1;
    const double* coef;
    int degree;
    int channels;
    unsigned long long* gacc;
    const double* scale;
    int diag;                  // diagnostic build only (-DEFGP_MFMA_DIAG, env EFGP_MFMA_DIAG): 1 skips the MFMA phase, 2 the window polynomials, 4 the LDS writes
};

// Window values of both dimensions from the offsets s in [-1, 1): symmetric Horner (see window_eval).
// DEG > 0: the degree is a compile-time constant and the ceil(W/2) x (DEG + 1) coefficients live in VGPRs for the
// whole kernel (loaded once per wave).  Measured alternatives at N = 1e7 (Horner share of the launch): scalar loads
// per degree (as window_eval does) 65 us -- with two waves per SIMD the scalar-cache round trips sit in the dependent
// chain of every batch; a table in LDS read by broadcast 68 us (the compiler keeps the reads one step ahead only);
// registers 37 us.  DEG == 0: run-time degree through scalar loads (uncommon degrees).
template <int W, int DEG>
struct WindowCoef {
    static constexpr int RH = (W + 1) / 2, RHP = sym_row(W);
    double cf[DEG > 0 ? (DEG + 1) * RH : 1];
    __device__ __forceinline__ void load(const double* __restrict__ coef_generic) {
        if (DEG > 0) {
            // a zero the compiler cannot see through: the loads become per-lane (vector) loads and the values stay in
            // VGPRs; as provably uniform values they would be allocated to SGPRs and spilled (79 spills measured)
            int z;
            asm volatile("v_mov_b32 %0, 0" : "=v"(z));
            const double* t = coef_generic + (kMaxDegree + 1) * W + z;
#pragma unroll
            for (int k = 0; k <= DEG; ++k)
#pragma unroll
                for (int j = 0; j < RH; ++j) cf[k * RH + j] = t[k * RHP + j];
        }
    }
};

template <int W, int DEG>
__device__ __forceinline__ void horner2(const WindowCoef<W, DEG>& wc, const double* __restrict__ coef_generic, int degree, double s0,
                                        double s1, double (&v0)[W], double (&v1)[W]) {
    // even / odd split (see window_eval): p_j(+-s) = E_j(s^2) +- s O_j(s^2), degree + 1 fused multiply-adds per mirror pair
    constexpr int RH = (W + 1) / 2, RHP = sym_row(W);
    double e0[RH], o0[RH], e1[RH], o1[RH];
    const double t0 = s0 * s0, t1 = s1 * s1;
    if (DEG > 0) {
        constexpr int KE = DEG & ~1, KO = (DEG - 1) | 1;
#pragma unroll
        for (int j = 0; j < RH; ++j) {
            e0[j] = e1[j] = wc.cf[KE * RH + j];
            o0[j] = o1[j] = wc.cf[KO * RH + j];
        }
#pragma unroll
        for (int k = KE - 2; k >= 0; k -= 2) {
#pragma unroll
            for (int j = 0; j < RH; ++j) {
                const double c = wc.cf[k * RH + j];
                e0[j] = fma(e0[j], t0, c);
                e1[j] = fma(e1[j], t1, c);
            }
        }
#pragma unroll
        for (int k = KO - 2; k >= 1; k -= 2) {
#pragma unroll
            for (int j = 0; j < RH; ++j) {
                const double c = wc.cf[k * RH + j];
                o0[j] = fma(o0[j], t0, c);
                o1[j] = fma(o1[j], t1, c);
            }
        }
    } else {
        const_coef_ptr coef = (const_coef_ptr)(coef_generic + (kMaxDegree + 1) * W);
        const int ke = degree & ~1, ko = (degree - 1) | 1;
#pragma unroll
        for (int j = 0; j < RH; ++j) {
            e0[j] = e1[j] = coef[ke * RHP + j];
            o0[j] = o1[j] = coef[ko * RHP + j];
        }
        for (int k = ke - 2; k >= 0; k -= 2) {
#pragma unroll
            for (int j = 0; j < RH; ++j) {
                const double c = coef[k * RHP + j];
                e0[j] = fma(e0[j], t0, c);
                e1[j] = fma(e1[j], t1, c);
            }
        }
        for (int k = ko - 2; k >= 1; k -= 2) {
#pragma unroll
            for (int j = 0; j < RH; ++j) {
                const double c = coef[k * RHP + j];
                o0[j] = fma(o0[j], t0, c);
                o1[j] = fma(o1[j], t1, c);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < RH; ++j) {
        v0[j] = fma(s0, o0[j], e0[j]);
        v1[j] = fma(s1, o1[j], e1[j]);
        if (W - 1 - j != j) {
            v0[W - 1 - j] = fma(-s0, o0[j], e0[j]);
            v1[W - 1 - j] = fma(-s1, o1[j], e1[j]);
        }
    }
}

__device__ __forceinline__ int pos_mod(int a, int n) {
    int r = a % n;
    return r < 0 ? r + n : r;
}

// `live`: this lane's tile column exists (c16 < W + HB; the columns beyond accumulate whatever the clamped operand reads return)
template <int W>
__device__ __forceinline__ void flush_tile(const d4& acc, int cur_bx, int q, int ycell, bool live, int nf0, int nf1, int64_t cells,
                                           int channels, double S0, double S1, unsigned long long* __restrict__ gch) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = q + 4 * r, ch = row >> 3, i = row & 7;       // ch is a compile-time function of r
        const double v = acc[r];
        if (live && i < W && ch < channels && v != 0.0) {
            const int xc = pos_mod(cur_bx + i, nf0);
            const long long f = __double2ll_rn(v * (ch ? S1 : S0));
            __hip_atomic_fetch_add(gch + (int64_t)ch * cells + (int64_t)xc * nf1 + ycell, (unsigned long long)f, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Tile flush of the 4 x 4 x 4 layout (tiles of at most 8 columns): lane holds D[row = 4 blk + (lane >> 4)][col = (lane & 3) + 4 half]
// of accumulator `half`, blk = (lane >> 2) & 3.
template <int W>
__device__ __forceinline__ void flush_tile_small(double acc0, double acc1, int cur_bx, int lane, int by0, int ncols, int nf0, int nf1,
                                                 int64_t cells, int channels, double S0, double S1, unsigned long long* __restrict__ gch) {
    const int row = 4 * ((lane >> 2) & 3) + (lane >> 4), ch = row >> 3, i = row & 7;
    if (i < W && ch < channels) {
        const int xc = pos_mod(cur_bx + i, nf0);
        const double S = ch ? S1 : S0;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const double v = half ? acc1 : acc0;
            const int col = (lane & 3) + 4 * half;
            if (col < ncols && v != 0.0) {
                const long long f = __double2ll_rn(v * S);
                __hip_atomic_fetch_add(gch + (int64_t)ch * cells + (int64_t)xc * nf1 + pos_mod(by0 + col, nf1), (unsigned long long)f,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

struct PointIn {
    double2 xy;
    double c0, c1;
};

// SORTED: strengths come from the level-ordered copy (the fit-time (y, 1) pair or y alone): two coalesced loads and
// no strength-mode dispatch in the hot loop.
template <bool SORTED>
__device__ __forceinline__ PointIn load_point(const MfmaSpreadArgs& a, int batch, int p) {
    PointIn r;
    r.xy = reinterpret_cast<const double2*>(a.xs)[p];
    if (SORTED) {
        r.c0 = a.ys[p];
        r.c1 = 1.0;
    } else {
        fetch_strength(a.src, batch, a.perm[p], r.c0, r.c1);
    }
    return r;
}

#ifdef EFGP_MFMA_DIAG
#define EFGP_DIAG(bit_) (a.diag & (bit_))
#else
#define EFGP_DIAG(bit_) false
#endif

template <int W, int DEG, bool SORTED, int HB>
__global__ __launch_bounds__(64 * kMfmaMaxWaves, HB == 1 ? 3 : 2) void spread_mfma_kernel(MfmaSpreadArgs a) {
    extern __shared__ double lds_raw[];
    constexpr int BR = W + HB;                    // B rows = tile columns in use
    constexpr int kWaveDoubles = mfma_rows(W, HB) * kMfmaRow;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    double* A = lds_raw + (size_t)wave * kWaveDoubles;
    double* B = A + 16 * kMfmaRow;
    // x cell of point l: an int in the padding of row l >> 2 (two doubles = four ints behind the 64 points of every row)
    int* bxs = reinterpret_cast<int*>(A + 64);
    constexpr int kBxRow = 2 * kMfmaRow;          // ints per row
    WindowCoef<W, DEG> wc;
    wc.load(a.coef);
    const int batch = blockIdx.y;
    const int nf0 = a.nf0, nf1 = a.nf1;
    const int64_t cells = (int64_t)nf0 * nf1;
    const double S0 = a.scale[0], S1 = a.scale[2];
    unsigned long long* gch = a.gacc + (int64_t)batch * a.channels * cells;
    // Tiles of at most 8 columns (one-cell bands, W <= 7) go through v_mfma_f64_4x4x4: four independent 4 x 4 x 4 blocks per
    // instruction, 9.3 ns against the 29.4 ns of the 16 x 16 x 4 (tools/r3/mfma_f64_4x4_bench.hip) -- two instructions cover the
    // 16 x 8 tile of four points, 18.6 ns instead of 29.4 with half the columns idle.  Lane map (measured): A and B elements of
    // block blk = (lane >> 2) & 3, point k = lane >> 4, row / column lane & 3; D[blk][i = lane >> 4][j = lane & 3].  Block blk
    // takes the tile rows 4 blk .. 4 blk + 3 (so the A element of a lane is row lane & 15 of point k: the same LDS read as for the
    // large shape) and the columns 0..3 (first instruction) / 4..7 (second).
    constexpr bool SMALL = BR <= 8;
    const int q = lane >> 4, c16 = lane & 15;
    const bool live = c16 < BR;
    if (W < 8) {                                  // A rows of the unused stencil cells stay zero
#pragma unroll
        for (int r = W; r < 8; ++r) {
            A[r * kMfmaRow + lane] = 0.0;
            A[(8 + r) * kMfmaRow + lane] = 0.0;
        }
    }
    const double* Ard = A + c16 * kMfmaRow + q;   // this lane's operand elements of step g: Ard[4 g], Brd[4 g]
    const double* Brd = SMALL ? B + ((lane & 3) < BR ? (lane & 3) : 0) * kMfmaRow + q      // columns lane & 3 and 4 + (lane & 3)
                              : B + (live ? c16 : 0) * kMfmaRow + q;  // columns beyond BR: a valid row, the products are never flushed
    const double* Brd2 = B + (4 + (lane & 3) < BR ? 4 + (lane & 3) : 0) * kMfmaRow + q;
    for (int chunk = blockIdx.x * nwaves + wave; chunk < a.nchunks; chunk += a.total_waves) {
        const int4 ci = reinterpret_cast<const int4*>(a.chunks)[chunk];
        const int start = __builtin_amdgcn_readfirstlane(ci.x), count = __builtin_amdgcn_readfirstlane(ci.y);
        const int band = __builtin_amdgcn_readfirstlane(ci.z);
        const int by0 = (int)ceil(a.scale1 * (a.band_lo[band] - a.xcen1) - 0.5 * W);
        const int ycell = pos_mod(by0 + c16, nf1);
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        double sa0 = 0.0, sa1 = 0.0;                                          // SMALL: the two 4 x 4 x 4 accumulators
        int cur_bx = 0;
        PointIn nxt = load_point<SORTED>(a, batch, start + (lane < count ? lane : count - 1));
        for (int b0 = 0; b0 < count; b0 += 64) {
            const int rem = count - b0;                                       // wave-uniform
            const bool valid = lane < rem;
            const PointIn cur = nxt;
            if (rem > 64) {                                                   // next batch's loads fly during this one
                const int r2 = rem - 64;
                nxt = load_point<SORTED>(a, batch, start + b0 + 64 + (lane < r2 ? lane : r2 - 1));
            }
            double c0 = cur.c0, c1 = a.channels == 1 ? 0.0 : cur.c1;
            if (!valid) {                                                     // tail lanes repeat the last point with zero strength
                c0 = 0.0;
                c1 = 0.0;
            }
            const double X0 = a.scale0 * (cur.xy.x - a.xcen0), X1 = a.scale1 * (cur.xy.y - a.xcen1);
            const double i0 = ceil(X0 - 0.5 * W), j0 = ceil(X1 - 0.5 * W);
            double v0[W], v1[W];
            if (EFGP_DIAG(2)) {
#pragma unroll
                for (int i = 0; i < W; ++i) {
                    v0[i] = X0 + i;
                    v1[i] = X1 - i;
                }
            } else {
                horner2<W, DEG>(wc, a.coef, a.degree, 2.0 * (i0 - X0 + 0.5 * W) - 1.0, 2.0 * (j0 - X1 + 0.5 * W) - 1.0, v0, v1);
            }
            const int bx = (int)i0, off = (int)j0 - by0;                      // 0 <= off <= HB (bands are at most HB cells high)
            if (!EFGP_DIAG(4)) {
#pragma unroll
                for (int i = 0; i < W; ++i) {
                    A[i * kMfmaRow + lane] = c0 * v0[i];
                    A[(8 + i) * kMfmaRow + lane] = c1 * v0[i];
                    B[(off + i) * kMfmaRow + lane] = v1[i];
                }
#pragma unroll
                for (int i = 0; i < HB; ++i)                                  // the HB rows outside [off, off + W) are zero
                    B[((i < off) ? i : i + W) * kMfmaRow + lane] = 0.0;
            } else {
                acc[0] += v0[0] * c0 + v1[W - 1] * c1 + v0[W / 2] + v1[W / 2];
            }
            bxs[(lane >> 2) * kBxRow + (lane & 3)] = bx;
            if (b0 == 0) cur_bx = __builtin_amdgcn_readfirstlane(bx);
            int prev = __shfl_up(bx, 1, 64);
            if (lane == 0) prev = cur_bx;
            const unsigned long long mask = __ballot(bx != prev);              // run starts inside this batch
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (EFGP_DIAG(1)) {
                acc[1] += A[lane] + B[lane];
            } else if (SMALL) {
                if (mask == 0ull && rem >= 64) {
#pragma unroll
                    for (int g = 0; g < 16; ++g) {
                        const double av = Ard[4 * g], b0v = Brd[4 * g], b1v = Brd2[4 * g];
                        sa0 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, b0v, sa0, 0, 0, 0);
                        sa1 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, b1v, sa1, 0, 0, 0);
                    }
                } else {
                    for (int g = 0; g < 16; ++g) {
                        if (4 * g >= rem) break;
                        const double av = Ard[4 * g], b0v = Brd[4 * g], b1v = Brd2[4 * g];
                        const unsigned bits = (unsigned)(mask >> (4 * g)) & 15u;   // wave-uniform
                        if (bits == 0u) {
                            sa0 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, b0v, sa0, 0, 0, 0);
                            sa1 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, b1v, sa1, 0, 0, 0);
                        } else {
                            for (int k = 0; k < 4; ++k) {                      // a run ends inside these four points
                                if ((bits >> k) & 1u) {
                                    flush_tile_small<W>(sa0, sa1, cur_bx, lane, by0, BR, nf0, nf1, cells, a.channels, S0, S1, gch);
                                    sa0 = sa1 = 0.0;
                                    cur_bx = __builtin_amdgcn_readfirstlane(bxs[g * kBxRow + k]);
                                }
                                const double am = q == k ? av : 0.0;
                                sa0 = __builtin_amdgcn_mfma_f64_4x4x4f64(am, b0v, sa0, 0, 0, 0);
                                sa1 = __builtin_amdgcn_mfma_f64_4x4x4f64(am, b1v, sa1, 0, 0, 0);
                            }
                        }
                    }
                }
            } else if (mask == 0ull && rem >= 64) {
                // whole batch continues the current run: 16 operand pairs, 16 back-to-back MFMAs, no branches
                // (round 3, measured: a sched_barrier between the reads and the MFMAs -- all 16 operand reads in flight, counted
                // lgkmcnt waits -- changes nothing, 212 us either way at N = 1e7: the other waves of the SIMD hide the round
                // trips, and the 64 operand VGPRs would stand in the way of a third wave)
                double av[16], bv[16];
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    av[g] = Ard[4 * g];
                    bv[g] = Brd[4 * g];
                }
#pragma unroll
                for (int g = 0; g < 16; ++g) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[g], bv[g], acc, 0, 0, 0);
            } else {
                for (int g = 0; g < 16; ++g) {
                    if (4 * g >= rem) break;
                    const double av = Ard[4 * g], bv = Brd[4 * g];
                    const unsigned bits = (unsigned)(mask >> (4 * g)) & 15u;   // wave-uniform
                    if (bits == 0u) {
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                    } else {
                        for (int k = 0; k < 4; ++k) {                          // a run ends inside these four points
                            if ((bits >> k) & 1u) {
                                flush_tile<W>(acc, cur_bx, q, ycell, live, nf0, nf1, cells, a.channels, S0, S1, gch);
                                acc = d4{0.0, 0.0, 0.0, 0.0};
                                cur_bx = __builtin_amdgcn_readfirstlane(bxs[g * kBxRow + k]);
                            }
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(q == k ? av : 0.0, bv, acc, 0, 0, 0);
                        }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();                                   // the next batch overwrites the rows
        }
        if (SMALL) flush_tile_small<W>(sa0, sa1, cur_bx, lane, by0, BR, nf0, nf1, cells, a.channels, S0, S1, gch);
        else flush_tile<W>(acc, cur_bx, q, ycell, live, nf0, nf1, cells, a.channels, S0, S1, gch);
    }
}

template <int W, int HB>
static hipError_t launch_w(dim3 grid, int waves, hipStream_t s, const MfmaSpreadArgs& a) {
    // W + 1 is what es_make_params picks for every standard tolerance at upsampling ratios around 2, W + 2 on the finer grids of
    // dense point sets (W = 7 at ratio 4, tolerance 6e-8: degree 9); other degrees take the run-time Horner loop (DEG = 0)
    const int fixed = a.degree == W + 1 ? 1 : (a.degree == W + 2 ? 2 : 0);
    const bool sorted = a.ys != nullptr;
    const size_t lds = mfma_lds_bytes(waves, W, HB);
    const dim3 block(64 * waves);
#define EFGP_GO(kern_)                                                                                               \
    do {                                                                                                             \
        auto k = kern_;                                                                                              \
        if (lds > 65536) {                                                                                           \
            hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return e;                                                                           \
        }                                                                                                            \
        hipLaunchKernelGGL(k, grid, block, lds, s, a);                                                               \
    } while (0)
    if (fixed == 1 && sorted) EFGP_GO((spread_mfma_kernel<W, W + 1, true, HB>));
    else if (fixed == 1) EFGP_GO((spread_mfma_kernel<W, W + 1, false, HB>));
    else if (fixed == 2 && sorted) EFGP_GO((spread_mfma_kernel<W, W + 2, true, HB>));
    else if (fixed == 2) EFGP_GO((spread_mfma_kernel<W, W + 2, false, HB>));
    else if (sorted) EFGP_GO((spread_mfma_kernel<W, 0, true, HB>));
    else EFGP_GO((spread_mfma_kernel<W, 0, false, HB>));
#undef EFGP_GO
    return hipGetLastError();
}

int spread_mfma_launch(DeviceCtx* ctx, const SortedLevel* lvl, int band_cells, const double* ys_sorted, const StrengthSrc& src,
                       const GridGeom& g, int W, const double* coef, int degree, int channels, int nbatch, unsigned long long* gacc,
                       const double* scale, hipStream_t stream) {
    EFGP_REQUIRE(band_cells == 1 || band_cells == 8, "spread_mfma: band height bound must be 1 or 8 cells");
    EFGP_REQUIRE(W >= 2 && W <= kMfmaMaxW, "spread_mfma: window width %d outside 2..%d", W, kMfmaMaxW);
    EFGP_REQUIRE(channels == 1 || channels == 2, "spread_mfma: channels must be 1 or 2");
    EFGP_REQUIRE(lvl && lvl->nchunks > 0, "spread_mfma: empty level");
    MfmaSpreadArgs a;
    a.xs = lvl->xs;
    a.ys = ys_sorted;
    a.perm = lvl->perm;
    a.chunks = lvl->chunks;
    a.nchunks = lvl->nchunks;
    a.band_lo = lvl->d_band_lo;
    a.src = src;
    a.scale0 = g.scale[0];
    a.scale1 = g.scale[1];
    a.xcen0 = g.xcen[0];
    a.xcen1 = g.xcen[1];
    a.nf0 = (int)g.nf[0];
    a.nf1 = (int)g.nf[1];
    a.coef = coef;
    a.degree = degree;
    a.channels = channels;
    a.gacc = gacc;
    a.scale = scale;
    // Two workgroups of four waves per CU (176 VGPRs and 16.9 KB of LDS rows per wave: two waves per SIMD); a wave strides
    // over the chunks (points_layout.hip sizes them for 8 waves per CU and several rounds).  Measured at N = 1e7:
    // 4 x 2 221 us, 3 x 3 (LDS table) 270 us, 1 x 9 265 us.
    // (Tried and measured at N = 1e7, all slower than 4 x 2 = 221 us: a Horner table in LDS with 3 x 3 waves 270 us / 4 x 4
    // waves and compact rows 231-241 us; a software-pipelined one-wave-per-SIMD variant with double-buffered rows that
    // interleaves the polynomials of batch n + 1 with the MFMAs of batch n 288-296 us -- the compiler's conservative
    // s_waitcnt vmcnt(0) at the merge of its two paths exposes the HBM latency of the prefetched points.)
    // Round 3, bands at most ONE cell high (band_cells = 1; pick_level chooses it for dense point sets): 13.2 KB of rows per
    // wave and 168 VGPRs -> three workgroups of four waves per CU.
    int waves = 4, per_cu = band_cells == 1 ? 3 : 2;
    if (const char* e1 = std::getenv("EFGP_MFMA_WAVES")) waves = std::max(1, std::min(kMfmaMaxWaves, std::atoi(e1)));
    if (const char* e2 = std::getenv("EFGP_MFMA_BLOCKS_PER_CU")) per_cu = std::max(1, std::atoi(e2));
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(((int64_t)lvl->nchunks + waves - 1) / waves, (int64_t)ctx->num_cu * per_cu));
    a.total_waves = blocks * waves;
    a.diag = std::getenv("EFGP_MFMA_DIAG") ? std::atoi(std::getenv("EFGP_MFMA_DIAG")) : 0;
    hipError_t e;
    // (Several channel pairs in ONE pass -- windows evaluated once, one accumulator tile per pair, A = c wx formed per step --
    // was measured for the probe transforms of the gradient: 0.876 ms instead of 0.779 ms for T = 5 at N = 1e7; the pass is
    // bound by its MFMA phase, which that variant lengthens.  Pairs stay separate launches of grid.y.)
    {
        KernelTimer timer("spread", stream);
        switch (W) {
#define EFGP_CASE(w_)                                                                                  \
    case w_:                                                                                           \
        e = band_cells == 1 ? launch_w<w_, 1>(dim3(blocks, nbatch), waves, stream, a)                  \
                            : launch_w<w_, 8>(dim3(blocks, nbatch), waves, stream, a);                 \
        break;
            EFGP_CASE(2) EFGP_CASE(3) EFGP_CASE(4) EFGP_CASE(5) EFGP_CASE(6) EFGP_CASE(7) EFGP_CASE(8)
#undef EFGP_CASE
            default: e = hipErrorInvalidValue;
        }
    }
    if (e != hipSuccess) {
        set_error("MFMA spread kernel launch failed: %s", hipGetErrorString(e));
        return EFGP_EHIP;
    }
    return EFGP_OK;
}

}  // namespace efgp
