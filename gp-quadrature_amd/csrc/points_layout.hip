// Per-model point layouts (see points_layout.hpp) and their C ABI.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "points_layout.hpp"

#include <rocprim/device/device_radix_sort.hpp>

namespace efgp {

// order-preserving map double -> uint64 (and back)
__host__ __device__ __forceinline__ unsigned long long enc_f64(double v) {
    unsigned long long b;
#if defined(__HIP_DEVICE_COMPILE__)
    b = (unsigned long long)__double_as_longlong(v);
#else
    std::memcpy(&b, &v, 8);
#endif
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
static double dec_f64(unsigned long long e) {
    unsigned long long b = (e >> 63) ? (e & 0x7FFFFFFFFFFFFFFFull) : ~e;
    double v;
    std::memcpy(&v, &b, 8);
    return v;
}

// per-dimension min / max of the coordinates: stats[2a] = enc(min), stats[2a+1] = enc(max)
template <int D>
__global__ __launch_bounds__(256) void bbox_kernel(const double* __restrict__ x, int64_t npts, unsigned long long* __restrict__ stats) {
    unsigned long long mn[D], mx[D];
#pragma unroll
    for (int a = 0; a < D; ++a) {
        mn[a] = ~0ull;
        mx[a] = 0ull;
    }
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < npts; n += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int a = 0; a < D; ++a) {
            const unsigned long long e = enc_f64(x[n * D + a]);
            mn[a] = e < mn[a] ? e : mn[a];
            mx[a] = e > mx[a] ? e : mx[a];
        }
    }
#pragma unroll
    for (int a = 0; a < D; ++a) {
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o1 = __shfl_down(mn[a], off, 64), o2 = __shfl_down(mx[a], off, 64);
            mn[a] = o1 < mn[a] ? o1 : mn[a];
            mx[a] = o2 > mx[a] ? o2 : mx[a];
        }
    }
    // one pair of global atomics per WORKGROUP and dimension (round 3: one per wave of 2048 workgroups was 32 k serialised
    // 64-bit atomics on four addresses -- 377 us of the 0.5-ms pass at N = 1e6)
    __shared__ unsigned long long smn[4][D], smx[4][D];
    const int wid = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < D; ++a) {
            smn[wid][a] = mn[a];
            smx[wid][a] = mx[a];
        }
    }
    __syncthreads();
    if (threadIdx.x < D) {
        const int a = threadIdx.x;
        unsigned long long lo = smn[0][a], hi = smx[0][a];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) {
            lo = smn[w][a] < lo ? smn[w][a] : lo;
            hi = smx[w][a] > hi ? smx[w][a] : hi;
        }
        atomicMin(&stats[2 * a], lo);
        atomicMax(&stats[2 * a + 1], hi);
    }
}

// sort key (band << 56 | top 56 bits of the ordered x_0) and per-band count / min / max of x_1.  The band statistics are
// reduced per workgroup in LDS first: one global atomic per (workgroup, band) -- a global atomic per POINT on 3 x nbands
// addresses serialised the whole pass (65 ms at N = 1e7, measured with rocprofv3).
__global__ __launch_bounds__(256) void band_key_kernel(const double* __restrict__ x, int64_t npts, double lo1, double inv_h1, int nbands,
                                                       unsigned long long* __restrict__ keys, int* __restrict__ vals,
                                                       unsigned long long* __restrict__ bstat /* [nbands][3] count, enc min, enc max */) {
    __shared__ unsigned long long cnt[kMaxBands], mn[kMaxBands], mx[kMaxBands];
    for (int b = threadIdx.x; b < nbands; b += blockDim.x) {
        cnt[b] = 0ull;
        mn[b] = ~0ull;
        mx[b] = 0ull;
    }
    __syncthreads();
    // wave-uniform trip count: every lane stays in the loop (the wave-level reduction below shuffles across all 64 lanes)
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < npts; base += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = base + threadIdx.x;
        const bool valid = n < npts;
        int b = -1;
        unsigned long long e = 0ull;
        if (valid) {
            const double2 p = reinterpret_cast<const double2*>(x)[n];
            b = (int)floor((p.y - lo1) * inv_h1);
            b = b < 0 ? 0 : (b >= nbands ? nbands - 1 : b);
            keys[n] = ((unsigned long long)b << 56) | (enc_f64(p.x) >> 8);
            vals[n] = (int)n;
            e = enc_f64(p.y);
        }
        if (nbands >= 64) {                      // many bands: the lanes of a wave rarely meet on an address
            if (valid) {
                atomicAdd(&cnt[b], 1ull);
                atomicMin(&mn[b], e);
                atomicMax(&mx[b], e);
            }
        } else {
            // few bands (8-16 at N = 1e6): 256 threads on 3 x nbands LDS addresses serialise -- 231 us of this 1e6-point pass
            // (rocprofv3, round 4).  One lane per DISTINCT band of the wave adds the wave's count / min / max for it.
            unsigned long long todo = __ballot(valid);
            while (todo) {
                const int leader = __ffsll((long long)todo) - 1;
                const int bb = __shfl(b, leader, 64);
                const bool mine = b == bb;
                const unsigned long long m = __ballot(mine);
                unsigned long long lo_ = mine ? e : ~0ull, hi_ = mine ? e : 0ull;
                for (int off = 32; off > 0; off >>= 1) {
                    const unsigned long long l2 = __shfl_xor(lo_, off, 64), h2 = __shfl_xor(hi_, off, 64);
                    lo_ = l2 < lo_ ? l2 : lo_;
                    hi_ = h2 > hi_ ? h2 : hi_;
                }
                if ((int)(threadIdx.x & 63) == leader) {
                    atomicAdd(&cnt[bb], (unsigned long long)__popcll(m));
                    atomicMin(&mn[bb], lo_);
                    atomicMax(&mx[bb], hi_);
                }
                todo &= ~m;
            }
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nbands; b += blockDim.x) {
        if (cnt[b]) {
            atomicAdd(&bstat[3 * b], cnt[b]);
            atomicMin(&bstat[3 * b + 1], mn[b]);
            atomicMax(&bstat[3 * b + 2], mx[b]);
        }
    }
}

__global__ __launch_bounds__(256) void gather_points_kernel(const double* __restrict__ x, const int* __restrict__ perm, int64_t npts,
                                                            double* __restrict__ xs) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npts) return;
    reinterpret_cast<double2*>(xs)[p] = reinterpret_cast<const double2*>(x)[perm[p]];
}

__global__ __launch_bounds__(256) void gather_values_kernel(const double* __restrict__ y, const int* __restrict__ perm, int64_t npts,
                                                            double* __restrict__ ys) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npts) return;
    ys[p] = y[perm[p]];
}

// max|y| of the attached strengths as an ordered bit pattern (non-negative doubles compare like integers)
__global__ __launch_bounds__(256) void values_max_kernel(const double* __restrict__ y, int64_t n, unsigned long long* __restrict__ out) {
    double m = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        m = fmax(m, isfinite(y[i]) ? fabs(y[i]) : INFINITY);            // non-finite targets poison the fit (see fixed_scale_write)
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_down(m, off, 64));
    // one atomic per WORKGROUP: 8192 wave atomics on one address were 80 of this pass's 95 us at N = 1e6
    __shared__ double wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmax(fmax(wmax[0], wmax[1]), fmax(wmax[2], wmax[3]));
        if (m > 0.0) atomicMax(out, (unsigned long long)__double_as_longlong(m));
    }
}

static void free_level(SortedLevel* l) {
    if (!l) return;
    if (l->xs) (void)hipFree(l->xs);
    if (l->perm) (void)hipFree(l->perm);
    if (l->ys) (void)hipFree(l->ys);
    if (l->chunks) (void)hipFree(l->chunks);
    if (l->d_band_lo) (void)hipFree(l->d_band_lo);
    delete l;
}

// Chunk length: every resident wave gets R chunks of at most ~1536 points (R rounds keep the tail of the grid-stride loop
// short); multiples of 64 (one point per lane per batch).  The spreader runs 8 waves per CU (band levels up to 8 cells high) or
// 12 (one-cell bands, dense point sets): from 4e6 points on the chunk count is sized for 24 waves per CU, a multiple of both
// (N = 1e7: 12 k chunks of 832 points = 4.0 / 5.9 rounds).  Smaller sets keep the 8-wave sizing: every chunk end is a tile
// flush (256 atomics), which must stay rare next to the points.
static int chunk_length(int64_t npts, int num_cu) {
    const int64_t waves = (int64_t)num_cu * (npts >= 4000000 ? 24 : 8);
    const int64_t rounds = std::max<int64_t>(1, (npts + waves * 1536 - 1) / (waves * 1536));
    int64_t len = (npts + waves * rounds - 1) / (waves * rounds);
    len = (len + 63) / 64 * 64;
    return (int)std::max<int64_t>(64, len);
}

int points_level(efgp_points_s* pts, int nbands, hipStream_t stream, SortedLevel** out) {
    for (SortedLevel* l : pts->levels)
        if (l->nbands == nbands) {
            *out = l;
            return EFGP_OK;
        }
    EFGP_REQUIRE(pts->dim == 2, "sorted point layouts exist for 2-D points only");
    EFGP_REQUIRE(nbands >= 1 && nbands <= kMaxBands && (nbands & (nbands - 1)) == 0, "band count must be a power of two <= %d", kMaxBands);
    EFGP_REQUIRE(pts->npts > 0 && pts->npts < (int64_t)1 << 31, "sorted point layouts need 1 <= N < 2^31");
    const int64_t N = pts->npts;
    auto* l = new SortedLevel();
    l->nbands = nbands;
    l->npts = N;
    unsigned long long *keys = nullptr, *keys2 = nullptr, *bstat = nullptr;
    int *vals = nullptr;
    void* temp = nullptr;
    auto fail = [&](int code) {
        if (keys) (void)hipFree(keys);
        if (keys2) (void)hipFree(keys2);
        if (vals) (void)hipFree(vals);
        if (bstat) (void)hipFree(bstat);
        if (temp) (void)hipFree(temp);
        free_level(l);
        return code;
    };
#define LVL_CHECK(expr)                                                                              \
    do {                                                                                             \
        hipError_t e__ = (expr);                                                                     \
        if (e__ != hipSuccess) {                                                                     \
            set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__);   \
            return fail(EFGP_EHIP);                                                                  \
        }                                                                                            \
    } while (0)
    l->xs_bytes = (size_t)N * 2 * sizeof(double);
    l->perm_bytes = (size_t)N * sizeof(int);
    LVL_CHECK(hipMalloc((void**)&l->xs, l->xs_bytes));
    LVL_CHECK(hipMalloc((void**)&l->perm, l->perm_bytes));
    LVL_CHECK(hipMalloc((void**)&keys, (size_t)N * 8));
    LVL_CHECK(hipMalloc((void**)&keys2, (size_t)N * 8));
    LVL_CHECK(hipMalloc((void**)&vals, (size_t)N * 4));
    LVL_CHECK(hipMalloc((void**)&bstat, (size_t)nbands * 3 * 8));
    std::vector<unsigned long long> init((size_t)nbands * 3);
    for (int b = 0; b < nbands; ++b) {
        init[3 * b] = 0;
        init[3 * b + 1] = ~0ull;
        init[3 * b + 2] = 0;
    }
    LVL_CHECK(hipMemcpyAsync(bstat, init.data(), init.size() * 8, hipMemcpyHostToDevice, stream));
    const double span = pts->hi[1] - pts->lo[1];
    const double inv_h1 = span > 0.0 ? (double)nbands / span : 0.0;
    const int blocks = (int)((N + 255) / 256);
    const int key_blocks = (int)std::max<int64_t>(1, std::min<int64_t>(blocks, (int64_t)pts->ctx->num_cu * 16));
    hipLaunchKernelGGL(band_key_kernel, dim3(key_blocks), dim3(256), 0, stream, pts->x, N, pts->lo[1], inv_h1, nbands, keys, vals, bstat);
    LVL_CHECK(hipGetLastError());
    size_t temp_bytes = 0;
    LVL_CHECK(rocprim::radix_sort_pairs(nullptr, temp_bytes, keys, keys2, vals, l->perm, (size_t)N, 0, 64, stream));
    LVL_CHECK(hipMalloc(&temp, std::max<size_t>(temp_bytes, 256)));
    LVL_CHECK(rocprim::radix_sort_pairs(temp, temp_bytes, keys, keys2, vals, l->perm, (size_t)N, 0, 64, stream));
    hipLaunchKernelGGL(gather_points_kernel, dim3(blocks), dim3(256), 0, stream, pts->x, (const int*)l->perm, N, l->xs);
    LVL_CHECK(hipGetLastError());
    std::vector<unsigned long long> st((size_t)nbands * 3);
    LVL_CHECK(hipMemcpyAsync(st.data(), bstat, st.size() * 8, hipMemcpyDeviceToHost, stream));
    LVL_CHECK(hipStreamSynchronize(stream));
    l->band_lo.resize(nbands);
    l->band_hi.resize(nbands);
    l->band_start.assign(nbands + 1, 0);
    for (int b = 0; b < nbands; ++b) {
        const int64_t cnt = (int64_t)st[3 * b];
        l->band_start[b + 1] = l->band_start[b] + cnt;
        l->band_lo[b] = cnt ? dec_f64(st[3 * b + 1]) : 1.0;
        l->band_hi[b] = cnt ? dec_f64(st[3 * b + 2]) : 0.0;
    }
    if (l->band_start[nbands] != N) {
        set_error("band histogram does not add up (%lld of %lld points)", (long long)l->band_start[nbands], (long long)N);
        return fail(EFGP_EHIP);
    }
    l->chunk_len = chunk_length(N, pts->ctx->num_cu);
    std::vector<int> chunks;
    for (int b = 0; b < nbands; ++b)
        for (int64_t s = l->band_start[b]; s < l->band_start[b + 1]; s += l->chunk_len) {
            chunks.push_back((int)s);
            chunks.push_back((int)std::min<int64_t>(l->chunk_len, l->band_start[b + 1] - s));
            chunks.push_back(b);
            chunks.push_back(0);
        }
    l->nchunks = (int)(chunks.size() / 4);
    l->chunk_bytes = chunks.size() * sizeof(int);
    l->lo_bytes = (size_t)nbands * sizeof(double);
    LVL_CHECK(hipMalloc((void**)&l->chunks, std::max<size_t>(l->chunk_bytes, 16)));
    LVL_CHECK(hipMalloc((void**)&l->d_band_lo, l->lo_bytes));
    LVL_CHECK(hipMemcpyAsync(l->chunks, chunks.data(), l->chunk_bytes, hipMemcpyHostToDevice, stream));
    LVL_CHECK(hipMemcpyAsync(l->d_band_lo, l->band_lo.data(), l->lo_bytes, hipMemcpyHostToDevice, stream));
    LVL_CHECK(hipStreamSynchronize(stream));       // the host vectors go out of scope
#undef LVL_CHECK
    (void)hipFree(keys);
    (void)hipFree(keys2);
    (void)hipFree(vals);
    (void)hipFree(bstat);
    (void)hipFree(temp);
    pts->levels.push_back(l);
    *out = l;
    return EFGP_OK;
}

int points_level_values(efgp_points_s* pts, SortedLevel* lvl, hipStream_t stream) {
    if (!pts->values) {
        lvl->ys_src = nullptr;
        return EFGP_OK;
    }
    if (lvl->ys && lvl->ys_src == pts->values) return EFGP_OK;
    if (!lvl->ys) {
        lvl->ys_bytes = (size_t)lvl->npts * sizeof(double);
        EFGP_HIP_CHECK(hipMalloc((void**)&lvl->ys, lvl->ys_bytes));
    }
    const int blocks = (int)((lvl->npts + 255) / 256);
    hipLaunchKernelGGL(gather_values_kernel, dim3(blocks), dim3(256), 0, stream, pts->values, (const int*)lvl->perm, lvl->npts, lvl->ys);
    EFGP_HIP_CHECK(hipGetLastError());
    lvl->ys_src = pts->values;
    return EFGP_OK;
}

}  // namespace efgp

using namespace efgp;

extern "C" {

int efgp_points_create(efgp_points_t** out, int device, int dim, int64_t npts, const double* x, void* stream_) {
    EFGP_REQUIRE(out, "efgp_points_create: null out");
    EFGP_REQUIRE(dim >= 1 && dim <= 3, "efgp_points_create: dim must be 1, 2 or 3 (got %d)", dim);
    EFGP_REQUIRE(npts >= 0, "efgp_points_create: negative point count");
    EFGP_REQUIRE(npts == 0 || x, "efgp_points_create: null x");
    DeviceCtx* ctx = device_ctx(device);
    if (!ctx) return EFGP_EHIP;
    DeviceGuard guard(device, (hipStream_t)stream_);
    hipStream_t stream = (hipStream_t)stream_;
    auto* p = new efgp_points_s();
    p->device = device;
    p->dim = dim;
    p->npts = npts;
    p->x = x;
    p->ctx = ctx;
    if (npts > 0) {
        unsigned long long* stats = (unsigned long long*)scratch(ctx, SLOT_MISC, 64);
        if (!stats) {
            delete p;
            return EFGP_ENOMEM;
        }
        unsigned long long init[6] = {~0ull, 0ull, ~0ull, 0ull, ~0ull, 0ull};
        hipError_t e = hipMemcpyAsync(stats, init, sizeof(init), hipMemcpyHostToDevice, stream);
        const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((npts + 255) / 256, 1024));
        if (e == hipSuccess) {
            if (dim == 1) hipLaunchKernelGGL(bbox_kernel<1>, dim3(blocks), dim3(256), 0, stream, x, npts, stats);
            else if (dim == 2) hipLaunchKernelGGL(bbox_kernel<2>, dim3(blocks), dim3(256), 0, stream, x, npts, stats);
            else hipLaunchKernelGGL(bbox_kernel<3>, dim3(blocks), dim3(256), 0, stream, x, npts, stats);
            e = hipGetLastError();
        }
        unsigned long long res[6];
        if (e == hipSuccess) e = hipMemcpyAsync(res, stats, sizeof(res), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) {
            set_error("efgp_points_create: bounding-box pass failed: %s", hipGetErrorString(e));
            delete p;
            return EFGP_EHIP;
        }
        for (int a = 0; a < dim; ++a) {
            p->lo[a] = dec_f64(res[2 * a]);
            p->hi[a] = dec_f64(res[2 * a + 1]);
        }
    }
    *out = p;
    return EFGP_OK;
}

int efgp_points_destroy(efgp_points_t* pts) {
    if (!pts) return EFGP_OK;
    DeviceGuard guard(pts->device);
    (void)hipDeviceSynchronize();
    for (SortedLevel* l : pts->levels) free_level(l);
    if (pts->d_values_max) (void)hipFree(pts->d_values_max);
    if (pts->d_fixed_scale) (void)hipFree(pts->d_fixed_scale);
    delete pts;
    return EFGP_OK;
}

int efgp_points_bounds(efgp_points_t* pts, double* lo_out, double* hi_out) {
    EFGP_REQUIRE(pts && lo_out && hi_out, "efgp_points_bounds: null argument");
    for (int a = 0; a < pts->dim; ++a) {
        lo_out[a] = pts->lo[a];
        hi_out[a] = pts->hi[a];
    }
    return EFGP_OK;
}

int efgp_points_attach_values(efgp_points_t* pts, const double* y, void* stream_) {
    EFGP_REQUIRE(pts, "efgp_points_attach_values: null layout");
    DeviceGuard guard(pts->device, (hipStream_t)stream_);
    hipStream_t stream = (hipStream_t)stream_;
    pts->values = y;
    pts->pair_scale_ready = false;
    for (SortedLevel* l : pts->levels) l->ys_src = nullptr;      // sorted copies are rebuilt on next use
    if (y && pts->npts > 0) {
        if (!pts->d_values_max) {
            EFGP_HIP_CHECK(hipMalloc((void**)&pts->d_values_max, 128));          // [0] max|y| bits, [64..128) the pair scale block
            pts->d_pair_scale = reinterpret_cast<double*>(reinterpret_cast<char*>(pts->d_values_max) + 64);
        }
        pts->pair_scale_ready = false;
        EFGP_HIP_CHECK(hipMemsetAsync(pts->d_values_max, 0, 8, stream));
        const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((pts->npts + 255) / 256, 1024));
        hipLaunchKernelGGL(values_max_kernel, dim3(blocks), dim3(256), 0, stream, y, pts->npts, pts->d_values_max);
        EFGP_HIP_CHECK(hipGetLastError());
    }
    return EFGP_OK;
}

}  // extern "C"
