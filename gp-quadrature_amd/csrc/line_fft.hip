// In-house batched complex FFT (double precision, in place, rank 1..3) for the sizes the EFGP path produces: fine grids of the
// NUFFT (2^a 3^b 5^c, es_fine_size) and the power-of-two circulant grids of the Toeplitz operator.
//
// Why (round 3): hipFFT / rocFFT compiles its kernels at RUN TIME for every new transform length (0.7-2.7 s per length on this
// box, profiles/r2_cold_start.txt): the first fit of a 1-D or 3-D model cost 2.7-3.6 s, and a hyper-parameter optimisation that
// walks through grid sizes pays it again for every new size -- hundreds of 10-ms steps' worth.  The 2-D path left rocFFT in
// round 2 (pruned DFT kernels); this is the same for everything else.  One kernel, no code generation: a pass transforms the
// lines along ONE axis; a workgroup stages a tile of L lines in LDS (tile chosen so that global accesses are coalesced along
// the contiguous direction, whatever the axis), runs Stockham autosort stages of radix 4 / 2 / 3 / 5 between two LDS buffers
// with twiddles from a per-length table exp(-2 pi i q / n) (host long double, cached per context), and writes the tile back.
// A rank-3 transform is three passes = three reads + writes of the array, what rocFFT's own three kernels move.
//
// Reference operations replaced: torch.fft.fftn / ifftn inside ToeplitzND (efgpnd.py:1275-1290, 1331-1393) and the FFT
// inside finufft's type-1 / type-2 transforms (efgpnd.py:1395-1421, 1533-1536).
#include "line_fft.hpp"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace efgp {

constexpr int kFftThreads = 256;
constexpr int kFftMaxLine = 4096;            // two LDS buffers of one line: 128 KB
constexpr int kFftMaxFactors = 12;

struct OwnFftPass {
    const double2* src;      // lines of n_src stored entries (may alias dst when n_src == n_dst == n)
    double2* dst;            // lines of n_dst stored entries
    const double2* tw;       // exp(-2 pi i q / n), q < n
    int n;                   // transform length
    int n_src, n_dst;        // stored entries per line: the n_x lowest-|frequency| bins in FFT order (bin k mod n_x); n_x == n: all
    int nfac;
    int fac[kFftMaxFactors];
    int64_t stride;          // between consecutive points of a line (1: the contiguous axis)
    int64_t nouter;          // batch * extents of the slower axes
    int L;                   // lines per workgroup
    int ld;                  // LDS line pitch
    int backward;            // conjugate on load and on store: the unnormalised inverse transform
    // windows on the slower axes (zero-padded inputs / cropped outputs of the Toeplitz products): only the lines whose slower
    // indices lie in [slow_lo, slow_lo + slow_cnt) are transformed; nouter counts those lines (times the batch)
    int nslow;
    int64_t slow_ext[2], slow_lo[2], slow_cnt[2];
    // contiguous first pass of a type-1 transform straight from the spreaders' int64 accumulator (FftAccSource), or null
    long long* acc;
    int acc_channels, acc_reset;
    int64_t acc_cells;
    const double* acc_scale;
};
__device__ __forceinline__ int64_t decode_outer(const OwnFftPass& a, int64_t oc) {
    if (a.nslow == 0) return oc;
    int64_t o = 0, mul = 1, rem = oc;
    for (int q = a.nslow - 1; q >= 0; --q) {
        const int64_t c = rem % a.slow_cnt[q];
        rem /= a.slow_cnt[q];
        o += (a.slow_lo[q] + c) * mul;
        mul *= a.slow_ext[q];
    }
    return o + rem * mul;       // rem = batch index
}
// bin p of the length-n transform -> index among the n_x stored entries (bins -n_x/2 .. (n_x-1)/2 in FFT order), or -1
__device__ __forceinline__ int stored_index(int p, int n, int n_x) {
    if (p < (n_x + 1) / 2) return p;
    return p >= n - n_x / 2 ? p - (n - n_x) : -1;
}

__device__ __forceinline__ double2 fmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ double2 fadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 fsub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double2 mul_mi(double2 a) { return make_double2(a.y, -a.x); }       // a * (-i)

// one Stockham butterfly of radix R: inputs x[j + r nb], twiddled by w^(r k), outputs y[j0 + r Ns]
template <int R>
__device__ __forceinline__ void butterfly(const double2* __restrict__ x, double2* __restrict__ y, const double2* __restrict__ tw, int j,
                                          int nb, int k, int tstep, int j0, int Ns) {
    double2 v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = x[j + r * nb];
    if (k != 0) {
#pragma unroll
        for (int r = 1; r < R; ++r) v[r] = fmul(v[r], tw[(int64_t)r * k * tstep]);
    }
    if (R == 2) {
        const double2 a = v[0], b = v[1];
        v[0] = fadd(a, b);
        v[1] = fsub(a, b);
    } else if (R == 4) {
        const double2 t0 = fadd(v[0], v[2]), t1 = fsub(v[0], v[2]), t2 = fadd(v[1], v[3]), t3 = mul_mi(fsub(v[1], v[3]));
        v[0] = fadd(t0, t2);
        v[1] = fadd(t1, t3);
        v[2] = fsub(t0, t2);
        v[3] = fsub(t1, t3);
    } else if (R == 3) {
        const double s3 = 0.86602540378443864676;
        const double2 t = fadd(v[1], v[2]), d = mul_mi(fsub(v[1], v[2]));
        const double2 m = make_double2(v[0].x - 0.5 * t.x, v[0].y - 0.5 * t.y);
        v[0] = fadd(v[0], t);
        v[1] = make_double2(m.x + s3 * d.x, m.y + s3 * d.y);
        v[2] = make_double2(m.x - s3 * d.x, m.y - s3 * d.y);
    } else {      // R == 5
        const double c1 = 0.30901699437494742410, c2 = -0.80901699437494742410;      // cos(2 pi / 5), cos(4 pi / 5)
        const double s1 = 0.95105651629515357212, s2 = 0.58778525229247312917;       // sin(2 pi / 5), sin(4 pi / 5)
        const double2 t1 = fadd(v[1], v[4]), t2 = fadd(v[2], v[3]), t3 = fsub(v[1], v[4]), t4 = fsub(v[2], v[3]);
        const double2 a1 = make_double2(v[0].x + c1 * t1.x + c2 * t2.x, v[0].y + c1 * t1.y + c2 * t2.y);
        const double2 a2 = make_double2(v[0].x + c2 * t1.x + c1 * t2.x, v[0].y + c2 * t1.y + c1 * t2.y);
        const double2 b1 = mul_mi(make_double2(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y));
        const double2 b2 = mul_mi(make_double2(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y));
        v[0] = fadd(v[0], fadd(t1, t2));
        v[1] = fadd(a1, b1);
        v[4] = fsub(a1, b1);
        v[2] = fadd(a2, b2);
        v[3] = fsub(a2, b2);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) y[j0 + r * Ns] = v[r];
}

__global__ __launch_bounds__(kFftThreads) void own_fft_pass_kernel(OwnFftPass a) {
    extern __shared__ double2 fft_lds[];
    double2* X = fft_lds;
    double2* Y = fft_lds + (size_t)a.L * a.ld;
    const int n = a.n, L = a.L, ld = a.ld, tid = threadIdx.x;
    const bool contig = a.stride == 1;
    __shared__ int64_t lineo[32];   // contiguous axis: the lines of the tile (they are not adjacent when the slower axes are windowed)
    int64_t sbase = 0, dbase = 0;   // strided axes: addresses of stored entry 0 of line 0 of the tile
    int nl;
    if (contig) {
        const int64_t oc = (int64_t)blockIdx.x * L;
        nl = (int)min((int64_t)L, a.nouter - oc);
        if (tid < nl) lineo[tid] = decode_outer(a, oc + tid);
        __syncthreads();
        if (a.acc) {                 // fixed-point accumulator -> complex values (reduce_slabs_kernel<true>'s conversion), full lines
            const int64_t per_batch = a.acc_cells / n;
            const double s0 = a.acc_scale[1], s1 = a.acc_scale[3];
            for (int w = tid; w < nl * n; w += kFftThreads) {
                const int l = w / n, p = w - l * n;
                const int64_t b = lineo[l] / per_batch, line = lineo[l] - b * per_batch;
                long long* c0 = a.acc + b * a.acc_channels * a.acc_cells + line * n + p;
                double2 v = make_double2((double)c0[0] * s0, a.acc_channels == 2 ? (double)c0[a.acc_cells] * s1 : 0.0);
                if (a.acc_reset) {
                    c0[0] = 0;
                    if (a.acc_channels == 2) c0[a.acc_cells] = 0;
                }
                if (a.backward) v.y = -v.y;
                X[l * ld + p] = v;
            }
        } else
        for (int w = tid; w < nl * n; w += kFftThreads) {
            const int l = w / n, p = w - l * n;
            const int q = stored_index(p, n, a.n_src);
            double2 v = q >= 0 ? a.src[lineo[l] * a.n_src + q] : make_double2(0.0, 0.0);
            if (a.backward) v.y = -v.y;
            X[l * ld + p] = v;
        }
    } else {
        const int64_t groups = (a.stride + L - 1) / L;
        const int64_t ocs = (int64_t)blockIdx.x / groups, i0 = ((int64_t)blockIdx.x - ocs * groups) * L;
        const int64_t o = decode_outer(a, ocs);
        nl = (int)min((int64_t)L, a.stride - i0);
        sbase = o * a.n_src * a.stride + i0;
        dbase = o * a.n_dst * a.stride + i0;
        for (int w = tid; w < n * L; w += kFftThreads) {
            const int p = w / L, l = w - p * L;
            if (l < nl) {
                const int q = stored_index(p, n, a.n_src);
                double2 v = q >= 0 ? a.src[sbase + (int64_t)q * a.stride + l] : make_double2(0.0, 0.0);
                if (a.backward) v.y = -v.y;
                X[l * ld + p] = v;
            }
        }
    }
    __syncthreads();
    int Ns = 1;
    for (int s = 0; s < a.nfac; ++s) {
        const int R = a.fac[s];
        const int nb = n / R;
        const int tstep = n / (Ns * R);
        for (int w = tid; w < nl * nb; w += kFftThreads) {
            const int l = w / nb, j = w - l * nb;
            const int k = j % Ns;
            const int j0 = (j - k) * R + k;
            const double2* x = X + l * ld;
            double2* y = Y + l * ld;
            if (R == 4) butterfly<4>(x, y, a.tw, j, nb, k, tstep, j0, Ns);
            else if (R == 2) butterfly<2>(x, y, a.tw, j, nb, k, tstep, j0, Ns);
            else if (R == 3) butterfly<3>(x, y, a.tw, j, nb, k, tstep, j0, Ns);
            else butterfly<5>(x, y, a.tw, j, nb, k, tstep, j0, Ns);
        }
        __syncthreads();
        double2* t = X;
        X = Y;
        Y = t;
        Ns *= R;
    }
    const int nd = a.n_dst, shift = n - nd, npos = (nd + 1) / 2;
    if (contig) {
        for (int w = tid; w < nl * nd; w += kFftThreads) {
            const int l = w / nd, q = w - l * nd;
            double2 v = X[l * ld + (q < npos ? q : q + shift)];
            if (a.backward) v.y = -v.y;
            a.dst[lineo[l] * nd + q] = v;
        }
    } else {
        for (int w = tid; w < nd * L; w += kFftThreads) {
            const int q = w / L, l = w - q * L;
            if (l < nl) {
                double2 v = X[l * ld + (q < npos ? q : q + shift)];
                if (a.backward) v.y = -v.y;
                a.dst[dbase + (int64_t)q * a.stride + l] = v;
            }
        }
    }
}

static bool factorize(int64_t n, int* fac, int* nfac) {
    int k = 0;
    while (n % 4 == 0 && k < kFftMaxFactors) {
        fac[k++] = 4;
        n /= 4;
    }
    for (int r : {2, 3, 5})
        while (n % r == 0 && k < kFftMaxFactors) {
            fac[k++] = r;
            n /= r;
        }
    *nfac = k;
    return n == 1;
}

bool own_fft_supported(int rank, const int64_t* n) {
    static const bool off = std::getenv("EFGP_FFT_ROCFFT") != nullptr;        // test / comparison hook: every transform through hipFFT
    if (off || rank < 1 || rank > 3) return false;
    for (int a = 0; a < rank; ++a) {
        int fac[kFftMaxFactors], nf;
        if (n[a] < 1 || n[a] > kFftMaxLine || !factorize(n[a], fac, &nf)) return false;
    }
    return true;
}

static const double2* twiddle_table(DeviceCtx* ctx, int64_t n, hipStream_t stream) {
    auto it = ctx->twiddles.find(n);
    if (it != ctx->twiddles.end()) return (const double2*)it->second;
    std::vector<double2> tw((size_t)n);
    const long double two_pi = 2.0L * acosl(-1.0L);
    for (int64_t q = 0; q < n; ++q) {
        const long double ang = -two_pi * (long double)q / (long double)n;
        tw[(size_t)q] = make_double2((double)cosl(ang), (double)sinl(ang));
    }
    double2* dtw = nullptr;
    if (hipMalloc((void**)&dtw, (size_t)n * sizeof(double2)) != hipSuccess) return nullptr;
    if (hipMemcpyAsync(dtw, tw.data(), (size_t)n * sizeof(double2), hipMemcpyHostToDevice, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess) {
        (void)hipFree(dtw);
        return nullptr;
    }
    ctx->twiddles[n] = dtw;
    return dtw;
}

int own_fft_prepare(DeviceCtx* ctx, int rank, const int64_t* n, hipStream_t stream) {
    for (int a = 0; a < rank; ++a)
        if (n[a] > 1 && !twiddle_table(ctx, n[a], stream)) {
            set_error("own_fft: twiddle table for length %lld failed", (long long)n[a]);
            return EFGP_ENOMEM;
        }
    return EFGP_OK;
}

// one pass: `outer` x `stride` lines of transform length len; n_src stored entries per source line, n_dst kept per destination line
struct SlowWindow {
    int nslow = 0;
    int64_t ext[2] = {1, 1}, lo[2] = {0, 0}, cnt[2] = {1, 1};
};
static int launch_pass(DeviceCtx* ctx, const double2* src, double2* dst, int64_t len, int64_t n_src, int64_t n_dst, int64_t stride, int64_t outer,
                       bool forward, hipStream_t stream, const SlowWindow* win = nullptr, const FftAccSource* acc = nullptr) {
    // the most the L rule below can ask for: one line of the longest length (131104 B) or 150 KB of strided lines
    constexpr int kPassLdsMax = 150 << 10;
    static_assert(2 * (kFftMaxLine + 1) * (int)sizeof(double2) <= kPassLdsMax, "a single line must fit the LDS budget");
    bool& attr_set = per_device_flag("own_fft_pass");
    if (!attr_set) {
        if (kPassLdsMax + 256 > ctx->max_lds) {
            set_error("own_fft: the device offers %d bytes of LDS per workgroup, the line transform needs %d", ctx->max_lds, kPassLdsMax + 256);
            return EFGP_EUNSUPPORTED;
        }
        EFGP_HIP_CHECK(hipFuncSetAttribute((const void*)own_fft_pass_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kPassLdsMax));
        attr_set = true;
    }
    OwnFftPass p;
    p.src = src;
    p.dst = dst;
    p.tw = twiddle_table(ctx, len, stream);
    if (!p.tw) return EFGP_ENOMEM;
    p.n = (int)len;
    p.n_src = (int)n_src;
    p.n_dst = (int)n_dst;
    factorize(len, p.fac, &p.nfac);
    p.stride = stride;
    p.nouter = outer;
    p.ld = (int)len + 1;
    p.backward = forward ? 0 : 1;
    p.acc = acc ? acc->acc : nullptr;
    p.acc_channels = acc ? acc->channels : 0;
    p.acc_reset = acc ? acc->reset : 0;
    p.acc_cells = acc ? acc->cells : 0;
    p.acc_scale = acc ? acc->scale : nullptr;
    p.nslow = win ? win->nslow : 0;
    for (int q = 0; q < 2; ++q) {
        p.slow_ext[q] = win ? win->ext[q] : 1;
        p.slow_lo[q] = win ? win->lo[q] : 0;
        p.slow_cnt[q] = win ? win->cnt[q] : 1;
    }
    // lines per workgroup: ~48 KB of LDS for the two buffers (three workgroups per CU), at least 8 adjacent lines on a strided axis
    // (128-byte segments), all of them resident for short lines
    const int64_t bytes_per_line = 2 * (int64_t)p.ld * (int64_t)sizeof(double2);
    int L = (int)std::max<int64_t>(1, (48 << 10) / bytes_per_line);
    if (stride > 1) L = std::max(L, (int)std::min<int64_t>(8, (int64_t)kPassLdsMax / bytes_per_line));
    L = std::min(L, 32);
    const int64_t avail = stride > 1 ? stride : outer;
    L = (int)std::max<int64_t>(1, std::min<int64_t>(L, avail));
    p.L = L;
    const int64_t blocks = stride > 1 ? outer * ((stride + L - 1) / L) : (outer + L - 1) / L;
    if (blocks > 2147483647LL) return EFGP_EUNSUPPORTED;
    hipLaunchKernelGGL(own_fft_pass_kernel, dim3((unsigned)blocks), dim3(kFftThreads), (size_t)L * bytes_per_line, stream, p);
    EFGP_HIP_CHECK(hipGetLastError());
    return EFGP_OK;
}

int own_fft_exec(DeviceCtx* ctx, int rank, const int64_t* n, int64_t batch, double2* data, bool forward, hipStream_t stream) {
    if (!own_fft_supported(rank, n)) return EFGP_EUNSUPPORTED;
    int64_t stride = 1;
    for (int ax = rank - 1; ax >= 0; --ax) {
        const int64_t len = n[ax];
        if (len > 1) {
            int64_t outer = batch;
            for (int q = 0; q < ax; ++q) outer *= n[q];
            const int rc = launch_pass(ctx, data, data, len, len, len, stride, outer, forward, stream);
            if (rc != EFGP_OK) return rc;
        }
        stride *= len;
    }
    return EFGP_OK;
}

// Transforms around a zero-padded Toeplitz product: the array has extents n, but on every axis q only the window
// [lo[q], lo[q] + cnt[q]) matters -- as non-zero input of a forward transform (fastest axis first: a pass along ax skips the lines
// whose slower indices lie outside their windows, they are zero and stay zero), or as wanted output of a backward one (slowest
// axis first: after the pass along an axis only its window is carried on).  With cnt = n / 2 a rank-3 pair of transforms runs
// 3.3 array passes instead of 6.
int own_fft_exec_windowed(DeviceCtx* ctx, int rank, const int64_t* n, int64_t batch, double2* data, bool forward, const int64_t* lo,
                          const int64_t* cnt, bool slowest_first, hipStream_t stream) {
    if (!own_fft_supported(rank, n)) return EFGP_EUNSUPPORTED;
    for (int step = 0; step < rank; ++step) {
        const int ax = slowest_first ? step : rank - 1 - step;
        if (n[ax] == 1) continue;
        int64_t stride = 1, outer = batch;
        for (int q = ax + 1; q < rank; ++q) stride *= n[q];
        SlowWindow win;
        win.nslow = ax;
        for (int q = 0; q < ax; ++q) {
            win.ext[q] = n[q];
            win.lo[q] = lo[q];
            win.cnt[q] = cnt[q];
            outer *= cnt[q];
        }
        const int rc = launch_pass(ctx, data, data, n[ax], n[ax], n[ax], stride, outer, forward, stream, &win);
        if (rc != EFGP_OK) return rc;
    }
    return EFGP_OK;
}

// Type-1 side: transform of `batch` arrays of extents nf, keeping per axis only the nc[a] lowest-|frequency| bins (FFT order on the
// nc torus).  Fastest axis first, every pass writes the cropped lines only, so the later (strided) passes run on what is left:
// with nc = nf / 2 the three passes of a rank-3 transform move 2.6 array volumes instead of 6 -- and the mode extraction behind
// them reads the small result.  `fine` is destroyed; the result lands in `work` or `fine` (returned through *out).
// work: >= batch * prod_{a < rank-1} nf[a] * nc[rank-1] elements.
int own_fft_pruned_forward(DeviceCtx* ctx, int rank, const int64_t* nf, const int64_t* nc, int64_t batch, double2* fine, double2* work,
                           bool forward, double2** out, hipStream_t stream, const FftAccSource* acc) {
    if (!own_fft_supported(rank, nf)) return EFGP_EUNSUPPORTED;
    if (acc && nf[rank - 1] == 1) return EFGP_EUNSUPPORTED;       // the accumulator is read by the pass along the contiguous axis
    int64_t cur[3];
    for (int a = 0; a < rank; ++a) cur[a] = nf[a];
    double2* src = fine;
    for (int ax = rank - 1; ax >= 0; --ax) {
        int64_t stride = 1, outer = batch;
        for (int q = ax + 1; q < rank; ++q) stride *= cur[q];
        for (int q = 0; q < ax; ++q) outer *= cur[q];
        const int64_t keep = std::min(nc[ax], nf[ax]);
        if (nf[ax] == 1) continue;
        const bool from_acc = acc && ax == rank - 1;             // out of place by nature: the complex values go to `fine` / `work`
        double2* dst = (keep == nf[ax] && !from_acc) ? src : (src == fine && !from_acc ? work : fine);
        if (from_acc && keep < nf[ax]) dst = work;
        const int rc = launch_pass(ctx, src, dst, nf[ax], nf[ax], keep, stride, outer, forward, stream, nullptr, from_acc ? acc : nullptr);
        if (rc != EFGP_OK) return rc;
        cur[ax] = keep;
        src = dst;
    }
    *out = src;
    return EFGP_OK;
}

// Type-2 side: `modes` holds batch arrays of extents nc (bins in FFT order on the nc torus; every other bin of the nf torus is
// zero); the full arrays of extents nf are written to `fine`.  Slowest axis first: the array grows pass by pass and only the last
// (contiguous) pass touches the full volume.  work: two regions of >= batch * prod_{a < rank-1} nf[a] * nc[rank-1] elements each;
// `modes` may be region 0.
int own_fft_pruned_backward(DeviceCtx* ctx, int rank, const int64_t* nc, const int64_t* nf, int64_t batch, const double2* modes, double2* fine,
                            double2* work, int64_t region, bool forward, hipStream_t stream) {
    if (!own_fft_supported(rank, nf)) return EFGP_EUNSUPPORTED;
    int64_t cur[3];
    for (int a = 0; a < rank; ++a) cur[a] = std::min(nc[a], nf[a]);
    const double2* src = modes;
    int flip = modes == work ? 1 : 0;
    for (int ax = 0; ax < rank; ++ax) {
        int64_t stride = 1, outer = batch;
        for (int q = ax + 1; q < rank; ++q) stride *= cur[q];
        for (int q = 0; q < ax; ++q) outer *= cur[q];
        const bool last = ax == rank - 1;
        double2* dst = last ? fine : work + (int64_t)flip * region;
        const int rc = launch_pass(ctx, src, dst, nf[ax], cur[ax], nf[ax], stride, outer, forward, stream);
        if (rc != EFGP_OK) return rc;
        cur[ax] = nf[ax];
        src = dst;
        flip ^= 1;
    }
    return EFGP_OK;
}

int fft_c2c(DeviceCtx* ctx, int rank, const int64_t* n, int64_t batch, double2* data, bool forward, hipStream_t stream) {
    if (own_fft_supported(rank, n)) return own_fft_exec(ctx, rank, n, batch, data, forward, stream);
    hipfftHandle fh;
    const int rc = fft_plan(ctx, rank, n, batch, stream, &fh);
    if (rc != EFGP_OK) return rc;
    EFGP_FFT_CHECK(hipfftExecZ2Z(fh, (hipfftDoubleComplex*)data, (hipfftDoubleComplex*)data, forward ? HIPFFT_FORWARD : HIPFFT_BACKWARD));
    return EFGP_OK;
}

}  // namespace efgp

extern "C" int efgp_fft_c2c(int device, int rank, const long long* n, long long batch, void* data, int forward, int use_rocfft, void* stream_) {
    using namespace efgp;
    EFGP_REQUIRE(rank >= 1 && rank <= 3 && n && data && batch >= 1, "efgp_fft_c2c: bad argument");
    DeviceGuard guard(device, (hipStream_t)stream_);
    DeviceCtx* ctx = device_ctx(device);
    if (!ctx) return EFGP_EHIP;
    int64_t nn[3];
    for (int a = 0; a < rank; ++a) nn[a] = n[a];
    hipStream_t stream = (hipStream_t)stream_;
    if (use_rocfft) {
        hipfftHandle fh;
        const int rc = fft_plan(ctx, rank, nn, batch, stream, &fh);
        if (rc != EFGP_OK) return rc;
        EFGP_FFT_CHECK(hipfftExecZ2Z(fh, (hipfftDoubleComplex*)data, (hipfftDoubleComplex*)data, forward ? HIPFFT_FORWARD : HIPFFT_BACKWARD));
        return EFGP_OK;
    }
    return own_fft_exec(ctx, rank, nn, batch, (double2*)data, forward != 0, stream);
}
