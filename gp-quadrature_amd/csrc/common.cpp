// Host-side plumbing: error string, device contexts, scratch buffers, hipFFT plan cache,
// and the host-only window entry points of the C ABI.
#include "common.hpp"

#include <chrono>
#include <map>
#include <cstdlib>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <mutex>

#include "es_kernel.hpp"

namespace efgp {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static std::mutex g_mu;
static std::map<int, std::unique_ptr<DeviceCtx>> g_ctx;

bool& per_device_flag(const char* key) {
    static std::map<std::pair<int, std::string>, bool> flags;
    int dev = 0;
    (void)hipGetDevice(&dev);
    return flags[std::make_pair(dev, std::string(key))];
}

DeviceCtx* device_ctx(int device) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_ctx.find(device);
    if (it != g_ctx.end()) return it->second.get();
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) {
        set_error("device %d not available (hipGetDeviceCount=%d): the HIP path needs a GPU", device, count);
        return nullptr;
    }
    auto ctx = std::make_unique<DeviceCtx>();
    ctx->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
        set_error("hipGetDeviceProperties(%d) failed", device);
        return nullptr;
    }
    ctx->num_cu = prop.multiProcessorCount;
    int lds = 0;
    if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device) != hipSuccess || lds <= 0)
        lds = 65536;
    ctx->max_lds = lds;
    DeviceCtx* raw = ctx.get();
    g_ctx[device] = std::move(ctx);
    return raw;
}

DeviceCtx* device_ctx_lock(int device) {
    DeviceCtx* ctx = device_ctx(device);
    if (ctx) ctx->mu.lock();
    return ctx;
}
void device_ctx_unlock(DeviceCtx* ctx) {
    if (ctx) ctx->mu.unlock();
}

void stream_handover(int device, hipStream_t stream) {
    DeviceCtx* ctx = device_ctx(device);
    if (!ctx) return;
    if (ctx->last_stream_set && ctx->last_stream == stream) return;
    if (ctx->last_stream_set && std::getenv("EFGP_NO_STREAM_HANDOVER") == nullptr) {      // (the switch is for the test that shows the race)
        bool ordered = false;
        if (!ctx->handover_event && hipEventCreateWithFlags(&ctx->handover_event, hipEventDisableTiming) != hipSuccess)
            ctx->handover_event = nullptr;
        if (ctx->handover_event && hipEventRecord(ctx->handover_event, ctx->last_stream) == hipSuccess &&
            hipStreamWaitEvent(stream, ctx->handover_event, 0) == hipSuccess)
            ordered = true;
        if (!ordered) {                       // e.g. the previous stream no longer exists
            (void)hipGetLastError();
            (void)hipDeviceSynchronize();
        }
    }
    ctx->last_stream = stream;
    ctx->last_stream_set = true;
}

void* scratch(DeviceCtx* ctx, Slot slot, size_t bytes) {
    if (bytes == 0) bytes = 256;
    if (slot == SLOT_SLABS) ctx->slabs_zero_bytes = 0;      // callers that rely on it re-establish it themselves
    if (ctx->cap[slot] >= bytes) return ctx->buf[slot];
    size_t want = bytes + bytes / 4;
    want = (want + 255) & ~size_t(255);
    if (ctx->buf[slot]) {
        (void)hipDeviceSynchronize();
        (void)hipFree(ctx->buf[slot]);
        ctx->buf[slot] = nullptr;
        ctx->cap[slot] = 0;
    }
    void* p = nullptr;
    if (hipMalloc(&p, want) != hipSuccess) {
        set_error("hipMalloc(%zu bytes) for scratch slot %d failed", want, (int)slot);
        return nullptr;
    }
    ctx->buf[slot] = p;
    ctx->cap[slot] = want;
    return p;
}

// A long hyper-parameter optimisation visits many grid sizes; rocFFT plans own device work buffers, so the cache
// is bounded and evicts the least recently used plan (hipfftDestroy frees its buffers behind an implicit sync).
constexpr size_t kMaxFftPlans = 32;

int fft_plan(DeviceCtx* ctx, int rank, const int64_t* n, int64_t batch, hipStream_t stream, hipfftHandle* out) {
    auto key = std::make_tuple(rank, n[0], rank > 1 ? n[1] : 0, rank > 2 ? n[2] : 0, batch);
    auto it = ctx->fft_plans.find(key);
    hipfftHandle h;
    if (it == ctx->fft_plans.end()) {
        if (ctx->fft_plans.size() >= kMaxFftPlans) {
            auto victim = ctx->fft_plans.begin();
            for (auto jt = ctx->fft_plans.begin(); jt != ctx->fft_plans.end(); ++jt)
                if (jt->second.stamp < victim->second.stamp) victim = jt;
            (void)hipfftDestroy(victim->second.handle);
            ctx->fft_plans.erase(victim);
        }
        int dims[3];
        int64_t dist = 1;
        for (int a = 0; a < rank; ++a) {
            dims[a] = (int)n[a];
            dist *= n[a];
        }
        EFGP_FFT_CHECK(hipfftCreate(&h));
        size_t work = 0;
        EFGP_FFT_CHECK(hipfftMakePlanMany(h, rank, dims, nullptr, 1, (int)dist, nullptr, 1, (int)dist, HIPFFT_Z2Z,
                                          (int)batch, &work));
        ctx->fft_plans[key] = DeviceCtx::FftEntry{h, ++ctx->fft_clock};
    } else {
        h = it->second.handle;
        it->second.stamp = ++ctx->fft_clock;
    }
    EFGP_FFT_CHECK(hipfftSetStream(h, stream));
    *out = h;
    return EFGP_OK;
}

int* pinned_host(DeviceCtx* ctx, size_t bytes) {
    if (ctx->host_pinned_bytes >= bytes) return ctx->host_pinned;
    if (ctx->host_pinned) (void)hipHostFree(ctx->host_pinned);
    ctx->host_pinned = nullptr;
    ctx->host_pinned_bytes = 0;
    void* p = nullptr;
    size_t want = bytes < 4096 ? 4096 : bytes * 2;
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) {
        set_error("hipHostMalloc(%zu) failed", want);
        return nullptr;
    }
    ctx->host_pinned = (int*)p;
    ctx->host_pinned_bytes = want;
    return ctx->host_pinned;
}

int upload_small(DeviceCtx* ctx, const void* src, size_t bytes, void* dst, hipStream_t stream) {
    if (bytes == 0) return EFGP_OK;
    if (bytes > DeviceCtx::kUploadSlotBytes) {
        EFGP_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream));
        EFGP_HIP_CHECK(hipStreamSynchronize(stream));
        return EFGP_OK;
    }
    if (!ctx->up_ring) {
        void* p = nullptr;
        if (hipHostMalloc(&p, DeviceCtx::kUploadSlots * DeviceCtx::kUploadSlotBytes, hipHostMallocDefault) != hipSuccess) {
            set_error("hipHostMalloc of the upload ring failed");
            return EFGP_ENOMEM;
        }
        ctx->up_ring = (char*)p;
        for (int i = 0; i < DeviceCtx::kUploadSlots; ++i) EFGP_HIP_CHECK(hipEventCreateWithFlags(&ctx->up_event[i], hipEventDisableTiming));
    }
    const int k = ctx->up_turn;
    ctx->up_turn = (k + 1) % DeviceCtx::kUploadSlots;
    if (ctx->up_used[k]) EFGP_HIP_CHECK(hipEventSynchronize(ctx->up_event[k]));      // eight uploads ago: long done
    char* slot = ctx->up_ring + (size_t)k * DeviceCtx::kUploadSlotBytes;
    std::memcpy(slot, src, bytes);
    EFGP_HIP_CHECK(hipMemcpyAsync(dst, slot, bytes, hipMemcpyHostToDevice, stream));
    EFGP_HIP_CHECK(hipEventRecord(ctx->up_event[k], stream));
    ctx->up_used[k] = true;
    return EFGP_OK;
}

static size_t pool_round(size_t bytes) { return (std::max<size_t>(bytes, 1) + 4095) & ~size_t(4095); }

void* pool_alloc(DeviceCtx* ctx, size_t bytes) {
    const size_t sz = pool_round(bytes);
    auto it = ctx->pool.find(sz);
    if (it != ctx->pool.end() && !it->second.empty()) {
        void* p = it->second.back();
        it->second.pop_back();
        ctx->pool_bytes -= sz;
        return p;
    }
    void* p = nullptr;
    if (hipMalloc(&p, sz) != hipSuccess) {
        set_error("hipMalloc(%zu) failed", sz);
        return nullptr;
    }
    return p;
}

// Blocks go back to per-size free lists; the lists together are capped (changing grid sizes would otherwise park
// one block per size forever): beyond the cap the block is released to the runtime instead.
constexpr size_t kPoolCapBytes = size_t(4) << 30;      // small next to 288 GB of HBM; large plans park several 100-MB blocks

void pool_free(DeviceCtx* ctx, void* p, size_t bytes) {
    if (!p) return;
    const size_t sz = pool_round(bytes);
    if (ctx->pool_bytes + sz > kPoolCapBytes) {
        (void)hipFree(p);          // implicit device synchronisation: nothing queued can still be using it
        return;
    }
    ctx->pool[sz].push_back(p);
    ctx->pool_bytes += sz;
}

void release_ctx(int device) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto it = g_ctx.begin(); it != g_ctx.end();) {
        if (device >= 0 && it->first != device) {
            ++it;
            continue;
        }
        DeviceCtx* c = it->second.get();
        int prev = -1;
        (void)hipGetDevice(&prev);
        (void)hipSetDevice(c->device);
        (void)hipDeviceSynchronize();
        for (auto& kv : c->fft_plans) (void)hipfftDestroy(kv.second.handle);
        for (int s = 0; s < SLOT_COUNT; ++s)
            if (c->buf[s]) (void)hipFree(c->buf[s]);
        if (c->host_pinned) (void)hipHostFree(c->host_pinned);
        if (c->up_ring) {
            (void)hipHostFree(c->up_ring);
            for (int i = 0; i < DeviceCtx::kUploadSlots; ++i)
                if (c->up_event[i]) (void)hipEventDestroy(c->up_event[i]);
        }
        if (c->aux_event) (void)hipEventDestroy(c->aux_event);
        if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
        if (c->handover_event) (void)hipEventDestroy(c->handover_event);
        for (auto& kv : c->pool)
            for (void* p : kv.second) (void)hipFree(p);
        for (auto& kv : c->twiddles) (void)hipFree(kv.second);
        // window_cache entries hold device pointers too; they are leaked deliberately at teardown only if
        // the NUFFT unit did not clear them (it registers no destructor to keep this unit independent)
        if (prev >= 0) (void)hipSetDevice(prev);
        it = g_ctx.erase(it);
    }
}

// ---- CG residual history hook ----------------------------------------------------------------
// per host thread: the hook is set and consumed by the thread that runs the solve (two models solving on different threads must
// not write into each other's buffer)
static thread_local CgHistory g_cg_history;
CgHistory cg_history() { return g_cg_history; }

}  // namespace efgp

extern "C" int efgp_cg_record_history(double* history_dev, int capacity) {
    efgp::g_cg_history.buf = capacity > 0 ? history_dev : nullptr;
    efgp::g_cg_history.capacity = history_dev ? (capacity > 0 ? capacity : 0) : 0;
    return EFGP_OK;
}

namespace efgp {

// ---- kernel timing ------------------------------------------------------------------------
struct TimingRec {
    std::string name;
    hipEvent_t start, stop;
    int device;
    bool valid;          // both records succeeded
};
static bool g_timing = false;
static std::string g_timing_only;            // non-empty: only timers of this name record (efgp_kernel_timing_only)
static std::vector<TimingRec> g_recs;
// recycled events, per device (an event belongs to the device that was current when it was created; recording it on another
// device's stream fails): hipEventCreate per launch cost the host ~10 us per timed kernel
static std::map<int, std::vector<hipEvent_t>> g_free_events;
constexpr size_t kMaxTimingRecs = 1 << 16;

static int current_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) (void)hipGetLastError();
    return d;
}

static bool take_event(int dev, hipEvent_t* e) {
    auto& pool = g_free_events[dev];
    if (!pool.empty()) {
        *e = pool.back();
        pool.pop_back();
        return true;
    }
    return hipEventCreate(e) == hipSuccess;
}

bool timing_enabled() { return g_timing; }

hipError_t stream_wait(hipStream_t stream) {
    if (std::getenv("EFGP_BLOCKING_WAIT") == nullptr) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            const hipError_t e = hipStreamQuery(stream);
            if (e != hipErrorNotReady) return e;
            if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) break;
        }
        (void)hipGetLastError();     // hipErrorNotReady is not an error of ours
    }
    return hipStreamSynchronize(stream);
}

KernelTimer::KernelTimer(const char* name, hipStream_t s) : stream(s) {
    if (!g_timing || g_recs.size() >= kMaxTimingRecs) return;
    if (!g_timing_only.empty() && g_timing_only != name) return;
    TimingRec r;
    r.name = name;
    r.device = current_device();          // the library's entry points set the plan's / operator's device before they launch
    if (!take_event(r.device, &r.start)) return;
    if (!take_event(r.device, &r.stop)) {
        g_free_events[r.device].push_back(r.start);
        return;
    }
    r.valid = hipEventRecord(r.start, s) == hipSuccess;
    if (!r.valid) (void)hipGetLastError();
    g_recs.push_back(r);
    slot = (int)g_recs.size() - 1;
}

KernelTimer::~KernelTimer() {
    if (slot < 0) return;
    TimingRec& r = g_recs[(size_t)slot];
    if (r.valid && hipEventRecord(r.stop, stream) != hipSuccess) {
        (void)hipGetLastError();
        r.valid = false;                  // a record that failed must not be read as a duration
    }
}

static void clear_timing() {
    for (auto& r : g_recs) {
        g_free_events[r.device].push_back(r.start);
        g_free_events[r.device].push_back(r.stop);
    }
    g_recs.clear();
}

}  // namespace efgp

using namespace efgp;

extern "C" {

int efgp_version(void) { return 1000 * 0 + 1; }

const char* efgp_last_error(void) { return g_err; }

int efgp_release_workspaces(int device) {
    release_ctx(device);
    return EFGP_OK;
}

int efgp_kernel_timing(int enable) {
    (void)hipDeviceSynchronize();
    clear_timing();
    g_timing = enable != 0;
    return EFGP_OK;
}

int efgp_kernel_timing_only(const char* name) {
    g_timing_only = name ? name : "";
    return EFGP_OK;
}

int efgp_kernel_timing_read(const char* name, double* total_ms_out, int64_t* launches_out) {
    EFGP_REQUIRE(name && total_ms_out && launches_out, "efgp_kernel_timing_read: null argument");
    EFGP_HIP_CHECK(hipDeviceSynchronize());
    double tot = 0.0;
    int64_t cnt = 0;
    for (auto& r : g_recs) {
        if (r.name != name || !r.valid) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess) {
            tot += ms;
            ++cnt;
        }
    }
    *total_ms_out = tot;
    *launches_out = cnt;
    return EFGP_OK;
}

int efgp_window_width(double tol, double sigma) { return es_width_for_tol(tol, sigma); }

int efgp_window_width_nd(double tol, double sigma, int dim) { return es_width_for_tol(tol, sigma, dim); }

int64_t efgp_fine_grid_size(int64_t n_modes, double tol) {
    if (n_modes < 1) return 0;
    return es_fine_size(n_modes, tol, 2);
}

int64_t efgp_fine_grid_size_nd(int64_t n_modes, double tol, int dim, int dense) {
    if (n_modes < 1 || dim < 1 || dim > 3) return 0;
    return es_fine_size(n_modes, tol, dim, dense != 0);
}

int efgp_window_eval(double tol, double sigma, double X, int64_t* first_cell_out, double* vals_out, int* w_out,
                     double* beta_out) {
    EFGP_REQUIRE(first_cell_out && vals_out, "efgp_window_eval: null output");
    EsParams p;
    es_make_params(tol, sigma, &p);
    const int W = p.w;
    int64_t i0 = (int64_t)std::ceil(X - 0.5 * W);
    double s = 2.0 * ((double)i0 - X + 0.5 * W) - 1.0;
    const int stride = kMaxDegree + 1;
    for (int j = 0; j < W; ++j) {
        double acc = p.coef[j * stride + p.degree];
        for (int k = p.degree - 1; k >= 0; --k) acc = std::fma(acc, s, p.coef[j * stride + k]);
        vals_out[j] = acc;
    }
    *first_cell_out = i0;
    if (w_out) *w_out = W;
    if (beta_out) *beta_out = p.beta;
    return EFGP_OK;
}

int efgp_window_deconv(double tol, int64_t nf, int64_t n_modes, double* out) {
    EFGP_REQUIRE(out && nf > 0 && n_modes > 0 && n_modes <= nf, "efgp_window_deconv: bad sizes");
    EsParams p;
    es_make_params(tol, (double)nf / (double)n_modes, &p);
    std::vector<double> f;
    es_deconv_factors(p, nf, n_modes, &f);
    std::memcpy(out, f.data(), sizeof(double) * (size_t)n_modes);
    return EFGP_OK;
}

}  // extern "C"
