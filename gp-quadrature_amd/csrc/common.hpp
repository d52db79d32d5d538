// Shared host-side plumbing for libefgp_hip: error reporting, per-device scratch buffers and a
// hipFFT plan cache.  gfx950 only; no portability layers.
#pragma once
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/efgp_hip.h"

namespace efgp {

void set_error(const char* fmt, ...);

#define EFGP_HIP_CHECK(expr)                                                                   \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess) {                                                               \
            ::efgp::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, \
                              __LINE__);                                                       \
            return EFGP_EHIP;                                                                  \
        }                                                                                      \
    } while (0)

#define EFGP_FFT_CHECK(expr)                                                              \
    do {                                                                                  \
        hipfftResult r__ = (expr);                                                        \
        if (r__ != HIPFFT_SUCCESS) {                                                      \
            ::efgp::set_error("%s failed: hipfft status %d (%s:%d)", #expr, (int)r__,     \
                              __FILE__, __LINE__);                                        \
            return EFGP_EHIP;                                                             \
        }                                                                                 \
    } while (0)

#define EFGP_REQUIRE(cond, ...)              \
    do {                                     \
        if (!(cond)) {                       \
            ::efgp::set_error(__VA_ARGS__);  \
            return EFGP_EINVAL;              \
        }                                    \
    } while (0)

// scratch slots (grow-only device buffers, one set per device)
enum Slot {
    SLOT_FINE = 0,     // complex fine grids of the NUFFT (nbatch * prod nf)
    SLOT_SLABS,        // per-workgroup partial fine grids of the spreader
    SLOT_TOEP_PAD,     // zero-padded FFT buffer of the Toeplitz apply / CG
    SLOT_CG_VEC,       // r, p, Ap of the CG solver
    SLOT_CG_SCALARS,   // per-row scalars and flags of the CG solver
    SLOT_MISC,         // reductions
    SLOT_SCALE,        // fixed-point scale of the spreader
    SLOT_G2M,          // grid_to_modes_kernel: per-tile arrival counters (first 64 KB, zero between launches) + partial sums
    SLOT_FFT_WORK,     // intermediate arrays of the pruned in-house transforms (line_fft.hip)
    SLOT_COUNT
};

struct DeviceCtx {
    int device = -1;
    int num_cu = 0;
    int max_lds = 0;            // bytes of LDS one workgroup may use
    void* buf[SLOT_COUNT] = {nullptr};
    size_t cap[SLOT_COUNT] = {0};
    // key: rank, n0, n1, n2, batch
    struct FftEntry {
        hipfftHandle handle;
        unsigned long long stamp;   // last use (monotone counter): the least recently used plan is evicted
    };
    std::map<std::tuple<int, int64_t, int64_t, int64_t, int64_t>, FftEntry> fft_plans;
    unsigned long long fft_clock = 0;
    int* host_pinned = nullptr;   // small pinned host buffer for status read-back
    size_t host_pinned_bytes = 0;
    // free lists of small device blocks (size -> pointers) so that per-fit objects (Toeplitz spectra,
    // twiddle tables) do not pay hipMalloc/hipFree each time
    std::map<size_t, std::vector<void*>> pool;
    size_t pool_bytes = 0;      // bytes parked in the free lists (bounded: see pool_free)
    // cached spreading-window data, owned by the NUFFT translation unit (opaque here)
    std::vector<void*> window_cache;
    // cached FFT twiddle tables of the persistent CG: length -> device table
    std::map<int64_t, void*> twiddles;
    // SLOT_SCALE holds the spreader's max|c| accumulator: zeroed once, then reset by the kernel that consumes it
    bool scale_slot_ready = false;
    // SLOT_G2M's counter block has been zeroed for the buffer currently allocated (the kernel leaves it zero)
    const void* g2m_zeroed_for = nullptr;
    // leading bytes of SLOT_SLABS known to be zero: the MFMA spreader's int64 grid is reset by the kernel that converts it,
    // so the next spread needs no memset launch (any other request for the slot clears this)
    size_t slabs_zero_bytes = 0;
    // side stream for hipGraph capture (capture is not allowed on the legacy default stream torch usually hands us)
    hipStream_t aux_stream = nullptr;
    hipEvent_t aux_event = nullptr;
    // the stream the context's work was last enqueued on, and the event of a hand-over to another one (stream_handover)
    hipStream_t last_stream = nullptr;
    bool last_stream_set = false;
    hipEvent_t handover_event = nullptr;
    // one entry point at a time per device (taken by DeviceGuard; recursive: entry points nest)
    std::recursive_mutex mu;
    // ring of pinned staging slots for small host -> device uploads that must not drain the stream (upload_small)
    static constexpr int kUploadSlots = 8;
    static constexpr size_t kUploadSlotBytes = size_t(64) << 10;
    char* up_ring = nullptr;
    hipEvent_t up_event[kUploadSlots] = {nullptr};
    bool up_used[kUploadSlots] = {false};
    int up_turn = 0;
};

// a once-per-DEVICE flag (function attributes such as hipFuncAttributeMaxDynamicSharedMemorySize are per device: a
// process-wide static would leave the second GPU of a process without them); keyed by a site name, for the current device
bool& per_device_flag(const char* key);

// returns the context of `device` (creates it, queries properties); nullptr + error on failure
DeviceCtx* device_ctx(int device);
// grow-only scratch; synchronises the device before replacing a buffer.  nullptr on failure.
void* scratch(DeviceCtx* ctx, Slot slot, size_t bytes);
// cached batched complex-to-complex double plan bound to `stream`
int fft_plan(DeviceCtx* ctx, int rank, const int64_t* n, int64_t batch, hipStream_t stream, hipfftHandle* out);
int* pinned_host(DeviceCtx* ctx, size_t bytes);
// Stream-ordered upload of a small host array WITHOUT waiting for the stream: the bytes are staged in one of a few pinned slots
// (a slot is reused only after the event behind its last copy has passed).  A copy from pageable memory, or a
// hipStreamSynchronize behind it to keep the source alive, stalls the host until everything queued has run -- in a training
// loop that is every step with a new mode count.  Arrays beyond a slot fall back to the blocking copy.
int upload_small(DeviceCtx* ctx, const void* src, size_t bytes, void* dst, hipStream_t stream);
// pooled device blocks (rounded up to 4 KB multiples); pool_free returns the block to the free list
void* pool_alloc(DeviceCtx* ctx, size_t bytes);
void pool_free(DeviceCtx* ctx, void* p, size_t bytes);
void release_ctx(int device);

// Wait for `stream` by polling hipStreamQuery (for up to 50 ms, then hipStreamSynchronize; EFGP_BLOCKING_WAIT=1 goes
// straight to the latter).  Our waits are short -- a burst of CG iterations, a read-back of a few scalars -- so the
// polling thread never reaches the runtime's interrupt sleep.  (The 90-ms stalls once seen inside these waits were CFS
// CPU-quota throttling of the whole process by torch's oversized OpenMP pool: efgp_hip/cpu_quota.py.)
hipError_t stream_wait(hipStream_t stream);

// Diagnostic hook of the CG solvers (efgp_cg_record_history): device buffer that receives row 0's relative residual
// |r_i| / |b| of every iteration of the solves enqueued while it is set.
struct CgHistory {
    double* buf = nullptr;
    int capacity = 0;
};
CgHistory cg_history();

// Optional HIP-event timing of selected kernels (see efgp_kernel_timing in the C ABI).
bool timing_enabled();
struct KernelTimer {     // RAII: records start at construction, stop at destruction, on `stream`
    KernelTimer(const char* name, hipStream_t stream);
    ~KernelTimer();
    int slot = -1;
    hipStream_t stream = nullptr;
};

// A device context is single-stream at any moment (shared scratch, pooled blocks reused in stream order, caches): a caller that
// comes in on ANOTHER stream than the context's last one is ordered behind everything queued on that one (an event recorded
// there, awaited here; a device-wide wait when that fails) instead of racing with it.  Nothing happens while the stream stays
// the same.  Host threads are not serialised: one thread per device at a time, as before.
void stream_handover(int device, hipStream_t stream);
// the per-device lock behind DeviceGuard: returns the context it locked (nullptr when the device has none and none can be made)
DeviceCtx* device_ctx_lock(int device);
void device_ctx_unlock(DeviceCtx* ctx);

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        // host threads: the device's context (scratch, pool, caches, the stream it is bound to) belongs to one call at a time; the
        // lock is recursive because entry points call each other (efgp_gradient_step, the synchronous solver's fallbacks)
        locked = device_ctx_lock(dev);
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
        target = dev;
    }
    // entry points that enqueue work: the guard also hands the context over to the caller's stream
    DeviceGuard(int dev, hipStream_t stream) : DeviceGuard(dev) {
        if (ok) stream_handover(dev, stream);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
    ~DeviceGuard() {
        if (ok && prev >= 0 && prev != target) (void)hipSetDevice(prev);
        if (locked) device_ctx_unlock(locked);
    }
    int target = -1;
    DeviceCtx* locked = nullptr;
};

inline int64_t next_pow2(int64_t n) {
    int64_t p = 1;
    while (p < n) p <<= 1;
    return p;
}

}  // namespace efgp
