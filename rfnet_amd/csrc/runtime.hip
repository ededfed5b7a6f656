// runtime.hip -- version / status strings and the per-kernel hipEvent profiler behind
// rf_profile_enable() / rf_profile_collect().
#include <atomic>
#include <mutex>
#include <string.h>
#include <vector>

#include "common.hpp"

namespace {

struct Pending {
    const char *name;
    hipEvent_t start, stop;
};

// The profiler is the library's ONLY process-global state (documented in rfops.h).  The enable flag
// is an atomic; everything else is touched under g_mu.  A scope owns its two events from
// construction to destruction and only then files them, so a collect() in between cannot
// invalidate anything the scope refers to.
std::mutex g_mu;
std::atomic<bool> g_enabled{false};
std::vector<Pending> g_pending;
std::vector<hipEvent_t> g_pool;

hipEvent_t get_event() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

}  // namespace

namespace rf {

ProfScope::ProfScope(const char *n, hipStream_t s) : name(n), stream(s), start(nullptr), stop(nullptr) {
    if (!g_enabled.load(std::memory_order_relaxed)) return;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        start = get_event();
        stop = get_event();
    }
    if (start) (void)hipEventRecord((hipEvent_t)start, s);
}

ProfScope::~ProfScope() {
    if (!start || !stop) return;
    (void)hipEventRecord((hipEvent_t)stop, stream);
    std::lock_guard<std::mutex> lk(g_mu);
    g_pending.push_back(Pending{name, (hipEvent_t)start, (hipEvent_t)stop});
}

// RF_OK when the calling thread's current HIP device is a gfx950 (MI355X), RF_ENODEVICE otherwise
// (no device, or another architecture: the library carries gfx950 code objects only).  The verdict
// is cached per device ordinal.
int require_device() {
    static std::atomic<signed char> cache[64];  // 0 unknown, 1 gfx950, -1 anything else
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) {
        (void)hipGetLastError();
        return RF_ENODEVICE;
    }
    if (dev < 64) {
        const signed char c = cache[dev].load(std::memory_order_relaxed);
        if (c) return c > 0 ? RF_OK : RF_ENODEVICE;
    }
    hipDeviceProp_t prop;
    bool ok = hipGetDeviceProperties(&prop, dev) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0;
    if (!ok) (void)hipGetLastError();
    if (dev < 64) cache[dev].store(ok ? 1 : -1, std::memory_order_relaxed);
    return ok ? RF_OK : RF_ENODEVICE;
}

namespace {
// words [0, head) one by one up to the first 16-byte boundary, then 16-byte stores, then the tail words
__global__ __launch_bounds__(256) void zero_kernel(uint32_t *p, size_t head, size_t quads, size_t words) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    if (i < head) p[i] = 0u;
    uint4 *q = (uint4 *)(p + head);
    for (size_t k = i; k < quads; k += stride) q[k] = make_uint4(0u, 0u, 0u, 0u);
    const size_t t0 = head + quads * 4;
    if (t0 + i < words) p[t0 + i] = 0u;  // fewer than 4 words
}
}  // namespace

int zero_async(void *p, size_t bytes, hipStream_t s) {
    if (bytes == 0) return RF_OK;
    if (!p || (bytes & 3) || ((uintptr_t)p & 3)) return RF_EINVAL;
    const size_t words = bytes / 4;
    size_t head = ((16 - ((uintptr_t)p & 15)) & 15) / 4;
    if (head > words) head = words;
    const size_t quads = (words - head) / 4;
    size_t blocks = (quads + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 8192) blocks = 8192;  // grid-stride beyond 32 MB
    RF_LAUNCH("zero_fill", zero_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (uint32_t *)p, head, quads, words);
    return RF_OK;
}

}  // namespace rf

extern "C" {

const char *rf_version(void) { return "rfops-mi355x 0.1 (gfx950)"; }

const char *rf_status_string(int status) {
    switch (status) {
        case RF_OK: return "ok";
        case RF_EINVAL: return "invalid argument";
        case RF_EWORKSPACE: return "workspace too small";
        case RF_ENODEVICE: return "no usable gfx950 device";
        default: break;
    }
    if (status > 0) return hipGetErrorString((hipError_t)status);
    return "unknown status";
}

void rf_profile_enable(int on) { g_enabled.store(on != 0, std::memory_order_relaxed); }

int rf_device_check(void) { return rf::require_device(); }

// Diagnostic: the one place this library issues a hipMemsetAsync -- so that a host can capture it into a
// graph and see whether the runtime replays memset nodes correctly (rfnet_amd/_host.py:graph_replay_ok).
int rf_probe_memset_async(void *p, size_t bytes, rf_stream_t stream) {
    if (!p || bytes == 0) return RF_EINVAL;
    if (int dv = rf::require_device()) return dv;
    RF_HIP(hipMemsetAsync(p, 0, bytes, (hipStream_t)stream));
    return RF_OK;
}

int rf_profile_collect(const char **names, double *ms, long *launches, int cap) {
    std::lock_guard<std::mutex> lk(g_mu);
    int count = 0;
    for (auto &p : g_pending) {
        float t = 0.f;
        if (hipEventSynchronize(p.stop) == hipSuccess) (void)hipEventElapsedTime(&t, p.start, p.stop);
        int k = 0;
        for (; k < count; k++)
            if (strcmp(names[k], p.name) == 0) break;
        if (k == count) {
            if (count >= cap) {
                g_pool.push_back(p.start);
                g_pool.push_back(p.stop);
                continue;
            }
            names[k] = p.name;
            ms[k] = 0.0;
            launches[k] = 0;
            count++;
        }
        ms[k] += t;
        launches[k] += 1;
        g_pool.push_back(p.start);
        g_pool.push_back(p.stop);
    }
    g_pending.clear();
    return count;
}

}  // extern "C"
