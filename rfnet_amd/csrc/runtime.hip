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
