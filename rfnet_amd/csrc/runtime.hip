// runtime.hip -- version / status strings and the per-kernel hipEvent profiler behind
// rf_profile_enable() / rf_profile_collect().
#include <mutex>
#include <string.h>
#include <vector>

#include "common.hpp"

namespace {

struct Pending {
    const char *name;
    hipEvent_t start, stop;
};

std::mutex g_mu;
bool g_enabled = false;
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

ProfScope::ProfScope(const char *n, hipStream_t s) : name(n), stream(s), slot(-1) {
    if (!g_enabled) return;
    std::lock_guard<std::mutex> lk(g_mu);
    Pending p{n, get_event(), get_event()};
    (void)hipEventRecord(p.start, s);
    slot = (int)g_pending.size();
    g_pending.push_back(p);
}

ProfScope::~ProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_mu);
    (void)hipEventRecord(g_pending[slot].stop, stream);
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

void rf_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_enabled = on != 0;
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
