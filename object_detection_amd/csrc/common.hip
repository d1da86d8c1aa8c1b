// Error reporting, version and the per-kernel hipEvent timer of the detbox C ABI.
#include "detbox_internal.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace detbox {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char *what) {
    set_error("HIP error %d (%s) in %s", static_cast<int>(e), hipGetErrorString(e), what);
    return DETBOX_ERR_HIP;
}

// ---- profiler ---------------------------------------------------------------------------
struct Sample { hipEvent_t a, b; };
static std::mutex g_pmu;
static bool g_prof_on = false;
static std::map<std::string, std::vector<Sample>> g_samples;
static std::map<std::string, hipEvent_t> g_open;   // begin event waiting for its end

void prof_begin(const char *name, hipStream_t s) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_pmu);
    hipEvent_t ev;
    if (hipEventCreate(&ev) != hipSuccess) return;
    (void)hipEventRecord(ev, s);
    g_open[name] = ev;
}

void prof_end(const char *name, hipStream_t s) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_pmu);
    auto it = g_open.find(name);
    if (it == g_open.end()) return;
    hipEvent_t ev;
    if (hipEventCreate(&ev) != hipSuccess) return;
    (void)hipEventRecord(ev, s);
    g_samples[name].push_back({it->second, ev});
    g_open.erase(it);
}

}  // namespace detbox

using namespace detbox;

extern "C" {

int detbox_version(void) { return 100; }

const char *detbox_last_error(void) { return g_err; }

int detbox_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_pmu);
    g_prof_on = on != 0;
    return DETBOX_OK;
}

int detbox_profile_reset(void) {
    std::lock_guard<std::mutex> lk(g_pmu);
    for (auto &kv : g_samples)
        for (auto &s : kv.second) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
    g_samples.clear();
    for (auto &kv : g_open) (void)hipEventDestroy(kv.second);
    g_open.clear();
    return DETBOX_OK;
}

int detbox_profile_read(const char *name, double *total_ms, int64_t *launches) {
    DETBOX_CHECK_ARG(name && total_ms && launches, "null argument");
    std::lock_guard<std::mutex> lk(g_pmu);
    *total_ms = 0.0;
    *launches = 0;
    auto it = g_samples.find(name);
    if (it == g_samples.end()) return DETBOX_OK;
    for (auto &s : it->second) {
        DETBOX_HIP(hipEventSynchronize(s.b));
        float ms = 0.f;
        DETBOX_HIP(hipEventElapsedTime(&ms, s.a, s.b));
        *total_ms += ms;
        *launches += 1;
    }
    return DETBOX_OK;
}

}  // extern "C"
