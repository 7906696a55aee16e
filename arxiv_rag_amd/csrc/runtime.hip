// Library-wide plumbing: error string, version, per-kernel-class hipEvent timing (include/arx.h).
#include <stdarg.h>

#include <map>
#include <mutex>
#include <vector>

#include "arx_common.h"

static thread_local char g_err[1024] = "";
void arx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* arx_last_error(void) { return g_err; }
extern "C" int32_t arx_version(void) { return ARX_VERSION; }

// ---- per-device launch attributes ------------------------------------------------------------------------
static std::mutex g_attr_mu;
static std::map<std::pair<int, const void*>, int> g_attr_smem;
static std::map<int, int> g_dev_cus;

hipError_t arx_func_smem(const void* kernel, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(g_attr_mu);
    int& have = g_attr_smem[std::make_pair(dev, kernel)];
    if (have >= bytes) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) have = bytes;
    return e;
}

int arx_device_cus() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    std::lock_guard<std::mutex> lk(g_attr_mu);
    int& n = g_dev_cus[dev];
    if (n <= 0 && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
    return n;
}

// ---- profiling: event pairs per kernel class, recorded on the launch stream -----------------------
struct ProfPair { hipEvent_t a, b; };
static bool g_prof_on = false;
static uint32_t g_prof_mask = 0xffffffffu;      // kernel classes that record events while profiling is on
static std::vector<ProfPair> g_pairs[ARX_K_CLASSES];
static std::vector<ProfPair> g_free;

bool arx_prof_on() { return g_prof_on; }

int arx_prof_begin(int cls, hipStream_t st) {
    if (!g_prof_on || !((g_prof_mask >> cls) & 1u)) return -1;
    ProfPair p;
    if (!g_free.empty()) { p = g_free.back(); g_free.pop_back(); }
    else {
        if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return -1;
    }
    if (hipEventRecord(p.a, st) != hipSuccess) { g_free.push_back(p); return -1; }
    g_pairs[cls].push_back(p);
    return (int)g_pairs[cls].size() - 1;
}
void arx_prof_end(int cls, int token, hipStream_t st) {
    if (token < 0) return;
    (void)hipEventRecord(g_pairs[cls][token].b, st);
}

extern "C" int32_t arx_prof_enable(int32_t on) { g_prof_on = on != 0; return ARX_OK; }
extern "C" int32_t arx_prof_classes(uint32_t mask) { g_prof_mask = mask; return ARX_OK; }
extern "C" int32_t arx_prof_reset(void) {
    for (int c = 0; c < ARX_K_CLASSES; ++c) {
        for (auto& p : g_pairs[c]) g_free.push_back(p);
        g_pairs[c].clear();
    }
    return ARX_OK;
}
extern "C" int32_t arx_prof_read(int32_t cls, float* total_ms, int32_t* launches) {
    ARX_REQUIRE(cls >= 0 && cls < ARX_K_CLASSES && total_ms && launches, "bad kernel class");
    float tot = 0.f;
    for (auto& p : g_pairs[cls]) {
        ARX_HIP_CHECK(hipEventSynchronize(p.b));
        float ms = 0.f;
        ARX_HIP_CHECK(hipEventElapsedTime(&ms, p.a, p.b));
        tot += ms;
    }
    *total_ms = tot;
    *launches = (int32_t)g_pairs[cls].size();
    return ARX_OK;
}
