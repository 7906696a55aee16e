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

extern "C" int32_t arx_device_cu_count(void) { return arx_device_cus(); }

// A stream whose queue may only use the CUs of `cu_mask` (hipExtStreamCreateWithCUMask).  The driver deals consecutive mask bits round-robin
// to the XCDs (bit i -> XCD i % 8), so a contiguous run of 8 n bits is n CUs on each of the 8 XCDs.
extern "C" int32_t arx_stream_create_cu_mask(const uint32_t* cu_mask, int32_t n_words, void** out) {
    ARX_REQUIRE(cu_mask && out && n_words > 0 && n_words <= 32, "bad args");
    bool any = false;
    for (int i = 0; i < n_words; ++i) any = any || cu_mask[i] != 0;
    ARX_REQUIRE(any, "empty CU mask");
    hipStream_t st = nullptr;
    ARX_HIP_CHECK(hipExtStreamCreateWithCUMask(&st, (uint32_t)n_words, cu_mask));
    *out = (void*)st;
    return ARX_OK;
}
extern "C" int32_t arx_stream_destroy(void* stream) {
    ARX_REQUIRE(stream, "null stream");
    ARX_HIP_CHECK(hipStreamDestroy((hipStream_t)stream));
    return ARX_OK;
}

// Which compute units does a stream's queue actually use?  Every block records (XCC_ID << 16) | (HW_ID & 0xffff) — HW_ID bits 8-11 = CU,
// 12 = shader array, 13-15 = shader engine — and then idles for `spin_cycles`, so that a grid of a few thousand blocks visits every CU the
// queue may use.  The check behind arx_stream_create_cu_mask's claim about the mask's bit order (tests, tools/cu_mask_probe.py).
__global__ __launch_bounds__(64) void cu_census_kernel(uint32_t* __restrict__ out, int spin_cycles) {
    const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20), hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while ((long long)(__builtin_amdgcn_s_memtime() - t0) < (long long)spin_cycles) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc << 16) | (hw & 0xffffu);
}
extern "C" int32_t arx_debug_cu_census(uint32_t* out, int32_t n_blocks, int32_t spin_cycles, void* stream) {
    ARX_REQUIRE(out && n_blocks > 0 && spin_cycles >= 0 && spin_cycles <= 10000000, "bad args");
    cu_census_kernel<<<n_blocks, 64, 0, (hipStream_t)stream>>>(out, spin_cycles);
    ARX_HIP_CHECK(hipGetLastError());
    return ARX_OK;
}

// ---- profiling: event pairs per kernel class, recorded on the launch stream -----------------------
struct ProfPair { hipEvent_t a, b; };
static bool g_prof_on = false;
static uint32_t g_prof_mask = 0xffffffffu;      // kernel classes that record events while profiling is on
static std::vector<ProfPair> g_pairs[ARX_K_CLASSES];
static std::vector<ProfPair> g_free;
static std::mutex g_prof_mu;                    // the facility is process-wide (the one piece of shared state in the library): searches from
                                                // several host threads may record into it at once

bool arx_prof_on() { return g_prof_on; }

int arx_prof_begin(int cls, hipStream_t st) {
    if (!g_prof_on || !((g_prof_mask >> cls) & 1u)) return -1;
    std::lock_guard<std::mutex> lk(g_prof_mu);
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
    std::lock_guard<std::mutex> lk(g_prof_mu);
    (void)hipEventRecord(g_pairs[cls][token].b, st);
}

extern "C" int32_t arx_prof_enable(int32_t on) { g_prof_on = on != 0; return ARX_OK; }
extern "C" int32_t arx_prof_classes(uint32_t mask) { g_prof_mask = mask; return ARX_OK; }
extern "C" int32_t arx_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int c = 0; c < ARX_K_CLASSES; ++c) {
        for (auto& p : g_pairs[c]) g_free.push_back(p);
        g_pairs[c].clear();
    }
    return ARX_OK;
}
extern "C" int32_t arx_prof_read(int32_t cls, float* total_ms, int32_t* launches) {
    ARX_REQUIRE(cls >= 0 && cls < ARX_K_CLASSES && total_ms && launches, "bad kernel class");
    std::lock_guard<std::mutex> lk(g_prof_mu);
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
