// Shared device/host helpers for libarx_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/arx.h"

typedef __bf16 bf16_t;
typedef _Float16 f16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

void arx_set_error(const char* fmt, ...);
// per-kernel-class event timing (runtime.hip); token < 0 = profiling off
int arx_prof_begin(int cls, hipStream_t st);
void arx_prof_end(int cls, int token, hipStream_t st);
// hipFuncAttributeMaxDynamicSharedMemorySize, raised at most once per (device, kernel, size): the attribute belongs to the
// device's code object, so a second handle on another GPU of the same process must set it again (runtime.hip; thread-safe)
hipError_t arx_func_smem(const void* kernel, int bytes);
// compute units of the CURRENT device (cached per device)
int arx_device_cus();
struct ProfScope {
    int cls, tok; hipStream_t st;
    ProfScope(int c, hipStream_t s) : cls(c), tok(arx_prof_begin(c, s)), st(s) {}
    ~ProfScope() { arx_prof_end(cls, tok, st); }
};

#define ARX_HIP_CHECK(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            arx_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return ARX_ERR_HIP;                                                          \
        }                                                                                \
    } while (0)

#define ARX_REQUIRE(cond, ...)                \
    do {                                      \
        if (!(cond)) {                        \
            arx_set_error(__VA_ARGS__);       \
            return ARX_ERR_ARG;               \
        }                                     \
    } while (0)

static inline int64_t round_up64(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- device helpers ---------------------------------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    // vector convert: ONE v_cvt_pk_bf16_f32 (RNE, NaN-preserving) writing both halves of the dword
    const f32x2 f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2));
}

__device__ __forceinline__ void unpack_bf16x2(uint32_t v, float& lo, float& hi) {
    lo = __uint_as_float(v << 16);
    hi = __uint_as_float(v & 0xffff0000u);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// XCD-aware bijective remap of a 1-D block id: blocks that share an XCD (bid % 8 equal) get a
// contiguous chunk of the logical tile order, so neighbouring tiles hit the same private L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
