// Dev micro-benchmark: is the ENERGY per FLOP of the matrix pipe the same for the two bf16 MFMA shapes?  A register-resident loop of
// independent MFMAs (no LDS, no memory) on random operands, every CU busy with 2 waves per SIMD, run for a few seconds per shape while the
// caller samples the shader clock and the package power: under the 1.4-kW cap the sustained TFLOP/s IS the energy figure.
//   mode 0: v_mfma_f32_16x16x32_bf16, 32 independent accumulator blocks (128 acc registers), 8 A + 4 B fragments  (the GEMM's wave tile)
//   mode 1: v_mfma_f32_32x32x16_bf16,  8 independent accumulator blocks (128 acc registers), 4 A + 2 B fragments  (same 128 x 64 wave tile)
// usage: mfma_power <mode> <seconds> <zero_operands 0|1>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(512) void spin(const bf16x8* __restrict__ src, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 a[8], b[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = src[(i * 64 + lane + threadIdx.x) & 4095];
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = src[(2048 + i * 64 + lane + blockIdx.x) & 4095];
    float keep = 0.f;
    if constexpr (MODE == 0) {
        f32x4 acc[4][8];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[j][i], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) keep += acc[j][i][0] + acc[j][i][3];
    } else {
        f32x16 acc[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
            // the same 128 x 64 x 32 of work per iteration: 8 blocks x two 16-deep steps (second step on the other operand halves)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j * 2 + ks], a[i * 2 + ks], acc[j][i], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) keep += acc[j][i][0] + acc[j][i][15];
    }
    if (keep == 12345.678f) out[0] = keep;
}

// mode 2: the 16x16x32 loop with its 12 fragments RE-READ from LDS every iteration (12 ds_read_b128 per 32 MFMAs: the GEMM's ratio, 24 reads per
//         64-deep k-tile of a 128 x 64 wave tile); LDS filled once with the random operands
// mode 3: mode 2 + the operand stream: 4 x 1-KB LDS-DMA pieces per wave and iteration from an L2-resident 8-MB buffer (a 256 x 256 block tile
//         stages 64 KB per k-tile = 8 KB per wave per 64 MFMAs), behind a counted vmcnt
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) void gbl_void_t;
template <int MODE>
__global__ __launch_bounds__(512) void spin_lds(const bf16x8* __restrict__ src, const char* __restrict__ stream, float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    bf16x8* frag = reinterpret_cast<bf16x8*>(smem);                 // [12][512] fragments: one 16-B slot per thread and fragment index
    for (int i = 0; i < 12; ++i) frag[i * 512 + threadIdx.x] = src[(i * 64 + lane + threadIdx.x) & 4095];
    __syncthreads();
    char* dma = smem + 12 * 512 * 16 + wid * 4096;                  // 4 KB per wave of DMA landing zone
    f32x4 acc[4][8];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const size_t gbase = ((size_t)(blockIdx.x * 8 + wid) * 4096) % (8u << 20);        // this wave's 1-KB pieces start here (8-MB ring + 1 MB of slack)
    for (int it = 0; it < iters; ++it) {
        bf16x8 a[8], b[4];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = frag[i * 512 + ((threadIdx.x + it) & 511)];
#pragma unroll
        for (int i = 0; i < 4; ++i) b[i] = frag[(8 + i) * 512 + ((threadIdx.x + it) & 511)];
        if constexpr (MODE == 3) {
#pragma unroll
            for (int pce = 0; pce < 4; ++pce)
                __builtin_amdgcn_global_load_lds((gbl_void_t*)(stream + (gbase + (size_t)(it * 4 + pce) * 65536) % (8u << 20) + lane * 16),
                                                 (lds_void_t*)(dma + pce * 1024), 16, 0, 0);       // < 8 MB + 1 KB: inside the 9-MB allocation
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[j][i], 0, 0, 0);
        if constexpr (MODE == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    float keep = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) keep += acc[j][i][0] + acc[j][i][3];
    if (keep == 12345.678f) out[0] = keep + dma[0];
}

// mode 6: mode 3 with the B fragments NOT through LDS: each lane loads its 4 B fragments (16 B each) straight from the L2-resident ring
//         (W is an L2-resident operand in the encoder), A fragments from LDS (8 reads), the DMA stream carries the A half only (2 pieces)
__global__ __launch_bounds__(512) void spin_bdirect(const bf16x8* __restrict__ src, const char* __restrict__ stream, float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    bf16x8* frag = reinterpret_cast<bf16x8*>(smem);
    for (int i = 0; i < 8; ++i) frag[i * 512 + threadIdx.x] = src[(i * 64 + lane + threadIdx.x) & 4095];
    __syncthreads();
    char* dma = smem + 12 * 512 * 16 + wid * 4096;
    f32x4 acc[4][8];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const size_t gbase = ((size_t)(blockIdx.x * 8 + wid) * 4096) % (8u << 20);
    const size_t bbase = ((size_t)(wid & 3) * 65536 + lane * 16);            // waves of one wave column read the same B lines
    bf16x8 b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = *reinterpret_cast<const bf16x8*>(stream + (bbase + i * 1024) % (8u << 20));
    for (int it = 0; it < iters; ++it) {
        bf16x8 a[8], bn[4];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = frag[i * 512 + ((threadIdx.x + it) & 511)];
#pragma unroll
        for (int i = 0; i < 4; ++i)                                           // next iteration's B fragments, in flight under this iteration's MFMAs
            bn[i] = *reinterpret_cast<const bf16x8*>(stream + (bbase + (size_t)(it + 1) * 4096 + i * 1024) % (8u << 20));
#pragma unroll
        for (int pce = 0; pce < 2; ++pce)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(stream + (gbase + (size_t)(it * 2 + pce) * 65536) % (8u << 20) + lane * 16),
                                             (lds_void_t*)(dma + pce * 1024), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[j][i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) b[i] = bn[i];
    }
    float keep = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) keep += acc[j][i][0] + acc[j][i][3];
    if (keep == 12345.678f) out[0] = keep + dma[0];
}

// mode 4 / 5: the OTHER geometry — ONE wave per SIMD with a 128 x 128 wave tile (64 accumulator blocks = 256 accumulator registers; 256-thread
// blocks, one per CU): 16 fragment reads per 64 MFMAs (a third of mode 2's reads per FLOP); mode 5 adds the operand stream (a 256 x 256 block
// tile still stages 64 KB per k-tile: 16 KB = 16 pieces per wave per 128 MFMAs, i.e. 8 per iteration of 64)
template <int MODE>
__global__ __launch_bounds__(256) void spin_big(const bf16x8* __restrict__ src, const char* __restrict__ stream, float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    bf16x8* frag = reinterpret_cast<bf16x8*>(smem);                 // [16][256]
    for (int i = 0; i < 16; ++i) frag[i * 256 + threadIdx.x] = src[(i * 64 + lane + threadIdx.x) & 4095];
    __syncthreads();
    char* dma = smem + 16 * 256 * 16 + wid * 8192;
    f32x4 acc[8][8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const size_t gbase = ((size_t)(blockIdx.x * 4 + wid) * 8192) % (8u << 20);
    for (int it = 0; it < iters; ++it) {
        bf16x8 a[8], b[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = frag[i * 256 + ((threadIdx.x + it) & 255)];
#pragma unroll
        for (int i = 0; i < 8; ++i) b[i] = frag[(8 + i) * 256 + ((threadIdx.x + it) & 255)];
        if constexpr (MODE == 5) {
#pragma unroll
            for (int pce = 0; pce < 8; ++pce)
                __builtin_amdgcn_global_load_lds((gbl_void_t*)(stream + (gbase + (size_t)(it * 8 + pce) * 65536) % (8u << 20) + lane * 16),
                                                 (lds_void_t*)(dma + pce * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[j][i], 0, 0, 0);
        if constexpr (MODE == 5) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    float keep = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) keep += acc[j][i][0] + acc[j][i][3];
    if (keep == 12345.678f) out[0] = keep + dma[0];
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const double secs = argc > 2 ? atof(argv[2]) : 3.0;
    const int zero = argc > 3 ? atoi(argv[3]) : 0;
    bf16x8* src; float* out;
    hipMalloc(&src, 4096 * 16); hipMalloc(&out, 4);
    unsigned short* h = (unsigned short*)malloc(4096 * 16);
    srand(1);
    for (int i = 0; i < 4096 * 8; ++i) {                      // bf16 N(0,1)-ish: random sign / exponent around 1 / mantissa
        const unsigned short s = (rand() & 1) << 15, e = (unsigned short)(124 + rand() % 6) << 7, m = rand() & 127;
        h[i] = zero ? 0 : (unsigned short)(s | e | m);
    }
    hipMemcpy(src, h, 4096 * 16, hipMemcpyHostToDevice);
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int iters = 20000;                                  // 32 (or 16) MFMAs per iteration
    const double flop_per_launch = (mode == 4 || mode == 5) ? (double)cus * 4 /*waves*/ * iters * 2.0 * 128 * 128 * 32
                                             : (double)cus * 8 /*waves*/ * iters * 2.0 * 128 * 64 * 32;
    char* stream = nullptr;
    hipMalloc(&stream, (8u << 20) + (1u << 20));
    hipMemset(stream, 0x3c, (8u << 20) + (1u << 20));
    {                                                          // random bytes in the stream too
        unsigned* hs = (unsigned*)malloc(8u << 20);
        for (size_t i = 0; i < (8u << 20) / 4; ++i) hs[i] = (unsigned)rand() * 2654435761u;
        hipMemcpy(stream, hs, 8u << 20, hipMemcpyHostToDevice);
        if (zero) hipMemset(stream, 0, 8u << 20);
        free(hs);
    }
    const int smem_bytes = 12 * 512 * 16 + 8 * 4096;
    const int smem_big = 16 * 256 * 16 + 4 * 8192;
    auto launch = [&]() {
        if (mode == 0) spin<0><<<cus, 512>>>(src, out, iters);
        else if (mode == 1) spin<1><<<cus, 512>>>(src, out, iters);
        else if (mode == 2) spin_lds<2><<<cus, 512, smem_bytes>>>(src, stream, out, iters);
        else if (mode == 3) spin_lds<3><<<cus, 512, smem_bytes>>>(src, stream, out, iters);
        else if (mode == 6) spin_bdirect<<<cus, 512, smem_bytes>>>(src, stream, out, iters);
        else if (mode == 4) spin_big<4><<<cus, 256, smem_big>>>(src, stream, out, iters);
        else spin_big<5><<<cus, 256, smem_big>>>(src, stream, out, iters);
    };
    hipFuncSetAttribute((const void*)spin_bdirect, hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
    hipFuncSetAttribute((const void*)spin_big<4>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_big);
    hipFuncSetAttribute((const void*)spin_big<5>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_big);
    if (mode >= 2) {
        hipFuncSetAttribute((const void*)spin_lds<2>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
        hipFuncSetAttribute((const void*)spin_lds<3>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
    }
    launch(); hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    int n = 0;
    double el = 0;
    while (el < secs) {
        for (int k = 0; k < 10; ++k) launch();
        hipDeviceSynchronize();
        n += 10;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    printf("{\"mode\": \"%s\", \"zero_operands\": %d, \"launches\": %d, \"seconds\": %.3f, \"tflops\": %.1f}\n",
           mode == 0 ? "16x16x32" : mode == 1 ? "32x32x16" : mode == 2 ? "16x16x32 + LDS fragment reads" : mode == 3 ? "16x16x32 + LDS reads + L2->LDS DMA stream" :
           mode == 4 ? "128x128 wave tile, 1 wave/SIMD: 16x16x32 + LDS fragment reads" : mode == 5 ? "128x128 wave tile, 1 wave/SIMD: + L2->LDS DMA stream" :
           "16x16x32: A fragments from LDS, B fragments straight from L2, A-only DMA stream", zero, n, el,
           flop_per_launch * n / el / 1e12);
    return 0;
}
