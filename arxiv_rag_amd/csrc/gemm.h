// C[m][n] = sum_k A[m][k] * W[n][k]   (both operands K-contiguous: activations x nn.Linear weight,
// or queries x corpus rows) on gfx950 MFMA 16x16x32, LDS-staged with direct global->LDS loads.
//
// Layout / schedule:
//   block tile BM x BN x 64, WM x WN waves, wave tile (BM/WM) x (BN/WN);
//   LDS tile = [rows][64 elem] (128-B rows), 16-B chunk index XOR ((row>>1)&7): the ds_read_b128
//   fragment reads of one 16-lane group then hit 16 distinct 16-B slots of the 256-B bank row;
//   global_load_lds writes LDS lane-linearly, so the XOR is applied to the per-lane SOURCE address
//   (the same involution on the read side);
//   2 LDS stages: stage k+1 is issued before the MFMAs of stage k, one barrier per 64-deep step;
//   MFMA operand order (W fragment as A-operand, activation fragment as B-operand) puts 4
//   consecutive n in one lane's accumulator registers -> 8-byte bf16 stores along a C row.
#pragma once
#include "arx_common.h"

template <typename T> struct Mfma;
template <> struct Mfma<bf16_t> {
    using vec = bf16x8;
    static __device__ __forceinline__ f32x4 mma(vec a, vec b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mfma<f16_t> {
    using vec = f16x8;
    static __device__ __forceinline__ f32x4 mma(vec a, vec b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

// Stage a [ROWS x 64-element] tile (rows row0.., columns k0..k0+63 of G[ld]) into `tile` (LDS).
template <int ROWS, int NT, bool GLDS>
__device__ __forceinline__ void stage_issue(const uint16_t* __restrict__ G, int64_t ld, int row0, int row_max,
                                            int k0, char* tile, int tid, u32x4 (&regs)[ROWS * 8 / NT]) {
    constexpr int IT = ROWS * 8 / NT;
    static_assert(ROWS * 8 % NT == 0, "tile rows must fill whole passes");
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int cid = it * NT + tid;
        const int row = cid >> 3, pc = cid & 7;
        const int c = pc ^ ((row >> 1) & 7);
        int grow = row0 + row;
        grow = grow < row_max ? grow : row_max;
        const uint16_t* src = G + (int64_t)grow * ld + k0 + c * 8;
        if constexpr (GLDS) {
            char* dst = tile + (it * NT + (tid & ~63)) * 16;      // wave-uniform base; lane l lands at +16*l
            __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)dst, 16, 0, 0);
        } else {
            regs[it] = *reinterpret_cast<const u32x4*>(src);
        }
    }
}
template <int ROWS, int NT>
__device__ __forceinline__ void stage_commit(char* tile, int tid, const u32x4 (&regs)[ROWS * 8 / NT]) {
#pragma unroll
    for (int it = 0; it < ROWS * 8 / NT; ++it)
        *reinterpret_cast<u32x4*>(tile + (it * NT + tid) * 16) = regs[it];
}

// Main loop: fills acc[NI][MI] for the block tile at (m0, n0).
//   acc[j][i][r]:  n = n0 + wn*TN + j*16 + (lane>>4)*4 + r ,  m = m0 + wm*TM + i*16 + (lane&15)
template <typename T, int BM, int BN, int WM, int WN, bool GLDS>
struct GemmMainloop {
    static constexpr int NT = WM * WN * 64;
    static constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 16, NI = TN / 16;
    static constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE_BYTES = A_BYTES + W_BYTES;
    static constexpr int SMEM_BYTES = 2 * STAGE_BYTES;
    using vec = typename Mfma<T>::vec;

    static __device__ __forceinline__ void run(const T* __restrict__ A, int64_t lda, int M,
                                               const T* __restrict__ W, int64_t ldw, int N, int K,
                                               int m0, int n0, char* smem, f32x4 (&acc)[NI][MI]) {
        const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
        const int wm = wid / WN, wn = wid % WN;
        const uint16_t* Ag = reinterpret_cast<const uint16_t*>(A);
        const uint16_t* Wg = reinterpret_cast<const uint16_t*>(W);
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int i = 0; i < MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

        // per-lane fragment byte offsets inside a tile (row part + swizzled chunk part), ks = 0/1
        const int frow = lane & 15, fq = lane >> 4, sw = (lane >> 1) & 7;
        const int off0 = frow * 128 + (((0 + fq) ^ sw) << 4);
        const int off1 = frow * 128 + (((4 + fq) ^ sw) << 4);
        const int a_base = wm * TM * 128, w_base = wn * TN * 128;

        u32x4 ra[BM * 8 / NT], rw[BN * 8 / NT];
        const int nk = K >> 6;
        stage_issue<BM, NT, GLDS>(Ag, lda, m0, M - 1, 0, smem, tid, ra);
        stage_issue<BN, NT, GLDS>(Wg, ldw, n0, N - 1, 0, smem + A_BYTES, tid, rw);
        if constexpr (!GLDS) {
            stage_commit<BM, NT>(smem, tid, ra);
            stage_commit<BN, NT>(smem + A_BYTES, tid, rw);
        }
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            char* cur = smem + (kt & 1) * STAGE_BYTES;
            char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
            if (kt + 1 < nk) {
                stage_issue<BM, NT, GLDS>(Ag, lda, m0, M - 1, (kt + 1) << 6, nxt, tid, ra);
                stage_issue<BN, NT, GLDS>(Wg, ldw, n0, N - 1, (kt + 1) << 6, nxt + A_BYTES, tid, rw);
            }
            const char* As = cur + a_base;
            const char* Ws = cur + A_BYTES + w_base;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int off = ks ? off1 : off0;
                vec af[MI], wf[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const vec*>(As + i * 2048 + off);
#pragma unroll
                for (int j = 0; j < NI; ++j) wf[j] = *reinterpret_cast<const vec*>(Ws + j * 2048 + off);
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int i = 0; i < MI; ++i) acc[j][i] = Mfma<T>::mma(wf[j], af[i], acc[j][i]);
            }
            if constexpr (!GLDS) {
                if (kt + 1 < nk) {
                    stage_commit<BM, NT>(nxt, tid, ra);
                    stage_commit<BN, NT>(nxt + A_BYTES, tid, rw);
                }
            }
            __syncthreads();      // drains the in-flight global->LDS loads and fences LDS reuse
        }
    }
};

// ---- epilogues --------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

enum { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_BIAS_RESID = 2 };

struct EpiParams {
    uint16_t* out;          // bf16 [M, ldc]
    int64_t ldc;
    const float* bias;      // [N]
    const uint16_t* resid;  // bf16 [M, ldr] or null
    int64_t ldr;
};

template <int MODE, int NI, int MI>
__device__ __forceinline__ void epilogue_store(const f32x4 (&acc)[NI][MI], const EpiParams& p, int m_base, int n_base,
                                               int lane, int M, int N) {
    const int mq = lane & 15, nq = (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int n = n_base + j * 16 + nq;
        if (n >= N) continue;
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = m_base + i * 16 + mq;
            if (m >= M) continue;
            float v0 = acc[j][i][0] + b[0], v1 = acc[j][i][1] + b[1];
            float v2 = acc[j][i][2] + b[2], v3 = acc[j][i][3] + b[3];
            if constexpr (MODE == EPI_BIAS_GELU) {
                v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3);
            }
            if constexpr (MODE == EPI_BIAS_RESID) {
                const u32x2 r = *reinterpret_cast<const u32x2*>(p.resid + (int64_t)m * p.ldr + n);
                float r0, r1, r2, r3;
                unpack_bf16x2(r[0], r0, r1); unpack_bf16x2(r[1], r2, r3);
                v0 += r0; v1 += r1; v2 += r2; v3 += r3;
            }
            u32x2 o;
            o[0] = pack_bf16x2(v0, v1); o[1] = pack_bf16x2(v2, v3);
            *reinterpret_cast<u32x2*>(p.out + (int64_t)m * p.ldc + n) = o;
        }
    }
}

template <int BM, int BN, int WM, int WN, bool GLDS, int MODE>
__global__ __launch_bounds__(WM * WN * 64) void gemm_bf16_kernel(const bf16_t* __restrict__ A, int64_t lda,
                                                                   const bf16_t* __restrict__ W, int64_t ldw,
                                                                   int M, int N, int K, int tiles_m, int tiles_n,
                                                                   EpiParams ep) {
    using ML = GemmMainloop<bf16_t, BM, BN, WM, WN, GLDS>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tile_m = t / tiles_n, tile_n = t % tiles_n;       // n fastest: blocks sharing A rows are neighbours
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    f32x4 acc[ML::NI][ML::MI];
    ML::run(A, lda, M, W, ldw, N, K, m0, n0, smem, acc);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    epilogue_store<MODE, ML::NI, ML::MI>(acc, ep, m0 + (wid / WN) * ML::TM, n0 + (wid % WN) * ML::TN, lane, M, N);
}
