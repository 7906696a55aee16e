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

// int8 operands seen through the same byte geometry as the 2-byte types: a "row of K elements" of this type is a row of 2 K int8
// values (128-byte k-tiles, 16-byte fragments = 16 consecutive k), multiplied by v_mfma_i32_16x16x64_i8; the f32x4 accumulator
// registers carry the int32 sums bit for bit (zero is zero in both).  Used by the search pre-filter (search.hip).
struct i8pair_t { uint16_t v; };
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <> struct Mfma<i8pair_t> {
    using vec = i32x4;
    static __device__ __forceinline__ f32x4 mma(vec a, vec b, f32x4 c) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, __builtin_bit_cast(i32x4, c), 0, 0, 0));
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
// OPT bits: 1 = waves in the upper half issue the next stage's loads after their first MFMA cluster (stagger),
//   2 = s_setprio(1) around MFMA clusters, 64 = rotate the k-loop start by 2*tile_n (set by the kernel),
//   4096 = register double-buffered fragments (variant 34)
template <typename T, int BM, int BN, int WM, int WN, bool GLDS, int OPT = 0>
struct GemmMainloop {
    static constexpr int NT = WM * WN * 64;
    static constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 16, NI = TN / 16;
    static constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE_BYTES = A_BYTES + W_BYTES;
    static constexpr int SMEM_BYTES = 2 * STAGE_BYTES;
    using vec = typename Mfma<T>::vec;

    static __device__ __forceinline__ void run(const T* __restrict__ A, int64_t lda, int M,
                                               const T* __restrict__ W, int64_t ldw, int N, int K,
                                               int m0, int n0, char* smem, f32x4 (&acc)[NI][MI], int koff = 0) {
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
        // the k-steps may be visited in any rotation (the sum is order-free up to fp32 rounding): blocks that
        // share an A panel start at different k so that they do not all miss on the same lines at once
        koff = koff % nk;
        auto kstep = [&](int kt) { int k = kt + koff; return (k >= nk ? k - nk : k) << 6; };
        stage_issue<BM, NT, GLDS>(Ag, lda, m0, M - 1, kstep(0), smem, tid, ra);
        stage_issue<BN, NT, GLDS>(Wg, ldw, n0, N - 1, kstep(0), smem + A_BYTES, tid, rw);
        if constexpr (!GLDS) {
            stage_commit<BM, NT>(smem, tid, ra);
            stage_commit<BN, NT>(smem + A_BYTES, tid, rw);
        }
        __syncthreads();
        if constexpr ((OPT & 4096) && GLDS) {
            // REGISTER-DOUBLE-BUFFERED FRAGMENTS: every MFMA cluster runs while the ds_reads of the NEXT cluster are in
            // flight (two fragment sets); the second cluster of step kt executes at the start of step kt+1, after the
            // barrier, from fragments read before it.  Uniform control flow for all waves.
            vec af0[MI], wf0[NI], af1[MI], wf1[NI];
            auto rd = [&](const char* stage, int off, vec (&af)[MI], vec (&wf)[NI]) {
#pragma unroll
                for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const vec*>(stage + a_base + i * 2048 + off);
#pragma unroll
                for (int j = 0; j < NI; ++j) wf[j] = *reinterpret_cast<const vec*>(stage + A_BYTES + w_base + j * 2048 + off);
            };
            auto mm = [&](const vec (&af)[MI], const vec (&wf)[NI]) {
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int i = 0; i < MI; ++i) acc[j][i] = Mfma<T>::mma(wf[j], af[i], acc[j][i]);
                __builtin_amdgcn_s_setprio(0);
            };
            for (int kt = 0; kt < nk; ++kt) {
                const char* cur = smem + (kt & 1) * STAGE_BYTES;
                char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
                if (kt + 1 < nk) {
                    stage_issue<BM, NT, GLDS>(Ag, lda, m0, M - 1, kstep(kt + 1), nxt, tid, ra);
                    stage_issue<BN, NT, GLDS>(Wg, ldw, n0, N - 1, kstep(kt + 1), nxt + A_BYTES, tid, rw);
                }
                rd(cur, off0, af0, wf0);                 // first cluster's fragments of this step
                if (kt > 0) mm(af1, wf1);                // second cluster of the previous step
                rd(cur, off1, af1, wf1);                 // second cluster's fragments (consumed after the barrier)
                mm(af0, wf0);
                __syncthreads();
            }
            mm(af1, wf1);
            return;
        }
        for (int kt = 0; kt < nk; ++kt) {
            char* cur = smem + (kt & 1) * STAGE_BYTES;
            char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
            const bool late = (OPT & 1) && (__builtin_amdgcn_readfirstlane(wid) >= (WM * WN) / 2);
            if (kt + 1 < nk && !late) {
                stage_issue<BM, NT, GLDS>(Ag, lda, m0, M - 1, kstep(kt + 1), nxt, tid, ra);
                stage_issue<BN, NT, GLDS>(Wg, ldw, n0, N - 1, kstep(kt + 1), nxt + A_BYTES, tid, rw);
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
                if constexpr (OPT & 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int i = 0; i < MI; ++i) acc[j][i] = Mfma<T>::mma(wf[j], af[i], acc[j][i]);
                if constexpr (OPT & 2) __builtin_amdgcn_s_setprio(0);
                if (ks == 0 && late && kt + 1 < nk) {
                    stage_issue<BM, NT, GLDS>(Ag, lda, m0, M - 1, kstep(kt + 1), nxt, tid, ra);
                    stage_issue<BN, NT, GLDS>(Wg, ldw, n0, N - 1, kstep(kt + 1), nxt + A_BYTES, tid, rw);
                }
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

// LayerNorm is never run as a kernel inside the layer loop: a producer GEMM (modes 5/6) stores the PRE-LN sum y and
// writes per-row partial (sum, sum of squares) of each 64-column slice to a slab (no atomics: a tiny finalize kernel
// adds the N/64 partials in fixed order, so results stay bit-reproducible); the consumer GEMM (modes 3/4) takes y as its A
// operand with gamma folded into the weights (W' = W o gamma, s_n = sum_k W'[n][k], c_n = b_n + sum_k beta_k W[n][k]):
//     LN(y) W^T + b = rstd_m * (y W'^T - mean_m * s_n) + c_n
// and a residual that is itself a LayerNorm output is rebuilt on the fly from y, the row statistics, gamma, beta.
enum { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_BIAS_RESID = 2,
       EPI_LN_BIAS = 3,            // out = rstd*(acc - mean*s) + c
       EPI_LN_BIAS_GELU = 4,       // out = gelu(rstd*(acc - mean*s) + c)
       EPI_RESID_STATS = 5,        // out = acc + bias + resid ; row stats of out
       EPI_LNRESID_STATS = 6 };    // out = acc + bias + LN(resid) ; row stats of out

struct EpiParams {
    uint16_t* out;          // bf16 [M, ldc]
    int64_t ldc;
    const float* bias;      // [N]  (modes 3/4: the folded c vector)
    const uint16_t* resid;  // bf16 [M, ldr] or null
    int64_t ldr;
    // LayerNorm plumbing (modes 3..6)
    const float* a_sum = nullptr;   // [M] row MEAN of the A operand's source rows (modes 3/4)
    const float* a_sq = nullptr;    // [M] row RSTD
    const float* s_vec = nullptr;   // [N] s_n (modes 3/4)
    const float* r_sum = nullptr;   // [M] row MEAN / RSTD of the residual's source rows (mode 6)
    const float* r_sq = nullptr;
    const float* r_gamma = nullptr; // [N] LayerNorm affine of the residual (mode 6)
    const float* r_beta = nullptr;
    float* o_sum = nullptr;         // [N/64][o_ld] partial row sums of the output, one slab row per 64-column slice (modes 5/6)
    float* o_sq = nullptr;          // [N/64][o_ld] partial sums of squares
    int64_t o_ld = 0;
    float inv_h = 0.f, eps = 0.f;
    float* fin_mean = nullptr;      // small-batch path only (gemm_small.h, modes 5/6): the output row's mean / rstd, written in final form
    float* fin_rstd = nullptr;
#ifdef ARX_DEV_VARIANTS
    int dev_store = 0;              // dev A/B: 0 plain stores, 1 non-temporal, 2 sc1 (write-through, line dropped from L2)
    int dev_bw = 0;                 // dev A/B: band width of the tile walk in n-tiles (0 = TileWalk's own choice)
    int dev_stagger = 0, dev_slots = 8;   // dev A/B (persistent kernel): block b starts (b / 8 % dev_slots) * dev_stagger cycles late
#endif
#ifdef ARX_STAMP
    unsigned long long* stamps = nullptr;   // dev build only: [tiles][4] s_memtime at start / loop entry / loop exit / end (wave 0)
#endif
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

// =====================================================================================================
// Ring-pipelined main loop: BK = 32 "half-tiles" (64-B LDS rows) in a ring of SLOTS buffers, global->LDS
// loads kept SLOTS-1 half-tiles ahead behind a COUNTED s_waitcnt vmcnt + raw s_barrier (never vmcnt(0) in
// the loop), so HBM/L2 latency is covered by 2-3 half-steps of MFMAs instead of one.  256x128 tile with
// 4 waves keeps LDS at 72 KB -> two blocks per CU, whose epilogues overlap each other's main loops.
// LDS image: [rows][32 elem]; 16-B chunk p of row r holds logical chunk p ^ G[(r>>2)&3], G = {0,3,2,1}
// (conflict-free for the 16x16x32 fragment ds_read_b128; applied on the per-lane SOURCE address).
template <int ROWS, int NT>
__device__ __forceinline__ void ring_stage(const uint16_t* __restrict__ G, int64_t ld, int row0, int row_max, int k0,
                                           char* tile, int tid) {
    constexpr int IT = ROWS * 4 / NT;
    static_assert(ROWS * 4 % NT == 0, "half-tile rows must fill whole passes");
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int cid = it * NT + tid;
        const int row = cid >> 2, pc = cid & 3;
        const int c = pc ^ ((4 - ((row >> 2) & 3)) & 3);
        int grow = row0 + row;
        grow = grow < row_max ? grow : row_max;
        const uint16_t* src = G + (int64_t)grow * ld + k0 + c * 8;
        char* dst = tile + (it * NT + (tid & ~63)) * 16;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)dst, 16, 0, 0);
    }
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (N == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

template <typename T, int BM, int BN, int WM, int WN, int SLOTS, int OPT = 0>
struct GemmRing {
    static constexpr int NT = WM * WN * 64;
    static constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 16, NI = TN / 16;
    static constexpr int A_BYTES = BM * 64, W_BYTES = BN * 64, SLOT_BYTES = A_BYTES + W_BYTES;
    static constexpr int SMEM_BYTES = SLOTS * SLOT_BYTES;
    static constexpr int LPH = (BM + BN) * 4 / NT;          // global_load_lds per wave per half-tile
    static constexpr int DEPTH = SLOTS - 1;                 // half-tiles in flight ahead of the one being computed
    using vec = typename Mfma<T>::vec;

    static __device__ __forceinline__ void issue(const uint16_t* Ag, int64_t lda, int M, const uint16_t* Wg, int64_t ldw,
                                                 int N, int m0, int n0, int h, char* smem, int tid) {
        char* slot = smem + (h % SLOTS) * SLOT_BYTES;
        ring_stage<BM, NT>(Ag, lda, m0, M - 1, h * 32, slot, tid);
        ring_stage<BN, NT>(Wg, ldw, n0, N - 1, h * 32, slot + A_BYTES, tid);
    }

    static __device__ __forceinline__ void run(const T* __restrict__ A, int64_t lda, int M, const T* __restrict__ W,
                                               int64_t ldw, int N, int K, int m0, int n0, char* smem,
                                               f32x4 (&acc)[NI][MI]) {
        const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
        const int wm = wid / WN, wn = wid % WN;
        const uint16_t* Ag = reinterpret_cast<const uint16_t*>(A);
        const uint16_t* Wg = reinterpret_cast<const uint16_t*>(W);
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int i = 0; i < MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int frow = lane & 15, fq = lane >> 4;
        const int off = frow * 64 + ((fq ^ ((4 - ((lane >> 2) & 3)) & 3)) << 4);
        const int a_base = wm * TM * 64 + off, w_base = A_BYTES + wn * TN * 64 + off;
        const int nh = K >> 5;
        static_assert(DEPTH == 2 || DEPTH == 3, "ring depth");
#pragma unroll
        for (int h = 0; h < DEPTH; ++h)
            if (h < nh) issue(Ag, lda, M, Wg, ldw, N, m0, n0, h, smem, tid);
        for (int h = 0; h < nh; ++h) {
            // half-tile h must have landed: all but the (DEPTH-1) younger half-tiles' loads retired
            const int younger = (nh - 1 - h) < (DEPTH - 1) ? (nh - 1 - h) : (DEPTH - 1);
            if (younger == DEPTH - 1) wait_vmcnt<LPH * (DEPTH - 1)>();
            else if (DEPTH == 3 && younger == 1) wait_vmcnt<LPH>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();          // everyone's part of h landed; everyone done reading slot (h-1)
            __builtin_amdgcn_sched_barrier(0);
            if (h + DEPTH < nh) issue(Ag, lda, M, Wg, ldw, N, m0, n0, h + DEPTH, smem, tid);
            const char* slot = smem + (h % SLOTS) * SLOT_BYTES;
            vec af[MI], wf[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const vec*>(slot + a_base + i * 1024);
#pragma unroll
            for (int j = 0; j < NI; ++j) wf[j] = *reinterpret_cast<const vec*>(slot + w_base + j * 1024);
            if constexpr (OPT & 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int i = 0; i < MI; ++i) acc[j][i] = Mfma<T>::mma(wf[j], af[i], acc[j][i]);
            if constexpr (OPT & 2) __builtin_amdgcn_s_setprio(0);
        }
    }
};

// ---- epilogue v2: lane exchange -> each lane owns 8 consecutive n (16-B stores, 64-B row segments) ------
// fast erf-GELU: Abramowitz-Stegun 7.1.26 (|erf error| < 1.5e-7), one v_rcp + one v_exp per element
__device__ __forceinline__ float gelu_fast(float v) {
    const float x = fabsf(v) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p *= t;
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * x * x);
    const float erf_abs = 1.0f - p * e;                       // erf(|v|/sqrt2)
    return 0.5f * v + 0.5f * fabsf(v) * erf_abs;              // 0.5 v (1 + sign(v) erf(|v|/sqrt2))
}

// Packed erf-GELU, two elements per instruction (v_pk_mul/fma_f32; no transcendental-unit ops, which run at quarter rate):
// erf(t) ~ t * Q(t^2) on |t| <= T, clamped beyond.  Round 2: Q of degree 8 on |t| <= 3 (least-squares fit on Chebyshev nodes;
// |GELU error| <= 5.8e-5 absolute), about 32 VALU cycles per element against ~80 for gelu_fast (12 plain ops + v_rcp + v_exp).
// The FFN-1 epilogue is VALU-bound on exactly these instructions (15 per element pair, both waves of a SIMD in it at once), so round 3
// takes two Horner steps out: degree 6 on |t| <= 2.8 (below).  The output is rounded to bf16 (half-ulp 2e-3 relative) right after.
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef ARX_GELU_DEG
#define ARX_GELU_DEG 6          // round 3 (same-box A/B, profiles/r03/gelu_deg_ab.txt): FFN-1 1.300 -> 1.268 ms, parity margins unchanged; 8 = the round-2 form
#endif
#ifndef ARX_GELU_SAT
#define ARX_GELU_SAT 1          // 1: the saturating form below (9 instructions per element pair); 0: the clamped-argument forms (13)
#endif
__device__ __forceinline__ f32x2 gelu_poly_pk(f32x2 v) {
#if ARX_GELU_SAT
    // GELU(v) = v * Phi(v), Phi(v) ~ sat01(0.5 + v R(v^2)): R of degree 6 fitted on |v| <= 3.96 (Lawson / minimax of v (Phi_p - Phi)); its
    // leading coefficient is positive, so beyond the fit range v R(v^2) runs off to +-infinity on the correct side and the [0, 1] clamp of
    // the final fma — an output modifier, no instruction — is exact saturation; no argument clamp (two v_med3), no 0.5 v, no v / sqrt 2:
    // v*v, six Horner steps, the clamped fma, v * Phi = 9 packed instructions per element pair instead of 13.  |GELU error| <= 1.75e-4
    // absolute in fp32 arithmetic over |v| <= 1e4 (the argument-clamp form: 1.7e-4); an overflowing Horner chain ends in +infinity.
    const f32x2 s = v * v;
    f32x2 r = s * 2.426521917e-08f + -1.672932058e-06f;
    r = r * s + 4.939662904e-05f;
    r = r * s + -8.275752189e-04f;
    r = r * s + 8.835676126e-03f;
    r = r * s + -6.470493972e-02f;
    r = r * s + 3.979698718e-01f;
    f32x2 phi;
    asm("v_pk_fma_f32 %0, %1, %2, 0.5 op_sel_hi:[1,1,0] clamp" : "=v"(phi) : "v"(v), "v"(r));
    return v * phi;
#else
    f32x2 t = v * 0.70710678118654752f;
#if ARX_GELU_DEG == 6
    // Q of degree 6 on |t| <= 2.8 with t Q(t^2) = 1 exactly at the clamp (the saturated branch is exact): |GELU error| <= 1.7e-4
    // absolute (weighted minimax fit of 0.5 |v| |erf error|; the floor of ANY clamp at 2.8 is 0.5 * 3.96 * (1 - erf 2.8) = 1.5e-4),
    // two Horner steps fewer than the degree-8 form
    t.x = __builtin_amdgcn_fmed3f(t.x, -2.8f, 2.8f);
    t.y = __builtin_amdgcn_fmed3f(t.y, -2.8f, 2.8f);
    const f32x2 u = t * t;
    f32x2 q = u * 4.355317931e-06f + -1.504790554e-04f;
    q = q * u + 2.226357740e-03f;
    q = q * u + -1.868393674e-02f;
    q = q * u + 9.987160573e-02f;
    q = q * u + -3.659436702e-01f;
    q = q * u + 1.125613580e+00f;
#else
    t.x = __builtin_amdgcn_fmed3f(t.x, -3.0f, 3.0f);
    t.y = __builtin_amdgcn_fmed3f(t.y, -3.0f, 3.0f);
    const f32x2 u = t * t;
    f32x2 q = u * 4.071986126e-08f + -1.945750910e-06f;
    q = q * u + 4.110950977e-05f;
    q = q * u + -5.118074478e-04f;
    q = q * u + 4.241328686e-03f;
    q = q * u + -2.512698807e-02f;
    q = q * u + 1.111308783e-01f;
    q = q * u + -3.753655851e-01f;
    q = q * u + 1.128284454e+00f;
#endif
    const f32x2 hv = v * 0.5f;
    return (hv * t) * q + hv;                                  // 0.5 v (1 + erf(v / sqrt 2))
#endif
}

// Lane exchange: v_permlane16_swap(X, Y) swaps the odd 16-lane rows of X with the even rows of Y.  With X = this
// lane's 4 columns of block 2jp and Y = of block 2jp+1, afterwards (lo, hi) = (X', Y') are 8 CONSECUTIVE columns in
// every lane: even rows hold block 2jp columns 8*(q>>1).., odd rows block 2jp+1 — one instruction per register
// pair, no selects (checked against torch on the GPU box by tools/gemm_bench.py and tests/test_gpu_parity.py).
template <int MODE, bool CHECK, int NI, int MI>
__device__ __forceinline__ void epilogue_store_v2_impl(const f32x4 (&acc)[NI][MI], const EpiParams& p, int m_base, int n_base,
                                                       int lane, int M, int N) {
    static_assert(NI % 2 == 0, "pairs of 16-column blocks");
    constexpr bool LN_IN = (MODE == EPI_LN_BIAS || MODE == EPI_LN_BIAS_GELU);
    constexpr bool STATS = (MODE == EPI_RESID_STATS || MODE == EPI_LNRESID_STATS);
    constexpr bool RESID = (MODE == EPI_BIAS_RESID || STATS);
    const int mq = lane & 15, q = lane >> 4, odd = q & 1;
    float a_mean[MI], a_rstd[MI], r_mean[MI], r_rstd[MI], st_s[MI], st_q[MI];
    uint32_t orow[MI], rrow[MI];          // element offsets (M * ld < 2^32 for every encoder shape)
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        int m = m_base + i * 16 + mq;
        if (CHECK) m = m < M ? m : M - 1;
        orow[i] = (uint32_t)m * (uint32_t)p.ldc;
        rrow[i] = (uint32_t)m * (uint32_t)p.ldr;
        a_mean[i] = 0.f; a_rstd[i] = 1.f; r_mean[i] = 0.f; r_rstd[i] = 1.f; st_s[i] = 0.f; st_q[i] = 0.f;
        if constexpr (LN_IN) { a_mean[i] = p.a_sum[m]; a_rstd[i] = p.a_sq[m]; }
        if constexpr (MODE == EPI_LNRESID_STATS) { r_mean[i] = p.r_sum[m]; r_rstd[i] = p.r_sq[m]; }
    }
#pragma unroll
    for (int jp = 0; jp < NI / 2; ++jp) {
        const int n = n_base + (2 * jp + odd) * 16 + (q >> 1) * 8;
        const int nc = (!CHECK || n < N) ? n : 0;
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + nc), b1 = *reinterpret_cast<const f32x4*>(p.bias + nc + 4);
        f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0, g0 = s0, g1 = s0, e0 = s0, e1 = s0;
        if constexpr (LN_IN) { s0 = *reinterpret_cast<const f32x4*>(p.s_vec + nc); s1 = *reinterpret_cast<const f32x4*>(p.s_vec + nc + 4); }
        if constexpr (MODE == EPI_LNRESID_STATS) {
            g0 = *reinterpret_cast<const f32x4*>(p.r_gamma + nc); g1 = *reinterpret_cast<const f32x4*>(p.r_gamma + nc + 4);
            e0 = *reinterpret_cast<const f32x4*>(p.r_beta + nc); e1 = *reinterpret_cast<const f32x4*>(p.r_beta + nc + 4);
        }
        // all residual vectors of this column pair are requested BEFORE any of them is consumed: the loads overlap
        // each other and the exchange below instead of paying one L2 round trip per (row block) in sequence
        u32x4 rres[MI];
        if constexpr (RESID) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
                if (!CHECK || (m_base + i * 16 + mq < M && n < N)) rres[i] = *reinterpret_cast<const u32x4*>(p.resid + rrow[i] + n);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            f32x4 lo, hi;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[2 * jp][i][r]),
                                                                 __float_as_uint(acc[2 * jp + 1][i][r]), false, false);
                lo[r] = __uint_as_float(sw[0]); hi[r] = __uint_as_float(sw[1]);
            }
            if (CHECK && (m_base + i * 16 + mq >= M || n >= N)) continue;
            // two elements per VALU instruction from here on (v_pk_*_f32)
            f32x2 vv[4] = {f32x2{lo[0], lo[1]}, f32x2{lo[2], lo[3]}, f32x2{hi[0], hi[1]}, f32x2{hi[2], hi[3]}};
            const f32x2 bb[4] = {f32x2{b0[0], b0[1]}, f32x2{b0[2], b0[3]}, f32x2{b1[0], b1[1]}, f32x2{b1[2], b1[3]}};
            if constexpr (LN_IN) {
                const f32x2 ss[4] = {f32x2{s0[0], s0[1]}, f32x2{s0[2], s0[3]}, f32x2{s1[0], s1[1]}, f32x2{s1[2], s1[3]}};
                const f32x2 nm = f32x2{-a_mean[i], -a_mean[i]}, rs = f32x2{a_rstd[i], a_rstd[i]};
#pragma unroll
                for (int pi = 0; pi < 4; ++pi) vv[pi] = rs * (nm * ss[pi] + vv[pi]) + bb[pi];
            } else {
#pragma unroll
                for (int pi = 0; pi < 4; ++pi) vv[pi] = vv[pi] + bb[pi];
            }
            if constexpr (MODE == EPI_BIAS_GELU || MODE == EPI_LN_BIAS_GELU) {
#pragma unroll
                for (int pi = 0; pi < 4; ++pi) vv[pi] = gelu_poly_pk(vv[pi]);
            }
            float v[8] = {vv[0].x, vv[0].y, vv[1].x, vv[1].y, vv[2].x, vv[2].y, vv[3].x, vv[3].y};
            if constexpr (RESID) {
                const u32x4 rr = rres[i];
                float ra[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) unpack_bf16x2(rr[r], ra[2 * r], ra[2 * r + 1]);
                if constexpr (MODE == EPI_LNRESID_STATS) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        ra[r] = fmaf((ra[r] - r_mean[i]) * r_rstd[i], g0[r], e0[r]);
                        ra[4 + r] = fmaf((ra[4 + r] - r_mean[i]) * r_rstd[i], g1[r], e1[r]);
                    }
                }
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] += ra[r];
            }
            u32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = pack_bf16x2(v[2 * r], v[2 * r + 1]);
            *reinterpret_cast<u32x4*>(p.out + orow[i] + n) = o;
            if constexpr (STATS) {
                // statistics of the bf16-ROUNDED values (what the consumers will read back)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x0, x1;
                    unpack_bf16x2(o[r], x0, x1);
                    st_s[i] += x0 + x1;
                    st_q[i] = fmaf(x0, x0, fmaf(x1, x1, st_q[i]));
                }
            }
        }
    }
    if constexpr (STATS) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            float a = st_s[i], b = st_q[i];
            a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
            b += __shfl_xor(b, 16); b += __shfl_xor(b, 32);
            const int m = m_base + i * 16 + mq;
            const int64_t part = n_base >> 6;                    // this wave's 64-column slice
            if (q == 0 && m < M && n_base < N) { p.o_sum[part * p.o_ld + m] = a; p.o_sq[part * p.o_ld + m] = b; }
        }
    }
}

template <int MODE, int NI, int MI>
__device__ __forceinline__ void epilogue_store_v2(const f32x4 (&acc)[NI][MI], const EpiParams& p, int m_base, int n_base,
                                                  int lane, int M, int N) {
    // interior wave tiles (the common case) skip every bounds test
    if (m_base + MI * 16 <= M && n_base + NI * 16 <= N) epilogue_store_v2_impl<MODE, false, NI, MI>(acc, p, m_base, n_base, lane, M, N);
    else {
        epilogue_store_v2_impl<MODE, true, NI, MI>(acc, p, m_base, n_base, lane, M, N);
        // edge tiles load vectors whose use is exec-masked; retire them HERE (vmcnt(0), a real S_WAITCNT the compiler's
        // counter tracking sees) so that a persistent caller's loop header does not inherit a conservative vmcnt(0) that
        // would also drain the interior tiles' stores and the prefetch stream
        __builtin_amdgcn_s_waitcnt(0x0F70);
    }
}

template <int BM, int BN, int WM, int WN, int SLOTS, int MODE, int OPT = 0>
__global__ __launch_bounds__(WM * WN * 64, 2) void gemm_ring_kernel(const bf16_t* __restrict__ A, int64_t lda,
                                                                   const bf16_t* __restrict__ W, int64_t ldw,
                                                                   int M, int N, int K, int tiles_m, int tiles_n,
                                                                   EpiParams ep) {
    using ML = GemmRing<bf16_t, BM, BN, WM, WN, SLOTS, OPT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tile_m = t / tiles_n, tile_n = t % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    f32x4 acc[ML::NI][ML::MI];
    ML::run(A, lda, M, W, ldw, N, K, m0, n0, smem, acc);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    epilogue_store_v2<MODE, ML::NI, ML::MI>(acc, ep, m0 + (wid / WN) * ML::TM, n0 + (wid % WN) * ML::TN, lane, M, N);
}

// v0 main loop with the v2 epilogue (isolates the epilogue change)
template <int BM, int BN, int WM, int WN, int MODE, int OPT = 0>
__global__ __launch_bounds__(WM * WN * 64) void gemm_v0e2_kernel(const bf16_t* __restrict__ A, int64_t lda,
                                                                   const bf16_t* __restrict__ W, int64_t ldw,
                                                                   int M, int N, int K, int tiles_m, int tiles_n,
                                                                   EpiParams ep) {
    using ML = GemmMainloop<bf16_t, BM, BN, WM, WN, true, OPT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tile_m = t / tiles_n, tile_n = t % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    f32x4 acc[ML::NI][ML::MI];
    const int koff = (OPT & 64) ? tile_n * 2 : 0;
    ML::run(A, lda, M, W, ldw, N, K, m0, n0, smem, acc, koff);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    epilogue_store_v2<MODE, ML::NI, ML::MI>(acc, ep, m0 + (wid / WN) * ML::TM, n0 + (wid % WN) * ML::TN, lane, M, N);
}


// (A persistent one-block-per-CU variant with cross-tile prefetch was measured 5-15 % SLOWER than the plain
//  grid on all four encoder shapes and was removed; DESIGN.md §4 keeps the numbers.)

// (A 128 x 768 "row-block" GEMM with the residual + LayerNorm fused into its epilogue — every A element staged
//  once chip-wide, no pre-LN tensor in HBM — was built and measured: its 48 KB-per-half-step W stream left only
//  2 x BK=32 LDS stages and it ran 674 TF on FFN-2 / 398 TF on the O projection, slower than the 256^2 kernel
//  plus the separate LayerNorm kernel (1.40 / 0.65 ms vs 1.84 / 0.78 ms); removed.  DESIGN.md §4.)

// =====================================================================================================
// "A3W2" main loop: same 256x256x64 tile, LDS images and MFMA order as the default kernel, but the streamed A
// operand (activations, HBM/MALL latency) rides a THREE-slot ring kept two k-steps ahead while the small, L2-resident
// W operand stays double-buffered: 3*32 KB + 2*32 KB = 160 KB = the whole LDS of a CU.  The prefetch distance is
// expressed by a counted s_waitcnt (the A tile issued last stays in flight across the raw s_barrier).
template <int BM, int BN, int WM, int WN, int MODE, int KROT = 2>
__global__ __launch_bounds__(WM * WN * 64) void gemm_a3w2_kernel(const bf16_t* __restrict__ A, int64_t lda,
                                                                  const bf16_t* __restrict__ W, int64_t ldw,
                                                                  int M, int N, int K, int tiles_m, int tiles_n,
                                                                  EpiParams ep) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 16, NI = TN / 16;
    constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128;
    constexpr int LA = BM * 8 / NT, LW = BN * 8 / NT;          // global_load_lds per wave per tile
    using vec = typename Mfma<bf16_t>::vec;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Abuf = smem;                                   // 3 slots
    char* const Wbuf = smem + 3 * A_BYTES;                     // 2 slots
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tile_m = t / tiles_n, tile_n = t % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const uint16_t* Ag = reinterpret_cast<const uint16_t*>(A);
    const uint16_t* Wg = reinterpret_cast<const uint16_t*>(W);
    const int nk = K >> 6;
    const int koff = (tile_n * KROT) % nk;
    auto kcol = [&](int kt) { int k = kt + koff; return (k >= nk ? k - nk : k) << 6; };
    u32x4 dummy_a[LA], dummy_w[LW];
    auto issueA = [&](int kt) { stage_issue<BM, NT, true>(Ag, lda, m0, M - 1, kcol(kt), Abuf + (kt % 3) * A_BYTES, tid, dummy_a); };
    auto issueW = [&](int kt) { stage_issue<BN, NT, true>(Wg, ldw, n0, N - 1, kcol(kt), Wbuf + (kt & 1) * W_BYTES, tid, dummy_w); };
    f32x4 acc[NI][MI];
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fq = lane >> 4, sw = (lane >> 1) & 7;
    const int off0 = frow * 128 + (((0 + fq) ^ sw) << 4);
    const int off1 = frow * 128 + (((4 + fq) ^ sw) << 4);
    const int a_base = wm * TM * 128, w_base = wn * TN * 128;

    // prologue: A(0), W(0), A(1) in flight; A(0) and W(0) must land, A(1) may stay in flight
    issueA(0); issueW(0);
    if (nk > 1) { issueA(1); wait_vmcnt<LA>(); } else { wait_vmcnt<0>(); }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    for (int kt = 0; kt < nk; ++kt) {
        // queue on entry (oldest first): [A(kt+1)]           -> issue W(kt+1), A(kt+2)
        if (kt + 1 < nk) issueW(kt + 1);
        if (kt + 2 < nk) issueA(kt + 2);
        const char* As = Abuf + (kt % 3) * A_BYTES + a_base;
        const char* Ws = Wbuf + (kt & 1) * W_BYTES + w_base;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int off = ks ? off1 : off0;
            vec af[MI], wf[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const vec*>(As + i * 2048 + off);
#pragma unroll
            for (int j = 0; j < NI; ++j) wf[j] = *reinterpret_cast<const vec*>(Ws + j * 2048 + off);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int i = 0; i < MI; ++i) acc[j][i] = Mfma<bf16_t>::mma(wf[j], af[i], acc[j][i]);
            __builtin_amdgcn_s_setprio(0);
        }
        // A(kt+1) and W(kt+1) must have landed; only A(kt+2) (the youngest LA loads) may remain in flight
        if (kt + 2 < nk) wait_vmcnt<LA>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    epilogue_store_v2<MODE, NI, MI>(acc, ep, m0 + wm * TM, n0 + wn * TN, lane, M, N);
}
