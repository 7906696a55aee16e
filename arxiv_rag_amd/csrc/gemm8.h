// 256 x 256 x 64 "8-phase" main loop (two k-tiles = eight phases per unrolled iteration in the playbook's naming; here
// four phases per 64-deep k-tile).  Same tile, MFMA (16x16x32), LDS image ([rows][64 elem], 16-B chunk XOR (row>>1)&7 on the
// glds SOURCE address) and wave tile (128 x 64, 2 x 4 waves) as the default kernel, but a different schedule:
//
//   * the k-tile is staged as FOUR half-tiles of 16 KB (A0, A1, B0, B1), two global_load_lds per thread each; the halves are
//     interleaved row sets so that every wave needs rows of both:  A half h = block rows {wr*128 + h*64 + [0,64)},
//     B half g = block columns {wc*64 + g*32 + [0,32)}  -> the wave's output stays a contiguous 128 x 64 rectangle;
//   * each phase computes ONE quadrant of the wave tile (64 x 32 outputs x K=64 = 16 MFMAs) from register fragments, loads
//     the fragments the next quadrant is missing (phase 1: B-sub0 + A-sub0 = 12 ds_read_b128, 2: B-sub1 = 4, 3: A-sub1 = 8,
//     4: none) and issues ONE half-tile of prefetch;
//   * the prefetch stream B0 A0 B1 A1 runs three half-tiles ahead behind ONE counted s_waitcnt vmcnt(6) per k-tile (phase 4)
//     and raw s_barriers — never vmcnt(0) in the loop;
//   * the four waves with wr == 1 run one barrier interval behind the four with wr == 0 (waves w and w+4 share a SIMD):
//     while one wave of a SIMD is in its MFMA section the other is in its ds_read / glds section.
//
// Hazards (phase p = load section L(p), barrier, MFMA section, barrier; group 1 is one barrier late):
//   RAW  a half-tile is readable one phase after the counted wait that retires it (wait before phase 4's first barrier ->
//        read from phase 1 of the next k-tile);
//   WAR  a slot is restaged >= 2 phases after its last ds_read, or 1 phase after when an lgkmcnt before the reading phase's
//        first barrier retired those reads (B-sub0: lgkmcnt(8) in phase 1 -> B0 restaged in phase 2);
//        A0 (read in 1) restaged in 3, B1 (read in 2) in 4, A1 (read in 3) in phase 1 of the next k-tile.
#pragma once
#include "gemm.h"
#include "epi3.h"

#ifdef ARX_STAMP
#define g_stamp1 stamp1_ref
#endif

// Tile walk of the 256-wide kernels.  Block id -> (XCD = id % 8, position L = id / 8 inside that XCD's share).  An XCD owns a
// contiguous range of m-panels (all their n-tiles) and walks it BAND by band: a band is `bw` adjacent n-tiles chosen so that the
// band's slice of W (bw x 256 x K bf16) stays resident in the XCD's 4-MB L2 while the A panels stream through once per band;
// inside a band the order is n-fastest, so the CUs of an XCD share both the A panel and the W slices they are reading.
// Without bands (bw = tiles_n) FFN-1's 4.7-MB W and the A stream evict each other: 2.7-2.9 GB of L2 fills per launch against
// 0.41 GB of operands (rocprofv3 FETCH_SIZE); with two bands of six the A operand is filled twice and W about once (1.84 GB).
// Fills are not time, though (round-2 A/B, profiles/r02/gemm_store_band_ab.json: the Infinity Cache serves them): FFN-1 runs the same
// banded or not, and QKV (W = 3.5 MB, inside the L2) is 3 % FASTER unbanded at 1.8x the fills — so only a W beyond the L2 is banded.
struct TileWalk {
    int tiles_m, tiles_n, bw;
    __device__ __forceinline__ TileWalk(int tm, int tn, int K, int bw_override = 0) : tiles_m(tm), tiles_n(tn) {
        if (bw_override > 0) { bw = bw_override < tn ? bw_override : tn; return; }
        const int w_tile = 512 * K;                               // bytes of one n-tile's W slice
        const int total = tn * w_tile;
        bw = tn;
        if (total > (4 << 20) && K <= 1024) {                     // W larger than the XCD's 4-MB L2 and a small A operand: band it (~2.4 MB per band)
            const int nb = (total + (12 << 18) / 5 * 4 - 1) / ((12 << 18) / 5 * 4);
            bw = (tn + nb - 1) / nb;
        }
    }
    // grid size that covers every XCD's share
    static int grid(int tm, int tn) { return 8 * ((tm + 7) / 8) * tn; }
    __device__ __forceinline__ bool coords(int id, int& tile_m, int& tile_n) const {
        const int x = id & 7;
        int L = id >> 3;
        const int pq = tiles_m >> 3, pr = tiles_m & 7;
        const int panels = pq + (x < pr ? 1 : 0), m_lo = x * pq + (x < pr ? x : pr);
        if (L >= panels * tiles_n) return false;
        int b = 0, w = bw < tiles_n ? bw : tiles_n;
        while (L >= panels * w) { L -= panels * w; ++b; w = tiles_n - b * bw; w = w < bw ? w : bw; }
        const int q = L / w;
        tile_m = m_lo + q; tile_n = b * bw + (L - q * w);
        return true;
    }
};
// PERM: the 32 W rows (output columns) of a B half-tile slice are PLACED in LDS so that MFMA block j (0/1) of the slice holds the columns
// 8g + 4j + [0,4), g = 0..3, instead of 16j + [0,16): a lane's four accumulator values of block 0 and of block 1 are then EIGHT CONSECUTIVE
// output columns (16 B of bf16) and the epilogue needs no cross-lane exchange (epi3.h).  Only the source row of each DMA piece changes; the
// LDS image, the swizzle and every fragment read are the same.
__device__ __forceinline__ int b_slice_col(int rho) { return ((rho & 15) >> 2) * 8 + (rho >> 4) * 4 + (rho & 3); }

template <typename T, int KROT, bool PERM = false>
struct Gemm8Phase {
    static constexpr int BM = 256, BN = 256, NT = 512, MI = 8, NI = 4;
    static constexpr int HALF_BYTES = 128 * 128, BUF_BYTES = 4 * HALF_BYTES, STAGE_OFF = 2 * BUF_BYTES;
#ifdef ARX_STAMP
    static constexpr int KT_STAMP_OFF = STAGE_OFF + 2 * EpiStage::BYTES;    // dev build: 2 groups x 16 k-tile start stamps
    static constexpr int SMEM_BYTES = KT_STAMP_OFF + 256;
#else
    static constexpr int SMEM_BYTES = STAGE_OFF + 2 * EpiStage::BYTES;      // k-tile buffers + two epilogue-vector stages
#endif
    using vec = typename Mfma<T>::vec;

    static __device__ __forceinline__ void fence() { asm volatile("" ::: "memory"); }
    static __device__ __forceinline__ void bar() {
        fence(); __builtin_amdgcn_s_barrier(); fence();
    }

    static __device__ __forceinline__ void run(const T* __restrict__ A, int64_t lda, int M, const T* __restrict__ W,
                                               int64_t ldw, int N, int K, int m0, int n0, char* smem,
                                               f32x4 (&acc)[NI][MI], int koff
#ifdef ARX_STAMP
                                               , unsigned long long& stamp1_ref
#endif
                                               ) {
        const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: SGPRs
        const int wr = wid >> 2, wc = wid & 3;
        const uint16_t* Ag = reinterpret_cast<const uint16_t*>(A);
        const uint16_t* Wg = reinterpret_cast<const uint16_t*>(W);
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int i = 0; i < MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

        // per-thread source element offsets of its two 16-B pieces of every half-tile (k column added per k-tile)
        uint32_t aoff[2][2], boff[2][2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int cid = it * NT + tid, r = cid >> 3, c = (cid & 7) ^ ((r >> 1) & 7);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int gm = m0 + (r >> 6) * 128 + h * 64 + (r & 63);
                gm = gm < M ? gm : M - 1;
                aoff[h][it] = ((uint32_t)gm * (uint32_t)lda + c * 8) * 2u;      // BYTE offsets: the voffset operand of buffer_load ... lds
                int gn = n0 + (r >> 5) * 64 + h * 32 + (PERM ? b_slice_col(r & 31) : (r & 31));
                gn = gn < N ? gn : N - 1;
                boff[h][it] = ((uint32_t)gn * (uint32_t)ldw + c * 8) * 2u;
            }
        }
        const int nk = K >> 6;
        koff = koff % nk;
        auto kcol = [&](int kt) { int k = kt + koff; return (k >= nk ? k - nk : k) << 6; };
        char* const wave_dst = smem + wid * 1024;               // wave-uniform; lane l lands at +16*l
        // LDS-DMA through BUFFER loads (buffer_load_dwordx4 v, s[rsrc], s_off offen lds): the per-lane part of the address is one
        // loop-invariant 32-bit VGPR and the k-tile's column is the scalar offset — no 64-bit per-lane address arithmetic per piece,
        // half the address registers through the texture addresser (the global_load_lds form cost 60-100 issue cycles per piece, and
        // the load sections of phases 1 and 3 are what the k-loop waits for: profiles/r02/gemm_kloop_bisect.txt)
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Ag, 0, 0xffffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)Wg, 0, 0xffffffff, 0x00020000);
        auto issue_a = [&](int h, int kt, int buf) {
            const int kb = kcol(kt) * 2;
            char* dst = wave_dst + buf * BUF_BYTES + h * HALF_BYTES;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_t*)dst, 16, aoff[h][0], kb, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_t*)(dst + 8192), 16, aoff[h][1], kb, 0, 0);
        };
        auto issue_b = [&](int g, int kt, int buf) {
            const int kb = kcol(kt) * 2;
            char* dst = wave_dst + buf * BUF_BYTES + (2 + g) * HALF_BYTES;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_void_t*)dst, 16, boff[g][0], kb, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_void_t*)(dst + 8192), 16, boff[g][1], kb, 0, 0);
        };

        // fragment read offsets inside a half-tile
        const int frow = lane & 15, fq = lane >> 4, sw = (lane >> 1) & 7;
        const int fo0 = frow * 128 + (((0 + fq) ^ sw) << 4), fo1 = frow * 128 + (((4 + fq) ^ sw) << 4);
        const int a_row = wr * 64 * 128, b_row = wc * 32 * 128;

        vec af[4][2], wf0[2][2], wf1[2][2];
        auto read_a = [&](const char* buf, int h) {
            const char* p = buf + h * HALF_BYTES + a_row;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i][0] = *reinterpret_cast<const vec*>(p + i * 2048 + fo0);
                af[i][1] = *reinterpret_cast<const vec*>(p + i * 2048 + fo1);
            }
        };
        auto read_b = [&](const char* buf, int g, vec (&wf)[2][2]) {
            const char* p = buf + (2 + g) * HALF_BYTES + b_row;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                wf[j][0] = *reinterpret_cast<const vec*>(p + j * 2048 + fo0);
                wf[j][1] = *reinterpret_cast<const vec*>(p + j * 2048 + fo1);
            }
        };
        auto quad = [&](int h, int g, const vec (&wf)[2][2]) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[g * 2 + j][h * 4 + i] = Mfma<T>::mma(wf[j][ks], af[i][ks], acc[g * 2 + j][h * 4 + i]);
            __builtin_amdgcn_s_setprio(0);
        };

        // prologue: k-tile 0 whole + the first three half-tiles of k-tile 1, in stream order
        issue_b(0, 0, 0); issue_a(0, 0, 0); issue_b(1, 0, 0); issue_a(1, 0, 0);
        if (nk > 1) {
            issue_b(0, 1, 1); issue_a(0, 1, 1); issue_b(1, 1, 1);
            wait_vmcnt<6>();
        } else {
            wait_vmcnt<0>();
        }
        bar();
#ifdef ARX_STAMP
        g_stamp1 = __builtin_readcyclecounter();
#endif
        if (wr == 1) bar();                                     // group 1 runs one barrier interval late

        for (int kt = 0; kt < nk; ++kt) {
            const int b = kt & 1;
            const char* cur = smem + b * BUF_BYTES;
            const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
            // ---- phase 1: quadrant (A-sub0, B-sub0)
            read_b(cur, 0, wf0);
            __builtin_amdgcn_sched_barrier(0);
            read_a(cur, 0);
            if (more1) issue_a(1, kt + 1, b ^ 1);
            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");  // the B-sub0 reads (issued first) are retired: B0 may be restaged in phase 2
            bar();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            quad(0, 0, wf0);
            bar();
            // ---- phase 2: quadrant (A-sub0, B-sub1)
            read_b(cur, 1, wf1);
            if (more2) issue_b(0, kt + 2, b);
            bar();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            quad(0, 1, wf1);
            bar();
            // ---- phase 3: quadrant (A-sub1, B-sub1)
            read_a(cur, 1);
            if (more2) issue_a(0, kt + 2, b);
            bar();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            quad(1, 1, wf1);
            bar();
            // ---- phase 4: quadrant (A-sub1, B-sub0), fragments already in registers; the k-tile's one counted wait
            if (more2) { issue_b(1, kt + 2, b); wait_vmcnt<6>(); }
            else wait_vmcnt<0>();
            bar();
            quad(1, 0, wf0);
            if (more1 || wr == 0) bar();                         // group 1 skips its very last barrier (counts stay equal)
        }
    }
};

template <int MODE, int KROT = 0>
__global__ __launch_bounds__(512) void gemm_8phase_kernel(const bf16_t* __restrict__ A, int64_t lda,
                                                           const bf16_t* __restrict__ W, int64_t ldw,
                                                           int M, int N, int K, int tiles_m, int tiles_n, EpiParams ep) {
    using ML = Gemm8Phase<bf16_t, KROT, true>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef ARX_DEV_VARIANTS
    const TileWalk walk(tiles_m, tiles_n, K, ep.dev_bw);
#else
    const TileWalk walk(tiles_m, tiles_n, K);
#endif
    int tile_m, tile_n;
    if (!walk.coords(blockIdx.x, tile_m, tile_n)) return;
    const int m0 = tile_m * 256, n0 = tile_n * 256;
    f32x4 acc[ML::NI][ML::MI];
#ifdef ARX_STAMP
    const unsigned long long ts0 = __builtin_readcyclecounter();
#endif
    epi_stage_issue<MODE>(ep, m0, n0, smem + ML::STAGE_OFF, __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), threadIdx.x & 63, N);   // oldest loads of the tile
#ifdef ARX_STAMP
    unsigned long long ts1 = 0;
    ML::run(A, lda, M, W, ldw, N, K, m0, n0, smem, acc, tile_n * KROT, ts1);
#else
    ML::run(A, lda, M, W, ldw, N, K, m0, n0, smem, acc, tile_n * KROT);
#endif
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef ARX_STAMP
    const unsigned long long ts2 = __builtin_readcyclecounter();
#endif
    epilogue_store_v3<MODE>(acc, ep, m0, n0, wid >> 2, wid & 3, lane, M, N, smem + ML::STAGE_OFF);
#ifdef ARX_STAMP
    const unsigned long long ts3 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // every store of the epilogue acknowledged
    const unsigned long long ts4 = __builtin_readcyclecounter();
    if (ep.stamps && (threadIdx.x == 0 || threadIdx.x == 256)) {
        unsigned long long* o = ep.stamps + ((size_t)blockIdx.x * 2 + (threadIdx.x >> 8)) * 32;
        o[0] = ts0; o[1] = ts2; o[2] = ts3;
        o[3] = ts1; o[4] = ts4; o[5] = 0; o[6] = 0; o[7] = 0;
    }
#endif
}

// ---- persistent form: one block per CU walks tiles blockIdx.x, +gridDim.x, ... and the half-tile prefetch stream runs straight
// through the tile boundary: the seven free issue slots of a tile's last two k-tiles carry the first seven half-tiles of the
// next tile (exactly what the per-tile prologue above issues), so the first-load latency (~4 us) is paid once per CU instead of
// once per tile, and the three half-tiles in flight land under the epilogue.  Needs an even number of k-tiles (buffer parity).
//
// The loop is generic over the element type (bf16 / f16 / int8 pairs: Mfma<T>) and over a POLICY object that supplies
//   bool tile(int o, int& m0, int& n0, int& ko)        block-order index o -> tile origin (false: past the end)
//   void stage_issue(int m0, int n0, char* stage, int wid, int lane)   per-tile epilogue vectors -> LDS stage (may be a no-op)
//   void epilogue(acc, m0, n0, wr, wc, lane, stage)     the tile's output
//   REBASE_W   the W operand is addressed relative to the tile's first row through a per-tile buffer descriptor (a 15-GB corpus does
//              not fit the 32-bit offsets the encoder's weights use)
// The encoder's linear layers (EncoderTilePolicy below) and the search's pass A at >= 256 queries (search.hip) share it.
template <typename T, typename Pol>
__device__ __forceinline__ void gemm8_persistent_body(const T* __restrict__ A, int64_t lda, const T* __restrict__ W, int64_t ldw,
                                                      int M, int N, int K, const Pol& pol, char* smem) {
    using ML = Gemm8Phase<T, 0>;
    using vec = typename ML::vec;
    constexpr int HALF_BYTES = ML::HALF_BYTES, BUF_BYTES = ML::BUF_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: SGPRs
    const int wr = wid >> 2, wc = wid & 3;
    const uint16_t* Ag = reinterpret_cast<const uint16_t*>(A);
    const uint16_t* Wg = reinterpret_cast<const uint16_t*>(W);
    const int stride = gridDim.x, nk = K >> 6;
    auto kcol = [&](int kt, int ko) { int k = kt + ko; return (k >= nk ? k - nk : k) << 6; };

    uint32_t aoff[2][2], boff[2][2];
    auto set_aoff = [&](int h, int m0) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int cid = it * 512 + tid, r = cid >> 3, c = (cid & 7) ^ ((r >> 1) & 7);
            int gm = m0 + (r >> 6) * 128 + h * 64 + (r & 63);
            gm = gm < M ? gm : M - 1;
            aoff[h][it] = ((uint32_t)gm * (uint32_t)lda + c * 8) * 2u;      // byte offsets (buffer_load ... lds voffset)
        }
    };
    auto set_boff = [&](int g, int n0) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int cid = it * 512 + tid, r = cid >> 3, c = (cid & 7) ^ ((r >> 1) & 7);
            int gn = n0 + (r >> 5) * 64 + g * 32 + (Pol::PERMUTE_B ? b_slice_col(r & 31) : (r & 31));
            gn = gn < N ? gn : N - 1;
            if constexpr (Pol::REBASE_W) gn -= n0;                          // relative to the tile's descriptor base
            boff[g][it] = ((uint32_t)gn * (uint32_t)ldw + c * 8) * 2u;
        }
    };
    char* const wave_dst = smem + wid * 1024;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Ag, 0, 0xffffffff, 0x00020000);      // see Gemm8Phase::run
    auto w_rsrc = [&](int n0) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(Wg + (Pol::REBASE_W ? (int64_t)n0 * ldw : (int64_t)0)), 0, 0xffffffff, 0x00020000);
    };
    __amdgpu_buffer_rsrc_t rsW = w_rsrc(0);
    auto issue_a = [&](int h, int kc, int buf) {
        char* dst = wave_dst + buf * BUF_BYTES + h * HALF_BYTES;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_t*)dst, 16, aoff[h][0], kc * 2, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_t*)(dst + 8192), 16, aoff[h][1], kc * 2, 0, 0);
    };
    auto issue_b = [&](int g, int kc, int buf) {
        char* dst = wave_dst + buf * BUF_BYTES + (2 + g) * HALF_BYTES;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_void_t*)dst, 16, boff[g][0], kc * 2, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_void_t*)(dst + 8192), 16, boff[g][1], kc * 2, 0, 0);
    };
    const int frow = lane & 15, fq = lane >> 4, sw = (lane >> 1) & 7;
    const int fo0 = frow * 128 + (((0 + fq) ^ sw) << 4), fo1 = frow * 128 + (((4 + fq) ^ sw) << 4);
    const int a_row = wr * 64 * 128, b_row = wc * 32 * 128;

    f32x4 acc[4][8];
    vec af[4][2], wf0[2][2], wf1[2][2];
    auto read_a = [&](const char* buf, int h) {
        const char* p = buf + h * HALF_BYTES + a_row;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[i][0] = *reinterpret_cast<const vec*>(p + i * 2048 + fo0);
            af[i][1] = *reinterpret_cast<const vec*>(p + i * 2048 + fo1);
        }
    };
    auto read_b = [&](const char* buf, int g, vec (&wf)[2][2]) {
        const char* p = buf + (2 + g) * HALF_BYTES + b_row;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            wf[j][0] = *reinterpret_cast<const vec*>(p + j * 2048 + fo0);
            wf[j][1] = *reinterpret_cast<const vec*>(p + j * 2048 + fo1);
        }
    };
    auto quad = [&](int h, int g, const vec (&wf)[2][2]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[g * 2 + j][h * 4 + i] = Mfma<T>::mma(wf[j][ks], af[i][ks], acc[g * 2 + j][h * 4 + i]);
        __builtin_amdgcn_s_setprio(0);
    };

    int orig = blockIdx.x;
    int m0, n0, ko;
    if (!pol.tile(orig, m0, n0, ko)) return;
    if constexpr (Pol::REBASE_W) rsW = w_rsrc(n0);
    set_aoff(0, m0); set_aoff(1, m0); set_boff(0, n0); set_boff(1, n0);
    int sbuf = 0;                                                // epilogue-vector stage of the current tile (alternates)
    pol.stage_issue(m0, n0, smem + ML::STAGE_OFF, wid, lane);
    issue_b(0, kcol(0, ko), 0); issue_a(0, kcol(0, ko), 0); issue_b(1, kcol(0, ko), 0); issue_a(1, kcol(0, ko), 0);
    issue_b(0, kcol(1, ko), 1); issue_a(0, kcol(1, ko), 1); issue_b(1, kcol(1, ko), 1);
    wait_vmcnt<6>();
    ML::bar();
    if (wr == 1) ML::bar();

    for (;;) {
        const int onext = orig + stride;
        int m0n = 0, n0n = 0, kon = 0;
        const bool has_next = pol.tile(onext, m0n, n0n, kon);
        // interior -> interior tile steps move every source offset by a block-uniform amount
        const bool edge = (m0 + 256 > M) || (m0n + 256 > M);
        const bool nclamp = (n0 + 256 > N) || (n0n + 256 > N);
        const uint32_t d_a = (uint32_t)(m0n - m0) * (uint32_t)lda * 2u, d_b = (uint32_t)(n0n - n0) * (uint32_t)ldw * 2u;      // bytes
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef ARX_STAMP
        const unsigned long long pts0 = __builtin_readcyclecounter();
        unsigned long long wst[4] = {0, 0, 0, 0};                // k-tile 0 / 1: time around the counted wait
#endif

        for (int kt = 0; kt < nk; ++kt) {
            const int b = kt & 1;
            const char* cur = smem + b * BUF_BYTES;
            const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
            const bool go1 = more1 || has_next, go2 = more2 || has_next;
            const int kc1 = more1 ? kcol(kt + 1, ko) : kcol(0, kon);
            const int kc2 = more2 ? kcol(kt + 2, ko) : kcol(kt + 2 - nk, kon);
#ifdef ARX_STAMP
            if (kt < 16 && (tid & 255) == 0) *reinterpret_cast<volatile unsigned long long*>(smem + ML::KT_STAMP_OFF + (tid >> 8) * 128 + kt * 8) = __builtin_readcyclecounter();
#endif
            // ---- phase 1
            read_b(cur, 0, wf0);
            __builtin_amdgcn_sched_barrier(0);
            read_a(cur, 0);
            if (kt == nk - 1 && has_next)                        // next tile's epilogue vectors: ahead of (older than) its A1 pieces
                pol.stage_issue(m0n, n0n, smem + ML::STAGE_OFF + (sbuf ^ 1) * EpiStage::BYTES, wid, lane);
            if (!more1 && has_next) {                            // the stream's A1 pieces now come from the next tile
                if (edge) { asm volatile("" : "+s"(m0n)); set_aoff(1, m0n); }      // clamped rows: recompute (last tile row only)
                else { aoff[1][0] += d_a; aoff[1][1] += d_a; }
            }
            if (go1) issue_a(1, kc1, b ^ 1);
            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            ML::bar();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            quad(0, 0, wf0);
            ML::bar();
            // ---- phase 2
            read_b(cur, 1, wf1);
            if (kt == nk - 2 && has_next) {
                if constexpr (Pol::REBASE_W) {                   // every B piece from here on belongs to the next tile: its own descriptor
                    rsW = w_rsrc(n0n);
                    if (nclamp) { asm volatile("" : "+s"(n0n)); set_boff(0, n0n); set_boff(1, n0n); }
                } else {
                    if (nclamp) { asm volatile("" : "+s"(n0n)); set_boff(0, n0n); set_boff(1, n0n); }   // half-present last n-tile: clamped rows
                    else { boff[0][0] += d_b; boff[0][1] += d_b; boff[1][0] += d_b; boff[1][1] += d_b; }
                }
                if (edge) { asm volatile("" : "+s"(m0n)); set_aoff(0, m0n); }
                else { aoff[0][0] += d_a; aoff[0][1] += d_a; }
            }
            if (go2) issue_b(0, kc2, b);
            ML::bar();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            quad(0, 1, wf1);
            ML::bar();
            // ---- phase 3
            read_a(cur, 1);
            if (go2) issue_a(0, kc2, b);
            ML::bar();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            quad(1, 1, wf1);
            ML::bar();
            // ---- phase 4
#ifdef ARX_STAMP
#ifdef ARX_STAMP_WAITS
            if (kt == 0) wst[0] = __builtin_readcyclecounter(); else if (kt == 1) wst[2] = __builtin_readcyclecounter();
#endif
#endif
            if (go2) { issue_b(1, kc2, b); wait_vmcnt<6>(); }
            else wait_vmcnt<0>();
#ifdef ARX_STAMP
#ifdef ARX_STAMP_WAITS
            if (kt == 0) wst[1] = __builtin_readcyclecounter(); else if (kt == 1) wst[3] = __builtin_readcyclecounter();
#endif
#endif
            ML::bar();
            quad(1, 0, wf0);
            if (go1 || wr == 0) ML::bar();
        }
        // Both wave groups run their epilogues TOGETHER (sharing the VALU, 12.6 k cycles for FFN-1) instead of one after the
        // other (8.1 k + 8.1 k, the second under a stalled partner): group 0 sits out the one interval in which group 1
        // finishes its last MFMA section, and group 1 gives the lead back after its epilogue.  Barrier counts stay equal.
        if (has_next && wr == 0) ML::bar();
#ifdef ARX_STAMP
        const unsigned long long pts1 = __builtin_readcyclecounter();
#endif
        pol.epilogue(acc, m0, n0, wr, wc, lane, smem + ML::STAGE_OFF + sbuf * EpiStage::BYTES);
        if (has_next && wr == 1) ML::bar();
#ifdef ARX_STAMP
        pol.stamp(orig, tid, pts0, pts1, wst, smem + ML::KT_STAMP_OFF);
#endif
        if (!has_next) break;
        orig = onext; m0 = m0n; n0 = n0n; ko = kon; sbuf ^= 1;
    }
}

// the encoder's linear layers: XCD-banded tile walk, LDS-staged epilogue vectors, fused epilogues (epi3.h)
template <int MODE, int KROT>
struct EncoderTilePolicy {
    static constexpr bool REBASE_W = false;
    static constexpr bool PERMUTE_B = true;                      // epilogue v3 expects the permuted column placement (Gemm8Phase PERM)
    TileWalk walk;
    const EpiParams& ep;
    int M, N, nk;
    __device__ __forceinline__ bool tile(int o, int& m0, int& n0, int& ko) const {
        int tm = 0, tn = 0;
        const bool ok = walk.coords(o, tm, tn);
        m0 = tm * 256; n0 = tn * 256; ko = (tn * KROT) % nk;
        return ok;
    }
    __device__ __forceinline__ void stage_issue(int m0, int n0, char* stage, int wid, int lane) const {
        epi_stage_issue<MODE>(ep, m0, n0, stage, wid, lane, N);
    }
    __device__ __forceinline__ void epilogue(const f32x4 (&acc)[4][8], int m0, int n0, int wr, int wc, int lane, const char* stage) const {
        // statistics modes: a 4-step residual window keeps the kernel under 256 VGPRs
        epilogue_store_v3<MODE, (MODE == EPI_RESID_STATS || MODE == EPI_LNRESID_STATS) ? 4 : 8>(acc, ep, m0, n0, wr, wc, lane, M, N, stage);
    }
#ifdef ARX_STAMP
    __device__ __forceinline__ void stamp(int orig, int tid, unsigned long long pts0, unsigned long long pts1, const unsigned long long (&wst)[4], const char* kts) const {
        if (ep.stamps && (tid == 0 || tid == 256)) {
            unsigned long long* o = ep.stamps + ((size_t)orig * 2 + (tid >> 8)) * 32;
            o[0] = pts0; o[1] = pts1; o[2] = __builtin_readcyclecounter(); o[3] = pts0;
            o[4] = wst[0]; o[5] = wst[1]; o[6] = wst[2]; o[7] = wst[3];
#pragma unroll 1
            for (int k = 0; k < 16; ++k) o[8 + k] = *reinterpret_cast<const volatile unsigned long long*>(kts + (tid >> 8) * 128 + k * 8);
        }
    }
#endif
};

template <int MODE, int KROT = 0>
__global__ __launch_bounds__(512) void gemm_8phase_persistent_kernel(const bf16_t* __restrict__ A, int64_t lda,
                                                                      const bf16_t* __restrict__ W, int64_t ldw,
                                                                      int M, int N, int K, int tiles_m, int tiles_n, EpiParams ep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef ARX_DEV_VARIANTS
    const TileWalk walk(tiles_m, tiles_n, K, ep.dev_bw);
#else
    const TileWalk walk(tiles_m, tiles_n, K);
#endif
    const EncoderTilePolicy<MODE, KROT> pol{walk, ep, M, N, K >> 6};
#ifdef ARX_DEV_VARIANTS
    if (ep.dev_stagger > 0) {        // probe: CUs of an XCD out of phase with one another
        const long long wait = (long long)((blockIdx.x >> 3) % ep.dev_slots) * ep.dev_stagger, t0 = clock64();
        while (clock64() - t0 < wait) __builtin_amdgcn_s_sleep(16);
    }
#endif
    gemm8_persistent_body<bf16_t>(A, lda, W, ldw, M, N, K, pol, smem);
}
