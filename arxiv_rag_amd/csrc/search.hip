// arx_topk_*: brute-force cosine top-k over an HBM-resident fp16 shard (C ABI in include/arx.h).
//
// Exact top-k without ever writing the Q x N score matrix and without per-lane candidate lists:
//   pass A  (the hot kernel, HBM-bound for Qb <~ 256): f16 MFMA GEMM queries x corpus^T whose epilogue
//           keeps only max-over-64-corpus-rows ("group max") per query  -> gmax[N/64][Qpad] f32.
//   pass B1 per query: the groups with the largest (gmax desc, group asc).  Every true top-k row lies in one
//           of the first k: a group ranked below k others is beaten k times (ties: lower group = lower row).
//           KSEL > k groups are kept (16 for k <= 10) so that groups whose maxima differ only by MFMA
//           accumulation-order rounding are all rescored and ranked by the exact pass.
//   pass B2 per query: f32 FMA-chain dot products of those KSEL*64 rows, final (score desc, row asc) top-k.
// The corpus is read once per query batch (pass B2 touches KSEL*64 rows per query, ~3 % extra at N = 10 M).
#include <math.h>
#include <stdlib.h>

#include "arx_common.h"
#include "gemm.h"
#include "gemm8.h"

#define GROUP_ROWS 64
#define KMAX 32                      // largest k
#define KSEL_SMALL 12                 // groups rescored when k <= 10 (k + 2: the certificate, not a margin, answers for exactness)
#define KSEL_BIG 36                   // groups rescored when k <= 32
#define SUPER 16                      // groups per super-group in the selection pass
#define SEL_SPLIT_WAVES 4            // waves per select block
#define QBATCH_MAX 1024              // queries per internal pass (bounds the gmax workspace)
#define AUX16_MAX_NQ 256              // fp16 pass: query batches up to this size (one 256-query tile) write aux words and take the single-row tail
#define TAIL_INBLOCK_MAX_SUPER 1024   // ... on shards of up to this many super-groups (1 M rows): there the block also selects for itself
#define SURV_CAP 256                 // int8 pipeline: rows at or above the threshold kept per query
#define CNT_QCOUNT 2                 // layout of the int8 candidate pipeline's counter block: see collect_pairs_kernel
#define CNT_QOVER (2 + QBATCH_MAX)
#define CNT_INTS (2 + 2 * QBATCH_MAX)

// Reductions over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48 hold the same query) WITHOUT the LDS: v_permlane16_swap /
// v_permlane32_swap of a value with itself leave, in every lane, the pair {own row's value, partner row's value} in the two
// results (in an order that depends on the lane: use them symmetrically).  The ds_bpermute shuffles they replace are LDS round
// trips in the tail of every pass-A block, where nothing overlaps them (0.15 ms per 39 k-block pass: profiles/r03).
__device__ __forceinline__ void rows16(float x, float& a, float& b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void rows32(float x, float& a, float& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
__device__ __forceinline__ float max_over_rows(float x) {
    float a, b;
    rows16(x, a, b); x = fmaxf(a, b);
    rows32(x, a, b); return fmaxf(a, b);
}
__device__ __forceinline__ int min_over_rows(int x) {
    auto r = __builtin_amdgcn_permlane16_swap((uint32_t)x, (uint32_t)x, false, false);
    x = min((int)r[0], (int)r[1]);
    r = __builtin_amdgcn_permlane32_swap((uint32_t)x, (uint32_t)x, false, false);
    return min((int)r[0], (int)r[1]);
}

// After the row reduction the four 16-lane rows of a wave hold the SAME per-query results: v[i] = the value of query i*16 + (lane & 15),
// i < MI.  Storing them from row 0 alone is MI store instructions of 64 B each — ten million 64-B writes per array and pass at 1 024
// queries, which cost 1.2 ms of a 9.8-ms pass (profiles/r03, store probe).  Here row rr stores block i = rr + 4 k: one instruction
// covers 64 consecutive queries = 256 contiguous bytes.
template <int MI, typename V>
__device__ __forceinline__ void store_query_row(V* __restrict__ dst, const V (&v)[MI], int m_first, int nq, int lane) {
    const int rr = lane >> 4;
#pragma unroll
    for (int k = 0; k < (MI + 3) / 4; ++k) {
        V x = v[4 * k];
#pragma unroll
        for (int t = 1; t < 4; ++t)
            if (4 * k + t < MI) {
                V y = v[4 * k + t];
                asm volatile("" : "+v"(y));          // opaque: keeps hipcc from turning the select chain into an indexed load of a SCRATCH copy of v
                x = (rr == t) ? y : x;
            }
        const int i = 4 * k + rr;
        const int m = m_first + i * 16 + (lane & 15);
        if (i < MI && m < nq) dst[m] = x;
    }
}

// ---- pass-A epilogues as functions of a wave's accumulator tile (shared by the per-tile kernels and the persistent one) --------------
// fp16 pass: the maximum pass-A score of the wave's 64 corpus rows, per query
template <int MI, int NI>
__device__ __forceinline__ void groupmax_epilogue_f16(const f32x4 (&acc)[NI][MI], float* __restrict__ gmax_row, int m_first, int nq, int lane) {
    float gm[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, acc[j][i][r]);
        gm[i] = max_over_rows(mx);
    }
    store_query_row<MI, float>(gmax_row, gm, m_first, nq, lane);
}

// fp16 pass, small query batches on small shards: beside the group maximum, the aux word (below) of the group — the position of the 4-ROW
// BLOCK (rows j*16 + 4*(lane>>4) + 0..3: what one lane holds of one MFMA block) that contains the largest pass-A score, and an upper bound
// on the pass-A score of every row of the group OUTSIDE that block.  The tail kernel then reads four consecutive fp16 rows of a selected
// group (6 KB) instead of its 64 (98 KB) whenever that bound is below the query's threshold: on a 625 k-row shard the rescoring of
// 12 x 64 rows per query — one CU pulling 1.2 MB — was 0.07 ms of a 0.26-ms batch (profiles/r03).  Granularity is the epilogue's price:
// tracking the arg-max ROW (first version, profiles/r04/tail_single_row_ab.md) keys every accumulator value — 4 instructions per element
// where the plain maximum takes 1 — and cost pass A 4 % at <= 64 queries and 10-17 % at 256; the 4-row block keys one value in four:
// 9 % at 256 queries, still 4 % at <= 64 (there it is not the VALU work; profiles/r04/tail_block4_ab.jsonl).
// The block's position travels in the low 6 bits of a value's float image (63 ulp either way, covered by the 8e-6 inflation of the
// bound); the group maximum itself is taken from the untouched values, so the selection and the certificate see what they saw before.
__device__ __forceinline__ uint32_t pack_aux(float ub2, int arg_row);
template <int MI, int NI>
__device__ __forceinline__ void groupmax_epilogue_f16_aux(const f32x4 (&acc)[NI][MI], float* __restrict__ gmax_row, uint32_t* __restrict__ aux_row,
                                                          int m_first, int nq, int lane) {
    float gm[MI];
    uint32_t ga[MI];
    const uint32_t lrow = (uint32_t)(lane >> 4) * 4u;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        float mx = -INFINITY, m1 = -INFINITY, m2 = -INFINITY;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const float bj = fmaxf(fmaxf(acc[j][i][0], acc[j][i][1]), fmaxf(acc[j][i][2], acc[j][i][3]));
            mx = fmaxf(mx, bj);
            const float key = __uint_as_float((__float_as_uint(bj) & ~63u) | ((uint32_t)(j * 16) + lrow));
            m2 = __builtin_amdgcn_fmed3f(m1, m2, key);
            m1 = fmaxf(m1, key);
        }
        {
            float a1, b1, a2, b2;
            rows16(m1, a1, b1); rows16(m2, a2, b2);
            m1 = fmaxf(a1, b1); m2 = fmaxf(fmaxf(a2, b2), fminf(a1, b1));
            rows32(m1, a1, b1); rows32(m2, a2, b2);
            m1 = fmaxf(a1, b1); m2 = fmaxf(fmaxf(a2, b2), fminf(a1, b1));
        }
        gm[i] = max_over_rows(mx);
        const float b2 = m2 + fabsf(m2) * 8.0e-6f + 1e-12f;
        ga[i] = pack_aux(b2, (int)(__float_as_uint(m1) & 63u));
    }
    store_query_row<MI, float>(gmax_row, gm, m_first, nq, lane);
    store_query_row<MI, uint32_t>(aux_row, ga, m_first, nq, lane);
}

// aux word per (query, group), beside the group's upper bound: the SECOND largest row bound of the group rounded UP to 16 bits (bf16
// image, still an upper bound) in the high half, the position (0..63) of the row that holds the largest bound in the low bits.  When the
// second bound is below a query's threshold, only that one row of the group can reach the top-k: the candidate step then reads ONE
// fp16 row (1.5 KB at D = 768) instead of the group's 64 (98 KB) — on unit rows that is the case for all but a handful of the ~150
// candidate groups per query, and it is what lets the int8 pass pay above the ridge point too (profiles/r03).
__device__ __forceinline__ uint32_t pack_aux(float ub2, int arg_row) {
    const uint32_t b = __float_as_uint(ub2);
    const uint32_t up = (b & 0x80000000u) ? (b & 0xFFFF0000u)                      // negative: dropping mantissa bits moves towards zero = up
                                          : ((b + 0xFFFFu) & 0xFFFF0000u);         // positive: round the magnitude up
    return up | (uint32_t)(arg_row & 63);
}

// int8 pass (see "int8 PRE-FILTER" below for the bound): upper bound of the group + aux word, per query.
//   this lane's 16 corpus rows: wave's group row j*16 + (lane>>4)*4 + r   (acc[j][i][r]); cm_of(j, c4) gives their (s_c, L1) pairs,
//   qm_of(i) the (s_q, L1) of query block i.
// ub(q, c) = s_q * [ s_c * (dot + cq) + X_c ],  cq = ceil(0.5001 L1(q8)) (an integer: added to the exact int32 dot),
// X_c = s_c * (0.5001 L1(c8) + 0.2501 D) inflated by 2^-22 (its two roundings).  Three instructions per element (integer add,
// convert — exact below 2^24 —, ONE fma = one rounding of the exact value); s_q > 0 and the rounding allowance are monotone, so
// the group's two largest bounds are reduced FIRST and scaled / inflated afterwards, on two values instead of sixteen.
template <int MI, int NI, typename CM, typename QM>
__device__ __forceinline__ void groupmax_epilogue_i8(const f32x4 (&acc)[NI][MI], CM cm_of, QM qm_of, int D, float* __restrict__ gmax_row,
                                                     uint32_t* __restrict__ aux_row, int m_first, int nq, int lane) {
    float sc[NI][4], xc[NI][4];
    const float dterm = 0.2501f * (float)D;
    const int lrow = (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        float2 c4[4];
        cm_of(j, c4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sc[j][r] = c4[r].x;
            const float x = c4[r].x * fmaf(0.5001f, c4[r].y, dterm);
            xc[j][r] = fmaf(x, 2.4e-7f, x);
        }
    }
    float gm[MI];
    uint32_t ga[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const float2 qm = qm_of(i);
        const int cqi = (int)ceilf(0.5001f * qm.y) + 1;
        // top-2 of the 16 bounds with the arg-max for free: the low 6 bits of each value's float image are REPLACED by the row's
        // position in the group (j*16 + r; the lane's 4-row offset is OR-ed in after the lane-local pass), which moves a value by at
        // most 63 ulp either way — covered by the 2^-17 allowance below — and lets v_max / v_med3 carry the index along.
        float m1 = -INFINITY, m2 = -INFINITY;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const i32x4 it = __builtin_bit_cast(i32x4, acc[j][i]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float u = fmaf(sc[j][r], (float)(it[r] + cqi), xc[j][r]);
                const float key = __uint_as_float((__float_as_uint(u) & ~63u) | (uint32_t)(j * 16 + r));
                m2 = __builtin_amdgcn_fmed3f(m1, m2, key);               // second largest of {m1 >= m2, key}
                m1 = fmaxf(m1, key);
            }
        }
        m1 = __uint_as_float(__float_as_uint(m1) | (uint32_t)lrow);
        m2 = __uint_as_float(__float_as_uint(m2) | (uint32_t)lrow);
        {
            float a1, b1, a2, b2;
            rows16(m1, a1, b1); rows16(m2, a2, b2);
            m1 = fmaxf(a1, b1); m2 = fmaxf(fmaxf(a2, b2), fminf(a1, b1));
            rows32(m1, a1, b1); rows32(m2, a2, b2);
            m1 = fmaxf(a1, b1); m2 = fmaxf(fmaxf(a2, b2), fminf(a1, b1));
        }
        const int i1 = (int)(__float_as_uint(m1) & 63u);
        float b1 = m1 * qm.x, b2 = m2 * qm.x;
        b1 += fabsf(b1) * 8.0e-6f + 1e-12f;                              // 63 ulp of the index bits (2^-17.4) + the fma's and this product's roundings
        b2 += fabsf(b2) * 8.0e-6f + 1e-12f;
        gm[i] = b1; ga[i] = pack_aux(b2, i1);
    }
    store_query_row<MI, float>(gmax_row, gm, m_first, nq, lane);
    store_query_row<MI, uint32_t>(aux_row, ga, m_first, nq, lane);
}
// (s_c, L1) pairs / (s_q, L1) pairs read from an LDS stage: [512 floats] the tile's 256 corpus rows, then [2 x 256] its queries
struct MetaFromLds {
    const float* meta; int wn, lrow;
    __device__ __forceinline__ void operator()(int j, float2 (&c4)[4]) const {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(meta + (wn * GROUP_ROWS + j * 16 + lrow) * 2);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(meta + (wn * GROUP_ROWS + j * 16 + lrow) * 2 + 4);
        c4[0] = float2{lo[0], lo[1]}; c4[1] = float2{lo[2], lo[3]}; c4[2] = float2{hi[0], hi[1]}; c4[3] = float2{hi[2], hi[3]};
    }
};
// one 4-byte LDS-DMA per thread stages the tile's corpus pairs, one more its query pairs (BM queries from m0)
template <int BM>
__device__ __forceinline__ void stage_i8_meta(const float2* __restrict__ cmeta, const float2* __restrict__ qmeta, int64_t n0, int64_t n_rows,
                                              int m0, int nq, float* meta, int tid) {
    {
        int64_t e = n0 * 2 + tid;                                          // dword index into cmeta; rows past the shard repeat its last row
        const int64_t last = n_rows * 2 - 2 + (tid & 1);
        e = e < last ? e : last;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(reinterpret_cast<const float*>(cmeta) + e), (lds_void_t*)(meta + (tid & ~63)), 4, 0, 0);
    }
    if (tid < 2 * BM) {                                                    // wave-uniform (BM is a multiple of 32)
        int e = m0 * 2 + tid;
        const int last = nq * 2 - 2 + (tid & 1);
        e = e < last ? e : last;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(reinterpret_cast<const float*>(qmeta) + e), (lds_void_t*)(meta + 512 + (tid & ~63)), 4, 0, 0);
    }
}

// ---------------------------------------------------------------------------------------------------
// pass A
template <int BM, bool GLDS>
__global__ __launch_bounds__(512) void search_groupmax_kernel(const f16_t* __restrict__ Q, int nq,
                                                               const f16_t* __restrict__ C, int64_t n_rows, int D,
                                                               int tiles_q, int tiles_n, float* __restrict__ gmax,
                                                               int64_t ldg, uint32_t* __restrict__ aux,
                                                               unsigned long long* __restrict__ zero_stats) {
    using ML = GemmMainloop<f16_t, BM, 256, 2, 4, GLDS, GLDS ? 3 : 0>;      // stagger + setprio as in the encoder GEMM
    // (the certificate counters of a call whose tail has no select kernel start at zero here: the tail runs after this grid)
    if (zero_stats && blockIdx.x == 0 && threadIdx.x < 2) zero_stats[threadIdx.x] = 0ull;
    static_assert(ML::TN == GROUP_ROWS, "one wave column = one group");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = xcd_remap(blockIdx.x, tiles_q * tiles_n);
    const int tile_q = t % tiles_q, tile_n = t / tiles_q;        // q fastest: blocks sharing a corpus tile are neighbours
    const int m0 = tile_q * BM;
    const int64_t n0 = (int64_t)tile_n * 256;
    f32x4 acc[ML::NI][ML::MI];
    // rows are addressed relative to the tile so 32-bit row math stays in range for any shard size
    const int rows_here = (int)((n_rows - n0) < 256 ? (n_rows - n0) : 256);
    // k rotation by query tile: the tiles_q blocks that share this corpus tile do not miss on the same lines at once
    if constexpr (BM == 256 && GLDS) {      // large query batches are MFMA-bound: the encoder's 4-phase schedule (gemm8.h)
#ifdef ARX_STAMP
        unsigned long long dummy_stamp;
        Gemm8Phase<f16_t, 2>::run(Q, D, nq, C + n0 * D, D, rows_here, D, m0, 0, smem, acc, tile_q * 2, dummy_stamp);
#else
        Gemm8Phase<f16_t, 2>::run(Q, D, nq, C + n0 * D, D, rows_here, D, m0, 0, smem, acc, tile_q * 2);
#endif
    } else
        ML::run(Q, D, nq, C + n0 * D, D, rows_here, D, m0, 0, smem, acc, tile_q * 2);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wm = wid / 4, wn = wid % 4;
    if (wn * GROUP_ROWS >= rows_here) return;
    const int64_t g = (n0 >> 6) + wn;
    if (aux) groupmax_epilogue_f16_aux<ML::MI, ML::NI>(acc, gmax + g * ldg, aux + g * ldg, m0 + wm * ML::TM, nq, lane);
    else groupmax_epilogue_f16<ML::MI, ML::NI>(acc, gmax + g * ldg, m0 + wm * ML::TM, nq, lane);
}

// ---------------------------------------------------------------------------------------------------
// int8 PRE-FILTER (optional second representation of the shard, `arx_topk_build_i8`): pass A over int8 rows — half the bytes in the
// HBM-bound regime, twice the MFMA rate in the matrix-bound one — with nothing given up: what it writes per (query, 64-row group) is
// a rigorous UPPER BOUND on the true fp16 score of every row of the group, and the exact passes (select, fp32 rescoring of the fp16
// rows, certificate) run unchanged on it.
//   row x (fp16, exact) = s * x8 + e,  s = max|x| / 127,  x8 = rint(x / s),  |e_i| <= 0.5001 s      (quantize_rows_i8_kernel; the 0.0001
//   absorbs the fp32 division).  For a query q = s_q q8 + f and a corpus row c = s_c c8 + e:
//       q.c = s_q s_c (q8.c8) + q^.e + f.c^ + f.e ,   |q^.e| <= 0.5001 s_c |q^|_1 ,  |f.c^| <= 0.5001 s_q |c^|_1 ,  |f.e| <= 0.2501 D s_q s_c
//   with |q^|_1 = s_q L1(q8), |c^|_1 = s_c L1(c8):   q.c <= s_q s_c ( q8.c8 + 0.5001 (L1(q8) + L1(c8)) + 0.2501 D ) =: ub.
// q8.c8 is an exact int32 (|.| <= 127^2 D < 2^24: exact as fp32 too); how the bound is evaluated and its roundings covered: see
// groupmax_epilogue_i8.  The certificate then reads: every unscored row's TRUE score <= U; U < s_k - tau => the answer is exact.  With
// unit rows the slack is ~0.026, so ~150 groups per query reach the threshold; the candidate pipeline below (collect_pairs_kernel ->
// pair_rescore_kernel -> merge_survivors_kernel) rescoring ONE row of almost every such group (aux word) settles them.
__global__ __launch_bounds__(256) void quantize_rows_i8_kernel(const f16_t* __restrict__ X, int64_t n_rows, int D, int8_t* __restrict__ X8,
                                                                float2* __restrict__ meta, unsigned long long* __restrict__ zero_stats) {
    // (quantising a QUERY batch is the first kernel of an int8 search: the call's certificate counters start at zero here)
    if (zero_stats && blockIdx.x == 0 && threadIdx.x < 2) zero_stats[threadIdx.x] = 0ull;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int per = D / 64;                       // D % 128 == 0: per even, <= 16 (D <= 1024)
    float v[16];
    float amax = 0.f;
    const f16_t* x = X + row * D + lane * per;
    for (int e = 0; e < per; e += 2) {
        const uint32_t w2 = *reinterpret_cast<const uint32_t*>(x + e);
        f16_t h0, h1;
        __builtin_memcpy(&h0, &w2, 2); __builtin_memcpy(&h1, reinterpret_cast<const char*>(&w2) + 2, 2);
        v[e] = (float)h0; v[e + 1] = (float)h1;
        amax = fmaxf(amax, fmaxf(fabsf(v[e]), fabsf(v[e + 1])));
    }
    amax = wave_max(amax);
    const float s = amax / 127.0f;                 // an all-zero row: s = 0, every x8 = 0, upper bound 0
    const float inv_s = amax > 0.f ? 127.0f / amax : 0.f;
    float l1 = 0.f;
    int8_t* o = X8 + row * D + lane * per;
    for (int e = 0; e < per; e += 2) {
        const float q0 = fminf(fmaxf(rintf(v[e] * inv_s), -127.f), 127.f), q1 = fminf(fmaxf(rintf(v[e + 1] * inv_s), -127.f), 127.f);
        l1 += fabsf(q0) + fabsf(q1);
        const uint16_t pk = (uint16_t)((uint8_t)(int8_t)(int)q0) | (uint16_t)((uint16_t)(uint8_t)(int8_t)(int)q1 << 8);
        *reinterpret_cast<uint16_t*>(o + e) = pk;
    }
    l1 = wave_sum(l1);
    if (lane == 0) meta[row] = float2{s, l1};
}

template <int BM, bool GLDS>
__global__ __launch_bounds__(512) void search_groupmax_i8_kernel(const int8_t* __restrict__ Q8, const float2* __restrict__ qmeta, int nq,
                                                                  const int8_t* __restrict__ C8, const float2* __restrict__ cmeta,
                                                                  int64_t n_rows, int D, int tiles_q, int tiles_n,
                                                                  float* __restrict__ gmax, uint32_t* __restrict__ aux, int64_t ldg) {
    using ML = GemmMainloop<i8pair_t, BM, 256, 2, 4, GLDS, GLDS ? 3 : 0>;
    static_assert(ML::TN == GROUP_ROWS, "one wave column = one group");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = xcd_remap(blockIdx.x, tiles_q * tiles_n);
    const int tile_q = t % tiles_q, tile_n = t / tiles_q;
    const int m0 = tile_q * BM;
    const int64_t n0 = (int64_t)tile_n * 256;
    const int Dh = D >> 1;                        // the int8 rows as rows of D/2 two-byte elements: the f16 kernel's byte geometry
    const i8pair_t* Q = reinterpret_cast<const i8pair_t*>(Q8);
    const i8pair_t* C = reinterpret_cast<const i8pair_t*>(C8);
    f32x4 acc[ML::NI][ML::MI];
    const int rows_here = (int)((n_rows - n0) < 256 ? (n_rows - n0) : 256);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wm = wid / 4, wn = wid % 4;
    // The row / query constants of the epilogue (scale and L1 norm of the tile's 256 corpus rows and BM queries: 2 KB + BM x 8 B) are
    // requested BEFORE the main loop: loaded after it they are an exposed L2 round trip at the end of every tile, with nothing left to
    // overlap it.  BM >= 128: by one 4-byte LDS-DMA per thread into 4 KB behind the k-tile buffers (no registers held across the loop;
    // they are the oldest loads of the tile, retired by the loop's own waits and barriers).  BM = 64: the block must stay at 80 KB of
    // LDS (two blocks per CU), so the values ride in 36 registers, which that kernel can spare.
    constexpr bool META_LDS = BM >= 128;
    constexpr int MAIN_BYTES = (BM == 256 && GLDS) ? Gemm8Phase<i8pair_t, 2>::STAGE_OFF : ML::SMEM_BYTES;
    float* const meta = reinterpret_cast<float*>(smem + MAIN_BYTES);          // [512] corpus (s, L1) pairs, then [2 BM] query pairs
    const int lrow = (lane >> 4) * 4;
    // (the 16 corpus pairs a lane's accumulators belong to are the same for the 16 lanes of a row of the wave: lane t of the row loads ONE
    // pair — row (t>>2)*16 + lrow + (t&3) of the group — and the epilogue fetches the sixteen by ds_bpermute.  Sixteen 8-byte loads per lane
    // were a third of all the vector-memory requests of a D = 768 tile, half of them at D = 384: 5.8 / 5.0 TB/s where the fp16 pass, the
    // same bytes per tile at D = 384, reaches 6.3; after: 6.0-6.15 / 5.3-5.5, and 5.5 from 5.25 at 64 queries.  A STREAMED form of this kernel —
    // a block walking four corpus tiles with its loads running through the tile boundaries and the epilogue under the next tile's first
    // k-tile — was built beside it and measured: +0-2 % at D = 768, +6-9 % at D = 384, -2 % for the fp16 rows; not kept.
    // profiles/r04/pass_a_narrow_int8_ab.md)
    float2 cmr = float2{0.f, 0.f}, qmr[META_LDS ? 1 : ML::MI];
    if constexpr (META_LDS) {
        stage_i8_meta<BM>(cmeta, qmeta, n0, n_rows, m0, nq, meta, threadIdx.x);
    } else {
        {
            const int t = lane & 15;
            const int64_t n = n0 + wn * GROUP_ROWS + (t >> 2) * 16 + lrow + (t & 3);
            cmr = cmeta[n < n_rows ? n : n_rows - 1];
        }
#pragma unroll
        for (int i = 0; i < ML::MI; ++i) {
            const int m = m0 + wm * ML::TM + i * 16 + (lane & 15);
            qmr[i] = qmeta[m < nq ? m : nq - 1];
        }
    }
    if constexpr (BM == 256 && GLDS) {
#ifdef ARX_STAMP
        unsigned long long dummy_stamp;
        Gemm8Phase<i8pair_t, 2>::run(Q, Dh, nq, C + n0 * Dh, Dh, rows_here, Dh, m0, 0, smem, acc, tile_q * 2, dummy_stamp);
#else
        Gemm8Phase<i8pair_t, 2>::run(Q, Dh, nq, C + n0 * Dh, Dh, rows_here, Dh, m0, 0, smem, acc, tile_q * 2);
#endif
    } else
        ML::run(Q, Dh, nq, C + n0 * Dh, Dh, rows_here, Dh, m0, 0, smem, acc, tile_q * 2);
    if (wn * GROUP_ROWS >= rows_here) return;
    const int64_t g = (n0 >> 6) + wn;
    if constexpr (META_LDS) {
        const float* qmeta_l = meta + 512 + (wm * ML::TM + (lane & 15)) * 2;
        groupmax_epilogue_i8<ML::MI, ML::NI>(acc, MetaFromLds{meta, wn, lrow},
                                             [&](int i) { return *reinterpret_cast<const float2*>(qmeta_l + i * 32); },
                                             D, gmax + g * ldg, aux + g * ldg, m0 + wm * ML::TM, nq, lane);
    } else {
        groupmax_epilogue_i8<ML::MI, ML::NI>(acc, [&](int j, float2 (&c4)[4]) {
#pragma unroll
                                                 for (int r = 0; r < 4; ++r) {
                                                     const int src = (lane & 48) | (j * 4 + r);
                                                     c4[r] = float2{__shfl(cmr.x, src), __shfl(cmr.y, src)};
                                                 }
                                             },
                                             [&](int i) { return qmr[i]; }, D, gmax + g * ldg, aux + g * ldg, m0 + wm * ML::TM, nq, lane);
    }
}

// ---- pass A, persistent form (>= 256 queries, even number of k-tiles): gemm8.h's persistent 4-phase loop with the pass-A epilogues.
// Above the ridge point a 256 x 256 x D tile is SHORT (12 k-tiles of f16, 6 of int8 at D = 768): the per-tile kernel pays the first
// loads' latency, an idle matrix pipe during the epilogue and a block launch per tile — 29 k cycles per int8 tile against 6 k of matrix
// work (profiles/r03).  Here one block per CU walks its tiles with the operand stream running through the tile boundaries.
// Tile order: block b belongs to XCD b % 8 and takes corpus tiles = b % 8 (mod 8); inside an XCD the sequence is query-tile fastest, so
// the (up to four) blocks that read one corpus tile are neighbours in time on ONE L2.
template <bool I8, bool AUX16 = false>
struct SearchTilePolicy {
    static constexpr bool REBASE_W = true;
    static constexpr bool PERMUTE_B = false;                     // a group's arg-max row is a position inside the tile: corpus rows stay in order
    int tiles_q, tiles_n, nq, D;
    int64_t n_rows, ldg;
    float* gmax;
    uint32_t* aux;
    const float2* qmeta; const float2* cmeta;
    __device__ __forceinline__ bool tile(int o, int& m0, int& n0, int& ko) const {
        const int x = o & 7, L = o >> 3;
        const int tq = L % tiles_q, tn = (L / tiles_q) * 8 + x;
        m0 = tq * 256; n0 = tn * 256; ko = 0;
        return tn < tiles_n;
    }
    __device__ __forceinline__ void stage_issue(int m0, int n0, char* stage, int wid, int lane) const {
        if constexpr (I8) stage_i8_meta<256>(cmeta, qmeta, n0, n_rows, m0, nq, reinterpret_cast<float*>(stage), wid * 64 + lane);
    }
    __device__ __forceinline__ void epilogue(const f32x4 (&acc)[4][8], int m0, int n0, int wr, int wc, int lane, const char* stage) const {
        if ((int64_t)n0 + wc * GROUP_ROWS >= n_rows) return;             // the wave's group lies past the shard (wave-uniform)
        const int64_t g = ((int64_t)n0 >> 6) + wc;
        if constexpr (I8) {
            const float* meta = reinterpret_cast<const float*>(stage);
            const float* qmeta_l = meta + 512 + (wr * 128 + (lane & 15)) * 2;
            groupmax_epilogue_i8<8, 4>(acc, MetaFromLds{meta, wc, (lane >> 4) * 4},
                                       [&](int i) { return *reinterpret_cast<const float2*>(qmeta_l + i * 32); },
                                       D, gmax + g * ldg, aux + g * ldg, m0 + wr * 128, nq, lane);
        } else if constexpr (AUX16)
            groupmax_epilogue_f16_aux<8, 4>(acc, gmax + g * ldg, aux + g * ldg, m0 + wr * 128, nq, lane);
        else
            groupmax_epilogue_f16<8, 4>(acc, gmax + g * ldg, m0 + wr * 128, nq, lane);
    }
#ifdef ARX_STAMP
    __device__ __forceinline__ void stamp(int, int, unsigned long long, unsigned long long, const unsigned long long (&)[4], const char*) const {}
#endif
};

template <typename T, bool I8, bool AUX16>
__global__ __launch_bounds__(512) void search_groupmax_persistent_kernel(const T* __restrict__ Q, int nq, const T* __restrict__ C, int64_t n_rows,
                                                                          int Kt /* row length in T elements */, int D, int tiles_q, int tiles_n,
                                                                          const float2* __restrict__ qmeta, const float2* __restrict__ cmeta,
                                                                          float* __restrict__ gmax, uint32_t* __restrict__ aux, int64_t ldg,
                                                                          unsigned long long* __restrict__ zero_stats) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (zero_stats && blockIdx.x == 0 && threadIdx.x < 2) zero_stats[threadIdx.x] = 0ull;
    const SearchTilePolicy<I8, AUX16> pol{tiles_q, tiles_n, nq, D, n_rows, ldg, gmax, aux, qmeta, cmeta};
    gemm8_persistent_body<T>(Q, Kt, C, Kt, nq, (int)n_rows, Kt, pol, smem);
}

// ---------------------------------------------------------------------------------------------------
// sorted insert into a register-resident top-K list (score desc, id asc on ties; new element has the
// larger id when scanning in increasing id order, so strict '>' keeps the earlier one first)
template <int K>
__device__ __forceinline__ void topk_insert(float (&s)[K], int64_t (&id)[K], float v, int64_t vi) {
#pragma unroll
    for (int p = 0; p < K; ++p) {
        const bool take = (v > s[p]) || (v == s[p] && vi < id[p]);
        const float ts = s[p]; const int64_t ti = id[p];
        s[p] = take ? v : ts;  id[p] = take ? vi : ti;
        v = take ? ts : v;     vi = take ? ti : vi;
    }
}

// pass B1, stage 1: lane = query, each wave scans a slice of groups (coalesced over queries) keeping its
// top-K groups in registers; the block's 4 waves then merge through LDS -> ONE list per (block slice, query).
// grid (Qpad/64, nsplit), block 256.  out: part_s/part_g [nsplit][Qpad][K]
template <int K>
__global__ __launch_bounds__(256) void select_groups_kernel(const float* __restrict__ gmax, int64_t ldg, int64_t n_groups,
                                                             int64_t n_real, int nq, int nsplit,
                                                             float* __restrict__ part_s, int32_t* __restrict__ part_g,
                                                             unsigned long long* __restrict__ zero_stats) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // the certificate counters of this search call start at zero (first internal pass only; the rescore kernels that add to them run
    // after this one on the stream): saves a 16-byte memset node per call, which a 0.25-ms small-shard search can see
    if (zero_stats && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 2) zero_stats[threadIdx.x] = 0ull;
    float* ls = reinterpret_cast<float*>(smem);                     // [4][K][64]
    int32_t* lg = reinterpret_cast<int32_t*>(smem + 4 * K * 64 * 4);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int q = blockIdx.x * 64 + lane;
    const int slice = blockIdx.y * SEL_SPLIT_WAVES + w, nslices = nsplit * SEL_SPLIT_WAVES;
    const int64_t per = (n_groups + nslices - 1) / nslices;
    const int64_t g0 = slice * per, g1 = (g0 + per < n_groups) ? g0 + per : n_groups;
    float s[K]; int64_t id[K];
#pragma unroll
    for (int p = 0; p < K; ++p) { s[p] = -INFINITY; id[p] = 0x7fffffff; }
    const int qc = q < nq ? q : nq - 1;                                // clamp: every lane loads (results unused)
    const float* col = gmax + qc;
    // g0..g1 count SUPER-groups of SUPER consecutive groups: one max per super-group (no divergence), then
    // one threshold test per super-group.  The true top-K groups lie inside the top-K super-groups (same
    // "beaten K times" argument one level up); the rescore kernel expands them again.
    for (int64_t sg = g0; sg < g1; ++sg) {
        float v[SUPER];
#pragma unroll
        for (int u = 0; u < SUPER; ++u) {
            const int64_t g = sg * SUPER + u;
            v[u] = col[(g < n_real ? g : n_real - 1) * ldg];
        }
        float m = v[0];
#pragma unroll
        for (int u = 1; u < SUPER; ++u) m = fmaxf(m, v[u]);
        if (m > s[K - 1]) topk_insert<K>(s, id, m, sg);                // increasing sg: ties keep the lower one
    }
#pragma unroll
    for (int p = 0; p < K; ++p) { ls[(w * K + p) * 64 + lane] = s[p]; lg[(w * K + p) * 64 + lane] = (int32_t)id[p]; }
    __syncthreads();
    if (w == 0) {
        int ptr[SEL_SPLIT_WAVES] = {0, 0, 0, 0};
        const int64_t o = ((int64_t)blockIdx.y * ldg + q) * K;
        for (int p = 0; p < K; ++p) {
            float bs = -INFINITY; int bg = 0x7fffffff; int bw = 0;
#pragma unroll
            for (int ww = 0; ww < SEL_SPLIT_WAVES; ++ww) {
                const int pp = ptr[ww] < K ? ptr[ww] : K - 1;
                const float cs = ptr[ww] < K ? ls[(ww * K + pp) * 64 + lane] : -INFINITY;
                const int cg = ptr[ww] < K ? lg[(ww * K + pp) * 64 + lane] : 0x7fffffff;
                if (cs > bs || (cs == bs && cg < bg)) { bs = cs; bg = cg; bw = ww; }
            }
#pragma unroll
            for (int ww = 0; ww < SEL_SPLIT_WAVES; ++ww) ptr[ww] += (ww == bw) ? 1 : 0;
            part_s[o + p] = bs;
            part_g[o + p] = (bg == 0x7fffffff) ? -1 : bg;
        }
    }
}

// All-reduce of the best (score desc, id asc) candidate over the 64 lanes WITHOUT the LDS: four DPP exchanges inside a 16-lane row
// (quad permutes, half-row and row mirrors: every exchange pairs a lane with one that holds a different partial result, which is all an
// all-reduce needs) and the two permlane swaps across rows.  Six steps of three register moves + a compare/select each; the
// ds_bpermute form it replaces made 24 LDS round trips per extracted element, and a rescore block extracts ~55 of them one after the
// other (profiles/r03: rescore 0.098 -> see DESIGN).  The order is total (ids are distinct; an empty slot is (-inf, INT64_MAX)), so both
// partners of an exchange keep the same winner.
__device__ __forceinline__ bool cand_better(float os, int64_t oi, float ws, int64_t wi) { return os > ws || (os == ws && oi < wi); }
template <int CTRL>
__device__ __forceinline__ void argbest_dpp(float& ws, int64_t& wi) {
    const int s_ = __builtin_amdgcn_update_dpp(0, __float_as_int(ws), CTRL, 0xf, 0xf, true);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)wi, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)((uint64_t)wi >> 32), CTRL, 0xf, 0xf, true);
    const float os = __int_as_float(s_);
    const int64_t oi = (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
    const bool take = cand_better(os, oi, ws, wi);
    ws = take ? os : ws; wi = take ? oi : wi;
}
template <bool ROW32>
__device__ __forceinline__ void argbest_rows(float& ws, int64_t& wi) {
    const uint32_t s_ = __float_as_uint(ws), lo = (uint32_t)wi, hi = (uint32_t)((uint64_t)wi >> 32);
    const auto rs = ROW32 ? __builtin_amdgcn_permlane32_swap(s_, s_, false, false) : __builtin_amdgcn_permlane16_swap(s_, s_, false, false);
    const auto rl = ROW32 ? __builtin_amdgcn_permlane32_swap(lo, lo, false, false) : __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto rh = ROW32 ? __builtin_amdgcn_permlane32_swap(hi, hi, false, false) : __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const float as = __uint_as_float(rs[0]), bs = __uint_as_float(rs[1]);
    const int64_t ai = (int64_t)(((uint64_t)rh[0] << 32) | rl[0]), bi = (int64_t)(((uint64_t)rh[1] << 32) | rl[1]);
    const bool take = cand_better(bs, bi, as, ai);
    ws = take ? bs : as; wi = take ? bi : ai;
}
__device__ __forceinline__ void wave_argbest(float& ws, int64_t& wi) {
    argbest_dpp<0xB1>(ws, wi);        // quad_perm [1,0,3,2]
    argbest_dpp<0x4E>(ws, wi);        // quad_perm [2,3,0,1]
    argbest_dpp<0x141>(ws, wi);       // row_half_mirror
    argbest_dpp<0x140>(ws, wi);       // row_mirror
    argbest_rows<false>(ws, wi);
    argbest_rows<true>(ws, wi);
}

// Wave-synchronous top-k: every lane holds R candidates in registers; k rounds of (lane-local best, wave all-reduce of the best,
// winner retires its candidate).  No LDS, no block barrier.  Order: score desc, id asc; id < 0 = empty.
template <int R>
__device__ __forceinline__ void wave_topk(float (&s)[R], int64_t (&id)[R], int k, int lane, float* out_s, int64_t* out_i) {
    for (int r = 0; r < k; ++r) {
        float bs = -INFINITY; int64_t bi = INT64_MAX; int bj = -1;
#pragma unroll
        for (int j = 0; j < R; ++j)
            if (id[j] >= 0 && (bj < 0 || s[j] > bs || (s[j] == bs && id[j] < bi))) { bs = s[j]; bi = id[j]; bj = j; }
        float ws = bj >= 0 ? bs : -INFINITY; int64_t wi = bj >= 0 ? bi : INT64_MAX;
        wave_argbest(ws, wi);
        const bool found = wi != INT64_MAX;
#pragma unroll
        for (int j = 0; j < R; ++j)
            if (found && j == bj && bi == wi) id[j] = -1;          // ids are distinct: only the winner's lane holds it
        if (lane == 0) { out_s[r] = found ? ws : -INFINITY; out_i[r] = found ? wi : -1; }
    }
}

// exact score of corpus row `row` against the query row staged in LDS: 8 lanes per row (l8 = lane & 7), two FMA chains per
// lane over its 16-B chunks, then a 3-step butterfly -> every lane of the octet holds the sum.  The ONE definition of a score
// in this file: pass B2 and the certificate's fallback both rank by it.
__device__ __forceinline__ float exact_row_score(const f16_t* __restrict__ crow, const f16_t* qs, int nch, int l8, bool ok) {
    float a0 = 0.f, a1 = 0.f;
    // chunks l8, l8 + 8, ... in ascending order, whatever the batching below: the sum's order (hence its bits) is fixed; the batches
    // only decide how many of the row's 16-B loads are in flight at once (six: a 768-d row is two dependent round trips instead of three)
    auto fma8 = [&](const f16x8& cv, const f16x8& qq) {
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            a0 = fmaf((float)cv[e], (float)qq[e], a0);
            a1 = fmaf((float)cv[e + 1], (float)qq[e + 1], a1);
        }
    };
    if (ok) {
        int ch = l8;
#pragma unroll 1
        for (; ch + 40 < nch; ch += 48) {
            f16x8 cv[6];
#pragma unroll
            for (int u = 0; u < 6; ++u) cv[u] = *reinterpret_cast<const f16x8*>(crow + (ch + 8 * u) * 8);
#pragma unroll
            for (int u = 0; u < 6; ++u) fma8(cv[u], *reinterpret_cast<const f16x8*>(qs + (ch + 8 * u) * 8));
        }
#pragma unroll 1
        for (; ch + 8 < nch; ch += 16) {
            f16x8 cv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) cv[u] = *reinterpret_cast<const f16x8*>(crow + (ch + 8 * u) * 8);
#pragma unroll
            for (int u = 0; u < 2; ++u) fma8(cv[u], *reinterpret_cast<const f16x8*>(qs + (ch + 8 * u) * 8));
        }
        for (; ch < nch; ch += 8) fma8(*reinterpret_cast<const f16x8*>(crow + ch * 8), *reinterpret_cast<const f16x8*>(qs + ch * 8));
    }
    float a = a0 + a1;
    a += __shfl_xor(a, 4); a += __shfl_xor(a, 2); a += __shfl_xor(a, 1);
    return a;
}

// pass B1 stage 2 + pass B2 + exactness certificate: one block of NT/64 waves per query.
//   (1) each wave reduces its slice of the nslices*K partial super-groups to K+1, wave 0 reduces those to K+1: K selected
//       super-groups + the best one left out
//   (2) wave 0 expands to K*SUPER groups (their gmax), reduces to the K best groups + the best one left out
//   (3) wave w rescoring group w (, w+NW, ...): 8 lanes per corpus row, query row staged in LDS; keeps its top-k
//   (4) wave 0 reduces NW*k -> k
//   (5) CERTIFICATE.  U = an upper bound on the pass-A score of every row that was NOT rescored = max(best super-group left out
//       in (1), the K-th kept value of every select slice (bounds what that slice dropped), best group left out in (2)).
//       Pass A (f16 MFMA, f32 accumulate) and pass B2 (f32 FMA chains) both approximate the real dot product, within
//       eps_A + eps_B <= tau = tau_scale * |q|_2 for corpus rows of norm <= 1 + 2^-9 (unit rows, as the encoder writes them; tau_scale =
//       (0.3125 D + 4) 2^-24: 8 roundings per 32-deep MFMA step, D/16 + 4 for the FMA chains and the butterfly).  If
//       U < s_k - tau no row outside the rescored groups can belong to the top-k: the common case, nothing more to do.
//       Otherwise (near-ties across more than K groups: duplicate / boilerplate chunks, or rounding at the boundary) the block
//       scans this query's gmax column and rescoring EVERY group with gmax >= s_k - tau that was not rescored yet, then
//       merges; the answer is then exact whatever the data.  No query is ever answered from an uncertified selection.
template <int K, int NT>
__global__ __launch_bounds__(NT) void rescore_kernel(const float* __restrict__ part_s, const int32_t* __restrict__ part_g,
                                                      int nslices, int64_t ldg, const float* __restrict__ gmax,
                                                      int64_t n_groups, const f16_t* __restrict__ Q,
                                                      const f16_t* __restrict__ C, int64_t n_rows, int D, int k,
                                                      float* __restrict__ out_s, int64_t* __restrict__ out_i,
                                                      int64_t idx_base, float tau_scale, int debug_drop,
                                                      unsigned long long* __restrict__ stats,
                                                      float* __restrict__ thr_out, int32_t* __restrict__ selg_out,
                                                      const int32_t* __restrict__ only_if, int* __restrict__ cand_counters,
                                                      int* __restrict__ cand_nsurv) {
    // thr_out / selg_out (int8 pre-filter): COLLECT mode — write the provisional top-k, the threshold s_k - tau
    // and the K rescored groups, and leave the rest to collect_pairs / pair_rescore / merge_survivors (the in-block fallback below
    // walks this query's gmax column from ONE CU: fine for the rare uncertified query, far too slow when every query needs it).
    // only_if: run only for the queries it flags (the overflow re-run of that pipeline).
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (only_if && !only_if[blockIdx.x]) return;
    constexpr int NW = NT / 64;
    constexpr int K1 = K + 1;
    constexpr int R1 = (256 * K + NT - 1) / NT;              // candidates per lane in stage (1): nslices <= 256
    constexpr int KK = K1 > KMAX ? K1 : KMAX;
    static_assert(3 * K >= NW, "fallback scratch: one 64-float row per wave inside the score + id buffers (K*64*12 bytes)");
    __shared__ float w_s[NW][KK];
    __shared__ int64_t w_i[NW][KK];
    __shared__ int32_t sel_g[K];
    __shared__ float w_vb[NW];
    __shared__ float sh_u, sh_thr, sh_qn;
    __shared__ int sh_flag;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    f16_t* qs = reinterpret_cast<f16_t*>(smem);               // [D] query row
    float* gs = reinterpret_cast<float*>(smem + (((size_t)D * 2 + 15) & ~(size_t)15));          // [K*64] row scores
    int64_t* gi_ = reinterpret_cast<int64_t*>(reinterpret_cast<char*>(gs) + K * GROUP_ROWS * 4);  // [K*64] row ids
    for (int i = tid; i < (D >> 3); i += NT)
        reinterpret_cast<u32x4*>(qs)[i] = reinterpret_cast<const u32x4*>(Q + (int64_t)q * D)[i];
    // (1) partial super-groups -> K best (+ the best one left out)
    {
        const int ncand = nslices * K;
        float s[R1]; int64_t id[R1];
        float vb = -INFINITY;
#pragma unroll
        for (int j = 0; j < R1; ++j) {
            const int i = (w * R1 + j) * 64 + lane;            // wave w owns a contiguous range
            s[j] = -INFINITY; id[j] = -1;
            if (i < ncand) {
                const int sl = i / K, p = i - sl * K;
                const int64_t o = ((int64_t)sl * ldg + q) * K + p;
                s[j] = part_s[o]; id[j] = part_g[o];
                if (p == K - 1 && id[j] >= 0) vb = fmaxf(vb, s[j]);     // whatever this slice dropped scores <= its K-th kept value
            }
        }
        wave_topk<R1>(s, id, K1, lane, w_s[w], w_i[w]);
        vb = wave_max(vb);
        if (lane == 0) w_vb[w] = vb;
    }
    __syncthreads();
    if (w == 0) {
        constexpr int R2 = (NW * K1 + 63) / 64;
        float s[R2]; int64_t id[R2];
#pragma unroll
        for (int j = 0; j < R2; ++j) {
            const int i = j * 64 + lane;
            s[j] = i < NW * K1 ? w_s[i / K1][i % K1] : -INFINITY;
            id[j] = i < NW * K1 ? w_i[i / K1][i % K1] : -1;
        }
        wave_topk<R2>(s, id, K1, lane, gs, gi_);               // K best super-groups -> gs/gi_[0..K), best left out -> [K]
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float u = gi_[K] >= 0 ? gs[K] : -INFINITY;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) u = fmaxf(u, w_vb[ww]);
        // (2) expand to K*SUPER groups, reduce to the K best groups (+ the best one left out)
        constexpr int R3 = (K * SUPER + 63) / 64;
        float s3[R3]; int64_t id3[R3];
#pragma unroll
        for (int j = 0; j < R3; ++j) {
            const int i = j * 64 + lane;
            s3[j] = -INFINITY; id3[j] = -1;
            if (i < K * SUPER) {
                const int64_t sg = gi_[i / SUPER];
                const int64_t g = sg * SUPER + (i % SUPER);
                if (sg >= 0 && g < n_groups) { s3[j] = gmax[g * ldg + q]; id3[j] = g; }
            }
        }
        wave_topk<R3>(s3, id3, K1, lane, w_s[0], w_i[0]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (w_i[0][K] >= 0) u = fmaxf(u, w_s[0][K]);
        // test hook (ARX_TOPK_DEBUG_DROP): forget the best group, as if selection had missed it; the certificate must recover it
        if (debug_drop && w_i[0][0] >= 0) u = fmaxf(u, w_s[0][0]);
        if (lane < K) sel_g[lane] = !debug_drop ? (int32_t)w_i[0][lane] : (lane + 1 < K ? (int32_t)w_i[0][lane + 1] : -1);
        // |q|_2 for the certificate's tolerance
        float qq = 0.f;
        for (int i = lane; i < D; i += 64) { const float v = (float)qs[i]; qq = fmaf(v, v, qq); }
        qq = wave_sum(qq);
        if (lane == 0) { sh_u = u; sh_qn = sqrtf(qq); }
    }
    __syncthreads();
    // (3) exact scores: 8 lanes per corpus row, 8 rows per wave and step; the K x 64 rows are dealt to ALL the block's waves (a 16-wave
    // block scores its 768 rows in 6 steps; one group per wave left four waves idle for 8)
    const int nch = D >> 3, l8 = lane & 7, rsub = lane >> 3;
    constexpr int GPW = (K + NW - 1) / NW;                    // groups per wave in the per-wave top-k below
    for (int t0 = w * 8; t0 < K * GROUP_ROWS; t0 += NW * 8) {
        const int t = t0 + rsub, gidx = t >> 6, rr = t & 63;
        const int gsel = sel_g[gidx];
        const int64_t row = (int64_t)gsel * GROUP_ROWS + rr;
        const bool ok = gsel >= 0 && row < n_rows;
        const float a = exact_row_score(C + row * D, qs, nch, l8, ok);
        if (l8 == 0) { gs[t] = ok ? a : -INFINITY; gi_[t] = ok ? row : -1; }
    }
    __syncthreads();
    // each wave: top-k of the rows of its groups (read back one row per lane)
    {
        float s[GPW]; int64_t id[GPW];
#pragma unroll
        for (int gq = 0; gq < GPW; ++gq) {
            const int gidx = w + gq * NW;
            s[gq] = gidx < K ? gs[gidx * GROUP_ROWS + lane] : -INFINITY;
            id[gq] = gidx < K ? gi_[gidx * GROUP_ROWS + lane] : -1;
        }
        wave_topk<GPW>(s, id, k, lane, w_s[w], w_i[w]);
    }
    __syncthreads();
    // (4) NW*k -> k, (5) certificate
    if (w == 0) {
        constexpr int R4 = (NW * KMAX + 63) / 64;
        float s[R4]; int64_t id[R4];
#pragma unroll
        for (int j = 0; j < R4; ++j) {
            const int i = j * 64 + lane;
            s[j] = i < NW * k ? w_s[i / k][i % k] : -INFINITY;
            id[j] = i < NW * k ? w_i[i / k][i % k] : -1;
        }
        wave_topk<R4>(s, id, k, lane, gs, gi_);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const bool full = gi_[k - 1] >= 0;                    // k rows found
        const float thr = full ? gs[k - 1] - tau_scale * sh_qn : -INFINITY;
        const bool collect = thr_out != nullptr;
        const bool flag = !collect && sh_u > -INFINITY && sh_u >= thr;     // something unscored might belong to the top-k
        if (lane == 0) { sh_flag = flag ? 1 : 0; sh_thr = thr; }
        if (!flag && lane < k) {
            out_s[(int64_t)q * k + lane] = gs[lane];
            out_i[(int64_t)q * k + lane] = gi_[lane] >= 0 ? gi_[lane] + idx_base : -1;
        }
        if (collect) {
            if (lane == 0) {
                thr_out[q] = thr;
                // this query's candidate state for the steps that follow (no separate memset launch)
                cand_counters[CNT_QCOUNT + q] = 0; cand_counters[CNT_QOVER + q] = 0; cand_nsurv[q] = 0;
                if (q == 0) { cand_counters[0] = 0; cand_counters[1] = 0; }
            }
            if (lane < K) selg_out[q * K + lane] = sel_g[lane];
        }
    }
    __syncthreads();
    if (!sh_flag) return;                                       // block-uniform

    // ---- certificate fallback: rescoring every unscored group whose pass-A maximum reaches the threshold -----------------
    float cs = -INFINITY; int64_t ci = -1;                      // this wave's running top-k: entry `lane` (lanes >= k empty)
    if (w == 0 && lane < k) { cs = gs[lane]; ci = gi_[lane]; }
    __syncthreads();                                            // gs is scratch from here: one 64-float row per wave
    float* sc = gs + w * GROUP_ROWS;
    const float thr = sh_thr;
    unsigned long long extra = 0;
    for (int64_t g0 = (int64_t)w * 64; g0 < n_groups; g0 += (int64_t)NW * 64) {
        const int64_t g = g0 + lane;
        bool sus = g < n_groups && gmax[(g < n_groups ? g : 0) * ldg + q] >= thr;
#pragma unroll 4
        for (int j = 0; j < K; ++j) sus = sus && (sel_g[j] != (int32_t)g);
        unsigned long long mask = __ballot(sus);
        while (mask) {
            const int b = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const int64_t gsel = g0 + b;
            ++extra;
            for (int r8 = 0; r8 < GROUP_ROWS; r8 += 8) {
                const int rr = r8 + rsub;
                const int64_t row = gsel * GROUP_ROWS + rr;
                const bool ok = row < n_rows;
                const float a = exact_row_score(C + row * D, qs, nch, l8, ok);
                if (l8 == 0) sc[rr] = ok ? a : -INFINITY;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int64_t row = gsel * GROUP_ROWS + lane;
            float s2[2] = {cs, sc[lane]};
            int64_t i2[2] = {ci, row < n_rows ? row : -1};
            wave_topk<2>(s2, i2, k, lane, w_s[w], w_i[w]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            cs = lane < k ? w_s[w][lane] : -INFINITY;
            ci = lane < k ? w_i[w][lane] : -1;
        }
    }
    if (lane < k) { w_s[w][lane] = cs; w_i[w][lane] = ci; }
    __syncthreads();
    if (w == 0) {
        constexpr int R4 = (NW * KMAX + 63) / 64;
        float s[R4]; int64_t id[R4];
#pragma unroll
        for (int j = 0; j < R4; ++j) {
            const int i = j * 64 + lane;
            s[j] = i < NW * k ? w_s[i / k][i % k] : -INFINITY;
            id[j] = i < NW * k ? w_i[i / k][i % k] : -1;
        }
        wave_topk<R4>(s, id, k, lane, gs, gi_);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane < k) {
            out_s[(int64_t)q * k + lane] = gs[lane];
            out_i[(int64_t)q * k + lane] = gi_[lane] >= 0 ? gi_[lane] + idx_base : -1;
        }
    }
    if (stats && lane == 0) {
        if (w == 0) atomicAdd(&stats[0], 1ull);
        if (extra) atomicAdd(&stats[1], extra);
    }
}

// ---- the tail of a SMALL query batch on the fp16 pass (<= 128 queries; pass A wrote the aux words): ONE kernel, one block per query ------
// What the select kernel + rescore_kernel pair does (and still does for wide batches and the int8 pipeline), restructured around two facts
// of the 625 k-row / 8-rank shape (profiles/r03: pass A 0.16 ms, select 0.011 + a launch boundary, rescore 0.068):
//   * the selection reads ~10 k group maxima per query: the block reads its query's gmax column itself (strided 4-byte loads, L2 hits),
//     16 groups per lane and step, instead of waiting for a second grid and its partial lists;
//   * of the K x 64 rows it used to rescore, K matter: the row each selected group's maximum came from (aux: its position) — unless the
//     group's SECOND best pass-A score (aux: an upper bound on it) reaches the provisional threshold, in which case the whole group is
//     rescored as before.  One CU then pulls 18 KB per query instead of 1.2 MB.
//   (1) every lane: maxima of its super-groups (16 consecutive groups) -> RS candidates per lane; wave top-(K+1); wave 0 merges: the K best
//       super-groups + the best one left out
//   (2) wave 0 expands them to K x 16 groups -> the K best groups + the best one left out           [as rescore_kernel]
//   (3a) exact scores of the K arg-max rows (8 lanes per row); provisional s_k, thr' = s_k - tau; groups with ub2 >= thr' (or fewer
//        than k rows so far) are EXPANDED: (3b) all their rows rescored
//   (4) top-k of the candidates; (5) certificate exactly as rescore_kernel's: U bounds every row in a group that was not selected; rows of a
//       selected, unexpanded group other than its arg-max row have pass-A score <= ub2 < thr' <= thr (s_k only grows as rows are added),
//       so they are covered too.  The fallback (every unscored group with gmax >= thr) is the same code.
#define I8D_CAP_ROWS 2048            // single-kernel int8 tail: (query, row) candidates a block lists before it gives up and scans exhaustively
#define I8D_CAP_GROUPS 512           // ... (query, group) candidates
#define I8D_EXTRA_SMEM (I8D_CAP_ROWS * 4 + I8D_CAP_GROUPS * 4 + SURV_CAP * 12)
template <int K, int NT, int RS, bool INBLOCK, bool COLLECT, int CW, bool I8D = false>
__global__ __launch_bounds__(NT) void tail_single_kernel(const float* __restrict__ gmax, const uint32_t* __restrict__ aux, int64_t ldg,
                                                          int64_t n_groups, const float* __restrict__ part_s, const int32_t* __restrict__ part_g,
                                                          int nslices, const f16_t* __restrict__ Q, const f16_t* __restrict__ C,
                                                          int64_t n_rows, int D, int k, float* __restrict__ out_s, int64_t* __restrict__ out_i,
                                                          int64_t idx_base, float tau_scale, int debug_drop,
                                                          unsigned long long* __restrict__ stats,
                                                          float* __restrict__ thr_out, int32_t* __restrict__ selg_out,
                                                          int* __restrict__ cand_counters, int* __restrict__ cand_nsurv,
                                                          const int32_t* __restrict__ only_if, unsigned long long* __restrict__ zero_stats) {
    // only_if: run only for the queries it flags (the int8 pipeline's last step: a query whose own candidate lists overflowed is answered
    // by this kernel's certificate fallback, i.e. exhaustively above the threshold).  zero_stats (COLLECT, first internal pass): the call's
    // certificate counters start at zero here — no other block of a COLLECT launch touches them, the kernels that add to them run later.
    if (only_if && !only_if[blockIdx.x]) return;
    if (zero_stats && blockIdx.x == 0 && threadIdx.x < 2) zero_stats[threadIdx.x] = 0ull;
    // COLLECT (int8 pre-filter; gmax / aux are UPPER BOUNDS there): write the provisional top-k, the threshold s_k - tau and the K selected
    // groups, start this query's candidate counters at zero, and leave everything at or above the threshold to collect_pairs /
    // pair_rescore / merge_survivors.  A selected group that was not expanded is skipped there like an expanded one: its rows other than
    // the arg-max row have TRUE scores <= ub2 < thr.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = NT / 64;
    constexpr int K1 = K + 1;
    constexpr int KK = K1 > KMAX ? K1 : KMAX;
    // CW = rows per candidate block of a selected group: 1 (int8 pass: aux names the arg-max ROW) or 4 (fp16 pass: the arg-max 4-row block)
    constexpr int NC = K * CW;
    static_assert(3 * K >= NW && NC <= 64, "fallback scratch / one lane of wave 0 per candidate row");
    __shared__ float w_s[NW][KK];
    __shared__ int64_t w_i[NW][KK];
    __shared__ int32_t sel_g[K];
    __shared__ int64_t sel_row[K];
    __shared__ float sel_ub2[K];
    __shared__ float cand_s[NC];
    __shared__ int64_t cand_i[NC];
    __shared__ int32_t exp_list[K];
    __shared__ float sh_u, sh_thr, sh_qn;
    __shared__ int sh_flag, sh_nexp;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    f16_t* qs = reinterpret_cast<f16_t*>(smem);               // [D] query row
    float* gs = reinterpret_cast<float*>(smem + (((size_t)D * 2 + 15) & ~(size_t)15));          // [K*64] row scores
    int64_t* gi_ = reinterpret_cast<int64_t*>(reinterpret_cast<char*>(gs) + K * GROUP_ROWS * 4);  // [K*64] row ids
    for (int i = tid; i < (D >> 3); i += NT)
        reinterpret_cast<u32x4*>(qs)[i] = reinterpret_cast<const u32x4*>(Q + (int64_t)q * D)[i];
    for (int i = tid; i < K * GROUP_ROWS; i += NT) { gs[i] = -INFINITY; gi_[i] = -1; }
    // (1) this query's super-group maxima: read here (INBLOCK: shards of up to ~2 M rows — one CU walking the column of a 10 M-row shard,
    // 156 k strided loads, takes longer than the select grid it replaces: 0.21 against 0.13 ms per 64-query batch) or taken from the select
    // kernel's partial lists (RS = candidates per lane of up to 256 slices x K entries)
    __shared__ float w_vb[NW];
    // I8D (int8 pass, small shard): the candidate pass below examines EVERY group at or above the threshold anyway, so the selection only has
    // to produce a good threshold, not the exact K best groups: each wave contributes the best group of its threads' groups (one wave
    // all-reduce instead of two 13-round top-k reductions and an expansion), the 12 best of those 16 are the selected groups.  The bounds
    // a thread read stay in its registers: the candidate pass does not read the column again.
    constexpr int GPT = I8D ? (TAIL_INBLOCK_MAX_SUPER * SUPER + NT - 1) / NT : 1;      // groups per thread (16 at 1 024 threads)
    float gv[GPT];
    if constexpr (I8D) {
        static_assert(!I8D || (INBLOCK && K <= NW), "one selected group per wave at most");
        float bs = -INFINITY; int64_t bg = INT64_MAX;
#pragma unroll
        for (int j = 0; j < GPT; ++j) {
            const int64_t g = (int64_t)j * NT + tid;
            gv[j] = g < n_groups ? gmax[g * ldg + q] : -INFINITY;
            if (g < n_groups && gv[j] > bs) { bs = gv[j]; bg = g; }          // ascending g: ties keep the lower group
        }
        wave_argbest(bs, bg);
        if (lane == 0) { w_s[w][0] = bs; w_i[w][0] = bg == INT64_MAX ? -1 : bg; w_vb[w] = -INFINITY; }
    } else if constexpr (!INBLOCK) {
        const int ncand = nslices * K;
        float s[RS]; int64_t id[RS];
        float vb = -INFINITY;
#pragma unroll
        for (int j = 0; j < RS; ++j) {
            const int i = (w * RS + j) * 64 + lane;
            s[j] = -INFINITY; id[j] = -1;
            if (i < ncand) {
                const int sl = i / K, pp = i - sl * K;
                const int64_t o = ((int64_t)sl * ldg + q) * K + pp;
                s[j] = part_s[o]; id[j] = part_g[o];
                if (pp == K - 1 && id[j] >= 0) vb = fmaxf(vb, s[j]);       // whatever this slice dropped scores <= its K-th kept value
            }
        }
        wave_topk<RS>(s, id, K1, lane, w_s[w], w_i[w]);
        vb = wave_max(vb);
        if (lane == 0) w_vb[w] = vb;
    } else {
        if (lane == 0) w_vb[w] = -INFINITY;
        const int64_t n_super = (n_groups + SUPER - 1) / SUPER;
        const float* col = gmax + q;
        float s[RS]; int64_t id[RS];
#pragma unroll
        for (int j = 0; j < RS; ++j) {
            const int64_t sg = (int64_t)j * NT + tid;
            s[j] = -INFINITY; id[j] = -1;
            if ((int64_t)j * NT < n_super) {                    // block-uniform: skips the loads of unused rounds
                float m = -INFINITY;
#pragma unroll 1
                for (int h = 0; h < SUPER; h += 8) {            // eight loads in flight at a time (sixteen cost the K = 36 instance three spilled registers)
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int64_t g = sg * SUPER + h + u;
                        v[u] = col[(g < n_groups ? g : n_groups - 1) * ldg];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) m = fmaxf(m, v[u]);
                }
                if (sg < n_super) { s[j] = m; id[j] = sg; }
            }
        }
        wave_topk<RS>(s, id, K1, lane, w_s[w], w_i[w]);
    }
    __syncthreads();
    if (w == 0) {
      float u = -INFINITY;
      if constexpr (I8D) {
        // the waves' best groups ranked by (bound desc, group asc) with shuffles; ranks 0 .. K-1 are the selected groups
        const float my = lane < NW ? w_s[lane][0] : -INFINITY;
        const int64_t mg = lane < NW ? w_i[lane][0] : -1;
        int better = 0;
#pragma unroll 4
        for (int j = 0; j < NW; ++j) {
            const float sj = __shfl(my, j);
            const int64_t gj = __shfl(mg, j);
            better += (gj >= 0 && mg >= 0 && cand_better(sj, gj, my, mg)) ? 1 : 0;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane < K1) { w_s[0][lane] = -INFINITY; w_i[0][lane] = -1; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (mg >= 0 && better < K1) { w_s[0][better] = my; w_i[0][better] = mg; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      } else {
        constexpr int R2 = (NW * K1 + 63) / 64;
        float s[R2]; int64_t id[R2];
#pragma unroll
        for (int j = 0; j < R2; ++j) {
            const int i = j * 64 + lane;
            s[j] = i < NW * K1 ? w_s[i / K1][i % K1] : -INFINITY;
            id[j] = i < NW * K1 ? w_i[i / K1][i % K1] : -1;
        }
        wave_topk<R2>(s, id, K1, lane, gs, gi_);               // K best super-groups -> gs/gi_[0..K), best left out -> [K]
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        u = gi_[K] >= 0 ? gs[K] : -INFINITY;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) u = fmaxf(u, w_vb[ww]);
        // (2) expand to K*SUPER groups, reduce to the K best groups (+ the best one left out)
        constexpr int R3 = (K * SUPER + 63) / 64;
        float s3[R3]; int64_t id3[R3];
#pragma unroll
        for (int j = 0; j < R3; ++j) {
            const int i = j * 64 + lane;
            s3[j] = -INFINITY; id3[j] = -1;
            if (i < K * SUPER) {
                const int64_t sg = gi_[i / SUPER];
                const int64_t g = sg * SUPER + (i % SUPER);
                if (sg >= 0 && g < n_groups) { s3[j] = gmax[g * ldg + q]; id3[j] = g; }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // gs / gi_[0..K] were scratch for the super-group list: back to "empty" before the candidates go in
        if (lane <= K) { gs[lane] = -INFINITY; gi_[lane] = -1; }
        wave_topk<R3>(s3, id3, K1, lane, w_s[0], w_i[0]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (w_i[0][K] >= 0) u = fmaxf(u, w_s[0][K]);
      }
        if (debug_drop && w_i[0][0] >= 0) u = fmaxf(u, w_s[0][0]);
        if (lane < K) {
            const int32_t g = !debug_drop ? (int32_t)w_i[0][lane] : (lane + 1 < K ? (int32_t)w_i[0][lane + 1] : -1);
            sel_g[lane] = g;
            int64_t row = -1; float ub2 = INFINITY;
            if (g >= 0) {
                const uint32_t a = aux[(int64_t)g * ldg + q];
                row = (int64_t)g * GROUP_ROWS + (int64_t)(a & 63u);
                ub2 = __uint_as_float(a & 0xFFFF0000u);
                // a position past the shard's end is a COPY of its last row (pass A clamps row addresses): then the copies tie with it and
                // ub2 is its own score, so the group is expanded below whenever that row matters; CW = 1: the candidate is the real row
                // (CW = 4: rows past the end are skipped one by one)
                if (CW == 1) row = row < n_rows ? row : n_rows - 1;
            }
            sel_row[lane] = row; sel_ub2[lane] = ub2;
        }
        float qq = 0.f;
        for (int i = lane; i < D; i += 64) { const float v = (float)qs[i]; qq = fmaf(v, v, qq); }
        qq = wave_sum(qq);
        if (lane == 0) { sh_u = u; sh_qn = sqrtf(qq); }
    }
    __syncthreads();
    // (3a) the K candidate blocks (CW rows each), exactly
    const int nch = D >> 3, l8 = lane & 7, rsub = lane >> 3;
    for (int t0 = w * 8; t0 < NC; t0 += NW * 8) {
        const int t = t0 + rsub, tg = (t < NC ? t : 0) / CW;
        const int64_t row0 = sel_row[tg] + (t % CW);
        const bool ok = t < NC && sel_g[tg] >= 0 && row0 < n_rows;
        const int64_t row = ok ? row0 : 0;
        const float a = exact_row_score(C + row * D, qs, nch, l8, ok);
        if (l8 == 0 && t < NC) { cand_s[t] = ok ? a : -INFINITY; cand_i[t] = ok ? row : -1; }
    }
    __syncthreads();
    if (w == 0) {
        const float my = lane < NC ? cand_s[lane] : -INFINITY;
        const int64_t mi = lane < NC ? cand_i[lane] : -1;
        int better = 0;
#pragma unroll 4
        for (int j = 0; j < NC; ++j) {
            const float sj = __shfl(my, j);
            const int64_t ij = __shfl(mi, j);
            better += (ij >= 0 && mi >= 0 && cand_better(sj, ij, my, mi)) ? 1 : 0;
        }
        const unsigned long long vmask = __ballot(mi >= 0);
        const int nvalid = __popcll(vmask);
        const unsigned long long kth = __ballot(mi >= 0 && better == k - 1);
        const float sk = (nvalid >= k && kth) ? __shfl(my, __ffsll((long long)kth) - 1) : -INFINITY;
        const float thr0 = sk > -INFINITY ? sk - tau_scale * sh_qn : -INFINITY;
        const bool expand = lane < K && sel_g[lane < K ? lane : 0] >= 0 && (sel_ub2[lane < K ? lane : 0] >= thr0);
        const unsigned long long em = __ballot(expand);
        if (expand) exp_list[__popcll(em & ((1ull << lane) - 1ull))] = lane;
        // an unexpanded group contributes its candidate block: slots (group index, 0 .. CW-1)
        const int cg = (lane < NC ? lane : 0) / CW;
        if (lane < NC && !((em >> cg) & 1ull) && mi >= 0) { gs[cg * GROUP_ROWS + lane % CW] = my; gi_[cg * GROUP_ROWS + lane % CW] = mi; }
        if (lane == 0) sh_nexp = __popcll(em);
        if (em == 0ull) {
            // the common case: the answer is the K candidates in rank order; nothing else to score
            if (mi >= 0 && better < k) {
                out_s[(int64_t)q * k + better] = my;
                out_i[(int64_t)q * k + better] = mi + idx_base;
            }
            if (lane >= nvalid && lane < k) { out_s[(int64_t)q * k + lane] = -INFINITY; out_i[(int64_t)q * k + lane] = -1; }
            const bool flag = !COLLECT && !I8D && sh_u > -INFINITY && sh_u >= thr0;
            if (lane == 0) { sh_flag = flag ? 1 : 0; sh_thr = thr0; }
            if constexpr (COLLECT) {
                if (lane == 0) {
                    thr_out[q] = thr0;
                    cand_counters[CNT_QCOUNT + q] = 0; cand_counters[CNT_QOVER + q] = 0; cand_nsurv[q] = 0;
                    if (q == 0) { cand_counters[0] = 0; cand_counters[1] = 0; }
                }
                if (lane < K) selg_out[q * K + lane] = sel_g[lane];
            }
            if (flag || I8D) {                                   // the fallback / the candidate pass start from the sorted list in gs / gi_[0..k)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane < NC) { gs[cg * GROUP_ROWS + lane % CW] = -INFINITY; gi_[cg * GROUP_ROWS + lane % CW] = -1; }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (mi >= 0 && better < k) { gs[better] = my; gi_[better] = mi; }
            }
        }
    }
    __syncthreads();
    const int nexp = sh_nexp;
    if (nexp > 0) {
        // (3b) every row of the expanded groups (block-uniform branch)
        for (int t0 = w * 8; t0 < nexp * GROUP_ROWS; t0 += NW * 8) {
            const int t = t0 + rsub, gidx = exp_list[t >> 6], rr = t & 63;
            const int64_t row = (int64_t)sel_g[gidx] * GROUP_ROWS + rr;
            const bool ok = row < n_rows;
            const float a = exact_row_score(C + row * D, qs, nch, l8, ok);
            if (l8 == 0) { gs[gidx * GROUP_ROWS + rr] = ok ? a : -INFINITY; gi_[gidx * GROUP_ROWS + rr] = ok ? row : -1; }
        }
        __syncthreads();
        constexpr int GPW = (K + NW - 1) / NW;
        {
            float s[GPW]; int64_t id[GPW];
#pragma unroll
            for (int gq = 0; gq < GPW; ++gq) {
                const int gidx = w + gq * NW;
                s[gq] = gidx < K ? gs[gidx * GROUP_ROWS + lane] : -INFINITY;
                id[gq] = gidx < K ? gi_[gidx * GROUP_ROWS + lane] : -1;
            }
            wave_topk<GPW>(s, id, k, lane, w_s[w], w_i[w]);
        }
        __syncthreads();
        if (w == 0) {
            constexpr int R4 = (NW * KMAX + 63) / 64;
            float s[R4]; int64_t id[R4];
#pragma unroll
            for (int j = 0; j < R4; ++j) {
                const int i = j * 64 + lane;
                s[j] = i < NW * k ? w_s[i / k][i % k] : -INFINITY;
                id[j] = i < NW * k ? w_i[i / k][i % k] : -1;
            }
            wave_topk<R4>(s, id, k, lane, gs, gi_);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const bool full = gi_[k - 1] >= 0;
            const float thr = full ? gs[k - 1] - tau_scale * sh_qn : -INFINITY;
            const bool flag = !COLLECT && !I8D && sh_u > -INFINITY && sh_u >= thr;
            if (lane == 0) { sh_flag = flag ? 1 : 0; sh_thr = thr; }
            if (!flag && lane < k) {
                out_s[(int64_t)q * k + lane] = gs[lane];
                out_i[(int64_t)q * k + lane] = gi_[lane] >= 0 ? gi_[lane] + idx_base : -1;
            }
            if constexpr (COLLECT) {
                if (lane == 0) {
                    thr_out[q] = thr;
                    cand_counters[CNT_QCOUNT + q] = 0; cand_counters[CNT_QOVER + q] = 0; cand_nsurv[q] = 0;
                    if (q == 0) { cand_counters[0] = 0; cand_counters[1] = 0; }
                }
                if (lane < K) selg_out[q * K + lane] = sel_g[lane];
            }
        }
        __syncthreads();
    }
    if constexpr (I8D) {
        // ---- int8 pass, small shard: the candidate step INSIDE this block (round 4: one launch instead of collect_pairs -> pair_rescore ->
        // merge_survivors -> redo, whose four dependent launches were 0.08 ms of a 0.25-ms batch on the 625 k-row slice).  The block reads
        // its query's column of UPPER BOUNDS once more (L2 hits); every group at or above the threshold that is not one of the K selected
        // ones becomes a (row) candidate when its second bound is below the threshold, a (group) candidate otherwise; the block's 16
        // waves rescoring them (8 lanes per row: 128 rows per step), rows at or above the threshold survive and are merged with the
        // provisional top-k.  More candidates than the lists hold (adversarial data) -> the exhaustive fallback below, which is exact.
        uint32_t* lrow = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(gi_) + K * GROUP_ROWS * 8);
        uint32_t* lgrp = lrow + I8D_CAP_ROWS;
        float* sv_s = reinterpret_cast<float*>(lgrp + I8D_CAP_GROUPS);
        int64_t* sv_i = reinterpret_cast<int64_t*>(sv_s + SURV_CAP);
        __shared__ int n_r, n_g, n_s, ovf;
        if (tid == 0) { n_r = 0; n_g = 0; n_s = 0; ovf = 0; }
        __syncthreads();
        const float thr = sh_thr;
#pragma unroll
        for (int j = 0; j < GPT; ++j) {
            const int64_t g = (int64_t)j * NT + tid;
            const float v = gv[j];                                // the bound this thread read in the selection step
            if (g >= n_groups || !(v >= thr)) continue;
            bool sel = false;
#pragma unroll 4
            for (int jj = 0; jj < K; ++jj) sel = sel || (sel_g[jj] == (int32_t)g);
            if (sel) continue;
            const uint32_t a = aux[g * ldg + q];
            const int64_t row = g * GROUP_ROWS + (int64_t)(a & 63u);
            if (__uint_as_float(a & 0xFFFF0000u) < thr && row < n_rows) {
                const int sl = atomicAdd(&n_r, 1);
                if (sl < I8D_CAP_ROWS) lrow[sl] = (uint32_t)row; else ovf = 1;
            } else {
                const int sl = atomicAdd(&n_g, 1);
                if (sl < I8D_CAP_GROUPS) lgrp[sl] = (uint32_t)g; else ovf = 1;
            }
        }
        __syncthreads();
        if (!ovf) {                                               // block-uniform
            const int nr = n_r, ng = n_g;
            for (int base = w * 8; base < nr; base += NW * 8) {
                const int p0 = base + rsub;
                const bool ok = p0 < nr;
                const int64_t row = (int64_t)lrow[ok ? p0 : 0];
                const float a = exact_row_score(C + row * D, qs, nch, l8, ok);
                if (l8 == 0 && ok && a >= thr) {
                    const int sl = atomicAdd(&n_s, 1);
                    if (sl < SURV_CAP) { sv_s[sl] = a; sv_i[sl] = row; } else ovf = 1;
                }
            }
            for (int p0 = w; p0 < ng; p0 += NW) {
                const int64_t gg = (int64_t)lgrp[p0];
                for (int r8 = 0; r8 < GROUP_ROWS; r8 += 8) {
                    const int64_t row = gg * GROUP_ROWS + r8 + rsub;
                    const bool ok = row < n_rows;
                    const float a = exact_row_score(C + (ok ? row : 0) * D, qs, nch, l8, ok);
                    if (l8 == 0 && ok && a >= thr) {
                        const int sl = atomicAdd(&n_s, 1);
                        if (sl < SURV_CAP) { sv_s[sl] = a; sv_i[sl] = row; } else ovf = 1;
                    }
                }
            }
            __syncthreads();
            if (!ovf) {                                           // block-uniform
                if (w == 0) {
                    const int ns = n_s;
                    constexpr int R = SURV_CAP / 64 + 1;
                    float s[R]; int64_t id[R];
#pragma unroll
                    for (int j = 0; j < R - 1; ++j) {
                        const int i = j * 64 + lane;
                        s[j] = i < ns ? sv_s[i] : -INFINITY;
                        id[j] = i < ns ? sv_i[i] : -1;
                    }
                    s[R - 1] = lane < k ? gs[lane] : -INFINITY;
                    id[R - 1] = lane < k ? gi_[lane] : -1;
                    wave_topk<R>(s, id, k, lane, w_s[0], w_i[0]);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (lane < k) {
                        out_s[(int64_t)q * k + lane] = w_s[0][lane];
                        out_i[(int64_t)q * k + lane] = w_i[0][lane] >= 0 ? w_i[0][lane] + idx_base : -1;
                    }
                    if (stats && lane == 0) atomicAdd(&stats[1], (unsigned long long)(nr + ng));
                }
                return;
            }
        }
        // too many candidates or survivors for the lists: answered exhaustively below (counted as a slow-path query)
    }
    if (COLLECT || (!I8D && !sh_flag)) return;                  // block-uniform

    // ---- certificate fallback (as rescore_kernel's): every group whose pass-A maximum reaches the threshold and that is not among the K
    // selected ones is rescored in full.  A selected group that was NOT expanded is skipped with them: its rows other than the arg-max
    // row score below ub2 < thr' <= thr.  One that is needed after all (thr dropped? it cannot: s_k only grows) never arises.
    float cs = -INFINITY; int64_t ci = -1;
    if (w == 0 && lane < k) { cs = gs[lane]; ci = gi_[lane]; }
    __syncthreads();
    float* sc = gs + w * GROUP_ROWS;
    const float thr = sh_thr;
    unsigned long long extra = 0;
    for (int64_t g0 = (int64_t)w * 64; g0 < n_groups; g0 += (int64_t)NW * 64) {
        const int64_t g = g0 + lane;
        bool sus = g < n_groups && gmax[(g < n_groups ? g : 0) * ldg + q] >= thr;
#pragma unroll 4
        for (int j = 0; j < K; ++j) sus = sus && (sel_g[j] != (int32_t)g);
        unsigned long long mask = __ballot(sus);
        while (mask) {
            const int b = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const int64_t gsel = g0 + b;
            ++extra;
            for (int r8 = 0; r8 < GROUP_ROWS; r8 += 8) {
                const int rr = r8 + rsub;
                const int64_t row = gsel * GROUP_ROWS + rr;
                const bool ok = row < n_rows;
                const float a = exact_row_score(C + row * D, qs, nch, l8, ok);
                if (l8 == 0) sc[rr] = ok ? a : -INFINITY;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int64_t row = gsel * GROUP_ROWS + lane;
            float s2[2] = {cs, sc[lane]};
            int64_t i2[2] = {ci, row < n_rows ? row : -1};
            wave_topk<2>(s2, i2, k, lane, w_s[w], w_i[w]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            cs = lane < k ? w_s[w][lane] : -INFINITY;
            ci = lane < k ? w_i[w][lane] : -1;
        }
    }
    if (lane < k) { w_s[w][lane] = cs; w_i[w][lane] = ci; }
    __syncthreads();
    if (w == 0) {
        constexpr int R4 = (NW * KMAX + 63) / 64;
        float s[R4]; int64_t id[R4];
#pragma unroll
        for (int j = 0; j < R4; ++j) {
            const int i = j * 64 + lane;
            s[j] = i < NW * k ? w_s[i / k][i % k] : -INFINITY;
            id[j] = i < NW * k ? w_i[i / k][i % k] : -1;
        }
        wave_topk<R4>(s, id, k, lane, gs, gi_);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane < k) {
            out_s[(int64_t)q * k + lane] = gs[lane];
            out_i[(int64_t)q * k + lane] = gi_[lane] >= 0 ? gi_[lane] + idx_base : -1;
        }
    }
    if (stats && lane == 0) {
        if (w == 0) atomicAdd(&stats[0], 1ull);
        if (extra) atomicAdd(&stats[1], extra);
    }
}

// ---- int8 pre-filter: the candidates beyond the K selected groups, in three coalesced / parallel steps ---------------------------------
#define PAIR_CAP_PER_QUERY 4096      // (query, group) pairs kept per query; a query that needs more is re-run ALONE by the exhaustive kernel
// counters (ints): [0] (query, group) pairs appended, [1] (query, row) pairs appended, [2, 2 + QBATCH_MAX) pairs seen per query, [2 + QBATCH_MAX, 2 + 2 QBATCH_MAX)
// per-query overflow flag.  Overflow is PER QUERY: every query appends at most PAIR_CAP_PER_QUERY pairs, so the shared list
// (nq x PAIR_CAP_PER_QUERY slots) cannot overflow, and one clustered query whose bound lets thousands of groups through sends only itself
// to the exhaustive kernel, not the whole batch.

// every group whose upper bound reaches a query's threshold and that was not rescored yet -> a candidate: a (query, ROW) pair when the
// group's second-largest row bound is below the threshold (only the arg-max row can matter), a (query, group) pair otherwise.
// Thread t of a block owns queries 4t .. 4t+3 (one 16-B load per group row: a wave reads 1 KB of the row, fully coalesced); the block
// walks `gpb` consecutive groups.
#define COLLECT_LDS_CAP 1024         // candidates a block buffers per list before it reserves its range of the global list
__global__ __launch_bounds__(256) void collect_pairs_kernel(const float* __restrict__ gmax, const uint32_t* __restrict__ aux, int64_t ldg,
                                                             int64_t n_groups, int64_t n_rows, int nq,
                                                             const float* __restrict__ thr, const int32_t* __restrict__ selg, int K,
                                                             unsigned long long* __restrict__ pairs, unsigned long long* __restrict__ rpairs,
                                                             int* __restrict__ counters, int gpb) {
    // One global atomic per block and list, not one per candidate: ~150 candidates per query x 1 024 queries on ONE counter word is
    // 1.8 ms of serialised atomics (a word takes ~88 per microsecond; first version of this kernel, profiles/r03).  Candidates go to
    // two LDS buffers through LDS atomics; the block then reserves its ranges and copies them out.  A buffer that fills up (clustered
    // data: thousands of candidates in 64 groups) spills straight to the global list.
    __shared__ unsigned long long lp[COLLECT_LDS_CAP], lr[COLLECT_LDS_CAP];
    __shared__ int ln[2], lbase[2];
    if (threadIdx.x < 2) ln[threadIdx.x] = 0;
    __syncthreads();
    // ldg / 4 threads cover one group's row of queries (4 each); the block's 256 threads take 1 024 / ldg groups per step (round 3 gave a
    // block ONE group per step: with 256 queries per pass three quarters of its threads idled, and a 625 k-row shard had 153 such blocks —
    // 153 waves doing the whole candidate step of a 256-query batch, 0.25 ms of a 0.63-ms batch)
    const int qpr = (int)(ldg >> 2), gpi = 256 / qpr;
    const int q0 = (threadIdx.x % qpr) * 4, tg = threadIdx.x / qpr;
    {
        float t4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) t4[j] = (q0 + j < nq) ? thr[q0 + j] : INFINITY;
        const int64_t g0 = (int64_t)blockIdx.x * gpb;
        const int64_t g1 = g0 + gpb < n_groups ? g0 + gpb : n_groups;
        for (int64_t g = g0 + tg; g < g1; g += gpi) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(gmax + g * ldg + q0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // columns [nq, ldg) of gmax are never written by pass A (stale workspace bytes): skipped by index, not by the threshold
                if (q0 + j >= nq) continue;
                if (!(v[j] >= t4[j])) continue;
                const int q = q0 + j;
                bool sel = false;
                for (int jj = 0; jj < K; ++jj) sel = sel || (selg[q * K + jj] == (int32_t)g);
                if (sel) continue;
                const int qc = atomicAdd(&counters[CNT_QCOUNT + q], 1);
                if (qc >= PAIR_CAP_PER_QUERY) { counters[CNT_QOVER + q] = 1; continue; }
                const uint32_t a = aux[g * ldg + q];
                const int64_t row = g * GROUP_ROWS + (int64_t)(a & 63u);
                const bool single = __uint_as_float(a & 0xFFFF0000u) < t4[j] && row < n_rows;
                const unsigned long long e = ((unsigned long long)q << 32) | (unsigned long long)(uint32_t)(single ? row : g);
                const int slot = atomicAdd(&ln[single ? 1 : 0], 1);
                if (slot < COLLECT_LDS_CAP) (single ? lr : lp)[slot] = e;
                else (single ? rpairs : pairs)[atomicAdd(&counters[single ? 1 : 0], 1)] = e;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        const int n = ln[threadIdx.x] < COLLECT_LDS_CAP ? ln[threadIdx.x] : COLLECT_LDS_CAP;
        lbase[threadIdx.x] = n ? atomicAdd(&counters[threadIdx.x], n) : 0;
        ln[threadIdx.x] = n;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ln[0]; i += 256) pairs[lbase[0] + i] = lp[i];
    for (int i = threadIdx.x; i < ln[1]; i += 256) rpairs[lbase[1] + i] = lr[i];
}

// eight lanes per (query, row) candidate: the row's exact score (the same exact_row_score as everywhere); at or above the query's
// threshold it joins the survivor list
__device__ __forceinline__ void row_candidates(const unsigned long long* __restrict__ rpairs, const int* __restrict__ counters,
                                               const f16_t* __restrict__ Q, const f16_t* __restrict__ C, int D,
                                               const float* __restrict__ thr, float* __restrict__ surv_s,
                                               int64_t* __restrict__ surv_i, int* __restrict__ nsurv) {
    const int lane = threadIdx.x & 63, l8 = lane & 7, nch = D >> 3;
    const int np = counters[1];
    // a wave takes 8 candidates per step (8 lanes each); `base` is wave-uniform, so every lane runs every step (the butterfly inside
    // exact_row_score needs the whole wave)
    for (int base = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 8; base < np; base += gridDim.x * 32) {
        const int p0 = base + (lane >> 3);
        const bool ok = p0 < np;
        const unsigned long long pr = rpairs[ok ? p0 : 0];
        const int q = (int)(pr >> 32);
        const int64_t row = (int64_t)(uint32_t)pr;
        const bool go = ok && !counters[CNT_QOVER + q];
        const float a = exact_row_score(C + row * D, Q + (int64_t)q * D, nch, l8, go);
        if (l8 == 0 && go && a >= thr[q]) {
            const int sidx = atomicAdd(&nsurv[q], 1);
            if (sidx < SURV_CAP) { surv_s[q * SURV_CAP + sidx] = a; surv_i[q * SURV_CAP + sidx] = row; }
        }
    }
}

// Candidate rescoring, one launch: first the (query, row) candidates (see row_candidates), then the (query, group) ones — one wave per
// pair: exact scores of the group's 64 rows (the same exact_row_score as everywhere); rows at or above the query's threshold are appended
// to its survivor list.
__global__ __launch_bounds__(256) void pair_rescore_kernel(const unsigned long long* __restrict__ pairs, const unsigned long long* __restrict__ rpairs,
                                                            const int* __restrict__ counters,
                                                            const f16_t* __restrict__ Q, const f16_t* __restrict__ C, int64_t n_rows, int D,
                                                            const float* __restrict__ thr, float* __restrict__ surv_s,
                                                            int64_t* __restrict__ surv_i, int* __restrict__ nsurv) {
    row_candidates(rpairs, counters, Q, C, D, thr, surv_s, surv_i, nsurv);
    const int lane = threadIdx.x & 63, l8 = lane & 7, rsub = lane >> 3, nch = D >> 3;
    const int np = counters[0];
    for (int p = blockIdx.x * 4 + (threadIdx.x >> 6); p < np; p += gridDim.x * 4) {
        const unsigned long long pr = pairs[p];
        const int q = (int)(pr >> 32);
        if (counters[CNT_QOVER + q]) continue;                    // this query goes to the exhaustive kernel anyway (wave-uniform)
        const int64_t g = (int64_t)(uint32_t)pr;
        const float t = thr[q];
        const f16_t* qrow = Q + (int64_t)q * D;
        for (int r8 = 0; r8 < GROUP_ROWS; r8 += 8) {
            const int64_t row = g * GROUP_ROWS + r8 + rsub;
            const bool ok = row < n_rows;
            const float a = exact_row_score(C + (ok ? row : 0) * D, qrow, nch, l8, ok);
            if (l8 == 0 && ok && a >= t) {
                const int sidx = atomicAdd(&nsurv[q], 1);
                if (sidx < SURV_CAP) { surv_s[q * SURV_CAP + sidx] = a; surv_i[q * SURV_CAP + sidx] = row; }
            }
        }
    }
}

// provisional top-k (from the K selected groups) + survivors -> final top-k; a query whose OWN lists overflowed is flagged for the exhaustive re-run
__global__ __launch_bounds__(256) void merge_survivors_kernel(float* __restrict__ out_s, int64_t* __restrict__ out_i, int nq, int k, int64_t idx_base,
                                                               const float* __restrict__ surv_s, const int64_t* __restrict__ surv_i,
                                                               const int* __restrict__ nsurv, const int* __restrict__ counters,
                                                               int32_t* __restrict__ redo, unsigned long long* __restrict__ stats) {
    __shared__ float ms[4][KMAX];
    __shared__ int64_t mi[4][KMAX];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + w;
    if (q >= nq) return;
    const int n = nsurv[q];
    const bool over = n > SURV_CAP || counters[CNT_QOVER + q] != 0;
    if (lane == 0) {
        redo[q] = over ? 1 : 0;
        if (stats) { if (over) atomicAdd(&stats[0], 1ull); if (q == 0) atomicAdd(&stats[1], (unsigned long long)(counters[0] + counters[1])); }
    }
    if (over) return;
    constexpr int R = SURV_CAP / 64 + 1;
    float s[R]; int64_t id[R];
#pragma unroll
    for (int j = 0; j < R - 1; ++j) {
        const int i = j * 64 + lane;
        s[j] = i < n ? surv_s[q * SURV_CAP + i] : -INFINITY;
        id[j] = i < n ? surv_i[q * SURV_CAP + i] : -1;
    }
    s[R - 1] = lane < k ? out_s[(int64_t)q * k + lane] : -INFINITY;
    id[R - 1] = (lane < k && out_i[(int64_t)q * k + lane] >= 0) ? out_i[(int64_t)q * k + lane] - idx_base : -1;
    wave_topk<R>(s, id, k, lane, ms[w], mi[w]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane < k) {
        out_s[(int64_t)q * k + lane] = ms[w][lane];
        out_i[(int64_t)q * k + lane] = mi[w][lane] >= 0 ? mi[w][lane] + idx_base : -1;
    }
}

// merge P partial lists: one wave per query (n_parts*k candidates, k <= 32)
__global__ __launch_bounds__(256) void merge_kernel(const float* __restrict__ ps, const int64_t* __restrict__ pi, int P, int nq,
                                                     int k, float* __restrict__ out_s, int64_t* __restrict__ out_i) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    const int n = P * k;
    // each lane owns candidates lane, lane+64, ...; "taken" candidates are marked in a private bitmask
    uint64_t taken = 0;
    for (int r = 0; r < k; ++r) {
        float bs = -INFINITY; int64_t bi = INT64_MAX; int bslot = -1;
        for (int c = lane, sl = 0; c < n; c += 64, ++sl) {
            if (taken >> sl & 1) continue;
            const int p = c / k, e = c % k;
            const int64_t o = ((int64_t)p * nq + q) * k + e;
            const float v = ps[o]; const int64_t vi = pi[o];
            if (vi < 0) continue;
            if (bslot < 0 || v > bs || (v == bs && vi < bi)) { bs = v; bi = vi; bslot = sl; }
        }
        float ws = bslot >= 0 ? bs : -INFINITY; int64_t wi = bslot >= 0 ? bi : INT64_MAX;
        wave_argbest(ws, wi);
        const bool found = wi != INT64_MAX;
        if (found && bslot >= 0 && bi == wi && bs == ws) taken |= (1ull << bslot);      // (global ids are distinct across shards)
        if (lane == 0) {
            out_s[(int64_t)q * k + r] = found ? ws : -INFINITY;
            out_i[(int64_t)q * k + r] = found ? wi : -1;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// L2-normalised N(0,1) rows in fp16 from (seed, row, col) — bench cfg 3 corpus/queries generated in HBM.
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}
__global__ __launch_bounds__(256) void fill_unit_rows_kernel(f16_t* __restrict__ dst, int64_t n_rows, int D, uint64_t seed, int64_t row_base) {
    const int lane = threadIdx.x & 63;
    const int64_t lrow = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (lrow >= n_rows) return;
    const int64_t row = lrow + row_base;                  // the row's index in the whole corpus: what the values depend on
    float v[16];
    float sq = 0.f;
    const int per = D / 64;          // D % 128 == 0 -> per even, <= 16
    for (int e = 0; e < per; e += 2) {
        const uint64_t r = splitmix64(seed ^ splitmix64((uint64_t)row * 1024 + (uint64_t)(lane * per + e)));
        const float u1 = ((float)((r >> 40) + 1)) * (1.0f / 16777216.0f);          // (0,1]
        const float u2 = ((float)((r >> 8) & 0xffffff)) * (1.0f / 16777216.0f);
        const float rad = sqrtf(-2.0f * __logf(u1));
        float sn, cs;
        __sincosf(6.283185307179586f * u2, &sn, &cs);
        v[e] = rad * cs; v[e + 1] = rad * sn;
        sq += v[e] * v[e] + v[e + 1] * v[e + 1];
    }
    const float inv = rsqrtf(wave_sum(sq));
    f16_t* out = dst + lrow * D + lane * per;
    for (int e = 0; e < per; ++e) out[e] = (f16_t)(v[e] * inv);
}

// Embedding-like rows (arx_fill_clustered_rows_f16_at): centre(cluster(row)) + spread * noise(row), per-dimension gain, unit-normalised.
__device__ __forceinline__ void gauss2(uint64_t r, float& g0, float& g1) {
    const float u1 = ((float)((r >> 40) + 1)) * (1.0f / 16777216.0f);          // (0,1]
    const float u2 = ((float)((r >> 8) & 0xffffff)) * (1.0f / 16777216.0f);
    const float rad = sqrtf(-2.0f * __logf(u1));
    float sn, cs;
    __sincosf(6.283185307179586f * u2, &sn, &cs);
    g0 = rad * cs; g1 = rad * sn;
}
__global__ __launch_bounds__(256) void fill_clustered_rows_kernel(f16_t* __restrict__ dst, int64_t n_rows, int D, uint64_t seed, int64_t row_base,
                                                                   int n_clusters, float spread, int n_hot, float hot_gain) {
    const int lane = threadIdx.x & 63;
    const int64_t lrow = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (lrow >= n_rows) return;
    const int64_t row = lrow + row_base;
    // n_clusters > 0: cluster = hash(row) (members scattered over the shard); n_clusters < 0: TOPIC ORDER — consecutive runs of -n_clusters rows
    // share a centre (consecutive chunks of one paper: neighbours in row order AND in embedding space, so whole 64-row groups are near-tied)
    const uint64_t cl = n_clusters > 0 ? splitmix64(seed ^ splitmix64(0xC105ull + (uint64_t)row)) % (uint64_t)n_clusters
                                       : (uint64_t)row / (uint64_t)(-n_clusters);
    float v[16];
    float sq = 0.f;
    const int per = D / 64;
    for (int e = 0; e < per; e += 2) {
        const int col = lane * per + e;
        float c0, c1, n0, n1;
        gauss2(splitmix64(seed ^ splitmix64(0xCE17ull + cl * 1024 + (uint64_t)col)), c0, c1);
        gauss2(splitmix64((seed + 0x5EEDull) ^ splitmix64((uint64_t)row * 1024 + (uint64_t)col)), n0, n1);
        float g0 = 1.f, g1 = 1.f;
        for (int h = 0; h < n_hot; ++h) {
            const int hd = (int)(splitmix64(seed + 0x407ull * (uint64_t)(h + 1)) % (uint64_t)D);
            g0 = hd == col ? hot_gain : g0; g1 = hd == col + 1 ? hot_gain : g1;
        }
        v[e] = (c0 + spread * n0) * g0; v[e + 1] = (c1 + spread * n1) * g1;
        sq += v[e] * v[e] + v[e + 1] * v[e + 1];
    }
    const float inv = rsqrtf(wave_sum(sq));
    f16_t* out = dst + lrow * D + lane * per;
    for (int e = 0; e < per; ++e) out[e] = (f16_t)(v[e] * inv);
}

// ---------------------------------------------------------------------------------------------------
struct TopkWs { int64_t stats, gmax, aux, part_s, part_g, q8, qmeta, thr, selg, counters, nsurv, redo, pairs, rpairs, surv_s, surv_i, total;
                int64_t ldg; int nsplit; int64_t n_groups; };
// has_i8: the layout arx_topk_search_i8 needs (aux word per (query, group), the quantised query batch, the candidate pipeline's lists:
// about as much again as gmax + 64 KB per query); the fp16 pass reserves none of it.  Both layouts share their prefix
// (stats, gmax), so arx_topk_stats reads either.
static TopkWs topk_layout(int64_t n_rows, int nq, int k, int dim, bool has_i8) {
    TopkWs w;
    const int qb = nq < QBATCH_MAX ? nq : QBATCH_MAX;
    w.ldg = round_up64(qb, 64);
    w.n_groups = (n_rows + GROUP_ROWS - 1) / GROUP_ROWS;
    const int64_t n_super = (w.n_groups + SUPER - 1) / SUPER;
    int64_t ns = (n_super + 4 * 2 - 1) / (4 * 2);               // ~2 super-groups per wave-slice (each is one dependent round of 16 loads: a 625 k-row
                                                                 // shard spent 21 us in eight such rounds per wave; capped at 256 slices = rescore's merge width)
    w.nsplit = (int)(ns < 1 ? 1 : (ns > 256 ? 256 : ns));
    int64_t o = 0;
    auto take = [&](int64_t b) { int64_t r = o; o += round_up64(b, 256); return r; };
    auto take8 = [&](int64_t b) { return take(has_i8 ? b : 0); };
    w.stats = take(16);                                          // certificate counters, at the allocation's start (zeroed per call)
    w.gmax = take(w.n_groups * w.ldg * 4);
    w.part_s = take((int64_t)w.nsplit * w.ldg * KSEL_BIG * 4);
    w.part_g = take((int64_t)w.nsplit * w.ldg * KSEL_BIG * 4);
    // second bound + arg-max row per (query, group): written by the int8 pass and by the fp16 pass of small batches (<= 128 queries: the
    // single-row tail); a wide fp16 batch (ldg up to 1 024) reserves none
    w.aux = take((has_i8 || (qb <= AUX16_MAX_NQ && n_super <= TAIL_INBLOCK_MAX_SUPER)) ? w.n_groups * w.ldg * 4 : 0);
    w.q8 = take8(w.ldg * (int64_t)dim);                          // the query batch quantised
    w.qmeta = take8(w.ldg * 8);
    const int64_t qc = qb;                                       // candidate pipeline state, sized by the internal query batch
    w.thr = take8(qc * 4);
    w.selg = take8(qc * KSEL_BIG * 4);
    w.counters = take8(CNT_INTS * 4);                            // see CNT_* (zeroed per batch together with nsurv, which follows)
    w.nsurv = take8(QBATCH_MAX * 4);
    w.redo = take8(qc * 4);
    w.pairs = take8(qc * PAIR_CAP_PER_QUERY * 8);
    w.rpairs = take8(qc * PAIR_CAP_PER_QUERY * 8);
    w.surv_s = take8(qc * SURV_CAP * 4);
    w.surv_i = take8(qc * SURV_CAP * 8);
    w.total = o;
    return w;
}

extern "C" int64_t arx_topk_workspace_bytes(int64_t n_rows, int32_t n_queries, int32_t dim, int32_t k) {
    if (n_rows <= 0 || n_queries <= 0 || dim <= 0 || k <= 0 || k > KMAX) return -1;
    return topk_layout(n_rows, n_queries, k, dim, false).total;
}
extern "C" int64_t arx_topk_workspace_bytes_i8(int64_t n_rows, int32_t n_queries, int32_t dim, int32_t k) {
    if (n_rows <= 0 || n_queries <= 0 || dim <= 0 || k <= 0 || k > KMAX || dim % 128 != 0 || dim > 1024) return -1;
    return topk_layout(n_rows, n_queries, k, dim, true).total;
}

template <int BM, bool GLDS>
static int launch_groupmax(const f16_t* Q, int nq, const f16_t* C, int64_t n_rows, int D, float* gmax, int64_t ldg, uint32_t* aux,
                           unsigned long long* zero_stats, hipStream_t st) {
    using ML = GemmMainloop<f16_t, BM, 256, 2, 4, GLDS, GLDS ? 3 : 0>;
    auto kern = search_groupmax_kernel<BM, GLDS>;
    constexpr int smem_bytes = (BM == 256 && GLDS) ? Gemm8Phase<f16_t, 2>::STAGE_OFF : ML::SMEM_BYTES;
    ARX_HIP_CHECK(arx_func_smem((const void*)kern, smem_bytes));
    const int tq = cdiv(nq, BM);
    const int64_t tn = (n_rows + 255) / 256;
    ARX_REQUIRE(tq * tn < (1ll << 31), "grid too large");
    kern<<<(int)(tq * tn), 512, smem_bytes, st>>>(Q, nq, C, n_rows, D, tq, (int)tn, gmax, ldg, aux, zero_stats);
    ARX_HIP_CHECK(hipGetLastError());
    return ARX_OK;
}

template <int BM, bool GLDS>
static int launch_groupmax_i8(const int8_t* Q8, const float2* qmeta, int nq, const int8_t* C8, const float2* cmeta, int64_t n_rows, int D,
                              float* gmax, uint32_t* aux, int64_t ldg, hipStream_t st) {
    using ML = GemmMainloop<i8pair_t, BM, 256, 2, 4, GLDS, GLDS ? 3 : 0>;
    auto kern = search_groupmax_i8_kernel<BM, GLDS>;
    constexpr int smem_bytes = ((BM == 256 && GLDS) ? Gemm8Phase<i8pair_t, 2>::STAGE_OFF : ML::SMEM_BYTES) + (BM >= 128 ? 4096 : 0);
    ARX_HIP_CHECK(arx_func_smem((const void*)kern, smem_bytes));
    const int tq = cdiv(nq, BM);
    const int64_t tn = (n_rows + 255) / 256;
    ARX_REQUIRE(tq * tn < (1ll << 31), "grid too large");
    kern<<<(int)(tq * tn), 512, smem_bytes, st>>>(Q8, qmeta, nq, C8, cmeta, n_rows, D, tq, (int)tn, gmax, aux, ldg);
    ARX_HIP_CHECK(hipGetLastError());
    return ARX_OK;
}

// pass A in persistent form: >= 256 queries per pass, an even number of 64-element k-tiles, rows addressable as int
template <typename T, bool I8, bool AUX16 = false>
static int launch_groupmax_persistent(const T* Q, int nq, const T* C, int64_t n_rows, int Kt, int D, const float2* qmeta, const float2* cmeta,
                                      float* gmax, uint32_t* aux, int64_t ldg, int cu_limit, hipStream_t st,
                                      unsigned long long* zero_stats = nullptr) {
    auto kern = search_groupmax_persistent_kernel<T, I8, AUX16>;
    constexpr int smem_bytes = Gemm8Phase<T, 0>::SMEM_BYTES;
    ARX_HIP_CHECK(arx_func_smem((const void*)kern, smem_bytes));
    const int tq = cdiv(nq, 256);
    const int64_t tn = (n_rows + 255) / 256;
    int n_cu = arx_device_cus();
    if (cu_limit > 0 && cu_limit < n_cu) n_cu = cu_limit;          // a CU-masked stream: one block per CU it may use
    int64_t grid = tq * tn < n_cu ? tq * tn : n_cu;
    grid = grid / 8 * 8 > 0 ? grid / 8 * 8 : 8;                   // a multiple of 8: a block keeps its XCD (and its residue class of corpus tiles)
    kern<<<(int)grid, 512, smem_bytes, st>>>(Q, nq, C, n_rows, Kt, D, tq, (int)tn, qmeta, cmeta, gmax, aux, ldg, zero_stats);
    ARX_HIP_CHECK(hipGetLastError());
    return ARX_OK;
}
static bool persistent_pass_ok(int nq, int64_t n_rows, int k_elems, int flags) {
    return !(flags & ARX_TOPK_NO_PERSISTENT) && nq > 128 && n_rows < (1ll << 31) - 256 && k_elems % 128 == 0 &&
           (int64_t)k_elems * 2 * 256 < (1ll << 31);
}

static int64_t i8_meta_offset(int64_t n_rows, int dim) { return round_up64(n_rows * (int64_t)dim, 256); }

extern "C" int64_t arx_topk_i8_index_bytes(int64_t n_rows, int32_t dim) {
    if (n_rows <= 0 || dim <= 0 || dim % 128 != 0 || dim > 1024) return -1;
    return i8_meta_offset(n_rows, dim) + n_rows * 8;
}

extern "C" int32_t arx_topk_build_i8(const void* corpus, int64_t n_rows, int32_t dim, void* index_i8, void* stream) {
    ARX_REQUIRE(corpus && index_i8 && n_rows > 0, "bad args");
    ARX_REQUIRE(dim % 128 == 0 && dim <= 1024, "int8 pre-filter: dim=%d must be a multiple of 128, <= 1024", dim);
    const int64_t blocks = (n_rows + 3) / 4;
    ARX_REQUIRE(blocks < (1ll << 31), "too many rows for one launch");
    quantize_rows_i8_kernel<<<(int)blocks, 256, 0, (hipStream_t)stream>>>((const f16_t*)corpus, n_rows, dim, (int8_t*)index_i8,
                                                                         (float2*)((char*)index_i8 + i8_meta_offset(n_rows, dim)), nullptr);
    ARX_HIP_CHECK(hipGetLastError());
    return ARX_OK;
}

static int launch_collect_rescore(const TopkWs& L, char* ws, const float* gmax, const uint32_t* aux, int64_t n_rows, int nq, const float* thr,
                                  const int32_t* selg, int K, const f16_t* Q, const f16_t* C, int D, float* surv_s, int64_t* surv_i, int* nsurv,
                                  int* counters, hipStream_t st) {
    // Candidate step, two launches: collect_pairs (lists) -> pair_rescore (a balanced grid over ALL candidates).  A fused form — the block
    // that finds a candidate rescoring it, no lists — was built and measured in round 4 (profiles/r04/i8_candidate_step_fused_ab.md): it
    // wins 10 % at <= 128 queries and loses 25-50 % at 256 and 1 024 (a block is then ~25 us of dependent round trips at four blocks per
    // CU); the pair stays.
    unsigned long long* pairs = (unsigned long long*)(ws + L.pairs);
    unsigned long long* rpairs = (unsigned long long*)(ws + L.rpairs);
    // groups per block: a multiple of the 1 024 / ldg groups a block takes per step; ~2 048 blocks, at most 64 groups each
    const int gpi = (int)(1024 / L.ldg);
    int64_t gpb = (L.n_groups + 2047) / 2048;
    gpb = (gpb + gpi - 1) / gpi * gpi;
    if (gpb < gpi) gpb = gpi;
    if (gpb > 64) gpb = 64 > gpi ? 64 : gpi;
    const int64_t blocks = (L.n_groups + gpb - 1) / gpb;
    ARX_REQUIRE(blocks < (1ll << 31), "grid too large");
    collect_pairs_kernel<<<(int)blocks, 256, 0, st>>>(gmax, aux, L.ldg, L.n_groups, n_rows, nq, thr, selg, K, pairs, rpairs, counters, (int)gpb);
    ARX_HIP_CHECK(hipGetLastError());
    pair_rescore_kernel<<<2048, 256, 0, st>>>(pairs, rpairs, counters, Q, C, n_rows, D, thr, surv_s, surv_i, nsurv);
    ARX_HIP_CHECK(hipGetLastError());
    return ARX_OK;
}

// The int8 pipeline with ONE row per selected group in its first step (default; aux is written by the int8 pass anyway):
//   shards of up to TAIL_INBLOCK_MAX_SUPER super-groups (1 M rows), k <= 10: ONE launch, tail_single_kernel<I8D> (round 3: six, and 0.47 ms
//   of a 0.63-ms 256-query batch on 625 k rows);
//   larger shards: select_groups -> tail_single_kernel<COLLECT> (provisional top-k, threshold, selected groups) -> collect_pairs ->
//   pair_rescore (every other (query, group) at or above the threshold: one row of it where the aux word allows) -> merge_survivors ->
//   tail_single_kernel<only_if> (a query whose own lists overflowed: exhaustive above the threshold).
template <int K>
static int run_i8_single(const TopkWs& L, char* ws, const f16_t* Q, int nq, const f16_t* C, int64_t n_rows, int D, int k, float* out_s,
                         int64_t* out_i, int64_t idx_base, float tau_scale, int debug_drop, hipStream_t st, bool first_pass) {
    float* gmax = (float*)(ws + L.gmax);
    const uint32_t* aux = (const uint32_t*)(ws + L.aux);
    float* ps = (float*)(ws + L.part_s);
    int32_t* pg = (int32_t*)(ws + L.part_g);
    float* thr = (float*)(ws + L.thr);
    int32_t* selg = (int32_t*)(ws + L.selg);
    int* counters = (int*)(ws + L.counters);
    int* nsurv = (int*)(ws + L.nsurv);
    int32_t* redo = (int32_t*)(ws + L.redo);
    float* surv_s = (float*)(ws + L.surv_s);
    int64_t* surv_i = (int64_t*)(ws + L.surv_i);
    unsigned long long* stats = (unsigned long long*)(ws + L.stats);
    const int64_t n_super = (L.n_groups + SUPER - 1) / SUPER;
    // (k > 10: the K = 36 in-block COLLECT instance needs 3 registers more than a 1 024-thread block may have; it takes the select kernel)
    const bool inblock = n_super <= TAIL_INBLOCK_MAX_SUPER && K == KSEL_SMALL;
    const size_t smem = (((size_t)D * 2 + 15) & ~(size_t)15) + (size_t)K * GROUP_ROWS * 12;
    constexpr int R1 = (256 * K + 1023) / 1024;
    if (!inblock) {
        dim3 grid(cdiv(nq, 64), L.nsplit);
        const int smem_sel = SEL_SPLIT_WAVES * K * 64 * 8;
        auto ksel = select_groups_kernel<K>;
        if (smem_sel > 48 * 1024) ARX_HIP_CHECK(arx_func_smem((const void*)ksel, smem_sel));
        ProfScope psc(ARX_K_SEARCH_SELECT, st);
        ksel<<<grid, 256, smem_sel, st>>>(gmax, L.ldg, n_super, L.n_groups, nq, L.nsplit, ps, pg, first_pass ? stats : nullptr);
        ARX_HIP_CHECK(hipGetLastError());
    }
    ProfScope psc(ARX_K_SEARCH_RESCORE, st);
    auto k_pl_c = tail_single_kernel<K, 1024, R1, false, true, 1>;
    auto k_pl_r = tail_single_kernel<K, 1024, R1, false, false, 1>;
    if (smem > 48 * 1024 && !inblock) {
        ARX_HIP_CHECK(arx_func_smem((const void*)k_pl_c, (int)smem));
        ARX_HIP_CHECK(arx_func_smem((const void*)k_pl_r, (int)smem));
    }
    if (inblock) {
        // small shard: ONE launch — selection, one row per selected group, the candidate pass and the merge inside the query's block
        // (tail_single_kernel<I8D>); the counters were zeroed by the query batch's quantisation kernel
        auto kd = tail_single_kernel<KSEL_SMALL, 1024, 1, true, false, 1, true>;
        const size_t smem_d = smem + I8D_EXTRA_SMEM;
        if (smem_d > 48 * 1024) ARX_HIP_CHECK(arx_func_smem((const void*)kd, (int)smem_d));
        kd<<<nq, 1024, smem_d, st>>>(gmax, aux, L.ldg, L.n_groups, nullptr, nullptr, 0, Q, C, n_rows, D, k, out_s, out_i, idx_base, tau_scale,
                                     debug_drop, stats, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
        ARX_HIP_CHECK(hipGetLastError());
        return ARX_OK;
    }
    k_pl_c<<<nq, 1024, smem, st>>>(gmax, aux, L.ldg, L.n_groups, ps, pg, L.nsplit, Q, C, n_rows, D, k, out_s, out_i, idx_base, tau_scale,
                                   debug_drop, nullptr, thr, selg, counters, nsurv, nullptr, nullptr);
    ARX_HIP_CHECK(hipGetLastError());
    if (int rc = launch_collect_rescore(L, ws, gmax, aux, n_rows, nq, thr, selg, K, Q, C, D, surv_s, surv_i, nsurv, counters, st)) return rc;
    merge_survivors_kernel<<<cdiv(nq, 4), 256, 0, st>>>(out_s, out_i, nq, k, idx_base, surv_s, surv_i, nsurv, counters, redo, stats);
    ARX_HIP_CHECK(hipGetLastError());
    // (merge_survivors already counted the queries that go through this)
    k_pl_r<<<nq, 1024, smem, st>>>(gmax, aux, L.ldg, L.n_groups, ps, pg, L.nsplit, Q, C, n_rows, D, k, out_s, out_i, idx_base, tau_scale,
                                   debug_drop, nullptr, nullptr, nullptr, nullptr, nullptr, redo, nullptr);
    ARX_HIP_CHECK(hipGetLastError());
    return ARX_OK;
}

template <int K, int NT>
static int run_select_rescore(const TopkWs& L, char* ws, const f16_t* Q, int nq, const f16_t* C, int64_t n_rows, int D,
                              int k, float* out_s, int64_t* out_i, int64_t idx_base, float tau_scale, int debug_drop, hipStream_t st,
                              bool collect, bool first_pass) {
    float* gmax = (float*)(ws + L.gmax);
    float* ps = (float*)(ws + L.part_s);
    int32_t* pg = (int32_t*)(ws + L.part_g);
    {
        dim3 grid(cdiv(nq, 64), L.nsplit);
        const int smem_sel = SEL_SPLIT_WAVES * K * 64 * 8;
        auto ksel = select_groups_kernel<K>;
        if (smem_sel > 48 * 1024) ARX_HIP_CHECK(arx_func_smem((const void*)ksel, smem_sel));
        ProfScope psc(ARX_K_SEARCH_SELECT, st);
        ksel<<<grid, 256, smem_sel, st>>>(gmax, L.ldg, (L.n_groups + SUPER - 1) / SUPER, L.n_groups, nq, L.nsplit, ps, pg,
                                          first_pass ? (unsigned long long*)(ws + L.stats) : nullptr);
        ARX_HIP_CHECK(hipGetLastError());
    }
    const int nslices = L.nsplit;
    const size_t smem = (((size_t)D * 2 + 15) & ~(size_t)15) + (size_t)K * GROUP_ROWS * 12;
    auto kern = rescore_kernel<K, NT>;
    if (smem > 48 * 1024) ARX_HIP_CHECK(arx_func_smem((const void*)kern, (int)smem));
    ProfScope psc(ARX_K_SEARCH_RESCORE, st);
    unsigned long long* stats = (unsigned long long*)(ws + L.stats);
    if (!collect) {
        kern<<<nq, NT, smem, st>>>(ps, pg, nslices, L.ldg, gmax, L.n_groups, Q, C, n_rows, D, k, out_s, out_i, idx_base, tau_scale,
                                   debug_drop, stats, nullptr, nullptr, nullptr, nullptr, nullptr);
        ARX_HIP_CHECK(hipGetLastError());
        return ARX_OK;
    }
    // int8 pre-filter: provisional top-k + threshold, then the candidate step, then (only for queries whose own
    // lists overflowed) the exhaustive kernel
    float* thr = (float*)(ws + L.thr);
    int32_t* selg = (int32_t*)(ws + L.selg);
    int* counters = (int*)(ws + L.counters);
    int* nsurv = (int*)(ws + L.nsurv);
    int32_t* redo = (int32_t*)(ws + L.redo);
    const uint32_t* aux = (const uint32_t*)(ws + L.aux);
    float* surv_s = (float*)(ws + L.surv_s);
    int64_t* surv_i = (int64_t*)(ws + L.surv_i);
    kern<<<nq, NT, smem, st>>>(ps, pg, nslices, L.ldg, gmax, L.n_groups, Q, C, n_rows, D, k, out_s, out_i, idx_base, tau_scale,
                               debug_drop, nullptr, thr, selg, nullptr, counters, nsurv);
    ARX_HIP_CHECK(hipGetLastError());
    if (int rc = launch_collect_rescore(L, ws, gmax, aux, n_rows, nq, thr, selg, K, Q, C, D, surv_s, surv_i, nsurv, counters, st)) return rc;
    merge_survivors_kernel<<<cdiv(nq, 4), 256, 0, st>>>(out_s, out_i, nq, k, idx_base, surv_s, surv_i, nsurv, counters, redo, stats);
    ARX_HIP_CHECK(hipGetLastError());
    kern<<<nq, NT, smem, st>>>(ps, pg, nslices, L.ldg, gmax, L.n_groups, Q, C, n_rows, D, k, out_s, out_i, idx_base, tau_scale,
                               debug_drop, nullptr, nullptr, nullptr, redo, nullptr, nullptr);      // (merge_survivors already counted these queries)
    ARX_HIP_CHECK(hipGetLastError());
    return ARX_OK;
}

// the single-kernel tail of a small fp16 batch (tail_single_kernel)
// fp16 pass: the aux epilogue costs pass A 3-4 % at <= 64 queries, 7 % at 128, 10-17 % at 256 (VALU work the HBM- / MFMA-bound loop does not
// hide); the tail saves 0.03 ms (<= 64 queries) to 0.08 ms (256) per batch.  Same box, same binary, arms interleaved
// (profiles/r04/tail_single_row_ab.md): a win at 625 k rows for every batch size (0.228 against 0.258 ms at 64 queries), break-even at
// 2 M rows, a loss beyond.  Hence: shards of up to TAIL_INBLOCK_MAX_SUPER super-groups (1 M rows).
static bool tail_single_ok(bool use_i8, int nq, int64_t n_rows, int k) {
    const int64_t n_groups = (n_rows + GROUP_ROWS - 1) / GROUP_ROWS, n_super = (n_groups + SUPER - 1) / SUPER;
    return !use_i8 && k <= 10 && nq <= AUX16_MAX_NQ && n_super <= TAIL_INBLOCK_MAX_SUPER;      // (k <= 10: 12 groups x 4 rows = one lane each)
}
template <int K>
static int run_tail_single(const TopkWs& L, char* ws, const f16_t* Q, int nq, const f16_t* C, int64_t n_rows, int D, int k, float* out_s,
                           int64_t* out_i, int64_t idx_base, float tau_scale, int debug_drop, hipStream_t st) {
    const size_t smem = (((size_t)D * 2 + 15) & ~(size_t)15) + (size_t)K * GROUP_ROWS * 12;
    ProfScope psc(ARX_K_SEARCH_RESCORE, st);
    auto kern = tail_single_kernel<K, 1024, 1, true, false, 4>;    // <= 1 024 super-groups: one candidate per lane (pass A zeroed the counters)
    if (smem > 48 * 1024) ARX_HIP_CHECK(arx_func_smem((const void*)kern, (int)smem));
    kern<<<nq, 1024, smem, st>>>((const float*)(ws + L.gmax), (const uint32_t*)(ws + L.aux), L.ldg, L.n_groups, nullptr, nullptr, 0, Q, C, n_rows,
                                 D, k, out_s, out_i, idx_base, tau_scale, debug_drop, (unsigned long long*)(ws + L.stats), nullptr, nullptr,
                                 nullptr, nullptr, nullptr, nullptr);
    ARX_HIP_CHECK(hipGetLastError());
    return ARX_OK;
}

// query batches of more than this take the fp16 pass even when an int8 index is given (arx_topk_options.i8_max_queries).  Default = every
// batch size: with single-row candidates (aux word) and block-aggregated candidate lists the int8 pass wins at every Qb measured
// (10 M x 768, same box: 1.67x at Qb = 1, 1.45x at 64, 1.31x at 256, 1.40x at 1 024: profiles/r03); round 2's crossover was 128.
#define I8_MAX_NQ_DEFAULT QBATCH_MAX

// The library keeps no search policy of its own (round 3 had three process-wide knobs here): a call's policy is its options argument.
struct TopkPolicy { int i8_max_nq; float norm_bound; int cu_limit; int flags; float tau_mult; int drop; };
static int resolve_policy(const arx_topk_options* opt, TopkPolicy& P) {
    P = TopkPolicy{I8_MAX_NQ_DEFAULT, 1.0f + 1.0f / 512.0f, 0, 0, 1.0f, 0};
    if (!opt) return ARX_OK;
    ARX_REQUIRE(opt->struct_bytes == (int32_t)sizeof(arx_topk_options), "arx_topk_options.struct_bytes=%d, this library expects %d",
                opt->struct_bytes, (int)sizeof(arx_topk_options));
    if (opt->i8_max_queries != 0) P.i8_max_nq = opt->i8_max_queries < 0 ? 0 : opt->i8_max_queries;
    if (opt->max_row_norm != 0.0f) {
        ARX_REQUIRE(opt->max_row_norm > 0.0f && opt->max_row_norm < INFINITY, "max_row_norm=%g: must be a finite positive bound",
                    (double)opt->max_row_norm);
        P.norm_bound = opt->max_row_norm;
    }
    ARX_REQUIRE(opt->cu_limit >= 0, "cu_limit=%d", opt->cu_limit);
    P.cu_limit = opt->cu_limit;
    P.flags = opt->flags;
    if (opt->debug_tau_mult != 0.0f) {
        ARX_REQUIRE(opt->debug_tau_mult >= 1.0f, "debug_tau_mult=%g: the certificate tolerance may only be widened (>= 1)",
                    (double)opt->debug_tau_mult);
        P.tau_mult = opt->debug_tau_mult;
    }
    P.drop = opt->debug_drop_best ? 1 : 0;
    return ARX_OK;
}

static int topk_search_impl(const void* corpus, const void* index_i8, int64_t n_rows, const void* queries, int32_t n_queries, int32_t dim,
                            int32_t k, float* out_scores, int64_t* out_ids, int64_t idx_base, void* ws,
                            int64_t ws_bytes, const arx_topk_options* opt, void* stream) {
    TopkPolicy P;
    if (int prc = resolve_policy(opt, P)) return prc;
    ARX_REQUIRE(corpus && queries && out_scores && out_ids && ws, "null pointer argument");
    ARX_REQUIRE(!index_i8 || (dim % 128 == 0 && dim <= 1024), "int8 pre-filter: dim=%d must be a multiple of 128, <= 1024", dim);
    ARX_REQUIRE(!index_i8 || n_rows < (1ll << 32), "int8 pre-filter: row candidates are 32-bit");
    ARX_REQUIRE(n_rows > 0 && n_queries > 0, "empty corpus or query set");
    ARX_REQUIRE(dim > 0 && dim % 64 == 0 && dim <= 8192, "dim=%d must be a multiple of 64", dim);
    ARX_REQUIRE(k > 0 && k <= KMAX, "k=%d out of range 1..%d", k, KMAX);
    const TopkWs L = topk_layout(n_rows, n_queries, k, dim, index_i8 != nullptr);
    ARX_REQUIRE(ws_bytes >= L.total, "workspace too small: %lld < %lld (arx_topk_workspace_bytes%s)", (long long)ws_bytes, (long long)L.total,
                index_i8 ? "_i8" : "");
    ARX_REQUIRE(!(P.flags & (ARX_TOPK_SCAN_ONLY | ARX_TOPK_TAIL_ONLY)) || n_queries <= QBATCH_MAX,
                "a split (scan / tail) search takes at most %d queries per call", QBATCH_MAX);
    ARX_REQUIRE((P.flags & (ARX_TOPK_SCAN_ONLY | ARX_TOPK_TAIL_ONLY)) != (ARX_TOPK_SCAN_ONLY | ARX_TOPK_TAIL_ONLY), "scan-only and tail-only together");
    hipStream_t st = (hipStream_t)stream;
    const f16_t* C = (const f16_t*)corpus;
#ifdef ARX_DEV_VARIANTS
    const char* genv = getenv("ARX_GEMM_GLDS");
    const bool glds = !(genv && genv[0] == '0');
#endif
    // certificate tolerance (rescore_kernel step 5): eps_A + eps_B <= (0.3125 D + 4) 2^-24 |q|_2 |c|_2 (every rounding of either pass is
    // relative to a partial sum bounded by sum |q_i c_i| <= |q|_2 |c|_2), |c|_2 <= the caller's bound on the shard's row norms
    const float tau_scale = (0.3125f * (float)dim + 4.0f) * 5.9604645e-8f * P.norm_bound * P.tau_mult;
    const int debug_drop = P.drop;
    for (int q0 = 0; q0 < n_queries; q0 += QBATCH_MAX) {
        const int nq = (n_queries - q0) < QBATCH_MAX ? (n_queries - q0) : QBATCH_MAX;
        const f16_t* Q = (const f16_t*)queries + (int64_t)q0 * dim;
        float* gmax = (float*)((char*)ws + L.gmax);
        int rc = ARX_OK;
        const bool use_i8 = index_i8 && nq <= P.i8_max_nq;          // arx_topk_options.i8_max_queries
        // (decided on the CALL's query count, like the workspace layout: a call of more than 1 024 queries whose last internal pass is small
        // has no aux region to write — found by tools/search_soak.py as a memory fault, 1 100 queries on a 100 k-row shard)
        const bool single = tail_single_ok(use_i8, n_queries, n_rows, k) && !(P.flags & ARX_TOPK_NO_SINGLE_ROW_TAIL);
        const bool i8_single = use_i8 && !(P.flags & ARX_TOPK_NO_SINGLE_ROW_TAIL);          // the int8 pipeline's first step, one row per selected group
        if (P.flags & ARX_TOPK_TAIL_ONLY) {
            // pass A of this batch ran in an earlier ARX_TOPK_SCAN_ONLY call on this workspace (the caller ordered the two streams)
        } else if (use_i8) {                           // pass A over the int8 representation: upper bounds instead of scores, everything after it unchanged
            int8_t* q8 = (int8_t*)((char*)ws + L.q8);
            float2* qmeta = (float2*)((char*)ws + L.qmeta);
            quantize_rows_i8_kernel<<<cdiv(nq, 4), 256, 0, st>>>(Q, nq, dim, q8, qmeta, q0 == 0 ? (unsigned long long*)((char*)ws + L.stats) : nullptr);
            ARX_HIP_CHECK(hipGetLastError());
            const int8_t* C8 = (const int8_t*)index_i8;
            const float2* cmeta = (const float2*)((const char*)index_i8 + i8_meta_offset(n_rows, dim));
            ProfScope ps(ARX_K_SEARCH_GROUPMAX, st);
            uint32_t* aux = (uint32_t*)((char*)ws + L.aux);
            if (persistent_pass_ok(nq, n_rows, dim / 2, P.flags))
                rc = launch_groupmax_persistent<i8pair_t, true>((const i8pair_t*)q8, nq, (const i8pair_t*)C8, n_rows, dim / 2, dim, qmeta, cmeta, gmax,
                                                                aux, L.ldg, P.cu_limit, st);
            else
            rc = nq <= 64 ? launch_groupmax_i8<64, true>(q8, qmeta, nq, C8, cmeta, n_rows, dim, gmax, aux, L.ldg, st)
               : nq <= 128 ? launch_groupmax_i8<128, true>(q8, qmeta, nq, C8, cmeta, n_rows, dim, gmax, aux, L.ldg, st)
                           : launch_groupmax_i8<256, true>(q8, qmeta, nq, C8, cmeta, n_rows, dim, gmax, aux, L.ldg, st);
        } else {
        ProfScope ps(ARX_K_SEARCH_GROUPMAX, st);
        uint32_t* aux16 = single ? (uint32_t*)((char*)ws + L.aux) : nullptr;                 // small fp16 batch: aux words for the single-kernel tail
        unsigned long long* zs16 = (single && q0 == 0) ? (unsigned long long*)((char*)ws + L.stats) : nullptr;
#ifdef ARX_DEV_VARIANTS
        if (!glds) {
            rc = nq <= 64 ? launch_groupmax<64, false>(Q, nq, C, n_rows, dim, gmax, L.ldg, aux16, zs16, st)
               : nq <= 128 ? launch_groupmax<128, false>(Q, nq, C, n_rows, dim, gmax, L.ldg, aux16, zs16, st)
                           : launch_groupmax<256, false>(Q, nq, C, n_rows, dim, gmax, L.ldg, aux16, zs16, st);
        } else
#endif
            if (persistent_pass_ok(nq, n_rows, dim, P.flags))
                rc = single ? launch_groupmax_persistent<f16_t, false, true>(Q, nq, C, n_rows, dim, dim, nullptr, nullptr, gmax, aux16, L.ldg, P.cu_limit, st, zs16)
                            : launch_groupmax_persistent<f16_t, false>(Q, nq, C, n_rows, dim, dim, nullptr, nullptr, gmax, nullptr, L.ldg, P.cu_limit, st);
            else
            rc = nq <= 64 ? launch_groupmax<64, true>(Q, nq, C, n_rows, dim, gmax, L.ldg, aux16, zs16, st)
               : nq <= 128 ? launch_groupmax<128, true>(Q, nq, C, n_rows, dim, gmax, L.ldg, aux16, zs16, st)
                           : launch_groupmax<256, true>(Q, nq, C, n_rows, dim, gmax, L.ldg, aux16, zs16, st);
        }
        if (rc != ARX_OK) return rc;
        if (P.flags & ARX_TOPK_SCAN_ONLY) continue;
        float* os = out_scores + (int64_t)q0 * k;
        int64_t* oi = out_ids + (int64_t)q0 * k;
        // rescore geometry (same-box A/B, r02): a 16-wave block per query is fastest while the blocks fit the chip at once
        // (0.10 vs 0.13 ms at <= 64 queries); 4-wave blocks, eight to a CU, when there are thousands (2.9 vs 5.4 ms per 10 k)
        if (i8_single) rc = k > 10 ? run_i8_single<KSEL_BIG>(L, (char*)ws, Q, nq, C, n_rows, dim, k, os, oi, idx_base, tau_scale, debug_drop, st, q0 == 0)
                                   : run_i8_single<KSEL_SMALL>(L, (char*)ws, Q, nq, C, n_rows, dim, k, os, oi, idx_base, tau_scale, debug_drop, st, q0 == 0);
        else if (single) rc = run_tail_single<KSEL_SMALL>(L, (char*)ws, Q, nq, C, n_rows, dim, k, os, oi, idx_base, tau_scale, debug_drop, st);
        else if (k > 10) rc = run_select_rescore<KSEL_BIG, 1024>(L, (char*)ws, Q, nq, C, n_rows, dim, k, os, oi, idx_base, tau_scale, debug_drop, st, use_i8, q0 == 0);
        else if (nq <= 128) rc = run_select_rescore<KSEL_SMALL, 1024>(L, (char*)ws, Q, nq, C, n_rows, dim, k, os, oi, idx_base, tau_scale, debug_drop, st, use_i8, q0 == 0);
        else rc = run_select_rescore<KSEL_SMALL, 256>(L, (char*)ws, Q, nq, C, n_rows, dim, k, os, oi, idx_base, tau_scale, debug_drop, st, use_i8, q0 == 0);
        if (rc != ARX_OK) return rc;
    }
    return ARX_OK;
}

extern "C" int32_t arx_topk_search(const void* corpus, int64_t n_rows, const void* queries, int32_t n_queries, int32_t dim,
                                   int32_t k, float* out_scores, int64_t* out_ids, int64_t idx_base, void* ws,
                                   int64_t ws_bytes, void* stream) {
    return topk_search_impl(corpus, nullptr, n_rows, queries, n_queries, dim, k, out_scores, out_ids, idx_base, ws, ws_bytes, nullptr, stream);
}

extern "C" int32_t arx_topk_search_i8(const void* corpus, const void* index_i8, int64_t n_rows, const void* queries, int32_t n_queries,
                                      int32_t dim, int32_t k, float* out_scores, int64_t* out_ids, int64_t idx_base, void* ws,
                                      int64_t ws_bytes, void* stream) {
    ARX_REQUIRE(index_i8, "null int8 index");
    return topk_search_impl(corpus, index_i8, n_rows, queries, n_queries, dim, k, out_scores, out_ids, idx_base, ws, ws_bytes, nullptr, stream);
}

extern "C" int32_t arx_topk_search_opt(const void* corpus, const void* index_i8, int64_t n_rows, const void* queries, int32_t n_queries,
                                       int32_t dim, int32_t k, float* out_scores, int64_t* out_ids, int64_t idx_base, void* ws,
                                       int64_t ws_bytes, const arx_topk_options* opt, void* stream) {
    return topk_search_impl(corpus, index_i8, n_rows, queries, n_queries, dim, k, out_scores, out_ids, idx_base, ws, ws_bytes, opt, stream);
}

// max row norm of a shard: one wave per row, fp32 sum of squares; non-negative floats order like their bit patterns, so the maximum is an
// integer atomicMax (a NaN row's pattern is above +inf's and survives: the caller refuses such a shard)
__global__ __launch_bounds__(256) void rows_max_norm_kernel(const f16_t* __restrict__ X, int64_t n_rows, int D, uint32_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    float best = 0.f;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < n_rows; row += (int64_t)gridDim.x * 4) {
        const f16_t* x = X + row * D;
        float a = 0.f;
        for (int c = lane * 8; c < D; c += 512) {
            const f16x8 v = *reinterpret_cast<const f16x8*>(x + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) a = fmaf((float)v[e], (float)v[e], a);
        }
        a = wave_sum(a);
        best = (a > best || a != a) ? a : best;
    }
    if (lane == 0) {
        const float nrm = sqrtf(best) * (1.0f + 1.0f / 4096.0f);      // rounded UP: the fp32 sum's error is below D 2^-24 relative
        atomicMax(out, __float_as_uint(nrm));
    }
}
extern "C" int32_t arx_rows_max_norm_f16(const void* rows, int64_t n_rows, int32_t dim, float* out_max, void* stream) {
    ARX_REQUIRE(rows && out_max && n_rows > 0, "bad args");
    ARX_REQUIRE(dim > 0 && dim % 8 == 0, "dim=%d must be a multiple of 8", dim);
    hipStream_t st = (hipStream_t)stream;
    ARX_HIP_CHECK(hipMemsetAsync(out_max, 0, 4, st));
    const int64_t want = (n_rows + 3) / 4;
    const int blocks = (int)(want < 8192 ? want : 8192);
    rows_max_norm_kernel<<<blocks, 256, 0, st>>>((const f16_t*)rows, n_rows, dim, (uint32_t*)out_max);
    ARX_HIP_CHECK(hipGetLastError());
    return ARX_OK;
}

extern "C" int32_t arx_topk_stats(const void* ws, int64_t* flagged_queries, int64_t* extra_groups, void* stream) {
    ARX_REQUIRE(ws && flagged_queries && extra_groups, "null pointer argument");
    unsigned long long h[2] = {0, 0};
    ARX_HIP_CHECK(hipMemcpyAsync(h, ws, 16, hipMemcpyDeviceToHost, (hipStream_t)stream));
    ARX_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    *flagged_queries = (int64_t)h[0]; *extra_groups = (int64_t)h[1];
    return ARX_OK;
}

extern "C" int32_t arx_topk_merge(const float* scores, const int64_t* ids, int32_t n_parts, int32_t n_queries, int32_t k,
                                  float* out_scores, int64_t* out_ids, void* stream) {
    ARX_REQUIRE(scores && ids && out_scores && out_ids, "null pointer argument");
    ARX_REQUIRE(n_parts > 0 && n_queries > 0 && k > 0 && k <= KMAX, "bad sizes");
    ARX_REQUIRE((int64_t)n_parts * k <= 64 * 64, "too many candidates per query (%d parts x k=%d)", n_parts, k);
    merge_kernel<<<cdiv(n_queries, 4), 256, 0, (hipStream_t)stream>>>(scores, ids, n_parts, n_queries, k, out_scores, out_ids);
    ARX_HIP_CHECK(hipGetLastError());
    return ARX_OK;
}

extern "C" int32_t arx_fill_unit_rows_f16_at(void* dst, int64_t n_rows, int32_t dim, uint64_t seed, int64_t row_base, void* stream) {
    ARX_REQUIRE(dst && n_rows > 0 && row_base >= 0, "bad args");
    ARX_REQUIRE(dim % 128 == 0 && dim <= 1024, "dim=%d must be a multiple of 128, <= 1024", dim);
    const int64_t blocks = (n_rows + 3) / 4;
    ARX_REQUIRE(blocks < (1ll << 31), "too many rows for one launch");
    fill_unit_rows_kernel<<<(int)blocks, 256, 0, (hipStream_t)stream>>>((f16_t*)dst, n_rows, dim, seed, row_base);
    ARX_HIP_CHECK(hipGetLastError());
    return ARX_OK;
}

extern "C" int32_t arx_fill_unit_rows_f16(void* dst, int64_t n_rows, int32_t dim, uint64_t seed, void* stream) {
    return arx_fill_unit_rows_f16_at(dst, n_rows, dim, seed, 0, stream);
}

extern "C" int32_t arx_fill_clustered_rows_f16_at(void* dst, int64_t n_rows, int32_t dim, uint64_t seed, int64_t row_base, int32_t n_clusters,
                                                  float spread, int32_t n_hot_dims, float hot_gain, void* stream) {
    ARX_REQUIRE(dst && n_rows > 0 && row_base >= 0, "bad args");
    ARX_REQUIRE(dim % 128 == 0 && dim <= 1024, "dim=%d must be a multiple of 128, <= 1024", dim);
    ARX_REQUIRE(n_clusters != 0 && n_clusters <= (1 << 20) && n_clusters >= -(1 << 20) && spread >= 0.f && n_hot_dims >= 0 && n_hot_dims <= 16 &&
                hot_gain > 0.f, "bad mixture parameters");
    const int64_t blocks = (n_rows + 3) / 4;
    ARX_REQUIRE(blocks < (1ll << 31), "too many rows for one launch");
    fill_clustered_rows_kernel<<<(int)blocks, 256, 0, (hipStream_t)stream>>>((f16_t*)dst, n_rows, dim, seed, row_base, n_clusters, spread,
                                                                              n_hot_dims, hot_gain);
    ARX_HIP_CHECK(hipGetLastError());
    return ARX_OK;
}
