// Pass A of arx_topk_* (search.hip): the scan of the shard.  Per-tile kernels for narrow query batches (HBM-bound), the persistent 4-phase
// form above the ridge point, their epilogues (group maximum; group maximum + aux word; int8 upper bounds + aux word) and the int8
// quantisation of rows.  Included by search.hip after its constants; not a stand-alone header.
#pragma once

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
//   qo_of(i): the query's OFFSET — an upper bound of q . mu, mu the vector the index subtracted from every row before quantising it (the
//   shard's mean, arx_topk_build_i8): the int8 machinery bounds q . (c - mu), the offset makes it a bound of q . c again.
//   R1 (the query was centred too, ARX_TOPK_I8_CENTRE_QUERY): ct_of(j, t4) gives t_c = m^ . (c - mu) of the lane's rows, qg_of(i) the
//   query's gamma / s_q; one more fma per element adds the rank-one term gamma t_c (see quantize_rows_i8_kernel).
template <int MI, int NI, bool R1, typename CM, typename QM, typename QO, typename CT, typename QG>
__device__ __forceinline__ void groupmax_epilogue_i8(const f32x4 (&acc)[NI][MI], CM cm_of, QM qm_of, QO qo_of, CT ct_of, QG qg_of, int D,
                                                     float* __restrict__ gmax_row, uint32_t* __restrict__ aux_row, int m_first, int nq, int lane) {
    float sc[NI][4], xc[NI][4], tc[R1 ? NI : 1][4];
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
        if constexpr (R1) ct_of(j, tc[j]);
    }
    float gm[MI];
    uint32_t ga[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const float2 qm = qm_of(i);
        const int cqi = (int)ceilf(0.5001f * qm.y) + 1;
        float gq = 0.f;
        if constexpr (R1) gq = qg_of(i);
        // top-2 of the 16 bounds with the arg-max for free: the low 6 bits of each value's float image are REPLACED by the row's
        // position in the group (j*16 + r; the lane's 4-row offset is OR-ed in after the lane-local pass), which moves a value by at
        // most 63 ulp either way — covered by the 2^-17 allowance below — and lets v_max / v_med3 carry the index along.
        float m1 = -INFINITY, m2 = -INFINITY;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const i32x4 it = __builtin_bit_cast(i32x4, acc[j][i]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float u = fmaf(sc[j][r], (float)(it[r] + cqi), xc[j][r]);
                if constexpr (R1) u = fmaf(gq, tc[j][r], u);
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
        const float qo = qo_of(i);
        b1 += qo; b2 += qo;                                              // + the offset (itself an upper bound); the additions' roundings:
        b1 += fabsf(b1) * 1.2e-7f; b2 += fabsf(b2) * 1.2e-7f;
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
struct TFromLds {                                    // t_c of the lane's rows: [256 floats] behind the corpus pairs
    const float* meta; int wn, lrow;
    __device__ __forceinline__ void operator()(int j, float (&t4)[4]) const {
        const f32x4 v = *reinterpret_cast<const f32x4*>(meta + 512 + wn * GROUP_ROWS + j * 16 + lrow);
        t4[0] = v[0]; t4[1] = v[1]; t4[2] = v[2]; t4[3] = v[3];
    }
};
struct NoT { __device__ __forceinline__ void operator()(int, float (&)[4]) const {} };
#define I8_META_Q_OFF 768                            // LDS stage: [512] corpus (s, L1) pairs, [256] corpus t, then the queries' [2 BM] pairs, [BM] offsets, [BM] gamma / s_q
// one 4-byte LDS-DMA per thread stages the tile's corpus pairs, one more its query pairs (BM queries from m0), one more (threads < BM) the
// queries' offsets; R1: also the rows' t and the queries' gamma / s_q
template <int BM, bool R1>
__device__ __forceinline__ void stage_i8_meta(const float2* __restrict__ cmeta, const float* __restrict__ ctrow, const float2* __restrict__ qmeta,
                                              const float* __restrict__ qoff, const float* __restrict__ qg,
                                              int64_t n0, int64_t n_rows, int m0, int nq, float* meta, int tid) {
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
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(reinterpret_cast<const float*>(qmeta) + e), (lds_void_t*)(meta + I8_META_Q_OFF + (tid & ~63)), 4, 0, 0);
    }
    if (tid < BM) {                                                        // wave-uniform
        int e = m0 + tid;
        e = e < nq ? e : nq - 1;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(qoff + e), (lds_void_t*)(meta + I8_META_Q_OFF + 2 * BM + (tid & ~63)), 4, 0, 0);
        if constexpr (R1)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(qg + e), (lds_void_t*)(meta + I8_META_Q_OFF + 3 * BM + (tid & ~63)), 4, 0, 0);
    }
    if constexpr (R1) {
        if (tid < 256) {
            int64_t e = n0 + tid;
            e = e < n_rows ? e : n_rows - 1;
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(ctrow + e), (lds_void_t*)(meta + 512 + (tid & ~63)), 4, 0, 0);
        }
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
//
// CENTRED rows: every slack term above is proportional to s_c, i.e. to the largest |x_i| of the row that is quantised.  Rows that share a
// large common component (anisotropic embeddings: the encoder's own rows under seeded weights have a mean pairwise cosine of 0.98) differ
// from each other by much less than their own size, and a bound with unit-row slack cannot tell them apart.  The index therefore
// quantises c - mu (mu: the mean of a sample of the shard's rows, `sub_mu`; ANY vector keeps the bound rigorous):
//       q.c = q.mu + q.(c - mu) ,   fp32(c_i - mu_i) = (c_i - mu_i)(1 + d), |d| <= 2^-24: at most 127.5 s 2^-24 of the 0.0001 s allowance
// and a QUERY batch gets, beside its int8 form, the offset q.mu rounded UP (`dot_mu` -> `qoff`; the fp32 chain's error is below
// 18 roundings x 2^-24 relative to sum |q_i mu_i|), which pass A adds to every bound of that query.  On rows without a common component
// mu ~ 0 and nothing changes.
//
// CENTRED QUERIES (ARX_TOPK_I8_CENTRE_QUERY; the caller's choice per call, sensible when |mu| is large): the slack terms proportional to the
// QUERY's quantisation step and L1 norm still see the whole query.  For any unit-ish vector m^ (the index keeps mu / |mu|) and scalar gamma
//       q.(c - mu) = gamma t_c + (q - gamma m^).(c - mu) ,   t_c = m^.(c - mu)   (an identity in real arithmetic),
// so with gamma = q.m^ the query that is quantised is q' = q - gamma m^ (what distinguishes it from the common direction), the int8
// machinery bounds q'.(c - mu), and the rank-one term gamma t_c — neither a per-query nor a per-row constant — is added EXACTLY per element
// in pass A's epilogue (one fma with gamma / s_q and the row's t_c, which the index stores).  Roundings: q'_i = fma(-gamma, m^_i, q_i) is one
// rounding of the exact value (inside the 0.0001 s_q allowance, as for the rows); t_c as computed differs from m^.(c - mu) by at most
// 2.2e-6 sum |c'_i m^_i| <= 2.2e-6 E (E: the largest such sum over the shard, kept in the index header), gamma / s_q and the epilogue's fma
// add 2.4e-7 |gamma| T (T: the largest |t_c|): both go into the query's offset (with the first of the epilogue's two roundings, see the kernel's end).  A query that IS the common direction (q' = 0, s_q = 0) gets
// |gamma| T in its offset instead of the rank-one term.
struct I8Header { float mu_norm, t_max, e_max, c_amax; };      // |mu|; max |t_c|; max sum |c'_i m^_i|; max |c'_i| — over the shard's rows
__global__ __launch_bounds__(256) void quantize_rows_i8_kernel(const f16_t* __restrict__ X, int64_t n_rows, int D, int8_t* __restrict__ X8,
                                                                float2* __restrict__ meta, unsigned long long* __restrict__ zero_stats,
                                                                const float* __restrict__ sub_mu, const float* __restrict__ dot_mu,
                                                                float* __restrict__ qoff, const float* __restrict__ mhat,
                                                                float* __restrict__ trow, I8Header* __restrict__ hdr_w,
                                                                const I8Header* __restrict__ hdr_r, float* __restrict__ qg, int centre_q) {
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
        if (sub_mu) { v[e] -= sub_mu[lane * per + e]; v[e + 1] -= sub_mu[lane * per + e + 1]; }
        amax = fmaxf(amax, fmaxf(fabsf(v[e]), fabsf(v[e + 1])));
    }
    float off = 0.f, gamma = 0.f;
    if (trow) {                                    // (a corpus row, already minus mu: its t_c = m^ . c' for centred queries, and the header's maxima)
        float t = 0.f, mag = 0.f;
        for (int e = 0; e < per; ++e) {
            const float m = mhat[lane * per + e];
            t = fmaf(v[e], m, t); mag = fmaf(fabsf(v[e]), fabsf(m), mag);
        }
        t = wave_sum(t); mag = wave_sum(mag);
        const float amax_row = wave_max(amax);
        if (lane == 0) {
            trow[row] = t;
            atomicMax(reinterpret_cast<unsigned int*>(&hdr_w->t_max), __float_as_uint(fabsf(t)));      // non-negative floats order as their bit patterns
            atomicMax(reinterpret_cast<unsigned int*>(&hdr_w->e_max), __float_as_uint(mag));
            atomicMax(reinterpret_cast<unsigned int*>(&hdr_w->c_amax), __float_as_uint(amax_row));
        }
    }
    if (dot_mu) {                                  // (a query row: its offset q . mu, rounded up)
        float dot = 0.f, mag = 0.f;
        for (int e = 0; e < per; ++e) {
            const float m = dot_mu[lane * per + e];
            dot = fmaf(v[e], m, dot); mag = fmaf(fabsf(v[e]), fabsf(m), mag);
        }
        dot = wave_sum(dot); mag = wave_sum(mag);
        off = dot + mag * 2.0e-6f + 1e-30f;
        if (centre_q) {                            // q' = q - gamma m^ is what gets quantised
            float g = 0.f;
            for (int e = 0; e < per; ++e) g = fmaf(v[e], mhat[lane * per + e], g);
            gamma = wave_sum(g);
            amax = 0.f;
            for (int e = 0; e < per; ++e) {
                v[e] = fmaf(-gamma, mhat[lane * per + e], v[e]);
                amax = fmaxf(amax, fabsf(v[e]));
            }
            off += fabsf(gamma) * (2.2e-6f * hdr_r->e_max + 2.4e-7f * hdr_r->t_max);
        }
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
    if (lane == 0) {
        meta[row] = float2{s, l1};
        if (dot_mu) {
            if (centre_q && !(s > 0.f)) off += fabsf(gamma) * hdr_r->t_max * 1.0000003f;       // q' = 0: no int8 form to carry gamma t_c
            // the epilogue's value is now the sum of TWO roundings (u1 = the int8 bound, then + (gamma / s_q) t_c): where the two nearly cancel, the
            // first one's error, 2^-24 |u1|, is not small relative to the result.  |u1| s_q <= s_q s_c (127.5 L1(q8) + 64 D + 2), s_c <= c_amax / 127
            if (centre_q && s > 0.f) off += 1.2e-7f * s * (hdr_r->c_amax * (1.0f / 127.0f)) * (127.5f * l1 + 64.0f * (float)D + 2.0f);
            qoff[row] = off;
            if (qg) qg[row] = (centre_q && s > 0.f) ? gamma / s : 0.f;
        }
    }
}

// mu for the centred int8 index: the mean of up to MEAN_SAMPLE_ROWS evenly spaced rows, summed in a FIXED order (a block per 64 columns, a
// wave per residue class of the sample, its rows in sequence, the four waves' sums added in order): the same shard always gives the same mu,
// hence the same index and the same candidate lists.
#define MEAN_SAMPLE_ROWS 16384
__global__ __launch_bounds__(256) void rows_mean_kernel(const f16_t* __restrict__ X, int64_t n_rows, int D, float* __restrict__ mu) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    const int64_t S = n_rows < MEAN_SAMPLE_ROWS ? n_rows : MEAN_SAMPLE_ROWS;
    float acc = 0.f;
    if (col < D)
        for (int64_t j = w; j < S; j += 4) {
            const int64_t r = j * n_rows / S;
            acc += (float)X[r * D + col];
        }
    part[w][lane] = acc;
    __syncthreads();
    if (w == 0 && col < D) mu[col] = (((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane]) / (float)S;
}
// m^ = mu / |mu| (zero when mu is) and the header (|mu|; the maxima start at zero: quantize_rows_i8_kernel raises them).  One block, fixed order.
__global__ __launch_bounds__(256) void mean_finish_kernel(const float* __restrict__ mu, int D, float* __restrict__ mhat, I8Header* __restrict__ hdr) {
    __shared__ float part[256];
    float a = 0.f;
    for (int c = threadIdx.x; c < D; c += 256) a = fmaf(mu[c], mu[c], a);
    part[threadIdx.x] = a;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    const float nrm = sqrtf(part[0]);
    const float inv = nrm > 1e-20f ? 1.0f / nrm : 0.f;
    for (int c = threadIdx.x; c < D; c += 256) mhat[c] = mu[c] * inv;
    if (threadIdx.x == 0) *hdr = I8Header{nrm, 0.f, 0.f, 0.f};
}

template <int BM, bool GLDS, bool R1>
__global__ __launch_bounds__(512) void search_groupmax_i8_kernel(const int8_t* __restrict__ Q8, const float2* __restrict__ qmeta,
                                                                  const float* __restrict__ qoff, const float* __restrict__ qg, int nq,
                                                                  const int8_t* __restrict__ C8, const float2* __restrict__ cmeta,
                                                                  const float* __restrict__ ctrow,
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
    float* const meta = reinterpret_cast<float*>(smem + MAIN_BYTES);          // the stage of stage_i8_meta
    const int lrow = (lane >> 4) * 4;
    // (the 16 corpus pairs a lane's accumulators belong to are the same for the 16 lanes of a row of the wave: lane t of the row loads ONE
    // pair — row (t>>2)*16 + lrow + (t&3) of the group — and the epilogue fetches the sixteen by ds_bpermute.  Sixteen 8-byte loads per lane
    // were a third of all the vector-memory requests of a D = 768 tile, half of them at D = 384: 5.8 / 5.0 TB/s where the fp16 pass, the
    // same bytes per tile at D = 384, reaches 6.3; after: 6.0-6.15 / 5.3-5.5, and 5.5 from 5.25 at 64 queries.  A STREAMED form of this kernel —
    // a block walking four corpus tiles with its loads running through the tile boundaries and the epilogue under the next tile's first
    // k-tile — was built beside it and measured: +0-2 % at D = 768, +6-9 % at D = 384, -2 % for the fp16 rows; not kept.
    // profiles/r04/pass_a_narrow_int8_ab.md)
    float2 cmr = float2{0.f, 0.f}, qmr[META_LDS ? 1 : ML::MI];
    float qor[META_LDS ? 1 : ML::MI], qgr[(META_LDS || !R1) ? 1 : ML::MI], ctr = 0.f;
    if constexpr (META_LDS) {
        stage_i8_meta<BM, R1>(cmeta, ctrow, qmeta, qoff, qg, n0, n_rows, m0, nq, meta, threadIdx.x);
    } else {
        {
            const int t = lane & 15;
            const int64_t n = n0 + wn * GROUP_ROWS + (t >> 2) * 16 + lrow + (t & 3);
            cmr = cmeta[n < n_rows ? n : n_rows - 1];
            if constexpr (R1) ctr = ctrow[n < n_rows ? n : n_rows - 1];
        }
#pragma unroll
        for (int i = 0; i < ML::MI; ++i) {
            const int m = m0 + wm * ML::TM + i * 16 + (lane & 15);
            qmr[i] = qmeta[m < nq ? m : nq - 1];
            qor[i] = qoff[m < nq ? m : nq - 1];
            if constexpr (R1) qgr[i] = qg[m < nq ? m : nq - 1];
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
        const float* qmeta_l = meta + I8_META_Q_OFF + (wm * ML::TM + (lane & 15)) * 2;
        groupmax_epilogue_i8<ML::MI, ML::NI, R1>(acc, MetaFromLds{meta, wn, lrow},
                                                 [&](int i) { return *reinterpret_cast<const float2*>(qmeta_l + i * 32); },
                                                 [&](int i) { return meta[I8_META_Q_OFF + 2 * BM + wm * ML::TM + i * 16 + (lane & 15)]; },
                                                 TFromLds{meta, wn, lrow},
                                                 [&](int i) { return meta[I8_META_Q_OFF + 3 * BM + wm * ML::TM + i * 16 + (lane & 15)]; },
                                                 D, gmax + g * ldg, aux + g * ldg, m0 + wm * ML::TM, nq, lane);
    } else {
        groupmax_epilogue_i8<ML::MI, ML::NI, R1>(acc, [&](int j, float2 (&c4)[4]) {
#pragma unroll
                                                     for (int r = 0; r < 4; ++r) {
                                                         const int src = (lane & 48) | (j * 4 + r);
                                                         c4[r] = float2{__shfl(cmr.x, src), __shfl(cmr.y, src)};
                                                     }
                                                 },
                                                 [&](int i) { return qmr[i]; }, [&](int i) { return qor[i]; },
                                                 [&](int j, float (&t4)[4]) {
#pragma unroll
                                                     for (int r = 0; r < 4; ++r) t4[r] = __shfl(ctr, (lane & 48) | (j * 4 + r));
                                                 },
                                                 [&](int i) { return qgr[R1 ? i : 0]; }, D, gmax + g * ldg, aux + g * ldg, m0 + wm * ML::TM, nq, lane);
    }
}

// ---- pass A, persistent form (>= 256 queries, even number of k-tiles): gemm8.h's persistent 4-phase loop with the pass-A epilogues.
// Above the ridge point a 256 x 256 x D tile is SHORT (12 k-tiles of f16, 6 of int8 at D = 768): the per-tile kernel pays the first
// loads' latency, an idle matrix pipe during the epilogue and a block launch per tile — 29 k cycles per int8 tile against 6 k of matrix
// work (profiles/r03).  Here one block per CU walks its tiles with the operand stream running through the tile boundaries.
// Tile order: block b belongs to XCD b % 8 and takes corpus tiles = b % 8 (mod 8); inside an XCD the sequence is query-tile fastest, so
// the (up to four) blocks that read one corpus tile are neighbours in time on ONE L2.
template <bool I8, bool AUX16 = false, bool R1 = false>
struct SearchTilePolicy {
    static constexpr bool REBASE_W = true;
    static constexpr bool PERMUTE_B = false;                     // a group's arg-max row is a position inside the tile: corpus rows stay in order
    int tiles_q, tiles_n, nq, D;
    int64_t n_rows, ldg;
    float* gmax;
    uint32_t* aux;
    const float2* qmeta; const float2* cmeta; const float* qoff; const float* qg; const float* ctrow;
    __device__ __forceinline__ bool tile(int o, int& m0, int& n0, int& ko) const {
        const int x = o & 7, L = o >> 3;
        const int tq = L % tiles_q, tn = (L / tiles_q) * 8 + x;
        m0 = tq * 256; n0 = tn * 256; ko = 0;
        return tn < tiles_n;
    }
    __device__ __forceinline__ void stage_issue(int m0, int n0, char* stage, int wid, int lane) const {
        if constexpr (I8) stage_i8_meta<256, R1>(cmeta, ctrow, qmeta, qoff, qg, n0, n_rows, m0, nq, reinterpret_cast<float*>(stage), wid * 64 + lane);
    }
    __device__ __forceinline__ void epilogue(const f32x4 (&acc)[4][8], int m0, int n0, int wr, int wc, int lane, const char* stage) const {
        if ((int64_t)n0 + wc * GROUP_ROWS >= n_rows) return;             // the wave's group lies past the shard (wave-uniform)
        const int64_t g = ((int64_t)n0 >> 6) + wc;
        if constexpr (I8) {
            const float* meta = reinterpret_cast<const float*>(stage);
            const float* qmeta_l = meta + I8_META_Q_OFF + (wr * 128 + (lane & 15)) * 2;
            groupmax_epilogue_i8<8, 4, R1>(acc, MetaFromLds{meta, wc, (lane >> 4) * 4},
                                           [&](int i) { return *reinterpret_cast<const float2*>(qmeta_l + i * 32); },
                                           [&](int i) { return meta[I8_META_Q_OFF + 2 * 256 + wr * 128 + i * 16 + (lane & 15)]; },
                                           TFromLds{meta, wc, (lane >> 4) * 4},
                                           [&](int i) { return meta[I8_META_Q_OFF + 3 * 256 + wr * 128 + i * 16 + (lane & 15)]; },
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

template <typename T, bool I8, bool AUX16, bool R1>
__global__ __launch_bounds__(512) void search_groupmax_persistent_kernel(const T* __restrict__ Q, int nq, const T* __restrict__ C, int64_t n_rows,
                                                                          int Kt /* row length in T elements */, int D, int tiles_q, int tiles_n,
                                                                          const float2* __restrict__ qmeta, const float2* __restrict__ cmeta,
                                                                          const float* __restrict__ qoff, const float* __restrict__ qg,
                                                                          const float* __restrict__ ctrow,
                                                                          float* __restrict__ gmax, uint32_t* __restrict__ aux, int64_t ldg,
                                                                          unsigned long long* __restrict__ zero_stats) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (zero_stats && blockIdx.x == 0 && threadIdx.x < 2) zero_stats[threadIdx.x] = 0ull;
    const SearchTilePolicy<I8, AUX16, R1> pol{tiles_q, tiles_n, nq, D, n_rows, ldg, gmax, aux, qmeta, cmeta, qoff, qg, ctrow};
    gemm8_persistent_body<T>(Q, Kt, C, Kt, nq, (int)n_rows, Kt, pol, smem);
}
