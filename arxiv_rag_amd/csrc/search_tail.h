// What follows pass A in arx_topk_* (search.hip): group selection, exact fp32 rescoring with the exactness certificate, the single-kernel
// tails (fp16 aux words / int8 pipeline on small shards), the int8 candidate pipeline of large shards, the merge of per-shard partials.
// Included by search.hip after search_pass_a.h; not a stand-alone header.
#pragma once

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
