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

#include "search_pass_a.h"      // the scan: pass-A kernels and their epilogues, int8 quantisation
#include "search_tail.h"        // selection, exact rescoring + certificate, single-kernel tails, int8 candidate pipeline, merge

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
struct TopkWs { int64_t stats, gmax, aux, part_s, part_g, q8, qmeta, qoff, qg, thr, selg, counters, nsurv, redo, pairs, rpairs, surv_s, surv_i, total;
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
    w.qoff = take8(w.ldg * 4);                                    // q . mu per query (the centred int8 index)
    w.qg = take8(w.ldg * 4);                                      // gamma / s_q per query (ARX_TOPK_I8_CENTRE_QUERY)
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

template <int BM, bool GLDS, bool R1>
static int launch_groupmax_i8(const int8_t* Q8, const float2* qmeta, const float* qoff, const float* qg, int nq, const int8_t* C8, const float2* cmeta,
                              const float* ctrow, int64_t n_rows, int D,
                              float* gmax, uint32_t* aux, int64_t ldg, hipStream_t st) {
    using ML = GemmMainloop<i8pair_t, BM, 256, 2, 4, GLDS, GLDS ? 3 : 0>;
    auto kern = search_groupmax_i8_kernel<BM, GLDS, R1>;
    constexpr int smem_bytes = ((BM == 256 && GLDS) ? Gemm8Phase<i8pair_t, 2>::STAGE_OFF : ML::SMEM_BYTES) + (BM >= 128 ? 7168 : 0);       // stage_i8_meta's [768 + 4 BM] floats behind the k-tile buffers
    ARX_HIP_CHECK(arx_func_smem((const void*)kern, smem_bytes));
    const int tq = cdiv(nq, BM);
    const int64_t tn = (n_rows + 255) / 256;
    ARX_REQUIRE(tq * tn < (1ll << 31), "grid too large");
    kern<<<(int)(tq * tn), 512, smem_bytes, st>>>(Q8, qmeta, qoff, qg, nq, C8, cmeta, ctrow, n_rows, D, tq, (int)tn, gmax, aux, ldg);
    ARX_HIP_CHECK(hipGetLastError());
    return ARX_OK;
}

// pass A in persistent form: >= 256 queries per pass, an even number of 64-element k-tiles, rows addressable as int
template <typename T, bool I8, bool AUX16 = false, bool R1 = false>
static int launch_groupmax_persistent(const T* Q, int nq, const T* C, int64_t n_rows, int Kt, int D, const float2* qmeta, const float2* cmeta, const float* qoff,
                                      const float* qg, const float* ctrow,
                                      float* gmax, uint32_t* aux, int64_t ldg, int cu_limit, hipStream_t st,
                                      unsigned long long* zero_stats = nullptr) {
    auto kern = search_groupmax_persistent_kernel<T, I8, AUX16, R1>;
    constexpr int smem_bytes = Gemm8Phase<T, 0>::SMEM_BYTES;
    ARX_HIP_CHECK(arx_func_smem((const void*)kern, smem_bytes));
    const int tq = cdiv(nq, 256);
    const int64_t tn = (n_rows + 255) / 256;
    int n_cu = arx_device_cus();
    if (cu_limit > 0 && cu_limit < n_cu) n_cu = cu_limit;          // a CU-masked stream: one block per CU it may use
    int64_t grid = tq * tn < n_cu ? tq * tn : n_cu;
    grid = grid / 8 * 8 > 0 ? grid / 8 * 8 : 8;                   // a multiple of 8: a block keeps its XCD (and its residue class of corpus tiles)
    kern<<<(int)grid, 512, smem_bytes, st>>>(Q, nq, C, n_rows, Kt, D, tq, (int)tn, qmeta, cmeta, qoff, qg, ctrow, gmax, aux, ldg, zero_stats);
    ARX_HIP_CHECK(hipGetLastError());
    return ARX_OK;
}
static bool persistent_pass_ok(int nq, int64_t n_rows, int k_elems, int flags) {
    return !(flags & ARX_TOPK_NO_PERSISTENT) && nq > 128 && n_rows < (1ll << 31) - 256 && k_elems % 128 == 0 &&
           (int64_t)k_elems * 2 * 256 < (1ll << 31);
}

// int8 index: [n_rows x dim int8] [n_rows x (scale, L1)] [n_rows floats: t_c = m^ . (c - mu)] [dim floats: mu, the vector subtracted from every
// row before quantising it] [dim floats: m^ = mu / |mu|] [I8Header: |mu|, max |t_c|, max sum |c'_i m^_i|]
static int64_t i8_meta_offset(int64_t n_rows, int dim) { return round_up64(n_rows * (int64_t)dim, 256); }
static int64_t i8_t_offset(int64_t n_rows, int dim) { return round_up64(i8_meta_offset(n_rows, dim) + n_rows * 8, 256); }
static int64_t i8_mu_offset(int64_t n_rows, int dim) { return round_up64(i8_t_offset(n_rows, dim) + n_rows * 4, 256); }
static int64_t i8_mhat_offset(int64_t n_rows, int dim) { return i8_mu_offset(n_rows, dim) + (int64_t)dim * 4; }
static int64_t i8_hdr_offset(int64_t n_rows, int dim) { return i8_mhat_offset(n_rows, dim) + (int64_t)dim * 4; }

extern "C" int64_t arx_topk_i8_index_bytes(int64_t n_rows, int32_t dim) {
    if (n_rows <= 0 || dim <= 0 || dim % 128 != 0 || dim > 1024) return -1;
    return i8_hdr_offset(n_rows, dim) + 256;
}

extern "C" int32_t arx_topk_build_i8(const void* corpus, int64_t n_rows, int32_t dim, void* index_i8, void* stream) {
    ARX_REQUIRE(corpus && index_i8 && n_rows > 0, "bad args");
    ARX_REQUIRE(dim % 128 == 0 && dim <= 1024, "int8 pre-filter: dim=%d must be a multiple of 128, <= 1024", dim);
    const int64_t blocks = (n_rows + 3) / 4;
    ARX_REQUIRE(blocks < (1ll << 31), "too many rows for one launch");
    float* mu = (float*)((char*)index_i8 + i8_mu_offset(n_rows, dim));
    float* mhat = (float*)((char*)index_i8 + i8_mhat_offset(n_rows, dim));
    I8Header* hdr = (I8Header*)((char*)index_i8 + i8_hdr_offset(n_rows, dim));
    rows_mean_kernel<<<cdiv(dim, 64), 256, 0, (hipStream_t)stream>>>((const f16_t*)corpus, n_rows, dim, mu);
    ARX_HIP_CHECK(hipGetLastError());
    mean_finish_kernel<<<1, 256, 0, (hipStream_t)stream>>>(mu, dim, mhat, hdr);
    ARX_HIP_CHECK(hipGetLastError());
    quantize_rows_i8_kernel<<<(int)blocks, 256, 0, (hipStream_t)stream>>>((const f16_t*)corpus, n_rows, dim, (int8_t*)index_i8,
                                                                         (float2*)((char*)index_i8 + i8_meta_offset(n_rows, dim)), nullptr, mu, nullptr,
                                                                         nullptr, mhat, (float*)((char*)index_i8 + i8_t_offset(n_rows, dim)), hdr, nullptr,
                                                                         nullptr, 0);
    ARX_HIP_CHECK(hipGetLastError());
    return ARX_OK;
}

extern "C" int32_t arx_topk_i8_index_info(const void* index_i8, int64_t n_rows, int32_t dim, float* host_out, void* stream) {
    ARX_REQUIRE(index_i8 && host_out && n_rows > 0 && dim > 0 && dim % 128 == 0 && dim <= 1024, "bad args");
    ARX_HIP_CHECK(hipMemcpyAsync(host_out, (const char*)index_i8 + i8_hdr_offset(n_rows, dim), 3 * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream));
    ARX_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
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
            float* qoff = (float*)((char*)ws + L.qoff);
            float* qg = (float*)((char*)ws + L.qg);
            const bool r1 = (P.flags & ARX_TOPK_I8_CENTRE_QUERY) != 0;          // the query is centred too; pass A adds the rank-one term (search_pass_a.h)
            const float* ctrow = (const float*)((const char*)index_i8 + i8_t_offset(n_rows, dim));
            quantize_rows_i8_kernel<<<cdiv(nq, 4), 256, 0, st>>>(Q, nq, dim, q8, qmeta, q0 == 0 ? (unsigned long long*)((char*)ws + L.stats) : nullptr, nullptr,
                                                                 (const float*)((const char*)index_i8 + i8_mu_offset(n_rows, dim)), qoff,
                                                                 (const float*)((const char*)index_i8 + i8_mhat_offset(n_rows, dim)), nullptr, nullptr,
                                                                 (const I8Header*)((const char*)index_i8 + i8_hdr_offset(n_rows, dim)), qg, r1 ? 1 : 0);
            ARX_HIP_CHECK(hipGetLastError());
            const int8_t* C8 = (const int8_t*)index_i8;
            const float2* cmeta = (const float2*)((const char*)index_i8 + i8_meta_offset(n_rows, dim));
            ProfScope ps(ARX_K_SEARCH_GROUPMAX, st);
            uint32_t* aux = (uint32_t*)((char*)ws + L.aux);
            if (persistent_pass_ok(nq, n_rows, dim / 2, P.flags))
                rc = r1 ? launch_groupmax_persistent<i8pair_t, true, false, true>((const i8pair_t*)q8, nq, (const i8pair_t*)C8, n_rows, dim / 2, dim, qmeta, cmeta,
                                                                                   qoff, qg, ctrow, gmax, aux, L.ldg, P.cu_limit, st)
                        : launch_groupmax_persistent<i8pair_t, true>((const i8pair_t*)q8, nq, (const i8pair_t*)C8, n_rows, dim / 2, dim, qmeta, cmeta, qoff,
                                                                     nullptr, nullptr, gmax, aux, L.ldg, P.cu_limit, st);
            else
            rc = r1 ? (nq <= 64 ? launch_groupmax_i8<64, true, true>(q8, qmeta, qoff, qg, nq, C8, cmeta, ctrow, n_rows, dim, gmax, aux, L.ldg, st)
                       : nq <= 128 ? launch_groupmax_i8<128, true, true>(q8, qmeta, qoff, qg, nq, C8, cmeta, ctrow, n_rows, dim, gmax, aux, L.ldg, st)
                                   : launch_groupmax_i8<256, true, true>(q8, qmeta, qoff, qg, nq, C8, cmeta, ctrow, n_rows, dim, gmax, aux, L.ldg, st))
                    : (nq <= 64 ? launch_groupmax_i8<64, true, false>(q8, qmeta, qoff, qg, nq, C8, cmeta, ctrow, n_rows, dim, gmax, aux, L.ldg, st)
                       : nq <= 128 ? launch_groupmax_i8<128, true, false>(q8, qmeta, qoff, qg, nq, C8, cmeta, ctrow, n_rows, dim, gmax, aux, L.ldg, st)
                                   : launch_groupmax_i8<256, true, false>(q8, qmeta, qoff, qg, nq, C8, cmeta, ctrow, n_rows, dim, gmax, aux, L.ldg, st));
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
                rc = single ? launch_groupmax_persistent<f16_t, false, true>(Q, nq, C, n_rows, dim, dim, nullptr, nullptr, nullptr, nullptr, nullptr, gmax, aux16, L.ldg, P.cu_limit, st, zs16)
                            : launch_groupmax_persistent<f16_t, false>(Q, nq, C, n_rows, dim, dim, nullptr, nullptr, nullptr, nullptr, nullptr, gmax, nullptr, L.ldg, P.cu_limit, st);
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
