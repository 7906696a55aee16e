// Small-batch linear layer (the query-encode path: a handful of short texts, M <= ARX_SMALL_M packed token rows).
//
// The 256 x 256-tile kernels of gemm8.h are throughput kernels: with M = 12 rows a layer's four GEMMs occupy 3 to 12 CUs, each walking
// its whole K serially (17 us at K = 768, 70 us at K = 3072: 1.5 ms per 12-layer forward of ONE query, tools/query_latency.py).
// Here the problem is cut the other way: a block is ONE wave computing a 16-row x 64-column x (K / S)-deep partial product straight from
// global memory (every load of the block is in flight at once: one memory round trip, no LDS), S chosen so that a few hundred blocks
// cover the chip and each weight byte is read once; a second, row-wise kernel adds the S partial sums in a fixed order and applies the
// same epilogue arithmetic as epilogue_store_v2 (gemm.h): bias / LayerNorm fold / GELU / residual (+ its LayerNorm), and the output row's
// LayerNorm statistics in final form (mean, rstd).
// Opt-in (arx_encoder_set_low_latency): the summation order differs from the tile kernels', so rows agree with the main path to
// rounding, not bit for bit — corpus rows keep the batch-independent main path, only query batches come here.
#pragma once
#include "gemm.h"

#define ARX_SMALL_M 256            // packed token rows (16 m-tiles at most)
#define ARX_MEDIUM_M 8192          // up to here the low-latency schedule uses 128 x 128 tiles (encoder.hip, variant 71)
#define ARX_SMALL_WS_BYTES (64ll << 20)

// partial products: ws[(split * Mp + m) * N + n] fp32, Mp = M rounded up to 16
template <int KSTEPS_MAX>
__global__ __launch_bounds__(64) void gemm_small_partial_kernel(const uint16_t* __restrict__ A, int64_t lda, const uint16_t* __restrict__ W,
                                                                 int64_t ldw, int M, int N, int ksteps, float* __restrict__ ws) {
    const int lane = threadIdx.x;
    const int n0 = blockIdx.x * 64, m0 = blockIdx.y * 16, kz = blockIdx.z;
    const int Mp = gridDim.y * 16;
    const int r = lane & 15, kc = (lane >> 4) * 8;
    int am = m0 + r; am = am < M ? am : M - 1;
    const uint16_t* ap = A + (int64_t)am * lda + (int64_t)kz * ksteps * 32 + kc;
    const uint16_t* wp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int wn = n0 + j * 16 + r; wn = wn < N ? wn : N - 1;
        wp[j] = W + (int64_t)wn * ldw + (int64_t)kz * ksteps * 32 + kc;
    }
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // KSTEPS_MAX k-steps of 32 per pass, all their loads issued before the first MFMA
    for (int s0 = 0; s0 < ksteps; s0 += KSTEPS_MAX) {
        bf16x8 af[KSTEPS_MAX], wf[KSTEPS_MAX][4];
#pragma unroll
        for (int s = 0; s < KSTEPS_MAX; ++s) {
            if (s0 + s < ksteps) {
                af[s] = *reinterpret_cast<const bf16x8*>(ap + (s0 + s) * 32);
#pragma unroll
                for (int j = 0; j < 4; ++j) wf[s][j] = *reinterpret_cast<const bf16x8*>(wp[j] + (s0 + s) * 32);
            }
        }
#pragma unroll
        for (int s = 0; s < KSTEPS_MAX; ++s) {
            if (s0 + s < ksteps) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = Mfma<bf16_t>::mma(wf[s][j], af[s], acc[j]);
            }
        }
    }
    // acc[j][e] = C[m0 + r][n0 + 16 j + 4 (lane >> 4) + e]
    float* dst = ws + ((int64_t)kz * Mp + m0 + r) * N + n0 + (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (n0 + j * 16 < N) *reinterpret_cast<f32x4*>(dst + j * 16) = acc[j];
}

// one block per output row: sum the S partials (s = 0, 1, ... in order; eight loads in flight at a time — a dependent chain of S loads
// cost 12-20 us per call), epilogue, bf16 store.  Statistics modes: the block holds the whole row, so it writes the row's mean and
// rstd itself (p.fin_mean / p.fin_rstd: what ln_finalize_kernel would derive from the 64-column slabs of the tile kernels; one launch
// per LayerNorm less), from the bf16-ROUNDED values as everywhere else.
template <int MODE>
__global__ __launch_bounds__(256) void gemm_small_epilogue_kernel(const float* __restrict__ ws, int S, int Mp, int M, int N, EpiParams p) {
    constexpr bool LN_IN = (MODE == EPI_LN_BIAS || MODE == EPI_LN_BIAS_GELU);
    constexpr bool STATS = (MODE == EPI_RESID_STATS || MODE == EPI_LNRESID_STATS);
    constexpr bool RESID = (MODE == EPI_BIAS_RESID || STATS);
    __shared__ float red[2][4];
    const int m = blockIdx.x;
    float a_mean = 0.f, a_rstd = 1.f, r_mean = 0.f, r_rstd = 1.f;
    if constexpr (LN_IN) { a_mean = p.a_sum[m]; a_rstd = p.a_sq[m]; }
    if constexpr (MODE == EPI_LNRESID_STATS) { r_mean = p.r_sum[m]; r_rstd = p.r_sq[m]; }
    float st_a = 0.f, st_q = 0.f;
    // blockIdx.y: 1024-column slice of the row (modes without statistics are launched with one block per slice: the wide layers' rows —
    // N = 2304, 3072, 4096 — no longer walk their slices one after the other); statistics modes have N <= 1024 = one slice
    for (int n = blockIdx.y * 1024 + threadIdx.x * 4; n < N; n += 1024 * gridDim.y) {
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* src = ws + (int64_t)m * N + n;
        const int64_t sstride = (int64_t)Mp * N;
        int s = 0;
        for (; s + 8 <= S; s += 8) {
            f32x4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const f32x4*>(src + (s + u) * sstride);
#pragma unroll
            for (int u = 0; u < 8; ++u) v += t[u];
        }
        for (; s < S; ++s) v += *reinterpret_cast<const f32x4*>(src + s * sstride);
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n);
        if constexpr (LN_IN) {
            const f32x4 sv = *reinterpret_cast<const f32x4*>(p.s_vec + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = a_rstd * (-a_mean * sv[e] + v[e]) + b[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += b[e];
        }
        if constexpr (MODE == EPI_BIAS_GELU || MODE == EPI_LN_BIAS_GELU) {
            const f32x2 g0 = gelu_poly_pk(f32x2{v[0], v[1]}), g1 = gelu_poly_pk(f32x2{v[2], v[3]});
            v = f32x4{g0.x, g0.y, g1.x, g1.y};
        }
        if constexpr (RESID) {
            const u32x2 rr = *reinterpret_cast<const u32x2*>(p.resid + (int64_t)m * p.ldr + n);
            float ra[4];
            unpack_bf16x2(rr[0], ra[0], ra[1]); unpack_bf16x2(rr[1], ra[2], ra[3]);
            if constexpr (MODE == EPI_LNRESID_STATS) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(p.r_gamma + n), be = *reinterpret_cast<const f32x4*>(p.r_beta + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) ra[e] = fmaf((ra[e] - r_mean) * r_rstd, g[e], be[e]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += ra[e];
        }
        u32x2 o;
        o[0] = pack_bf16x2(v[0], v[1]); o[1] = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<u32x2*>(p.out + (int64_t)m * p.ldc + n) = o;
        if constexpr (STATS) {
            float x0, x1, x2, x3;
            unpack_bf16x2(o[0], x0, x1); unpack_bf16x2(o[1], x2, x3);
            st_a += (x0 + x1) + (x2 + x3);
            st_q += fmaf(x0, x0, fmaf(x1, x1, fmaf(x2, x2, x3 * x3)));
        }
    }
    if constexpr (STATS) {                                         // fixed reduction tree: lanes, then the four waves in order
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { st_a += __shfl_xor(st_a, d); st_q += __shfl_xor(st_q, d); }
        if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = st_a; red[1][threadIdx.x >> 6] = st_q; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const float sa = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3], sq = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
            const float mu = sa * p.inv_h;                         // ln_finalize_kernel's formulas
            p.fin_mean[m] = mu;
            p.fin_rstd[m] = rsqrtf(fmaxf(sq * p.inv_h - mu * mu, 0.f) + p.eps);
        }
    }
}

// number of K splits: a divisor of K / 32 that brings the grid to about `target` blocks
static inline int gemm_small_splits(int mt, int nt, int ksteps_total, int target) {
    int best = 1;
    for (int s = 1; s <= ksteps_total; ++s) {
        if (ksteps_total % s) continue;
        if ((int64_t)mt * nt * s <= target) best = s;
    }
    return best;
}

template <int MODE>
static int launch_gemm_small(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, int M, int N, int K, const EpiParams& ep,
                             float* ws, hipStream_t st) {
    if (K % 32 != 0 || N % 64 != 0 || M > ARX_SMALL_M || !ws) {
        arx_set_error("small-batch gemm: M=%d (<= %d) N=%d (%% 64) K=%d (%% 32), workspace %p", M, ARX_SMALL_M, N, K, (void*)ws);
        return ARX_ERR_ARG;
    }
    if ((MODE == EPI_RESID_STATS || MODE == EPI_LNRESID_STATS) && !(ep.fin_mean && ep.fin_rstd)) {
        arx_set_error("small-batch gemm: statistics modes write the row's mean / rstd (EpiParams::fin_mean / fin_rstd)");
        return ARX_ERR_ARG;
    }
    const int mt = cdiv(M, 16), nt = N / 64, kst = K / 32;
    int S = gemm_small_splits(mt, nt, kst, 768);
    while ((int64_t)S * mt * 16 * N * 4 > ARX_SMALL_WS_BYTES && S > 1) {       // keep the partials inside the workspace
        int s2 = S - 1;
        while (s2 > 1 && kst % s2) --s2;
        S = s2;
    }
    const int ksteps = kst / S;
    dim3 grid(nt, mt, S);
    if (ksteps <= 4) gemm_small_partial_kernel<4><<<grid, 64, 0, st>>>(A, lda, W, ldw, M, N, ksteps, ws);
    else gemm_small_partial_kernel<8><<<grid, 64, 0, st>>>(A, lda, W, ldw, M, N, ksteps, ws);
    ARX_HIP_CHECK(hipGetLastError());
    constexpr bool STATS = (MODE == EPI_RESID_STATS || MODE == EPI_LNRESID_STATS);
    if (STATS && N > 1024) {
        arx_set_error("small-batch gemm: statistics modes take N <= 1024 (the row is reduced inside one block), N=%d", N);
        return ARX_ERR_ARG;
    }
    gemm_small_epilogue_kernel<MODE><<<dim3(M, STATS ? 1 : cdiv(N, 1024)), 256, 0, st>>>(ws, S, mt * 16, M, N, ep);
    ARX_HIP_CHECK(hipGetLastError());
    return ARX_OK;
}
