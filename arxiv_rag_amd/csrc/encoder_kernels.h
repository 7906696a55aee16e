// Non-GEMM encoder kernels: position ids + packing offsets, embedding gather + LayerNorm, row LayerNorm,
// fused attention (S <= 512, whole K/V of one (sequence, head) resident in LDS), masked mean / CLS pool
// + L2 normalise.  Activations are token-PACKED: row t of every [T, *] buffer is token (t - cu[b]) of
// sequence b, no padding rows between sequences.
#pragma once
#include <type_traits>

#include "arx_common.h"

// ---------------------------------------------------------------------------------------------------
// cu[b] = sum_{i<b} lens[i]  (one block; n_seqs <= a few thousand)
__global__ void scan_lens_kernel(const int32_t* __restrict__ lens, int32_t* __restrict__ cu, int n) {
    __shared__ int32_t part[1024];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int per = (n + nt - 1) / nt;
    const int b0 = tid * per, b1 = min(n, b0 + per);
    int s = 0;
    for (int b = b0; b < b1; ++b) s += lens[b];
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < nt; ++i) { int v = part[i]; part[i] = run; run += v; }
        cu[n] = run;
    }
    __syncthreads();
    int run = part[tid];
    for (int b = b0; b < b1; ++b) { cu[b] = run; run += lens[b]; }
}

// ---------------------------------------------------------------------------------------------------
// Embedding gather + LayerNorm -> bf16 x[T,H].  One wave per token; grid (ceil(max_len/4), n_seqs), 256 thr.
// MPNet position id = cumsum(ids != pad)[s] * (ids[s] != pad) + pad  (TF modeling_mpnet.py:873-881);
// BERT position id = s, plus token-type row 0.
template <int ARCH>
__global__ __launch_bounds__(256) void embed_ln_kernel(const int32_t* __restrict__ ids, int seq_stride,
                                                        const int32_t* __restrict__ lens, const int32_t* __restrict__ cu,
                                                        const float* __restrict__ word, const float* __restrict__ pos,
                                                        const float* __restrict__ type0, const float* __restrict__ g,
                                                        const float* __restrict__ bta, uint16_t* __restrict__ x,
                                                        int H, int vocab, int max_pos, int pad_id, float eps) {
    const int b = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int s = blockIdx.x * 4 + w;
    const int L = lens[b];
    if (s >= L) return;
    const int32_t* row = ids + (int64_t)b * seq_stride;
    int id = row[s];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    int pid;
    if (ARCH == ARX_ARCH_MPNET) {
        int cnt = 0;                                   // non-pad tokens in [0, s]
        for (int c0 = 0; c0 <= s; c0 += 64) {
            const int i = c0 + lane;
            const bool np = (i <= s) && (row[i] != pad_id);
            cnt += __popcll(__ballot(np));
        }
        pid = (row[s] != pad_id) ? cnt + pad_id : pad_id;
    } else {
        pid = s;
    }
    pid = pid >= max_pos ? max_pos - 1 : pid;
    const float* wr = word + (int64_t)id * H;
    const float* pr = pos + (int64_t)pid * H;
    // H <= 1024: each lane holds up to 4 float4 chunks (chunk c covers columns 4*(lane + 64*c) ..)
    f32x4 v[4];
    float sum = 0.f;
    const int nch = H >> 2;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) {
            f32x4 a = *reinterpret_cast<const f32x4*>(wr + ch * 4);
            const f32x4 p4 = *reinterpret_cast<const f32x4*>(pr + ch * 4);
            a += p4;
            if (ARCH == ARX_ARCH_BERT) a += *reinterpret_cast<const f32x4*>(type0 + ch * 4);
            v[c] = a;
            sum += a[0] + a[1] + a[2] + a[3];
        } else {
            v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const float mean = wave_sum(sum) / (float)H;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) {
            const f32x4 d = v[c] - mean;
            sq += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)H + eps);
    uint16_t* out = x + (int64_t)(cu[b] + s) * H;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) {
            const f32x4 gg = *reinterpret_cast<const f32x4*>(g + ch * 4);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(bta + ch * 4);
            const f32x4 y = (v[c] - mean) * rstd * gg + bb;
            u32x2 o;
            o[0] = pack_bf16x2(y[0], y[1]); o[1] = pack_bf16x2(y[2], y[3]);
            *reinterpret_cast<u32x2*>(out + ch * 4) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Row LayerNorm bf16 -> bf16, one wave per row (H <= 1024, H % 8 == 0).  n_rows read from cu[n_seqs].
__global__ __launch_bounds__(256) void layernorm_kernel(const uint16_t* __restrict__ y, uint16_t* __restrict__ x,
                                                         const float* __restrict__ g, const float* __restrict__ bta,
                                                         const int32_t* __restrict__ n_rows_ptr, int H, float eps) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= *n_rows_ptr) return;
    const uint16_t* in = y + (int64_t)r * H;
    const int nch = H >> 3;                     // 16-B chunks of 8 bf16
    float v[2][8];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) {
            const u32x4 q = *reinterpret_cast<const u32x4*>(in + ch * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unpack_bf16x2(q[e], v[c][2 * e], v[c][2 * e + 1]);
                sum += v[c][2 * e] + v[c][2 * e + 1];
            }
        }
    }
    const float mean = wave_sum(sum) / (float)H;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c)
        if (lane + 64 * c < nch) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mean; sq += d * d; }
        }
    const float rstd = rsqrtf(wave_sum(sq) / (float)H + eps);
    uint16_t* out = x + (int64_t)r * H;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(g + ch * 8), g1 = *reinterpret_cast<const f32x4*>(g + ch * 8 + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(bta + ch * 8), b1 = *reinterpret_cast<const f32x4*>(bta + ch * 8 + 4);
            float o[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = (v[c][e] - mean) * rstd * g0[e] + b0[e];
                o[e + 4] = (v[c][e + 4] - mean) * rstd * g1[e] + b1[e];
            }
            u32x4 q;
            q[0] = pack_bf16x2(o[0], o[1]); q[1] = pack_bf16x2(o[2], o[3]);
            q[2] = pack_bf16x2(o[4], o[5]); q[3] = pack_bf16x2(o[6], o[7]);
            *reinterpret_cast<u32x4*>(out + ch * 8) = q;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Fused attention for one (sequence b, head h, block of 32*NW queries).
//   scores^T tile [32 keys x 32 queries] = K_tile . Q^T  via mfma_f32_32x32x16_bf16 (A = K rows from LDS,
//   B = Q fragments held in registers): the query sits on the lane, the keys in the 16 accumulator
//   registers, so softmax statistics are lane-local (+ one exchange between lane halves) and the
//   bf16-packed probabilities are directly the B operand of  O^T += V^T . P^T  (no LDS round trip).
//   V is transposed once per block while staging into LDS (key order inside each 16-key group permuted
//   to the MFMA k order: [0-3, 8-11 | 4-7, 12-15]) so the A-operand read is one ds_read_b128.
//   s = q.k * scale + bias[key - query] (MPNet), keys >= len masked; online softmax in base 2.
// LDS (dynamic): K [Lk][DH] bf16 (16-B chunk XOR-swizzled), V^T [DH][Lk+8] bf16,
//                4 shifted copies of the Toeplitz bias row [2*Lk+8] f32 (aligned float4 reads).
template <int DH> struct AttnSmem {
    static __host__ __device__ int k_bytes(int Lk) { return Lk * DH * 2; }
    static __host__ __device__ int vt_stride(int Lk) { return Lk + 8; }
    static __host__ __device__ int vt_bytes(int Lk) { return DH * (Lk + 8) * 2; }
    static __host__ __device__ int bias_stride(int Lk) { return 2 * Lk + 8; }
    static __host__ __device__ int bias_bytes(int Lk, bool has) { return has ? 4 * (2 * Lk + 8) * 4 : 0; }
    static __host__ __device__ int total(int Lk, bool has) { return k_bytes(Lk) + vt_bytes(Lk) + bias_bytes(Lk, has); }
};

#define ARX_BIAS_CENTER 640          // global bias table: [heads][2*640+1], entry d + 640 for d in [-640, 640]
#define ARX_BIAS_ROW (2 * ARX_BIAS_CENTER + 1)

template <int DH, bool HAS_BIAS, int NW>
__global__ __launch_bounds__(NW * 64) void attention_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ ctx,
                                                             const int32_t* __restrict__ cu,
                                                             const float* __restrict__ bias_tbl,   // pre-multiplied by log2(e)
                                                             int H, float scale_log2e) {
    constexpr int NT = NW * 64;
    constexpr int CPR = DH / 8;                 // 16-B chunks per K row
    constexpr int RPB = 256 / (DH * 2);         // K rows per 256-B bank row
    constexpr int KS = DH / 16;                 // QK^T k-steps
    constexpr int DB = DH / 32;                 // 32-row blocks of O^T
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int b = blockIdx.z, h = blockIdx.y;
    const int t0 = cu[b];
    const int L = cu[b + 1] - t0;
    const int q0 = blockIdx.x * (32 * NW);
    if (q0 >= L) return;
    const int Lk = (L + 31) & ~31;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int64_t ld = 3 * (int64_t)H;
    const uint16_t* Qg = qkv + (int64_t)t0 * ld + h * DH;
    const uint16_t* Kg = Qg + H;
    const uint16_t* Vg = Qg + 2 * H;

    char* Ks = smem;
    uint16_t* Vt = reinterpret_cast<uint16_t*>(smem + AttnSmem<DH>::k_bytes(Lk));
    float* Bs = reinterpret_cast<float*>(smem + AttnSmem<DH>::k_bytes(Lk) + AttnSmem<DH>::vt_bytes(Lk));
    const int vts = AttnSmem<DH>::vt_stride(Lk);
    const int bst = AttnSmem<DH>::bias_stride(Lk);

    // ---- stage K (swizzled), V^T (transposed, key-permuted), bias copies
    for (int cid = tid; cid < Lk * CPR; cid += NT) {
        const int row = cid / CPR, pc = cid % CPR;
        const int c = pc ^ ((row / RPB) & (CPR - 1));
        u32x4 kv = u32x4{0u, 0u, 0u, 0u}, vv = u32x4{0u, 0u, 0u, 0u};
        if (row < L) {
            kv = *reinterpret_cast<const u32x4*>(Kg + (int64_t)row * ld + c * 8);
            vv = *reinterpret_cast<const u32x4*>(Vg + (int64_t)row * ld + pc * 8);
        }
        *reinterpret_cast<u32x4*>(Ks + (int64_t)cid * 16) = kv;
        // V[row][pc*8 + e] -> Vt[pc*8 + e][pos(row)]
        const int k16 = row & 15;
        const int pos = (row & ~15) + (((k16 >> 2) & 1) << 3) + ((k16 >> 3) << 2) + (k16 & 3);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            Vt[(pc * 8 + 2 * e) * vts + pos] = (uint16_t)(vv[e] & 0xffffu);
            Vt[(pc * 8 + 2 * e + 1) * vts + pos] = (uint16_t)(vv[e] >> 16);
        }
    }
    if (HAS_BIAS) {
        const float* bt = bias_tbl + (int64_t)h * ARX_BIAS_ROW + ARX_BIAS_CENTER;
        for (int i = tid; i < 4 * bst; i += NT) {
            const int c = i / bst, j = i % bst;
            int d = j + c - Lk;                       // key - query
            d = d < -ARX_BIAS_CENTER ? -ARX_BIAS_CENTER : (d > ARX_BIAS_CENTER ? ARX_BIAS_CENTER : d);
            Bs[i] = bt[d];
        }
    }
    __syncthreads();

    // ---- per-wave: 32 queries
    const int qw = q0 + wid * 32;
    if (qw >= L) return;                              // no barriers below
    const int ql = lane & 31, hh = lane >> 5;
    const int q = qw + ql;                            // this lane's query (may be >= L: computed, not stored)
    const int qc = q < L ? q : L - 1;
    bf16x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
        qf[ks] = *reinterpret_cast<const bf16x8*>(Qg + (int64_t)qc * ld + ks * 16 + hh * 8);

    f32x16 o[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // bias: copy c = (-q) & 3, aligned float4 index (kb - q + Lk - c) for key base kb (multiple of 4)
    const int bc = (4 - (qc & 3)) & 3;
    const float* brow = Bs + bc * bst + (Lk - qc - bc);
    const int krow_sw = ((ql / RPB) & (CPR - 1));     // swizzle term of K row (32*kt + ql): 32*kt/RPB = 0 mod CPR

    const int nkt = Lk >> 5;
    for (int kt = 0; kt < nkt; ++kt) {
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const char* krow = Ks + (int64_t)(kt * 32 + ql) * (DH * 2);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(krow + (((2 * ks + hh) ^ krow_sw) << 4));
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
        }
        // scale + bias + mask; register r <-> key = 32*kt + (r&3) + 8*(r>>2) + 4*hh
        float mx = -INFINITY;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int kb = kt * 32 + 8 * g4 + 4 * hh;
            f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
            if (HAS_BIAS) bv = *reinterpret_cast<const f32x4*>(brow + kb);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = s[g4 * 4 + e] * scale_log2e + bv[e];
                v = (kb + e < L) ? v : -INFINITY;
                s[g4 * 4 + e] = v;
                mx = fmaxf(mx, v);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);           // finite: key 0 of tile 0 is always valid
        const float alpha = exp2f(m_run - m_new);
        m_run = m_new;
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = exp2f(s[r] - m_new); rs += s[r]; }
        l_run = l_run * alpha + rs;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
        // P^T fragments: k-step ss uses registers 8ss..8ss+7
        bf16x8 pf[2];
#pragma unroll
        for (int ss = 0; ss < 2; ++ss)
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[ss][e] = (bf16_t)s[8 * ss + e];
#pragma unroll
        for (int d = 0; d < DB; ++d) {
            const uint16_t* vrow = Vt + (int64_t)(d * 32 + ql) * vts + kt * 32 + hh * 8;
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                const bf16x8 vf = *reinterpret_cast<const bf16x8*>(vrow + ss * 16);
                o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[ss], o[d], 0, 0, 0);
            }
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    if (q < L) {
        uint16_t* orow = ctx + (int64_t)(t0 + q) * H + h * DH;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                u32x2 w2;
                w2[0] = pack_bf16x2(o[d][g4 * 4 + 0] * inv, o[d][g4 * 4 + 1] * inv);
                w2[1] = pack_bf16x2(o[d][g4 * 4 + 2] * inv, o[d][g4 * 4 + 3] * inv);
                *reinterpret_cast<u32x2*>(orow + d * 32 + 8 * g4 + 4 * hh) = w2;
            }
    }
}

// ---------------------------------------------------------------------------------------------------
// attention v2: same algorithm, but
//   * V is staged ROW-major (16-B ds_write_b128, chunk XOR 4*((row>>1)&1) for DH = 64) and the V^T MFMA operand
//     is produced by the hardware transposing read ds_read_b64_tr_b16 (two per k-step: keys 16s+4hh.. and
//     16s+8+4hh.. at this lane's dh column) — no b16 scatter while staging;
//   * NW = 8 waves x 32 queries: one block covers a whole 256-token sequence-head, K/V staged once;
//   * the key mask is applied only in the (uniform) ragged last tile; O is rescaled only when some lane's
//     running max moved;
//   * (round 2) the tile loop that normally runs is the OPTIMISTIC one — fixed softmax reference 0, hand-pipelined LDS reads —
//     and the running-maximum loop is its range-checked fallback (see the kernel body).
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <int DH> struct AttnSmem2 {
    static __host__ __device__ int kv_bytes(int Lk) { return Lk * DH * 2; }
    static __host__ __device__ int bias_stride(int Lk) { return 2 * Lk + 8; }
    static __host__ __device__ int total(int Lk, bool has) { return 2 * kv_bytes(Lk) + (has ? 4 * (2 * Lk + 8) * 4 : 0); }
};

// bias copies: copy c lives at c*bias_stride + BIAS_OFF[c] floats; the offsets make the 16-lane ds_read_b128 groups
// (lanes with q = 0-3,12-15,20-27 / 4-11,16-19,28-31) land on 16 distinct 16-B slots (brute-forced; 0 conflicts)
template <int DH> struct AttnSmem3 {
    static __host__ __device__ int kv_bytes(int Lk) { return Lk * DH * 2; }
    static __host__ __device__ int bias_stride(int Lk) { return ((2 * Lk + 8 + 63) & ~63) + 64; }
    static __host__ __device__ int total(int Lk, bool has) { return 2 * kv_bytes(Lk) + (has ? 4 * bias_stride(Lk) * 4 : 0); }
};
__device__ __forceinline__ int attn_bias_off(int c) { return c == 0 ? 0 : (c == 1 ? 20 : (c == 2 ? 36 : 52)); }

template <int DH, bool HAS_BIAS, int NW>
__global__ __launch_bounds__(NW * 64, NW / 2) void attention_tr_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ ctx,
                                                                        const int32_t* __restrict__ cu,
                                                                        const float* __restrict__ bias_tbl, int H,
                                                                        float scale_log2e, int dev_word, unsigned long long* dev_stamps) {
    // dev_word / dev_stamps: 0 / nullptr in the product path.  The dev build (tools/attn_probe.py) uses them for per-block time
    // stamps (start, barrier passed, slowest wave's end, hardware id) and for two probes: word 3 = staging only, 5 = compute only,
    // 6 = the exact (running-maximum) tile loop instead of the optimistic one.
    constexpr int NT = NW * 64;
    constexpr int CPR = DH / 8;
    constexpr int RPB = 256 / (DH * 2);
#ifdef ARX_DEV_VARIANTS
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (dev_stamps && threadIdx.x == 0) {
        dev_stamps[4 * lin] = wall_clock64();
        dev_stamps[4 * lin + 3] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
#endif
    constexpr int KS = DH / 16;
    constexpr int DB = DH / 32;
    constexpr int ROWB = DH * 2;                  // bytes per K / V row in LDS
    constexpr int RPI = NT / CPR;                 // rows staged per pass of the block
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.z, h = blockIdx.y;
    const int t0 = cu[b];
    const int L = cu[b + 1] - t0;
    const int q0 = blockIdx.x * (32 * NW);
    if (q0 >= L) return;
    const int Lk = (L + 31) & ~31;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t ld = 3u * (uint32_t)H;          // elements; in-sequence offsets stay 32-bit (rows < 512)
    const uint16_t* Qg = qkv + (int64_t)t0 * ld + h * DH;
    const uint16_t* Kg = Qg + H;
    const uint16_t* Vg = Qg + 2 * H;
    char* Ks = smem;
    char* Vs = smem + AttnSmem3<DH>::kv_bytes(Lk);
    float* Bs = reinterpret_cast<float*>(smem + 2 * AttnSmem3<DH>::kv_bytes(Lk));
    const int bst = AttnSmem3<DH>::bias_stride(Lk);

    // this lane's Q fragments: requested BEFORE the K/V stream so that their latency runs under it (they used to be loaded after the
    // barrier: one more exposed memory round trip per block)
    const int ql = lane & 31, hh = lane >> 5;
    const int qw = q0 + wid * 32;
    const int q = qw + ql;
    const int qc = q < L ? q : L - 1;
    bf16x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
        qf[ks] = *reinterpret_cast<const bf16x8*>(Qg + (uint32_t)qc * ld + ks * 16 + hh * 8);
    {   // stage K (chunk ^ row swizzle) and V (chunk ^ 4*((row>>1)&1) for DH=64) with direct global->LDS loads: every
        // request of the block is in flight at once (one memory round trip instead of one per pass), no staging
        // registers.  This thread always owns physical chunk pc of rows r0, r0+RPI, ... (RPI % 16 == 0 keeps both
        // swizzle terms per-thread constants); the LDS image is lane-linear, the swizzle sits on the source address.
        // Rows >= L re-read row L-1: finite values that the key mask / zero probabilities neutralise.
        static_assert(RPI % 16 == 0, "staging pass must keep the swizzle terms constant per thread");
        typedef __attribute__((address_space(3))) void lds_v;
        typedef const __attribute__((address_space(1))) void gbl_v;
        const int pc = tid % CPR, r0 = tid / CPR;
        const int kc = pc ^ ((r0 / RPB) & (CPR - 1));
        const int vc = (DH == 64) ? (pc ^ (((r0 >> 1) & 1) << 2)) : pc;
        uint32_t lo = (uint32_t)(tid & ~63) * 16;
        int LkS = Lk;
#ifdef ARX_DEV_VARIANTS
        if (dev_word == 5) LkS = 0;                        // probe: compute only (LDS holds whatever it holds)
#endif
        for (int row = r0; row < LkS; row += RPI, lo += NT * 16) {
            const uint32_t gr = (uint32_t)(row < L ? row : L - 1) * ld;
            __builtin_amdgcn_global_load_lds((gbl_v*)(Kg + gr + kc * 8), (lds_v*)(Ks + lo), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_v*)(Vg + gr + vc * 8), (lds_v*)(Vs + lo), 16, 0, 0);
        }
    }
    if (HAS_BIAS) {
        const float* bt = bias_tbl + (int64_t)h * ARX_BIAS_ROW + ARX_BIAS_CENTER;
        const int span = 2 * Lk + 8;
        for (int i = tid; i < 4 * span; i += NT) {
            const int c = (i >= span) + (i >= 2 * span) + (i >= 3 * span), j = i - c * span;      // i / span without the software division
            int d = j + c - Lk;
            d = d < -ARX_BIAS_CENTER ? -ARX_BIAS_CENTER : (d > ARX_BIAS_CENTER ? ARX_BIAS_CENTER : d);
            Bs[c * bst + attn_bias_off(c) + j] = bt[d];
        }
    }
    __syncthreads();
#ifdef ARX_DEV_VARIANTS
    if (dev_stamps && threadIdx.x == 0) dev_stamps[4 * lin + 1] = wall_clock64();
    if (dev_word == 3) {                                   // probe: staging only (a never-taken store keeps the Q loads alive)
        if (q < L && qf[0][0] == 0x7fc1 && qf[1 % KS][1] == 0x7fc2) ctx[(int64_t)(t0 + q) * H + h * DH] = 1;
        return;
    }
#endif

    if (qw >= L) return;
    f32x16 o[DB];
    // LDS read bases of this lane (a tile's reads are base + 32-key tile offset + constant); recomputed by whichever pass runs, so that
    // the optimistic loop does not carry the fallback's copies through its 128-register budget
    auto lds_bases = [&](const float*& bb, const char* (&kb)[KS], const char* (&vb)[DB]) {
        int lane = threadIdx.x & 63;
        asm volatile("" : "+v"(lane));                  // opaque: a pass derives its addresses from scratch, nothing is shared (and kept live)
        const int ql = lane & 31, hh = lane >> 5;
        int qc = q0 + wid * 32 + ql;
        qc = qc < L ? qc : L - 1;
        const int bc = (4 - (qc & 3)) & 3;
        bb = Bs + bc * bst + attn_bias_off(bc) + (Lk - qc - bc) + 4 * hh;
        const int krow_sw = ((ql / RPB) & (CPR - 1));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kb[ks] = Ks + ql * ROWB + (((2 * ks + hh) ^ krow_sw) << 4);
        const int g16 = lane >> 4, q_ = (lane & 15) >> 2, p_ = lane & 3;
        const int vsw = (DH == 64) ? (((q_ >> 1) & 1) << 2) : 0;
#pragma unroll
        for (int d = 0; d < DB; ++d) {
            const int col = 32 * d + 16 * (g16 & 1) + 4 * p_;
            vb[d] = Vs + (4 * hh + q_) * ROWB + (((col >> 3) ^ vsw) << 4) + ((p_ & 1) << 3);
        }
    };
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    f32x4 lsum;
    const s16x4 one4 = s16x4{0x3f80, 0x3f80, 0x3f80, 0x3f80};
    const int nkt = Lk >> 5;
    const int nfull = (L & 31) ? nkt - 1 : nkt;           // tiles with no masked key
    float l_tot;

    // OPTIMISTIC pass: softmax with the FIXED reference 0 — P = exp2(v) straight from the scaled, biased score, no running maximum, no
    // accumulator pre-load, no cross-lane exchange, no rescaling: per 32-key tile that removes 16 v_mov, 16 max, a ds_bpermute round
    // trip and a branch from the ~135 vector instructions of the exact tile below (85 remain).  fp32 and bf16 share an 8-bit exponent,
    // so P, the row sum and O keep their relative precision at any magnitude; what a fixed reference cannot do is keep them in RANGE.
    // The pass is therefore accepted only if every row sum of the wave ended inside [2^-100, 2^100] (with |V| < 2^27 nothing overflowed on
    // the way; an inf / NaN / fully underflowed row fails the test) — otherwise the wave redoes its queries with the exact loop.  Scores of
    // real encoders sit within +-30 base-2 units; tests/test_gpu_parity.py drives both outcomes.
    // What it buys is modest (0.415 -> 0.40 ms at 1024 x 256 tokens, tools/attn_probe.py): with the tile loop emptied out entirely the
    // launch still takes 0.33 ms (staging alone 0.215 ms) — the kernel is bound by what a CU can take in and put out per block
    // (96 KB in, 32 KB out, ~12.5 B/clk/CU = the chip's 6.4 TB/s), not by its arithmetic.
    bool redo = false;
#ifdef ARX_DEV_VARIANTS
    redo = dev_word == 6;
#endif
    if (!redo) {
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
        lsum = f32x4{0.f, 0.f, 0.f, 0.f};
        typedef const __attribute__((address_space(3))) char* lds_cptr;      // 32-bit LDS pointers: one VGPR each, no flat-address casts in the loop
        typedef const __attribute__((address_space(3))) float* lds_fptr;
        lds_fptr bp;
        lds_cptr kp[KS], vp[DB];
        {
            const float* bb;
            const char* kb[KS];
            const char* vb[DB];
            lds_bases(bb, kb, vb);
            bp = (lds_fptr)bb;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) kp[ks] = (lds_cptr)kb[ks];
#pragma unroll
            for (int d = 0; d < DB; ++d) vp[d] = (lds_cptr)vb[d];
        }
        // One tile, software-pipelined by hand (sched_barrier pins the stage order; the waits hipcc inserts are counted, LDS returns in
        // order).  Left to itself hipcc issues every LDS read right before its use — twelve exposed LDS round trips per tile and wave
        // (tools/attn_probe.py).  Here every read is issued a stage ahead of its use:
        //   S = K(t) Q^T (K fragments requested during tile t-1) | bias rows(t), V^T fragments(t) | exp2 / pack | K fragments(t+1) | row sums, PV
        // Tiles go in pairs: the second tile's reads are immediate offsets of the first's address registers, which advance once per pair.
        bf16x8 kf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kf[ks] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(kp[ks]);
        auto advance = [&](int n) {
            bp += 32 * n;
            asm volatile("" : "+v"(bp));                       // opaque: otherwise the loop passes rewrite the seven addresses as base + induction
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { kp[ks] += 32 * ROWB * n; asm volatile("" : "+v"(kp[ks])); }      // variable and copy each one
#pragma unroll
            for (int d = 0; d < DB; ++d) { vp[d] += 32 * ROWB * n; asm volatile("" : "+v"(vp[d])); }           // (v_add 0) per use: +7 registers
            __builtin_amdgcn_sched_barrier(0);
        };
        auto fast = [&](auto masked_tag, auto off_tag, int kt) {
            constexpr bool MASKED = decltype(masked_tag)::value;
            constexpr int OFF = decltype(off_tag)::value;      // tile inside the pair
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;           // a constant-zero C operand: no register initialisation
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], s, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 bv[4];                                       // requested behind the S MFMAs: their 4 x 32 cycles cover the LDS round trip
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) bv[g4] = HAS_BIAS ? *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(bp + OFF * 32 + 8 * g4) : f32x4{0.f, 0.f, 0.f, 0.f};
            s16x4 vt[DB][4];
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int j = 0; j < 4; ++j) vt[d][j] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vp[d] + (OFF * 32 + 8 * j) * ROWB));
            __builtin_amdgcn_sched_barrier(0);
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x2 sc2 = f32x2{scale_log2e, scale_log2e};
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                for (int e = 0; e < 4; e += 2) {               // v_pk_fma_f32: two scores per instruction
                    f32x2 v = __builtin_elementwise_fma(f32x2{s[g4 * 4 + e], s[g4 * 4 + e + 1]}, sc2, f32x2{bv[g4][e], bv[g4][e + 1]});
                    if (MASKED) {
                        v[0] = (kt * 32 + 8 * g4 + 4 * hh + e < L) ? v[0] : -INFINITY;
                        v[1] = (kt * 32 + 8 * g4 + 4 * hh + e + 1 < L) ? v[1] : -INFINITY;
                    }
                    s[g4 * 4 + e] = v[0]; s[g4 * 4 + e + 1] = v[1];
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(s[r]);
            uint32_t pw[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) pw[j] = pack_bf16x2(s[2 * j], s[2 * j + 1]);
            __builtin_amdgcn_sched_barrier(0);
            if (!MASKED) {                                     // after the last K tile this reads the head of V: in bounds, unused
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) kf[ks] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(kp[ks] + (OFF + 1) * 32 * ROWB);
            }
            __builtin_amdgcn_sched_barrier(0);
            bf16x8 pf[2];
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) pf[ss] = __builtin_bit_cast(bf16x8, u32x4{pw[4 * ss], pw[4 * ss + 1], pw[4 * ss + 2], pw[4 * ss + 3]});
#pragma unroll
            for (int j = 0; j < 4; ++j)
                lsum = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(one4, __builtin_bit_cast(s16x4, u32x2{pw[2 * j], pw[2 * j + 1]}), lsum, 0, 0, 0);
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int ss = 0; ss < 2; ++ss) {
                    s16x8 v8;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v8[e] = vt[d][2 * ss][e]; v8[4 + e] = vt[d][2 * ss + 1][e]; }
                    o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v8), pf[ss], o[d], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        };
        int kt = 0;
        for (; kt + 2 <= nfull; kt += 2) {
            fast(std::false_type{}, std::integral_constant<int, 0>{}, kt);
            fast(std::false_type{}, std::integral_constant<int, 1>{}, kt + 1);
            advance(2);
        }
        if (kt < nfull) { fast(std::false_type{}, std::integral_constant<int, 0>{}, kt); advance(1); }
        if (nfull < nkt) fast(std::true_type{}, std::integral_constant<int, 0>{}, nkt - 1);
        l_tot = lsum[0] + __shfl_xor(lsum[0], 32);         // the two lane halves hold disjoint keys of the same query
        redo = !__all(l_tot > 0x1p-100f && l_tot < 0x1p100f);
    }
    if (redo) {
    // EXACT pass (the wave's fallback; wave-uniform branch)
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -INFINITY;
    const float* bptr;
    const char* kptr[KS];
    const char* vptr[DB];
    lds_bases(bptr, kptr, vptr);
    // Online softmax with a LAZY reference: the running reference m_run is folded into the S accumulator's initial value
    // (s0 = -m_run / scale), so fma(acc, scale, bias) already yields v - m_run and exp2 needs no subtraction; the
    // reference is only moved (and O, l rescaled) when some lane's tile maximum exceeds it by more than THR (base-2
    // units; P then stays <= 2^THR, exact up to the final normalisation).  Tile 0 always takes the exact-maximum path
    // (no reference exists yet).  Row sums ride on the matrix pipe: the 4x4x4 MFMA with A = ones adds each lane's own four bf16 B
    // values into its accumulator (4 registers and 4 x 8 cycles per tile; the 32x32x16 form of round 1 took 16 registers and 64
    // cycles to compute the same sum in every row); the two lane halves of a query are added once, at the end.
    constexpr float THR = 8.0f;
    const float inv_scale = 1.0f / scale_log2e;
    lsum = f32x4{0.f, 0.f, 0.f, 0.f};
    auto tile = [&](auto masked_tag, auto first_tag, int kt) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        f32x16 s;
        const float s0 = FIRST ? 0.f : -m_run * inv_scale;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = s0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kptr[ks]);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
        }
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
            if (HAS_BIAS) bv = *reinterpret_cast<const f32x4*>(bptr + 8 * g4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = fmaf(s[g4 * 4 + e], scale_log2e, bv[e]);
                if (MASKED) v = (kt * 32 + 8 * g4 + 4 * hh + e < L) ? v : -INFINITY;
                s[g4 * 4 + e] = v;
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; r += 2) mx = fmaxf(mx, fmaxf(s[r], s[r + 1]));
        mx = fmaxf(mx, __shfl_xor(mx, 32));                   // tile maximum relative to the reference (FIRST: absolute)
        if (FIRST) {
            m_run = mx;                                       // finite: key 0 is always valid
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] -= mx;
        } else if (__any(mx > THR)) {
            const float d = fmaxf(mx, 0.f);                   // lanes whose maximum did not grow keep their reference
            const float alpha = __builtin_amdgcn_exp2f(-d);
            m_run += d;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] -= d;
#pragma unroll
            for (int r = 0; r < 4; ++r) lsum[r] *= alpha;
#pragma unroll
            for (int dd = 0; dd < DB; ++dd)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dd][r] *= alpha;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(s[r]);
        uint32_t pw[8];                                       // P as bf16, one v_cvt_pk_bf16_f32 per pair
#pragma unroll
        for (int j = 0; j < 8; ++j) pw[j] = pack_bf16x2(s[2 * j], s[2 * j + 1]);
        bf16x8 pf[2];
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) pf[ss] = __builtin_bit_cast(bf16x8, u32x4{pw[4 * ss], pw[4 * ss + 1], pw[4 * ss + 2], pw[4 * ss + 3]});
#pragma unroll
        for (int j = 0; j < 4; ++j)
            lsum = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(one4, __builtin_bit_cast(s16x4, u32x2{pw[2 * j], pw[2 * j + 1]}), lsum, 0, 0, 0);
#pragma unroll
        for (int d = 0; d < DB; ++d) {
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vptr[d] + (16 * ss) * ROWB));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vptr[d] + (16 * ss + 8) * ROWB));
                s16x8 v8;
#pragma unroll
                for (int e = 0; e < 4; ++e) { v8[e] = lo[e]; v8[4 + e] = hi[e]; }
                o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v8), pf[ss], o[d], 0, 0, 0);
            }
        }
        // advance to the next 32-key tile
        bptr += 32;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kptr[ks] += 32 * ROWB;
#pragma unroll
        for (int d = 0; d < DB; ++d) vptr[d] += 32 * ROWB;
    };
    if (nfull > 0) tile(std::false_type{}, std::true_type{}, 0); else tile(std::true_type{}, std::true_type{}, 0);
    for (int kt = 1; kt < nfull; ++kt) tile(std::false_type{}, std::false_type{}, kt);
    if (nfull < nkt && nkt > 1) tile(std::true_type{}, std::false_type{}, nkt - 1);
    l_tot = lsum[0] + __shfl_xor(lsum[0], 32);
    }

    const float inv = 1.0f / l_tot;
    // Output row of a query: lanes (q, hh = 0 / 1) hold interleaved 8-B pieces of it (d = 8 g4 + 4 hh + [0, 4) per 32-wide block).  Four
    // v_permlane32_swap per block regroup them — the lower lane keeps d 0..15, the upper lane d 16..31 — so that a lane stores 16 B at a
    // time: four dwordx4 stores per lane instead of eight dwordx2 (the store tail of a block is issue-bound: half the instructions).
    u32x4 pc[DB][2];
#pragma unroll
    for (int d = 0; d < DB; ++d) {
        uint32_t w[4][2];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            w[g4][0] = pack_bf16x2(o[d][g4 * 4 + 0] * inv, o[d][g4 * 4 + 1] * inv);
            w[g4][1] = pack_bf16x2(o[d][g4 * 4 + 2] * inv, o[d][g4 * 4 + 3] * inv);
        }
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                // swap(upper half of w[g], lower half of w[g + 2]): lower lanes end with {own g, partner's g}, upper lanes with {partner's g + 2, own g + 2}
                const auto sw = __builtin_amdgcn_permlane32_swap(w[g][k], w[g + 2][k], false, false);
                pc[d][g][k] = sw[0]; pc[d][g][2 + k] = sw[1];
            }
    }
    if (q < L) {
        uint16_t* orow = ctx + (int64_t)(t0 + q) * H + h * DH + 16 * hh;
#pragma unroll
        for (int d = 0; d < DB; ++d) {
            *reinterpret_cast<u32x4*>(orow + d * 32) = pc[d][0];            // plain stores: `nt` on these 16-B pieces gives the gain back (0.397 ms)
            *reinterpret_cast<u32x4*>(orow + d * 32 + 8) = pc[d][1];
        }
    }
#ifdef ARX_DEV_VARIANTS
    if (dev_stamps && (threadIdx.x & 63) == 0) atomicMax(&dev_stamps[4 * lin + 2], (unsigned long long)wall_clock64());
#endif
}

#ifdef ARX_DEV_VARIANTS
// (dev build only: the two streaming attention kernels of round 2 — both correct and tested, both slower than v2 on this chip;
//  DESIGN.md "attention: what the streaming kernels showed" has the numbers)
// ---------------------------------------------------------------------------------------------------
// attention v3 ("ring"): the same arithmetic as attention_tr_kernel, bit for bit — same tiles, same MFMA order, same lazy softmax
// reference — under a different memory schedule.  The v2 kernel stages a whole (sequence, head) before its single barrier and
// loads Q after it: 34 % of its wave time is parked there and the matrix pipe is 28 % busy, while its HBM traffic is already
// the algorithmic 1.0x (rocprofv3, r01).  Here a block is PERSISTENT (grid = heads x G, G = blocks per CU x CUs / heads; block
// (g, h) walks sequences g, g + G, ... of head h) and everything it reads arrives through ONE stream of 16-KB slot-loads
// (DH = 64; 8 KB for DH = 32) into an 8-slot LDS ring, by LDS-DMA, seven slot-loads ahead of the consumer and straight through
// item boundaries (one 8-wave block per CU: the body needs ~170 VGPRs — at the 128 of two blocks per CU hipcc spills into the
// tile loop, and every scratch reload is a vmcnt(0) behind the stream):
//     item stream = [ Q part 0 (queries 0..127) , Q part 1 (if L > 128) , KV slot 0 (keys 0..63: K rows then V rows) , KV slot 1 , ... ]
// Step j of the consumer: wait until slot-load j has landed (counted vmcnt: only the YOUNGER operations may stay in flight),
// barrier, issue slot-load j + 7 into the slot step j - 1 just released, then either copy this wave's Q fragments to registers
// (Q step) or run the slot's two 32-key tiles.  One barrier per step, never vmcnt(0), ~100 KB in flight per CU at any time.
// The MPNet bias rows are staged ONCE per block (the head is fixed) around a fixed centre C0 = round32(max_len).
// vmcnt bookkeeping is exact and wave-uniform: every slot-load is GPS global_load_lds per thread whatever the item's length
// (rows past L re-read row L - 1); the only other vector-memory operations in the loop are the DH/8 output stores of a wave
// that has queries in the item, written as asm so that their count is the source's (AttnStream counts both kinds).
template <int DH, int NSLOT_ = 8> struct AttnRing {
    static constexpr int NT = 512, NSLOT = NSLOT_, AHEAD = NSLOT - 1, SLOT_KEYS = 64;
    static constexpr int ROWB = DH * 2;                              // bytes per K / V / Q row
    static constexpr int SLOT_BYTES = 2 * SLOT_KEYS * ROWB;          // 64 K rows + 64 V rows, or 128 Q rows
    static constexpr int GPS = SLOT_BYTES / (NT * 16);               // global_load_lds per thread per slot-load
    static_assert(GPS >= 1 && GPS * NT * 16 == SLOT_BYTES, "slot must be whole passes of the block");
    static __host__ __device__ int bias_stride(int C0) { return (2 * C0 + 8 + 52 + 63) & ~63; }
    static __host__ __device__ int total(int C0, bool has) { return NSLOT * SLOT_BYTES + (has ? 4 * bias_stride(C0) * 4 : 0); }
};

#define ARX_VMCNT_CASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
#define ARX_VMCNT_CASE4(A, B, C_, D) ARX_VMCNT_CASE(A) ARX_VMCNT_CASE(B) ARX_VMCNT_CASE(C_) ARX_VMCNT_CASE(D)
__device__ __forceinline__ void attn_wait_vm(int n) {      // n is wave-uniform: the younger operations that may stay in flight
    switch (n) {
    ARX_VMCNT_CASE4(0, 1, 2, 3) ARX_VMCNT_CASE4(4, 5, 6, 7) ARX_VMCNT_CASE4(8, 9, 10, 11) ARX_VMCNT_CASE4(12, 13, 14, 15)
    ARX_VMCNT_CASE4(16, 17, 18, 19) ARX_VMCNT_CASE4(20, 21, 22, 23) ARX_VMCNT_CASE4(24, 25, 26, 27) ARX_VMCNT_CASE4(28, 29, 30, 31)
    ARX_VMCNT_CASE4(32, 33, 34, 35) ARX_VMCNT_CASE4(36, 37, 38, 39) ARX_VMCNT_CASE4(40, 41, 42, 43) ARX_VMCNT_CASE(44)
    // AHEAD - 1 = 6 younger slot-loads (12 pieces) + the stores of up to four short items that ended among them (4 x 8)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;      // anything larger: over-wait (safe)
    }
}
#undef ARX_VMCNT_CASE4
#undef ARX_VMCNT_CASE

// The stream state of one block (every member wave-uniform: SGPRs).  Plain struct + force-inlined members: the lambda form of
// the same code left its closures in scratch memory.
template <int DH, int NSLOT> struct AttnStream {
    using R = AttnRing<DH, NSLOT>;
    static constexpr int NT = R::NT, ROWB = R::ROWB, SLOT_BYTES = R::SLOT_BYTES, GPS = R::GPS, CPR = DH / 8, RPB = 256 / R::ROWB;
    const uint16_t* qkv; const int32_t* cu; char* ring;
    int H, h, G, n_seqs;
    uint32_t ld;
    int p_b, p_t0, p_L, p_step, p_nqp, p_nsteps, p_slot, c_slot;
    // vmcnt bookkeeping without arrays: `inflight` slot-loads are outstanding (oldest = index 0); bit i of `st_bits` says that one
    // item's output stores were issued just before slot-load i; `pend_st` counts the store operations issued since the youngest
    // slot-load.  Operations younger than the oldest slot-load = (inflight - 1) GPS + NSTORE popcount(st_bits >> 1) + pend_st.
    int inflight, pend_st;
    unsigned st_bits;

    __device__ __forceinline__ void p_load_item() {      // first non-empty sequence at or after p_b (stride G); p_L = 0 at the end
        p_L = 0; p_step = 0;
        while (p_b < n_seqs) {
            const __attribute__((address_space(4))) int32_t* cuc = (const __attribute__((address_space(4))) int32_t*)cu;
            const int t0 = cuc[p_b], L = cuc[p_b + 1] - t0;
            if (L > 0) { p_t0 = t0; p_L = L; p_nqp = L > 128 ? 2 : 1; p_nsteps = p_nqp + ((L + 63) >> 6); break; }
            p_b += G;
        }
    }
    __device__ __forceinline__ void issue_next() {       // one slot-load of the producer's item into ring slot p_slot; advances the cursor
        const uint16_t* base = qkv + (int64_t)p_t0 * ld + h * DH;
        const bool isq = p_step < p_nqp;
        const int r0 = isq ? 128 * p_step : 64 * (p_step - p_nqp);
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));               // opaque: the per-thread source offsets are rebuilt here (a dozen VALU ops per slot-load)
                                                  // instead of living in a dozen VGPRs across the whole tile loop
        char* dst = ring + p_slot * SLOT_BYTES + (__builtin_amdgcn_readfirstlane(t) & ~63) * 16;
#pragma unroll
        for (int it = 0; it < GPS; ++it) {
            const int cid = it * NT + t, r128 = cid / CPR, pcn = cid % CPR;
            int row, col0, c;
            if (isq) { row = r0 + r128; col0 = 0; c = pcn ^ ((r128 / RPB) & (CPR - 1)); }
            else if (r128 < 64) { row = r0 + r128; col0 = H; c = pcn ^ ((r128 / RPB) & (CPR - 1)); }
            else { row = r0 + r128 - 64; col0 = 2 * H; c = (DH == 64) ? (pcn ^ (((r128 >> 1) & 1) << 2)) : pcn; }
            row = row < p_L ? row : p_L - 1;      // rows past the sequence re-read its last row: finite, masked or unused
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (uint32_t)row * ld + col0 + c * 8),
                                             (__attribute__((address_space(3))) void*)(dst + it * NT * 16), 16, 0, 0);
        }
        if (pend_st) st_bits |= 1u << inflight;
        pend_st = 0;
        ++inflight;
        p_slot = (p_slot + 1) & (R::NSLOT - 1);
        if (++p_step == p_nsteps) { p_b += G; p_load_item(); }
    }
    __device__ __forceinline__ const char* step_begin() {  // start of a consumer step: returns the slot that has just become readable
        attn_wait_vm((inflight - 1) * GPS + (DH / 8) * __builtin_popcount(st_bits >> 1) + pend_st);   // this step's slot-load has landed (this wave's pieces) ...
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();             // ... and everyone's; every wave is also done with the slot of the step before
        asm volatile("" ::: "memory");
        st_bits >>= 1;
        --inflight;
        if (p_L > 0) issue_next();                // slot-load j + AHEAD -> the slot released by step j - 1
        const char* slot = ring + c_slot * SLOT_BYTES;
        c_slot = (c_slot + 1) & (R::NSLOT - 1);
        return slot;
    }
};

// one 32-key tile of the ring kernel: the tile body of attention_tr_kernel with the row sums on the 4x4x4 MFMA
template <int DH, bool HAS_BIAS, bool MASKED, bool first>
__device__ __forceinline__ void attn_ring_tile(const char* kbase, int k0, int v0, const bf16x8 (&qf)[DH / 16], f32x16 (&o)[DH / 32],
                                               f32x4& lsum, float& m_run, const float* bp, int kt, int hh, int L,
                                               float scale_log2e, float inv_scale) {
    constexpr int KS = DH / 16, DB = DH / 32, ROWB = DH * 2;
    constexpr float THR = 8.0f;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    f32x16 s;
    const float s0 = first ? 0.f : -m_run * inv_scale;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = s0;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kbase + (k0 ^ (ks << 5)));
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
    }
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
        if (HAS_BIAS) bv = *reinterpret_cast<const f32x4*>(bp + 8 * g4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = fmaf(s[g4 * 4 + e], scale_log2e, bv[e]);
            if (MASKED) v = (kt * 32 + 8 * g4 + 4 * hh + e < L) ? v : -INFINITY;
            s[g4 * 4 + e] = v;
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; r += 2) mx = fmaxf(mx, fmaxf(s[r], s[r + 1]));
    mx = fmaxf(mx, __shfl_xor(mx, 32));           // tile maximum relative to the reference (first tile: absolute, finite: key 0 is valid)
    if (first) {                                  // no reference yet: take the exact maximum (O and the row sums are still zero)
        m_run = mx;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] -= mx;
    } else if (__any(mx > THR)) {                 // move the reference: lanes whose maximum did not grow keep theirs (d = 0)
        const float d = fmaxf(mx, 0.f);
        const float alpha = __builtin_amdgcn_exp2f(-d);
        m_run += d;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] -= d;
#pragma unroll
        for (int r = 0; r < 4; ++r) lsum[r] *= alpha;
#pragma unroll
        for (int dd = 0; dd < DB; ++dd)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dd][r] *= alpha;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(s[r]);
    // P as bf16: one v_cvt_pk_bf16_f32 per pair; pw[4 ss + j] holds elements 2j, 2j + 1 of k-step ss
    uint32_t pw[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) pw[j] = pack_bf16x2(s[2 * j], s[2 * j + 1]);
    bf16x8 pf[2];
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) pf[ss] = __builtin_bit_cast(bf16x8, u32x4{pw[4 * ss], pw[4 * ss + 1], pw[4 * ss + 2], pw[4 * ss + 3]});
    // row sums on the matrix pipe: the 4x4x4 MFMA with A = ones adds each lane's own four B values into its accumulator
    const s16x4 one4 = s16x4{0x3f80, 0x3f80, 0x3f80, 0x3f80};
#pragma unroll
    for (int j = 0; j < 4; ++j)
        lsum = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(one4, __builtin_bit_cast(s16x4, u32x2{pw[2 * j], pw[2 * j + 1]}), lsum, 0, 0, 0);
    // V^T operands by the transposing LDS read, written as asm: behind the BUILTIN hipcc puts s_waitcnt vmcnt(0) (it cannot tell
    // the read from the ring's in-flight LDS-DMA writes), which would drain the stream twice per tile.  LDS operations return in
    // order, so the compiler's own counted lgkmcnt waits stay valid beside these; ours is lgkmcnt(0) and names every destination.
    const uint32_t vaddr = (uint32_t)(size_t)(const __attribute__((address_space(3))) char*)kbase + (uint32_t)v0;
#pragma unroll
    for (int d = 0; d < DB; ++d) {
        const uint32_t va = vaddr ^ (uint32_t)(d << 6);       // kbase is 64-B aligned and v0 < 2^16: the xor acts on v0's bit 6 only
        s16x4 vt[2][2];
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vt[ss][0]) : "v"(va), "n"((16 * ss) * ROWB));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vt[ss][1]) : "v"(va), "n"((16 * ss + 8) * ROWB));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vt[0][0]), "+v"(vt[0][1]), "+v"(vt[1][0]), "+v"(vt[1][1]) :: "memory");
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            s16x8 v8;
#pragma unroll
            for (int e = 0; e < 4; ++e) { v8[e] = vt[ss][0][e]; v8[4 + e] = vt[ss][1][e]; }
            o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v8), pf[ss], o[d], 0, 0, 0);
        }
    }
}

template <int DH, bool HAS_BIAS, int NSLOT>
__global__ __launch_bounds__(512, 2) void attention_ring_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ ctx,
                                                                 const int32_t* __restrict__ cu,
                                                                 const float* __restrict__ bias_tbl, int H, int n_seqs,
                                                                 int C0, float scale_log2e) {
    using R = AttnRing<DH, NSLOT>;
    constexpr int NT = R::NT, ROWB = R::ROWB, SLOT_BYTES = R::SLOT_BYTES, GPS = R::GPS;
    constexpr int CPR = DH / 8;                   // 16-B chunks per row
    constexpr int RPB = 256 / ROWB;               // rows per 256-B bank row
    constexpr int KS = DH / 16;
    constexpr int DB = DH / 32;
    constexpr int NSTORE = DB * 4;                // output stores per lane per item
    static_assert((R::AHEAD - 1) * GPS + 4 * NSTORE <= 44, "attn_wait_vm covers AHEAD - 1 younger slot-loads plus four items' stores");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int heads = H / DH;
    const int h = blockIdx.x % heads, g = blockIdx.x / heads, G = gridDim.x / heads;
    float* const Bs = reinterpret_cast<float*>(smem + R::NSLOT * SLOT_BYTES);
    const int bst = R::bias_stride(C0);

    if (HAS_BIAS) {      // four shifted copies of this head's Toeplitz row around the fixed centre C0: once per block
        const float* bt = bias_tbl + (int64_t)h * ARX_BIAS_ROW + ARX_BIAS_CENTER;
        const int span = 2 * C0 + 8;
        for (int i = tid; i < 4 * span; i += NT) {
            const int c = i / span, j = i - c * span;
            int d = j + c - C0;
            d = d < -ARX_BIAS_CENTER ? -ARX_BIAS_CENTER : (d > ARX_BIAS_CENTER ? ARX_BIAS_CENTER : d);
            Bs[c * bst + attn_bias_off(c) + j] = bt[d];
        }
    }
    __syncthreads();                              // (also drains the bias loads: nothing is in flight when the stream starts)

    AttnStream<DH, NSLOT> st;
    st.qkv = qkv; st.cu = cu; st.ring = smem; st.H = H; st.h = h; st.G = G; st.n_seqs = n_seqs; st.ld = 3u * (uint32_t)H;
    st.p_b = g; st.p_t0 = 0; st.p_L = 0; st.p_step = 0; st.p_nqp = 0; st.p_nsteps = 0; st.p_slot = 0; st.c_slot = 0;
    st.inflight = 0; st.pend_st = 0; st.st_bits = 0u;
    st.p_load_item();
    if (st.p_L == 0) return;                      // block-uniform: no work for this block
#pragma unroll 1
    for (int i = 0; i < R::AHEAD && st.p_L > 0; ++i) st.issue_next();

    // ---- per-lane constants of the tile body (byte offsets inside a slot)
    const int ql = lane & 31, hh = lane >> 5;
    // fragment ks of this lane's K row sits at k0 ^ (ks << 5): 2 ks + hh and the row swizzle occupy disjoint bits of the chunk index
    const int k0 = ql * ROWB + ((hh ^ ((ql / RPB) & (CPR - 1))) << 4);
    const int q_ = (lane & 15) >> 2, p_ = lane & 3;
    const int vsw = (DH == 64) ? (((q_ >> 1) & 1) << 2) : 0;
    // V fragment offsets: block d of the head dimension sits at v0 ^ (d << 6) (d selects bit 2 of the chunk index)
    const int v0 = 64 * ROWB + (4 * hh + q_) * ROWB + ((((16 * ((lane >> 4) & 1) + 4 * p_) >> 3) ^ vsw) << 4) + ((p_ & 1) << 3);
    const float inv_scale = 1.0f / scale_log2e;

    int b = g;
#pragma unroll 1
    for (;;) {
        // ---- next item of the consumer (same walk as the producer's)
        int t0 = 0, L = 0;
        while (b < n_seqs) {                      // scalar loads (constant address space): a vector load here would be a vmcnt(0) per item
            const __attribute__((address_space(4))) int32_t* cuc = (const __attribute__((address_space(4))) int32_t*)cu;
            t0 = cuc[b]; L = cuc[b + 1] - t0;
            if (L > 0) break;
            b += G;
        }
        if (b >= n_seqs) break;
        const int nqp = L > 128 ? 2 : 1, nkv = (L + 63) >> 6, nkt = (L + 31) >> 5;
        const bool wave_on = wid * 32 < L;        // wave-uniform (wid is an SGPR)
        bf16x8 qf[KS];
        f32x16 o[DB];
        f32x4 lsum = f32x4{0.f, 0.f, 0.f, 0.f};   // per-lane sum of its bf16-rounded probabilities (all four entries equal)
        float m_run = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[ks][e] = (bf16_t)0.f;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
        const int q = wid * 32 + ql, qc = q < L ? q : L - 1;
        const int bc = (4 - (qc & 3)) & 3;
        const float* bptr = Bs + bc * bst + attn_bias_off(bc) + (C0 - qc - bc) + 4 * hh;

        // ---- Q steps: the waves of each 128-query part copy their fragments to registers
#pragma unroll 1
        for (int p = 0; p < nqp; ++p) {
            const char* slot = st.step_begin();
            if (wave_on && (wid >> 2) == p) {
                const int r = (wid & 3) * 32 + ql;
                const char* qrow = slot + r * ROWB;
                const int sw = (r / RPB) & (CPR - 1);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qrow + (((2 * ks + hh) ^ sw) << 4));
            }
        }
        // ---- KV steps: two 32-key tiles each
        const bool ragged = (L & 31) != 0;
#pragma unroll 1
        for (int sidx = 0; sidx < nkv; ++sidx) {
            const char* slot = st.step_begin();
            if (wave_on) {
#pragma unroll 1
                for (int t2 = 0; t2 < 2; ++t2) {
                    const int kt = 2 * sidx + t2;
                    if (kt >= nkt) break;
                    const char* kbase = slot + t2 * 32 * ROWB;
                    const bool msk = ragged && kt == nkt - 1;
                    if (kt == 0) {
                        if (msk) attn_ring_tile<DH, HAS_BIAS, true, true>(kbase, k0, v0, qf, o, lsum, m_run, bptr, kt, hh, L, scale_log2e, inv_scale);
                        else attn_ring_tile<DH, HAS_BIAS, false, true>(kbase, k0, v0, qf, o, lsum, m_run, bptr, kt, hh, L, scale_log2e, inv_scale);
                    } else {
                        if (msk) attn_ring_tile<DH, HAS_BIAS, true, false>(kbase, k0, v0, qf, o, lsum, m_run, bptr + 32 * kt, kt, hh, L, scale_log2e, inv_scale);
                        else attn_ring_tile<DH, HAS_BIAS, false, false>(kbase, k0, v0, qf, o, lsum, m_run, bptr + 32 * kt, kt, hh, L, scale_log2e, inv_scale);
                    }
                }
            }
        }
        // ---- item done: normalise and store this wave's rows
        if (wave_on) {
            const float l_tot = lsum[0] + __shfl_xor(lsum[0], 32);      // the two lane halves hold disjoint keys of the same query
            const float inv = 1.0f / l_tot;
            uint16_t* orow = ctx + (int64_t)(t0 + qc) * H + h * DH + 4 * hh;
            if (q < L) {
#pragma unroll
                for (int d = 0; d < DB; ++d)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        u32x2 w2;
                        w2[0] = pack_bf16x2(o[d][g4 * 4 + 0] * inv, o[d][g4 * 4 + 1] * inv);
                        w2[1] = pack_bf16x2(o[d][g4 * 4 + 2] * inv, o[d][g4 * 4 + 3] * inv);
                        asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 1" :: "v"(orow + d * 32 + 8 * g4), "v"(w2) : "memory");
                    }
            }
            st.pend_st += NSTORE;                 // lane 0 of an active wave always has q < L: the stores were issued
        }
        b += G;
    }
}

// ---------------------------------------------------------------------------------------------------
// attention v4 ("ring16"): the v3 stream (persistent block per CU, 8-slot LDS ring, LDS-DMA seven slot-loads ahead, counted
// vmcnt) under SIXTEEN waves of 16 queries on mfma_f32_16x16x32_bf16.  v3 showed where the time goes once staging no longer
// stalls: a 32-query wave's tile is a ~2 k-cycle dependency chain (LDS reads -> 4 chained MFMAs -> max / exchange / exp / convert
// -> transposing reads -> PV), its state needs ~165 VGPRs, so one 8-wave block per CU leaves two waves per SIMD and nothing to
// cover that chain with (0.71 ms against v2's 0.42 ms, same box).  A 16-query wave halves every per-wave array (O^T 16 regs,
// S^T 8, Q 8), fits 128 VGPRs, and a 16-wave block puts FOUR waves on each SIMD with the whole LDS still free for the ring.
//   S^T block [16 keys x 16 queries] = K rows . Q^T: lane l holds query l & 15 and keys 4 (l >> 4) + r of the block in its 4
//   accumulator registers; two blocks (keys 0-15, 16-31 of the tile) give the lane 8 probabilities which ARE its B fragment of
//   O^T[16 dh x 16 q] += V^T[16 dh x 32 k] . P^T[32 k x 16 q] under the k <-> key map  k = 8 g + j  ->  key 16 (j >> 2) + 4 g + (j & 3):
//   no lane movement; the V^T operand under the same map is two ds_read_b64_tr_b16 (rows 4 g .. 4 g + 3 and 16 + 4 g ..).
//   Row maxima cross the four lane groups by v_permlane16/32 swaps; row sums stay per lane until the item ends.
// LDS images: K and Q rows as in v2/v3 (16-B chunk ^ ((row >> 1) & 7): the 16-row x 4-chunk fragment read is conflict-free);
// V rows with chunk ^ (((row >> 1) & 3) << 1), which spreads the transposing read's rows 4 g + q over the four 32-B column
// windows of a bank half.
template <int NSLOT_ = 8> struct AttnRing16 {
    static constexpr int DH = 64, NT = 1024, NSLOT = NSLOT_, AHEAD = NSLOT - 1;
    static constexpr int ROWB = DH * 2, SLOT_BYTES = 128 * ROWB;     // 64 K rows + 64 V rows, or 128 Q rows: one 16-B piece per thread
    static __host__ __device__ int bias_stride(int C0) { return (2 * C0 + 8 + 52 + 63) & ~63; }
    static __host__ __device__ int total(int C0, bool has) { return NSLOT * SLOT_BYTES + (has ? 4 * bias_stride(C0) * 4 : 0); }
};

struct AttnStream16 {
    using R = AttnRing16<8>;
    const uint16_t* qkv; const int32_t* cu; char* ring;
    int H, h, G, n_seqs;
    uint32_t ld;
    int p_b, p_t0, p_L, p_step, p_nqp, p_nsteps, p_slot, c_slot;
    int inflight, pend_st;                        // bookkeeping as in AttnStream (one piece per slot-load, 4 stores per item and lane)
    unsigned st_bits;

    __device__ __forceinline__ void p_load_item() {
        p_L = 0; p_step = 0;
        while (p_b < n_seqs) {
            const __attribute__((address_space(4))) int32_t* cuc = (const __attribute__((address_space(4))) int32_t*)cu;
            const int t0 = cuc[p_b], L = cuc[p_b + 1] - t0;
            if (L > 0) { p_t0 = t0; p_L = L; p_nqp = L > 128 ? 2 : 1; p_nsteps = p_nqp + ((L + 63) >> 6); break; }
            p_b += G;
        }
    }
    __device__ __forceinline__ void issue_next() {
        const uint16_t* base = qkv + (int64_t)p_t0 * ld + h * 64;
        const bool isq = p_step < p_nqp;
        const int r0 = isq ? 128 * p_step : 64 * (p_step - p_nqp);
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));               // opaque: source offsets rebuilt per slot-load, not kept in VGPRs across the tile loop
        char* dst = ring + p_slot * R::SLOT_BYTES + (__builtin_amdgcn_readfirstlane(t) & ~63) * 16;
        const int r128 = t >> 3, pcn = t & 7;
        int row, col0, c;
        if (isq) { row = r0 + r128; col0 = 0; c = pcn ^ ((r128 >> 1) & 7); }
        else if (r128 < 64) { row = r0 + r128; col0 = H; c = pcn ^ ((r128 >> 1) & 7); }
        else { row = r0 + r128 - 64; col0 = 2 * H; c = pcn ^ (((r128 >> 1) & 3) << 1); }
        row = row < p_L ? row : p_L - 1;          // rows past the sequence re-read its last row: finite, masked or unused
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (uint32_t)row * ld + col0 + c * 8),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        if (pend_st) st_bits |= 1u << inflight;
        pend_st = 0;
        ++inflight;
        p_slot = (p_slot + 1) & (R::NSLOT - 1);
        if (++p_step == p_nsteps) { p_b += G; p_load_item(); }
    }
    __device__ __forceinline__ const char* step_begin() {
        attn_wait_vm((inflight - 1) + 4 * __builtin_popcount(st_bits >> 1) + pend_st);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        st_bits >>= 1;
        --inflight;
        if (p_L > 0) issue_next();
        const char* slot = ring + c_slot * R::SLOT_BYTES;
        c_slot = (c_slot + 1) & (R::NSLOT - 1);
        return slot;
    }
};

template <bool HAS_BIAS, bool MASKED, bool FIRST>
__device__ __forceinline__ void attn_ring16_tile(const char* kbase, int k0, uint32_t v0, const bf16x8 (&qf)[2], f32x4 (&o)[4], f32x4& lsum,
                                                 float& m_run, const float* bp, int kt, int g, int L, float scale_log2e, float inv_scale) {
    constexpr int ROWB = 128;
    constexpr float THR = 8.0f;
    // ---- S^T for the tile's two 16-key blocks: 2 x 2 MFMAs, two independent accumulators
    f32x4 s[2];
    const float s0 = FIRST ? 0.f : -m_run * inv_scale;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        s[kb] = f32x4{s0, s0, s0, s0};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kbase + kb * 16 * ROWB + (k0 ^ (ks << 6)));
            s[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s[kb], 0, 0, 0);
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
        if (HAS_BIAS) bv = *reinterpret_cast<const f32x4*>(bp + 16 * kb);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = fmaf(s[kb][e], scale_log2e, bv[e]);
            if (MASKED) v = (kt * 32 + 16 * kb + 4 * g + e < L) ? v : -INFINITY;
            s[kb][e] = v;
            mx = fmaxf(mx, v);
        }
    }
    // maximum over the four lane groups that share this query (lanes l, l ^ 16, l ^ 32, l ^ 48)
    {
        const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mx = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
        const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mx = fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
    }
    if (FIRST) {
        m_run = mx;                               // finite: key 0 is valid
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 4; ++e) s[kb][e] -= mx;
    } else if (__any(mx > THR)) {
        const float d = fmaxf(mx, 0.f);
        const float alpha = __builtin_amdgcn_exp2f(-d);
        m_run += d;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 4; ++e) s[kb][e] -= d;
#pragma unroll
        for (int r = 0; r < 4; ++r) lsum[r] *= alpha;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[db][r] *= alpha;
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 4; ++e) s[kb][e] = __builtin_amdgcn_exp2f(s[kb][e]);
    // P^T fragment: element j = 4 kb + e  <->  key 16 kb + 4 g + e  (the k <-> key map of the header)
    uint32_t pw[4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        pw[2 * kb] = pack_bf16x2(s[kb][0], s[kb][1]);
        pw[2 * kb + 1] = pack_bf16x2(s[kb][2], s[kb][3]);
    }
    const bf16x8 pf = __builtin_bit_cast(bf16x8, u32x4{pw[0], pw[1], pw[2], pw[3]});
    typedef short s16x4_ __attribute__((ext_vector_type(4)));
    typedef short s16x8_ __attribute__((ext_vector_type(8)));
    const s16x4_ one4 = s16x4_{0x3f80, 0x3f80, 0x3f80, 0x3f80};
    lsum = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(one4, __builtin_bit_cast(s16x4_, u32x2{pw[0], pw[1]}), lsum, 0, 0, 0);
    lsum = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(one4, __builtin_bit_cast(s16x4_, u32x2{pw[2], pw[3]}), lsum, 0, 0, 0);
    // ---- O^T += V^T . P^T: per 16-row block of the head dimension two transposing reads (keys 4 g + q, 16 + 4 g + q) and one MFMA
    const uint32_t kb_lds = (uint32_t)(size_t)(const __attribute__((address_space(3))) char*)kbase;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        s16x4_ vt[2][2];
#pragma unroll
        for (int d2 = 0; d2 < 2; ++d2) {
            const uint32_t va = kb_lds + (v0 ^ (uint32_t)((2 * half + d2) << 5));   // block db moves the 32-B column window: bits 5, 6 of the row offset
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(vt[d2][0]) : "v"(va));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vt[d2][1]) : "v"(va), "n"(16 * ROWB));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vt[0][0]), "+v"(vt[0][1]), "+v"(vt[1][0]), "+v"(vt[1][1]) :: "memory");
#pragma unroll
        for (int d2 = 0; d2 < 2; ++d2) {
            s16x8_ v8;
#pragma unroll
            for (int e = 0; e < 4; ++e) { v8[e] = vt[d2][0][e]; v8[4 + e] = vt[d2][1][e]; }
            o[2 * half + d2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, v8), pf, o[2 * half + d2], 0, 0, 0);
        }
    }
}

template <bool HAS_BIAS>
__global__ __launch_bounds__(1024, 4) void attention_ring16_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ ctx,
                                                                    const int32_t* __restrict__ cu,
                                                                    const float* __restrict__ bias_tbl, int H, int n_seqs,
                                                                    int C0, float scale_log2e) {
    using R = AttnRing16<8>;
    constexpr int NT = R::NT, ROWB = R::ROWB, SLOT_BYTES = R::SLOT_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int heads = H / 64;
    const int h = blockIdx.x % heads, g_blk = blockIdx.x / heads, G = gridDim.x / heads;
    float* const Bs = reinterpret_cast<float*>(smem + R::NSLOT * SLOT_BYTES);
    const int bst = R::bias_stride(C0);
    if (HAS_BIAS) {
        const float* bt = bias_tbl + (int64_t)h * ARX_BIAS_ROW + ARX_BIAS_CENTER;
        const int span = 2 * C0 + 8;
        for (int i = tid; i < 4 * span; i += NT) {
            const int c = i / span, j = i - c * span;
            int d = j + c - C0;
            d = d < -ARX_BIAS_CENTER ? -ARX_BIAS_CENTER : (d > ARX_BIAS_CENTER ? ARX_BIAS_CENTER : d);
            Bs[c * bst + attn_bias_off(c) + j] = bt[d];
        }
    }
    __syncthreads();

    AttnStream16 st;
    st.qkv = qkv; st.cu = cu; st.ring = smem; st.H = H; st.h = h; st.G = G; st.n_seqs = n_seqs; st.ld = 3u * (uint32_t)H;
    st.p_b = g_blk; st.p_t0 = 0; st.p_L = 0; st.p_step = 0; st.p_nqp = 0; st.p_nsteps = 0; st.p_slot = 0; st.c_slot = 0;
    st.inflight = 0; st.pend_st = 0; st.st_bits = 0u;
    st.p_load_item();
    if (st.p_L == 0) return;
#pragma unroll 1
    for (int i = 0; i < R::AHEAD && st.p_L > 0; ++i) st.issue_next();

    // ---- per-lane constants: query column qi = lane & 15, lane group g = lane >> 4
    const int qi = lane & 15, g = lane >> 4;
    // K / Q fragment (row qi of a 16-row block, 16-B chunk 4 ks + g, swizzled by the row): k0 ^ (ks << 6)
    const int k0 = qi * ROWB + ((g ^ ((qi >> 1) & 7)) << 4);
    // transposing read of V: lane 4 q' + p' of its 16-lane group addresses row 4 g + q', columns 16 db + 4 p' .. + 3
    const int qp = qi >> 2, pp = qi & 3;
    const int vrow = 4 * g + qp;
    const uint32_t v0 = (uint32_t)(64 * ROWB + vrow * ROWB + ((((pp >> 1)) ^ (((vrow >> 1) & 3) << 1)) << 4) + ((pp & 1) << 3));
    const float inv_scale = 1.0f / scale_log2e;

    int b = g_blk;
#pragma unroll 1
    for (;;) {
        int t0 = 0, L = 0;
        while (b < n_seqs) {
            const __attribute__((address_space(4))) int32_t* cuc = (const __attribute__((address_space(4))) int32_t*)cu;
            t0 = cuc[b]; L = cuc[b + 1] - t0;
            if (L > 0) break;
            b += G;
        }
        if (b >= n_seqs) break;
        const int nqp = L > 128 ? 2 : 1, nkv = (L + 63) >> 6, nkt = (L + 31) >> 5;
        const bool wave_on = wid * 16 < L;
        bf16x8 qf[2];
        f32x4 o[4];
        f32x4 lsum = f32x4{0.f, 0.f, 0.f, 0.f};
        float m_run = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[ks][e] = (bf16_t)0.f;
#pragma unroll
        for (int db = 0; db < 4; ++db) o[db] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int q = wid * 16 + qi, qc = q < L ? q : L - 1;
        const int bc = (4 - (qc & 3)) & 3;
        const float* bptr = Bs + bc * bst + attn_bias_off(bc) + (C0 - qc - bc) + 4 * g;

#pragma unroll 1
        for (int p = 0; p < nqp; ++p) {
            const char* slot = st.step_begin();
            if (wave_on && (wid >> 3) == p) {
                const char* qrow = slot + (wid & 7) * 16 * ROWB;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qrow + (k0 ^ (ks << 6)));
            }
        }
        const bool ragged = (L & 31) != 0;
#pragma unroll 1
        for (int sidx = 0; sidx < nkv; ++sidx) {
            const char* slot = st.step_begin();
            if (wave_on) {
#pragma unroll 1
                for (int t2 = 0; t2 < 2; ++t2) {
                    const int kt = 2 * sidx + t2;
                    if (kt >= nkt) break;
                    const char* kbase = slot + t2 * 32 * ROWB;
                    const bool msk = ragged && kt == nkt - 1;
                    if (kt == 0) {
                        if (msk) attn_ring16_tile<HAS_BIAS, true, true>(kbase, k0, v0, qf, o, lsum, m_run, bptr, kt, g, L, scale_log2e, inv_scale);
                        else attn_ring16_tile<HAS_BIAS, false, true>(kbase, k0, v0, qf, o, lsum, m_run, bptr, kt, g, L, scale_log2e, inv_scale);
                    } else {
                        if (msk) attn_ring16_tile<HAS_BIAS, true, false>(kbase, k0, v0, qf, o, lsum, m_run, bptr + 32 * kt, kt, g, L, scale_log2e, inv_scale);
                        else attn_ring16_tile<HAS_BIAS, false, false>(kbase, k0, v0, qf, o, lsum, m_run, bptr + 32 * kt, kt, g, L, scale_log2e, inv_scale);
                    }
                }
            }
        }
        if (wave_on) {
            // row sum over the four lane groups of the query, then normalise and store: lane holds O[q][16 db + 4 g .. + 3]
            float l_tot = lsum[0];
            {
                const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(l_tot), __float_as_uint(l_tot), false, false);
                l_tot = __uint_as_float(a[0]) + __uint_as_float(a[1]);
                const auto c2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_tot), __float_as_uint(l_tot), false, false);
                l_tot = __uint_as_float(c2[0]) + __uint_as_float(c2[1]);
            }
            const float inv = 1.0f / l_tot;
            uint16_t* orow = ctx + (int64_t)(t0 + qc) * H + h * 64 + 4 * g;
            if (q < L) {
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    u32x2 w2;
                    w2[0] = pack_bf16x2(o[db][0] * inv, o[db][1] * inv);
                    w2[1] = pack_bf16x2(o[db][2] * inv, o[db][3] * inv);
                    asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 1" :: "v"(orow + 16 * db), "v"(w2) : "memory");
                }
            }
            st.pend_st += 4;
        }
        b += G;
    }
}

#endif  // ARX_DEV_VARIANTS

// ---------------------------------------------------------------------------------------------------
// Pool (masked mean over the sequence's tokens, or CLS row) + optional L2 normalise.
// One block (256 thr) per sequence; writes f32 and/or fp16 rows.
__global__ __launch_bounds__(256) void pool_norm_kernel(const uint16_t* __restrict__ x, const int32_t* __restrict__ cu,
                                                         int H, int pool_mode, int normalize,
                                                         float* __restrict__ out32, int64_t ld32,
                                                         f16_t* __restrict__ out16, int64_t ld16,
                                                         const float* __restrict__ ln_sum, const float* __restrict__ ln_sq,
                                                         const float* __restrict__ ln_g, const float* __restrict__ ln_b, float eps) {
    // ln_sum != null: x holds the PRE-LayerNorm sums of the last layer; the row LayerNorm is applied while pooling:
    //   mean_t LN(y_t) = gamma o mean_t((y_t - mu_t) rstd_t) + beta
    __shared__ float acc_s[4][1024];
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int t0 = cu[b], L = cu[b + 1] - t0;
    const int nch = H >> 3;
    float a[2][8];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) a[c][e] = 0.f;
    const int nrows = (pool_mode == ARX_POOL_CLS) ? (L > 0 ? 1 : 0) : L;
    for (int r = w; r < nrows; r += 4) {
        const uint16_t* in = x + (int64_t)(t0 + r) * H;
        float mu = 0.f, rs = 1.f;
        if (ln_sum) { mu = ln_sum[t0 + r]; rs = ln_sq[t0 + r]; }      // row mean / rstd from ln_finalize_kernel
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nch) {
                const u32x4 q = *reinterpret_cast<const u32x4*>(in + ch * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float lo, hi;
                    unpack_bf16x2(q[e], lo, hi);
                    a[c][2 * e] += (lo - mu) * rs; a[c][2 * e + 1] += (hi - mu) * rs;
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc_s[w][ch * 8 + e] = a[c][e];
    }
    __syncthreads();
    float v[4];
    float sq = 0.f;
    const float denom = (pool_mode == ARX_POOL_CLS) ? 1.0f : fmaxf((float)L, 1e-9f);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int col = tid + 256 * c;
        v[c] = 0.f;
        if (col < H) {
            v[c] = (acc_s[0][col] + acc_s[1][col] + acc_s[2][col] + acc_s[3][col]) / denom;
            if (ln_sum && L > 0) v[c] = fmaf(v[c], ln_g[col], ln_b[col]);
            sq += v[c] * v[c];
        }
    }
    sq = wave_sum(sq);
    if (lane == 0) red[w] = sq;
    __syncthreads();
    const float nrm = sqrtf(red[0] + red[1] + red[2] + red[3]);
    const float sc = normalize ? 1.0f / fmaxf(nrm, 1e-12f) : 1.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int col = tid + 256 * c;
        if (col < H) {
            const float y = v[c] * sc;
            if (out32) out32[(int64_t)b * ld32 + col] = y;
            if (out16) out16[(int64_t)b * ld16 + col] = (f16_t)y;
        }
    }
}

// Row statistics from the producers' partial slabs, added in fixed order: mean[m], rstd[m].  NP = H / 64 partials per row and
// statistic, all 2 NP loads of a thread in flight at once (the loop over a runtime count kept one pair in flight: 11 us per call
// at 262 144 rows, 24 calls per forward).
template <int NP>
__global__ __launch_bounds__(256) void ln_finalize_kernel(const float* __restrict__ ps, const float* __restrict__ pq, int64_t ld,
                                                           int nparts, const int32_t* __restrict__ n_rows_ptr, float inv_h, float eps,
                                                           float* __restrict__ mean, float* __restrict__ rstd) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= *n_rows_ptr) return;
    float s = 0.f, q = 0.f;
    if constexpr (NP > 0) {
        float vs[NP], vq[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) { vs[p] = ps[p * ld + m]; vq[p] = pq[p * ld + m]; }
#pragma unroll
        for (int p = 0; p < NP; ++p) { s += vs[p]; q += vq[p]; }
    } else {
        for (int p = 0; p < nparts; ++p) { s += ps[p * ld + m]; q += pq[p * ld + m]; }
    }
    const float mu = s * inv_h;
    mean[m] = mu;
    rstd[m] = rsqrtf(fmaxf(q * inv_h - mu * mu, 0.f) + eps);
}

// Fold a LayerNorm into the linear layer that consumes it (one wave per output row n):
//   W'[n][k] = bf16(W[n][k] * gamma[k]);  s[n] = sum_k W'[n][k];  c[n] = bias[n] + sum_k beta[k] * W[n][k]
__global__ __launch_bounds__(256) void fold_ln_kernel(const uint16_t* __restrict__ W, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ bias,
                                                       uint16_t* __restrict__ Wf, float* __restrict__ s, float* __restrict__ c,
                                                       int N, int K) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float ss = 0.f, cc = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float w = bf16_bits_to_f32(W[(int64_t)n * K + k]);
        const bf16_t wf = (bf16_t)(w * gamma[k]);
        Wf[(int64_t)n * K + k] = __builtin_bit_cast(uint16_t, wf);
        ss += (float)wf;
        cc = fmaf(beta[k], w, cc);
    }
    ss = wave_sum(ss); cc = wave_sum(cc);
    if (lane == 0) { s[n] = ss; c[n] = bias[n] + cc; }
}

// cos(e_i, e_{i+1}) for consecutive rows of an f32 matrix: the similarity the stage-3 semantic chunker thresholds at 0.7
// (text_processor.py:1555-1561, _cosine_similarity :1601-1605 = dot / (|a| |b|)).  One wave per pair.
__global__ __launch_bounds__(256) void adjacent_cosine_kernel(const float* __restrict__ e, int64_t ld, int n, int D,
                                                               float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i + 1 >= n) return;
    const float* a = e + (int64_t)i * ld;
    const float* b = a + ld;
    float dot = 0.f, na = 0.f, nb = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(a + c), y = *reinterpret_cast<const f32x4*>(b + c);
#pragma unroll
        for (int r = 0; r < 4; ++r) { dot = fmaf(x[r], y[r], dot); na = fmaf(x[r], x[r], na); nb = fmaf(y[r], y[r], nb); }
    }
    dot = wave_sum(dot); na = wave_sum(na); nb = wave_sum(nb);
    if (lane == 0) out[i] = dot / (sqrtf(na) * sqrtf(nb));
}

// bf16 [n, H] -> f32 (debug tap)
__global__ void bf16_to_f32_kernel(const uint16_t* __restrict__ src, float* __restrict__ dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = bf16_bits_to_f32(src[i]);
}
__global__ void f32_to_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { bf16_t v = (bf16_t)src[i]; dst[i] = __builtin_bit_cast(uint16_t, v); }
}
