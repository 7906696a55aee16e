// arx_encoder_*: host side of the encoder forward (C ABI in include/arx.h).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "arx_common.h"
#include "encoder_kernels.h"
#include "gemm.h"
#include "gemm8.h"
#include "gemm_small.h"

// MPNet / T5 bidirectional bucket (TF modeling_mpnet.py:330-349). rel = key_pos - query_pos.
// Integer-exact restatement: the float32 expression there lands on the expected side of every
// boundary (n = 16, 32, 64 are exact powers); the +1e-6 reproduces that (golden: mpnet_tables.npz).
extern "C" int32_t arx_mpnet_bucket(int32_t rel, int32_t num_buckets, int32_t max_distance) {
    int n = -rel;
    const int nb = num_buckets / 2;
    int ret = 0;
    if (n < 0) { ret = nb; n = -n; }
    const int max_exact = nb / 2;
    if (n < max_exact) return ret + n;
    int v = max_exact + (int)floor(log((double)n / max_exact) / log((double)max_distance / max_exact) * (nb - max_exact) + 1e-6);
    if (v > nb - 1) v = nb - 1;
    return ret + v;
}

// ---- handle -----------------------------------------------------------------------------------
struct arx_encoder {
    arx_encoder_config cfg;
    arx_encoder_weights w;
    std::vector<arx_layer_weights> layers;
    int max_tokens, max_seqs, tok_pad;
    char* ws = nullptr;
    int64_t ws_bytes = 0;
    int32_t* cu = nullptr;
    uint16_t *x = nullptr, *x1 = nullptr, *qkv = nullptr, *ctx = nullptr, *y = nullptr, *hbuf = nullptr;
    float* bias_tbl = nullptr;
    uint16_t* tap = nullptr;
    int tap_layer = -1;
    // LayerNorm folding (ln_fold = 1): derived weights + per-row statistics, all private to the handle
    int ln_fold = 1;
    char* fold_ws = nullptr;
    std::vector<uint16_t*> wqkv_f, wfc1_f;      // per layer: gamma-folded copies (wqkv_f[0] unused)
    std::vector<float*> s_qkv, c_qkv, s_fc1, c_fc1;
    float *st1_sum = nullptr, *st1_sq = nullptr, *st2_sum = nullptr, *st2_sq = nullptr;   // row mean / rstd of y1 / y2
    float *part_s = nullptr, *part_q = nullptr;                                              // [H/64][tok_pad] partial slabs
    int variant = 89;
    int low_latency = 0;                          // arx_encoder_set_low_latency: forwards of <= ARX_SMALL_M rows take gemm_small.h
    float* small_ws = nullptr;                    // its split-K partial sums (ARX_SMALL_WS_BYTES, allocated when the option is first set)
    int attn_variant = 1;
    int attn_dev_word = 0;                        // dev: probes of attention_tr_kernel (0 in the product path)
    unsigned long long* attn_stamps = nullptr;    // dev: per-block time stamps
};

struct WsLayout {
    int64_t cu, x, x1, qkv, ctx, y, hbuf, bias, stats, total;
};
static WsLayout ws_layout(const arx_encoder_config& c, int max_tokens, int max_seqs) {
    const int64_t tp = round_up64(max_tokens, 256);
    WsLayout l;
    int64_t o = 0;
    auto take = [&](int64_t bytes) { int64_t r = o; o += round_up64(bytes, 256); return r; };
    l.cu = take((int64_t)(max_seqs + 1) * 4);
    l.x = take(tp * c.hidden * 2);
    l.x1 = take(tp * c.hidden * 2);
    l.qkv = take(tp * c.hidden * 3 * 2);
    l.ctx = take(tp * c.hidden * 2);
    l.y = take(tp * c.hidden * 2);
    l.hbuf = take(tp * c.ffn * 2);
    l.bias = take((int64_t)c.heads * ARX_BIAS_ROW * 4);
    l.stats = take(tp * 4 * 4 + 2 * (int64_t)(c.hidden / 64) * tp * 4);   // mean/rstd x2 + partial slabs
    l.total = o;
    return l;
}

static int check_cfg(const arx_encoder_config* c) {
    ARX_REQUIRE(c != nullptr, "cfg is null");
    ARX_REQUIRE(c->arch == ARX_ARCH_MPNET || c->arch == ARX_ARCH_BERT, "unknown arch %d", c->arch);
    ARX_REQUIRE(c->hidden > 0 && c->hidden % 64 == 0 && c->hidden <= 1024, "hidden=%d must be a multiple of 64, <= 1024", c->hidden);
    ARX_REQUIRE(c->heads > 0 && c->hidden % c->heads == 0, "hidden %% heads != 0");
    const int dh = c->hidden / c->heads;
    ARX_REQUIRE(dh == 32 || dh == 64, "head_dim=%d unsupported (32 or 64)", dh);
    ARX_REQUIRE(c->ffn > 0 && c->ffn % 64 == 0, "ffn=%d must be a multiple of 64", c->ffn);
    ARX_REQUIRE(c->layers > 0 && c->vocab_size > 0 && c->max_pos > 0, "bad layers/vocab/max_pos");
    ARX_REQUIRE(c->pool == ARX_POOL_MEAN || c->pool == ARX_POOL_CLS, "unknown pool %d", c->pool);
    return ARX_OK;
}

extern "C" int64_t arx_encoder_workspace_bytes(const arx_encoder_config* cfg, int32_t max_tokens, int32_t max_seqs) {
    if (check_cfg(cfg) != ARX_OK || max_tokens <= 0 || max_seqs <= 0) return -1;
    return ws_layout(*cfg, max_tokens, max_seqs).total;
}

extern "C" int32_t arx_encoder_create(const arx_encoder_config* cfg, const arx_encoder_weights* w, int32_t max_tokens,
                                      int32_t max_seqs, arx_encoder** out) {
    int rc = check_cfg(cfg);
    if (rc != ARX_OK) return rc;
    ARX_REQUIRE(w && out, "null weights/out");
    ARX_REQUIRE(max_tokens > 0 && max_seqs > 0, "max_tokens/max_seqs must be positive");
    ARX_REQUIRE(w->word_emb && w->pos_emb && w->emb_ln_g && w->emb_ln_b && w->layers, "missing embedding weights");
    ARX_REQUIRE(cfg->arch != ARX_ARCH_BERT || w->type_emb, "BERT needs type_emb");
    ARX_REQUIRE(cfg->arch != ARX_ARCH_MPNET || w->rel_bias, "MPNet needs rel_bias");
    for (int i = 0; i < cfg->layers; ++i) {
        const arx_layer_weights& L = w->layers[i];
        ARX_REQUIRE(L.w_qkv && L.b_qkv && L.w_o && L.b_o && L.ln1_g && L.ln1_b && L.w_fc1 && L.b_fc1 && L.w_fc2 &&
                        L.b_fc2 && L.ln2_g && L.ln2_b, "layer %d: missing weights", i);
    }
    arx_encoder* h = new arx_encoder();
    auto fail = [&](int code) {                        // every error exit releases whatever the handle already owns
        if (h->ws) (void)hipFree(h->ws);
        if (h->fold_ws) (void)hipFree(h->fold_ws);
        delete h;
        return code;
    };
    h->cfg = *cfg;
    h->w = *w;
    h->layers.assign(w->layers, w->layers + cfg->layers);
    h->w.layers = h->layers.data();
    h->max_tokens = max_tokens;
    h->max_seqs = max_seqs;
    h->tok_pad = (int)round_up64(max_tokens, 256);
    const char* e = getenv("ARX_GEMM_VARIANT");
    h->variant = e ? atoi(e) : 89;
    const char* av = getenv("ARX_ATTN_VARIANT");
    h->attn_variant = av ? atoi(av) : 1;
#ifdef ARX_DEV_VARIANTS
    const char* g = getenv("ARX_GEMM_GLDS");          // dev switch: 0 = register-staged reference loop
    if (g && g[0] == '0') h->variant = 4;
#endif
    const WsLayout l = ws_layout(*cfg, max_tokens, max_seqs);
    hipError_t he = hipMalloc((void**)&h->ws, l.total);
    if (he != hipSuccess) {
        arx_set_error("hipMalloc(%lld bytes workspace): %s", (long long)l.total, hipGetErrorString(he));
        h->ws = nullptr;
        return fail(ARX_ERR_HIP);
    }
    h->ws_bytes = l.total;
    // padded rows are read by GEMM tiles (results discarded): keep them finite
    (void)hipMemset(h->ws, 0, l.total);
    h->cu = (int32_t*)(h->ws + l.cu);
    h->x = (uint16_t*)(h->ws + l.x);
    h->x1 = (uint16_t*)(h->ws + l.x1);
    h->qkv = (uint16_t*)(h->ws + l.qkv);
    h->ctx = (uint16_t*)(h->ws + l.ctx);
    h->y = (uint16_t*)(h->ws + l.y);
    h->hbuf = (uint16_t*)(h->ws + l.hbuf);
    h->bias_tbl = (float*)(h->ws + l.bias);
    {
        const int64_t tp = round_up64(max_tokens, 256);
        float* st = (float*)(h->ws + l.stats);
        h->st1_sum = st; h->st1_sq = st + tp; h->st2_sum = st + 2 * tp; h->st2_sq = st + 3 * tp;
        h->part_s = st + 4 * tp; h->part_q = h->part_s + (int64_t)(cfg->hidden / 64) * tp;
    }
    const char* lf = getenv("ARX_LN_FOLD");
    h->ln_fold = lf ? atoi(lf) : 1;
    if (h->ln_fold) {
        // derived weights: per layer  W_fc1 o gamma1 (+ s, c), and for layers >= 1  W_qkv o gamma2 of the previous layer
        const int64_t H = cfg->hidden, F = cfg->ffn, Lc = cfg->layers;
        const int64_t per = round_up64(3 * H * H * 2, 256) + round_up64(F * H * 2, 256) + 2 * round_up64(3 * H * 4, 256) + 2 * round_up64(F * 4, 256);
        he = hipMalloc((void**)&h->fold_ws, per * Lc);
        if (he != hipSuccess) {
            arx_set_error("hipMalloc(folded weights): %s", hipGetErrorString(he));
            h->fold_ws = nullptr;
            return fail(ARX_ERR_HIP);
        }
        char* pws = h->fold_ws;
        auto takep = [&](int64_t bytes) { char* r = pws; pws += round_up64(bytes, 256); return r; };
        for (int i = 0; i < Lc; ++i) {
            const arx_layer_weights& L = h->layers[i];
            uint16_t* wq = (uint16_t*)takep(3 * H * H * 2); uint16_t* w1 = (uint16_t*)takep(F * H * 2);
            float* sq = (float*)takep(3 * H * 4); float* cq = (float*)takep(3 * H * 4);
            float* s1 = (float*)takep(F * 4); float* c1 = (float*)takep(F * 4);
            h->wqkv_f.push_back(wq); h->wfc1_f.push_back(w1);
            h->s_qkv.push_back(sq); h->c_qkv.push_back(cq); h->s_fc1.push_back(s1); h->c_fc1.push_back(c1);
            fold_ln_kernel<<<cdiv(F, 4), 256>>>((const uint16_t*)L.w_fc1, L.ln1_g, L.ln1_b, L.b_fc1, w1, s1, c1, (int)F, (int)H);
            if (i > 0) {
                const arx_layer_weights& P = h->layers[i - 1];
                fold_ln_kernel<<<cdiv(3 * H, 4), 256>>>((const uint16_t*)L.w_qkv, P.ln2_g, P.ln2_b, L.b_qkv, wq, sq, cq, (int)(3 * H), (int)H);
            }
        }
        he = hipDeviceSynchronize();
        if (he != hipSuccess) {
            arx_set_error("fold_ln_kernel: %s", hipGetErrorString(he));
            return fail(ARX_ERR_HIP);
        }
    }
    if (cfg->arch == ARX_ARCH_MPNET) {
        // Toeplitz bias rows, pre-multiplied by log2(e): tbl[h][d + C] = rel_bias[bucket(d)][h] * log2e
        std::vector<float> rb((size_t)cfg->rel_buckets * cfg->heads);
        he = hipMemcpy(rb.data(), w->rel_bias, rb.size() * 4, hipMemcpyDeviceToHost);
        if (he != hipSuccess) {
            arx_set_error("copy rel_bias: %s", hipGetErrorString(he));
            return fail(ARX_ERR_HIP);
        }
        std::vector<float> tbl((size_t)cfg->heads * ARX_BIAS_ROW);
        const float log2e = 1.4426950408889634f;
        for (int d = -ARX_BIAS_CENTER; d <= ARX_BIAS_CENTER; ++d) {
            const int bk = arx_mpnet_bucket(d, cfg->rel_buckets, cfg->rel_max_distance);
            for (int hd = 0; hd < cfg->heads; ++hd)
                tbl[(size_t)hd * ARX_BIAS_ROW + d + ARX_BIAS_CENTER] = rb[(size_t)bk * cfg->heads + hd] * log2e;
        }
        he = hipMemcpy(h->bias_tbl, tbl.data(), tbl.size() * 4, hipMemcpyHostToDevice);
        if (he != hipSuccess) {
            arx_set_error("upload bias table: %s", hipGetErrorString(he));
            return fail(ARX_ERR_HIP);
        }
    }
    *out = h;
    return ARX_OK;
}

extern "C" void arx_encoder_destroy(arx_encoder* h) {
    if (!h) return;
    if (h->ws) (void)hipFree(h->ws);
    if (h->tap) (void)hipFree(h->tap);
    if (h->fold_ws) (void)hipFree(h->fold_ws);
    if (h->small_ws) (void)hipFree(h->small_ws);
    delete h;
}

extern "C" int32_t arx_encoder_set_low_latency(arx_encoder* h, int32_t on) {
    ARX_REQUIRE(h, "null handle");
    if (on && !h->small_ws) ARX_HIP_CHECK(hipMalloc((void**)&h->small_ws, ARX_SMALL_WS_BYTES));
    h->low_latency = on ? 1 : 0;
    return ARX_OK;
}

extern "C" int32_t arx_encoder_set_tap(arx_encoder* h, int32_t layer) {
    ARX_REQUIRE(h, "null handle");
    ARX_REQUIRE(layer >= -1 && layer <= h->cfg.layers, "tap layer out of range");
    if (layer >= 0 && !h->tap) ARX_HIP_CHECK(hipMalloc((void**)&h->tap, (int64_t)h->tok_pad * h->cfg.hidden * 2));
    h->tap_layer = layer;
    return ARX_OK;
}

extern "C" int32_t arx_encoder_debug_hidden(arx_encoder* h, int32_t /*layer_slot*/, float* dst, int32_t n_tokens, void* stream) {
    ARX_REQUIRE(h && h->tap && dst, "no tap recorded");
    ARX_REQUIRE(n_tokens > 0 && n_tokens <= h->max_tokens, "n_tokens out of range");
    const int64_t n = (int64_t)n_tokens * h->cfg.hidden;
    bf16_to_f32_kernel<<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(h->tap, dst, n);
    ARX_HIP_CHECK(hipGetLastError());
    return ARX_OK;
}

// ---- GEMM dispatch ------------------------------------------------------------------------------
// Shipped schedules: 89 = default (persistent 4-phase kernel for K <= 1024, per-tile 4-phase kernel above), 8 / 9 = per-tile /
// persistent everywhere they apply, 13 = the 2-stage loop (64-bit offsets, any N % 8 == 0: the fallback for shapes the 4-phase
// kernels do not take).  The round-1 A/B schedules (0, 1, 2, 3, 4, 15, 33, 34) are compiled only with -DARX_DEV_VARIANTS.
template <typename Kern>
static int launch_gemm_kernel(Kern kern, int smem, int threads, int BM, int BN, const uint16_t* A, int64_t lda,
                              const uint16_t* W, int64_t ldw, int M, int N, int K, const EpiParams& ep, hipStream_t st) {
    ARX_HIP_CHECK(arx_func_smem((const void*)kern, smem));
    const int tm = cdiv(M, BM), tn = cdiv(N, BN);
    kern<<<tm * tn, threads, smem, st>>>((const bf16_t*)A, lda, (const bf16_t*)W, ldw, M, N, K, tm, tn, ep);
    ARX_HIP_CHECK(hipGetLastError());
    return ARX_OK;
}

// the 4-phase kernels address operands and the output with 32-bit element offsets
static bool fits_u32_offsets(int64_t M, int64_t lda, int64_t N, int64_t ldw, const EpiParams& ep) {
    return M * lda < (1ll << 31) && N * ldw < (1ll << 31) && M * ep.ldc < (1ll << 32) && (!ep.resid || M * ep.ldr < (1ll << 32));
}

template <int MODE>
int arx_launch_gemm(int variant, const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, int M, int N, int K,
                    const EpiParams& ep, hipStream_t st) {
    if (K % 64 != 0 || N % 8 != 0) {
        arx_set_error("gemm: K=%d must be a multiple of 64 and N=%d of 8", K, N);
        return ARX_ERR_ARG;
    }
    const bool wide = (N % 256 == 0);
    const bool wide8 = (N % 128 == 0 && N >= 256);      // the 4-phase kernels also take a half-present last n-tile (MiniLM: N = 384, 1152)
#ifdef ARX_DEV_VARIANTS
    if constexpr (MODE <= EPI_BIAS_RESID) {
        // first-version kernels (8-B stores, erff GELU): the A/B reference for the epilogue work
        if (variant == 0) {
            if (wide) return launch_gemm_kernel(gemm_bf16_kernel<256, 256, 2, 4, true, MODE>, GemmMainloop<bf16_t, 256, 256, 2, 4, true>::SMEM_BYTES, 512, 256, 256, A, lda, W, ldw, M, N, K, ep, st);
            return launch_gemm_kernel(gemm_bf16_kernel<256, 128, 4, 2, true, MODE>, GemmMainloop<bf16_t, 256, 128, 4, 2, true>::SMEM_BYTES, 512, 256, 128, A, lda, W, ldw, M, N, K, ep, st);
        }
        if (variant == 4) {      // register-staged loop (no global_load_lds)
            if (wide) return launch_gemm_kernel(gemm_bf16_kernel<256, 256, 2, 4, false, MODE>, GemmMainloop<bf16_t, 256, 256, 2, 4, false>::SMEM_BYTES, 512, 256, 256, A, lda, W, ldw, M, N, K, ep, st);
            return launch_gemm_kernel(gemm_bf16_kernel<256, 128, 4, 2, false, MODE>, GemmMainloop<bf16_t, 256, 128, 4, 2, false>::SMEM_BYTES, 512, 256, 128, A, lda, W, ldw, M, N, K, ep, st);
        }
    }
#endif
    // 71: the 2-stage loop on 128 x 128 tiles (4 waves, two blocks per CU) — the medium-batch half of the opt-in low-latency schedule:
    // between 256 and 8192 token rows a 256 x 256 grid leaves most CUs idle while each tile walks its whole K (17 us at K = 768, 70 us at
    // K = 3072 whatever the row count); four times the tiles, a quarter of the work each
    if (variant == 71)
        return launch_gemm_kernel(gemm_v0e2_kernel<128, 128, 2, 2, MODE, 67>, GemmMainloop<bf16_t, 128, 128, 2, 2, true, 67>::SMEM_BYTES, 256, 128, 128, A, lda, W, ldw, M, N, K, ep, st);
    // 89 = DEFAULT: persistent kernel for the short-K shapes (K <= 1024: QKV, O-projection, FFN-1 — the per-tile first-load latency is
    // 10-14 % of such a tile; +2..4 % measured in situ, same box), per-tile kernel for the long-K one (FFN-2: -3 % when persistent)
    if (variant == 89) variant = (K <= 1024) ? 9 : 8;
    if (variant == 9 && (K / 64) % 2 != 0) variant = 8;          // the persistent form needs an even number of k-tiles (buffer parity)
    if ((variant == 8 || variant == 9) && wide8 && !fits_u32_offsets(M, lda, N, ldw, ep)) {
        static bool said = false;                                // once per process: the shape is served, by the slower kernel
        if (!said) {
            said = true;
            fprintf(stderr, "[arx] gemm M=%d N=%d K=%d: operand or output offsets exceed 32 bits, using the 2-stage kernel (64-bit offsets)\n", M, N, K);
        }
        variant = 13;
    }
    if (variant == 9 && wide8) {                                 // persistent 4-phase schedule
        auto kern = gemm_8phase_persistent_kernel<MODE>;
        ARX_HIP_CHECK(arx_func_smem((const void*)kern, Gemm8Phase<bf16_t, 0>::SMEM_BYTES));
        const int n_cu = arx_device_cus();
        const int tm = cdiv(M, 256), tn = cdiv(N, 256);
        int grid = TileWalk::grid(tm, tn) < n_cu ? TileWalk::grid(tm, tn) : n_cu;
        grid = grid / 8 * 8 > 0 ? grid / 8 * 8 : grid;            // a multiple of 8: a block keeps its XCD across tiles
        kern<<<grid, 512, Gemm8Phase<bf16_t, 0>::SMEM_BYTES, st>>>((const bf16_t*)A, lda, (const bf16_t*)W, ldw, M, N, K, tm, tn, ep);
        ARX_HIP_CHECK(hipGetLastError());
        return ARX_OK;
    }
    if (variant == 8 && wide8) {                                 // 4-phase-per-k-tile schedule (gemm8.h)
        auto kern = gemm_8phase_kernel<MODE>;
        ARX_HIP_CHECK(arx_func_smem((const void*)kern, Gemm8Phase<bf16_t, 0>::SMEM_BYTES));
        const int tm = cdiv(M, 256), tn = cdiv(N, 256);
        kern<<<TileWalk::grid(tm, tn), 512, Gemm8Phase<bf16_t, 0>::SMEM_BYTES, st>>>((const bf16_t*)A, lda, (const bf16_t*)W, ldw, M, N, K, tm, tn, ep);
        ARX_HIP_CHECK(hipGetLastError());
        return ARX_OK;
    }
#ifdef ARX_DEV_VARIANTS
    if (variant == 34 && wide)       // ping-pong: the two waves of a SIMD half a k-step apart
        return launch_gemm_kernel(gemm_v0e2_kernel<256, 256, 2, 4, MODE, 64 + 4096>, GemmMainloop<bf16_t, 256, 256, 2, 4, true, 64 + 4096>::SMEM_BYTES, 512, 256, 256, A, lda, W, ldw, M, N, K, ep, st);
    if (variant == 33 && wide)       // A operand 3-deep (two k-steps ahead), W double-buffered, 160 KB LDS
        return launch_gemm_kernel(gemm_a3w2_kernel<256, 256, 2, 4, MODE>, 3 * 256 * 128 + 2 * 256 * 128, 512, 256, 256, A, lda, W, ldw, M, N, K, ep, st);
    if (variant == 1) {              // 2-stage loop + 16-B-store epilogue, no stagger / setprio / k rotation
        if (wide) return launch_gemm_kernel(gemm_v0e2_kernel<256, 256, 2, 4, MODE>, GemmMainloop<bf16_t, 256, 256, 2, 4, true>::SMEM_BYTES, 512, 256, 256, A, lda, W, ldw, M, N, K, ep, st);
        return launch_gemm_kernel(gemm_v0e2_kernel<256, 128, 4, 2, MODE>, GemmMainloop<bf16_t, 256, 128, 4, 2, true>::SMEM_BYTES, 512, 256, 128, A, lda, W, ldw, M, N, K, ep, st);
    }
    if (variant == 15 && wide)       // stagger + setprio, no k rotation
        return launch_gemm_kernel(gemm_v0e2_kernel<256, 256, 2, 4, MODE, 3>, GemmMainloop<bf16_t, 256, 256, 2, 4, true, 3>::SMEM_BYTES, 512, 256, 256, A, lda, W, ldw, M, N, K, ep, st);
    if (variant == 3 && wide)        // ring 256x256, 8 waves, 4 half-tile slots
        return launch_gemm_kernel(gemm_ring_kernel<256, 256, 2, 4, 4, MODE>, GemmRing<bf16_t, 256, 256, 2, 4, 4>::SMEM_BYTES, 512, 256, 256, A, lda, W, ldw, M, N, K, ep, st);
    if (variant == 2 || variant == 3)   // ring 256x128, 4 waves, 3 half-tile slots, 2 blocks/CU
        return launch_gemm_kernel(gemm_ring_kernel<256, 128, 2, 2, 3, MODE>, GemmRing<bf16_t, 256, 128, 2, 2, 3>::SMEM_BYTES, 256, 256, 128, A, lda, W, ldw, M, N, K, ep, st);
#endif
    // 13: 2-stage loop, upper-half waves issue loads mid-step, setprio around MFMA clusters, k-loop start rotated by 2*tile_n
    // (blocks sharing an A panel do not miss on the same lines at once); 64-bit offsets, any N % 8 == 0
    if (!wide) return launch_gemm_kernel(gemm_v0e2_kernel<256, 128, 4, 2, MODE, 67>, GemmMainloop<bf16_t, 256, 128, 4, 2, true, 67>::SMEM_BYTES, 512, 256, 128, A, lda, W, ldw, M, N, K, ep, st);
    return launch_gemm_kernel(gemm_v0e2_kernel<256, 256, 2, 4, MODE, 67>, GemmMainloop<bf16_t, 256, 256, 2, 4, true, 67>::SMEM_BYTES, 512, 256, 256, A, lda, W, ldw, M, N, K, ep, st);
}

template <int MODE>
static int launch_gemm(int cls, int variant, const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, int M, int N, int K,
                       const EpiParams& ep, hipStream_t st, float* small_ws = nullptr) {
    ProfScope ps(cls, st);
    if (variant == 70) return launch_gemm_small<MODE>(A, lda, W, ldw, M, N, K, ep, small_ws, st);      // small-batch path (gemm_small.h)
#ifdef ARX_DEV_VARIANTS
    static const int dev_bw = getenv("ARX_DEV_BW") ? atoi(getenv("ARX_DEV_BW")) : 0;      // dev A/B in situ: tile-walk band width
    const char* stg = getenv("ARX_DEV_STAGGER");      // "cycles[,slots[,store]]"
    if (dev_bw || stg) {
        EpiParams e2 = ep; e2.dev_bw = dev_bw;
        if (stg) { int a = 0, b = 8, c = 0; sscanf(stg, "%d,%d,%d", &a, &b, &c); e2.dev_stagger = a; e2.dev_slots = b > 0 ? b : 8; e2.dev_store = c; }
        return arx_launch_gemm<MODE>(variant, A, lda, W, ldw, M, N, K, e2, st);
    }
#endif
    return arx_launch_gemm<MODE>(variant, A, lda, W, ldw, M, N, K, ep, st);
}

// Raw linear layer C = epi(A W^T + bias [+ resid]) — building block exposed for unit tests / tuning.
extern "C" int32_t arx_gemm_bf16(const void* A, const void* W, const float* bias, const void* resid, void* C, int32_t M,
                                 int32_t N, int32_t K, int32_t mode, int32_t variant, void* stream) {
    ARX_REQUIRE(A && W && bias && C, "null pointer argument");
    ARX_REQUIRE(M > 0 && N > 0 && K > 0, "bad sizes");
    ARX_REQUIRE(mode != EPI_BIAS_RESID || resid, "mode 2 needs resid");
    EpiParams ep{(uint16_t*)C, N, bias, (const uint16_t*)resid, N};
    hipStream_t st = (hipStream_t)stream;
#ifdef ARX_DEV_VARIANTS
    // dev A/B encoding: variant = schedule + 100 * store mode (1 nt, 2 sc1) + 1000 * band width of the tile walk
    ep.dev_store = (variant / 100) % 10; ep.dev_bw = variant / 1000; variant %= 100;
    if (const char* stg = getenv("ARX_DEV_STAGGER")) { int a = 0, b = 8; sscanf(stg, "%d,%d", &a, &b); ep.dev_stagger = a; ep.dev_slots = b > 0 ? b : 8; }
#endif
#ifdef ARX_STAMP
    {   // dev build: per-tile cycle stamps of one launch, summarised on stderr
        static unsigned long long* d = nullptr;
        const int tiles = cdiv(M, 256) * cdiv(N, 256);
        if (!d) ARX_HIP_CHECK(hipMalloc(&d, (size_t)8 * 32 * 2 * 16384));
        if (getenv("ARX_STAMP_DUMP") && tiles <= 16384) {
            ep.stamps = d;
            int rc = ARX_ERR_ARG;
            if (mode == EPI_BIAS) rc = arx_launch_gemm<EPI_BIAS>(variant, (const uint16_t*)A, K, (const uint16_t*)W, K, M, N, K, ep, st);
            if (mode == EPI_BIAS_GELU) rc = arx_launch_gemm<EPI_BIAS_GELU>(variant, (const uint16_t*)A, K, (const uint16_t*)W, K, M, N, K, ep, st);
            if (mode == EPI_BIAS_RESID) rc = arx_launch_gemm<EPI_BIAS_RESID>(variant, (const uint16_t*)A, K, (const uint16_t*)W, K, M, N, K, ep, st);
            ARX_HIP_CHECK(hipStreamSynchronize(st));
            std::vector<unsigned long long> hbuf((size_t)tiles * 64);
            ARX_HIP_CHECK(hipMemcpy(hbuf.data(), d, hbuf.size() * 8, hipMemcpyDeviceToHost));
            double loop[2] = {0, 0}, epi[2] = {0, 0}, pro[2] = {0, 0}, x45[2] = {0, 0}, x67[2] = {0, 0};
            unsigned long long tmin = ~0ull, tmax = 0;
            for (int t = 0; t < tiles; ++t)
                for (int g = 0; g < 2; ++g) {
                    const unsigned long long* o = &hbuf[((size_t)t * 2 + g) * 32];
                    loop[g] += (double)(o[1] - o[3]); epi[g] += (double)(o[2] - o[1]); pro[g] += (double)(o[3] - o[0]);
                    if (variant == 9) { x45[g] += (double)(o[5] - o[4]); x67[g] += (double)(o[7] - o[6]); }      // stall in the counted wait of k-tile 0 / 1
                    else x45[g] += (double)(o[4] - o[2]);                                                        // per-tile: last store issued -> all acknowledged
                    tmin = o[0] < tmin ? o[0] : tmin; tmax = o[2] > tmax ? o[2] : tmax;
                }
            if (variant == 9) {   // persistent: tile t+256 follows tile t on the same CU -> gap between epilogue end and next loop start, tile period
                double gap = 0, period = 0; int n = 0;
                for (int t = 0; t + 256 < tiles; ++t) { const unsigned long long* a = &hbuf[(size_t)t * 64]; const unsigned long long* b = &hbuf[(size_t)(t + 256) * 64];
                    gap += (double)(b[0] - a[2]); period += (double)(b[0] - a[0]); ++n; }
                if (n) fprintf(stderr, "[stamp] persistent: g0 tile period %.0f clk, epilogue-end -> next loop start %.0f clk\n", period / n, gap / n);
            }
            if (const char* sf = getenv("ARX_STAMP_FILE")) {      // raw stamps for offline analysis: [tiles][2 groups][32] u64
                if (FILE* f = fopen(sf, "wb")) { fwrite(hbuf.data(), 8, hbuf.size(), f); fclose(f); }
            }
            if (variant == 9) {      // k-tile start to k-tile start, first 16 k-tiles, group 0 (tiles after a CU's first)
                const int nkt = K / 64 < 16 ? K / 64 : 16;
                fprintf(stderr, "[stamp] k-tile durations g0:");
                for (int k = 0; k + 1 < nkt; ++k) { double a = 0; int n = 0; for (int t = 256; t < tiles; ++t) { const unsigned long long* o = &hbuf[(size_t)t * 64]; a += (double)(o[9 + k] - o[8 + k]); ++n; } fprintf(stderr, " %.0f", a / (n ? n : 1)); }
                fprintf(stderr, "\n");
            }
            fprintf(stderr, "[stamp] %s g0 %.0f g1 %.0f clk%s g0 %.0f g1 %.0f\n", variant == 9 ? "wait at k-tile 0:" : "store drain after the epilogue:", x45[0] / tiles, x45[1] / tiles,
                    variant == 9 ? ", at k-tile 1:" : " (-)", x67[0] / tiles, x67[1] / tiles);
            fprintf(stderr, "[stamp] M=%d N=%d K=%d mode=%d tiles=%d: prologue %.0f / %.0f, loop g0 %.0f g1 %.0f clk, epilogue g0 %.0f g1 %.0f clk, span %.0f clk, tiles/CU %.1f\n",
                    M, N, K, mode, tiles, pro[0] / tiles, pro[1] / tiles, loop[0] / tiles, loop[1] / tiles, epi[0] / tiles, epi[1] / tiles, (double)(tmax - tmin), tiles / 256.0);
            return rc;
        }
    }
#endif
    ProfScope ps(ARX_K_GEMM_RAW, st);       // its own class: a raw call has whatever shape the caller chose, not FFN-1's
    if (variant == 70) {          // small-batch path alone (tests / tuning): a stream-ordered scratch for this call, released behind its kernels
        float* ws = nullptr;
        ARX_HIP_CHECK(hipMallocAsync((void**)&ws, ARX_SMALL_WS_BYTES, st));
        int rc = ARX_ERR_ARG;
        switch (mode) {
        case EPI_BIAS: rc = launch_gemm_small<EPI_BIAS>((const uint16_t*)A, K, (const uint16_t*)W, K, M, N, K, ep, ws, st); break;
        case EPI_BIAS_GELU: rc = launch_gemm_small<EPI_BIAS_GELU>((const uint16_t*)A, K, (const uint16_t*)W, K, M, N, K, ep, ws, st); break;
        case EPI_BIAS_RESID: rc = launch_gemm_small<EPI_BIAS_RESID>((const uint16_t*)A, K, (const uint16_t*)W, K, M, N, K, ep, ws, st); break;
        default: arx_set_error("unknown gemm mode %d", mode);
        }
        (void)hipFreeAsync(ws, st);
        return rc;
    }
    switch (mode) {
    case EPI_BIAS: return arx_launch_gemm<EPI_BIAS>(variant, (const uint16_t*)A, K, (const uint16_t*)W, K, M, N, K, ep, st);
    case EPI_BIAS_GELU: return arx_launch_gemm<EPI_BIAS_GELU>(variant, (const uint16_t*)A, K, (const uint16_t*)W, K, M, N, K, ep, st);
    case EPI_BIAS_RESID: return arx_launch_gemm<EPI_BIAS_RESID>(variant, (const uint16_t*)A, K, (const uint16_t*)W, K, M, N, K, ep, st);
    }
    arx_set_error("unknown gemm mode %d", mode);
    return ARX_ERR_ARG;
}

// ---- attention dispatch -------------------------------------------------------------------------
template <int DH, bool HB, int NW>
static int launch_attn_cfg(arx_encoder* h, int n_seqs, int max_len, hipStream_t st) {
    auto kern = attention_kernel<DH, HB, NW>;
    const int Lk = (max_len + 31) & ~31;
    const int smem = AttnSmem<DH>::total(Lk, HB);
    ARX_HIP_CHECK(arx_func_smem((const void*)kern, smem));
    const float scale_log2e = 1.4426950408889634f / sqrtf((float)DH);
    dim3 grid(cdiv(max_len, 32 * NW), h->cfg.heads, n_seqs);
    ProfScope ps(ARX_K_ATTENTION, st);
    kern<<<grid, NW * 64, smem, st>>>(h->qkv, h->ctx, h->cu, h->bias_tbl, h->cfg.hidden, scale_log2e);
    ARX_HIP_CHECK(hipGetLastError());
    return ARX_OK;
}

template <int DH, bool HB, int NW>
static int launch_attn_tr_cfg(arx_encoder* h, int n_seqs, int max_len, hipStream_t st) {
    auto kern = attention_tr_kernel<DH, HB, NW>;
    const int Lk = (max_len + 31) & ~31;
    const int smem = AttnSmem3<DH>::total(Lk, HB);
    ARX_HIP_CHECK(arx_func_smem((const void*)kern, smem));
    const float scale_log2e = 1.4426950408889634f / sqrtf((float)DH);
    dim3 grid(cdiv(max_len, 32 * NW), h->cfg.heads, n_seqs);
    ProfScope ps(ARX_K_ATTENTION, st);
    kern<<<grid, NW * 64, smem, st>>>(h->qkv, h->ctx, h->cu, h->bias_tbl, h->cfg.hidden, scale_log2e, h->attn_dev_word, h->attn_stamps);
    ARX_HIP_CHECK(hipGetLastError());
    return ARX_OK;
}

#ifdef ARX_DEV_VARIANTS
template <int DH, bool HB, int NSLOT>
static int launch_attn_ring_cfg(arx_encoder* h, int n_seqs, int max_len, hipStream_t st) {
    auto kern = attention_ring_kernel<DH, HB, NSLOT>;
    const int C0 = (max_len + 31) & ~31;
    const int smem = AttnRing<DH, NSLOT>::total(C0, HB);
    ARX_HIP_CHECK(arx_func_smem((const void*)kern, smem));
    int G = arx_device_cus() / h->cfg.heads;                      // one persistent 8-wave block per CU (the ring takes most of its LDS)
    G = G < 1 ? 1 : (G > n_seqs ? n_seqs : G);
    const float scale_log2e = 1.4426950408889634f / sqrtf((float)DH);
    ProfScope ps(ARX_K_ATTENTION, st);
    kern<<<G * h->cfg.heads, 512, smem, st>>>(h->qkv, h->ctx, h->cu, h->bias_tbl, h->cfg.hidden, n_seqs, C0, scale_log2e);
    ARX_HIP_CHECK(hipGetLastError());
    return ARX_OK;
}

template <bool HB>
static int launch_attn_ring16_cfg(arx_encoder* h, int n_seqs, int max_len, hipStream_t st) {
    auto kern = attention_ring16_kernel<HB>;
    const int C0 = (max_len + 31) & ~31;
    const int smem = AttnRing16<8>::total(C0, HB);
    ARX_HIP_CHECK(arx_func_smem((const void*)kern, smem));
    int G = arx_device_cus() / h->cfg.heads;                      // one persistent 16-wave block per CU
    G = G < 1 ? 1 : (G > n_seqs ? n_seqs : G);
    const float scale_log2e = 1.4426950408889634f / sqrtf(64.0f);
    ProfScope ps(ARX_K_ATTENTION, st);
    kern<<<G * h->cfg.heads, 1024, smem, st>>>(h->qkv, h->ctx, h->cu, h->bias_tbl, h->cfg.hidden, n_seqs, C0, scale_log2e);
    ARX_HIP_CHECK(hipGetLastError());
    return ARX_OK;
}

#endif

static int launch_attn(arx_encoder* h, int n_seqs, int max_len, hipStream_t st) {
    const int dh = h->cfg.hidden / h->cfg.heads;
    const bool hb = h->cfg.arch == ARX_ARCH_MPNET;
#ifdef ARX_DEV_VARIANTS
    if (h->attn_variant == 4 && dh == 64 && max_len > 128 && max_len <= 256)      // 16-wave ring kernel (encoder_kernels.h "attention v4")
        return hb ? launch_attn_ring16_cfg<true>(h, n_seqs, max_len, st) : launch_attn_ring16_cfg<false>(h, n_seqs, max_len, st);
    if (h->attn_variant == 2 && max_len > 128 && max_len <= 256) {      // streaming ring kernel (encoder_kernels.h "attention v3"), one block per CU
        if (dh == 64) return hb ? launch_attn_ring_cfg<64, true, 8>(h, n_seqs, max_len, st) : launch_attn_ring_cfg<64, false, 8>(h, n_seqs, max_len, st);
        return hb ? launch_attn_ring_cfg<32, true, 8>(h, n_seqs, max_len, st) : launch_attn_ring_cfg<32, false, 8>(h, n_seqs, max_len, st);
    }
#endif
    if (h->attn_variant >= 1) {      // transposing-read kernel; 8 waves cover 256 queries, 4 waves for short batches
        const bool w8 = max_len > 128;
        if (dh == 64) {
            if (hb) return w8 ? launch_attn_tr_cfg<64, true, 8>(h, n_seqs, max_len, st) : launch_attn_tr_cfg<64, true, 4>(h, n_seqs, max_len, st);
            return w8 ? launch_attn_tr_cfg<64, false, 8>(h, n_seqs, max_len, st) : launch_attn_tr_cfg<64, false, 4>(h, n_seqs, max_len, st);
        }
        if (hb) return w8 ? launch_attn_tr_cfg<32, true, 8>(h, n_seqs, max_len, st) : launch_attn_tr_cfg<32, true, 4>(h, n_seqs, max_len, st);
        return w8 ? launch_attn_tr_cfg<32, false, 8>(h, n_seqs, max_len, st) : launch_attn_tr_cfg<32, false, 4>(h, n_seqs, max_len, st);
    }
    const bool big = max_len > 256;     // > 80 KB of LDS per block anyway: use 8 waves per block
    if (dh == 64) {
        if (hb) return big ? launch_attn_cfg<64, true, 8>(h, n_seqs, max_len, st) : launch_attn_cfg<64, true, 4>(h, n_seqs, max_len, st);
        return big ? launch_attn_cfg<64, false, 8>(h, n_seqs, max_len, st) : launch_attn_cfg<64, false, 4>(h, n_seqs, max_len, st);
    }
    if (hb) return big ? launch_attn_cfg<32, true, 8>(h, n_seqs, max_len, st) : launch_attn_cfg<32, true, 4>(h, n_seqs, max_len, st);
    return big ? launch_attn_cfg<32, false, 8>(h, n_seqs, max_len, st) : launch_attn_cfg<32, false, 4>(h, n_seqs, max_len, st);
}

// Attention block alone (parity tap for the fused kernel): ctx[T,H] = softmax(q k^T / sqrt(dh) + bias, key mask) v
// on caller-provided packed qkv [T, 3H] bf16; lens device int32 [n_seqs].
extern "C" int32_t arx_encoder_attention(arx_encoder* h, const void* qkv, const int32_t* lens, int32_t n_seqs, int32_t max_len,
                                         void* ctx, void* stream) {
    ARX_REQUIRE(h && qkv && lens && ctx, "null pointer argument");
    ARX_REQUIRE(n_seqs > 0 && n_seqs <= h->max_seqs && max_len > 0 && max_len <= 512, "bad n_seqs/max_len");
    hipStream_t st = (hipStream_t)stream;
    scan_lens_kernel<<<1, 256, 0, st>>>(lens, h->cu, n_seqs);
    ARX_HIP_CHECK(hipGetLastError());
    uint16_t* q0 = h->qkv; uint16_t* c0 = h->ctx;
    h->qkv = (uint16_t*)qkv; h->ctx = (uint16_t*)ctx;
    const int rc = launch_attn(h, n_seqs, max_len, st);
    h->qkv = q0; h->ctx = c0;
    return rc;
}

static void launch_ln_finalize(int T, int nparts, const float* ps, const float* pq, int64_t ld, const int32_t* n_rows, float inv_h,
                               float eps, float* mean, float* rstd, hipStream_t st) {
    const int grid = cdiv(T, 256);
    switch (nparts) {      // same order of additions whatever the instantiation
    case 6: ln_finalize_kernel<6><<<grid, 256, 0, st>>>(ps, pq, ld, nparts, n_rows, inv_h, eps, mean, rstd); break;
    case 12: ln_finalize_kernel<12><<<grid, 256, 0, st>>>(ps, pq, ld, nparts, n_rows, inv_h, eps, mean, rstd); break;
    case 16: ln_finalize_kernel<16><<<grid, 256, 0, st>>>(ps, pq, ld, nparts, n_rows, inv_h, eps, mean, rstd); break;
    default: ln_finalize_kernel<0><<<grid, 256, 0, st>>>(ps, pq, ld, nparts, n_rows, inv_h, eps, mean, rstd); break;
    }
}

// ---- forward ------------------------------------------------------------------------------------
extern "C" int32_t arx_encoder_forward(arx_encoder* h, const int32_t* ids, int32_t seq_stride, const int32_t* lens,
                                       int32_t n_seqs, int32_t max_len, int32_t total_tokens, float* out_f32,
                                       int64_t out_stride, void* out_f16, int64_t out16_stride, int32_t normalize,
                                       void* stream) {
    ARX_REQUIRE(h && ids && lens, "null handle/ids/lens");
    ARX_REQUIRE(out_f32 || out_f16, "no output buffer");
    ARX_REQUIRE(n_seqs > 0 && max_len > 0 && max_len <= 512, "n_seqs=%d max_len=%d (max_len must be in 1..512)", n_seqs, max_len);
    ARX_REQUIRE(seq_stride >= max_len, "seq_stride < max_len");
    ARX_REQUIRE(total_tokens > 0 && (int64_t)total_tokens <= (int64_t)n_seqs * max_len, "total_tokens out of range");
    if (n_seqs > h->max_seqs || total_tokens > h->max_tokens) {
        arx_set_error("batch (%d seqs, %d tokens) exceeds handle capacity (%d, %d)", n_seqs, total_tokens, h->max_seqs, h->max_tokens);
        return ARX_ERR_CAPACITY;
    }
    ARX_REQUIRE(!out_f32 || out_stride >= h->cfg.hidden, "out_stride < hidden");
    ARX_REQUIRE(!out_f16 || out16_stride >= h->cfg.hidden, "out16_stride < hidden");
    hipStream_t st = (hipStream_t)stream;
    const arx_encoder_config& c = h->cfg;
    const int H = c.hidden, F = c.ffn, T = total_tokens;
    int rc;

    scan_lens_kernel<<<1, 256, 0, st>>>(lens, h->cu, n_seqs);
    ARX_HIP_CHECK(hipGetLastError());
    {
        ProfScope ps(ARX_K_EMBED, st);
        dim3 grid(cdiv(max_len, 4), n_seqs);
        if (c.arch == ARX_ARCH_MPNET)
            embed_ln_kernel<ARX_ARCH_MPNET><<<grid, 256, 0, st>>>(ids, seq_stride, lens, h->cu, h->w.word_emb, h->w.pos_emb, nullptr,
                                                                  h->w.emb_ln_g, h->w.emb_ln_b, h->x, H, c.vocab_size, c.max_pos, c.pad_id, c.ln_eps);
        else
            embed_ln_kernel<ARX_ARCH_BERT><<<grid, 256, 0, st>>>(ids, seq_stride, lens, h->cu, h->w.word_emb, h->w.pos_emb, h->w.type_emb,
                                                                 h->w.emb_ln_g, h->w.emb_ln_b, h->x, H, c.vocab_size, c.max_pos, c.pad_id, c.ln_eps);
        ARX_HIP_CHECK(hipGetLastError());
    }
    const int ln_grid = cdiv(T, 4);
    const int64_t tp = round_up64(h->max_tokens, 256);
    if (!h->ln_fold) {
        // ---- reference schedule: explicit LayerNorm kernels --------------------------------------------
        auto tap = [&](int layer) -> int {
            if (h->tap_layer == layer && h->tap) ARX_HIP_CHECK(hipMemcpyAsync(h->tap, h->x, (int64_t)T * H * 2, hipMemcpyDeviceToDevice, st));
            return ARX_OK;
        };
        if ((rc = tap(0)) != ARX_OK) return rc;
        for (int li = 0; li < c.layers; ++li) {
            const arx_layer_weights& L = h->layers[li];
            EpiParams ep;
            ep = EpiParams{h->qkv, 3 * (int64_t)H, L.b_qkv, nullptr, 0};
            if ((rc = launch_gemm<EPI_BIAS>(ARX_K_GEMM_QKV, h->variant, h->x, H, (const uint16_t*)L.w_qkv, H, T, 3 * H, H, ep, st)) != ARX_OK) return rc;
            if ((rc = launch_attn(h, n_seqs, max_len, st)) != ARX_OK) return rc;
            ep = EpiParams{h->y, H, L.b_o, h->x, H};
            if ((rc = launch_gemm<EPI_BIAS_RESID>(ARX_K_GEMM_OPROJ, h->variant, h->ctx, H, (const uint16_t*)L.w_o, H, T, H, H, ep, st)) != ARX_OK) return rc;
            { ProfScope ps(ARX_K_LAYERNORM, st);
              layernorm_kernel<<<ln_grid, 256, 0, st>>>(h->y, h->x1, L.ln1_g, L.ln1_b, h->cu + n_seqs, H, c.ln_eps); }
            ARX_HIP_CHECK(hipGetLastError());
            ep = EpiParams{h->hbuf, F, L.b_fc1, nullptr, 0};
            if ((rc = launch_gemm<EPI_BIAS_GELU>(ARX_K_GEMM_FC1, h->variant, h->x1, H, (const uint16_t*)L.w_fc1, H, T, F, H, ep, st)) != ARX_OK) return rc;
            ep = EpiParams{h->y, H, L.b_fc2, h->x1, H};
            if ((rc = launch_gemm<EPI_BIAS_RESID>(ARX_K_GEMM_FC2, h->variant, h->hbuf, F, (const uint16_t*)L.w_fc2, F, T, H, F, ep, st)) != ARX_OK) return rc;
            { ProfScope ps(ARX_K_LAYERNORM, st);
              layernorm_kernel<<<ln_grid, 256, 0, st>>>(h->y, h->x, L.ln2_g, L.ln2_b, h->cu + n_seqs, H, c.ln_eps); }
            ARX_HIP_CHECK(hipGetLastError());
            if ((rc = tap(li + 1)) != ARX_OK) return rc;
        }
        ProfScope pps(ARX_K_POOL, st);
        pool_norm_kernel<<<n_seqs, 256, 0, st>>>(h->x, h->cu, H, c.pool, normalize, out_f32, out_stride, (f16_t*)out_f16, out16_stride,
                                                 nullptr, nullptr, nullptr, nullptr, 0.f);
        ARX_HIP_CHECK(hipGetLastError());
        return ARX_OK;
    }

    // ---- default schedule: no LayerNorm kernel inside the layer loop (gemm.h "LayerNorm is never run as a kernel") ----
    //   x  : layer 0 -> embeddings (already normalised);  afterwards the PRE-LN2 sum y2 of the previous layer
    //   x1 : PRE-LN1 sum y1 of the current layer;  (st1, st2) : row sums / sums of squares of y1 / y2
    const float inv_h = 1.0f / (float)H;
    // opt-in low-latency schedule: a query batch takes split-K wave tiles (<= 256 rows) or 128 x 128 tiles (<= 8192 rows) instead of 256 x 256 tiles
    const int gv = !h->low_latency ? h->variant : (T <= ARX_SMALL_M ? 70 : (T <= ARX_MEDIUM_M ? 71 : h->variant));
    if (h->tap_layer == 0 && h->tap) ARX_HIP_CHECK(hipMemcpyAsync(h->tap, h->x, (int64_t)T * H * 2, hipMemcpyDeviceToDevice, st));
    for (int li = 0; li < c.layers; ++li) {
        const arx_layer_weights& L = h->layers[li];
        EpiParams ep;
        if (li == 0) {
            ep = EpiParams{h->qkv, 3 * (int64_t)H, L.b_qkv, nullptr, 0};
            if ((rc = launch_gemm<EPI_BIAS>(ARX_K_GEMM_QKV, gv, h->x, H, (const uint16_t*)L.w_qkv, H, T, 3 * H, H, ep, st, h->small_ws)) != ARX_OK) return rc;
        } else {
            ep = EpiParams{h->qkv, 3 * (int64_t)H, h->c_qkv[li], nullptr, 0};
            ep.a_sum = h->st2_sum; ep.a_sq = h->st2_sq; ep.s_vec = h->s_qkv[li]; ep.inv_h = inv_h; ep.eps = c.ln_eps;
            if ((rc = launch_gemm<EPI_LN_BIAS>(ARX_K_GEMM_QKV, gv, h->x, H, h->wqkv_f[li], H, T, 3 * H, H, ep, st, h->small_ws)) != ARX_OK) return rc;
        }
        if ((rc = launch_attn(h, n_seqs, max_len, st)) != ARX_OK) return rc;
        // y1 = ctx Wo^T + b + (x0 | LN2_prev(y2))  -> x1, statistics -> st1
        ep = EpiParams{h->x1, H, L.b_o, h->x, H};
        ep.o_sum = h->part_s; ep.o_sq = h->part_q; ep.o_ld = tp; ep.inv_h = inv_h; ep.eps = c.ln_eps;
        ep.fin_mean = h->st1_sum; ep.fin_rstd = h->st1_sq;          // (small-batch path: statistics in final form, no ln_finalize)
        if (li == 0) {
            if ((rc = launch_gemm<EPI_RESID_STATS>(ARX_K_GEMM_OPROJ, gv, h->ctx, H, (const uint16_t*)L.w_o, H, T, H, H, ep, st, h->small_ws)) != ARX_OK) return rc;
        } else {
            const arx_layer_weights& P = h->layers[li - 1];
            ep.r_sum = h->st2_sum; ep.r_sq = h->st2_sq; ep.r_gamma = P.ln2_g; ep.r_beta = P.ln2_b;
            if ((rc = launch_gemm<EPI_LNRESID_STATS>(ARX_K_GEMM_OPROJ, gv, h->ctx, H, (const uint16_t*)L.w_o, H, T, H, H, ep, st, h->small_ws)) != ARX_OK) return rc;
        }
        if (gv != 70) { ProfScope ps(ARX_K_LAYERNORM, st);
          launch_ln_finalize(T, H / 64, h->part_s, h->part_q, tp, h->cu + n_seqs, inv_h, c.ln_eps, h->st1_sum, h->st1_sq, st); }
        ARX_HIP_CHECK(hipGetLastError());
        // hbuf = gelu(LN1(y1) W1^T + b1)   (gamma1 folded into W1')
        ep = EpiParams{h->hbuf, F, h->c_fc1[li], nullptr, 0};
        ep.a_sum = h->st1_sum; ep.a_sq = h->st1_sq; ep.s_vec = h->s_fc1[li]; ep.inv_h = inv_h; ep.eps = c.ln_eps;
        if ((rc = launch_gemm<EPI_LN_BIAS_GELU>(ARX_K_GEMM_FC1, gv, h->x1, H, h->wfc1_f[li], H, T, F, H, ep, st, h->small_ws)) != ARX_OK) return rc;
        // y2 = hbuf W2^T + b2 + LN1(y1)  -> x, statistics -> st2
        ep = EpiParams{h->x, H, L.b_fc2, h->x1, H};
        ep.r_sum = h->st1_sum; ep.r_sq = h->st1_sq; ep.r_gamma = L.ln1_g; ep.r_beta = L.ln1_b;
        ep.o_sum = h->part_s; ep.o_sq = h->part_q; ep.o_ld = tp; ep.inv_h = inv_h; ep.eps = c.ln_eps;
        ep.fin_mean = h->st2_sum; ep.fin_rstd = h->st2_sq;
        if ((rc = launch_gemm<EPI_LNRESID_STATS>(ARX_K_GEMM_FC2, gv, h->hbuf, F, (const uint16_t*)L.w_fc2, F, T, H, F, ep, st, h->small_ws)) != ARX_OK) return rc;
        if (gv != 70) { ProfScope ps(ARX_K_LAYERNORM, st);
          launch_ln_finalize(T, H / 64, h->part_s, h->part_q, tp, h->cu + n_seqs, inv_h, c.ln_eps, h->st2_sum, h->st2_sq, st); }
        ARX_HIP_CHECK(hipGetLastError());
        if (h->tap_layer == li + 1 && h->tap) {       // parity tap: materialise LN2(y2) with the stand-alone kernel
            layernorm_kernel<<<ln_grid, 256, 0, st>>>(h->x, h->tap, L.ln2_g, L.ln2_b, h->cu + n_seqs, H, c.ln_eps);
            ARX_HIP_CHECK(hipGetLastError());
        }
    }
    {
        const arx_layer_weights& Ll = h->layers[c.layers - 1];
        ProfScope pps(ARX_K_POOL, st);
        pool_norm_kernel<<<n_seqs, 256, 0, st>>>(h->x, h->cu, H, c.pool, normalize, out_f32, out_stride, (f16_t*)out_f16, out16_stride,
                                                 h->st2_sum, h->st2_sq, Ll.ln2_g, Ll.ln2_b, c.ln_eps);
        ARX_HIP_CHECK(hipGetLastError());
    }
    return ARX_OK;
}

#ifdef ARX_DEV_VARIANTS
// dev build only (not in include/arx.h): probe word and per-block stamp buffer (4 x u64 per block) of attention_tr_kernel
extern "C" int32_t arx_dev_attn_set(arx_encoder* h, int32_t word, void* stamps) {
    h->attn_dev_word = word;
    h->attn_stamps = (unsigned long long*)stamps;
    return ARX_OK;
}
#endif

extern "C" int32_t arx_build_info(void) {
#ifdef ARX_DEV_VARIANTS
    return 1;
#else
    return 0;
#endif
}

extern "C" int32_t arx_adjacent_cosine(const float* emb, int64_t ld, int32_t n, int32_t dim, float* out, void* stream) {
    ARX_REQUIRE(n >= 0 && dim > 0 && dim % 4 == 0 && ld >= dim, "bad sizes (dim must be a multiple of 4)");
    if (n < 2) return ARX_OK;   // no pairs: out may be an empty (null) buffer
    ARX_REQUIRE(emb && out, "null pointer argument");
    adjacent_cosine_kernel<<<cdiv(n - 1, 4), 256, 0, (hipStream_t)stream>>>(emb, ld, n, dim, out);
    ARX_HIP_CHECK(hipGetLastError());
    return ARX_OK;
}

extern "C" int32_t arx_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
    ARX_REQUIRE(src && dst && n >= 0, "bad args");
    if (n == 0) return ARX_OK;
    f32_to_bf16_kernel<<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(src, (uint16_t*)dst, n);
    ARX_HIP_CHECK(hipGetLastError());
    return ARX_OK;
}
