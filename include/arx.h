/*
 * arx.h — C ABI of the MI355X-native embed + retrieve hot path (libarx_hip.so).
 *
 * This is the replacement surface for the arithmetic the reference reaches through
 * `SentenceTransformer.encode(...)`
 *   /root/reference/4-embed/generation/generate_embeddings_parallel.py:146-153 (batch encode)
 *   /root/reference/4-embed/generation/generate_embeddings_parallel.py:160-165 (per-item retry)
 *   /root/reference/3-chunks/pipeline/src/processors/text_processor.py:1383-1396 (semantic chunker)
 * plus the cosine top-k step the reference configures but never implements
 *   /root/reference/3-chunks/pipeline/config.yaml:63-64 (`retrieval.top_k: 10`)
 *   /root/reference/3-chunks/pipeline/src/processors/text_processor.py:1601-1605 (cosine helper).
 * The reference has no FFI of its own (it is pure Python); INTEGRATION.md shows the ctypes stub a
 * maintainer would add at those call sites.
 *
 * Conventions
 *   - plain C types only; every pointer marked "device" is a HIP device pointer owned by the CALLER
 *     (e.g. a PyTorch-ROCm tensor's data_ptr); the library never allocates outputs.
 *   - an opaque handle owns only its private workspace (create/destroy).
 *   - every launch goes on the caller's hipStream_t (passed as void*); no hidden synchronisation.
 *   - return 0 on success, negative on error; arx_last_error() gives the thread-local message.
 *   - one host thread per handle.  The search entry points keep NO process-wide state: everything a call depends on is in its
 *     arguments (arx_topk_options) and its workspace, so different host threads may search different (or the same) shards at once,
 *     each with its own workspace and stream.  The only process-wide state in the library is the opt-in arx_prof_* timing facility.
 */
#ifndef ARX_H
#define ARX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ARX_VERSION 111            /* 0.1.1: search policy per call (arx_topk_options), no process-wide search state; 111: the int8 index is centred (layout + 4 dim bytes) */

#define ARX_OK            0
#define ARX_ERR_ARG      -1        /* bad argument / unsupported shape */
#define ARX_ERR_HIP      -2        /* HIP runtime error */
#define ARX_ERR_CAPACITY -3        /* batch exceeds the handle's workspace */

#define ARX_ARCH_MPNET 0           /* relative-position bias, pad-aware position ids */
#define ARX_ARCH_BERT  1           /* absolute positions + token type 0 */
#define ARX_POOL_MEAN  0
#define ARX_POOL_CLS   1

typedef struct {
    int32_t arch;                  /* ARX_ARCH_* */
    int32_t vocab_size;
    int32_t hidden;                /* H: multiple of 64 */
    int32_t layers;
    int32_t heads;                 /* H / heads must be 32 or 64 */
    int32_t ffn;                   /* F: multiple of 64 */
    int32_t max_pos;               /* rows of the position table */
    int32_t pool;                  /* ARX_POOL_* */
    int32_t pad_id;                /* MPNet 1, BERT 0 */
    int32_t rel_buckets;           /* MPNet: 32 */
    int32_t rel_max_distance;      /* MPNet: 128 */
    float   ln_eps;
} arx_encoder_config;

/* All weight pointers: device memory, caller-owned, must outlive the handle.
 * Matrices are nn.Linear layout [out, in] row-major, bf16 (uint16 storage).  Vectors are f32. */
typedef struct {
    const void*  w_qkv;            /* [3H, H] bf16: rows 0..H-1 = q, H..2H-1 = k, 2H..3H-1 = v */
    const float* b_qkv;            /* [3H] */
    const void*  w_o;              /* [H, H] bf16 */
    const float* b_o;              /* [H] */
    const float* ln1_g; const float* ln1_b;   /* attention output LayerNorm [H] */
    const void*  w_fc1;            /* [F, H] bf16 */
    const float* b_fc1;            /* [F] */
    const void*  w_fc2;            /* [H, F] bf16 */
    const float* b_fc2;            /* [H] */
    const float* ln2_g; const float* ln2_b;   /* output LayerNorm [H] */
} arx_layer_weights;

typedef struct {
    const float* word_emb;         /* [vocab, H] f32 */
    const float* pos_emb;          /* [max_pos, H] f32 */
    const float* type_emb;         /* [H] f32: BERT token_type row 0; NULL for MPNet */
    const float* emb_ln_g; const float* emb_ln_b;   /* [H] */
    const float* rel_bias;         /* [rel_buckets, heads] f32 (HF encoder.relative_attention_bias.weight); NULL for BERT */
    const arx_layer_weights* layers;   /* host array of `layers` entries */
} arx_encoder_weights;

typedef struct arx_encoder arx_encoder;

int32_t     arx_version(void);
const char* arx_last_error(void);

/* MPNet relative-position bucket of (key_pos - query_pos) — host function, exported so the table
 * the kernels consume can be checked against the golden vectors without a GPU. */
int32_t arx_mpnet_bucket(int32_t relative_position, int32_t num_buckets, int32_t max_distance);

/* Workspace bytes a handle for (cfg, max_tokens, max_seqs) allocates. */
int64_t arx_encoder_workspace_bytes(const arx_encoder_config* cfg, int32_t max_tokens, int32_t max_seqs);

/* Build a handle: validates shapes, allocates the private workspace (activations for up to
 * max_tokens packed tokens / max_seqs sequences), precomputes the relative-position table. */
int32_t arx_encoder_create(const arx_encoder_config* cfg, const arx_encoder_weights* w,
                           int32_t max_tokens, int32_t max_seqs, arx_encoder** out);
void    arx_encoder_destroy(arx_encoder* h);

/* Encode one batch of token sequences -> unit-norm sentence embeddings.
 *   ids   device int32 [n_seqs, seq_stride]: right-padded token ids (pad beyond lens[i] is ignored)
 *   lens  device int32 [n_seqs]: valid tokens per row (0 <= lens[i] <= max_len <= 512)
 *   total_tokens  host-side upper bound on sum(lens) (exact sum is best; n_seqs*max_len always valid)
 *   out_f32  device float  [n_seqs, out_stride] or NULL
 *   out_f16  device fp16   [n_seqs, out16_stride] or NULL   (corpus-shard row, written in place)
 *   normalize  1: x / max(||x||, 1e-12)   0: raw pooled vector
 * Semantics = SentenceTransformer.encode(..., normalize_embeddings=True) on the same token ids:
 * encoder forward, masked mean (or CLS) pool, L2 normalise.  Rows with lens[i] == 0 give zeros. */
int32_t arx_encoder_forward(arx_encoder* h, const int32_t* ids, int32_t seq_stride,
                            const int32_t* lens, int32_t n_seqs, int32_t max_len, int32_t total_tokens,
                            float* out_f32, int64_t out_stride,
                            void* out_f16, int64_t out16_stride,
                            int32_t normalize, void* stream);

/* The fused attention block alone, on caller-provided packed qkv [sum(lens), 3H] bf16 -> ctx [sum(lens), H] bf16
 * (parity tap: lets a test drive the online-softmax rescale path with crafted scores). */
int32_t arx_encoder_attention(arx_encoder* h, const void* qkv, const int32_t* lens, int32_t n_seqs, int32_t max_len,
                              void* ctx, void* stream);

/* Debug / parity tap: copy the packed hidden state after `layer` (0 = embeddings, L = last) as f32
 * [total_tokens, H] into dst (device).  Valid after a forward on the same stream. */
int32_t arx_encoder_debug_hidden(arx_encoder* h, int32_t layer_slot, float* dst, int32_t n_tokens, void* stream);

/* Opt-in small-batch schedule for QUERY batches (no reference counterpart: the reference encodes its queries with the same
 * `model.encode` as the corpus, GEN:146-153).  With on != 0, a forward of at most 256 packed token rows runs its linear layers as
 * split-K wave tiles (csrc/gemm_small.h: every CU takes part and each weight byte is read once) instead of 256 x 256 tiles that
 * leave all but a few CUs idle; forwards of up to 8 192 rows take 128 x 128 tiles; longer forwards are unaffected.  Rows agree with the default schedule to rounding (another
 * summation order), not bit for bit: leave it off for corpus rows if their bits must not depend on the batch they were in.
 * Allocates the handle's 64-MiB partial-sum workspace on first use (freed by arx_encoder_destroy). */
int32_t arx_encoder_set_low_latency(arx_encoder* h, int32_t on);

/* Ask the next forward() to snapshot the hidden state after `layer` (-1 = off). */
int32_t arx_encoder_set_tap(arx_encoder* h, int32_t layer);

/* Build flags of the loaded library: bit 0 = built with -DARX_DEV_VARIANTS (the A/B schedules of DESIGN.md's negative-result
 * tables are compiled in and selectable through ARX_GEMM_VARIANT / ARX_ATTN_VARIANT); 0 for the shipped build. */
int32_t arx_build_info(void);

/* ---- brute-force cosine top-k over an HBM-resident fp16 shard ------------------------------------
 *   corpus  device fp16 [n_rows, dim] row-major, dim % 64 == 0.  Scores are DOT PRODUCTS (cosine when rows and queries are unit
 *           vectors, as the encoder writes them); the answer is the exact top-k of those for rows of ANY norm, provided
 *           arx_topk_options.max_row_norm bounds the rows' L2 norms (default: unit rows; arx_rows_max_norm_f16 measures it).
 *   queries device fp16 [n_queries, dim]
 *   out_scores device f32 [n_queries, k]; out_ids device int64 [n_queries, k] = local row + idx_base
 *   order: score descending, ties -> lower row index; if k > n_rows the tail is (-inf, -1)
 *   ws: device workspace of arx_topk_workspace_bytes(...) bytes, caller-owned.  k <= 32. */
int64_t arx_topk_workspace_bytes(int64_t n_rows, int32_t n_queries, int32_t dim, int32_t k);
int32_t arx_topk_search(const void* corpus, int64_t n_rows, const void* queries, int32_t n_queries,
                        int32_t dim, int32_t k, float* out_scores, int64_t* out_ids, int64_t idx_base,
                        void* ws, int64_t ws_bytes, void* stream);

/* Optional int8 PRE-FILTER (no reference counterpart; same exact answers, fewer bytes).  arx_topk_build_i8 writes a second, int8
 * representation of the shard (int8 values of the rows MINUS a vector mu the build samples from them — their mean, so that rows sharing a large
 * common component are told apart by what distinguishes them —, row-wise scale, the row's L1 norm, mu: dim + 8 bytes per row + 4 dim, caller-owned device buffer of
 * arx_topk_i8_index_bytes(...) bytes); arx_topk_search_i8 then runs its first pass over THAT — half the bytes where the pass is
 * HBM-bound, twice the MFMA rate where it is matrix-bound — computing for every (query, 64-row group) a rigorous UPPER BOUND on the
 * true fp16 score of the group's rows (quantisation error bounded analytically, csrc/search_pass_a.h).  Selection, the fp32 rescoring
 * of the fp16 rows and the exactness certificate are those of arx_topk_search, so out_scores / out_ids are the same exact top-k
 * (the certificate's exhaustive-by-threshold step absorbs the bound's slack: a few hundred 64-row groups per query on unit rows).
 * dim % 128 == 0, dim <= 1024.  `corpus` is still needed (the rescoring reads it).  The int8 pass needs the LARGER workspace of
 * arx_topk_workspace_bytes_i8 (candidate lists, a second word per (query, group)); the fp16 pass does not pay for it. */
int64_t arx_topk_i8_index_bytes(int64_t n_rows, int32_t dim);
int64_t arx_topk_workspace_bytes_i8(int64_t n_rows, int32_t n_queries, int32_t dim, int32_t k);
int32_t arx_topk_build_i8(const void* corpus, int64_t n_rows, int32_t dim, void* index_i8, void* stream);
/* {|mu|, max |t_c|, max sum |c'_i m^_i|} of a built index -> host_out[3] (synchronises the stream): |mu| is what a caller needs to decide on
 * ARX_TOPK_I8_CENTRE_QUERY. */
int32_t arx_topk_i8_index_info(const void* index_i8, int64_t n_rows, int32_t dim, float* host_out, void* stream);
int32_t arx_topk_search_i8(const void* corpus, const void* index_i8, int64_t n_rows, const void* queries, int32_t n_queries,
                           int32_t dim, int32_t k, float* out_scores, int64_t* out_ids, int64_t idx_base,
                           void* ws, int64_t ws_bytes, void* stream);

/* Per-CALL policy of a search (nothing here is remembered by the library: two indices with different policies can be searched from two
 * host threads at once).  Zero-initialise, set struct_bytes = sizeof(arx_topk_options), fill what differs from the defaults. */
#define ARX_TOPK_NO_PERSISTENT 1   /* flags: take the per-tile pass-A kernel even where the persistent one applies (A/B measurements) */
#define ARX_TOPK_NO_SINGLE_ROW_TAIL 8 /* flags: take the select + rescore kernel pair (all 64 rows of each selected group) where the library would take the
                                      single-kernel tail that rescoring only the arg-max 4-row block (fp16 pass: batches of <= 256 queries, k <= 10, shards
                                      of <= 1 M rows) or the arg-max row (int8 pipeline's first step) of each selected group — A/B measurements, tests */
#define ARX_TOPK_I8_CENTRE_QUERY 16 /* flags (int8 pass): quantise the QUERY minus its component along the index's mean direction as well, the rank-one term
                                      that leaves added exactly in pass A (one more instruction per value of its epilogue).  For indexes whose rows share a large
                                      common component (arx_topk_i8_index_info: |mu|^2 = the mean pairwise cosine of unit rows); same exact answers either way */
#define ARX_TOPK_SCAN_ONLY     2   /* flags: run only pass A (the scan of the shard: every CU, HBM-bound) and leave its result in the workspace */
#define ARX_TOPK_TAIL_ONLY     4   /* flags: run only what follows pass A (select, exact rescoring, certificate) on a workspace a SCAN_ONLY call
                                      with the same arguments filled; the caller orders the two calls (possibly on two streams with different
                                      CU masks: the pipelined search).  Split calls take at most 1 024 queries. */
typedef struct {
    int32_t struct_bytes;          /* sizeof(arx_topk_options) of the caller's header */
    int32_t i8_max_queries;        /* with an int8 index: internal query batches of more than this many queries take the fp16 first pass;
                                      0 = library default (every batch size: the measured crossover), negative = never int8 */
    float   max_row_norm;          /* upper bound on the L2 norm of every corpus row; 0 = unit rows (<= 1 + 2^-9, what the encoder
                                      writes).  The exactness certificate's rounding tolerance scales with it: a bound that is too
                                      SMALL voids the proof (answers could then differ among ~1e-5-level near-ties), too large only
                                      sends more queries through the slow path.  arx_rows_max_norm_f16 computes it. */
    int32_t cu_limit;              /* compute units the launch stream may use (a stream created with a CU mask); 0 = the whole device.
                                      Sizes the persistent pass-A grid (one block per CU). */
    int32_t flags;                 /* ARX_TOPK_* */
    float   debug_tau_mult;        /* TEST HOOK: multiplies the certificate tolerance; 0 = 1; values in (0, 1) would SHRINK the tolerance
                                      and void the guarantee, so they are refused (1e9 = every group rescored: an exhaustive exact scan) */
    int32_t debug_drop_best;       /* TEST HOOK: the selection forgets its best group, as a rounding accident at the boundary would;
                                      the certificate must recover it */
} arx_topk_options;
/* arx_topk_search (index_i8 == NULL) / arx_topk_search_i8 with an explicit policy; opt == NULL = defaults. */
int32_t arx_topk_search_opt(const void* corpus, const void* index_i8, int64_t n_rows, const void* queries, int32_t n_queries,
                            int32_t dim, int32_t k, float* out_scores, int64_t* out_ids, int64_t idx_base,
                            void* ws, int64_t ws_bytes, const arx_topk_options* opt, void* stream);

/* max over rows of the L2 norm of fp16 rows [n_rows, dim] -> *out_max (ONE device float, written by the kernels on `stream`; fp32
 * accumulation, rounded up).  What arx_topk_options.max_row_norm wants for a shard that was not written by the encoder (rows loaded
 * from a user's .npy).  A non-finite row gives +inf/NaN: refuse to index such a shard. */
int32_t arx_rows_max_norm_f16(const void* rows, int64_t n_rows, int32_t dim, float* out_max, void* stream);

/* Exactness certificate of the LAST arx_topk_search on this workspace (csrc/search_tail.h, rescore_kernel step 5): the number of
 * queries whose first selection could not be certified (an unscored 64-row group reached the k-th exact score minus the
 * rounding tolerance) and the number of extra groups that were then rescored for them.  Every answer is exact either way; the
 * counters say how often the slow path ran (near-duplicate chunks).  Copies 16 bytes to the host and waits on `stream`. */
int32_t arx_topk_stats(const void* ws, int64_t* flagged_queries, int64_t* extra_groups, void* stream);

/* A HIP stream restricted to a subset of the compute units (hipExtStreamCreateWithCUMask): bit i of cu_mask = CU i in the driver's
 * enumeration, which deals consecutive bits round-robin to the 8 XCDs — a contiguous run of 8 n bits is n CUs on every XCD.  The
 * pipelined search (ShardIndex.search_many) runs pass A of batch b + 1 on most CUs and the select / rescore / merge tail of batch b on a
 * few, so that the tail's whole-CU blocks never wait for pass-A blocks to drain.  *out is a hipStream_t. */
int32_t arx_stream_create_cu_mask(const uint32_t* cu_mask, int32_t n_words, void** out);
int32_t arx_stream_destroy(void* stream);
/* compute units of the current device */
int32_t arx_device_cu_count(void);
/* Debug: out[b] (device uint32 [n_blocks]) = (XCC_ID << 16) | (HW_ID & 0xffff) of the CU that ran block b of a grid of one-wave blocks that
 * each idle for spin_cycles — which compute units the stream's queue really uses (HW_ID: bits 8-11 CU, 12 shader array, 13-15 engine). */
int32_t arx_debug_cu_census(uint32_t* out, int32_t n_blocks, int32_t spin_cycles, void* stream);

/* Merge P partial top-k lists (e.g. the all-gathered per-shard results) into the global top-k.
 *   scores f32 [P, n_queries, k], ids int64 [P, n_queries, k] (device) -> out [n_queries, k]. */
int32_t arx_topk_merge(const float* scores, const int64_t* ids, int32_t n_parts, int32_t n_queries,
                       int32_t k, float* out_scores, int64_t* out_ids, void* stream);

/* Raw linear layer of the path: C[M,N] (bf16) = epi(A[M,K] (bf16) x W[N,K]^T (bf16) + bias[N] (f32)),
 * mode 0 = bias, 1 = bias + erf-GELU, 2 = bias + resid[M,N] (bf16).  K % 64 == 0, N % 8 == 0.
 * `variant` selects the main-loop schedule (see csrc/encoder.hip); exposed for unit tests and tuning. */
int32_t arx_gemm_bf16(const void* A, const void* W, const float* bias, const void* resid, void* C,
                      int32_t M, int32_t N, int32_t K, int32_t mode, int32_t variant, void* stream);

/* out[i] = cos(emb[i], emb[i+1]) for i in [0, n-1): the adjacent-sentence similarity the stage-3 semantic chunker
 * thresholds (/root/reference/3-chunks/pipeline/src/processors/text_processor.py:1555-1561, helper :1601-1605).
 * emb device f32 [n, ld >= dim]; out device f32 [n-1]. */
int32_t arx_adjacent_cosine(const float* emb, int64_t ld, int32_t n, int32_t dim, float* out, void* stream);

/* ---- host-side WordPiece feeder (no device code) ------------------------------------------------------------------------------
 * Replaces, for pure-ASCII texts, the tokenisation sentence-transformers performs inside `model.encode(batch, ...)`
 * (/root/reference/4-embed/generation/generate_embeddings_parallel.py:146-153; HF `tokenizers` pipeline BertNormalizer ->
 * BertPreTokenizer -> WordPiece("##") -> "<bos> $A <eos>" -> truncation, transformers models/mpnet/tokenization_mpnet.py:108-163).
 * Texts containing one of the `triggers` (the tokenizer's added-token strings) are FLAGGED, not tokenised: the caller sends them
 * through the reference pipeline; non-ASCII segments go through a cache of that pipeline's output (below).  Multi-threaded; writes a
 * padded id matrix and lengths directly. */
int32_t arx_wp_create(const char* vocab_blob, const int64_t* vocab_off /* [n_vocab+1] */, int32_t n_vocab, int32_t unk_id,
                      int32_t bos_id, int32_t eos_id, int32_t pad_id, int32_t lowercase, int32_t max_chars_per_word,
                      const char* trigger_blob, const int64_t* trigger_off /* [n_triggers+1] */, int32_t n_triggers, void** out);
void arx_wp_destroy(void* tokenizer);
/* ids: host int32 [n, max_len] (rows padded with pad_id), lens: host int32 [n], fallback: host uint8 [n] (1 = not tokenised here) */
int32_t arx_wp_encode(void* tokenizer, const char* text_blob, const int64_t* text_off /* [n+1] */, int64_t n, int32_t max_len,
                      int32_t* ids, int32_t* lens, uint8_t* fallback, int32_t n_threads);
/* fallback[i]: 0 = tokenised; 1 = send the whole text through the reference pipeline (added-token string present, or a non-ASCII
 * run > 512 bytes); 2 = the text has non-ASCII whitespace-delimited segments the tokenizer has not been taught yet: fetch them
 * (arx_wp_miss_count / arx_wp_miss_fetch), tokenise each ONCE with the reference pipeline (no specials, no truncation), hand the
 * pieces back (arx_wp_cache_add) and encode those texts again.  tokens(text) is the concatenation of tokens(segment) because every
 * stage of the pipeline is local to a whitespace-delimited segment. */
int32_t arx_wp_miss_count(void* tokenizer, int64_t* n_strings, int64_t* n_bytes);
int32_t arx_wp_miss_fetch(void* tokenizer, char* blob, int64_t* off /* [n_strings+1] */);
int32_t arx_wp_cache_add(void* tokenizer, const char* seg_blob, const int64_t* seg_off /* [n+1] */, int64_t n, const int32_t* ids,
                         const int64_t* ids_off /* [n+1] */);
int64_t arx_wp_cache_size(void* tokenizer);
int32_t arx_wp_version(void);

/* ---- small device helpers the host code needs (all on `stream`) -------------------------------- */
/* f32 [n] -> bf16 [n] round-to-nearest-even (weight upload). */
int32_t arx_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);
/* L2-normalised N(0,1) rows in fp16, generated on device from (seed, row index): bench cfg 3. */
int32_t arx_fill_unit_rows_f16(void* dst, int64_t n_rows, int32_t dim, uint64_t seed, void* stream);
/* The same generator for a row RANGE of a larger corpus: dst row j = row (row_base + j) of the corpus arx_fill_unit_rows_f16 would
 * write for this seed — every rank of a sharded bench fills its own slice of ONE corpus (bench cfg 4: 5 M rows cut 8 ways), so that
 * the merged answer can be checked against a single-index search of the same rows. */
int32_t arx_fill_unit_rows_f16_at(void* dst, int64_t n_rows, int32_t dim, uint64_t seed, int64_t row_base, void* stream);
/* EMBEDDING-LIKE synthetic rows (bench: search on something harder than iid Gaussian directions): row r belongs to cluster
 * hash(seed, r) % n_clusters and is centre(cluster) + spread * noise(r), both N(0,1) per dimension times a per-dimension gain that is
 * hot_gain on n_hot_dims dimensions chosen by the seed (outlier dimensions: they set max|x| and with it the int8 scale of every row)
 * and 1 elsewhere; L2-normalised, fp16.  Rows of the same seed share the centres whatever row_base is, so queries drawn with a
 * row_base beyond the corpus are new points of the same mixture.  n_clusters < 0 = TOPIC ORDER: cluster(r) = r / (-n_clusters), i.e.
 * consecutive runs of -n_clusters rows share a centre (the chunks of one paper: neighbours in row order and in embedding space — whole
 * 64-row groups of near-tied rows, the hard layout for group-max selection).  dim % 128 == 0, dim <= 1024. */
int32_t arx_fill_clustered_rows_f16_at(void* dst, int64_t n_rows, int32_t dim, uint64_t seed, int64_t row_base, int32_t n_clusters,
                                       float spread, int32_t n_hot_dims, float hot_gain, void* stream);

/* ---- live per-kernel timing (bench.py roofline leg) ----------------------------------------------
 * When enabled, every launch of a hot kernel class is bracketed by hipEvents recorded on the launch
 * stream; arx_prof_read synchronises on them and returns the summed device time and launch count. */
#define ARX_K_SEARCH_GROUPMAX 0    /* search pass A: f16 MFMA GEMM + max over 64-row groups */
#define ARX_K_GEMM_QKV        1
#define ARX_K_GEMM_OPROJ      2
#define ARX_K_GEMM_FC1        3
#define ARX_K_GEMM_FC2        4
#define ARX_K_ATTENTION       5
#define ARX_K_LAYERNORM       6
#define ARX_K_EMBED           7
#define ARX_K_POOL            8
#define ARX_K_SEARCH_SELECT   9
#define ARX_K_SEARCH_RESCORE 10
#define ARX_K_GEMM_RAW       11    /* arx_gemm_bf16 called directly (unit tests, tuning): whatever its shape */
#define ARX_K_CLASSES        12
int32_t arx_prof_enable(int32_t on);
/* bit c set = kernel class c (ARX_K_*) records its event pair while profiling is on (default: all).  A timed run enables only the
 * class it reports, so that the other ~170 event packets per forward do not sit between the kernels being timed. */
int32_t arx_prof_classes(uint32_t mask);
int32_t arx_prof_reset(void);
int32_t arx_prof_read(int32_t kernel_class, float* total_ms, int32_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* ARX_H */
