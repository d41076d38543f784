/* rmu.h -- C-ABI of librmu.so: the MI355X-native retrieval hot path of RAGMeUp.
 *
 * The reference (AI-Commandos/RAGMeUp @ 2025-01-03) has no FFI for this path: its boundary is the
 * LangChain plug-in surface that server/RAGHelper.py touches.  librmu.so is what
 * OUR implementations of those plug-ins bind through ctypes; every entry point below names the
 * reference call it serves.  Plain pointers and sizes only -- no torch / C++ types cross this line.
 *
 * Conventions
 *   - return 0 = OK, <0 = error class (RMU_E_*); rmu_last_error() gives the thread-local message.
 *   - never throws; the caller owns every in/out buffer; the library owns what *_create made.
 *   - device pointers are plain HIP device addresses (e.g. torch.Tensor.data_ptr()).
 *   - hip_stream: 0 = an internal per-thread stream, results complete on return;
 *                 non-zero = a hipStream_t the work is ordered on (caller synchronises).  Scratch space belongs to the
 *                 calling THREAD: a call that returns with work in flight marks its end with an event, and the thread's
 *                 next call on any other stream (or stream 0) is ordered behind it -- consecutive calls from one thread
 *                 never overlap on the device, whichever streams they name; use one thread per concurrent stream.
 *   - thread-safe: searches on one index run concurrently (shared lock); add/remove are exclusive.
 *   - a good neighbour in the host process (the reference runs its LLM in PyTorch on the same GPU, RAGHelper_local.py:42-105): no entry point
 *     synchronises the whole device or issues a synchronous copy -- a writer (add that re-allocates, remove_rows, free) waits for exactly
 *     the streams on which searches of that index are still in flight -- and every entry point runs with the calling thread's stream-capture
 *     interaction mode set to relaxed for its duration, so a hipGraph / torch.cuda.graph capture on another thread is never invalidated.
 *   - row ids are int64 row numbers in insertion order; pk/metadata mapping stays in the host language.
 */
#ifndef RMU_H_
#define RMU_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RMU_OK 0
#define RMU_E_INVALID (-1) /* bad argument / unsupported shape */
#define RMU_E_HIP (-2)     /* HIP runtime error */
#define RMU_E_OOM (-3)     /* device or host allocation failed */
#define RMU_E_RCCL (-4)    /* RCCL error (librccl.so missing, communicator or collective failure) */

#define RMU_METRIC_IP 0     /* larger = better */
#define RMU_METRIC_COSINE 1 /* rows are stored L2-normalised, queries normalised per call */
#define RMU_METRIC_L2SQ 2   /* squared L2 distance on the stored (un-normalised) rows, smaller = better (Milvus "L2");
                             * dim <= 767 (rows carry -|x|^2 in one pad column: 384-d rows are stored 768 wide; at dim 384 the index also keeps the
                             * fp16 screening image and one fp32 norm per row) */

/* flags for rmu_index_search / rmu_topk_merge */
#define RMU_F_Q_DEVICE 1u   /* query pointer is a device address */
#define RMU_F_OUT_DEVICE 2u /* output pointers are device addresses */
#define RMU_F_SMALLER_BETTER 4u /* rmu_topk_merge / rmu_shard_allgather_topk: the scores are distances (RMU_METRIC_L2SQ lists) */

/* options for rmu_index_set_option */
#define RMU_OPT_SCREEN 1    /* 1 (default): searches may take the fp16 screening path; 0: always the exact fp32 scan.
                             * Results are identical either way (bench.py times both through this switch). */

#define RMU_OPT_SCREEN_MIN_NQ 2 /* n > 0: take the screening path for every batch of >= n queries whatever the corpus size (by
                             * default small batches over small corpora take the exact scan, which is faster there); 0: default.
                             * Results are identical either way (the tests force the path through this switch). */

#define RMU_OPT_LADDER_RATIO 3   /* tuning: growth ratio of the screening path's threshold ladder above 64k rows (0 = default: 3, or the
                             * small-batch default for <= 128 queries).  Results are identical for every value. */
#define RMU_OPT_LADDER_FIRST 4   /* tuning: rows of the ladder's smallest first range (0 = default).  Results identical for every value. */

#define RMU_MAX_K 112       /* largest k the fused scan keeps in LDS */
#define RMU_MAX_DIM 768

typedef struct rmu_index rmu_index_t;
typedef struct rmu_bert rmu_bert_t;
typedef struct rmu_comm rmu_comm_t;

/* ---- runtime ------------------------------------------------------------------------------- */
/* hipSetDevice(device_ordinal); idempotent.  Serves: RAGHelper_local.py:107-117 (device choice). */
int rmu_init(int device_ordinal);
const char* rmu_last_error(void);
/* "librmu <ver> gfx950" -- lets the host fail loudly on a wrong build. */
const char* rmu_version(void);

/* ---- HBM-resident flat index ------------------------------------------------------------------
 * Serves: RAGHelper.py:385-404 (Milvus.from_documents / PGVector ctor -> an empty collection). */
int rmu_index_create(rmu_index_t** out, int dim, int metric, int64_t capacity_hint);
int rmu_index_free(rmu_index_t* idx);
int rmu_index_size(rmu_index_t* idx, int64_t* n_rows);
int rmu_index_dim(rmu_index_t* idx, int* dim);
int rmu_index_metric(rmu_index_t* idx, int* metric);
int rmu_index_set_option(rmu_index_t* idx, int option, int64_t value);
/* Bookkeeping a host may want to report (bench.py's indexing leg does): allocated row capacity, how often rmu_index_add had to
 * re-allocate and copy the corpus matrix (+ screening image) and the wall time that took, live (non-tombstoned) rows.
 * Serves: the growth of the Milvus collection under RAGHelper.py:423-434's 1000-document inserts (no capacity is known
 * up front there either). */
#define RMU_STAT_CAPACITY 1
#define RMU_STAT_GROW_COUNT 2
#define RMU_STAT_GROW_MS 3
#define RMU_STAT_LIVE_ROWS 4
int rmu_index_stat(rmu_index_t* idx, int what, double* out);
/* Make room for `rows` rows in all (grow-only; at most ONE re-allocation, none if the capacity is there).  A re-allocation waits for
 * everything in flight on the device before the old matrix is freed: a caller that knows how many rows are coming -- or that is about to
 * leave work in flight (the insert pipeline's worker, between two forwards) -- pays for it at a moment of its choosing instead of inside
 * an rmu_index_add behind a forward.  Serves: RAGHelper.py:423-434's insert loop (the collection grows batch by batch). */
int rmu_index_reserve(rmu_index_t* idx, int64_t rows);

/* Append n rows ([n, dim] fp32 row-major, host or device).  *first_row = row id of vecs[0].
 * Serves: RAGHelper.py:431, :525 (db.add_documents -> add_texts -> insert). */
int rmu_index_add(rmu_index_t* idx, const float* vecs, int64_t n, int is_device, int64_t* first_row);

/* Tombstone rows (they stop appearing in results; storage is not compacted).  *n_removed counts rows
 * that were live.  Serves: server.py:373-377 (collection.delete('source == ...') -> delete_count). */
int rmu_index_remove_rows(rmu_index_t* idx, const int64_t* rows, int64_t n, int64_t* n_removed);

/* Gather stored rows to the host ([n, dim] fp32).  Serves: the MMR retriever's
 * `col.query(expr="pk in [...]", output_fields=[vector])` round trip (RAGHelper.py:497-499). */
int rmu_index_get_rows(rmu_index_t* idx, const int64_t* rows, int64_t n, float* out_host);

/* Batched greedy maximal-marginal-relevance selection on the device (fp64, one wave per query): for query i pick k of the
 * fetch_k candidate rows rows[i, :] (index-local ids as rmu_index_search returned them, -1 = absent): first the candidate
 * most similar to the query, then repeatedly argmax lambda*cos(q,x) - (1-lambda)*max cos(x, picked), lowest position on
 * ties.  out_pos [nq, k] int32 = positions in the candidate list (-1 past the number of candidates).
 * flags: RMU_F_Q_DEVICE (q), RMU_F_OUT_DEVICE (rows and out_pos).  fetch_k <= 64.
 * Serves: VectorStoreRetriever(search_type="mmr") -> maximal_marginal_relevance (RAGHelper.py:497-499) without the
 * per-query "fetch 20 vectors by pk" round trip. */
int rmu_index_mmr(rmu_index_t* idx, const float* q, int64_t nq, const int64_t* rows, int fetch_k, int k,
                  double lambda_mult, unsigned flags, int32_t* out_pos);

/* The reference's per-request retrieval in ONE call (VectorStoreRetriever.invoke with search_type="mmr", RAGHelper.py:497-499:
 * dense top-fetch_k, then maximal_marginal_relevance over those candidates): rmu_index_search followed by rmu_index_mmr on
 * the device-resident candidate list, one host round trip.  HOST q [nq, dim]; HOST outputs out_rows [nq, k] int64 (row ids +
 * row_base in pick order, -1 past the number of candidates) and out_scores [nq, k] fp32 (the search score of each pick; may
 * be NULL).  fetch_k <= 64, k <= fetch_k.  Results equal rmu_index_search + rmu_index_mmr called one after the other. */
int rmu_index_search_mmr(rmu_index_t* idx, const float* q, int64_t nq, int fetch_k, int k, double lambda_mult, int64_t row_base,
                         int64_t* out_rows, float* out_scores);

/* Persist / restore the corpus matrix (flat file: 64-byte header, liveness bytes, fp32 rows; restart = one H2D copy).
 * Serves: the Milvus-Lite `data.db` the reference re-opens when vector_store_initial_load is False
 * (RAGHelper.py:391, :417; .env.template:33,36).  Tombstones survive (poisoned rows are stored as they are). */
int rmu_index_save(rmu_index_t* idx, const char* path);
int rmu_index_load(rmu_index_t** out, const char* path);

/* Exact top-k of every query against all live rows (dim 384, k <= 104, any metric: fp16 screening + exact fp32 re-score under a
 * per-query sufficiency test; otherwise, and for every query that fails the test, the exact fp32 fused scan -- the
 * returned ids and scores are those of the exact scan either way; the failing queries are re-run by launches that are
 * predicated on the device, so the call never waits on the host for a decision and, given a caller stream with device
 * buffers and nq <= 8192, returns without synchronising it).
 *   q [nq, dim] fp32; out_scores [nq, k] fp32, out_rows [nq, k] int64, best first,
 *   order (score, then lower row id); slots beyond the live row count hold (-inf | +inf for L2SQ, -1).
 *   row_base is added to every returned row id (shard offset).
 * Serves: RAGHelper.py:497-499 -> vector-store similarity search (Milvus col.search FLAT). */
int rmu_index_search(rmu_index_t* idx, const float* q, int64_t nq, int k, unsigned flags,
                     int64_t row_base, float* out_scores, int64_t* out_rows, uint64_t hip_stream);

/* Merge `parts` per-shard top-k lists ([parts, nq, k] each, best first; larger score = better unless
 * RMU_F_SMALLER_BETTER) into one [nq, k].  Ties: lower part index first (give shards in ascending row order).
 * Serves: the 8-GPU shard merge after the RCCL all-gather; no reference counterpart. */
int rmu_topk_merge(const float* scores, const int64_t* rows, int parts, int64_t nq, int k,
                   unsigned flags, float* out_scores, int64_t* out_rows, uint64_t hip_stream);

/* ---- multi-GPU: the one exchange step of the row-sharded search -----------------------------------------
 * One process per GPU.  Rank 0 calls rmu_comm_unique_id and hands the 128 bytes to the other ranks by any side channel
 * (a file, a socket, torch.distributed's store); every rank then calls rmu_comm_init on ITS device (rmu_init first).
 * RCCL is bound at run time (dlopen of librccl.so; RMU_E_RCCL if absent).  No reference counterpart. */
#define RMU_COMM_ID_BYTES 128
int rmu_comm_unique_id(void* id_out /* RMU_COMM_ID_BYTES */);
int rmu_comm_init(rmu_comm_t** out, const void* id, int world, int rank);
int rmu_comm_free(rmu_comm_t* comm);
int rmu_comm_world(rmu_comm_t* comm, int* world, int* rank);
/* Every rank passes its local [nq, k] lists (what rmu_index_search returned with row_base = the shard's first row):
 * ONE RCCL all-gather of nq*k*12 bytes per rank over xGMI, then the W lists are merged on the device; every rank ends
 * with the same global [nq, k].  flags: RMU_F_Q_DEVICE (inputs), RMU_F_OUT_DEVICE (outputs), RMU_F_SMALLER_BETTER. */
int rmu_shard_allgather_topk(rmu_comm_t* comm, const float* scores, const int64_t* rows, int64_t nq, int k, unsigned flags,
                             float* out_scores, int64_t* out_rows, uint64_t hip_stream);

/* Test hook (tests/test_search_gpu.py), not a product entry point: the screening pass's K' = 32 candidates of each of nq
 * host queries -- approximate scores, row ids, the exact fp32 score of the same rows, and the error bound EPS(q) of the
 * sufficiency test -- so |approx - exact| <= EPS can be checked on the hardware.  All outputs are host arrays
 * ([nq, 32] x 3 and [nq]); absent candidates are (-inf, -1, -inf). */
int rmu_index_screen_candidates(rmu_index_t* idx, const float* q_host, int64_t nq, float* out_approx, int64_t* out_rows,
                                float* out_exact, float* out_eps);

/* Timing hook for bench.py: duration in ms of the last fused scan kernel launched by the calling
 * thread, measured with hipEvents on the stream the kernel ran on; <0 if none. */
float rmu_last_scan_ms(void);
/* Same for the whole search (scan + merge), and the launch geometry of the last scan. */
float rmu_last_search_ms(void);
int rmu_last_scan_geometry(int* grid, int* block, int* lds_bytes, int* passes);
/* How the calling thread's last rmu_index_search was answered: >0 by the fp16 screening ladder + exact fp32 re-score;
 * <0 the same, with that many queries failing the sufficiency test and re-run on the exact fp32 scan (patched in; the
 * count is known only when the call itself drained the stream, i.e. not for an un-synchronised caller stream);
 * 0 by the exact fp32 scan alone.  Results are identical in all three cases. */
int rmu_last_screened(void);
/* Enable (1) / disable (0) the event timing above for the calling thread (off by default). */
int rmu_set_timing(int on);
/* Diagnostic for bench.py (no reference counterpart): the rate in TFLOP/s THIS GPU sustains on v_mfma_f32_32x32x16 with operands that change
 * from instruction to instruction (random N(0, 3.3) values: the fp16 image's distribution) -- the power-limited roof a kernel working on data
 * can approach, as opposed to the nominal peak reached on constant operands (tools/ubench/mfma_power.hip, profiles/r06_mfma_power.txt).
 * dtype 0 = f16, 1 = bf16.  variant 0 = nothing but MFMAs (two waves per SIMD on every CU); variant 1 (f16 only) = the screening kernel's operand
 * delivery beside them: one 1-KiB LDS fragment read per MFMA + one 1-KiB LDS-DMA piece per wave and 8 MFMAs.  Runs ~millis ms of back-to-back
 * launches on a stream of its own and returns the mean of the last half of them.  Thread-safe; allocates and frees its own buffers. */
int rmu_probe_mfma_rate(int dtype, int variant, int millis, double* tflops_out);

/* ---- BERT-6x384 encoder (bi-encoder and cross-encoder forwards) -------------------------------
 * Serves: HuggingFaceEmbeddings.embed_documents / embed_query (RAGHelper_local.py:107-117 via
 * RAGHelper.py:423-434) and HuggingFaceCrossEncoder.score (RAGHelper.py:483-486 ->
 * ScoredCrossEncoderReranker.py:42). */
typedef struct rmu_bert_cfg {
    int vocab_size;   /* 30522 */
    int hidden;       /* 384 (must be 384 in this build) */
    int layers;       /* 6 */
    int heads;        /* 12 (head_dim 32) */
    int ffn;          /* 1536 */
    int max_pos;      /* 512 */
    int type_vocab;   /* 2 */
    float ln_eps;     /* 1e-12 */
    int has_head;     /* 1: pooler.dense + classifier (cross-encoder) weights are supplied */
} rmu_bert_cfg;

/* weights: array of DEVICE pointers to fp32 tensors in HF layout, order documented in
 * ragmeup_amd/bert.py (WEIGHT_ORDER).  The library converts them to bf16 MFMA operand layout once. */
int rmu_bert_create(rmu_bert_t** out, const rmu_bert_cfg* cfg, const void* const* weight_ptrs, int n_weights);
int rmu_bert_free(rmu_bert_t* m);
/* What rmu_bert_encode writes (`mode`): the head behind the transformer as the checkpoint declares it --
 * sentence-transformers `1_Pooling/config.json` (pooling_mode_mean_tokens | pooling_mode_cls_token) and `modules.json`
 * (Normalize present or not) for the bi-encoder, BertForSequenceClassification(num_labels = 1) for the cross-encoder. */
#define RMU_BERT_POOL_MEAN 0   /* masked mean over the tokens (+ L2 normalise) -> out_dev fp32 [batch, out_stride] (first `hidden` cols) */
#define RMU_BERT_CE_LOGIT 1    /* pooler tanh + Linear(hidden, 1) logit        -> out_dev fp32 [batch] */
#define RMU_BERT_POOL_CLS 2    /* first token's state (+ L2 normalise)         -> out_dev fp32 [batch, out_stride] */
#define RMU_BERT_TOKENS 3      /* final hidden state of every real token (sentence-transformers output_value="token_embeddings"):
                                * packed rows, sequence b at rows [sum(len[<b]), +len[b]), len = min(lens, max_len)
                                *                                              -> out_dev fp32 [sum len, out_stride] */
#define RMU_BERT_NO_NORMALIZE 0x100 /* OR-ed into POOL_MEAN / POOL_CLS: the checkpoint has no Normalize module */
/* ids/type_ids: device int32 [batch, max_len] (row padded), lens: device int32 [batch].
 * Rounding and batch shape: activations are bf16, and which kernels serve a call depends on batch * max_len (<= 256 tokens: the
 * small-batch GEMMs; <= 16384: the GEMM pair; above: the fused FFN kernel) -- a sequence's result is bit-identical across
 * calls that take the same kernels and equal up to bf16 rounding noise (|d| < 2e-3 on unit vectors, cosine > 0.9999)
 * otherwise: a query embedded alone reproduces the vector its text got at indexing time to that noise, not bit for bit.
 * hip_stream != 0: the forward is left in flight on that stream (inputs and out_dev must stay valid until it has run); the model's next
 * call on any OTHER stream -- stream 0 and the host-path entry points included -- is ordered behind it on the device (one workspace per
 * model), so a caller may queue the next block's forward while this one runs. */
int rmu_bert_encode(rmu_bert_t* m, const int32_t* ids, const int32_t* type_ids, const int32_t* lens,
                    int batch, int max_len, int mode, float* out_dev, int64_t out_stride,
                    uint64_t hip_stream);

/* The same forward for a HANDFUL of tokens (batch * max_len <= 4096 and <= 256 result rows: embed_query, the <= 14 (query, passage)
 * pairs of one rerank call) from HOST buffers to a HOST result: the interactive per-request pattern of the reference (one query per
 * /chat call, RAGHelper.py:497-499; ScoredCrossEncoderReranker.py:42).  The library replays one captured hipGraph per input shape
 * (H2D, ~45 launches, D2H: one graph launch, one synchronisation) and keeps the 64 most recently used shapes: callers should bucket
 * batch and max_len (a padded sequence has lens = 0 and costs nothing).
 * ids / type_ids (may be NULL) [batch, max_len], lens [batch]: host int32; out_host as out_dev above, on the host. */
int rmu_bert_encode_host(rmu_bert_t* m, const int32_t* ids, const int32_t* type_ids, const int32_t* lens,
                         int batch, int max_len, int mode, float* out_host, int64_t out_stride);

/* The reference's per-request retrieval as ONE call with ONE synchronisation (VectorStoreRetriever.invoke with search_type="mmr",
 * RAGHelper.py:497-499: embed_query -> dense top-fetch_k -> maximal_marginal_relevance): HOST token ids of `batch` queries in (as
 * rmu_bert_encode_host; mode = RMU_BERT_POOL_MEAN | RMU_BERT_POOL_CLS [| RMU_BERT_NO_NORMALIZE]), HOST out_rows [batch, k] int64 (row ids
 * + row_base in pick order, -1 past the number of candidates) and out_scores [batch, k] fp32 (the search score of each pick; may be
 * NULL) out.  The pooled query vectors stay on the device between the forward and the search (out_vecs, if not NULL, receives
 * them: [batch, 384] fp32).  lambda_mult < 0: no selection -- the top-k in score order (k <= fetch_k), i.e. rmu_index_search.
 * fetch_k <= 64, k <= fetch_k; the index must hold 384-d rows.  Results equal rmu_bert_encode_host + rmu_index_search_mmr. */
int rmu_bert_search_mmr(rmu_bert_t* m, rmu_index_t* idx, const int32_t* ids, const int32_t* type_ids, const int32_t* lens, int batch,
                        int max_len, int mode, int fetch_k, int k, double lambda_mult, int64_t row_base, int64_t* out_rows,
                        float* out_scores, float* out_vecs);

/* ---- WordPiece tokenizer (host C++; the step in front of both encoder forwards) --------------------
 * Restates transformers' BertTokenizer (BasicTokenizer + WordPiece) as used by sentence-transformers `tokenize`
 * (HuggingFaceEmbeddings.embed_documents, RAGHelper.py:423-434) and CrossEncoder pair tokenisation
 * (HuggingFaceCrossEncoder.score, RAGHelper.py:483-486).  vocab_path: one token per line (vocab.txt). */
typedef struct rmu_tok rmu_tok_t;
int rmu_tok_create(rmu_tok_t** out, const char* vocab_path, int do_lower_case);
int rmu_tok_free(rmu_tok_t* tk);
int rmu_tok_vocab_size(rmu_tok_t* tk);
/* n UTF-8 strings (texts_b may be NULL, or hold NULL entries, for single sentences).  Host outputs: ids/type_ids
 * [n, max_len] int32 ([CLS] a [SEP] (b [SEP]), [PAD]-filled; type_ids may be NULL), lens [n].  Single sequences keep
 * their first max_len-2 tokens; pairs use "longest_first" truncation. */
int rmu_tok_encode(rmu_tok_t* tk, const char* const* texts_a, const char* const* texts_b, int n, int max_len,
                   int32_t* ids, int32_t* type_ids, int32_t* lens);
/* The same over NUL-separated blobs: blob_a holds n strings back to back, each terminated by '\0' (bytes_a = total size including
 * the terminators); blob_b likewise or NULL.  One host buffer per call instead of n pointers: what a Python caller builds with a
 * single join + encode.  RMU_E_INVALID when a blob does not hold exactly n terminated strings. */
int rmu_tok_encode_blob(rmu_tok_t* tk, const char* blob_a, int64_t bytes_a, const char* blob_b, int64_t bytes_b, int n, int max_len,
                        int32_t* ids, int32_t* type_ids, int32_t* lens);

#ifdef __cplusplus
}
#endif
#endif /* RMU_H_ */
