/*
 * leann_backend.h — C ABI of the MI355X-native ANN search path that replaces leann-rs's
 * backend layer (src/backend) and the arithmetic of its recompute search (src/index/recompute.rs).
 *
 * Every entry point cites the reference interface it replaces (path:line in decisiongraph/leann-rs).
 * Plain pointers and sizes only; no torch / HIP types.  Thread-safety: `*_search*` are re-entrant
 * on one handle (BackendSearcher: Send + Sync, src/backend/traits.rs:11; concurrent callers at
 * src/cli/serve.rs:289-292); open/close/build are not.
 *
 * Return value: 0 on success, non-zero error code otherwise; the message is available from
 * leann_last_error() (thread-local), mirroring the anyhow::Result strings of the reference.
 *
 * Distances: dist = 1 - <q, x> (MetricKind::IP, src/backend/hnsw.rs:45; DistDot,
 * src/backend/diskann.rs:8,36), ascending, ties broken by lower key.  Keys are 0-based positions
 * in embedding order (src/backend/hnsw.rs:129), widened to u64 (src/backend/diskann.rs:58).
 */
#ifndef LEANN_BACKEND_H
#define LEANN_BACKEND_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct leann_backend leann_backend;

/* enum BackendType { Hnsw, DiskAnn }  — src/backend/mod.rs:15-19 */
enum { LEANN_BACKEND_HNSW = 0, LEANN_BACKEND_DISKANN = 1 };

enum {
    LEANN_OK = 0,
    LEANN_ERR_INVALID = 1,      /* bad argument */
    LEANN_ERR_NOT_FOUND = 2,    /* "Index file not found" hnsw.rs:34-40 / diskann.rs:26-32 */
    LEANN_ERR_FORMAT = 3,       /* FAISS / foreign format, hnsw.rs:24-32,57-69; compat.rs:15-38 */
    LEANN_ERR_DEVICE = 4,       /* HIP runtime / no GPU */
    LEANN_ERR_UNSUPPORTED = 5,  /* e.g. DiskANN incremental add, mod.rs:93-98 */
    LEANN_ERR_IO = 6,
    LEANN_ERR_OVERFLOW = 7      /* a query ran out of visited-set space (defensive; see leann_backend_search_batch_device) */
};

const char *leann_last_error(void);
const char *leann_version(void);

/* ---- BackendType::load_searcher(index_path, dimensions)  src/backend/mod.rs:23-45 -------------
 * `index_path_stem` is ".../documents.leann"; the backend derives "<stem minus .leann>.index"
 * (hnsw.rs:19) or ".diskann" (diskann.rs:22) itself.
 * `device_spec`: NULL / "" / "0" ... = one HIP device ordinal; a list or range ("0,1,2,3", "0-7"; the same ordinal repeated =
 * several shards on one device) opens the index SHARDED: the rows (from "<stem>.embeddings", else from this library's own index
 * file) are split into contiguous position ranges, one sub-index with its own graph per entry, and the returned handle fans every
 * search out to them and merges the per-shard lists by (dist, key) — see "sharded indexes" below.  The Rust side passes
 * std::env::var("LEANN_DEVICES") here.
 * Files: this library's own format ("LEANNGX1": graph + vectors, or graph + encoder inputs for a recompute-on index).  A usearch /
 * diskann-rs file written by stock leann-rs is not readable; when the directory also holds "<stem>.embeddings"
 * (src/index/embeddings.rs:21-153) the graph is rebuilt from it on the GPU and cached as "<stem>.gpu.index" / ".gpu.diskann".
 * Every file is validated (header against file length, every neighbour id / list offset against n) before it reaches the device. */
int leann_backend_open(const char *index_path_stem, int backend, size_t dims,
                       const char *device_spec, leann_backend **out);

/* ---- BackendSearcher::search(&self, query, top_k, complexity)  src/backend/traits.rs:16-21 ----
 * Caller allocates keys/dists[top_k]; the first *n_out are filled, best first; *n_out <= top_k
 * (short results allowed, src/index/searcher.rs:139-143).  ef = max(complexity, top_k)
 * (diskann.rs:54).  Unlike hnsw.rs:83 `complexity` is honoured for HNSW too; with LEANN_HNSW_REFERENCE_EF=1 in the environment
 * when the handle is made, an HNSW handle behaves like the reference instead: ef = max(64, top_k) whatever `complexity` says
 * (expansion_search: 64, hnsw.rs:49; `_complexity` unused, :83). */
int leann_backend_search(const leann_backend *h, const float *query, size_t top_k,
                         size_t complexity, uint64_t *keys, float *dists, size_t *n_out);

/* Additive: nq queries [nq x dims] row-major in one launch; equals nq single calls.
 * keys/dists are [nq x top_k] (unused tail: key = UINT64_MAX, dist = +inf), counts[nq]. */
int leann_backend_search_batch(const leann_backend *h, const float *queries, size_t nq,
                               size_t top_k, size_t complexity, uint64_t *keys, float *dists,
                               uint32_t *counts);

/* Additive: metadata-filtered search with the filter evaluated INSIDE the traversal,
 * replacing IndexSearcher's fetch_k = 5*top_k over-fetch + post-filter (src/index/searcher.rs:129-133,:190-194).
 * `allow` is a bitmap over positions (bit i&7 of byte i>>3 set = position i may be returned; positions are local to
 * the handle, i.e. key - key_offset), ceil(len/8) bytes; in the batch call query i uses allow + i*allow_stride
 * (allow_stride == 0: one bitmap for the whole batch).  The graph is walked exactly as by the unfiltered search
 * (disallowed nodes still route); the answer is the top_k best allowed positions among EVERY node whose distance
 * the level-0 walk evaluated (~30x the beam), best first.  allow == NULL: the unfiltered search. */
int leann_backend_search_filtered(const leann_backend *h, const float *query, size_t top_k,
                                  size_t complexity, const uint8_t *allow, uint64_t *keys,
                                  float *dists, size_t *n_out);
int leann_backend_search_filtered_batch(const leann_backend *h, const float *queries, size_t nq,
                                        size_t top_k, size_t complexity, const uint8_t *allow,
                                        size_t allow_stride, uint64_t *keys, float *dists,
                                        uint32_t *counts);

/* Additive: the same filter answered EXACTLY, for filters that allow only a small share of the rows (a graph walk finds
 * few allowed nodes then: 1 % allowed -> filtered recall@10 0.86 at complexity 256).  The bitmap is compacted into the
 * list of allowed positions and only those rows are scanned (f32 matrix cores, one k-ordered fmaf chain per pair, like
 * leann_scan_topk_device): top_k best allowed positions by distance 1 - <x, q>, ties to the lower position; cost is
 * proportional to the number of allowed rows, independent of the graph.  The host-side planner
 * (host/leann_host.hpp IndexSearcher) switches to it below 5 % allowed (single queries; ~1.5 % is the crossover in 16k-query batches).  Needs stored vectors. */
int leann_backend_search_filtered_exact_batch(const leann_backend *h, const float *queries, size_t nq,
                                              size_t top_k, const uint8_t *allow, size_t allow_stride,
                                              uint64_t *keys, float *dists, uint32_t *counts);

/* Additive: a filter registered on the device.  A server answers many queries under the same metadata filter; the calls above
 * re-send the N/8-byte bitmap and (exact path) re-compact it for every query.  leann_backend_filter_create uploads the bitmap
 * and builds the list of allowed positions once; leann_backend_search_filter_batch then answers nq queries under it.
 * mode: 0 = walk the graph with the filter inside (leann_backend_search_filtered), 1 = exact scan of the allowed rows
 * (leann_backend_search_filtered_exact_batch), 2 = choose: exact when the filter allows <= 5 % of the rows or <= 64k rows
 * (batches of <= 64 queries) / <= 1.5 % (larger batches) and the index stores vectors, the walk otherwise.  Results are those of
 * the corresponding call above, bit for bit.  Free the filter only when no search is using it. */
typedef struct leann_filter leann_filter;
int leann_backend_filter_create(const leann_backend *h, const uint8_t *allow, leann_filter **out);
size_t leann_backend_filter_count(const leann_filter *f); /* allowed positions */
void leann_backend_filter_free(leann_filter *f);
int leann_backend_search_filter_batch(const leann_backend *h, const float *queries, size_t nq, size_t top_k,
                                      size_t complexity, const leann_filter *filter, int mode,
                                      uint64_t *keys, float *dists, uint32_t *counts);

/* Additive: request coalescing for servers that call leann_backend_search from many threads (one query per
 * call, src/cli/serve.rs:289-292).  Concurrent callers are gathered for up to wait_us microseconds (or
 * max_batch queries) and answered by one batched launch; results are identical.  (0, 0) disables.
 * Without this call a handle coalesces automatically: a leann_backend_search caller that finds another one in
 * flight queues behind a dispatcher (50 us or 64 queries, ending early once the callers of the previous round
 * are all waiting again) created at that moment, a lone caller is answered
 * directly.  LEANN_COALESCE=off in the environment switches the automatic mode off for the process. */
int leann_backend_set_coalescing(leann_backend *h, uint32_t wait_us, uint32_t max_batch);
int leann_backend_coalescing_stats(const leann_backend *h, uint64_t *n_launches, uint64_t *n_queries);

/* BackendSearcher::len  src/backend/traits.rs:24 */
size_t leann_backend_len(const leann_backend *h);
size_t leann_backend_dims(const leann_backend *h);
/* Drop of Box<dyn BackendSearcher> */
void leann_backend_close(leann_backend *h);

/* ---- BackendBuilder::build(embeddings, ids, index_path, dims, graph_degree, complexity)
 *      src/backend/mod.rs:55-79 -> hnsw.rs:96-139 / diskann.rs:70-105 ---------------------------
 * vectors: [n x dims] row-major host memory.  Writes "<stem>.index" / "<stem>.diskann". */
int leann_backend_build(int backend, const float *vectors, size_t n, size_t dims,
                        size_t graph_degree, size_t complexity, const char *index_path_stem);
/* BackendBuilder::add_to_index(embeddings, index_path, dims, start_id)  mod.rs:82-100,
 * hnsw.rs:142-191: appends to the index on disk (the new rows continue the insertion from the loaded graph; cost
 * proportional to the rows added).  DiskANN -> LEANN_ERR_UNSUPPORTED with the reference's message. */
int leann_backend_add(int backend, const float *vectors, size_t n, size_t dims, size_t start_id,
                      const char *index_path_stem);

/* ---- roofline accounting -------------------------------------------------------
 * Totals accumulated over every search call on the handle since the last reset. */
typedef struct {
    uint64_t n_queries;
    uint64_t n_dist_evals;   /* base vectors whose distance was computed (distinct per level) */
    uint64_t n_hops_base;    /* expansions on level 0 */
    uint64_t n_hops_upper;   /* expansions on levels >= 1 */
    uint64_t n_table_overflow; /* queries re-run with the global-memory visited table */
    uint64_t algorithmic_bytes; /* n_dist_evals*dims*4 + n_hops_base*M0*4 + n_hops_upper*M*4 */
} leann_search_stats;
int leann_backend_stats(const leann_backend *h, leann_search_stats *out, int reset);

/* =================================================================================================
 * Device-resident additive API (no reference counterpart): the same operations with operands that
 * already live in HBM, on a caller-supplied HIP stream (void* = hipStream_t, NULL = default).
 * Used by bench.py, the sharded searcher and the recompute path.
 * ============================================================================================== */

/* In-memory index straight from device-resident rows (no file round trip).
 * d_vectors: [n x ld] f32 in HBM, ld % 4 == 0, ld >= dims, padding zero; borrowed if `take_copy`
 * is 0 (must outlive the handle).  key_offset is added to every returned key (shard rebasing,
 * SURVEY.md §8e). */
int leann_backend_build_device(int backend, const float *d_vectors, size_t n, size_t dims,
                               size_t ld, size_t graph_degree, size_t complexity, int device,
                               uint64_t key_offset, int take_copy, leann_backend **out);
/* Wrap host-side flat graph arrays + host rows into a device index. */
int leann_backend_from_arrays(int backend, const float *vectors, size_t n, size_t dims,
                              uint32_t M, uint32_t M0, uint32_t max_level, uint32_t entry,
                              const uint8_t *levels, const uint32_t *upper_off,
                              const uint32_t *adj0, const uint32_t *adjU, size_t n_upper_lists,
                              int device, uint64_t key_offset, leann_backend **out);
/* Copy the graph back to host arrays (sizes from leann_backend_graph_info). */
int leann_backend_graph_info(const leann_backend *h, uint64_t *info /* n,dims,ld,M,M0,max_level,entry,n_upper_lists */);
int leann_backend_graph_export(const leann_backend *h, uint8_t *levels, uint32_t *upper_off,
                               uint32_t *adj0, uint32_t *adjU, float *vectors /* [n x dims] or NULL */);
/* Persist an in-memory index: "<stem>.index" / ".diskann" */
int leann_backend_save(const leann_backend *h, const char *index_path_stem);

/* d_stats: optional [nq x 4] u32: distance evaluations, level-0 hops, upper-level hops, visited-set level (0 = the LDS
 * table sufficed, 1 / 2 = the query moved to a pooled table in HBM, 3 = even the last level filled up: that query's count is 0
 * and the host-pointer entry points return LEANN_ERR_OVERFLOW; the pools are sized so that 3 cannot occur). */
int leann_backend_search_batch_device(const leann_backend *h, const float *d_queries, size_t nq,
                                      size_t top_k, size_t complexity, uint64_t *d_keys,
                                      float *d_dists, uint32_t *d_counts, uint32_t *d_stats,
                                      void *stream);
/* ... with a device-resident allow-bitmap (see leann_backend_search_filtered) */
int leann_backend_search_filtered_batch_device(const leann_backend *h, const float *d_queries,
                                               size_t nq, size_t top_k, size_t complexity,
                                               const uint8_t *d_allow, size_t allow_stride,
                                               uint64_t *d_keys, float *d_dists, uint32_t *d_counts,
                                               uint32_t *d_stats, void *stream);
int leann_backend_search_filtered_exact_batch_device(const leann_backend *h, const float *d_queries,
                                                     size_t nq, size_t top_k, const uint8_t *d_allow,
                                                     size_t allow_stride, uint64_t *d_keys, float *d_dists,
                                                     uint32_t *d_counts, void *stream);
/* device pointer of the rows (for ground-truth scans) */
const float *leann_backend_device_rows(const leann_backend *h);

/* Deterministic synthetic rows written straight into HBM; bit-identical to
 * oracle/oracle.c:orc_gen_rows. */
int leann_synth_rows_device(uint64_t seed, uint32_t dims, uint32_t ld, uint32_t r,
                            uint32_t n_clusters, float sigma, uint32_t stream_id, uint64_t i0,
                            uint64_t n, float *d_out, void *stream);

/* Exact inner-product top-k of every query against all rows — the arithmetic of
 * RecomputeSearcher::search, src/index/recompute.rs:96-109 (scores = raw dot, descending,
 * ties -> lower position), with the embeddings already materialised in HBM.
 * allow_mask: optional N-bit device bitmap (early filter, recompute.rs:66-71). */
int leann_scan_topk_device(const float *d_rows, size_t n, size_t dims, size_t ld,
                           const float *d_queries, size_t nq, size_t top_k,
                           const uint8_t *d_allow_mask, uint64_t key_offset, uint64_t *d_keys,
                           float *d_scores, uint32_t *d_counts, void *stream);

/* ---- recompute search ("pruned" index, no stored vectors) -----------------------------------------
 * RecomputeSearcher::search(query_embedding, embedding_provider, top_k, filter)
 * src/index/recompute.rs:52-123 with the provider (src/embedding/mod.rs:112-120; dense + L2-normalise
 * tail of src/embedding/candle.rs:165,218-225) resident on the device:
 *     features [n x h] bf16 (borrowed), weights [h x dims] bf16 (copied, re-tiled),
 *     embedding_i = l2_normalize(W^T f_i)  (bf16 MFMA, f32 accumulate)
 * scores = raw dot product, descending, ties -> lower position; allow_mask = early filter (:66-71). */
typedef struct leann_recompute leann_recompute;
int leann_recompute_create(const uint16_t *d_features, size_t n, size_t h, const uint16_t *d_weights,
                           size_t dims, int device, uint64_t key_offset, leann_recompute **out);
/* token-level provider: features [n x L x h] bf16 + attention mask [n x L] (0 = padding, NULL = none), L in
 * {1,2,4,8}: embedding_i = l2_normalize(masked_mean_t(W^T f_it)) — mean_pooling of candle.rs:191-216 with
 * count.clamp(1e-9), then l2_normalize :218-225 */
int leann_recompute_create_pooled(const uint16_t *d_features, const uint8_t *d_mask, size_t n,
                                  size_t tokens_per_passage, size_t h, const uint16_t *d_weights,
                                  size_t dims, int device, uint64_t key_offset, leann_recompute **out);
int leann_recompute_search_batch_device(const leann_recompute *r, const float *d_queries, size_t nq,
                                        size_t top_k, const uint8_t *d_allow_mask, uint64_t *d_keys,
                                        float *d_scores, uint32_t *d_counts, void *stream);
/* host-memory twins (SURVEY.md §8b "Recompute boundary"): plain host pointers in, results out; the handle made by
 * create_host owns a device copy of the features.  keys/scores are [nq x top_k] (unused tail: UINT64_MAX / -inf). */
int leann_recompute_create_host(const uint16_t *features, size_t n, size_t h, const uint16_t *weights,
                                size_t dims, int device, uint64_t key_offset, leann_recompute **out);
int leann_recompute_search_batch(const leann_recompute *r, const float *queries, size_t nq, size_t top_k,
                                 const uint8_t *allow_mask, uint64_t *keys, float *scores, uint32_t *counts);
/* Sharded form: `parts` cover consecutive position ranges (part g's key_offset = part g-1's key_offset + its length,
 * a multiple of 8 past the first part's) and may sit on different devices; a search runs every part's fused scan concurrently and
 * merges the per-part lists by (score descending, key ascending) — bit for bit the answer of one handle over all passages.  The
 * queries, the allow mask (over ALL positions, first part's first position = bit 0) and the results live on the first part's
 * device.  The parts are borrowed: close the composite handle first.  encode / build_index apply to the parts. */
int leann_recompute_create_sharded(const leann_recompute *const *parts, size_t n_parts, leann_recompute **out);
/* materialise embeddings of rows [row0, row0+rows) into d_out [rows x ceil4(dims)] (validation) */
int leann_recompute_encode_device(const leann_recompute *r, uint64_t row0, uint64_t rows, float *d_out,
                                  void *stream);
size_t leann_recompute_len(const leann_recompute *r);
/* HIP-event milliseconds of the last search call: [0] encode GEMM, [1] scoring GEMM, [2] top-k */
int leann_recompute_last_timing(const leann_recompute *r, float *ms3);
void leann_recompute_close(leann_recompute *r);
/* Recompute-on GRAPH index: HNSW / Vamana whose distances are recomputed from the encoder inputs.  The graph is
 * built on transiently materialised embeddings; the returned searcher keeps the graph, the bf16 features and one
 * f32 per passage (||W^T f||) — no vectors — and evaluates dist = 1 - <f, W q> / ||W^T f|| (== 1 - <e, q>).
 * It is an ordinary leann_backend handle: leann_backend_search* work unchanged (queries in embedding space). */
int leann_recompute_build_index(const leann_recompute *r, int backend, size_t graph_degree, size_t complexity,
                                leann_backend **out);
int leann_backend_feature_rows_export(const leann_backend *h, uint32_t *feat_h, uint32_t *row_bytes, void *out);
/* synthetic encoder inputs (bf16), the recompute twin of leann_synth_rows_device */
/* r_int = 0: h-dimensional cluster+noise features; r_int > 0: intrinsic dimension r_int lifted to width h */
int leann_synth_features_device(uint64_t seed, uint32_t h, uint32_t r_int, uint32_t n_clusters, float sigma,
                                uint32_t stream_id, uint64_t i0, uint64_t n, uint16_t *d_out, void *stream);
int leann_synth_weights_device(uint64_t seed, uint32_t h, uint32_t dims, uint16_t *d_out, void *stream);

/* G-way merge of per-shard top-k lists gathered by RCCL: inputs
 * [n_shards x nq x k_in], ascending (dist, key) per list (descending = 1 for recompute scores);
 * outputs [nq x k_out]. */
int leann_merge_topk_device(const uint64_t *d_keys, const float *d_dists, const uint32_t *d_counts,
                            size_t n_shards, size_t nq, size_t k_in, size_t k_out, int descending,
                            uint64_t *d_out_keys, float *d_out_dists, uint32_t *d_out_counts,
                            void *stream);

/* ---- hybrid rerank for batches (BASELINE configs[4]: DiskANN + hybrid BM25 rerank) ------------------------------------------------
 * The hybrid branch of IndexSearcher::search_with_options (src/index/searcher.rs:146-169) + hybrid_rerank (src/index/bm25.rs:135-170)
 * for nq queries at once, on the lists a leann_backend_search_batch_device call with top_k = fetch_k = 5 * top_k left in HBM:
 * BM25-only hits of bm25_top (the first min(fetch_k, count) positives) are appended with vector score 0.0, both score lists are
 * min-max normalised (the BM25 side over ALL n_docs scores), blended alpha * v + (1 - alpha) * b and stable-sorted descending;
 * the first top_k entries are returned (unused tail: UINT64_MAX / -inf).  Every f32 operation in the reference's order.
 * BM25 scores arrive SPARSE, as the host's Bm25Scorer::search produces them: per query `d_bm25_count[q]` positives (position, score),
 * sorted by score descending, ties by position ascending (Rust's stable sort, bm25.rs:118), rows `bm25_stride` entries apart; every
 * other passage scores 0.0.  compat_polarity != 0: the backend's DISTANCES enter the blend as the reference has it (the
 * worst ANN hit gets the largest vector term); 0: corrected, 1 - dist.  fetch_k <= 256. */
int leann_hybrid_rerank_device(const uint64_t *d_keys, const float *d_dists, const uint32_t *d_counts, size_t nq, size_t fetch_k,
                               const uint32_t *d_bm25_pos, const float *d_bm25_score, const uint32_t *d_bm25_count,
                               size_t bm25_stride, size_t n_docs, float alpha, int compat_polarity, size_t top_k,
                               uint64_t *d_out_keys, float *d_out_scores, uint32_t *d_out_counts, void *stream);

/* ---- sharded indexes (the reference has no counterpart: IndexSearcher owns one Box<dyn BackendSearcher>,
 * src/index/searcher.rs:68) --------------------------------------------------------------------------------------------------
 * One process, G devices: the composite handle leann_backend_open returns for a device list, or built here from rows / handles.
 * Every leann_backend_search* entry point works on it: the in-traversal allow-bitmap, exact filtered search and registered filters
 * (the bitmap is sliced per shard — interior shard boundaries are multiples of 64 —, every shard answers for its slice, lists are merged
 * by (dist, key): the answer of an unsharded index for exact searches, up to the order of entries with equal distances).  leann_backend_save writes one self-contained file
 * per shard, "<stem>.shard<g>of<G>.index" / ".diskann", which leann_backend_open with a list of G devices loads again; graph export is
 * per shard (leann_backend_shard).  Queries and results of the *_device calls live on the first device of the list.
 * One process per GPU: leann_sharded_attach joins this rank's shard to an RCCL communicator (librccl.so is resolved at run time);
 * a search is the local traversal + ONE ncclAllGather of the packed per-shard block {u64 keys | f32 dists | u32 counts} + the merge
 * kernel on every rank.  All ranks must issue the same calls with the same nq / top_k.
 * In both modes exchange + merge run on the handle's own stream, so the *_async form lets batch i + 1's traversal overlap batch
 * i's exchange (two result slots rotate: wait for ticket t before issuing t + 2). */
typedef struct leann_sharded leann_sharded;
int leann_sharded_open(const char *index_path_stem, int backend, size_t dims, const char *device_spec, leann_sharded **out);
/* the sub-index of shard g of a composite handle as an ordinary handle (borrowed — do not close it; it returns global keys):
 * graph_info / graph_export / feature_rows_export work on it.  leann_backend_shard_count: 0 for a plain handle. */
size_t leann_backend_shard_count(const leann_backend *h);
int leann_backend_shard(const leann_backend *h, size_t g, leann_backend **out);
/* d_vectors[g]: rows of shard g on devices[g] ([rows[g] x ld] f32, borrowed); keys are rebased by the prefix sums of rows[] */
int leann_sharded_build_device(int backend, const float *const *d_vectors, const size_t *rows, size_t n_shards, size_t dims,
                               size_t ld, size_t graph_degree, size_t complexity, const int *devices, leann_sharded **out);
/* existing handles (each built with its key_offset); take_ownership: leann_sharded_close closes them */
int leann_sharded_from_handles(leann_backend *const *shards, size_t n_shards, int take_ownership, leann_sharded **out);
/* wrap a one-process group as an ordinary leann_backend handle (closing that handle closes the group) */
int leann_sharded_as_backend(leann_sharded *s, leann_backend **out);
/* RCCL: rank 0 makes the id (128 bytes), the host distributes it by its own means, every rank attaches its shard (collective) */
int leann_rccl_get_unique_id(void *id128);
int leann_sharded_attach(leann_backend *local_shard, const void *unique_id128, int world, int rank, size_t total_rows,
                         leann_sharded **out);
/* d_stats: optional per-query counters [nq x 4], the layout of leann_backend_search_batch_device: one process — evaluations and
 * hops summed over the shards, visited-set level = the highest any shard needed; RCCL — the local shard's */
int leann_sharded_search_batch_device(const leann_sharded *s, const float *d_queries, size_t nq, size_t top_k, size_t complexity,
                                      uint64_t *d_keys, float *d_dists, uint32_t *d_counts, uint32_t *d_stats, void *stream);
int leann_sharded_search_batch_device_async(const leann_sharded *s, const float *d_queries, size_t nq, size_t top_k,
                                            size_t complexity, uint64_t *d_keys, float *d_dists, uint32_t *d_counts,
                                            uint32_t *d_stats, void *stream, uint64_t *ticket);
int leann_sharded_wait(const leann_sharded *s, uint64_t ticket, void *stream); /* `stream` waits for that exchange + merge */
size_t leann_sharded_len(const leann_sharded *s);    /* rows over all shards */
size_t leann_sharded_shards(const leann_sharded *s);
void leann_sharded_close(leann_sharded *s);

/* Environment knobs (LEANN_COALESCE, LEANN_HNSW_REFERENCE_EF, the LEANN_DEBUG_* test hooks) are read ONCE, when the library is first
 * used — no search call touches the environment.  A test that changes one of them afterwards calls this to have it read again. */
void leann_debug_reload_env(void);

/* raw device memory helpers so that non-torch hosts (the C++ CLI, ctypes tests) need no HIP binding */
int leann_device_count(int *n);
int leann_device_malloc(int device, size_t bytes, void **out);
int leann_device_free(void *p);
int leann_device_upload(void *d_dst, const void *h_src, size_t bytes);
int leann_device_download(void *h_dst, const void *d_src, size_t bytes);
int leann_device_sync(int device);

#ifdef __cplusplus
}
#endif
#endif
