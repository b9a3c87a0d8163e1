// internal.h — structs shared by api.hip / build.hip (not part of the C ABI).
#pragma once
#include "search.cuh"
#include "../../include/leann_backend.h"
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

struct Workspace { // staging of the host-pointer API; one per concurrent caller
    float *d_q = nullptr;
    uint64_t *d_keys = nullptr;
    float *d_dists = nullptr;
    uint32_t *d_counts = nullptr, *d_stats = nullptr;
    uint8_t *d_allow = nullptr; // filtered searches: staged allow-bitmap(s)
    size_t cap_q = 0, cap_keys = 0, cap_dists = 0, cap_counts = 0, cap_stats = 0, cap_allow = 0;
    // small calls (a single query is the reference's own call, traits.rs:16-21): one block of pinned, device-mapped host memory
    // {queries | keys | dists | counts | stats}: the kernels read the queries from it and write the results into it, so the call
    // is kernel launches + one stream synchronisation, with no copy engine in between
    unsigned char *pin = nullptr;
    size_t cap_pin = 0;
    hipStream_t stream = nullptr;
};

// Environment knobs of the SEARCH path.  The environment is read ONCE — when the library is first used — into this struct; a search
// call never calls getenv (it used to, five times per call: a 50-us call paid five environment scans, and getenv racing a setenv in a
// multi-threaded Rust host is undefined behaviour).  Tests that flip a diagnostic knob on a live handle call leann_debug_reload_env().
struct LeannKnobs {
    int hash_bits = 0;      // LEANN_DEBUG_HASH_BITS: force the LDS visited table to 2^bits slots (6..15; 0 = automatic)
    int nw = 0;             // LEANN_DEBUG_NW: waves per query (0 = by batch size)
    int gpool_bits = 0, gpool2_bits = 0; // LEANN_DEBUG_GPOOL_BITS / _GPOOL2_BITS (0 = default sizes)
    bool no_feat256 = false, no_zero_copy = false, no_emit = false, fused_v1 = false, no_list = false, no_tiled = false;
    bool force_remote = false;       // LEANN_DEBUG_FORCE_REMOTE: sharded searches treat EVERY shard as if it sat on another device (staging
                                     // buffers + peer copies, with source = destination device) — exercises that branch on a one-GPU box
    bool coalesce_off = false;       // LEANN_COALESCE=off|0
    bool hnsw_reference_ef = false;  // LEANN_HNSW_REFERENCE_EF=1: HNSW handles opened from now on search with ef = 64 whatever
                                     // `complexity` says, as the reference does (hnsw.rs:49 expansion_search: 64, :83 _complexity unused)
    unsigned long long stamp_buf = 0; // LEANN_STAMP_BUF (diagnostic builds only)
};
const LeannKnobs &leann_knobs();
extern "C" void leann_debug_reload_env(void);

struct Coalescer;
struct leann_sharded;
struct leann_backend {
    leann_sharded *sharded = nullptr; // composite handle (shard.hip): searches fan out to the sub-indexes; g holds only n and d
    int kind = LEANN_BACKEND_HNSW, device = 0;
    GraphView g{};
    uint64_t key_offset = 0;
    bool owns_rows = true;
    uint8_t *d_levels = nullptr;
    uint32_t efc = 64;
    float alpha = 1.2f;
    uint32_t fixed_ef = leann_knobs().hnsw_reference_ef ? 64u : 0u; // reference-exact mode, latched when the handle is made (HNSW only)
    bool nav_levels = false; // Vamana built with entry layers (build.hip); searches need nothing but g.max_level / g.entry
    uint64_t n_upper_lists = 0;
    std::mutex mu;
    std::vector<Workspace *> free_ws; // host-pointer API: one per concurrent caller
    // pooled HBM visited tables (search.cuh): shared by every stream, handed out by in-kernel locks
    unsigned long long *gpool = nullptr, *gpool2 = nullptr;
    uint32_t *gpool_lock = nullptr, *gpool_ctr = nullptr, *gpool2_lock = nullptr;
    uint32_t gpool_bits = GPOOL_BITS, gpool_tables = GPOOL_TABLES, gpool2_bits = 0, gpool2_tables = 0;
    leann_search_stats stats{};
    // recompute-on graph mode (g.feat_h != 0): encoder weights as f32 [feat_h x dims] for the query projection and
    // per-stream scratch for the projected queries
    float *Wf32 = nullptr;
    std::map<hipStream_t, std::pair<float *, size_t>> proj_scratch;
    std::shared_ptr<Coalescer> coalescer; // optional request coalescing for single-query callers (api.hip); swapped under `mu`
    // 0 = automatic (default): a leann_backend_search caller that finds another one in flight goes through a dispatcher created on the
    // spot, a lone caller is answered directly; 1 = configured by leann_backend_set_coalescing; 2 = switched off by it
    int coalesce_mode = 0;
    std::atomic<int> singles_in_flight{0};
};

// LEANN_LOG=error|warn|info|debug (default warn) -> stderr, "LEVEL leann_hip: message" (the reference logs through tracing, cli/mod.rs:38-43)
enum { LEANN_LOG_ERROR = 0, LEANN_LOG_WARN = 1, LEANN_LOG_INFO = 2, LEANN_LOG_DEBUG = 3 };
void leann_log(int level, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
// host-side graph arrays -> device index; exactly one of `vectors` (f32 rows [n x dims]) and `feat_rows` (recompute-on rows
// [n x row_bytes] + Wf32 [feat_h x dims]) is given.  Validates every array against n before anything is uploaded.
int leann_internal_from_host(int backend, size_t n, size_t dims, uint32_t M, uint32_t M0, uint32_t max_level, uint32_t entry,
                             const uint8_t *levels, const uint32_t *upper_off, const uint32_t *adj0, const uint32_t *adjU,
                             size_t n_upper_lists, const float *vectors, const unsigned char *feat_rows, uint32_t feat_h, uint32_t row_bytes,
                             const float *Wf32, int device, uint64_t key_offset, leann_backend **out);
int leann_internal_save_to(const leann_backend *h, const std::string &path);
int leann_internal_parse_device(const char *spec, int *device);
// sharded handles (shard.hip)
bool leann_internal_spec_is_sharded(const char *spec);
int leann_internal_open_sharded_backend(const char *stem, int backend, size_t dims, const char *spec, leann_backend **out);
size_t leann_internal_sharded_count(const leann_sharded *s);
leann_backend *leann_internal_sharded_shard(const leann_sharded *s, size_t g);
// A filter registered on the device (leann_backend_filter_create): the bitmap, the ascending list of allowed positions and its length.
// On a composite handle: one sub-filter per shard (its slice of the bitmap, on the shard's device); n / n_allowed are the totals.
struct leann_filter {
    int device = 0;
    size_t n = 0, n_allowed = 0;
    uint8_t *d_allow = nullptr;
    uint32_t *d_list = nullptr; // scratch-pool block (scan.hip), held for the filter's lifetime
    std::vector<leann_filter *> parts;
};
// how a sharded search filters: a bitmap over GLOBAL positions on the first device (sliced per shard at byte boundaries), or one
// registered sub-filter per shard; `exact`: scan the allowed rows of every shard instead of walking its graph
struct ShardFilterArgs {
    const uint8_t *d_allow = nullptr;
    size_t allow_stride = 0;
    const leann_filter *const *sub = nullptr;
    bool exact = false;
};
int leann_internal_sharded_search(leann_sharded *s, const float *d_queries, size_t nq, size_t top_k, size_t complexity, const ShardFilterArgs &fa,
                                  uint64_t *d_keys, float *d_dists, uint32_t *d_counts, uint32_t *d_stats, hipStream_t st, uint64_t *ticket);
int leann_internal_sharded_save(const leann_sharded *s, const char *index_path_stem);
uint64_t leann_internal_sharded_lo(const leann_sharded *s, size_t g);
int leann_internal_filtered_exact_list(const float *d_rows, size_t dims, size_t ld, const float *d_queries, size_t nq, size_t top_k,
                                       const uint32_t *d_list, size_t m, uint64_t key_offset, uint64_t *d_keys, float *d_dists,
                                       uint32_t *d_counts, hipStream_t st);
extern "C" void leann_sharded_close(leann_sharded *s);

int leann_internal_launch_search(leann_backend *h, SearchArgs a, hipStream_t st);
size_t leann_internal_effective_complexity(const leann_backend *h, size_t complexity);

// Candidate emission of the exhaustive searches (recompute_fstat.cuh, scan.hip): instead of writing an nq x rows score slab for a
// separate top-k pass, a scoring kernel compares each score with the query's running k-th best (fixed for the launch) and appends
// the few survivors (~k * rows / rows_seen per query) to a per-query list; leann_internal_fold_candidates merges the lists into the
// running best-k (ascending keys ~orderable(score) << 32 | position), publishes the new thresholds and resets the counters.
struct CandEmit {
    const float *thr;      // [slots] score of the running k-th best per query (+inf: query slot unused); null = the kernel writes its slab
    uint32_t *cnt;         // [slots] survivors appended so far (may exceed cap: the list then overflowed)
    uint64_t *list;        // [slots x cap] keys
    uint32_t cap;
    const uint8_t *allow;  // optional early filter over positions (recompute.rs:66-71)
    uint64_t pos0;         // position of the launch's first row
};
int leann_internal_fold_candidates(const CandEmit &em, uint32_t k, uint32_t nq, uint32_t slots, uint64_t *best, uint32_t *d_overflow,
                                   hipStream_t st);
int leann_internal_score(const float *X, size_t rows, size_t dims, size_t ld, const float *d_queries, size_t nq, size_t ldq, float *S,
                         hipStream_t st);
void leann_internal_free_graph(leann_backend *h);
size_t leann_internal_feat_file_row_bytes(const GraphView &g);
int leann_internal_feat_rows_to_host(const leann_backend *h, size_t r0, size_t rows, unsigned char *out);
std::string leann_internal_index_file(const char *stem, int backend);
std::string leann_internal_with_extension(const std::string &stem, const char *ext);
