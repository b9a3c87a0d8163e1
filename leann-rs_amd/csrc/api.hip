// api.hip — C ABI (include/leann_backend.h) over the gfx950 kernels: handle lifetime, index files,
// host<->HBM staging, kernel dispatch.  No CPU fallback exists in this library by design: every
// search / scan / build entry point runs HIP kernels or returns LEANN_ERR_DEVICE.
#include "common.cuh"
#include "search.cuh"
#include "../../include/leann_backend.h"
#include "internal.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <strings.h>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include <sys/stat.h>
#include <chrono>
#include <condition_variable>
#include <thread>

static thread_local char g_err[2048] = "";
extern "C" void leann_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char *leann_last_error(void) { return g_err; }
void leann_log(int level, const char *fmt, ...) {
    static const int threshold = [] {
        const char *e = getenv("LEANN_LOG");
        if (!e) return (int)LEANN_LOG_WARN;
        if (!strcasecmp(e, "error")) return (int)LEANN_LOG_ERROR;
        if (!strcasecmp(e, "info")) return (int)LEANN_LOG_INFO;
        if (!strcasecmp(e, "debug") || !strcasecmp(e, "trace")) return (int)LEANN_LOG_DEBUG;
        return (int)LEANN_LOG_WARN;
    }();
    if (level > threshold) return;
    static const char *names[] = {"ERROR", "WARN", "INFO", "DEBUG"};
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    fprintf(stderr, "%s leann_hip: %s\n", names[level < 0 ? 0 : (level > 3 ? 3 : level)], buf);
}
extern "C" const char *leann_version(void) { return "leann-rs_amd 0.1 (gfx950)"; }

// ---- environment knobs: read once (internal.h) -----------------------------------------------------
static LeannKnobs *read_knobs_from_env() {
    LeannKnobs *k = new LeannKnobs();
    auto num = [](const char *name, int lo, int hi) { const char *e = getenv(name); if (!e || !*e) return 0; int v = atoi(e); return v >= lo && v <= hi ? v : 0; };
    auto flag = [](const char *name) { const char *e = getenv(name); return e && *e && strcmp(e, "0") != 0; };
    k->hash_bits = num("LEANN_DEBUG_HASH_BITS", 6, 15);
    k->nw = num("LEANN_DEBUG_NW", 1, 64);
    k->gpool_bits = num("LEANN_DEBUG_GPOOL_BITS", 6, 31);
    k->gpool2_bits = num("LEANN_DEBUG_GPOOL2_BITS", 6, 31);
    k->no_feat256 = flag("LEANN_DEBUG_NO_FEAT256");
    k->no_zero_copy = flag("LEANN_DEBUG_NO_ZERO_COPY");
    k->no_emit = flag("LEANN_DEBUG_NO_EMIT");
    k->fused_v1 = flag("LEANN_DEBUG_FUSED_V1");
    k->no_list = flag("LEANN_RECOMPUTE_NO_LIST");
    k->no_tiled = flag("LEANN_RECOMPUTE_NO_TILED");
    k->hnsw_reference_ef = flag("LEANN_HNSW_REFERENCE_EF");
    k->force_remote = flag("LEANN_DEBUG_FORCE_REMOTE");
    if (const char *e = getenv("LEANN_COALESCE")) k->coalesce_off = !strcmp(e, "off") || !strcmp(e, "0");
    if (const char *e = getenv("LEANN_STAMP_BUF")) k->stamp_buf = strtoull(e, nullptr, 0);
    return k;
}
static std::atomic<const LeannKnobs *> g_knobs{nullptr};
const LeannKnobs &leann_knobs() {
    const LeannKnobs *k = g_knobs.load(std::memory_order_acquire);
    if (!k) {
        LeannKnobs *fresh = read_knobs_from_env();
        const LeannKnobs *expected = nullptr;
        if (g_knobs.compare_exchange_strong(expected, fresh, std::memory_order_acq_rel)) k = fresh;
        else { delete fresh; k = expected; }
    }
    return *k;
}
// test hook: re-read the environment (the previous struct is left alive on purpose: a search in flight may still be reading it)
extern "C" void leann_debug_reload_env(void) { g_knobs.store(read_knobs_from_env(), std::memory_order_release); }

// ---- raw device helpers ---------------------------------------------------------------------------
extern "C" int leann_device_count(int *n) {
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { c = 0; (void)hipGetLastError(); }
    *n = c;
    return LEANN_OK;
}
extern "C" int leann_device_malloc(int device, size_t bytes, void **out) {
    HIP_CHECK_RET(hipSetDevice(device));
    HIP_CHECK_RET(hipMalloc(out, bytes ? bytes : 16));
    return LEANN_OK;
}
extern "C" int leann_device_free(void *p) {
    if (p) HIP_CHECK_RET(hipFree(p));
    return LEANN_OK;
}
extern "C" int leann_device_upload(void *d, const void *h, size_t bytes) {
    HIP_CHECK_RET(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice));
    return LEANN_OK;
}
extern "C" int leann_device_download(void *h, const void *d, size_t bytes) {
    HIP_CHECK_RET(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost));
    return LEANN_OK;
}
extern "C" int leann_device_sync(int device) {
    HIP_CHECK_RET(hipSetDevice(device));
    HIP_CHECK_RET(hipDeviceSynchronize());
    return LEANN_OK;
}

extern "C" int leann_backend_set_coalescing(leann_backend *h, uint32_t wait_us, uint32_t max_batch);
// ---- handle (structs in internal.h) ---------------------------------------------------------------
static void ws_free(Workspace *w) {
    if (!w) return;
    (void)hipFree(w->d_q);
    (void)hipFree(w->d_keys);
    (void)hipFree(w->d_dists);
    (void)hipFree(w->d_counts);
    (void)hipFree(w->d_stats);
    (void)hipFree(w->d_allow);
    if (w->pin) (void)hipHostFree(w->pin);
    if (w->stream) (void)hipStreamDestroy(w->stream);
    delete w;
}
static uint32_t debug_bits(int knob, uint32_t lo, uint32_t hi, uint32_t dflt) { // test hooks: force tiny visited tables
    return knob >= (int)lo && knob <= (int)hi ? (uint32_t)knob : dflt;
}
// Pool 1: GPOOL_TABLES tables of 2^GPOOL_BITS slots (512 MB) for queries that outgrow their LDS table.  Pool 2, only when the
// index has more rows than a pool-1 table holds at 75 % load: a few tables of >= (n + 128) / 0.75 slots each (<= 1 GiB in all), so
// that no search can run out of visited-set space whatever the beam (search.cuh).
static int ensure_gpool(leann_backend *h) {
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->gpool) return LEANN_OK;
    unsigned long long *p = nullptr, *p2 = nullptr;
    uint32_t *lock = nullptr, *lock2 = nullptr;
    const uint32_t bits = debug_bits(leann_knobs().gpool_bits, 6, GPOOL_BITS, GPOOL_BITS), tables = GPOOL_TABLES;
    const size_t slots = (size_t)tables << bits;
    uint32_t bits2 = 0, tables2 = 0;
    const uint64_t need = h->g.n + 128;
    if (need > (1ull << bits) - (1ull << (bits - 2))) {
        bits2 = bits + 1;
        while (bits2 < 31 && (1ull << bits2) - (1ull << (bits2 - 2)) < need) bits2++;
        bits2 = debug_bits(leann_knobs().gpool2_bits, 6, 31, bits2);
        const uint64_t per = 8ull << bits2;
        tables2 = (uint32_t)std::min<uint64_t>(64, std::max<uint64_t>(2, (1ull << 30) / per));
    }
    auto fail = [&](const char *what) {
        (void)hipFree(p); (void)hipFree(p2); (void)hipFree(lock); (void)hipFree(lock2);
        leann_set_error("visited-table pool: %s failed: %s", what, hipGetErrorString(hipGetLastError()));
        return LEANN_ERR_DEVICE;
    };
    if (hipMalloc((void **)&p, slots * 8) != hipSuccess) return fail("hipMalloc");
    if (hipMalloc((void **)&lock, (tables + 4) * 4) != hipSuccess) return fail("hipMalloc");
    if (hipMemset(p, 0, slots * 8) != hipSuccess) return fail("hipMemset"); // generation 0 is never issued
    if (hipMemset(lock, 0, (tables + 4) * 4) != hipSuccess) return fail("hipMemset");
    if (tables2) {
        if (hipMalloc((void **)&p2, ((size_t)tables2 << bits2) * 8) != hipSuccess) return fail("hipMalloc");
        if (hipMalloc((void **)&lock2, tables2 * 4) != hipSuccess) return fail("hipMalloc");
        if (hipMemset(p2, 0, ((size_t)tables2 << bits2) * 8) != hipSuccess) return fail("hipMemset");
        if (hipMemset(lock2, 0, tables2 * 4) != hipSuccess) return fail("hipMemset");
    }
    // hipMemset on device memory returns before the fill has run (it is queued on the null stream), and searches run on NON-BLOCKING
    // streams, which do not wait for the null stream: without this wait the first launch could migrate queries into a table that was
    // still being zeroed under them (entries lost -> nodes visited twice -> results off the oracle's; seen once the migration came a
    // few microseconds earlier, tests/test_gpu_parity.py::test_visited_table_overflow_moves_to_hbm_pool).  Once per handle.
    if (hipDeviceSynchronize() != hipSuccess) return fail("hipDeviceSynchronize");
    h->gpool_lock = lock;
    h->gpool_ctr = lock + tables;
    h->gpool_bits = bits;
    h->gpool_tables = tables;
    h->gpool2 = p2;
    h->gpool2_lock = lock2;
    h->gpool2_bits = bits2;
    h->gpool2_tables = tables2;
    h->gpool = p;
    return LEANN_OK;
}
static void set_pool_args(const leann_backend *h, SearchArgs &a) {
    a.gpool = h->gpool; a.gpool_lock = h->gpool_lock; a.gpool_ctr = h->gpool_ctr;
    a.gpool_bits = h->gpool_bits; a.gpool_tables = h->gpool_tables;
    a.gpool2 = h->gpool2; a.gpool2_lock = h->gpool2_lock; a.gpool2_bits = h->gpool2_bits; a.gpool2_tables = h->gpool2_tables;
}

void leann_internal_free_graph(leann_backend *h) {
    if (h->owns_rows) (void)hipFree((void *)h->g.X);
    (void)hipFree((void *)h->g.norms);
    (void)hipFree((void *)h->g.adj0);
    (void)hipFree((void *)h->g.adjU);
    (void)hipFree((void *)h->g.upper_off);
    (void)hipFree(h->d_levels);
    (void)hipFree(h->gpool);
    (void)hipFree(h->gpool_lock);
    (void)hipFree(h->gpool2);
    (void)hipFree(h->gpool2_lock);
    (void)hipFree(h->Wf32);
    for (auto &kv : h->proj_scratch) (void)hipFree(kv.second.first);
}

extern "C" void leann_backend_close(leann_backend *h) {
    if (!h) return;
    leann_backend_set_coalescing(h, 0, 0); // stops the dispatcher thread, if any
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    for (auto *w : h->free_ws) ws_free(w);
    if (h->sharded) leann_sharded_close(h->sharded); // composite handle: the sub-indexes own the device memory
    else leann_internal_free_graph(h);
    delete h;
}
extern "C" size_t leann_backend_len(const leann_backend *h) { return h ? (size_t)h->g.n : 0; }
extern "C" size_t leann_backend_dims(const leann_backend *h) { return h ? (size_t)h->g.d : 0; }
extern "C" const float *leann_backend_device_rows(const leann_backend *h) { return h && !h->sharded ? h->g.X : nullptr; }
#define NOT_ON_SHARDED(h, what)                                                                                                    \
    do {                                                                                                                           \
        if ((h) && (h)->sharded) {                                                                                                 \
            leann_set_error("%s is not available on a sharded handle (use leann_backend_shard(h, g) for one shard's)", what);       \
            return LEANN_ERR_UNSUPPORTED;                                                                                          \
        }                                                                                                                          \
    } while (0)

extern "C" int leann_backend_stats(const leann_backend *hc, leann_search_stats *out, int reset) {
    leann_backend *h = const_cast<leann_backend *>(hc);
    if (!h || !out) { leann_set_error("leann_backend_stats: null argument"); return LEANN_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(h->mu);
    *out = h->stats;
    if (reset) h->stats = leann_search_stats{};
    return LEANN_OK;
}

// Reference-exact mode: the reference's HNSW searcher is opened with expansion_search = 64
// (src/backend/hnsw.rs:49) and drops the caller's `complexity` (`_complexity`, :83), so every `leann search --complexity N` runs at
// ef = max(64, top_k).  LEANN_HNSW_REFERENCE_EF=1 in the environment when the handle is made latches that behaviour into it (HNSW
// handles only; DiskANN honours complexity in the reference too, diskann.rs:54).  Default: complexity is honoured.
size_t leann_internal_effective_complexity(const leann_backend *h, size_t complexity) {
    return (h->fixed_ef && h->kind == LEANN_BACKEND_HNSW) ? (size_t)h->fixed_ef : complexity;
}

// ---- kernel dispatch --------------------------------------------------------------------------------
// LDS visited table: 4 workgroups per CU are register-limited anyway, so 32 KiB (8192 slots) per
// query is free; larger beams take 64 / 128 KiB.  A query that outgrows it moves to the HBM pool.
static uint32_t pick_hash_bits(uint32_t ef) {
    if (const int v = leann_knobs().hash_bits) return (uint32_t)v; // test hook: force tiny tables to exercise the HBM pool
    // measured on 10M x 768: ~20-25 distance evaluations per unit of ef on average, p99.9 ~ 50 x ef.
    // 4 workgroups per CU need <= 32 KiB tables; a 64 KiB table halves occupancy and throughput, so
    // beams up to 256 keep the 8 192-slot table and let the ~1 % heaviest queries migrate to HBM.
    uint32_t want = ef * 24u, b = 13;
    while ((1u << b) < want && b < 15) b++;
    return b;
}

template <typename K>
static int launch_one(K kernel, int nthreads, size_t lds, const GraphView &g, const SearchArgs &a, hipStream_t st) {
    if (lds > 64 * 1024)
        HIP_CHECK_RET(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(kernel, dim3(a.nq), dim3(nthreads), lds, st, g, a);
    HIP_CHECK_RET(hipGetLastError());
    return LEANN_OK;
}
static int search_lds_checked(const GraphView &g, const SearchArgs &a, size_t *lds, int nw) {
    *lds = search_lds_bytes(a.ef, std::max(g.M0, g.M), a.hash_bits, a.allow ? a.k : 0u, nw > 4 ? 2u : 1u);
    if (*lds > 160 * 1024) {
        leann_set_error("search: complexity %u needs %zu B of LDS per query (> 160 KiB)", a.ef, *lds);
        return LEANN_ERR_INVALID;
    }
    return LEANN_OK;
}

template <int T, int R, int NW>
static int launch_search_NW(const GraphView &g, const SearchArgs &a, hipStream_t st) {
    size_t lds;
    if (int rc = search_lds_checked(g, a, &lds, NW)) return rc;
    if (a.allow) return launch_one(beam_search_filtered_kernel<T, R, NW>, NW * 64, lds, g, a, st);
    if (a.q_rows) return launch_one(beam_search_kernel<T, R, NW, true>, NW * 64, lds, g, a, st);
    return launch_one(beam_search_kernel<T, R, NW, false>, NW * 64, lds, g, a, st);
}

// Waves per query: 4 for throughput batches (4 workgroups per CU hide each other's dependent hops);
// 16 for small batches, where the chip is mostly idle and the per-hop row fetch is the critical path —
// all ~40 new rows of a hop are then in flight at once (results are identical: same order, same sums).
template <int T, int R>
static int launch_search_T(const GraphView &g, const SearchArgs &a, hipStream_t st) {
    int nw = a.nq <= 384 ? 16 : a.nq <= 640 ? 8 : 4; // 10M x 768, ef = 56: 16 waves win up to 256 queries, 8 at 512, 4 from 768 on (scripts/exp/batch_sweep.py)
    if (const int v = leann_knobs().nw) nw = v;
    if (nw >= 16) return launch_search_NW<T, R, 16>(g, a, st);
    if (nw >= 8) return launch_search_NW<T, R, 8>(g, a, st);
    return launch_search_NW<T, R, 4>(g, a, st);
}

template <int T, int R>
static int launch_search_feat(const GraphView &g, const SearchArgs &a, hipStream_t st) {
    size_t lds;
    if (int rc = search_lds_checked(g, a, &lds, a.nq <= 512 ? 16 : 4)) return rc;
    if (T == 1 && g.feat_h == 256 && !leann_knobs().no_feat256) { // four rows per wave instruction
        if (a.nq <= 512) {
            if (a.allow) return launch_one(beam_search_feat256_filtered_kernel<1, 16>, 16 * 64, lds, g, a, st);
            return launch_one(beam_search_feat256_kernel<1, 16>, 16 * 64, lds, g, a, st);
        }
        if (a.allow) return launch_one(beam_search_feat256_filtered_kernel<LEANN_FEAT_G, 4>, 4 * 64, lds, g, a, st);
        return launch_one(beam_search_feat256_kernel<LEANN_FEAT_G, 4>, 4 * 64, lds, g, a, st);
    }
    if (a.nq <= 512) {
        if (a.allow) return launch_one(beam_search_feat_filtered_kernel<T, R, 16>, 16 * 64, lds, g, a, st);
        return launch_one(beam_search_feat_kernel<T, R, 16>, 16 * 64, lds, g, a, st);
    }
    if (a.allow) return launch_one(beam_search_feat_filtered_kernel<T, R, 4>, 4 * 64, lds, g, a, st);
    return launch_one(beam_search_feat_kernel<T, R, 4>, 4 * 64, lds, g, a, st);
}

// recompute-on mode: queries [nq x dims] -> g = W q [nq x feat_h] into the stream's scratch (f32 MFMA, k-ordered chains)
static int project_queries(leann_backend *h, const float *d_queries, size_t nq, hipStream_t st, const float **out) {
    float *G = nullptr;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        auto &sc = h->proj_scratch[st];
        const size_t need = nq * h->g.feat_h;
        if (sc.second < need) {
            (void)hipFree(sc.first);
            sc.first = nullptr;
            sc.second = 0;
            if (hipMalloc((void **)&sc.first, need * 4) != hipSuccess) { leann_set_error("hipMalloc(%zu) failed", need * 4); return LEANN_ERR_DEVICE; }
            sc.second = need;
        }
        G = sc.first;
    }
    int rc = leann_internal_score(h->Wf32, h->g.feat_h, h->g.d, h->g.d, d_queries, nq, h->g.d, G, st);
    *out = G;
    return rc;
}

int leann_internal_launch_search(leann_backend *h, SearchArgs a, hipStream_t st) {
    if (a.nq == 0) return LEANN_OK;
    if (h->g.feat_h) {
        if (a.q_rows) { leann_set_error("recompute-on index: construction searches are not supported"); return LEANN_ERR_UNSUPPORTED; }
        if (a.ef < a.k) a.ef = a.k;
        // 520-B rows make this mode latency- rather than bandwidth-bound: favour occupancy (16 KiB visited table ->
        // 6-8 workgroups per CU) for narrow beams; heavier queries migrate to the HBM pool
        a.hash_bits = leann_knobs().hash_bits ? pick_hash_bits(a.ef) : (a.ef <= 64 ? 12u : pick_hash_bits(a.ef));
        int rc = ensure_gpool(h);
        if (rc) return rc;
        set_pool_args(h, a);
        const float *G = nullptr;
        rc = project_queries(h, a.queries, a.nq, st, &G);
        if (rc) return rc;
        a.queries = G;
        a.ldq = h->g.feat_h;
        const int T = (int)((h->g.feat_h + 255) / 256);
        switch (T) {
            case 1: return launch_search_feat<1, LEANN_FEAT_R1>(h->g, a, st);
            case 2: return launch_search_feat<2, 6>(h->g, a, st);
            case 3: case 4: return launch_search_feat<4, 4>(h->g, a, st);
            default: leann_set_error("recompute-on index: feature width %u > 1024 not supported", h->g.feat_h); return LEANN_ERR_INVALID;
        }
    }
    if (a.ef < a.k) a.ef = a.k; // diskann.rs:54
    // Rows of up to 512 floats leave registers for 5-7 workgroups per CU where the 32 KiB visited table allows 4, and rows this short
    // do not hide a hop's dependent phases behind their own transfer: narrow beams take the 16 KiB table (the heaviest queries
    // move to the HBM pool).  10M rows, ef = 64: 128-d 2.41 -> 3.10 M queries/s, 256-d 2.16 -> 2.72 M, 384-d 1.66 -> 1.85 M, 512-d
    // unchanged; 768-d and wider are bound by HBM either way and keep the larger table (scripts/exp/dims_sweep.py).
    a.hash_bits = (h->g.ld <= 512 && a.ef <= 64 && !a.q_rows && !leann_knobs().hash_bits) ? 12u : pick_hash_bits(a.ef);
    int rc = ensure_gpool(h);
    if (rc) return rc;
    set_pool_args(h, a);
    const GraphView &g = h->g;
    int T = (int)((g.ld + 255) / 256);
    switch (T) {
        case 1: return launch_search_T<1, 4>(g, a, st);
        case 2: return launch_search_T<2, 4>(g, a, st);
        case 3: return launch_search_T<3, 4>(g, a, st);
        case 4: return launch_search_T<4, 3>(g, a, st);
        case 5: case 6: return launch_search_T<6, 2>(g, a, st);
        case 7: case 8: return launch_search_T<8, 2>(g, a, st);
        case 9: case 10: case 11: case 12: return launch_search_T<12, 1>(g, a, st); // 3 072-d: text-embedding-3-large (embedding/models.rs:113)
        case 13: case 14: case 15: case 16: return launch_search_T<16, 1>(g, a, st);
        default:
            leann_set_error("search: dims %u > 4096 not supported", g.d);
            return LEANN_ERR_INVALID;
    }
}

extern "C" int leann_backend_search_batch_device(const leann_backend *hc, const float *d_queries, size_t nq,
                                                 size_t top_k, size_t complexity, uint64_t *d_keys, float *d_dists,
                                                 uint32_t *d_counts, uint32_t *d_stats, void *stream) {
    return leann_backend_search_filtered_batch_device(hc, d_queries, nq, top_k, complexity, nullptr, 0, d_keys, d_dists,
                                                      d_counts, d_stats, stream);
}

// Filtered traversal: the allow-bitmap is evaluated inside the kernel instead of the
// reference's fetch_k = 5*top_k over-fetch + post-filter (src/index/searcher.rs:129-133,:190-194).
extern "C" int leann_backend_search_filtered_batch_device(const leann_backend *hc, const float *d_queries, size_t nq,
                                                          size_t top_k, size_t complexity, const uint8_t *d_allow,
                                                          size_t allow_stride, uint64_t *d_keys, float *d_dists,
                                                          uint32_t *d_counts, uint32_t *d_stats, void *stream) {
    leann_backend *h = const_cast<leann_backend *>(hc);
    if (!h || !d_queries || !d_keys || !d_dists || !d_counts || top_k == 0) {
        leann_set_error("leann_backend_search_batch_device: null/zero argument");
        return LEANN_ERR_INVALID;
    }
    if (d_allow && allow_stride && allow_stride < (h->g.n + 7) / 8) {
        leann_set_error("filtered search: allow_stride %zu is smaller than the %zu-byte bitmap", allow_stride, (size_t)(h->g.n + 7) / 8);
        return LEANN_ERR_INVALID;
    }
    if (nq == 0) return LEANN_OK;
    HIP_CHECK_RET(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    if (h->g.n == 0) {
        HIP_CHECK_RET(hipMemsetAsync(d_counts, 0, nq * 4, st));
        return LEANN_OK;
    }
    if (h->sharded) { // composite handle: fan out, gather, merge (shard.hip); d_stats stays [nq x 4]: summed over the shards
        ShardFilterArgs fa;
        fa.d_allow = d_allow; fa.allow_stride = allow_stride;
        return leann_internal_sharded_search(h->sharded, d_queries, nq, top_k, complexity, fa, d_keys, d_dists, d_counts, d_stats, st, nullptr);
    }
    SearchArgs a{};
    a.queries = d_queries;
    a.ldq = h->g.d;
    a.nq = (uint32_t)nq;
    a.k = (uint32_t)top_k;
    a.ef = (uint32_t)std::max(leann_internal_effective_complexity(h, complexity), top_k);
    a.target_level = 0;
    a.key_offset = h->key_offset;
    a.out_keys = d_keys;
    a.out_dists = d_dists;
    a.out_counts = d_counts;
    a.out_stats = d_stats;
    a.allow = d_allow;
    a.allow_stride = allow_stride;
#ifdef LEANN_STAMPS
    if (const unsigned long long e = leann_knobs().stamp_buf) { // diagnostic build: [nq x 8] u64 device buffer address in the environment
        a.out_nexp = reinterpret_cast<uint32_t *>(e);
        a.exp_cap = 0xFEED;
    }
#endif
    return leann_internal_launch_search(h, a, st);
}

// BackendSearcher::search batched over host pointers.
extern "C" int leann_backend_search_batch(const leann_backend *hc, const float *queries, size_t nq, size_t top_k,
                                          size_t complexity, uint64_t *keys, float *dists, uint32_t *counts) {
    return leann_backend_search_filtered_batch(hc, queries, nq, top_k, complexity, nullptr, 0, keys, dists, counts);
}

// ... with an optional allow-bitmap over positions (host memory; one shared bitmap when allow_stride == 0)
enum { FILTER_WALK = 0, FILTER_EXACT = 1, FILTER_AUTO = 2 };
static int search_filtered_batch_host(const leann_backend *hc, const float *queries, size_t nq, size_t top_k, size_t complexity,
                                      const uint8_t *allow, size_t allow_stride, uint64_t *keys, float *dists, uint32_t *counts, int mode,
                                      const leann_filter *flt = nullptr);
int leann_internal_compact_allow(const uint8_t *d_allow, size_t n, uint32_t **d_list, size_t *n_list, hipStream_t st);
void leann_internal_scratch_release(void *p);

// Registered filters: a server that answers many queries under the same metadata filter uploads and compacts the bitmap once
// (the host-pointer calls above re-send N/8 bytes and re-compact them for every query: 1.25 MB at 10M rows).
extern "C" int leann_backend_filter_create(const leann_backend *hc, const uint8_t *allow, leann_filter **out) {
    if (!hc || !allow || !out) {
        leann_set_error("leann_backend_filter_create: null argument");
        return LEANN_ERR_INVALID;
    }
    *out = nullptr;
    if (hc->sharded) { // composite handle: one sub-filter per shard — its slice of the bitmap (shard boundaries are multiples of 64), on its device
        const size_t G = leann_internal_sharded_count(hc->sharded);
        leann_filter *f = new leann_filter();
        f->device = hc->device;
        f->n = hc->g.n;
        for (size_t g = 0; g < G; g++) {
            leann_backend *sh = leann_internal_sharded_shard(hc->sharded, g);
            const uint64_t lo = leann_internal_sharded_lo(hc->sharded, g);
            leann_filter *part = nullptr;
            int rc = (!sh || (lo & 7)) ? (leann_set_error("registered filter: shard %zu does not start at a multiple of 8", g), (int)LEANN_ERR_UNSUPPORTED)
                                       : leann_backend_filter_create(sh, allow + lo / 8, &part);
            if (rc) { leann_backend_filter_free(f); return rc; }
            f->parts.push_back(part);
            f->n_allowed += part->n_allowed;
        }
        (void)hipSetDevice(hc->device);
        *out = f;
        return LEANN_OK;
    }
    if (hc->g.n >= (1ull << 32)) {
        leann_set_error("leann_backend_filter_create: the index has 2^32 rows or more");
        return LEANN_ERR_INVALID;
    }
    HIP_CHECK_RET(hipSetDevice(hc->device));
    leann_filter *f = new leann_filter();
    f->device = hc->device;
    f->n = hc->g.n;
    const size_t nbytes = std::max<size_t>((f->n + 7) / 8, 1);
    if (hipMalloc((void **)&f->d_allow, nbytes) != hipSuccess || hipMemcpy(f->d_allow, allow, (f->n + 7) / 8, hipMemcpyHostToDevice) != hipSuccess) {
        leann_set_error("leann_backend_filter_create: device allocation / copy of %zu bytes failed", nbytes);
        if (f->d_allow) (void)hipFree(f->d_allow);
        delete f;
        return LEANN_ERR_DEVICE;
    }
    int rc = f->n ? leann_internal_compact_allow(f->d_allow, f->n, &f->d_list, &f->n_allowed, nullptr) : LEANN_OK;
    if (rc != LEANN_OK) {
        (void)hipFree(f->d_allow);
        delete f;
        return rc;
    }
    *out = f;
    return LEANN_OK;
}
extern "C" size_t leann_backend_filter_count(const leann_filter *f) { return f ? f->n_allowed : 0; }
extern "C" void leann_backend_filter_free(leann_filter *f) {
    if (!f) return;
    if (!f->parts.empty()) {
        for (auto *p : f->parts) leann_backend_filter_free(p);
        delete f;
        return;
    }
    (void)hipSetDevice(f->device);
    (void)hipDeviceSynchronize(); // searches on any stream may still be reading the bitmap / the list
    leann_internal_scratch_release(f->d_list);
    if (f->d_allow) (void)hipFree(f->d_allow);
    delete f;
}
extern "C" int leann_backend_search_filter_batch(const leann_backend *hc, const float *queries, size_t nq, size_t top_k, size_t complexity,
                                                 const leann_filter *filter, int mode, uint64_t *keys, float *dists, uint32_t *counts) {
    if (!filter || !hc || filter->n != hc->g.n || filter->device != hc->device || mode < FILTER_WALK || mode > FILTER_AUTO ||
        filter->parts.empty() != (hc->sharded == nullptr)) { // (a filter registered on a sharded handle holds one sub-filter per shard, and only those)
        leann_set_error("leann_backend_search_filter_batch: null / foreign filter (made for %zu rows, the index has %zu) or bad mode %d",
                        filter ? filter->n : (size_t)0, hc ? (size_t)hc->g.n : (size_t)0, mode);
        return LEANN_ERR_INVALID;
    }
    return search_filtered_batch_host(hc, queries, nq, top_k, complexity, nullptr, 0, keys, dists, counts, mode, filter);
}

extern "C" int leann_backend_search_filtered_batch(const leann_backend *hc, const float *queries, size_t nq, size_t top_k,
                                                   size_t complexity, const uint8_t *allow, size_t allow_stride,
                                                   uint64_t *keys, float *dists, uint32_t *counts) {
    return search_filtered_batch_host(hc, queries, nq, top_k, complexity, allow, allow_stride, keys, dists, counts, FILTER_WALK);
}
// ... answered exactly: the allowed rows are compacted and scanned (scan.hip), no graph involved
extern "C" int leann_backend_search_filtered_exact_batch(const leann_backend *hc, const float *queries, size_t nq, size_t top_k,
                                                         const uint8_t *allow, size_t allow_stride, uint64_t *keys, float *dists,
                                                         uint32_t *counts) {
    if (!allow) {
        leann_set_error("leann_backend_search_filtered_exact_batch: null allow-bitmap");
        return LEANN_ERR_INVALID;
    }
    return search_filtered_batch_host(hc, queries, nq, top_k, 0, allow, allow_stride, keys, dists, counts, FILTER_EXACT);
}
int leann_internal_filtered_exact(const float *d_rows, size_t n, size_t dims, size_t ld, const float *d_queries, size_t nq, size_t top_k,
                                  const uint8_t *d_allow, size_t allow_stride, uint64_t key_offset, uint64_t *d_keys, float *d_dists,
                                  uint32_t *d_counts, hipStream_t st);
extern "C" int leann_backend_search_filtered_exact_batch_device(const leann_backend *hc, const float *d_queries, size_t nq, size_t top_k,
                                                                const uint8_t *d_allow, size_t allow_stride, uint64_t *d_keys,
                                                                float *d_dists, uint32_t *d_counts, void *stream) {
    leann_backend *h = const_cast<leann_backend *>(hc);
    if (!h || !d_queries || !d_keys || !d_dists || !d_counts || !d_allow || top_k == 0) {
        leann_set_error("leann_backend_search_filtered_exact_batch_device: null/zero argument");
        return LEANN_ERR_INVALID;
    }
    if (h->sharded) { // every shard scans its own allowed rows; lists merged by (dist, key)
        if (allow_stride && allow_stride < (h->g.n + 7) / 8) {
            leann_set_error("filtered search: allow_stride %zu is smaller than the %zu-byte bitmap", allow_stride, (size_t)(h->g.n + 7) / 8);
            return LEANN_ERR_INVALID;
        }
        if (nq == 0) return LEANN_OK;
        ShardFilterArgs fa;
        fa.d_allow = d_allow; fa.allow_stride = allow_stride; fa.exact = true;
        return leann_internal_sharded_search(h->sharded, d_queries, nq, top_k, 0, fa, d_keys, d_dists, d_counts, nullptr, (hipStream_t)stream, nullptr);
    }
    if (h->g.feat_h) {
        leann_set_error("exact filtered search needs stored vectors; this index recomputes them from features (use leann_recompute_search_batch_device with an allow mask)");
        return LEANN_ERR_UNSUPPORTED;
    }
    if (allow_stride && allow_stride < (h->g.n + 7) / 8) {
        leann_set_error("filtered search: allow_stride %zu is smaller than the %zu-byte bitmap", allow_stride, (size_t)(h->g.n + 7) / 8);
        return LEANN_ERR_INVALID;
    }
    if (nq == 0) return LEANN_OK;
    HIP_CHECK_RET(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    if (h->g.n == 0) {
        HIP_CHECK_RET(hipMemsetAsync(d_counts, 0, nq * 4, st));
        return LEANN_OK;
    }
    return leann_internal_filtered_exact(h->g.X, h->g.n, h->g.d, h->g.ld, d_queries, nq, top_k, d_allow, allow_stride, h->key_offset, d_keys,
                                         d_dists, d_counts, st);
}
static int search_filtered_batch_host_impl(const leann_backend *hc, const float *queries, size_t nq, size_t top_k, size_t complexity,
                                      const uint8_t *allow, size_t allow_stride, uint64_t *keys, float *dists, uint32_t *counts, int mode,
                                      const leann_filter *flt) {
    leann_backend *h = const_cast<leann_backend *>(hc);
    // exact scan of the allowed rows, or the walk with the filter inside?  FILTER_AUTO (registered filters only: the count is known):
    // exact up to 5 % of the rows / 64k rows for small batches, 1.5 % for large ones, when the index stores vectors
    bool exact = mode == FILTER_EXACT;
    const bool stored_vectors = hc && !(hc->sharded ? leann_internal_sharded_shard(hc->sharded, 0)->g.feat_h : hc->g.feat_h);
    if (mode == FILTER_AUTO && flt && hc && stored_vectors && top_k <= 1024) {
        const double frac = nq <= 64 ? 0.05 : 0.015;
        exact = flt->n_allowed <= std::max<size_t>((size_t)(frac * (double)hc->g.n), nq <= 64 ? 65536 : 0);
    }
    if (!h || !queries || !keys || !dists || !counts) {
        leann_set_error("leann_backend_search_batch: null argument");
        return LEANN_ERR_INVALID;
    }
    if (nq == 0 || top_k == 0) {
        for (size_t i = 0; i < nq; i++) counts[i] = 0;
        return LEANN_OK;
    }
    if (h->g.n == 0) {
        for (size_t i = 0; i < nq; i++) counts[i] = 0;
        return LEANN_OK;
    }
    HIP_CHECK_RET(hipSetDevice(h->device));
    Workspace *w = nullptr;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        if (!h->free_ws.empty()) { w = h->free_ws.back(); h->free_ws.pop_back(); }
    }
    if (!w) w = new Workspace();
    int rc = LEANN_OK;
    auto fail = [&](int code) { std::lock_guard<std::mutex> lk(h->mu); h->free_ws.push_back(w); return code; };
    if (!w->stream) {
        if (hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking) != hipSuccess) {
            leann_set_error("hipStreamCreate failed");
            return fail(LEANN_ERR_DEVICE);
        }
    }
    const size_t d = h->g.d, qf = nq * d, no = nq * top_k;
    const size_t ns = 1; // (a composite handle reduces its per-shard counters to [nq x 4] itself, shard.hip)
    auto grow = [&](void **p, size_t &cap, size_t need, size_t elt) -> int {
        if (need <= cap) return 0;
        (void)hipFree(*p);
        *p = nullptr;
        cap = 0;
        if (hipMalloc(p, need * elt) != hipSuccess) { leann_set_error("hipMalloc(%zu) failed", need * elt); return 1; }
        cap = need;
        return 0;
    };
    // Small call on a plain handle: zero-copy through the workspace's pinned block (internal.h).  Layout, 16-byte aligned pieces:
    const size_t o_keys = (qf * 4 + 15) & ~(size_t)15, o_dists = o_keys + no * 8, o_counts = (o_dists + no * 4 + 15) & ~(size_t)15,
                 o_stats = o_counts + ((nq * 4 + 15) & ~(size_t)15), pin_bytes = o_stats + nq * 16;
    bool zero_copy = !h->sharded && !exact && pin_bytes <= ((size_t)256 << 10) && !leann_knobs().no_zero_copy;
    if (zero_copy && w->cap_pin < pin_bytes) {
        if (w->pin) (void)hipHostFree(w->pin);
        w->pin = nullptr;
        w->cap_pin = 0;
        const size_t want = std::max<size_t>(pin_bytes, (size_t)64 << 10);
        if (hipHostMalloc((void **)&w->pin, want, hipHostMallocDefault) == hipSuccess) w->cap_pin = want;
        else { (void)hipGetLastError(); zero_copy = false; } // no pinned memory to be had: the staged path below
    }
    // (queries: read in place by up to 4 queries' workgroups — every wave of a query reads all of it, and mapped host memory is not
    // cached on the device; larger small calls copy them from the pinned block with one DMA)
    const bool zc_in = zero_copy && nq <= 4;
    if (zero_copy && !zc_in && grow((void **)&w->d_q, w->cap_q, qf, 4)) return fail(LEANN_ERR_DEVICE);
    if (!zero_copy && (grow((void **)&w->d_q, w->cap_q, qf, 4) || grow((void **)&w->d_keys, w->cap_keys, no, 8) ||
        grow((void **)&w->d_dists, w->cap_dists, no, 4) || grow((void **)&w->d_counts, w->cap_counts, nq, 4) ||
        grow((void **)&w->d_stats, w->cap_stats, ns * nq * 4, 4)))
        return fail(LEANN_ERR_DEVICE);
    hipStream_t st = w->stream;
    if (allow) {
        const size_t nbytes = (h->g.n + 7) / 8;
        if (allow_stride && allow_stride < nbytes) {
            leann_set_error("filtered search: allow_stride %zu is smaller than the %zu-byte bitmap", allow_stride, nbytes);
            return fail(LEANN_ERR_INVALID);
        }
        const size_t total = allow_stride ? allow_stride * nq : nbytes;
        void *pa = w->d_allow;
        if (grow(&pa, w->cap_allow, total, 1)) { w->d_allow = (uint8_t *)pa; return fail(LEANN_ERR_DEVICE); }
        w->d_allow = (uint8_t *)pa;
        if (hipMemcpyAsync(w->d_allow, allow, total, hipMemcpyHostToDevice, st) != hipSuccess) {
            leann_set_error("H2D copy of the allow-bitmap failed");
            return fail(LEANN_ERR_DEVICE);
        }
    }
    if (zero_copy) memcpy(w->pin, queries, qf * 4);
    if (!zc_in && hipMemcpyAsync(w->d_q, zero_copy ? reinterpret_cast<const float *>(w->pin) : queries, qf * 4, hipMemcpyHostToDevice, st) != hipSuccess) {
        leann_set_error("H2D copy of queries failed");
        return fail(LEANN_ERR_DEVICE);
    }
    SearchArgs a{};
    a.queries = zc_in ? reinterpret_cast<const float *>(w->pin) : w->d_q;
    a.ldq = (uint32_t)d;
    a.nq = (uint32_t)nq;
    a.k = (uint32_t)top_k;
    a.ef = (uint32_t)std::max(leann_internal_effective_complexity(h, complexity), top_k);
    a.key_offset = h->key_offset;
    a.out_keys = zero_copy ? reinterpret_cast<uint64_t *>(w->pin + o_keys) : w->d_keys;
    a.out_dists = zero_copy ? reinterpret_cast<float *>(w->pin + o_dists) : w->d_dists;
    a.out_counts = zero_copy ? reinterpret_cast<uint32_t *>(w->pin + o_counts) : w->d_counts;
    a.out_stats = zero_copy ? reinterpret_cast<uint32_t *>(w->pin + o_stats) : w->d_stats;
    a.allow = flt ? flt->d_allow : (allow ? w->d_allow : nullptr);
    a.allow_stride = flt ? 0 : allow_stride;
    if (h->sharded) { // composite handle: the same decision (exact / walk) for every shard, per-shard lists merged by (dist, key)
        ShardFilterArgs fa;
        fa.exact = exact;
        if (flt) fa.sub = flt->parts.data();
        else { fa.d_allow = a.allow; fa.allow_stride = a.allow_stride; }
        if (flt && flt->parts.size() != leann_internal_sharded_count(h->sharded)) {
            leann_set_error("registered filter was not made for this sharded handle");
            return fail(LEANN_ERR_INVALID);
        }
        rc = leann_internal_sharded_search(h->sharded, w->d_q, nq, top_k, complexity, fa, w->d_keys, w->d_dists, w->d_counts, w->d_stats, st, nullptr);
    } else if (exact && flt) {
        if (h->g.feat_h) {
            leann_set_error("exact filtered search needs stored vectors; this index recomputes them from features");
            return fail(LEANN_ERR_UNSUPPORTED);
        }
        if (hipMemsetAsync(w->d_stats, 0, nq * 16, st) != hipSuccess) return fail(LEANN_ERR_DEVICE);
        rc = leann_internal_filtered_exact_list(h->g.X, h->g.d, h->g.ld, w->d_q, nq, top_k, flt->d_list, flt->n_allowed, h->key_offset,
                                                w->d_keys, w->d_dists, w->d_counts, st);
    } else if (exact) {
        if (h->g.feat_h) {
            leann_set_error("exact filtered search needs stored vectors; this index recomputes them from features");
            return fail(LEANN_ERR_UNSUPPORTED);
        }
        if (hipMemsetAsync(w->d_stats, 0, nq * 16, st) != hipSuccess) return fail(LEANN_ERR_DEVICE);
        rc = leann_internal_filtered_exact(h->g.X, h->g.n, h->g.d, h->g.ld, w->d_q, nq, top_k, w->d_allow, allow_stride, h->key_offset,
                                           w->d_keys, w->d_dists, w->d_counts, st);
    } else {
        rc = leann_internal_launch_search(h, a, st);
    }
    if (rc) return fail(rc);
    std::vector<uint32_t> hstats(ns * nq * 4);
    if (zero_copy) {
        if (hipStreamSynchronize(st) != hipSuccess) {
            leann_set_error("search: device error: %s", hipGetErrorString(hipGetLastError()));
            return fail(LEANN_ERR_DEVICE);
        }
        memcpy(keys, w->pin + o_keys, no * 8);
        memcpy(dists, w->pin + o_dists, no * 4);
        memcpy(counts, w->pin + o_counts, nq * 4);
        memcpy(hstats.data(), w->pin + o_stats, nq * 16);
    } else
    if (hipMemcpyAsync(keys, w->d_keys, no * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(dists, w->d_dists, no * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(counts, w->d_counts, nq * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(hstats.data(), w->d_stats, ns * nq * 16, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
        leann_set_error("search: device error: %s", hipGetErrorString(hipGetLastError()));
        return fail(LEANN_ERR_DEVICE);
    }
    size_t n_lost = 0;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        const GraphView &gv = h->sharded ? leann_internal_sharded_shard(h->sharded, 0)->g : h->g; // (shards share one configuration)
        for (size_t i = 0; i < ns * nq; i++) {
            h->stats.n_dist_evals += hstats[i * 4 + 0];
            h->stats.n_hops_base += hstats[i * 4 + 1];
            h->stats.n_hops_upper += hstats[i * 4 + 2];
            h->stats.n_table_overflow += hstats[i * 4 + 3] ? 1 : 0;
            n_lost += hstats[i * 4 + 3] == 3;
            h->stats.algorithmic_bytes += (uint64_t)hstats[i * 4 + 0] * (gv.feat_h ? (gv.norms ? 2 * (size_t)gv.feat_h + 4 : (size_t)gv.row_bytes) : d * 4) + (uint64_t)hstats[i * 4 + 1] * gv.M0 * 4 +
                                          (uint64_t)hstats[i * 4 + 2] * gv.M * 4;
        }
        h->stats.n_queries += nq;
        h->free_ws.push_back(w);
    }
    if (n_lost) { // search.cuh: the last visited-table level filled up (not reachable with the pools ensure_gpool sizes)
        leann_set_error("search: %zu of %zu queries ran out of visited-set space at complexity %zu; lower the complexity", n_lost, nq,
                        (size_t)a.ef);
        return LEANN_ERR_OVERFLOW;
    }
    return LEANN_OK;
}

static int search_filtered_batch_host(const leann_backend *hc, const float *queries, size_t nq, size_t top_k, size_t complexity,
                                      const uint8_t *allow, size_t allow_stride, uint64_t *keys, float *dists, uint32_t *counts, int mode,
                                      const leann_filter *flt) {
    try { // std::bad_alloc and friends must not cross the C ABI (every host-pointer search entry point funnels through here)
        return search_filtered_batch_host_impl(hc, queries, nq, top_k, complexity, allow, allow_stride, keys, dists, counts, mode, flt);
    } catch (const std::exception &e) {
        leann_set_error("search: %s", e.what());
        return LEANN_ERR_DEVICE;
    }
}

// ---- request coalescing --------------------------------------------------------------
// The reference serves one query per call from many threads (cli/serve.rs:289-292).  One query is one
// workgroup: ~1 ms of latency and 1/1000 of the chip.  With coalescing enabled on a handle, concurrent
// leann_backend_search callers are gathered for up to `wait_us` (or `max_batch` queries) and answered by ONE
// batched launch, transparently to the callers — the Rust server needs no change.
struct PendingQuery {
    const float *q;
    size_t k, ef;
    uint64_t *keys;
    float *dists;
    size_t *n_out;
    int rc = 0;
    bool done = false, answered = false;
    std::string err;
};
struct Coalescer {
    leann_backend *h = nullptr;
    uint32_t wait_us = 200, max_batch = 4096;
    std::mutex mu;
    std::condition_variable cv_submit, cv_done;
    std::vector<PendingQuery *> queue;
    std::thread th;
    bool stop = false;
    uint64_t n_launches = 0, n_queries = 0;
    size_t recent_callers = 0; // size of the previous batch + the queue behind it when it completed

    void run() {
        (void)hipSetDevice(h->device);
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            // The window ends early once as many callers as were seen lately (the previous batch + what had queued up behind it) are
            // waiting again.  A batch of this size costs about what one query costs (the walk is latency-bound), so one full herd beats
            // two alternating halves (8 threads: 17 k against 12 k queries/s) and firing at once on whatever is there (ditto).
            cv_submit.wait(lk, [&] { return stop || !queue.empty(); });
            if (stop && queue.empty()) return;
            const size_t fire_at = std::min<size_t>(max_batch, std::max<size_t>(1, recent_callers));
            auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(wait_us);
            cv_submit.wait_until(lk, deadline, [&] { return stop || queue.size() >= fire_at; });
            std::vector<PendingQuery *> batch;
            batch.swap(queue);
            lk.unlock();
            // one launch per distinct (top_k, complexity)
            std::map<std::pair<size_t, size_t>, std::vector<PendingQuery *>> groups;
            for (auto *p : batch) groups[{p->k, p->ef}].push_back(p);
            const size_t d = h->g.d;
            try { // (an allocation failure in here must reach the callers as an error, not end the process from a library thread)
            for (auto &kv : groups) {
                auto &g = kv.second;
                const size_t nq = g.size(), k = kv.first.first;
                std::vector<float> Q(nq * d);
                for (size_t i = 0; i < nq; i++) memcpy(Q.data() + i * d, g[i]->q, d * 4);
                std::vector<uint64_t> keys(nq * std::max<size_t>(k, 1));
                std::vector<float> dists(nq * std::max<size_t>(k, 1));
                std::vector<uint32_t> counts(nq);
                int rc = leann_backend_search_batch(h, Q.data(), nq, k, kv.first.second, keys.data(), dists.data(), counts.data());
                std::string err = rc ? leann_last_error() : "";
                for (size_t i = 0; i < nq; i++) {
                    if (!rc) {
                        memcpy(g[i]->keys, keys.data() + i * k, counts[i] * 8);
                        memcpy(g[i]->dists, dists.data() + i * k, counts[i] * 4);
                        *g[i]->n_out = counts[i];
                    }
                    g[i]->rc = rc;
                    g[i]->err = err;
                    g[i]->answered = true;
                }
                n_launches++;
                n_queries += nq;
            }
            } catch (const std::exception &e) {
                for (auto *p : batch)
                    if (!p->answered) { p->rc = LEANN_ERR_DEVICE; p->err = std::string("request coalescing: ") + e.what(); *p->n_out = 0; }
            }
            lk.lock();
            recent_callers = batch.size() + queue.size();
            for (auto *p : batch) p->done = true;
            cv_done.notify_all();
        }
    }
};

// Safe against searches in flight: the handle's coalescer is swapped out under h->mu, the lock is RELEASED, and only then is the old
// dispatcher stopped and joined (it drains its queue first; its launches take h->mu themselves).  Callers hold a shared_ptr, so the
// object outlives every waiter; a caller that arrives after `stop` answers its query directly.
static std::shared_ptr<Coalescer> make_coalescer(leann_backend *h, uint32_t wait_us, uint32_t max_batch) {
    auto fresh = std::make_shared<Coalescer>();
    fresh->h = h;
    fresh->wait_us = wait_us;
    fresh->max_batch = max_batch ? max_batch : 4096;
    Coalescer *c = fresh.get();
    fresh->th = std::thread([c] { c->run(); });
    return fresh;
}
static void retire_coalescer(std::shared_ptr<Coalescer> old) {
    if (!old) return;
    { std::lock_guard<std::mutex> l2(old->mu); old->stop = true; }
    old->cv_submit.notify_all();
    old->th.join();
}
extern "C" int leann_backend_set_coalescing(leann_backend *h, uint32_t wait_us, uint32_t max_batch) {
    if (!h) { leann_set_error("leann_backend_set_coalescing: null handle"); return LEANN_ERR_INVALID; }
    std::shared_ptr<Coalescer> fresh, old;
    const bool off = wait_us == 0 && max_batch == 0;
    if (!off) fresh = make_coalescer(h, wait_us, max_batch);
    {
        std::lock_guard<std::mutex> lk(h->mu);
        old.swap(h->coalescer);
        h->coalescer = fresh;
        h->coalesce_mode = off ? 2 : 1;
    }
    retire_coalescer(old);
    return LEANN_OK;
}
// Automatic mode (the default; LEANN_COALESCE=off in the environment or leann_backend_set_coalescing(h, 0, 0) switch it off): the
// reference's server calls search from many threads without knowing about batches (cli/serve.rs:289-292).  The first caller that
// finds another single-query call in flight on the handle installs a dispatcher (50 us window or 64 queries for a query that arrives at an idle dispatcher; whatever
// queued up while a batch ran leaves at once) and queues behind it; a
// caller that is alone is answered directly, at the latency of one launch.
static std::shared_ptr<Coalescer> auto_coalescer(leann_backend *h) {
    if (leann_knobs().coalesce_off) return nullptr;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        if (h->coalesce_mode != 0) return h->coalescer;
        if (h->coalescer) return h->coalescer;
    }
    std::shared_ptr<Coalescer> fresh = make_coalescer(h, 50, 64), lost;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        if (h->coalesce_mode == 0 && !h->coalescer) { h->coalescer = fresh; return fresh; }
        lost = fresh; // somebody else installed one (or configured the handle) meanwhile
        fresh = h->coalescer;
    }
    retire_coalescer(lost);
    return fresh;
}
extern "C" int leann_backend_coalescing_stats(const leann_backend *hc, uint64_t *n_launches, uint64_t *n_queries) {
    leann_backend *h = const_cast<leann_backend *>(hc);
    std::shared_ptr<Coalescer> c;
    if (h) { std::lock_guard<std::mutex> lk(h->mu); c = h->coalescer; }
    if (!c) { leann_set_error("coalescing is not enabled on this handle"); return LEANN_ERR_INVALID; }
    std::lock_guard<std::mutex> l2(c->mu);
    if (n_launches) *n_launches = c->n_launches;
    if (n_queries) *n_queries = c->n_queries;
    return LEANN_OK;
}

extern "C" int leann_backend_search(const leann_backend *hc, const float *query, size_t top_k, size_t complexity,
                                    uint64_t *keys, float *dists, size_t *n_out) {
    if (!n_out) { leann_set_error("leann_backend_search: n_out is null"); return LEANN_ERR_INVALID; }
    leann_backend *h = const_cast<leann_backend *>(hc);
    std::shared_ptr<Coalescer> c;
    struct InFlight { // single-query calls on this handle right now (automatic coalescing)
        std::atomic<int> *a;
        int before;
        explicit InFlight(std::atomic<int> *p) : a(p), before(p ? p->fetch_add(1) : 0) {}
        ~InFlight() { if (a) a->fetch_sub(1); }
    } inflight(h ? &h->singles_in_flight : nullptr);
    if (h && query && keys && dists && top_k > 0 && h->g.n > 0) {
        int mode;
        { std::lock_guard<std::mutex> lk(h->mu); c = h->coalescer; mode = h->coalesce_mode; }
        if (!c && mode == 0 && inflight.before > 0) c = auto_coalescer(h);
    }
    if (c) {
        PendingQuery p{query, top_k, complexity, keys, dists, n_out};
        std::unique_lock<std::mutex> lk(c->mu);
        if (!c->stop) { // a stopped dispatcher takes no new work: fall through to the direct call
            c->queue.push_back(&p);
            c->cv_submit.notify_one();
            c->cv_done.wait(lk, [&] { return p.done; });
            if (p.rc) leann_set_error("%s", p.err.c_str());
            return p.rc;
        }
    }
    uint32_t cnt = 0;
    int rc = leann_backend_search_batch(h, query, 1, top_k, complexity, keys, dists, &cnt);
    *n_out = cnt;
    return rc;
}

// One filtered query (host bitmap of ceil(len/8) bytes, uploaded with the call; callers with a recurring filter keep the
// bitmap in HBM and use the *_device entry point).  Not coalesced.
extern "C" int leann_backend_search_filtered(const leann_backend *hc, const float *query, size_t top_k, size_t complexity,
                                             const uint8_t *allow, uint64_t *keys, float *dists, size_t *n_out) {
    if (!n_out) { leann_set_error("leann_backend_search_filtered: n_out is null"); return LEANN_ERR_INVALID; }
    if (!allow) return leann_backend_search(hc, query, top_k, complexity, keys, dists, n_out);
    uint32_t cnt = 0;
    int rc = leann_backend_search_filtered_batch(hc, query, 1, top_k, complexity, allow, 0, keys, dists, &cnt);
    *n_out = cnt;
    return rc;
}

// ---- in-memory construction from host arrays ----------------------------------------------------------
// Every array is checked against n BEFORE anything is uploaded: the traversal kernel indexes rows by neighbour id, upper lists by
// upper_off[node] + level - 1 and trusts both (search.cuh), so one bad entry in an index file would be an out-of-bounds read on the GPU.
static const char *validate_graph(size_t n, uint32_t M, uint32_t M0, uint32_t max_level, uint32_t entry, const uint8_t *levels,
                                  const uint32_t *upper_off, const uint32_t *adj0, const uint32_t *adjU, size_t n_upper_lists) {
    if (M == 0 || M0 == 0 || M > 64 || M0 > 64) return "graph degree outside [1, 64]";
    if (max_level > 15) return "max_level > 15";
    if (n == 0) return nullptr;
    if (entry >= n) return "entry point is not a row of the index";
    if (!levels && (max_level != 0 || n_upper_lists != 0)) return "upper levels without a level table";
    if (n_upper_lists && !adjU) return "upper lists missing";
    if (levels && levels[entry] < max_level) return "the entry point does not reach max_level";
    for (size_t v = 0; v < n; v++) {
        const uint32_t lv = levels ? levels[v] : 0;
        if (lv > 15) return "node level > 15";
        if (lv && (size_t)upper_off[v] + lv > n_upper_lists) return "upper_off + level runs past the upper lists";
        const uint32_t *l0 = adj0 + v * (size_t)M0;
        for (uint32_t j = 0; j < M0; j++)
            if (l0[j] != LEANN_EMPTY && l0[j] >= n) return "level-0 neighbour id >= n";
        for (uint32_t l = 1; l <= lv; l++) {
            const uint32_t *lu = adjU + ((size_t)upper_off[v] + l - 1) * M;
            for (uint32_t j = 0; j < M; j++) {
                const uint32_t e = lu[j];
                if (e == LEANN_EMPTY) continue;
                if (e >= n) return "upper-level neighbour id >= n";
                if (levels[e] < l) return "upper-level neighbour does not exist on that level";
            }
        }
    }
    return nullptr;
}

int leann_internal_from_host(int backend, size_t n, size_t dims, uint32_t M, uint32_t M0, uint32_t max_level, uint32_t entry,
                             const uint8_t *levels, const uint32_t *upper_off, const uint32_t *adj0, const uint32_t *adjU,
                             size_t n_upper_lists, const float *vectors, const unsigned char *feat_rows, uint32_t feat_h, uint32_t row_bytes,
                             const float *Wf32, int device, uint64_t key_offset, leann_backend **out) {
    const bool feat = feat_rows != nullptr;
    if (!out || dims == 0 || dims > 4096 || n >= (1ull << 31) || (n && ((!vectors && !feat) || !adj0 || !upper_off)) ||
        (backend != LEANN_BACKEND_HNSW && backend != LEANN_BACKEND_DISKANN) ||
        (feat && (!Wf32 || feat_h == 0 || (feat_h & 3) || feat_h > 1024 || row_bytes < 2 * feat_h + 4 || (row_bytes & 7)))) {
        leann_set_error("leann_backend_from_arrays: invalid arguments");
        return LEANN_ERR_INVALID;
    }
    if (const char *why = validate_graph(n, M, M0, max_level, entry, levels, upper_off, adj0, adjU, n_upper_lists)) {
        leann_set_error("leann_backend_from_arrays: inconsistent graph arrays: %s", why);
        return LEANN_ERR_FORMAT;
    }
    int ndev = 0;
    leann_device_count(&ndev);
    if (device < 0 || device >= ndev) {
        leann_set_error("HIP device %d not available (%d visible). This library has no CPU fallback.", device, ndev);
        return LEANN_ERR_DEVICE;
    }
    HIP_CHECK_RET(hipSetDevice(device));
    leann_backend *h = new leann_backend();
    h->kind = backend;
    h->device = device;
    h->key_offset = key_offset;
    h->g.n = n;
    h->g.d = (uint32_t)dims;
    h->g.ld = (uint32_t)((dims + 3) & ~(size_t)3);
    h->g.M = M;
    h->g.M0 = M0;
    h->g.max_level = max_level;
    h->g.entry = entry;
    h->n_upper_lists = n_upper_lists;
    // every allocation lands in the handle at once, so that one leann_backend_close frees whatever a failure leaves behind
    auto fail = [&](const char *what) {
        leann_set_error("leann_backend_from_arrays: %s failed: %s", what, hipGetErrorString(hipGetLastError()));
        leann_backend_close(h);
        return LEANN_ERR_DEVICE;
    };
    const size_t nn = std::max<size_t>(n, 1), nu = std::max<size_t>(n_upper_lists, 1), ld = h->g.ld;
    if (feat) {
        h->g.feat_h = feat_h;
        if (feat_h == 256) { // split device layout (GraphView::norms): 512-B rows of whole lines, norms beside them
            h->g.row_bytes = 512;
            if (hipMalloc((void **)&h->g.X, nn * 512) != hipSuccess || hipMalloc((void **)&h->g.norms, nn * 4) != hipSuccess) return fail("hipMalloc(rows)");
            const size_t slab = (size_t)1 << 20; // de-interleaved on the host, uploaded contiguously
            std::vector<unsigned char> fr(std::min(slab, nn) * 512);
            std::vector<float> nr(std::min(slab, nn));
            for (size_t s0 = 0; s0 < n; s0 += slab) {
                const size_t m = std::min(slab, n - s0);
                for (size_t i = 0; i < m; i++) {
                    memcpy(fr.data() + i * 512, feat_rows + (s0 + i) * row_bytes, 512);
                    memcpy(&nr[i], feat_rows + (s0 + i) * row_bytes + 512, 4);
                }
                if (hipMemcpy((unsigned char *)h->g.X + s0 * 512, fr.data(), m * 512, hipMemcpyHostToDevice) != hipSuccess ||
                    hipMemcpy((void *)(h->g.norms + s0), nr.data(), m * 4, hipMemcpyHostToDevice) != hipSuccess)
                    return fail("upload of the feature rows");
            }
        } else {
            h->g.row_bytes = row_bytes;
            if (hipMalloc((void **)&h->g.X, nn * row_bytes) != hipSuccess) return fail("hipMalloc(rows)");
            if (n && hipMemcpy((void *)h->g.X, feat_rows, n * (size_t)row_bytes, hipMemcpyHostToDevice) != hipSuccess) return fail("upload of the feature rows");
        }
        if (hipMalloc((void **)&h->Wf32, (size_t)feat_h * dims * 4) != hipSuccess) return fail("hipMalloc(weights)");
        if (hipMemcpy(h->Wf32, Wf32, (size_t)feat_h * dims * 4, hipMemcpyHostToDevice) != hipSuccess) return fail("upload of the weights");
    } else {
        if (hipMalloc((void **)&h->g.X, std::max<size_t>(n * ld, 4) * 4) != hipSuccess) return fail("hipMalloc(rows)");
        if (n) {
            if (ld == dims) { if (hipMemcpy((void *)h->g.X, vectors, n * dims * 4, hipMemcpyHostToDevice) != hipSuccess) return fail("upload of the rows"); }
            else if (hipMemset((void *)h->g.X, 0, n * ld * 4) != hipSuccess ||
                     hipMemcpy2D((void *)h->g.X, ld * 4, vectors, dims * 4, dims * 4, n, hipMemcpyHostToDevice) != hipSuccess) return fail("upload of the rows");
        }
    }
    if (hipMalloc((void **)&h->g.adj0, nn * M0 * 4) != hipSuccess || hipMalloc((void **)&h->g.adjU, nu * M * 4) != hipSuccess ||
        hipMalloc((void **)&h->g.upper_off, nn * 4) != hipSuccess || hipMalloc((void **)&h->d_levels, nn) != hipSuccess)
        return fail("hipMalloc(graph arrays)");
    if (hipMemset((void *)h->g.adjU, 0xFF, nu * M * 4) != hipSuccess) return fail("hipMemset");
    if (n) {
        if (hipMemcpy((void *)h->g.adj0, adj0, n * M0 * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy((void *)h->g.upper_off, upper_off, n * 4, hipMemcpyHostToDevice) != hipSuccess ||
            (levels ? hipMemcpy(h->d_levels, levels, n, hipMemcpyHostToDevice) : hipMemset(h->d_levels, 0, n)) != hipSuccess ||
            (n_upper_lists && hipMemcpy((void *)h->g.adjU, adjU, n_upper_lists * M * 4, hipMemcpyHostToDevice) != hipSuccess))
            return fail("upload of the graph arrays");
    }
    if (hipDeviceSynchronize() != hipSuccess) return fail("hipDeviceSynchronize"); // the fills above are null-stream work; searches run on non-blocking streams
    *out = h;
    return LEANN_OK;
}

extern "C" int leann_backend_from_arrays(int backend, const float *vectors, size_t n, size_t dims, uint32_t M,
                                         uint32_t M0, uint32_t max_level, uint32_t entry, const uint8_t *levels,
                                         const uint32_t *upper_off, const uint32_t *adj0, const uint32_t *adjU,
                                         size_t n_upper_lists, int device, uint64_t key_offset, leann_backend **out) {
    try {
        return leann_internal_from_host(backend, n, dims, M, M0, max_level, entry, levels, upper_off, adj0, adjU, n_upper_lists, vectors,
                                        nullptr, 0, 0, nullptr, device, key_offset, out);
    } catch (const std::exception &e) {
        leann_set_error("leann_backend_from_arrays: %s", e.what());
        return LEANN_ERR_IO;
    }
}

// A composite (sharded) handle holds G graphs, so graph_info / graph_export / feature_rows_export apply per shard: the sub-index of a
// shard is an ordinary handle (borrowed: it lives as long as the composite one; keys it returns are global positions).
extern "C" size_t leann_backend_shard_count(const leann_backend *h) { return h && h->sharded ? leann_internal_sharded_count(h->sharded) : 0; }
extern "C" int leann_backend_shard(const leann_backend *h, size_t g, leann_backend **out) {
    if (!h || !out) { leann_set_error("leann_backend_shard: null argument"); return LEANN_ERR_INVALID; }
    *out = nullptr;
    if (!h->sharded || g >= leann_internal_sharded_count(h->sharded) || !leann_internal_sharded_shard(h->sharded, g)) {
        leann_set_error("leann_backend_shard: %s", h->sharded ? "no such shard" : "not a sharded handle");
        return LEANN_ERR_INVALID;
    }
    *out = leann_internal_sharded_shard(h->sharded, g);
    return LEANN_OK;
}

extern "C" int leann_backend_graph_info(const leann_backend *h, uint64_t *info) {
    if (!h || !info) { leann_set_error("leann_backend_graph_info: null argument"); return LEANN_ERR_INVALID; }
    info[0] = h->g.n; info[1] = h->g.d; info[2] = h->g.ld; info[3] = h->g.M; info[4] = h->g.M0;
    info[5] = h->g.max_level; info[6] = h->g.entry; info[7] = h->n_upper_lists;
    return LEANN_OK;
}
extern "C" int leann_backend_graph_export(const leann_backend *h, uint8_t *levels, uint32_t *upper_off, uint32_t *adj0,
                                          uint32_t *adjU, float *vectors) {
    if (!h) { leann_set_error("leann_backend_graph_export: null handle"); return LEANN_ERR_INVALID; }
    NOT_ON_SHARDED(h, "graph export");
    HIP_CHECK_RET(hipSetDevice(h->device));
    HIP_CHECK_RET(hipDeviceSynchronize());
    const size_t n = h->g.n;
    if (n == 0) return LEANN_OK;
    if (levels) HIP_CHECK_RET(hipMemcpy(levels, h->d_levels, n, hipMemcpyDeviceToHost));
    if (upper_off) HIP_CHECK_RET(hipMemcpy(upper_off, h->g.upper_off, n * 4, hipMemcpyDeviceToHost));
    if (adj0) HIP_CHECK_RET(hipMemcpy(adj0, h->g.adj0, n * h->g.M0 * 4, hipMemcpyDeviceToHost));
    if (adjU && h->n_upper_lists)
        HIP_CHECK_RET(hipMemcpy(adjU, h->g.adjU, h->n_upper_lists * h->g.M * 4, hipMemcpyDeviceToHost));
    if (vectors && h->g.feat_h) { leann_set_error("graph_export: a recompute-on index holds no vectors"); return LEANN_ERR_UNSUPPORTED; }
    if (vectors)
        HIP_CHECK_RET(hipMemcpy2D(vectors, (size_t)h->g.d * 4, h->g.X, (size_t)h->g.ld * 4, (size_t)h->g.d * 4, n,
                                  hipMemcpyDeviceToHost));
    return LEANN_OK;
}

