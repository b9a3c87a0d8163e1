// shard.hip — the corpus partitioned across GPUs, behind the C ABI (north_star: "Shard the passage/vector corpus
// across the 8 GPUs of one node with RCCL all-gather of per-shard top-k candidates over xGMI").
//
// The reference has no sharding: IndexSearcher owns ONE Box<dyn BackendSearcher> (src/index/searcher.rs:68) and calls
// BackendSearcher::search on it (src/backend/traits.rs:16-21).  A sharded index is therefore one handle too:
//   * one process, G devices   leann_backend_open(stem, backend, dims, "0-7") / leann_sharded_build_device / _from_handles:
//                              a `leann_backend` whose searches fan the batch out to G sub-indexes (contiguous position ranges,
//                              own graph per shard, keys rebased by the range start), gather the per-shard {key, dist} lists on the
//                              first device by peer copies and run merge_topk_kernel there.  Every leann_backend_search* entry point
//                              works on it unchanged, so a Rust IndexSearcher uses 8 GPUs without knowing.
//   * one process per GPU      leann_sharded_attach(local shard, unique id, world, rank): local search, ONE ncclAllGather of the packed
//                              per-shard block {u64 keys | f32 dists | u32 counts} per batch (RCCL, resolved from librccl.so at run time),
//                              merge_topk_kernel on every rank.  This is the mode `bench.py --gpus N` and the driver's scaling run use.
// In both modes the exchange + merge run on the handle's own stream: leann_sharded_search_batch_device_async returns a ticket and
// the next batch's traversal overlaps the (latency-bound: 12 B per entry) exchange of this one; two result slots rotate.
// Merge order is (dist, key) — independent of the number of shards (tested for G in {1, 2, 4, 8}).
#include "common.cuh"
#include "search.cuh"
#include "../../include/leann_backend.h"
#include "internal.h"

#include <algorithm>
#include <cstring>
#include <dlfcn.h>
#include <string>
#include <sys/stat.h>
#include <thread>
#include <vector>

int leann_internal_merge_strided(const void *keys, const void *dists, const void *counts, size_t kstride, size_t dstride, size_t cstride,
                                 size_t n_shards, size_t nq, size_t k_in, size_t k_out, int descending, uint64_t *d_out_keys,
                                 float *d_out_dists, uint32_t *d_out_counts, hipStream_t st);
int leann_internal_load_own_file(const std::string &path, int backend, size_t dims, int device, leann_backend **out, std::string *why);

// ---- RCCL, resolved at run time (the library has no link-time dependency on it; one-process handles never load it) ---------------
namespace {
typedef struct { char internal[128]; } rcclUniqueId; // NCCL_UNIQUE_ID_BYTES
typedef void *rcclComm_t;
struct RcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(rcclUniqueId *) = nullptr;
    int (*CommInitRank)(rcclComm_t *, int, rcclUniqueId, int) = nullptr;
    int (*CommDestroy)(rcclComm_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, rcclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;
std::mutex g_rccl_mu;
const RcclApi *rccl_api() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.lib) return &g_rccl;
    const char *names[] = {getenv("LEANN_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
    void *lib = nullptr;
    for (const char *nm : names)
        if (nm && *nm && (lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!lib) { leann_set_error("RCCL is not available: dlopen(librccl.so) failed: %s", dlerror()); return nullptr; }
    RcclApi a;
    a.lib = lib;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(lib, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(lib, "ncclCommDestroy");
    a.AllGather = (decltype(a.AllGather))dlsym(lib, "ncclAllGather");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather || !a.GetErrorString) {
        leann_set_error("RCCL: librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather");
        return nullptr;
    }
    g_rccl = a;
    return &g_rccl;
}
constexpr int RCCL_UINT8 = 1; // ncclUint8
} // namespace

#define RCCL_CHECK_RET(api, expr)                                                                                        \
    do {                                                                                                                 \
        int _r = (expr);                                                                                                 \
        if (_r != 0) {                                                                                                   \
            leann_set_error("RCCL: %s failed: %s", #expr, (api)->GetErrorString(_r));                                    \
            return LEANN_ERR_DEVICE;                                                                                     \
        }                                                                                                                \
    } while (0)

// ---- the handle ------------------------------------------------------------------------------------------------------------
struct ShardDev {
    leann_backend *h = nullptr;
    int device = 0;
    uint64_t lo = 0;              // first global position of the shard (== h->key_offset)
    hipStream_t st = nullptr;     // traversal stream on the shard's device
    hipEvent_t ev[2] = {nullptr, nullptr};
    float *d_q = nullptr;         // staged queries (devices other than the first)
    unsigned char *d_out = nullptr, *d_allow = nullptr; // result block / staged allow-bitmaps (devices other than the first)
    uint32_t *d_stats = nullptr;  // per-query counters (devices other than the first)
    size_t cap_q = 0, cap_out = 0, cap_allow = 0, cap_stats = 0;
};
struct ShardSlot {
    unsigned char *gather = nullptr, *local = nullptr; // [G x block] on the first device; RCCL mode: this rank's own block
    uint32_t *stats = nullptr;                          // [G x nq x 4] (one-process mode)
    size_t cap_gather = 0, cap_local = 0, cap_stats = 0;
    hipEvent_t ev_q = nullptr, done = nullptr;
    bool used = false;
};
struct leann_sharded {
    bool rccl = false, owns_shards = true;
    std::vector<ShardDev> shards; // one-process mode: G; RCCL mode: 1 (the local shard)
    int primary = 0, world = 1, rank = 0;
    size_t total_rows = 0, dims = 0;
    hipStream_t xstream = nullptr;
    ShardSlot slots[2];
    uint64_t next_ticket = 0;
    std::mutex mu;
    rcclComm_t comm = nullptr;
};

static size_t block_bytes(size_t nq, size_t k) { return (nq * k * 12 + nq * 4 + 15) & ~(size_t)15; }
// per-query counters of a composite handle: the caller's d_stats is [nq x 4] like a plain handle's (include/leann_backend.h) —
// evaluations and hops summed over the shards, the visited-set level the highest any shard needed
__global__ void reduce_shard_stats_kernel(const uint32_t *__restrict__ per_shard /* [G x nq x 4] */, uint32_t G, uint32_t nq,
                                          uint32_t *__restrict__ out /* [nq x 4] */) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq * 4) return;
    uint32_t acc = 0;
    for (uint32_t g = 0; g < G; g++) {
        const uint32_t v = per_shard[(size_t)g * nq * 4 + i];
        acc = (i & 3) == 3 ? max(acc, v) : acc + v;
    }
    out[i] = acc;
}
static int grow_dev(void **p, size_t *cap, size_t bytes) {
    if (bytes <= *cap) return LEANN_OK;
    (void)hipFree(*p); // (synchronises the device: nothing still reads the old block)
    *p = nullptr;
    *cap = 0;
    if (hipMalloc(p, bytes) != hipSuccess) { leann_set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(hipGetLastError())); return LEANN_ERR_DEVICE; }
    *cap = bytes;
    return LEANN_OK;
}

size_t leann_internal_sharded_count(const leann_sharded *s) { return s ? (s->rccl ? (size_t)s->world : s->shards.size()) : 0; }
leann_backend *leann_internal_sharded_shard(const leann_sharded *s, size_t g) { return s && g < s->shards.size() ? s->shards[g].h : nullptr; }
int leann_internal_sharded_primary(const leann_sharded *s) { return s ? s->primary : 0; }

static int init_common(leann_sharded *s) {
    HIP_CHECK_RET(hipSetDevice(s->primary));
    HIP_CHECK_RET(hipStreamCreateWithFlags(&s->xstream, hipStreamNonBlocking));
    for (auto &sl : s->slots) {
        HIP_CHECK_RET(hipEventCreateWithFlags(&sl.ev_q, hipEventDisableTiming));
        HIP_CHECK_RET(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    }
    for (auto &sd : s->shards) {
        HIP_CHECK_RET(hipSetDevice(sd.device));
        HIP_CHECK_RET(hipStreamCreateWithFlags(&sd.st, hipStreamNonBlocking));
        for (auto &e : sd.ev) HIP_CHECK_RET(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    HIP_CHECK_RET(hipSetDevice(s->primary));
    return LEANN_OK;
}

extern "C" void leann_sharded_close(leann_sharded *s) {
    if (!s) return;
    for (auto &sd : s->shards) {
        (void)hipSetDevice(sd.device);
        (void)hipDeviceSynchronize();
    }
    (void)hipSetDevice(s->primary);
    (void)hipDeviceSynchronize();
    if (s->comm) {
        if (const RcclApi *api = rccl_api()) (void)api->CommDestroy(s->comm);
    }
    for (auto &sl : s->slots) {
        (void)hipFree(sl.gather); (void)hipFree(sl.local); (void)hipFree(sl.stats);
        if (sl.ev_q) (void)hipEventDestroy(sl.ev_q);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    if (s->xstream) (void)hipStreamDestroy(s->xstream);
    for (auto &sd : s->shards) {
        (void)hipSetDevice(sd.device);
        (void)hipFree(sd.d_q); (void)hipFree(sd.d_out); (void)hipFree(sd.d_allow); (void)hipFree(sd.d_stats);
        for (auto &e : sd.ev) if (e) (void)hipEventDestroy(e);
        if (sd.st) (void)hipStreamDestroy(sd.st);
        if (s->owns_shards && sd.h) leann_backend_close(sd.h);
    }
    (void)hipSetDevice(s->primary);
    delete s;
}

// ---- one process, G shards ---------------------------------------------------------------------------------------------------
extern "C" int leann_sharded_from_handles(leann_backend *const *shards, size_t n_shards, int take_ownership, leann_sharded **out) {
    if (!shards || !out || n_shards == 0 || n_shards > 64) { leann_set_error("leann_sharded_from_handles: invalid arguments"); return LEANN_ERR_INVALID; }
    *out = nullptr;
    for (size_t g = 0; g < n_shards; g++) {
        if (!shards[g] || shards[g]->sharded || shards[g]->g.d != shards[0]->g.d || shards[g]->kind != shards[0]->kind ||
            (shards[g]->g.feat_h != 0) != (shards[0]->g.feat_h != 0)) {
            leann_set_error("leann_sharded_from_handles: shard %zu is null, itself sharded, or differs from shard 0 in kind / dimensions", g);
            return LEANN_ERR_INVALID;
        }
    }
    leann_sharded *s = new leann_sharded();
    s->owns_shards = take_ownership != 0;
    s->primary = shards[0]->device;
    s->dims = shards[0]->g.d;
    for (size_t g = 0; g < n_shards; g++) {
        ShardDev sd;
        sd.h = shards[g];
        sd.device = shards[g]->device;
        sd.lo = shards[g]->key_offset;
        s->total_rows += shards[g]->g.n;
        s->shards.push_back(sd);
    }
    int rc = init_common(s);
    if (rc) { s->owns_shards = false; leann_sharded_close(s); return rc; }
    *out = s;
    return LEANN_OK;
}

static int parse_device_list(const char *spec, std::vector<int> *out) { // "0-7", "0,1,2,3", "0,0,0,0" (simulated shards on one device)
    out->clear();
    std::string s = spec ? spec : "";
    size_t pos = 0;
    while (pos <= s.size()) {
        size_t comma = s.find(',', pos);
        std::string tok = s.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
        size_t dash = tok.find('-');
        char *e1 = nullptr, *e2 = nullptr;
        long a = strtol(tok.c_str(), &e1, 10), b = a;
        bool ok = e1 != tok.c_str();
        if (ok && dash != std::string::npos) { b = strtol(tok.c_str() + dash + 1, &e2, 10); ok = e1 == tok.c_str() + dash && e2 != tok.c_str() + dash + 1 && *e2 == 0; }
        else if (ok) ok = *e1 == 0;
        if (!ok || a < 0 || b < a || b > 1023) { leann_set_error("device_spec \"%s\": expected ordinals, lists or ranges such as \"0\", \"0,1,2,3\", \"0-7\"", s.c_str()); return LEANN_ERR_INVALID; }
        for (long v = a; v <= b; v++) out->push_back((int)v);
        if (comma == std::string::npos) break;
        pos = comma + 1;
    }
    if (out->empty() || out->size() > 64) { leann_set_error("device_spec \"%s\": between 1 and 64 shards", s.c_str()); return LEANN_ERR_INVALID; }
    return LEANN_OK;
}
bool leann_internal_spec_is_sharded(const char *spec) { return spec && (strchr(spec, ',') || strchr(spec, '-')); }

// contiguous ranges [lo_g, lo_{g+1}); interior boundaries are multiples of 64 so that an allow-bitmap slices at byte boundaries
static uint64_t shard_lo(uint64_t n, size_t G, size_t g) { return g == 0 ? 0 : (g >= G ? n : ((n * g) / G) & ~(uint64_t)63); }

extern "C" int leann_sharded_build_device(int backend, const float *const *d_vectors, const size_t *rows, size_t n_shards, size_t dims,
                                          size_t ld, size_t graph_degree, size_t complexity, const int *devices, leann_sharded **out) {
    if (!d_vectors || !rows || !devices || !out || n_shards == 0 || n_shards > 64) {
        leann_set_error("leann_sharded_build_device: invalid arguments");
        return LEANN_ERR_INVALID;
    }
    *out = nullptr;
    std::vector<leann_backend *> hs(n_shards, nullptr);
    std::vector<int> rcs(n_shards, LEANN_OK);
    std::vector<std::string> errs(n_shards);
    std::vector<uint64_t> lo(n_shards, 0);
    for (size_t g = 1; g < n_shards; g++) lo[g] = lo[g - 1] + rows[g - 1];
    // one builder thread per shard: shards on different devices build concurrently (leann_last_error is thread-local: carried over)
    std::vector<std::thread> th;
    for (size_t g = 0; g < n_shards; g++)
        th.emplace_back([&, g] {
            rcs[g] = leann_backend_build_device(backend, d_vectors[g], rows[g], dims, ld, graph_degree, complexity, devices[g], lo[g], 0, &hs[g]);
            if (rcs[g]) errs[g] = leann_last_error();
        });
    for (auto &t : th) t.join();
    for (size_t g = 0; g < n_shards; g++)
        if (rcs[g]) {
            for (auto *h : hs) if (h) leann_backend_close(h);
            leann_set_error("shard %zu: %s", g, errs[g].c_str());
            return rcs[g];
        }
    int rc = leann_sharded_from_handles(hs.data(), n_shards, 1, out);
    if (rc) for (auto *h : hs) leann_backend_close(h);
    return rc;
}

// Rows of a sharded open come from "<stem>.embeddings" (the reference builder's file, src/index/embeddings.rs) or, failing that, from
// the rows inside this library's own version-1 index file.  Each shard's graph is cached as "<stem>.shard<g>of<G>.gpu.index".
// "<stem>.shard<g>of<G>.index": a shard saved by leann_backend_save (self-contained); "....gpu.index": a graph cache derived from the row source
static std::string shard_file(const char *stem, size_t g, size_t G, int backend, bool cache) {
    char tag[80];
    snprintf(tag, sizeof tag, "shard%zuof%zu.%s%s", g, G, cache ? "gpu." : "", backend == LEANN_BACKEND_DISKANN ? "diskann" : "index");
    return leann_internal_with_extension(stem, tag);
}
// This library's own version-1 index file comes FIRST: `leann update` appends to the index (leann_backend_add rewrites the file with
// n_old + n rows) but never to "<stem>.embeddings" — only the builder writes that (src/index/builder.rs:105-113, src/cli/update.rs:169-223)
// — so after an update the embeddings file is the stale one, and a sharded open that preferred it served an index without the new
// passages.  The embeddings file is the source for foreign (stock usearch / diskann-rs) index files only.
static int rows_source(const char *stem, int backend, size_t dims, std::string *path, uint64_t *offset, size_t *n) {
    const std::string emb = leann_internal_with_extension(stem, "embeddings");
    const std::string own = leann_internal_index_file(stem, backend);
    struct stat st{};
    size_t emb_rows = 0;
    const bool have_emb = stat(emb.c_str(), &st) == 0 && dims && st.st_size > 0 && (uint64_t)st.st_size % (dims * 4) == 0;
    if (have_emb) emb_rows = (size_t)st.st_size / (dims * 4);
    FILE *f = fopen(own.c_str(), "rb");
    if (f) {
        unsigned char hd[128];
        const bool ok = fread(hd, 1, 128, f) == 128 && !memcmp(hd, "LEANNGX1", 8);
        fclose(f);
        uint32_t version = 0, d = 0, M = 0, M0 = 0;
        uint64_t nn = 0, nul = 0;
        if (ok) { memcpy(&version, hd + 8, 4); memcpy(&nn, hd + 16, 8); memcpy(&d, hd + 24, 4); memcpy(&M, hd + 28, 4); memcpy(&M0, hd + 32, 4); memcpy(&nul, hd + 56, 8); }
        if (ok && version == 1 && (!dims || d == dims) && stat(own.c_str(), &st) == 0 &&
            (uint64_t)st.st_size == 128 + nn + 4 * nn + 4 * nn * M0 + 4 * nul * M + 4 * nn * d) {
            if (have_emb && emb_rows != nn)
                leann_log(LEANN_LOG_WARN, "%s holds %zu rows, %s holds %llu: partitioning the index file's rows (the embeddings file is not updated by `leann update`)",
                          emb.c_str(), emb_rows, own.c_str(), (unsigned long long)nn);
            *path = own; *offset = 128 + nn + 4 * nn + 4 * nn * M0 + 4 * nul * M; *n = (size_t)nn;
            return LEANN_OK;
        }
        if (!have_emb) {
            leann_set_error("sharded open: %s holds no rows to partition (foreign, recompute-on or corrupt file) and %s is missing", own.c_str(), emb.c_str());
            return LEANN_ERR_FORMAT;
        }
    }
    if (have_emb) { *path = emb; *offset = 0; *n = emb_rows; return LEANN_OK; }
    leann_set_error("sharded open: neither %s nor %s exists", emb.c_str(), own.c_str());
    return LEANN_ERR_NOT_FOUND;
}

// shards written by leann_backend_save on a composite handle: all G files present and loadable -> the index, no rebuild
static int saved_shards(const char *stem, int backend, size_t dims, const std::vector<int> &devs, std::vector<leann_backend *> *hs) {
    const size_t G = devs.size();
    struct stat st{}, so{};
    // an index file written AFTER the shards (leann_backend_build / leann_backend_add on the same stem, i.e. `leann build --force` or
    // `leann update`) makes them stale: they are ignored, the rows are partitioned afresh
    const bool have_own = stat(leann_internal_index_file(stem, backend).c_str(), &so) == 0;
    for (size_t g = 0; g < G; g++) {
        if (stat(shard_file(stem, g, G, backend, false).c_str(), &st) != 0) return LEANN_ERR_NOT_FOUND;
        if (have_own && so.st_mtime > st.st_mtime) {
            leann_log(LEANN_LOG_WARN, "ignoring saved shard files of %s: the index file is newer", stem);
            return LEANN_ERR_NOT_FOUND;
        }
    }
    uint64_t lo = 0;
    for (size_t g = 0; g < G; g++) {
        std::string why;
        leann_backend *h = nullptr;
        int rc = leann_internal_load_own_file(shard_file(stem, g, G, backend, false), backend, dims, devs[g], &h, &why);
        if (rc) {
            for (auto *p : *hs) if (p) leann_backend_close(p);
            hs->assign(G, nullptr);
            leann_log(LEANN_LOG_WARN, "ignoring saved shard files of %s: shard %zu: %s", stem, g, why.empty() ? leann_last_error() : why.c_str());
            return rc;
        }
        h->key_offset = lo;
        lo += h->g.n;
        (*hs)[g] = h;
    }
    return LEANN_OK;
}

static int open_one_shard(const std::string &src, uint64_t src_off, const std::string &cache, time_t src_mtime, int backend, size_t dims,
                          uint64_t lo, size_t rows, int device, leann_backend **out) {
    struct stat sc{};
    if (stat(cache.c_str(), &sc) == 0 && sc.st_mtime >= src_mtime) {
        std::string why;
        int rc = leann_internal_load_own_file(cache, backend, dims, device, out, &why);
        if (rc == LEANN_OK && (*out)->g.n == rows) { (*out)->key_offset = lo; return LEANN_OK; }
        if (rc == LEANN_OK) { leann_backend_close(*out); *out = nullptr; }
        leann_log(LEANN_LOG_WARN, "ignoring stale or unreadable shard cache %s", cache.c_str());
    }
    int ndev = 0;
    leann_device_count(&ndev);
    if (device < 0 || device >= ndev) { leann_set_error("HIP device %d not available (%d visible). This library has no CPU fallback.", device, ndev); return LEANN_ERR_DEVICE; }
    HIP_CHECK_RET(hipSetDevice(device));
    const size_t ld = (dims + 3) & ~(size_t)3;
    FILE *f = fopen(src.c_str(), "rb");
    if (!f || fseeko(f, (off_t)(src_off + lo * dims * 4), SEEK_SET) != 0) { if (f) fclose(f); leann_set_error("cannot read %s", src.c_str()); return LEANN_ERR_IO; }
    float *dX = nullptr;
    if (hipMalloc((void **)&dX, std::max<size_t>(rows * ld, 4) * 4) != hipSuccess) { fclose(f); leann_set_error("hipMalloc for shard rows failed"); return LEANN_ERR_DEVICE; }
    const size_t slab_rows = std::max<size_t>(1, ((size_t)256 << 20) / (dims * 4));
    std::vector<float> slab(std::min(slab_rows, std::max<size_t>(rows, 1)) * dims);
    bool ok = ld == dims || hipMemset(dX, 0, rows * ld * 4) == hipSuccess;
    for (size_t r0 = 0; ok && r0 < rows; r0 += slab_rows) {
        const size_t nr = std::min(slab_rows, rows - r0);
        ok = fread(slab.data(), dims * 4, nr, f) == nr &&
             hipMemcpy2D(dX + r0 * ld, ld * 4, slab.data(), dims * 4, dims * 4, nr, hipMemcpyHostToDevice) == hipSuccess;
    }
    fclose(f);
    if (!ok) { (void)hipFree(dX); leann_set_error("reading rows [%llu, +%zu) of %s failed", (unsigned long long)lo, rows, src.c_str()); return LEANN_ERR_IO; }
    size_t degree = backend == LEANN_BACKEND_HNSW ? 32 : 64, complexity = 128;
    if (const char *e = getenv("LEANN_REBUILD_DEGREE")) degree = (size_t)atoi(e);
    if (const char *e = getenv("LEANN_REBUILD_COMPLEXITY")) complexity = (size_t)atoi(e);
    int rc = leann_backend_build_device(backend, dX, rows, dims, ld, degree, complexity, device, lo, 0, out);
    if (rc) { (void)hipFree(dX); return rc; }
    (*out)->owns_rows = true;
    if (leann_internal_save_to(*out, cache) != LEANN_OK)
        leann_log(LEANN_LOG_WARN, "could not cache shard graph %s (%s)", cache.c_str(), leann_last_error());
    return LEANN_OK;
}

extern "C" int leann_sharded_open(const char *index_path_stem, int backend, size_t dims, const char *device_spec, leann_sharded **out) {
    if (!index_path_stem || !out || dims == 0) { leann_set_error("leann_sharded_open: null argument / dims == 0"); return LEANN_ERR_INVALID; }
    *out = nullptr;
    std::vector<int> devs;
    if (int rc = parse_device_list(device_spec, &devs)) return rc;
    try {
        {
            std::vector<leann_backend *> saved(devs.size(), nullptr);
            if (saved_shards(index_path_stem, backend, dims, devs, &saved) == LEANN_OK) {
                int rc = leann_sharded_from_handles(saved.data(), saved.size(), 1, out);
                if (rc) for (auto *h : saved) leann_backend_close(h);
                return rc;
            }
        }
        std::string src;
        uint64_t off = 0;
        size_t n = 0;
        if (int rc = rows_source(index_path_stem, backend, dims, &src, &off, &n)) return rc;
        struct stat st{};
        (void)stat(src.c_str(), &st);
        const size_t G = devs.size();
        std::vector<leann_backend *> hs(G, nullptr);
        std::vector<int> rcs(G, LEANN_OK);
        std::vector<std::string> errs(G);
        std::vector<std::thread> th;
        for (size_t g = 0; g < G; g++)
            th.emplace_back([&, g] {
                const uint64_t lo = shard_lo(n, G, g), hi = shard_lo(n, G, g + 1);
                try {
                    rcs[g] = open_one_shard(src, off, shard_file(index_path_stem, g, G, backend, true), st.st_mtime, backend, dims, lo,
                                            (size_t)(hi - lo), devs[g], &hs[g]);
                } catch (const std::exception &e) {
                    leann_set_error("%s", e.what());
                    rcs[g] = LEANN_ERR_IO;
                }
                if (rcs[g]) errs[g] = leann_last_error();
            });
        for (auto &t : th) t.join();
        for (size_t g = 0; g < G; g++)
            if (rcs[g]) {
                for (auto *h : hs) if (h) leann_backend_close(h);
                leann_set_error("shard %zu (device %d): %s", g, devs[g], errs[g].c_str());
                return rcs[g];
            }
        int rc = leann_sharded_from_handles(hs.data(), G, 1, out);
        if (rc) for (auto *h : hs) leann_backend_close(h);
        else leann_log(LEANN_LOG_INFO, "sharded index: %zu rows of %s in %zu shards", n, src.c_str(), G);
        return rc;
    } catch (const std::exception &e) {
        leann_set_error("leann_sharded_open: %s", e.what());
        return LEANN_ERR_IO;
    }
}

// The same thing as ONE leann_backend handle (leann_backend_open with a list / range of devices): what a Rust IndexSearcher holds.
int leann_internal_open_sharded_backend(const char *stem, int backend, size_t dims, const char *spec, leann_backend **out) {
    leann_sharded *s = nullptr;
    int rc = leann_sharded_open(stem, backend, dims, spec, &s);
    if (rc) return rc;
    leann_backend *h = new leann_backend();
    h->kind = backend;
    h->device = s->primary;
    h->g.n = s->total_rows;
    h->g.d = (uint32_t)dims;
    h->g.ld = (uint32_t)((dims + 3) & ~(size_t)3);
    h->owns_rows = false;
    h->sharded = s;
    *out = h;
    return LEANN_OK;
}
extern "C" int leann_sharded_as_backend(leann_sharded *s, leann_backend **out) {
    if (!s || !out || s->rccl) { leann_set_error("leann_sharded_as_backend: null argument, or a one-process-per-GPU (RCCL) group"); return LEANN_ERR_INVALID; }
    leann_backend *h = new leann_backend();
    h->kind = s->shards[0].h->kind;
    h->device = s->primary;
    h->g.n = s->total_rows;
    h->g.d = (uint32_t)s->dims;
    h->g.ld = (uint32_t)((s->dims + 3) & ~(size_t)3);
    h->owns_rows = false;
    h->sharded = s;
    *out = h;
    return LEANN_OK;
}

// ---- one process per GPU (RCCL) -----------------------------------------------------------------------------------------------
extern "C" int leann_rccl_get_unique_id(void *id128) {
    if (!id128) { leann_set_error("leann_rccl_get_unique_id: null argument"); return LEANN_ERR_INVALID; }
    const RcclApi *api = rccl_api();
    if (!api) return LEANN_ERR_DEVICE;
    rcclUniqueId id;
    RCCL_CHECK_RET(api, api->GetUniqueId(&id));
    memcpy(id128, &id, 128);
    return LEANN_OK;
}
extern "C" int leann_sharded_attach(leann_backend *local, const void *unique_id128, int world, int rank, size_t total_rows,
                                    leann_sharded **out) {
    if (!local || local->sharded || !unique_id128 || !out || world < 1 || rank < 0 || rank >= world || world > 64) {
        leann_set_error("leann_sharded_attach: invalid arguments (world=%d rank=%d)", world, rank);
        return LEANN_ERR_INVALID;
    }
    *out = nullptr;
    const RcclApi *api = rccl_api();
    if (!api) return LEANN_ERR_DEVICE;
    HIP_CHECK_RET(hipSetDevice(local->device));
    leann_sharded *s = new leann_sharded();
    s->rccl = true;
    s->owns_shards = false;
    s->primary = local->device;
    s->world = world;
    s->rank = rank;
    s->dims = local->g.d;
    s->total_rows = total_rows ? total_rows : local->g.n;
    ShardDev sd;
    sd.h = local;
    sd.device = local->device;
    sd.lo = local->key_offset;
    s->shards.push_back(sd);
    int rc = init_common(s);
    if (rc) { leann_sharded_close(s); return rc; }
    rcclUniqueId id;
    memcpy(&id, unique_id128, 128);
    int r = api->CommInitRank(&s->comm, world, id, rank); // collective: every rank of the group calls attach
    if (r != 0) {
        s->comm = nullptr;
        leann_set_error("RCCL: ncclCommInitRank(world=%d, rank=%d) failed: %s", world, rank, api->GetErrorString(r));
        leann_sharded_close(s);
        return LEANN_ERR_DEVICE;
    }
    *out = s;
    return LEANN_OK;
}

// ---- search ---------------------------------------------------------------------------------------------------------------------
// d_queries [nq x dims] and the outputs live on the handle's first device (RCCL mode: the rank's device).  d_stats: optional
// [nq x 4] — one-process mode: summed over the shards (reduce_shard_stats_kernel, after the merge); RCCL mode: the local shard's.
// The traversal is queued behind `stream`; exchange and merge run on the handle's own stream.  ticket == nullptr: `stream` also waits
// for the merge (results are ordered on `stream`).
int leann_internal_sharded_search(leann_sharded *s, const float *d_queries, size_t nq, size_t top_k, size_t complexity,
                                  const ShardFilterArgs &fa, uint64_t *d_keys, float *d_dists, uint32_t *d_counts,
                                  uint32_t *d_stats, hipStream_t st, uint64_t *ticket) {
    if (!s || !d_queries || !d_keys || !d_dists || !d_counts || top_k == 0) { leann_set_error("sharded search: null/zero argument"); return LEANN_ERR_INVALID; }
    const uint8_t *d_allow = fa.d_allow;
    const size_t allow_stride = fa.allow_stride;
    const size_t G = s->rccl ? (size_t)s->world : s->shards.size();
    if (G * top_k > 12288) { leann_set_error("sharded search: shards x top_k = %zu x %zu exceeds the merge kernel's 12288 entries", G, top_k); return LEANN_ERR_INVALID; }
    if (s->rccl && (fa.sub || fa.exact)) { leann_set_error("sharded search over RCCL: registered / exact filters are a one-process feature"); return LEANN_ERR_UNSUPPORTED; }
    if (d_allow && s->rccl && (s->shards[0].lo & 7)) { leann_set_error("sharded filtered search: the shard does not start at a multiple of 8"); return LEANN_ERR_UNSUPPORTED; }
    if (nq == 0) return LEANN_OK;
    std::lock_guard<std::mutex> lk(s->mu); // enqueue phase only; the work itself is asynchronous
    HIP_CHECK_RET(hipSetDevice(s->primary));
    const uint64_t tk = s->next_ticket++;
    ShardSlot &sl = s->slots[tk & 1];
    const size_t blk = block_bytes(nq, top_k), koff = 0, doff = nq * top_k * 8, coff = nq * top_k * 12;
    if (int rc = grow_dev((void **)&sl.gather, &sl.cap_gather, G * blk)) return rc;
    if (d_stats && !s->rccl)
        if (int rc = grow_dev((void **)&sl.stats, &sl.cap_stats, G * nq * 16)) return rc;
    HIP_CHECK_RET(hipEventRecord(sl.ev_q, st)); // the queries (and the allow-bitmaps) are ready
    if (s->rccl) {
        const RcclApi *api = rccl_api();
        if (!api) return LEANN_ERR_DEVICE;
        if (int rc = grow_dev((void **)&sl.local, &sl.cap_local, blk)) return rc;
        if (sl.used) HIP_CHECK_RET(hipStreamWaitEvent(st, sl.done, 0)); // the all-gather that last read this slot's block has finished
        ShardDev &sd = s->shards[0];
        int rc = leann_backend_search_filtered_batch_device(sd.h, d_queries, nq, top_k, complexity, d_allow ? d_allow + sd.lo / 8 : nullptr,
                                                            allow_stride, (uint64_t *)(sl.local + koff), (float *)(sl.local + doff),
                                                            (uint32_t *)(sl.local + coff), d_stats, st);
        if (rc) return rc;
        HIP_CHECK_RET(hipEventRecord(sd.ev[tk & 1], st));
        HIP_CHECK_RET(hipStreamWaitEvent(s->xstream, sd.ev[tk & 1], 0));
        RCCL_CHECK_RET(api, api->AllGather(sl.local, sl.gather, blk, RCCL_UINT8, s->comm, s->xstream));
    } else {
        for (size_t g = 0; g < G; g++) {
            ShardDev &sd = s->shards[g];
            if (d_allow && (sd.lo & 7)) { leann_set_error("sharded filtered search: shard %zu does not start at a multiple of 8", g); return LEANN_ERR_UNSUPPORTED; }
            HIP_CHECK_RET(hipSetDevice(sd.device));
            HIP_CHECK_RET(hipStreamWaitEvent(sd.st, sl.ev_q, 0));
            if (sl.used) HIP_CHECK_RET(hipStreamWaitEvent(sd.st, sl.done, 0)); // the merge that last read this slot has finished
            const bool remote = sd.device != s->primary || leann_knobs().force_remote;
            const float *q = d_queries;
            const uint8_t *allow_g = d_allow ? d_allow + sd.lo / 8 : nullptr; // the shard's slice of the bitmap(s): its first position is bit 0
            unsigned char *blkp = sl.gather + g * blk;
            if (remote) {
                if (int rc = grow_dev((void **)&sd.d_q, &sd.cap_q, nq * s->dims * 4)) return rc;
                if (int rc = grow_dev((void **)&sd.d_out, &sd.cap_out, blk)) return rc;
                HIP_CHECK_RET(hipMemcpyPeerAsync(sd.d_q, sd.device, d_queries, s->primary, nq * s->dims * 4, sd.st));
                q = sd.d_q;
                blkp = sd.d_out;
                if (d_allow) { // only the bytes this shard reads travel: [lo / 8, ceil(hi / 8)) of the shared bitmap, or of the strided block
                    const size_t slice = (sd.h->g.n + 7) / 8;
                    const size_t span = allow_stride ? allow_stride * (nq - 1) + slice : slice;
                    if (int rc = grow_dev((void **)&sd.d_allow, &sd.cap_allow, span)) return rc;
                    HIP_CHECK_RET(hipMemcpyPeerAsync(sd.d_allow, sd.device, d_allow + sd.lo / 8, s->primary, span, sd.st));
                    allow_g = sd.d_allow;
                }
            }
            uint32_t *stats_g = nullptr;
            if (d_stats && !remote) stats_g = sl.stats + g * nq * 4;
            if (d_stats && remote) {
                if (int rc = grow_dev((void **)&sd.d_stats, &sd.cap_stats, nq * 16)) return rc;
                stats_g = sd.d_stats;
            }
            uint64_t *ok = (uint64_t *)(blkp + koff);
            float *od = (float *)(blkp + doff);
            uint32_t *oc = (uint32_t *)(blkp + coff);
            int rc;
            if (fa.exact && stats_g) HIP_CHECK_RET(hipMemsetAsync(stats_g, 0, nq * 16, sd.st)); // no walk: no evaluations / hops to count
            if (fa.sub && fa.exact) // registered filter, answered exactly: the shard's compacted list of allowed rows is scanned
                rc = sd.h->g.feat_h ? (leann_set_error("exact filtered search needs stored vectors; this index recomputes them from features"), (int)LEANN_ERR_UNSUPPORTED)
                                    : leann_internal_filtered_exact_list(sd.h->g.X, sd.h->g.d, sd.h->g.ld, q, nq, top_k, fa.sub[g]->d_list, fa.sub[g]->n_allowed,
                                                                         sd.h->key_offset, ok, od, oc, sd.st);
            else if (fa.sub) // registered filter inside the walk: the sub-filter's bitmap lives on the shard's device already
                rc = leann_backend_search_filtered_batch_device(sd.h, q, nq, top_k, complexity, fa.sub[g]->d_allow, 0, ok, od, oc, stats_g, sd.st);
            else if (fa.exact)
                rc = leann_backend_search_filtered_exact_batch_device(sd.h, q, nq, top_k, allow_g, allow_stride, ok, od, oc, sd.st);
            else
                rc = leann_backend_search_filtered_batch_device(sd.h, q, nq, top_k, complexity, allow_g, allow_stride, ok, od, oc, stats_g, sd.st);
            if (rc) { (void)hipSetDevice(s->primary); return rc; }
            if (remote) HIP_CHECK_RET(hipMemcpyPeerAsync(sl.gather + g * blk, s->primary, sd.d_out, sd.device, blk, sd.st));
            if (remote && d_stats) HIP_CHECK_RET(hipMemcpyPeerAsync(sl.stats + g * nq * 4, s->primary, sd.d_stats, sd.device, nq * 16, sd.st));
            HIP_CHECK_RET(hipEventRecord(sd.ev[tk & 1], sd.st));
        }
        HIP_CHECK_RET(hipSetDevice(s->primary));
        for (size_t g = 0; g < G; g++) HIP_CHECK_RET(hipStreamWaitEvent(s->xstream, s->shards[g].ev[tk & 1], 0));
    }
    int rc = leann_internal_merge_strided(sl.gather + koff, sl.gather + doff, sl.gather + coff, blk, blk, blk, G, nq, top_k, top_k, 0, d_keys,
                                          d_dists, d_counts, s->xstream);
    if (rc) return rc;
    if (d_stats && !s->rccl) {
        hipLaunchKernelGGL(reduce_shard_stats_kernel, dim3((unsigned)((nq * 4 + 255) / 256)), dim3(256), 0, s->xstream, sl.stats, (uint32_t)G,
                           (uint32_t)nq, d_stats);
        HIP_CHECK_RET(hipGetLastError());
    }
    HIP_CHECK_RET(hipEventRecord(sl.done, s->xstream));
    sl.used = true;
    if (ticket) *ticket = tk;
    else HIP_CHECK_RET(hipStreamWaitEvent(st, sl.done, 0));
    return LEANN_OK;
}

extern "C" int leann_sharded_search_batch_device(const leann_sharded *s, const float *d_queries, size_t nq, size_t top_k, size_t complexity,
                                                 uint64_t *d_keys, float *d_dists, uint32_t *d_counts, uint32_t *d_stats, void *stream) {
    return leann_internal_sharded_search(const_cast<leann_sharded *>(s), d_queries, nq, top_k, complexity, ShardFilterArgs{}, d_keys, d_dists, d_counts,
                                         d_stats, (hipStream_t)stream, nullptr);
}
uint64_t leann_internal_sharded_lo(const leann_sharded *s, size_t g) { return s && g < s->shards.size() ? s->shards[g].lo : 0; }

// leann_backend_save on a composite handle: every shard as a self-contained file of this library's own format,
// "<stem>.shard<g>of<G>.index" / ".diskann"; leann_backend_open(stem, ..., a list of G devices) finds them again (saved_shards below).
int leann_internal_sharded_save(const leann_sharded *s, const char *index_path_stem) {
    if (!s || s->rccl) { leann_set_error("leann_backend_save: a one-process-per-GPU (RCCL) group saves each rank's own shard handle"); return LEANN_ERR_UNSUPPORTED; }
    const size_t G = s->shards.size();
    for (size_t g = 0; g < G; g++) {
        const ShardDev &sd = s->shards[g];
        HIP_CHECK_RET(hipSetDevice(sd.device));
        if (int rc = leann_internal_save_to(sd.h, shard_file(index_path_stem, g, G, sd.h->kind, false))) { (void)hipSetDevice(s->primary); return rc; }
    }
    HIP_CHECK_RET(hipSetDevice(s->primary));
    return LEANN_OK;
}
extern "C" int leann_sharded_search_batch_device_async(const leann_sharded *s, const float *d_queries, size_t nq, size_t top_k,
                                                       size_t complexity, uint64_t *d_keys, float *d_dists, uint32_t *d_counts,
                                                       uint32_t *d_stats, void *stream, uint64_t *ticket) {
    if (!ticket) { leann_set_error("leann_sharded_search_batch_device_async: null ticket"); return LEANN_ERR_INVALID; }
    return leann_internal_sharded_search(const_cast<leann_sharded *>(s), d_queries, nq, top_k, complexity, ShardFilterArgs{}, d_keys, d_dists, d_counts,
                                         d_stats, (hipStream_t)stream, ticket);
}
// `stream` waits for the exchange + merge of `ticket`.  At most two tickets may be outstanding (two result slots rotate): wait for
// ticket t before issuing t + 2.
extern "C" int leann_sharded_wait(const leann_sharded *sc, uint64_t ticket, void *stream) {
    leann_sharded *s = const_cast<leann_sharded *>(sc);
    if (!s) { leann_set_error("leann_sharded_wait: null handle"); return LEANN_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(s->mu);
    if (ticket >= s->next_ticket || ticket + 2 < s->next_ticket) {
        leann_set_error("leann_sharded_wait: ticket %llu is not outstanding (next %llu)", (unsigned long long)ticket, (unsigned long long)s->next_ticket);
        return LEANN_ERR_INVALID;
    }
    HIP_CHECK_RET(hipSetDevice(s->primary));
    HIP_CHECK_RET(hipStreamWaitEvent((hipStream_t)stream, s->slots[ticket & 1].done, 0));
    return LEANN_OK;
}
extern "C" size_t leann_sharded_len(const leann_sharded *s) { return s ? s->total_rows : 0; }
extern "C" size_t leann_sharded_shards(const leann_sharded *s) { return leann_internal_sharded_count(s); }
