// indexfile.hip — index files behind leann_backend_open / leann_backend_save.
//
//     HnswSearcher::load     src/backend/hnsw.rs:18-75      "<stem minus .leann>.index"   (hnsw.rs:19)
//     DiskAnnSearcher::load  src/backend/diskann.rs:21-43   "<stem minus .leann>.diskann" (diskann.rs:22)
//
// Own format "LEANNGX1" (DESIGN.md §2): 128-byte header + levels + upper_off + adj0 + adjU + rows.
//   version 1: rows = unpadded f32 vectors [n x d];
//   version 2: recompute-on index (no vectors): rows = [n x row_bytes] {bf16 features | f32 ||W^T f|| | pad}, followed by the encoder
//              weights as f32 [feat_h x d] — what `is_pruned` (src/index/meta.rs:38-42, src/cli/prune.rs:17-79) means for a graph index.
// The header is checked against the file length BEFORE anything is allocated, every array is validated against n before it is
// uploaded (api.hip:validate_graph), and no exception crosses the C boundary.
//
// Index directories written by stock leann-rs hold a usearch / diskann-rs file this library cannot read (their format sources are
// not available offline).  When the reference builder also left `<stem>.embeddings` (raw LE f32 [n x dims],
// src/index/embeddings.rs:21-153, written in --recompute mode, src/index/builder.rs:105-113) the graph is rebuilt from it on the GPU
// (seconds: 10M x 768 in 27 s) and cached beside the original as "<stem>.gpu.index" / ".gpu.diskann"; the foreign file is left alone.
#include "common.cuh"
#include "search.cuh"
#include "../../include/leann_backend.h"
#include "internal.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <sys/stat.h>
#include <vector>

// Path::with_extension replaces the text after the last '.' of the file name.
std::string leann_internal_with_extension(const std::string &stem, const char *ext) {
    size_t slash = stem.find_last_of('/');
    size_t dot = stem.find_last_of('.');
    std::string base = (dot != std::string::npos && (slash == std::string::npos || dot > slash) && dot != slash + 1)
                           ? stem.substr(0, dot) : stem;
    return base + "." + ext;
}
std::string leann_internal_index_file(const char *stem, int backend) {
    return leann_internal_with_extension(stem, backend == LEANN_BACKEND_DISKANN ? "diskann" : "index");
}

#pragma pack(push, 1)
struct FileHeader {
    char magic[8]; // "LEANNGX1"
    uint32_t version, kind;
    uint64_t n;
    uint32_t d, M, M0, max_level, entry, efc;
    float alpha;
    uint32_t feat_h; // version 2: feature width (multiple of 4); 0 in version 1
    uint64_t n_upper_lists;
    uint32_t row_bytes; // version 2
    uint8_t pad[60];
};
#pragma pack(pop)
static_assert(sizeof(FileHeader) == 128, "index file header is 128 bytes");

static uint64_t payload_bytes(const FileHeader &hd) {
    const uint64_t n = hd.n;
    uint64_t b = n + 4 * n + 4 * n * hd.M0 + 4 * hd.n_upper_lists * hd.M;
    if (hd.version == 2) b += n * hd.row_bytes + 4ull * hd.feat_h * hd.d;
    else b += 4 * n * hd.d;
    return b;
}

int leann_internal_save_to(const leann_backend *h, const std::string &path) {
    const size_t n = h->g.n, d = h->g.d;
    const bool feat = h->g.feat_h != 0;
    std::vector<uint8_t> levels(std::max<size_t>(n, 1));
    std::vector<uint32_t> uo(std::max<size_t>(n, 1)), a0(std::max<size_t>(n * h->g.M0, 1)),
        aU(std::max<size_t>(h->n_upper_lists * h->g.M, 1));
    int rc = leann_backend_graph_export(h, levels.data(), uo.data(), a0.data(), aU.data(), nullptr);
    if (rc) return rc;
    // written beside the target and renamed over it: add_to_index rewrites the file it has just loaded, and a crash or a full disk
    // half-way must not leave a truncated index behind (the loader would refuse it, but the old one would be gone)
    const std::string tmp = path + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) { leann_set_error("cannot create %s", tmp.c_str()); return LEANN_ERR_IO; }
    FileHeader hd{};
    memcpy(hd.magic, "LEANNGX1", 8);
    hd.version = feat ? 2 : 1; hd.kind = (uint32_t)h->kind; hd.n = n; hd.d = (uint32_t)d; hd.M = h->g.M; hd.M0 = h->g.M0;
    hd.max_level = h->g.max_level; hd.entry = h->g.entry; hd.efc = h->efc; hd.alpha = h->alpha;
    hd.n_upper_lists = h->n_upper_lists;
    hd.feat_h = h->g.feat_h;
    hd.row_bytes = feat ? (uint32_t)leann_internal_feat_file_row_bytes(h->g) : h->g.row_bytes;
    bool ok = fwrite(&hd, sizeof(hd), 1, f) == 1;
    ok = ok && (n == 0 || fwrite(levels.data(), 1, n, f) == n);
    ok = ok && (n == 0 || fwrite(uo.data(), 4, n, f) == n);
    ok = ok && (n == 0 || fwrite(a0.data(), 4, n * h->g.M0, f) == n * h->g.M0);
    ok = ok && (h->n_upper_lists == 0 || fwrite(aU.data(), 4, h->n_upper_lists * h->g.M, f) == h->n_upper_lists * h->g.M);
    // rows in slabs of <= 256 MiB: a 10M x 768 index is 30 GB, more than some hosts want to hold twice
    const size_t row_b = feat ? leann_internal_feat_file_row_bytes(h->g) : d * 4, dev_pitch = feat ? h->g.row_bytes : (size_t)h->g.ld * 4;
    const size_t slab_rows = std::max<size_t>(1, ((size_t)256 << 20) / std::max<size_t>(row_b, 1));
    std::vector<unsigned char> slab(std::min(slab_rows, std::max<size_t>(n, 1)) * row_b);
    for (size_t r0 = 0; ok && r0 < n; r0 += slab_rows) {
        const size_t rows = std::min(slab_rows, n - r0);
        if (feat ? leann_internal_feat_rows_to_host(h, r0, rows, slab.data()) != LEANN_OK
                 : hipMemcpy2D(slab.data(), row_b, reinterpret_cast<const unsigned char *>(h->g.X) + r0 * dev_pitch, dev_pitch, row_b, rows,
                               hipMemcpyDeviceToHost) != hipSuccess) {
            fclose(f);
            (void)remove(tmp.c_str());
            leann_set_error("leann_backend_save: device read failed: %s", hipGetErrorString(hipGetLastError()));
            return LEANN_ERR_DEVICE;
        }
        ok = fwrite(slab.data(), row_b, rows, f) == rows;
    }
    if (ok && feat) {
        std::vector<float> W((size_t)h->g.feat_h * d);
        if (hipMemcpy(W.data(), h->Wf32, W.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) {
            fclose(f);
            (void)remove(tmp.c_str());
            leann_set_error("leann_backend_save: device read failed: %s", hipGetErrorString(hipGetLastError()));
            return LEANN_ERR_DEVICE;
        }
        ok = fwrite(W.data(), 4, W.size(), f) == W.size();
    }
    ok = (fclose(f) == 0) && ok;
    if (!ok) { (void)remove(tmp.c_str()); leann_set_error("short write to %s", tmp.c_str()); return LEANN_ERR_IO; }
    if (rename(tmp.c_str(), path.c_str()) != 0) { (void)remove(tmp.c_str()); leann_set_error("cannot move %s into place", tmp.c_str()); return LEANN_ERR_IO; }
    return LEANN_OK;
}

extern "C" int leann_backend_save(const leann_backend *h, const char *index_path_stem) {
    if (!h || !index_path_stem) { leann_set_error("leann_backend_save: null argument"); return LEANN_ERR_INVALID; }
    if (h->sharded) { // every shard as its own self-contained file "<stem>.shard<g>of<G>.index"; leann_backend_open with G devices finds them
        try { return leann_internal_sharded_save(h->sharded, index_path_stem); }
        catch (const std::exception &e) { leann_set_error("leann_backend_save: %s", e.what()); return LEANN_ERR_IO; }
    }
    try {
        HIP_CHECK_RET(hipSetDevice(h->device));
        return leann_internal_save_to(h, leann_internal_index_file(index_path_stem, h->kind));
    } catch (const std::exception &e) {
        leann_set_error("leann_backend_save: %s", e.what());
        return LEANN_ERR_IO;
    }
}

int leann_internal_parse_device(const char *spec, int *device) {
    *device = 0;
    if (!spec || !*spec) return LEANN_OK;
    char *end = nullptr;
    long v = strtol(spec, &end, 10);
    if (end == spec || *end != 0 || v < 0 || v > 1023) {
        leann_set_error("device_spec \"%s\": expected a HIP device ordinal (\"0\"), a list / range of ordinals for a sharded index "
                        "(\"0,1,2,3\", \"0-7\") or \"\"", spec);
        return LEANN_ERR_INVALID;
    }
    *device = (int)v;
    return LEANN_OK;
}

static const char *FOREIGN_MSG = // hnsw.rs:57-69
    "Failed to load index: incompatible format.\n"
    "This may be a FAISS index from Python LEANN, or a usearch/diskann-rs file written by stock leann-rs.\n"
    "Rebuild with: leann build <name> --docs <path> --force\n\n"
    "Original error: %s in %s";

enum { LOAD_OK = 0, LOAD_FOREIGN = -1 }; // > 0: LEANN_ERR_*
// Load one file of our format.  LOAD_FOREIGN: the magic is not ours (the caller decides what that means).
static int load_own_file(const std::string &path, int backend, size_t dims, int device, leann_backend **out, std::string *why) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) { *why = "cannot open"; return LEANN_ERR_NOT_FOUND; }
    struct stat stt{};
    const bool have_size = fstat(fileno(f), &stt) == 0;
    FileHeader hd{};
    const size_t got = fread(&hd, 1, sizeof(hd), f);
    if (got != sizeof(hd) || memcmp(hd.magic, "LEANNGX1", 8) != 0) {
        fclose(f);
        *why = "bad magic/header";
        return LOAD_FOREIGN;
    }
    auto bad = [&](const char *msg) {
        fclose(f);
        *why = msg;
        leann_set_error("Failed to load index: %s in %s", msg, path.c_str());
        return (int)LEANN_ERR_FORMAT;
    };
    if (hd.version != 1 && hd.version != 2) return bad("unsupported version");
    if (hd.kind != (uint32_t)backend) return bad(backend == LEANN_BACKEND_HNSW ? "the file holds a DiskANN graph, an HNSW index was asked for"
                                                                                 : "the file holds an HNSW graph, a DiskANN index was asked for");
    if (hd.d == 0 || hd.d > 4096 || hd.M == 0 || hd.M > 64 || hd.M0 == 0 || hd.M0 > 64 || hd.n >= (1ull << 31) || hd.max_level > 15 ||
        hd.n_upper_lists > hd.n * 15ull || (hd.n && hd.entry >= hd.n))
        return bad("header fields out of range");
    if (hd.version == 2 && (hd.feat_h == 0 || (hd.feat_h & 3) || hd.feat_h > 1024 || hd.row_bytes < 2 * hd.feat_h + 4 || (hd.row_bytes & 7) ||
                            hd.row_bytes > 4096))
        return bad("recompute-on header fields out of range");
    if (have_size && (uint64_t)stt.st_size != sizeof(hd) + payload_bytes(hd)) return bad("file length does not match the header (truncated or corrupt)");
    if (dims && hd.d != dims) {
        fclose(f);
        leann_set_error("index has %u dimensions, expected %zu", hd.d, dims);
        return LEANN_ERR_FORMAT;
    }
    const size_t n = hd.n, d = hd.d;
    const bool feat = hd.version == 2;
    std::vector<uint8_t> levels(std::max<size_t>(n, 1));
    std::vector<uint32_t> uo(std::max<size_t>(n, 1)), a0(std::max<size_t>(n * hd.M0, 1)), aU(std::max<size_t>(hd.n_upper_lists * hd.M, 1));
    std::vector<unsigned char> rows(std::max<size_t>(n * (feat ? (size_t)hd.row_bytes : d * 4), 4));
    std::vector<float> W(feat ? (size_t)hd.feat_h * d : 0);
    bool ok = (n == 0 || fread(levels.data(), 1, n, f) == n);
    ok = ok && (n == 0 || fread(uo.data(), 4, n, f) == n);
    ok = ok && (n == 0 || fread(a0.data(), 4, n * hd.M0, f) == n * hd.M0);
    ok = ok && (hd.n_upper_lists == 0 || fread(aU.data(), 4, hd.n_upper_lists * hd.M, f) == hd.n_upper_lists * hd.M);
    ok = ok && (n == 0 || fread(rows.data(), feat ? hd.row_bytes : d * 4, n, f) == n);
    ok = ok && (!feat || fread(W.data(), 4, W.size(), f) == W.size());
    fclose(f);
    if (!ok) { *why = "truncated file"; leann_set_error("Failed to load index: truncated file %s", path.c_str()); return LEANN_ERR_FORMAT; }
    int rc = leann_internal_from_host(backend, n, d, hd.M, hd.M0, hd.max_level, hd.entry, levels.data(), uo.data(), a0.data(), aU.data(),
                                      hd.n_upper_lists, feat ? nullptr : reinterpret_cast<const float *>(rows.data()), feat ? rows.data() : nullptr,
                                      hd.feat_h, hd.row_bytes, feat ? W.data() : nullptr, device, 0, out);
    if (rc == LEANN_ERR_FORMAT) { // validate_graph: keep the reason, name the file
        std::string msg = leann_last_error();
        leann_set_error("Failed to load index: %s (%s)", msg.c_str(), path.c_str());
    }
    if (rc == LEANN_OK) { (*out)->efc = hd.efc ? hd.efc : 64; (*out)->alpha = hd.alpha; }
    return rc;
}

// A directory written by stock leann-rs: rebuild the graph from "<stem>.embeddings" (see the file comment).
static int rebuild_from_embeddings(const std::string &emb_path, const std::string &sidecar, int backend, size_t dims, int device,
                                   leann_backend **out) {
    struct stat st{};
    if (stat(emb_path.c_str(), &st) != 0 || dims == 0 || st.st_size == 0 || (uint64_t)st.st_size % (dims * 4) != 0) {
        leann_set_error("%s: %lld bytes is not a whole number of %zu-dimensional f32 embeddings", emb_path.c_str(), (long long)st.st_size, dims);
        return LEANN_ERR_FORMAT;
    }
    const size_t n = (size_t)st.st_size / (dims * 4), ld = (dims + 3) & ~(size_t)3;
    if (n >= (1ull << 31)) { leann_set_error("%s: too many embeddings (%zu)", emb_path.c_str(), n); return LEANN_ERR_FORMAT; }
    FILE *f = fopen(emb_path.c_str(), "rb");
    if (!f) { leann_set_error("cannot open %s", emb_path.c_str()); return LEANN_ERR_IO; }
    int ndev = 0;
    leann_device_count(&ndev);
    if (device >= ndev) {
        fclose(f);
        leann_set_error("HIP device %d not available (%d visible). This library has no CPU fallback.", device, ndev);
        return LEANN_ERR_DEVICE;
    }
    if (hipSetDevice(device) != hipSuccess) { fclose(f); leann_set_error("hipSetDevice(%d) failed", device); return LEANN_ERR_DEVICE; }
    float *dX = nullptr;
    if (hipMalloc((void **)&dX, n * ld * 4) != hipSuccess) {
        fclose(f);
        leann_set_error("hipMalloc(%zu) for the embeddings failed", n * ld * 4);
        return LEANN_ERR_DEVICE;
    }
    const size_t slab_rows = std::max<size_t>(1, ((size_t)256 << 20) / (dims * 4));
    std::vector<float> slab(std::min(slab_rows, n) * dims);
    bool ok = ld == dims || hipMemset(dX, 0, n * ld * 4) == hipSuccess;
    for (size_t r0 = 0; ok && r0 < n; r0 += slab_rows) {
        const size_t rows = std::min(slab_rows, n - r0);
        ok = fread(slab.data(), dims * 4, rows, f) == rows &&
             hipMemcpy2D(dX + r0 * ld, ld * 4, slab.data(), dims * 4, dims * 4, rows, hipMemcpyHostToDevice) == hipSuccess;
    }
    fclose(f);
    if (!ok) { (void)hipFree(dX); leann_set_error("reading %s into device memory failed", emb_path.c_str()); return LEANN_ERR_IO; }
    // the reference's own build defaults are graph_degree 32, complexity 64 (src/cli/build.rs:78-83); a wider construction beam costs
    // seconds here.  Vamana needs R = 64 beyond ~1M rows.
    size_t degree = backend == LEANN_BACKEND_HNSW ? 32 : 64, complexity = 128;
    if (const char *e = getenv("LEANN_REBUILD_DEGREE")) degree = (size_t)atoi(e);
    if (const char *e = getenv("LEANN_REBUILD_COMPLEXITY")) complexity = (size_t)atoi(e);
    leann_backend *h = nullptr;
    int rc = leann_backend_build_device(backend, dX, n, dims, ld, degree, complexity, device, 0, 0, &h);
    if (rc) { (void)hipFree(dX); return rc; }
    h->owns_rows = true;
    if (leann_internal_save_to(h, sidecar) != LEANN_OK) // read-only directory: serve from memory, rebuild again next time
        leann_log(LEANN_LOG_WARN, "could not cache the rebuilt graph as %s (%s); it will be rebuilt on the next open", sidecar.c_str(), leann_last_error());
    *out = h;
    return LEANN_OK;
}

int leann_internal_load_own_file(const std::string &path, int backend, size_t dims, int device, leann_backend **out, std::string *why) {
    int rc = load_own_file(path, backend, dims, device, out, why);
    return rc == LOAD_FOREIGN ? (int)LEANN_ERR_FORMAT : rc;
}

static int open_impl(const char *index_path_stem, int backend, size_t dims, int device, leann_backend **out) {
    const std::string path = leann_internal_index_file(index_path_stem, backend);
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) {
        if (backend == LEANN_BACKEND_HNSW) // hnsw.rs:34-40
            leann_set_error("Index file not found: \"%s\"\nRun 'leann build' to create an index first.", path.c_str());
        else // diskann.rs:26-32
            leann_set_error("DiskANN index not found: \"%s\"\nRun 'leann build' with --backend-name diskann to create an index first.", path.c_str());
        return LEANN_ERR_NOT_FOUND;
    }
    unsigned char m[4] = {0, 0, 0, 0};
    const size_t got = fread(m, 1, 4, f);
    fclose(f);
    if (got >= 4 && ((m[0] == 'I' && m[1] == 'x') || !memcmp(m, "CSR\0", 4) || !memcmp(m, "HNSW", 4))) { // compat.rs:15-38, hnsw.rs:24-32
        leann_set_error("This index was built with Python LEANN (FAISS format).\n"
                        "Rust LEANN uses usearch which has a different binary format.\n\n"
                        "To use this index with Rust LEANN, you need to rebuild it:\n"
                        "  leann build <name> --docs <path> --force\n\n"
                        "The passages and metadata files are compatible and will be preserved.");
        return LEANN_ERR_FORMAT;
    }
    std::string why;
    int rc = load_own_file(path, backend, dims, device, out, &why);
    if (rc != LOAD_FOREIGN) return rc;
    // Not ours: a usearch / diskann-rs file of stock leann-rs.  Cached rebuild, or rebuild from the embeddings file.
    const std::string sidecar = leann_internal_with_extension(index_path_stem, backend == LEANN_BACKEND_DISKANN ? "gpu.diskann" : "gpu.index");
    const std::string emb = leann_internal_with_extension(index_path_stem, "embeddings");
    struct stat se{}, ss{};
    const bool have_emb = stat(emb.c_str(), &se) == 0, have_side = stat(sidecar.c_str(), &ss) == 0;
    if (have_side && (!have_emb || ss.st_mtime >= se.st_mtime)) {
        std::string why2;
        rc = load_own_file(sidecar, backend, dims, device, out, &why2);
        if (rc == LEANN_OK) {
            if (!have_emb || dims == 0 || (uint64_t)se.st_size == (uint64_t)(*out)->g.n * dims * 4) {
                leann_log(LEANN_LOG_INFO, "%s is not a LEANNGX1 file; using the cached GPU graph %s", path.c_str(), sidecar.c_str());
                return LEANN_OK;
            }
            leann_backend_close(*out); // the embeddings changed size since the cache was written
            *out = nullptr;
        }
        leann_log(LEANN_LOG_WARN, "ignoring stale or unreadable %s (%s)", sidecar.c_str(), why2.empty() ? "row count differs from the embeddings file" : why2.c_str());
    }
    if (have_emb && dims) {
        leann_log(LEANN_LOG_WARN, "%s is a stock leann-rs (usearch / diskann-rs) file; rebuilding the graph on the GPU from %s", path.c_str(), emb.c_str());
        return rebuild_from_embeddings(emb, sidecar, backend, dims, device, out);
    }
    leann_set_error(FOREIGN_MSG, (why + "; no " + emb + " to rebuild the graph from").c_str(), path.c_str());
    return LEANN_ERR_FORMAT;
}

extern "C" int leann_backend_open(const char *index_path_stem, int backend, size_t dims, const char *device_spec,
                                  leann_backend **out) {
    if (!index_path_stem || !out) { leann_set_error("leann_backend_open: null argument"); return LEANN_ERR_INVALID; }
    *out = nullptr;
    if (backend != LEANN_BACKEND_HNSW && backend != LEANN_BACKEND_DISKANN) {
        leann_set_error("Unknown backend: %d", backend); // searcher.rs:98
        return LEANN_ERR_INVALID;
    }
    if (leann_internal_spec_is_sharded(device_spec)) // "0-7", "0,1,2,3": one handle over per-device shards (shard.hip)
        return leann_internal_open_sharded_backend(index_path_stem, backend, dims, device_spec, out);
    int device = 0;
    if (int rc = leann_internal_parse_device(device_spec, &device)) return rc;
    try { // (device availability is checked where the index goes to the device: file errors are reported without a GPU too)
        return open_impl(index_path_stem, backend, dims, device, out);
    } catch (const std::bad_alloc &) {
        leann_set_error("Failed to load index: out of host memory reading %s", leann_internal_index_file(index_path_stem, backend).c_str());
        return LEANN_ERR_IO;
    } catch (const std::exception &e) {
        leann_set_error("Failed to load index: %s", e.what());
        return LEANN_ERR_IO;
    }
}
