// internal.h — structs shared by api.hip / build.hip (not part of the C ABI).
#pragma once
#include "search.cuh"
#include "../../include/leann_backend.h"
#include <map>
#include <mutex>
#include <string>
#include <vector>

struct Workspace {
    uint32_t *overflow_list = nullptr; // [cap_nq]
    uint32_t *ctrs = nullptr;          // [0] overflow count, [1] work-queue head
    uint32_t *gtables = nullptr;       // [GT_BLOCKS << GT_BITS]
    size_t cap_nq = 0;
    // staging for the host-pointer API
    float *d_q = nullptr;
    uint64_t *d_keys = nullptr;
    float *d_dists = nullptr;
    uint32_t *d_counts = nullptr, *d_stats = nullptr;
    size_t cap_q = 0, cap_keys = 0, cap_dists = 0, cap_counts = 0, cap_stats = 0;
    hipStream_t stream = nullptr; // owned stream of the host-pointer API
};
static constexpr uint32_t GT_BLOCKS = 32, GT_BITS = 20;

struct leann_backend {
    int kind = LEANN_BACKEND_HNSW, device = 0;
    GraphView g{};
    uint64_t key_offset = 0;
    bool owns_rows = true;
    uint8_t *d_levels = nullptr;
    uint32_t efc = 64;
    float alpha = 1.2f;
    uint64_t n_upper_lists = 0;
    std::mutex mu;
    std::vector<Workspace *> free_ws;             // host-pointer API: one per concurrent caller
    std::map<hipStream_t, Workspace *> stream_ws; // device API: one per caller stream
    leann_search_stats stats{};
};

int leann_internal_launch_search(const GraphView &g, SearchArgs a, Workspace *w, hipStream_t st);
Workspace *leann_internal_stream_ws(leann_backend *h, hipStream_t st);
void leann_internal_free_graph(leann_backend *h);
std::string leann_internal_index_file(const char *stem, int backend);
