// serve_bench.cpp — the reference server's call pattern without the HTTP layer: T threads, each issuing one
// BackendSearcher::search per request (src/cli/serve.rs:289-292 calls searcher.search under RwLock::read from tokio workers;
// src/backend/traits.rs:16-21 is one query per call).  Links only the C ABI.  Measures queries/s and per-call latency with and
// without request coalescing (leann_backend_set_coalescing) — natively, so that no interpreter lock sits between the callers.
//   serve_bench [rows=1000000] [dims=768] [threads=64] [calls_per_thread=200] [ef=64] [k=10]
#include "../../include/leann_backend.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#define CHECK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, leann_last_error()); return 1; } } while (0)

int main(int argc, char **argv) {
    const size_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1000000, d = argc > 2 ? strtoull(argv[2], nullptr, 10) : 768;
    const int T = argc > 3 ? atoi(argv[3]) : 64, calls = argc > 4 ? atoi(argv[4]) : 200;
    const size_t ef = argc > 5 ? strtoull(argv[5], nullptr, 10) : 64, k = argc > 6 ? strtoull(argv[6], nullptr, 10) : 10;
    const size_t nq = 4096;
    float *dX = nullptr, *dQ = nullptr;
    CHECK(leann_device_malloc(0, n * d * 4, (void **)&dX));
    CHECK(leann_device_malloc(0, nq * d * 4, (void **)&dQ));
    CHECK(leann_synth_rows_device(0x5EED0001ull, (uint32_t)d, (uint32_t)d, 64, 4096, 1.0f, 0, 0, n, dX, nullptr));
    CHECK(leann_synth_rows_device(0x5EED0001ull, (uint32_t)d, (uint32_t)d, 64, 4096, 1.0f, 1, 0, nq, dQ, nullptr));
    CHECK(leann_device_sync(0));
    std::vector<float> Q(nq * d);
    CHECK(leann_device_download(Q.data(), dQ, nq * d * 4));
    leann_backend *h = nullptr;
    const auto tb = std::chrono::steady_clock::now();
    CHECK(leann_backend_build_device(LEANN_BACKEND_HNSW, dX, n, d, d, 32, 128, 0, 0, 0, &h));
    fprintf(stderr, "index: %zu x %zu, built in %.1f s\n", n, d, std::chrono::duration<double>(std::chrono::steady_clock::now() - tb).count());
    for (int mode = 0; mode < 3; mode++) { // 0: the handle as opened (automatic coalescing), 1: switched off, 2: configured
        const uint32_t wait_us = getenv("SERVE_WAIT_US") ? (uint32_t)atoi(getenv("SERVE_WAIT_US")) : 100, max_b = getenv("SERVE_MAX_BATCH") ? (uint32_t)atoi(getenv("SERVE_MAX_BATCH")) : 64;
        if (mode) CHECK(leann_backend_set_coalescing(h, mode == 2 ? wait_us : 0, mode == 2 ? max_b : 0));
        std::vector<std::vector<double>> lat(T);
        std::atomic<int> failed{0};
        auto worker = [&](int t, int ncalls, bool record) {
            std::vector<uint64_t> keys(k);
            std::vector<float> dists(k);
            for (int c = 0; c < ncalls; c++) {
                const float *q = Q.data() + (size_t)((t * 7919 + c * 104729) % nq) * d;
                size_t n_out = 0;
                const auto t0 = std::chrono::steady_clock::now();
                if (leann_backend_search(h, q, k, ef, keys.data(), dists.data(), &n_out) || n_out == 0) failed++;
                if (record) lat[t].push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
            }
        };
        { std::vector<std::thread> th; for (int t = 0; t < T; t++) th.emplace_back(worker, t, 20, false); for (auto &x : th) x.join(); } // warm-up
        const auto t0 = std::chrono::steady_clock::now();
        { std::vector<std::thread> th; for (int t = 0; t < T; t++) th.emplace_back(worker, t, calls, true); for (auto &x : th) x.join(); }
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::vector<double> all;
        for (auto &v : lat) all.insert(all.end(), v.begin(), v.end());
        std::sort(all.begin(), all.end());
        printf("{\"threads\": %d, \"coalescing\": %s, \"queries_per_s\": %.0f, \"p50_us\": %.1f, \"p99_us\": %.1f, \"failed\": %d, \"rows\": %zu, \"dims\": %zu, \"ef\": %zu}\n",
               T, mode == 0 ? "\"automatic (default)\"" : mode == 2 ? "\"configured (SERVE_WAIT_US / SERVE_MAX_BATCH, default 100 us / 64)\"" : "\"off\"", (double)T * calls / secs, all[all.size() / 2], all[(size_t)(all.size() * 0.99)], failed.load(), n, d, ef);
    }
    leann_backend_close(h);
    leann_device_free(dX);
    leann_device_free(dQ);
    return 0;
}
