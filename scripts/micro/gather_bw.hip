// gather_bw.hip — what the memory system sustains for the traversal kernels' access pattern: every wave reads whole rows at random
// addresses (one row = `row_bytes` contiguous bytes, R rows in flight per wave), no compute to speak of.  The ceiling the
// recompute-on graph search (520-B rows) and the f32 traversal (3 072-B rows) are priced against in DESIGN.md.
//   hipcc -O3 --offload-arch=gfx950 gather_bw.hip -o gather_bw.bin && ./gather_bw.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}
// LPL = loads of 16 B (or 8 B when W8) per lane per row
// NORM: 0 = the f32 norm sits behind the features in the row, 1 = no norm read, 2 = norms in an array of their own (n_rows x 4 B)
template <int R, int LPL, bool W8, int NORM = 0>
__global__ void __launch_bounds__(256) gather_kernel(const char *__restrict__ base, uint64_t n_rows, uint32_t row_bytes, uint32_t iters,
                                                     float *__restrict__ sink, const float *__restrict__ norms = nullptr) {
    const int lane = threadIdx.x & 63;
    const uint64_t wid = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    float acc = 0.f;
    for (uint32_t it = 0; it < iters; it++) {
        float4 v[R][LPL];
        float nrm[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint64_t row = mix64(wid * 0x100000001b3ull + (uint64_t)it * R + r) % n_rows;
            const char *p = base + row * row_bytes;
#pragma unroll
            for (int t = 0; t < LPL; t++) {
                if (W8) { const uint2 u = *reinterpret_cast<const uint2 *>(p + t * 512 + lane * 8); v[r][t] = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), 0.f, 0.f); }
                else v[r][t] = *reinterpret_cast<const float4 *>(p + t * 1024 + lane * 16);
            }
            nrm[r] = !W8 || NORM == 1 ? 0.f : NORM == 2 ? norms[row] : *reinterpret_cast<const float *>(p + LPL * 512);
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
#pragma unroll
            for (int t = 0; t < LPL; t++) acc += v[r][t].x + v[r][t].y + v[r][t].z + v[r][t].w;
            acc += nrm[r];
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}

// the recompute-on kernel's own access shape: 16 lanes per 520-B row (two 16-byte loads per lane, each instruction a contiguous 256-B
// half of four rows), the norm read by lane 15 of the row; G groups of four rows in flight per wave
template <int G, bool SPLIT = false>
__global__ void __launch_bounds__(256) gather4_kernel(const char *__restrict__ base, uint64_t n_rows, uint32_t row_bytes, uint32_t iters,
                                                      float *__restrict__ sink, const float *__restrict__ norms = nullptr) {
    typedef uint32_t u32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
    const int lane = threadIdx.x & 63, m = lane & 15, sub = lane >> 4;
    const uint64_t wid = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    uint32_t acc = 0;
    float facc = 0.f;
    for (uint32_t it = 0; it < iters; it++) {
        u32x4_a8 va[G], vb[G];
        float nrm[G];
#pragma unroll
        for (int g = 0; g < G; g++) {
            const uint64_t row = mix64(wid * 0x100000001b3ull + ((uint64_t)it * G + g) * 4 + sub) % n_rows;
            const char *p = base + row * row_bytes;
            va[g] = *reinterpret_cast<const u32x4_a8 *>(p + 16 * m);
            vb[g] = *reinterpret_cast<const u32x4_a8 *>(p + 256 + 16 * m);
            nrm[g] = 0.f;
            if (m == 15) nrm[g] = SPLIT ? norms[row] : *reinterpret_cast<const float *>(p + 512);
        }
#pragma unroll
        for (int g = 0; g < G; g++) {
            acc += va[g].x + va[g].y + va[g].z + va[g].w + vb[g].x + vb[g].y + vb[g].z + vb[g].w;
            facc += nrm[g];
        }
    }
    if (acc == 123456u && facc == 1.5f) sink[0] = facc;
}
template <int G, bool SPLIT = false>
static int run4(const char *name, const char *d, uint64_t bytes, int wg_per_cu, float *sink, const float *norms = nullptr) {
    const uint32_t rb = SPLIT ? 512u : 520u;
    const uint64_t n_rows = bytes / rb;
    const uint32_t iters = 400;
    const int grid = 256 * wg_per_cu;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((gather4_kernel<G, SPLIT>), dim3(grid), dim3(256), 0, 0, d, n_rows, rb, 20u, sink, norms);
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((gather4_kernel<G, SPLIT>), dim3(grid), dim3(256), 0, 0, d, n_rows, rb, iters, sink, norms);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double rows = (double)grid * 4 * iters * G * 4;
    printf("%-34s G=%d WG/CU=%d: %7.1f Mrows/s  %6.2f TB/s algorithmic (520 B/row)  ~%6.2f TB/s in whole 128-B lines\n", name, G, wg_per_cu,
           rows / ms / 1e3, rows * 520 / ms / 1e9, rows * 5 * 128.0 / ms / 1e9);
    return 0;
}

template <int R, int LPL, bool W8, int NORM = 0>
static int run(const char *name, const char *d, uint64_t bytes, uint32_t row_bytes, uint32_t algo_bytes, int wg_per_cu, float *sink,
               const float *norms = nullptr) {
    const uint64_t n_rows = bytes / row_bytes;
    const uint32_t iters = 400;
    const int grid = 256 * wg_per_cu;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((gather_kernel<R, LPL, W8, NORM>), dim3(grid), dim3(256), 0, 0, d, n_rows, row_bytes, 20u, sink, norms);
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((gather_kernel<R, LPL, W8, NORM>), dim3(grid), dim3(256), 0, 0, d, n_rows, row_bytes, iters, sink, norms);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double rows = (double)grid * 4 * iters * R;
    const uint32_t lines = (algo_bytes + 127) / 128; // 520 B at any 8-byte alignment touch exactly 5 lines
    printf("%-34s R=%d WG/CU=%d: %7.1f Mrows/s  %6.2f TB/s algorithmic (%u B/row)  ~%6.2f TB/s in whole 128-B lines\n", name, R, wg_per_cu,
           rows / ms / 1e3, rows * algo_bytes / ms / 1e9, algo_bytes, rows * lines * 128.0 / ms / 1e9);
    return 0;
}

int main() {
    const uint64_t bytes = 24ull << 30; // far beyond L2 + Infinity Cache
    char *d = nullptr;
    float *sink = nullptr;
    CHECK(hipMalloc((void **)&d, bytes + 4096));
    CHECK(hipMalloc((void **)&sink, 16));
    CHECK(hipMemset(d, 1, bytes + 4096));
    {   // the recompute-on index's own size: 10M passages.  Does a 512-B aligned row with the norm elsewhere gather faster?
        const uint64_t n = 10000000;
        float *norms = nullptr;
        CHECK(hipMalloc((void **)&norms, n * 4));
        CHECK(hipMemset(norms, 0, n * 4));
        for (int wg : {4, 6, 8}) {
            run4<1>("10M rows: 4 rows / instruction", d, n * 520, wg, sink);
            run4<2>("10M rows: 4 rows / instruction", d, n * 520, wg, sink);
            run4<4>("10M rows: 4 rows / instruction", d, n * 520, wg, sink);
            run4<2, true>("10M rows: 4/instr, 512 B + norms[]", d, n * 512, wg, sink, norms);
            run4<4, true>("10M rows: 4/instr, 512 B + norms[]", d, n * 512, wg, sink, norms);
        }
        for (int wg : {4, 8}) {
            run<8, 1, true, 0>("10M rows: 512+8 B inline norm", d, n * 520, 520, 520, wg, sink);
            run<8, 1, true, 1>("10M rows: 512 B, no norm read", d, n * 512, 512, 512, wg, sink);
            run<8, 1, true, 2>("10M rows: 512 B + norm array", d, n * 512, 512, 516, wg, sink, norms);
            run<4, 1, true, 2>("10M rows: 512 B + norm array, R=4", d, n * 512, 512, 516, wg, sink, norms);
        }
        CHECK(hipFree(norms));
    }
    for (int wg : {2, 4, 6, 8}) {
        run<8, 1, true>("feature rows 512+8 B (stride 520)", d, bytes, 520, 520, wg, sink);
        run<8, 1, true>("feature rows padded to 640 B", d, bytes, 640, 520, wg, sink);
        run<4, 1, true>("feature rows 520 B, 4 in flight", d, bytes, 520, 520, wg, sink);
        run<4, 3, false>("f32 rows 3 072 B", d, bytes, 3072, 3072, wg, sink);
        run<2, 6, false>("f32 rows 6 144 B", d, bytes, 6144, 6144, wg, sink);
    }
    return 0;
}
