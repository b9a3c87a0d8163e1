// gen.hip — deterministic synthetic corpus / query rows written straight into HBM.
// Bit-identical to oracle/oracle.c:orc_gen_rows: integer hashing + exactly rounded f32 ops only.
//   row i of stream s:  c = hash(i) mod C;  z = centre_c + sigma * noise_i  (r dims)
//                       x = P^T z  (per element an fmaf chain in increasing k),  x /= max(||x||, 1e-12)
//   r == 0: x_j = gauss(i, j) i.i.d.
// One wave per row: lane l owns elements 256t+4l..+3 — the canonical mapping of common.cuh, so the
// squared norm uses the same wave tree as every distance in the traversal kernel.
#include "common.cuh"
#include "../../include/leann_backend.h"
#include <map>
#include <mutex>
#include <tuple>

#define TAG_P 0x50524F4A00000000ull
#define TAG_C 0x43454E5400000000ull
#define TAG_A 0x4153534700000000ull
#define TAG_N 0x4E4F495300000000ull

__global__ void gen_projection_kernel(uint64_t seed, uint32_t d, uint32_t ld, uint32_t r, float *__restrict__ P) {
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= r * ld) return;
    uint32_t k = idx / ld, j = idx % ld;
    P[idx] = j < d ? gauss_ih4(seed ^ TAG_P, k, j) : 0.f;
}

template <int T>
__global__ void __launch_bounds__(256) gen_rows_kernel(uint64_t seed, uint32_t d, uint32_t ld, uint32_t r,
                                                       uint32_t n_clusters, float sigma, uint32_t stream_id,
                                                       uint64_t i0, uint64_t n, const float *__restrict__ P,
                                                       float *__restrict__ out) {
    __shared__ volatile float s_z[4][256]; // volatile: wave-private producer/consumer, LDS is in-order per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t nseed = seed ^ TAG_N ^ ((uint64_t)stream_id * 0x9E3779B97F4A7C15ull);
    for (uint64_t ii = (uint64_t)blockIdx.x * 4 + wave; ii < n; ii += (uint64_t)gridDim.x * 4) {
        const uint64_t i = i0 + ii;
        float4 x[T];
        if (r == 0) {
#pragma unroll
            for (int t = 0; t < T; t++) {
                uint32_t j = 256u * t + 4u * lane;
                x[t].x = j + 0 < d ? gauss_ih4(nseed, i, j + 0) : 0.f;
                x[t].y = j + 1 < d ? gauss_ih4(nseed, i, j + 1) : 0.f;
                x[t].z = j + 2 < d ? gauss_ih4(nseed, i, j + 2) : 0.f;
                x[t].w = j + 3 < d ? gauss_ih4(nseed, i, j + 3) : 0.f;
            }
        } else {
            const uint64_t c = hash3(seed ^ TAG_A, stream_id, i) % n_clusters;
            for (uint32_t k = lane; k < r; k += 64)
                s_z[wave][k] = fmaf(sigma, gauss_ih4(nseed, i, k), gauss_ih4(seed ^ TAG_C, c, k));
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < T; t++) x[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            for (uint32_t k = 0; k < r; k++) {
                const float zk = s_z[wave][k];
                const float *Pk = P + (size_t)k * ld;
#pragma unroll
                for (int t = 0; t < T; t++) {
                    float4 p = row_load4(Pk, ld, t, lane);
                    x[t].x = fmaf(p.x, zk, x[t].x);
                    x[t].y = fmaf(p.y, zk, x[t].y);
                    x[t].z = fmaf(p.z, zk, x[t].z);
                    x[t].w = fmaf(p.w, zk, x[t].w);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int t = 0; t < T; t++) fma4(acc, x[t], x[t]);
        float nrm = sqrtf(wave_tree_sum(lane4_sum(acc)));
        nrm = nrm < 1e-12f ? 1e-12f : nrm;
        float *row = out + ii * ld;
#pragma unroll
        for (int t = 0; t < T; t++) {
            uint32_t j = 256u * t + 4u * lane;
            if (j < ld) {
                float4 o;
                o.x = x[t].x / nrm;
                o.y = x[t].y / nrm;
                o.z = x[t].z / nrm;
                o.w = x[t].w / nrm;
                *reinterpret_cast<float4 *>(row + j) = o;
            }
        }
    }
}

template <int T>
static int launch_gen(uint64_t seed, uint32_t d, uint32_t ld, uint32_t r, uint32_t n_clusters, float sigma,
                      uint32_t stream_id, uint64_t i0, uint64_t n, const float *P, float *out, hipStream_t st) {
    uint64_t blocks = (n + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(gen_rows_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, st, seed, d, ld, r, n_clusters, sigma,
                       stream_id, i0, n, P, out);
    HIP_CHECK_RET(hipGetLastError());
    return LEANN_OK;
}

extern "C" int leann_synth_rows_device(uint64_t seed, uint32_t dims, uint32_t ld, uint32_t r, uint32_t n_clusters,
                                       float sigma, uint32_t stream_id, uint64_t i0, uint64_t n, float *d_out,
                                       void *stream) {
    if (!d_out || dims == 0 || ld < dims || (ld & 3) || ld > 4096 || r > 256 || (r && n_clusters == 0)) {
        leann_set_error("leann_synth_rows_device: invalid arguments (dims=%u ld=%u r=%u)", dims, ld, r);
        return LEANN_ERR_INVALID;
    }
    if (n == 0) return LEANN_OK;
    hipStream_t st = (hipStream_t)stream;
    float *P = nullptr;
    if (r) {
        // projection matrices are tiny (r x ld floats) and reused by every call with the same
        // parameters: keep them for the life of the process (no stream-ordered allocator involved)
        static std::mutex mu;
        static std::map<std::tuple<int, uint64_t, uint32_t, uint32_t, uint32_t>, float *> cache;
        int dev = 0;
        HIP_CHECK_RET(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lk(mu);
        auto key = std::make_tuple(dev, seed, dims, ld, r);
        auto it = cache.find(key);
        if (it == cache.end()) {
            HIP_CHECK_RET(hipMalloc((void **)&P, (size_t)r * ld * sizeof(float)));
            uint32_t total = r * ld;
            hipLaunchKernelGGL(gen_projection_kernel, dim3((total + 255) / 256), dim3(256), 0, st, seed, dims, ld, r, P);
            HIP_CHECK_RET(hipGetLastError());
            HIP_CHECK_RET(hipStreamSynchronize(st)); // other streams may use the cached matrix next
            cache[key] = P;
        } else {
            P = it->second;
        }
    }
    int T = (int)((ld + 255) / 256), rc;
    switch (T) {
        case 1: rc = launch_gen<1>(seed, dims, ld, r, n_clusters, sigma, stream_id, i0, n, P, d_out, st); break;
        case 2: rc = launch_gen<2>(seed, dims, ld, r, n_clusters, sigma, stream_id, i0, n, P, d_out, st); break;
        case 3: rc = launch_gen<3>(seed, dims, ld, r, n_clusters, sigma, stream_id, i0, n, P, d_out, st); break;
        case 4: rc = launch_gen<4>(seed, dims, ld, r, n_clusters, sigma, stream_id, i0, n, P, d_out, st); break;
        case 5: case 6: rc = launch_gen<6>(seed, dims, ld, r, n_clusters, sigma, stream_id, i0, n, P, d_out, st); break;
        case 7: case 8: rc = launch_gen<8>(seed, dims, ld, r, n_clusters, sigma, stream_id, i0, n, P, d_out, st); break;
        case 9: case 10: case 11: case 12: rc = launch_gen<12>(seed, dims, ld, r, n_clusters, sigma, stream_id, i0, n, P, d_out, st); break;
        default: rc = launch_gen<16>(seed, dims, ld, r, n_clusters, sigma, stream_id, i0, n, P, d_out, st); break;
    }
    return rc;
}
