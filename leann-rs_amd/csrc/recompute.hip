// recompute.hip — the LEANN "no stored vectors" search: embeddings are recomputed on the fly from a
// device-resident encoder and never kept.
//
// Reference path (src/index/recompute.rs:52-123): for every passage, embedding_provider.embed(..)
// (:86-93) -> dot with the query (:96-103) -> stable sort descending (:106) -> take(k) (:109-120).
// The provider's last stage is a dense layer + L2 normalisation (candle.rs:165, :218-225:
// x / max(sqrt(sum x^2), 1e-12)); the transformer in front of it is out of scope (no weights offline),
// so the device-resident "embedding provider" here is:
//     per-passage compact feature  f_i  [h]      bf16   (h = 256: 512 B instead of 3 072 B per passage)
//     encoder weights              W    [h x d]  bf16
//     embedding_i = l2_normalize(W^T f_i)  with f32 accumulation
//
// Search (leann_recompute_search_batch[_device]), per tile of <= 64 queries: G = W Q^T once (project_queries_kernel), then
//   * common shape (h = 256, dims % 128 == 0, one token row per passage): fused_fstat_kernel (recompute_fstat.cuh) — encode GEMM,
//     row norms, feature-space scoring and candidate emission in one persistent kernel; first 16k rows through a score slab +
//     segment top-k (scan.hip) to establish the running k-th best, the rest emits survivors only (fold_candidates_kernel);
//   * other shapes: encode_kernel<CT, FUSED, POOL> below (128 passages x all columns per workgroup, both operands through LDS,
//     masked mean pooling over L token rows in the epilogue) + score slab + segment top-k per chunk.
// Embeddings are never materialised by a search; leann_recompute_encode_device writes them for validation, and
// leann_recompute_build_index builds a graph on a transient copy and keeps only features + norms (recompute-on graph).
#include "common.cuh"
#include "../../include/leann_backend.h"
#include "internal.h"
#include <thread>
#include <cstring>
#include <string>
#include "recompute_fstat.cuh" // fused_fstat_kernel, tile_features_kernel; bf16x8 / f32x16
#include <algorithm>
#include <mutex>
#include <utility>
#include <vector>


__host__ __device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
    uint32_t u;
#if defined(__HIP_DEVICE_COMPILE__)
    u = __float_as_uint(f);
#else
    memcpy(&u, &f, 4);
#endif
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40); // NaN stays NaN
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

// ---- synthetic features / weights: the recompute twin of gen.hip -----------------------------------
// f_i[k] = bf16(centre_c[k] + sigma * noise_i[k]),  W[k][j] = bf16(gauss(k, j))  (same hash streams as
// gen.hip with r = h), so normalise(W^T f_i) is the clustered synthetic corpus at bf16 input precision.
#define TAG_P 0x50524F4A00000000ull
#define TAG_C 0x43454E5400000000ull
#define TAG_A 0x4153534700000000ull
#define TAG_N 0x4E4F495300000000ull
#define TAG_F 0x4645415400000000ull /* feature lift */
// r_int == 0: f = bf16(z), z in R^h.   r_int > 0: z in R^r_int (clusters + noise, as gen.hip), f[k] = bf16(sum_m A[k][m] z[m])
// (fmaf chain in increasing m) — features of width h with intrinsic dimension r_int, like real encoder inputs.
__global__ void __launch_bounds__(256) synth_features_kernel(uint64_t seed, uint32_t h, uint32_t r_int, uint32_t n_clusters, float sigma,
                                                             uint32_t stream_id, uint64_t i0, uint64_t n, uint16_t *__restrict__ out) {
    extern __shared__ float s_feat[]; // [h * r_int] lift matrix A, then [r_int] z
    const uint64_t nseed = seed ^ TAG_N ^ ((uint64_t)stream_id * 0x9E3779B97F4A7C15ull);
    if (r_int == 0) {
        for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n * h; idx += (uint64_t)gridDim.x * blockDim.x) {
            const uint64_t i = i0 + idx / h;
            const uint32_t k = (uint32_t)(idx % h);
            const uint64_t c = hash3(seed ^ TAG_A, stream_id, i) % n_clusters;
            out[idx] = f32_to_bf16_rne(fmaf(sigma, gauss_ih4(nseed, i, k), gauss_ih4(seed ^ TAG_C, c, k)));
        }
        return;
    }
    float *A = s_feat, *z = s_feat + (size_t)h * r_int;
    for (uint32_t idx = threadIdx.x; idx < h * r_int; idx += blockDim.x) A[idx] = gauss_ih4(seed ^ TAG_F, idx / r_int, idx % r_int);
    for (uint64_t ii = blockIdx.x; ii < n; ii += gridDim.x) {
        const uint64_t i = i0 + ii;
        const uint64_t c = hash3(seed ^ TAG_A, stream_id, i) % n_clusters;
        __syncthreads();
        for (uint32_t m = threadIdx.x; m < r_int; m += blockDim.x)
            z[m] = fmaf(sigma, gauss_ih4(nseed, i, m), gauss_ih4(seed ^ TAG_C, c, m));
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < h; k += blockDim.x) {
            float acc = 0.f;
            for (uint32_t m = 0; m < r_int; m++) acc = fmaf(A[k * r_int + m], z[m], acc);
            out[ii * h + k] = f32_to_bf16_rne(acc);
        }
    }
}
__global__ void synth_weights_kernel(uint64_t seed, uint32_t h, uint32_t d, uint16_t *__restrict__ W /* [h x d] */) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= h * d) return;
    W[idx] = f32_to_bf16_rne(gauss_ih4(seed ^ TAG_P, idx / d, idx % d));
}
extern "C" int leann_synth_features_device(uint64_t seed, uint32_t h, uint32_t r_int, uint32_t n_clusters, float sigma,
                                           uint32_t stream_id, uint64_t i0, uint64_t n, uint16_t *d_out, void *stream) {
    if (!d_out || h == 0 || n_clusters == 0 || (size_t)(h + 1) * r_int * 4 > 150 * 1024) {
        leann_set_error("leann_synth_features_device: invalid arguments (h=%u r_int=%u)", h, r_int);
        return LEANN_ERR_INVALID;
    }
    if (n == 0) return LEANN_OK;
    const size_t lds = r_int ? ((size_t)h * r_int + r_int) * 4 : 0;
    if (lds > 64 * 1024)
        HIP_CHECK_RET(hipFuncSetAttribute((const void *)synth_features_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const uint64_t blocks = r_int ? std::min<uint64_t>(n, 4096) : std::min<uint64_t>((n * h + 255) / 256, 65536);
    hipLaunchKernelGGL(synth_features_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, seed, h, r_int, n_clusters,
                       sigma, stream_id, i0, n, d_out);
    HIP_CHECK_RET(hipGetLastError());
    return LEANN_OK;
}
extern "C" int leann_synth_weights_device(uint64_t seed, uint32_t h, uint32_t dims, uint16_t *d_out, void *stream) {
    if (!d_out || h == 0 || dims == 0) { leann_set_error("leann_synth_weights_device: invalid arguments"); return LEANN_ERR_INVALID; }
    hipLaunchKernelGGL(synth_weights_kernel, dim3((h * dims + 255) / 256), dim3(256), 0, (hipStream_t)stream, seed, h, dims, d_out);
    HIP_CHECK_RET(hipGetLastError());
    return LEANN_OK;
}

// ---- weights re-tiled for the MFMA B operand: Wp[kstep][col][16] bf16, col padded to CT*32 ------------
__global__ void tile_weights_kernel(const uint16_t *__restrict__ W, uint32_t h, uint32_t d, uint32_t hp, uint32_t dp,
                                    uint16_t *__restrict__ Wp) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= hp * dp) return;
    const uint32_t ks = idx / (dp * 16), rem = idx % (dp * 16), col = rem / 16, kk = rem % 16, k = ks * 16 + kk;
    Wp[idx] = (k < h && col < d) ? W[(size_t)k * d + col] : (uint16_t)0;
}

// ------------------------------------------------------------------------------------------------
// Query-side projection for the fused search: G = W Q^T  [h x 64] in f32 (k-ordered fmaf chain), split
// into three bf16 pieces g = hi + lo + lo2 (24 mantissa bits: since the features are exactly bf16, the
// three bf16 MFMAs F*hi + F*lo + F*lo2 reproduce the f32 products F*g), tiled like the weights:
// Gp[kstep][piece*64 + q][16].  By associativity  <l2norm(W^T f), q> = <f, W q> / ||W^T f||, so the
// d-dimensional scoring GEMM against materialised embeddings becomes an h-dimensional one fused into
// the encode loop, and the embeddings never leave the registers.
// ------------------------------------------------------------------------------------------------
// Workgroup = 4 rows of W x 64 queries; the queries go through LDS in chunks of 64 columns (coalesced 256-B row pieces, stored
// transposed so that consecutive queries sit in consecutive banks), the W values are wave-uniform.  Each g is still ONE fmaf chain,
// j ascending — the arithmetic did not change, only the access pattern (one thread per (k, q) reading its query row straight from
// global memory touched 64 cache lines per load instruction: 112 us per 64 queries, 0.45 ms per 256).
__global__ void __launch_bounds__(256) project_queries_kernel(const uint16_t *__restrict__ W, uint32_t h, uint32_t hp, uint32_t d,
                                                              const float *__restrict__ Q, uint32_t ldq, uint32_t nq,
                                                              uint16_t *__restrict__ Gp, int by_query_tile, uint32_t q_slots) {
    // q_slots: query slots of the image (64 for encode_kernel; a multiple of 32 up to FSTAT_MAX_QUERIES for fused_fstat_kernel)
    __shared__ float sQ[64][65];
    __shared__ float sW[4][64];
    const uint32_t tid = threadIdx.x, kl = tid >> 6, ql = tid & 63;
    const uint32_t k = blockIdx.x * 4 + kl, q = blockIdx.y * 64 + ql;
    float g = 0.f;
    for (uint32_t j0 = 0; j0 < d; j0 += 64) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 4; p++) { // 16 query rows per pass, 4 consecutive columns per thread
            const uint32_t qr = (tid >> 4) + 16 * p, jj = (tid & 15) * 4, qg = blockIdx.y * 64 + qr;
#pragma unroll
            for (int e = 0; e < 4; e++) sQ[jj + e][qr] = (qg < nq && j0 + jj + e < d) ? Q[(size_t)qg * ldq + j0 + jj + e] : 0.f;
        }
        {
            const uint32_t kr = blockIdx.x * 4 + (tid >> 6), jc = j0 + (tid & 63);
            sW[tid >> 6][tid & 63] = (kr < h && jc < d) ? __uint_as_float((uint32_t)W[(size_t)kr * d + jc] << 16) : 0.f;
        }
        __syncthreads();
        const uint32_t lim = min(64u, d - j0);
        if (lim == 64) {
#pragma unroll 16
            for (uint32_t j = 0; j < 64; j++) g = fmaf(sW[kl][j], sQ[j][ql], g);
        } else {
            for (uint32_t j = 0; j < lim; j++) g = fmaf(sW[kl][j], sQ[j][ql], g);
        }
    }
    if (k >= hp || q >= q_slots) return;
    if (!(k < h && q < nq)) g = 0.f;
    const uint16_t hi = f32_to_bf16_rne(g);
    const float r1 = g - __uint_as_float((uint32_t)hi << 16);
    const uint16_t lo = f32_to_bf16_rne(r1);
    const float r2 = r1 - __uint_as_float((uint32_t)lo << 16);
    const uint16_t lo2 = f32_to_bf16_rne(r2);
    const uint32_t ks = k / 16, kk = k % 16;
    uint16_t *base = Gp + (size_t)ks * (q_slots * 3) * 16;
    if (by_query_tile) { // fused_fstat_kernel: Gp[kstep][query tile][piece][32 q][16] — the three pieces of a query tile are contiguous
        base += ((q >> 5) * 96 + (q & 31)) * 16 + kk;
        base[0] = hi;
        base[32 * 16] = lo;
        base[64 * 16] = lo2;
        return;
    }
    base[(0 * 64 + q) * 16 + kk] = hi;
    base[(1 * 64 + q) * 16 + kk] = lo;
    base[(2 * 64 + q) * 16 + kk] = lo2;
}

// ------------------------------------------------------------------------------------------------
// encode_kernel<CT, FUSED>: 512 threads = 8 waves; workgroup tile = 128 passages x (CT*4*32) columns
// (CT col tiles of 32 per wave, 4 column groups, 2 row halves of 64 passages).  d <= CT*128.
// LDS: sF[128][hp + 8] bf16 (whole K, loaded once) + a 3-slot ring of sW[DP (+192)][16] bf16 k-step slabs (LDS-DMA,
// two slabs in flight behind counted vmcnt waits and one raw s_barrier per k-step).
//   FUSED = false: E = l2norm(F W) stored as f32 (validation / leann_recompute_encode_device)
//   FUSED = true : nothing but S[q][passage] = <f, W q> / ||W^T f|| leaves the workgroup; each wave adds
//                  3 MFMAs per k-step (hi/lo/lo2 pieces) for one 32-query x 32-passage score tile,
//                  reusing the feature fragment it already holds as the B operand.
// ------------------------------------------------------------------------------------------------
template <int CT, bool FUSED, bool POOL>
__global__ void __launch_bounds__(512) encode_kernel(const uint16_t *__restrict__ F, uint64_t n, uint32_t h, uint32_t hp,
                                                     const uint16_t *__restrict__ Wp, uint32_t d, uint32_t ld_out,
                                                     float *__restrict__ E, const uint16_t *__restrict__ Gp, uint32_t nq,
                                                     float *__restrict__ S, uint32_t n_rows_s, uint32_t L,
                                                     const uint8_t *__restrict__ mask, float *__restrict__ norms_out) {
    constexpr int DP = CT * 128;              // padded columns of W
    constexpr int DPX = DP + (FUSED ? 192 : 0); // + three 64-query pieces of G
    constexpr int RING = 3;                     // k-step slabs in LDS: one being consumed, two in flight
    constexpr int NI = DPX * 2 / 64;            // 1-KiB DMA instructions per k-step slab
    constexpr int PER = (NI + 7) / 8;           // ... per wave (the same count in every wave: counted vmcnt waits)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t fstride = hp + 8; // bf16 elements per sF row
    uint16_t *sF = reinterpret_cast<uint16_t *>(smem);
    uint16_t *sW = sF + 128 * fstride;                              // [RING][DPX][16] linear, swizzled 16-B slots
    float *sN = reinterpret_cast<float *>(sW + RING * DPX * 16);    // [4 col groups][128 rows] partial sum of squares
    float *sM = sN + 4 * 128;                                       // [128] attention mask of the tile's token rows (0/1)
    float *sC = sM + 128;                                           // [128] max(token count, 1e-9) at each passage's first row
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int rhalf = wave >> 2, cgrp = wave & 3;
    const uint64_t row_base = (uint64_t)blockIdx.x * 128;

    // ---- stage the feature tile (zero-padded rows / k) ------------------------------------------------
    for (uint32_t idx = tid; idx < 128 * (hp / 8); idx += 512) {
        const uint32_t r = idx / (hp / 8), k8 = (idx % (hp / 8)) * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        const uint64_t row = row_base + r;
        if (row < n) {
            if (k8 + 8 <= h && ((h & 7) == 0)) v = *reinterpret_cast<const uint4 *>(F + row * h + k8);
            else {
                uint16_t t[8];
                for (int e = 0; e < 8; e++) t[e] = (k8 + e < h) ? F[row * h + k8 + e] : (uint16_t)0;
                memcpy(&v, t, 16);
            }
        }
        *reinterpret_cast<uint4 *>(sF + r * fstride + k8) = v;
    }
    // k-step staging by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B = 1 KiB per wave instruction, no VGPRs,
    // asynchronous until the wave's vmcnt wait).  The LDS image is linear in 16-B slots; slot p holds global piece
    // p ^ ((p >> 4) & 1) (swizzle on the SOURCE address), which makes the 32-B-stride fragment reads conflict-free.
    auto stage_w = [&](int slot, uint32_t ks) {
        const char *wsrc = reinterpret_cast<const char *>(Wp + (size_t)ks * DP * 16);
        const char *gsrc = FUSED ? reinterpret_cast<const char *>(Gp + (size_t)ks * 192 * 16) : nullptr;
        char *dst = reinterpret_cast<char *>(sW + slot * DPX * 16);
#pragma unroll
        for (int j = 0; j < PER; j++) {
            uint32_t i = wave + 8 * j;
            if (i >= (uint32_t)NI) i = NI - 1; // padding instruction: re-fetches the last piece (same bytes, same place)
            const uint32_t p = i * 64 + lane, g = p ^ ((p >> 4) & 1);
            const char *src = (!FUSED || g < (uint32_t)(DP * 2)) ? wsrc + (size_t)g * 16 : gsrc + (size_t)(g - DP * 2) * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(dst + i * 1024), 16, 0, 0);
        }
    };
    auto frag = [&](const uint16_t *region, uint32_t col) -> bf16x8 { // 8 consecutive k of one column (16 B)
        const uint32_t g = 2 * col + lh, p = g ^ ((g >> 4) & 1);
        return *reinterpret_cast<const bf16x8 *>(reinterpret_cast<const char *>(region) + p * 16);
    };
    f32x16 acc[2][CT];
    f32x16 accs; // FUSED: scores of query tile (cgrp & 1) x passage tile (2*rhalf + (cgrp >> 1))
#pragma unroll
    for (int i = 0; i < 16; i++) accs[i] = 0.f;
#pragma unroll
    for (int rt = 0; rt < 2; rt++)
#pragma unroll
        for (int ct = 0; ct < CT; ct++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[rt][ct][i] = 0.f;
    const int st = cgrp >> 1, qt = cgrp & 1;

    if (POOL && tid < 128) {
        const uint64_t row = row_base + tid;
        sM[tid] = (row < n && (!mask || mask[row] != 0)) ? 1.f : 0.f;
    }
    const uint32_t nks = hp / 16;
    __syncthreads(); // feature tile + mask visible (no DMA in flight yet: a plain barrier)
    if (POOL && tid < 128 && (tid % L) == 0) {
        float c = 0.f;
        for (uint32_t j = 0; j < L; j++) c += sM[tid + j];
        sC[tid] = c < 1e-9f ? 1e-9f : c; // candle.rs:213 count.clamp(1e-9, inf)
    }
    stage_w(0, 0);
    if (nks > 1) stage_w(1, 1);
    for (uint32_t ks = 0; ks < nks; ks++) {
        // (a) my pieces of slab ks have landed (slab ks+1 may stay in flight), (b) everybody's have, and everybody
        // is done reading slab ks-1, so (c) its slot can take slab ks+2.  One raw barrier per k-step; __syncthreads()
        // would drain the DMA queue (vmcnt(0)) and serialise the stream.
        if (ks + 1 < nks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (ks + 2 < nks) stage_w((ks + 2) % RING, ks + 2);
        const uint16_t *w = sW + (ks % RING) * DPX * 16;
        bf16x8 a[2], b[CT];
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
            a[rt] = *reinterpret_cast<const bf16x8 *>(sF + (rhalf * 64 + rt * 32 + l31) * fstride + ks * 16 + lh * 8);
#pragma unroll
        for (int ct = 0; ct < CT; ct++)
            b[ct] = frag(w, (cgrp * CT + ct) * 32 + l31);
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
#pragma unroll
            for (int ct = 0; ct < CT; ct++)
                acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rt], b[ct], acc[rt][ct], 0, 0, 0);
        if (FUSED) {
            const bf16x8 fb = st ? a[1] : a[0]; // B operand: B[k][j = passage]  (same bytes as the A fragment of F)
#pragma unroll
            for (int p = 0; p < 3; p++) {
                const bf16x8 g = frag(w, DP + p * 64 + qt * 32 + l31);
                accs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g, fb, accs, 0, 0, 0); // C[i = query][j = passage]
            }
        }
    }
    __syncthreads();

    // ---- masked mean pooling over the L token rows of a passage (candle.rs:191-216) ------------------------
    // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5): the four registers of a
    // group are four consecutive token rows, so L <= 4 pools in registers and L = 8 adds the lane + 32 partner.
    // The pooled mean lands in the register of the passage's first token; the other registers become 0.
    if (POOL) {
#pragma unroll
    for (int rt = 0; rt < 2; rt++) {
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
            const int r0 = rhalf * 64 + rt * 32 + 8 * g4 + 4 * lh;
            const float m0 = sM[r0], m1 = sM[r0 + 1], m2 = sM[r0 + 2], m3 = sM[r0 + 3];
#pragma unroll
            for (int ct = 0; ct < CT; ct++) {
                float v0 = acc[rt][ct][4 * g4] * m0, v1 = acc[rt][ct][4 * g4 + 1] * m1;
                float v2 = acc[rt][ct][4 * g4 + 2] * m2, v3 = acc[rt][ct][4 * g4 + 3] * m3;
                if (L == 1) {
                    v0 /= sC[r0]; v1 /= sC[r0 + 1]; v2 /= sC[r0 + 2]; v3 /= sC[r0 + 3];
                } else if (L == 2) {
                    v0 = (v0 + v1) / sC[r0]; v2 = (v2 + v3) / sC[r0 + 2]; v1 = 0.f; v3 = 0.f;
                } else if (L == 4) {
                    v0 = (((v0 + v1) + v2) + v3) / sC[r0]; v1 = v2 = v3 = 0.f;
                } else { // L == 8: tokens 0-3 in the lane with lh = 0, tokens 4-7 in its lh = 1 partner
                    float a = ((v0 + v1) + v2) + v3;
                    float t = a + __shfl_xor(a, 32, 64);
                    v0 = lh == 0 ? t / sC[r0] : 0.f; v1 = v2 = v3 = 0.f;
                }
                acc[rt][ct][4 * g4] = v0; acc[rt][ct][4 * g4 + 1] = v1; acc[rt][ct][4 * g4 + 2] = v2; acc[rt][ct][4 * g4 + 3] = v3;
            }
        }
    }
    } // POOL
    // ---- row norms of the pooled embeddings (candle.rs:218-225) ---------------------------------------------
#pragma unroll
    for (int rt = 0; rt < 2; rt++) {
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            float p = 0.f;
#pragma unroll
            for (int ct = 0; ct < CT; ct++) p = fmaf(acc[rt][ct][reg], acc[rt][ct][reg], p);
            // sum over the 32 lanes that share lane >> 5 (the 32 columns of a tile)
            p += __shfl_xor(p, 1, 64);
            p += __shfl_xor(p, 2, 64);
            p += __shfl_xor(p, 4, 64);
            p += __shfl_xor(p, 8, 64);
            p += __shfl_xor(p, 16, 64);
            if (l31 == 0) sN[cgrp * 128 + rhalf * 64 + rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh] = p;
        }
    }
    __syncthreads();
    if (FUSED) {
        // score tile: rows (regs) = queries, col (lane & 31) = token row.  Mask, pool the L adjacent lanes of a
        // passage, then <mean, q> / ||mean||; 128-B coalesced stores into S[q][passage] when L = 1.
        const int r = rhalf * 64 + st * 32 + l31;
        const uint64_t row = row_base + r;
        const float ss = ((sN[r] + sN[128 + r]) + (sN[256 + r] + sN[384 + r]));
        float nrm = sqrtf(ss);
        nrm = nrm < 1e-12f ? 1e-12f : nrm;
        if (!POOL) {
            if (row < n) {
#pragma unroll
                for (int reg = 0; reg < 16; reg++) {
                    const uint32_t q = qt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                    if (q < nq) S[(size_t)q * n_rows_s + row] = accs[reg] / nrm;
                }
            }
            return;
        }
        const float m = sM[r];
        const bool first = (r % L) == 0;
        const float cnt = first ? sC[r] : 1.f;
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            float v = accs[reg] * m;
            if (L >= 2) v += __shfl_xor(v, 1, 64);
            if (L >= 4) v += __shfl_xor(v, 2, 64);
            if (L >= 8) v += __shfl_xor(v, 4, 64);
            const uint32_t q = qt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            if (first && row < n && q < nq) S[(size_t)q * n_rows_s + row / L] = (v / cnt) / nrm;
        }
        return;
    }
#pragma unroll
    for (int rt = 0; rt < 2; rt++) {
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const int r = rhalf * 64 + rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            const uint64_t row = row_base + r;
            const float ss = ((sN[r] + sN[128 + r]) + (sN[256 + r] + sN[384 + r])); // fixed order: reproducible
            float nrm = sqrtf(ss);
            nrm = nrm < 1e-12f ? 1e-12f : nrm;
            if (row < n && (r % L) == 0) {
                if (norms_out && cgrp == 0 && l31 == 0) norms_out[row / L] = nrm; // ||W^T f|| before normalisation
#pragma unroll
                for (int ct = 0; ct < CT; ct++) {
                    const uint32_t col = (cgrp * CT + ct) * 32 + l31;
                    if (col < ld_out) E[(size_t)(row / L) * ld_out + col] = col < d ? acc[rt][ct][reg] / nrm : 0.f;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
struct leann_recompute {
    int device = 0;
    const uint16_t *F = nullptr; // borrowed [n x h]
    uint16_t *Wp = nullptr;      // owned, tiled
    uint16_t *Wraw = nullptr;    // owned copy [h x d] (query projection)
    size_t n = 0, h = 0, hp = 0, d = 0, dp = 0, ld = 0;
    uint64_t key_offset = 0;
    int ct = 0;
    uint32_t L = 1;                 // token rows per passage (masked mean pooling), 1 | 2 | 4 | 8
    const uint8_t *mask = nullptr;  // borrowed [n x L] attention mask or null (all ones)
    float last_ms[3] = {0, 0, 0}; // encode, score, top-k of the last search call (HIP events)
    // search scratch, grown on demand and kept (hipMalloc / hipFree of ~1 GB per call costs 2-30 ms depending on the box);
    // one search at a time per handle
    std::mutex mu;
    float *sS = nullptr;
    uint16_t *sGp = nullptr;
    uint64_t *sCandA = nullptr, *sCandB = nullptr, *sBest = nullptr;
    size_t capS = 0, capGp = 0, capCand = 0, capBest = 0;
    uint4 *Ft = nullptr;            // fragment-major copy of F (tile_features_kernel), made by the first exhaustive search
    uint16_t *ownF = nullptr;       // leann_recompute_create_host: the handle's own copy of the features
    unsigned char *sEmit = nullptr; // [64 f32 thr | 64 u32 cnt | u32 overflow | pad | 64 x EMIT_CAP u64 list]
    size_t capEmit = 0;
    // sharded form (leann_recompute_create_sharded): the passages live in `parts` (consecutive position ranges, possibly on different
    // devices); this handle owns only the gather block and the per-part staging
    struct Part {
        const leann_recompute *r = nullptr;
        hipStream_t st = nullptr;
        hipEvent_t done = nullptr;
        unsigned char *stage = nullptr; // remote parts: [queries | keys | scores | counts | mask slice] on the part's device
        size_t cap_stage = 0;
    };
    std::vector<Part> parts;
    unsigned char *gather = nullptr; // [n_parts x {keys | scores | counts}] on `device`
    size_t cap_gather = 0;
    hipEvent_t ev_q = nullptr;
};
static int grow_scratch(void **p, size_t *cap, size_t bytes) {
    if (bytes <= *cap) return LEANN_OK;
    (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    HIP_CHECK_RET(hipMalloc(p, bytes));
    *cap = bytes;
    return LEANN_OK;
}

static size_t encode_lds_bytes(size_t hp, size_t dp, bool fused = false) { return 128 * (hp + 8) * 2 + 3 * (dp + (fused ? 192 : 0)) * 16 * 2 + 6 * 128 * 4; }

// the feature-stationary fused kernel serves the common shape (one token row per passage, h = 256, dims a multiple of 128);
// LEANN_DEBUG_FUSED_V1 forces the general kernel (tests compare the two)
static bool use_fstat(const leann_recompute *r) {
    return r->L == 1 && !r->mask && r->h == 256 && r->dp % 128 == 0 && !leann_knobs().fused_v1;
}

static int launch_encode(const leann_recompute *r, uint64_t row0, uint64_t rows, float *E, hipStream_t st,
                         const uint16_t *Gp = nullptr, uint32_t nq = 0, float *S = nullptr, float *norms = nullptr,
                         const CandEmit *emit = nullptr, const uint32_t *idx = nullptr) {
    const bool fused = Gp != nullptr;
    if (idx && !(fused && use_fstat(r))) {
        leann_set_error("recompute: a row list needs the fused feature-stationary kernel");
        return LEANN_ERR_INVALID;
    }
    if (fused && use_fstat(r)) {
        // features stationary in registers (the common shape: h = 256, dims = 384 / 768)
        const size_t lds2 = 2 * 16 * 128 * 32 + 4 * 64 * 4 + FSTAT_MAX_QUERIES * 4; // two 64-KiB sub-slice buffers + per-wave norm exchange + thresholds
        const uint64_t unit_rows = 4 * LEANN_FSTAT_RB * 32, units = (rows + unit_rows - 1) / unit_rows; // NWV * RB * 32 passages per unit
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, r->device);
        const unsigned grid2 = (unsigned)std::min<uint64_t>(units, (uint64_t)cus);
        if (idx) { // rows [row0, row0 + rows) of the LIST: passage i = row idx[i] of the row-major features
            HIP_CHECK_RET(hipFuncSetAttribute((const void *)fused_fstat_kernel<16, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            hipLaunchKernelGGL((fused_fstat_kernel<16, false, true>), dim3(grid2), dim3(256), lds2, st, r->F, (uint64_t)rows, r->Wp,
                               (uint32_t)r->dp, Gp, nq, S, (uint32_t)rows, emit ? *emit : CandEmit{}, idx + row0);
        } else if (r->Ft) {
            HIP_CHECK_RET(hipFuncSetAttribute((const void *)fused_fstat_kernel<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            hipLaunchKernelGGL((fused_fstat_kernel<16, true>), dim3(grid2), dim3(256), lds2, st,
                               reinterpret_cast<const uint16_t *>(r->Ft) + row0 * r->h, (uint64_t)rows, r->Wp, (uint32_t)r->dp, Gp, nq, S,
                               (uint32_t)rows, emit ? *emit : CandEmit{});
        } else {
            HIP_CHECK_RET(hipFuncSetAttribute((const void *)fused_fstat_kernel<16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            hipLaunchKernelGGL((fused_fstat_kernel<16, false>), dim3(grid2), dim3(256), lds2, st, r->F + row0 * r->h, (uint64_t)rows, r->Wp,
                               (uint32_t)r->dp, Gp, nq, S, (uint32_t)rows, emit ? *emit : CandEmit{});
        }
        HIP_CHECK_RET(hipGetLastError());
        return LEANN_OK;
    }
    const size_t lds = encode_lds_bytes(r->hp, r->dp, fused);
    // row0 / rows count PASSAGES; the kernel works on token rows (L per passage, tiles of 128 token rows)
    const uint64_t tok0 = row0 * r->L, toks = rows * r->L;
    const unsigned grid = (unsigned)((toks + 127) / 128);
    const uint16_t *F = r->F + tok0 * r->h;
    const uint8_t *mk = r->mask ? r->mask + tok0 : nullptr;
#define LAUNCH_ONE(CT, FU, PO)                                                                                            \
    do {                                                                                                                  \
        HIP_CHECK_RET(hipFuncSetAttribute((const void *)encode_kernel<CT, FU, PO>, hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                          160 * 1024));                                                                   \
        hipLaunchKernelGGL((encode_kernel<CT, FU, PO>), dim3(grid), dim3(512), lds, st, F, (uint64_t)toks, (uint32_t)r->h,        \
                           (uint32_t)r->hp, r->Wp, (uint32_t)r->d, (uint32_t)r->ld, E, Gp, nq, S, (uint32_t)rows, r->L, mk, norms); \
    } while (0)
#define LAUNCH_CT(CT)                                                                                                     \
    do {                                                                                                                  \
        const bool pool = r->L > 1 || r->mask != nullptr;                                                                 \
        if (fused && pool) LAUNCH_ONE(CT, true, true);                                                                    \
        else if (fused) LAUNCH_ONE(CT, true, false);                                                                      \
        else if (pool) LAUNCH_ONE(CT, false, true);                                                                       \
        else LAUNCH_ONE(CT, false, false);                                                                                \
    } while (0)
    switch (r->ct) {
        case 1: LAUNCH_CT(1); break;
        case 2: LAUNCH_CT(2); break;
        case 3: LAUNCH_CT(3); break;
        case 4: LAUNCH_CT(4); break;
        case 6: LAUNCH_CT(6); break;
        default: leann_set_error("recompute: unsupported dims %zu", r->d); return LEANN_ERR_INVALID;
    }
#undef LAUNCH_CT
#undef LAUNCH_ONE
    HIP_CHECK_RET(hipGetLastError());
    return LEANN_OK;
}

extern "C" int leann_recompute_create(const uint16_t *d_features, size_t n, size_t h, const uint16_t *d_weights, size_t dims,
                                      int device, uint64_t key_offset, leann_recompute **out) {
    if (!out || (n && !d_features) || !d_weights || h == 0 || dims == 0 || dims > 768 || n >= (1ull << 32)) {
        leann_set_error("leann_recompute_create: invalid arguments (n=%zu h=%zu dims=%zu; dims <= 768)", n, h, dims);
        return LEANN_ERR_INVALID;
    }
    int ndev = 0;
    leann_device_count(&ndev);
    if (device < 0 || device >= ndev) {
        leann_set_error("HIP device %d not available (%d visible). This library has no CPU fallback.", device, ndev);
        return LEANN_ERR_DEVICE;
    }
    HIP_CHECK_RET(hipSetDevice(device));
    leann_recompute *r = new leann_recompute();
    r->device = device;
    r->F = d_features;
    r->n = n;
    r->h = h;
    r->hp = (h + 15) / 16 * 16;
    r->d = dims;
    r->ld = (dims + 3) & ~(size_t)3;
    int ct = (int)((dims + 127) / 128);
    if (ct == 5) ct = 6;
    r->ct = ct;
    r->dp = (size_t)ct * 128;
    r->key_offset = key_offset;
    if (encode_lds_bytes(r->hp, r->dp, true) > 160 * 1024) {
        delete r;
        leann_set_error("recompute: feature width %zu x dims %zu exceeds the 160 KiB LDS tile", h, dims);
        return LEANN_ERR_INVALID;
    }
    HIP_CHECK_RET(hipMalloc((void **)&r->Wp, r->hp * r->dp * 2));
    HIP_CHECK_RET(hipMalloc((void **)&r->Wraw, h * dims * 2));
    HIP_CHECK_RET(hipMemcpy(r->Wraw, d_weights, h * dims * 2, hipMemcpyDeviceToDevice));
    const uint32_t total = (uint32_t)(r->hp * r->dp);
    hipLaunchKernelGGL(tile_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, nullptr, d_weights, (uint32_t)h, (uint32_t)dims,
                       (uint32_t)r->hp, (uint32_t)r->dp, r->Wp);
    HIP_CHECK_RET(hipGetLastError());
    HIP_CHECK_RET(hipDeviceSynchronize());
    *out = r;
    return LEANN_OK;
}
// Token-level provider: features [n x L x h], attention mask [n x L] (0 = padding) -> masked mean pooling
// (candle.rs:191-216) between the dense layer and the normalisation.
extern "C" int leann_recompute_create_pooled(const uint16_t *d_features, const uint8_t *d_mask, size_t n, size_t tokens_per_passage,
                                             size_t h, const uint16_t *d_weights, size_t dims, int device, uint64_t key_offset,
                                             leann_recompute **out) {
    if (tokens_per_passage != 1 && tokens_per_passage != 2 && tokens_per_passage != 4 && tokens_per_passage != 8) {
        leann_set_error("leann_recompute_create_pooled: tokens_per_passage must be 1, 2, 4 or 8 (got %zu)", tokens_per_passage);
        return LEANN_ERR_INVALID;
    }
    int rc = leann_recompute_create(d_features, n, h, d_weights, dims, device, key_offset, out);
    if (rc) return rc;
    (*out)->L = (uint32_t)tokens_per_passage;
    (*out)->mask = d_mask;
    return LEANN_OK;
}
// Host-memory twins of create / search (SURVEY.md §8b "Recompute boundary": plain host pointers in, results out).
extern "C" int leann_recompute_create_host(const uint16_t *features, size_t n, size_t h, const uint16_t *weights, size_t dims,
                                           int device, uint64_t key_offset, leann_recompute **out) {
    if (!out || (n && !features) || !weights || h == 0 || dims == 0) {
        leann_set_error("leann_recompute_create_host: null/zero argument");
        return LEANN_ERR_INVALID;
    }
    int ndev = 0;
    leann_device_count(&ndev);
    if (device < 0 || device >= ndev) {
        leann_set_error("HIP device %d not available (%d visible). This library has no CPU fallback.", device, ndev);
        return LEANN_ERR_DEVICE;
    }
    HIP_CHECK_RET(hipSetDevice(device));
    uint16_t *dF = nullptr, *dW = nullptr;
    HIP_CHECK_RET(hipMalloc((void **)&dF, std::max<size_t>(n * h, 8) * 2));
    if (hipMalloc((void **)&dW, h * dims * 2) != hipSuccess) { (void)hipFree(dF); leann_set_error("hipMalloc failed"); return LEANN_ERR_DEVICE; }
    int rc = LEANN_OK;
    if ((n && hipMemcpy(dF, features, n * h * 2, hipMemcpyHostToDevice) != hipSuccess) ||
        hipMemcpy(dW, weights, h * dims * 2, hipMemcpyHostToDevice) != hipSuccess) {
        leann_set_error("H2D copy of the encoder inputs failed");
        rc = LEANN_ERR_DEVICE;
    }
    if (rc == LEANN_OK) rc = leann_recompute_create(dF, n, h, dW, dims, device, key_offset, out);
    (void)hipFree(dW); // copied and re-tiled by create
    if (rc != LEANN_OK) { (void)hipFree(dF); return rc; }
    (*out)->ownF = dF;
    return LEANN_OK;
}
extern "C" int leann_recompute_search_batch(const leann_recompute *r, const float *queries, size_t nq, size_t top_k,
                                            const uint8_t *allow_mask, uint64_t *keys, float *scores, uint32_t *counts) {
    if (!r || !queries || !keys || !scores || !counts || top_k == 0) {
        leann_set_error("leann_recompute_search_batch: null/zero argument");
        return LEANN_ERR_INVALID;
    }
    if (nq == 0) return LEANN_OK;
    HIP_CHECK_RET(hipSetDevice(r->device));
    const size_t nmask = allow_mask ? (r->n + 7) / 8 : 0;
    unsigned char *buf = nullptr; // [queries | keys | scores | counts | mask]
    const size_t oq = 0, ok = oq + nq * r->d * 4, os = ok + nq * top_k * 8, oc = os + nq * top_k * 4, om = (oc + nq * 4 + 15) & ~(size_t)15;
    HIP_CHECK_RET(hipMalloc((void **)&buf, om + nmask + 16));
    int rc = LEANN_OK;
    if (hipMemcpy(buf + oq, queries, nq * r->d * 4, hipMemcpyHostToDevice) != hipSuccess ||
        (nmask && hipMemcpy(buf + om, allow_mask, nmask, hipMemcpyHostToDevice) != hipSuccess)) {
        leann_set_error("H2D copy of the queries failed");
        rc = LEANN_ERR_DEVICE;
    }
    if (rc == LEANN_OK)
        rc = leann_recompute_search_batch_device(r, (const float *)(buf + oq), nq, top_k, nmask ? buf + om : nullptr, (uint64_t *)(buf + ok),
                                                 (float *)(buf + os), (uint32_t *)(buf + oc), nullptr);
    if (rc == LEANN_OK && (hipDeviceSynchronize() != hipSuccess || hipMemcpy(keys, buf + ok, nq * top_k * 8, hipMemcpyDeviceToHost) != hipSuccess ||
                           hipMemcpy(scores, buf + os, nq * top_k * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                           hipMemcpy(counts, buf + oc, nq * 4, hipMemcpyDeviceToHost) != hipSuccess)) {
        leann_set_error("recompute search: device error: %s", hipGetErrorString(hipGetLastError()));
        rc = LEANN_ERR_DEVICE;
    }
    (void)hipFree(buf);
    return rc;
}
int leann_internal_merge_strided(const void *keys, const void *dists, const void *counts, size_t kstride, size_t dstride, size_t cstride,
                                 size_t n_shards, size_t nq, size_t k_in, size_t k_out, int descending, uint64_t *d_out_keys,
                                 float *d_out_dists, uint32_t *d_out_counts, hipStream_t st);

// ---- sharded recompute search -------------------------------------------------------------------
// RecomputeSearcher::search scores EVERY passage (recompute.rs:86-103); the top-k of a union is the top-k of the per-part top-ks, so the
// passages split into consecutive position ranges, every part runs its own fused scan (concurrently: one host thread per part, the
// parts may sit on different devices) and the per-part lists are merged by (score descending, key ascending) — the order of the
// reference's stable sort (recompute.rs:106).  Per-passage arithmetic does not depend on a passage's neighbours in a tile, so the
// answer equals the unsharded handle's bit for bit (tests/test_gpu_shard.py).
extern "C" int leann_recompute_create_sharded(const leann_recompute *const *parts, size_t n_parts, leann_recompute **out) {
    if (!parts || !out || n_parts == 0 || n_parts > 64) { leann_set_error("leann_recompute_create_sharded: invalid arguments"); return LEANN_ERR_INVALID; }
    *out = nullptr;
    for (size_t g = 0; g < n_parts; g++) {
        const leann_recompute *p = parts[g];
        if (!p || !p->parts.empty() || p->d != parts[0]->d || p->h != parts[0]->h || p->L != parts[0]->L ||
            (g && p->key_offset != parts[g - 1]->key_offset + parts[g - 1]->n) || (g && ((p->key_offset - parts[0]->key_offset) & 7))) {
            leann_set_error("leann_recompute_create_sharded: part %zu is null, itself sharded, of another shape, or does not continue part %zu's "
                            "position range at a multiple of 8", g, g ? g - 1 : 0);
            return LEANN_ERR_INVALID;
        }
    }
    leann_recompute *r = new leann_recompute();
    r->device = parts[0]->device;
    r->d = parts[0]->d; r->h = parts[0]->h; r->hp = parts[0]->hp; r->dp = parts[0]->dp; r->ld = parts[0]->ld; r->L = parts[0]->L;
    r->key_offset = parts[0]->key_offset;
    bool ok = hipSetDevice(r->device) == hipSuccess && hipEventCreateWithFlags(&r->ev_q, hipEventDisableTiming) == hipSuccess;
    for (size_t g = 0; ok && g < n_parts; g++) {
        leann_recompute::Part pt;
        pt.r = parts[g];
        r->n += parts[g]->n;
        ok = hipSetDevice(parts[g]->device) == hipSuccess && hipStreamCreateWithFlags(&pt.st, hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&pt.done, hipEventDisableTiming) == hipSuccess;
        r->parts.push_back(pt);
    }
    (void)hipSetDevice(r->device);
    if (!ok) {
        leann_set_error("leann_recompute_create_sharded: stream / event creation failed: %s", hipGetErrorString(hipGetLastError()));
        leann_recompute_close(r);
        return LEANN_ERR_DEVICE;
    }
    *out = r;
    return LEANN_OK;
}

static int sharded_recompute_search(leann_recompute *r, const float *d_queries, size_t nq, size_t top_k, const uint8_t *d_allow_mask,
                                    uint64_t *d_keys, float *d_scores, uint32_t *d_counts, hipStream_t st) {
    const size_t G = r->parts.size();
    if (G * top_k > 12288) { leann_set_error("sharded recompute search: parts x top_k = %zu x %zu exceeds the merge kernel's 12288 entries", G, top_k); return LEANN_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(r->mu); // one search at a time per handle, like a plain handle's scratch
    HIP_CHECK_RET(hipSetDevice(r->device));
    const size_t okeys = 0, oscores = nq * top_k * 8, ocounts = nq * top_k * 12, blk = (nq * top_k * 12 + nq * 4 + 15) & ~(size_t)15;
    if (int rc = grow_scratch((void **)&r->gather, &r->cap_gather, G * blk)) return rc;
    HIP_CHECK_RET(hipEventRecord(r->ev_q, st)); // queries (and the mask) are ready
    std::vector<int> rcs(G, LEANN_OK);
    std::vector<std::string> errs(G);
    auto run_part = [&](size_t g) {
        leann_recompute::Part &pt = r->parts[g];
        const leann_recompute *p = pt.r;
        auto fail = [&](const char *what) { leann_set_error("sharded recompute search, part %zu: %s failed: %s", g, what, hipGetErrorString(hipGetLastError())); return (int)LEANN_ERR_DEVICE; };
        if (hipSetDevice(p->device) != hipSuccess) return fail("hipSetDevice");
        if (hipStreamWaitEvent(pt.st, r->ev_q, 0) != hipSuccess) return fail("hipStreamWaitEvent");
        const bool remote = p->device != r->device || leann_knobs().force_remote;
        const size_t mask_off = (size_t)((p->key_offset - r->key_offset) / 8), mask_b = d_allow_mask ? (p->n + 7) / 8 : 0;
        const float *q = d_queries;
        const uint8_t *mask = d_allow_mask ? d_allow_mask + mask_off : nullptr;
        unsigned char *outb = r->gather + g * blk;
        if (remote) { // queries (and the part's slice of the mask) travel to the part's device, its lists travel back
            const size_t oq = 0, ob = (nq * r->d * 4 + 15) & ~(size_t)15, om = ob + blk, need = om + mask_b + 16;
            if (need > pt.cap_stage) {
                (void)hipFree(pt.stage); pt.stage = nullptr; pt.cap_stage = 0;
                if (hipMalloc((void **)&pt.stage, need) != hipSuccess) return fail("hipMalloc");
                pt.cap_stage = need;
            }
            if (hipMemcpyPeerAsync(pt.stage + oq, p->device, d_queries, r->device, nq * r->d * 4, pt.st) != hipSuccess) return fail("peer copy of the queries");
            if (mask_b && hipMemcpyPeerAsync(pt.stage + om, p->device, d_allow_mask + mask_off, r->device, mask_b, pt.st) != hipSuccess) return fail("peer copy of the mask");
            q = (const float *)(pt.stage + oq);
            mask = mask_b ? pt.stage + om : nullptr;
            outb = pt.stage + ob;
        }
        int rc = leann_recompute_search_batch_device(p, q, nq, top_k, mask, (uint64_t *)(outb + okeys), (float *)(outb + oscores),
                                                     (uint32_t *)(outb + ocounts), pt.st);
        if (rc) return rc;
        if (remote && hipMemcpyPeerAsync(r->gather + g * blk, r->device, outb, p->device, blk, pt.st) != hipSuccess) return fail("peer copy of the results");
        if (hipEventRecord(pt.done, pt.st) != hipSuccess) return fail("hipEventRecord");
        return (int)LEANN_OK;
    };
    std::vector<std::thread> th;
    for (size_t g = 1; g < G; g++)
        th.emplace_back([&, g] { rcs[g] = run_part(g); if (rcs[g]) errs[g] = leann_last_error(); });
    rcs[0] = run_part(0);
    if (rcs[0]) errs[0] = leann_last_error();
    for (auto &t : th) t.join();
    (void)hipSetDevice(r->device);
    for (size_t g = 0; g < G; g++)
        if (rcs[g]) { leann_set_error("%s", errs[g].c_str()); return rcs[g]; }
    for (size_t g = 0; g < G; g++) HIP_CHECK_RET(hipStreamWaitEvent(st, r->parts[g].done, 0));
    r->last_ms[0] = r->last_ms[1] = r->last_ms[2] = 0.f;
    for (size_t g = 0; g < G; g++)
        for (int i = 0; i < 3; i++) r->last_ms[i] = std::max(r->last_ms[i], r->parts[g].r->last_ms[i]);
    return leann_internal_merge_strided(r->gather + okeys, r->gather + oscores, r->gather + ocounts, blk, blk, blk, G, nq, top_k, top_k, 1, d_keys,
                                        d_scores, d_counts, st);
}

extern "C" void leann_recompute_close(leann_recompute *r) {
    if (!r) return;
    for (auto &pt : r->parts) { // (the parts themselves are borrowed)
        (void)hipSetDevice(pt.r ? pt.r->device : r->device);
        if (pt.st) { (void)hipStreamSynchronize(pt.st); (void)hipStreamDestroy(pt.st); }
        if (pt.done) (void)hipEventDestroy(pt.done);
        (void)hipFree(pt.stage);
    }
    if (!r->parts.empty()) (void)hipSetDevice(r->device);
    if (r->ev_q) (void)hipEventDestroy(r->ev_q);
    (void)hipFree(r->gather);
    (void)hipFree(r->Wp);
    (void)hipFree(r->Wraw);
    (void)hipFree(r->sS);
    (void)hipFree(r->sGp);
    (void)hipFree(r->sCandA);
    (void)hipFree(r->sCandB);
    (void)hipFree(r->sBest);
    (void)hipFree(r->sEmit);
    (void)hipFree(r->Ft);
    (void)hipFree(r->ownF);
    delete r;
}
extern "C" size_t leann_recompute_len(const leann_recompute *r) { return r ? r->n : 0; }
extern "C" int leann_recompute_last_timing(const leann_recompute *r, float *ms3) {
    if (!r || !ms3) { leann_set_error("leann_recompute_last_timing: null argument"); return LEANN_ERR_INVALID; }
    ms3[0] = r->last_ms[0]; ms3[1] = r->last_ms[1]; ms3[2] = r->last_ms[2];
    return LEANN_OK;
}

// materialise the embeddings of rows [row0, row0 + rows) into d_out [rows x ld] (tests / validation)
extern "C" int leann_recompute_encode_device(const leann_recompute *r, uint64_t row0, uint64_t rows, float *d_out, void *stream) {
    if (!r || !d_out || row0 + rows > r->n) { leann_set_error("leann_recompute_encode_device: invalid arguments"); return LEANN_ERR_INVALID; }
    if (!r->parts.empty()) { leann_set_error("leann_recompute_encode_device: a sharded handle holds no features of its own (encode through its parts)"); return LEANN_ERR_UNSUPPORTED; }
    if (rows == 0) return LEANN_OK;
    HIP_CHECK_RET(hipSetDevice(r->device));
    return launch_encode(r, row0, rows, d_out, (hipStream_t)stream);
}

// scan.hip internals reused per chunk
int leann_internal_topk_chunk(const float *S, size_t rows, size_t nq, uint32_t k, const uint8_t *allow, uint64_t pos0, uint64_t *cand,
                              size_t cand_len, size_t seg_off, hipStream_t st, size_t *segs_out, uint64_t *best);
int leann_internal_scan_finish(uint64_t *candA, uint64_t *candB, size_t cand_len, size_t total_segs, size_t nq, uint32_t k,
                               uint64_t key_offset, uint64_t *d_keys, float *d_scores, uint32_t *d_counts, hipStream_t st);

// RecomputeSearcher::search arithmetic for a batch of queries — recompute.rs:86-109 on the GPU.
// Per tile of <= 64 queries: G = W Q^T once, then per chunk of passages ONE fused kernel (encode GEMM,
// row norms, feature-space scoring) + segment top-k; the embeddings are never written anywhere.
static int recompute_search_impl(const leann_recompute *r, const float *d_queries, size_t nq, size_t top_k, const uint8_t *d_allow_mask,
                                 uint64_t *d_keys, float *d_scores, uint32_t *d_counts, void *stream, bool emit_ok, bool *overflowed,
                                 const uint32_t *idx = nullptr, size_t n_list = 0);
int leann_internal_scan_finish_ex(uint64_t *candA, uint64_t *candB, size_t cand_len, size_t total_segs, size_t nq, uint32_t k, uint64_t key_offset,
                                  uint64_t *d_keys, float *d_scores, uint32_t *d_counts, hipStream_t st, const uint32_t *idx, int as_dist);
int leann_internal_compact_allow(const uint8_t *d_allow, size_t n, uint32_t **d_list, size_t *n_list, hipStream_t st);
int leann_internal_scratch_acquire(void **out, size_t bytes);
void leann_internal_scratch_release(void *p);
extern "C" int leann_recompute_search_batch_device(const leann_recompute *r, const float *d_queries, size_t nq, size_t top_k,
                                                   const uint8_t *d_allow_mask, uint64_t *d_keys, float *d_scores,
                                                   uint32_t *d_counts, void *stream) {
    if (!r || !d_queries || !d_keys || !d_scores || !d_counts || top_k == 0 || top_k > 1024) {
        leann_set_error("leann_recompute_search_batch_device: invalid arguments");
        return LEANN_ERR_INVALID;
    }
    if (nq == 0) return LEANN_OK;
    if (!r->parts.empty())
        return sharded_recompute_search(const_cast<leann_recompute *>(r), d_queries, nq, top_k, d_allow_mask, d_keys, d_scores, d_counts, (hipStream_t)stream);
    HIP_CHECK_RET(hipSetDevice(r->device));
    // The early filter of recompute.rs:62-79: the reference fetches, and therefore embeds, only the passages that pass the filter.
    // A selective mask is compacted into the list of allowed positions and only those rows go through the fused kernel (same
    // per-passage arithmetic, so the same scores as the masked pass over everything); LEANN_RECOMPUTE_NO_LIST=1 keeps the masked pass.
    uint32_t *list = nullptr;
    size_t n_list = 0;
    if (d_allow_mask && use_fstat(r) && r->n < (1ull << 32) && !leann_knobs().no_list) {
        int rc = leann_internal_compact_allow(d_allow_mask, r->n, &list, &n_list, (hipStream_t)stream);
        if (rc != LEANN_OK) return rc;
        if (n_list > r->n / 2) { // barely selective: the masked pass reads the fragment-major feature copy and skips the gather
            leann_internal_scratch_release(list); // (compact_allow synchronised the stream)
            list = nullptr;
        } else if (!list) { // nothing allowed: an empty list still needs a non-null marker for the list path
            if (int e = leann_internal_scratch_acquire((void **)&list, 4)) return e;
        }
    }
    bool overflowed = false;
    int rc = recompute_search_impl(r, d_queries, nq, top_k, d_allow_mask, d_keys, d_scores, d_counts, stream, true, &overflowed, list, n_list);
    // a candidate list overflowed (adversarial score order, e.g. ascending): repeat on the score-slab path
    if (rc == LEANN_OK && overflowed)
        rc = recompute_search_impl(r, d_queries, nq, top_k, d_allow_mask, d_keys, d_scores, d_counts, stream, false, &overflowed, list, n_list);
    if (list) {
        (void)hipStreamSynchronize((hipStream_t)stream); // (already idle unless an error cut the search short)
        leann_internal_scratch_release(list);
    }
    return rc;
}
static int recompute_search_impl(const leann_recompute *r, const float *d_queries, size_t nq, size_t top_k, const uint8_t *d_allow_mask,
                                 uint64_t *d_keys, float *d_scores, uint32_t *d_counts, void *stream, bool emit_ok, bool *overflowed,
                                 const uint32_t *idx, size_t n_list) {
    hipStream_t st = (hipStream_t)stream;
    const size_t N = idx ? n_list : r->n; // passages of this search (idx: the allowed ones, keys are list indices until the finalize step)
    if (idx) d_allow_mask = nullptr;
    const uint32_t k = (uint32_t)top_k;
    const size_t SEGSZ = 2048;
    // slab path: geometric chunk schedule 128k, 512k, 2M, 4M, 4M ... : the first small chunks fix the running k-th best, after
    // which topk_scores_kernel skips almost every segment unsorted
    std::vector<std::pair<size_t, size_t>> chunks; // (row0, rows)
    {
        // with candidate emission only the first chunk needs a score slab: 16k rows fix a first k-th-best bound, the next 496k
        // tighten it (~k * 496k / 16k survivors per query), everything else goes through ONE persistent launch
        const bool emit_schedule = emit_ok && use_fstat(r) && !leann_knobs().no_emit;
        size_t pos = 0, len = emit_schedule ? (size_t)16 << 10 : (size_t)128 << 10;
        while (pos < N) {
            size_t rows = std::min(len, N - pos);
            chunks.emplace_back(pos, rows);
            pos += rows;
            if (emit_schedule) len = chunks.size() == 1 ? (size_t)496 << 10 : N;
            else len = std::min<size_t>(len * 4, (size_t)4 << 20);
        }
    }
    const size_t n_chunks = chunks.size();
    size_t chunk = 0, total_segs = 0;
    for (auto &c : chunks) { chunk = std::max(chunk, c.second); total_segs += (c.second + SEGSZ - 1) / SEGSZ; }
    chunk = std::max<size_t>(chunk, SEGSZ);
    total_segs = std::max<size_t>(total_segs, 1);
    const size_t cand_len = std::max<size_t>(total_segs * k, k);
    leann_recompute *rw = const_cast<leann_recompute *>(r);
    std::lock_guard<std::mutex> scratch_lock(rw->mu);
    if (use_fstat(r) && !idx && !rw->Ft && r->n >= 4096 && !leann_knobs().no_tiled) {
        // first exhaustive search on this handle: keep a fragment-major copy of the features (+ n * h * 2 bytes; if HBM is short
        // the kernel reads the caller's row-major array instead, ~10 % slower)
        const uint64_t n_pad = (r->n + 255) / 256 * 256 + 256;
        if (hipMalloc((void **)&rw->Ft, n_pad * r->h * 2) == hipSuccess) {
            hipLaunchKernelGGL(tile_features_kernel, dim3(4096), dim3(256), 0, st, r->F, (uint64_t)r->n, (uint32_t)r->h, n_pad, rw->Ft);
            HIP_CHECK_RET(hipGetLastError());
        } else {
            (void)hipGetLastError();
            rw->Ft = nullptr;
        }
    }
    // queries per pass over the passages: the encode GEMM is shared by all of them (fused_fstat_kernel: up to 256, one G sub-slice
    // per 32 queries and unit); the general kernel is built for 64
    const size_t QT = use_fstat(r) ? FSTAT_MAX_QUERIES : 64;
    if (int e = grow_scratch((void **)&rw->sS, &rw->capS, sizeof(float) * QT * chunk)) return e;
    if (int e = grow_scratch((void **)&rw->sGp, &rw->capGp, r->hp * QT * 3 * 2 + 1024)) return e; // + the fourth (unused) KiB piece of the last k-step
    {
        size_t capB = rw->capCand;
        if (int e = grow_scratch((void **)&rw->sCandA, &rw->capCand, sizeof(uint64_t) * nq * cand_len)) return e;
        if (int e = grow_scratch((void **)&rw->sCandB, &capB, sizeof(uint64_t) * nq * cand_len)) return e;
    }
    if (int e = grow_scratch((void **)&rw->sBest, &rw->capBest, sizeof(uint64_t) * nq * k)) return e;
    float *S = rw->sS;
    uint16_t *Gp = rw->sGp;
    uint64_t *candA = rw->sCandA, *candB = rw->sCandB, *best = rw->sBest;
    // chunks after the first emit their few survivors straight from the fused kernel (no score slab, no segment sort)
    const uint32_t EMIT_CAP = std::max<uint32_t>(8192, 32 * k); // expected survivors per query ~ k * rows / rows_seen (x19 after 512k of 10M rows)
    const bool emit = emit_ok && use_fstat(r) && n_chunks > 1 && k <= 1024 && !leann_knobs().no_emit;
    CandEmit em{};
    uint32_t *d_overflow = nullptr;
    if (emit) {
        // [QT f32 thr | QT u32 cnt | u32 overflow | pad to 4 KiB | QT x EMIT_CAP u64 list]
        if (int e = grow_scratch((void **)&rw->sEmit, &rw->capEmit, 4096 + sizeof(uint64_t) * QT * EMIT_CAP)) return e;
        em.thr = reinterpret_cast<const float *>(rw->sEmit);
        em.cnt = reinterpret_cast<uint32_t *>(rw->sEmit + 1024);
        d_overflow = reinterpret_cast<uint32_t *>(rw->sEmit + 2048);
        em.list = reinterpret_cast<uint64_t *>(rw->sEmit + 4096);
        em.cap = EMIT_CAP;
        em.allow = d_allow_mask;
        HIP_CHECK_RET(hipMemsetAsync(rw->sEmit, 0, 4096, st));
    }
    HIP_CHECK_RET(hipMemsetAsync(candA, 0xFF, sizeof(uint64_t) * nq * cand_len, st));
    HIP_CHECK_RET(hipMemsetAsync(best, 0xFF, sizeof(uint64_t) * nq * k, st));
    int rc = LEANN_OK;
    const size_t n_tiles = (nq + QT - 1) / QT;
    struct EventSet { // destroyed on every exit, including the HIP_CHECK_RET returns below
        std::vector<hipEvent_t> v;
        ~EventSet() { for (auto &e : v) if (e) (void)hipEventDestroy(e); }
    } evset;
    evset.v.assign(n_tiles * n_chunks * 3 + 1, nullptr);
    std::vector<hipEvent_t> &evs = evset.v;
    for (auto &e : evs) HIP_CHECK_RET(hipEventCreate(&e));
    size_t ei = 0;
    for (size_t q0 = 0; q0 < nq && rc == LEANN_OK; q0 += QT) {
        const uint32_t nqt = (uint32_t)std::min<size_t>(QT, nq - q0);
        const uint32_t q_slots = use_fstat(r) ? (nqt + 31) / 32 * 32 : 64; // fused_fstat_kernel sizes its G image by 32-query tiles
        hipLaunchKernelGGL(project_queries_kernel, dim3((unsigned)((r->hp + 3) / 4), (unsigned)((q_slots + 63) / 64)), dim3(256), 0, st, r->Wraw, (uint32_t)r->h,
                           (uint32_t)r->hp, (uint32_t)r->d, d_queries + q0 * r->d, (uint32_t)r->d, nqt, Gp, use_fstat(r) ? 1 : 0, q_slots);
        size_t seg_off = 0;
        for (size_t c = 0; c < n_chunks && rc == LEANN_OK; c++) {
            const size_t row0 = chunks[c].first, rows = chunks[c].second;
            (void)hipEventRecord(evs[ei++], st);
            const bool emit_chunk = emit && c > 0;
            em.pos0 = row0;
            rc = launch_encode(r, row0, rows, nullptr, st, Gp, nqt, S, nullptr, emit_chunk ? &em : nullptr, idx);
            (void)hipEventRecord(evs[ei++], st);
            size_t segs = 0;
            if (rc == LEANN_OK && !emit_chunk)
                rc = leann_internal_topk_chunk(S, rows, nqt, k, d_allow_mask, row0, candA + q0 * cand_len, cand_len, seg_off, st, &segs,
                                               best + q0 * k);
            if (rc == LEANN_OK && emit) // merge the survivors (none after chunk 0), publish the k-th best as the next threshold
                rc = leann_internal_fold_candidates(em, k, nqt, (uint32_t)QT, best + q0 * k, d_overflow, st);
            (void)hipEventRecord(evs[ei++], st);
            seg_off += segs;
        }
    }
    if (rc == LEANN_OK) {
        // `best` is the exact running top-k; without emission the final answer is re-derived from all segment winners
        if (emit) rc = leann_internal_scan_finish_ex(best, candB, k, 1, nq, k, r->key_offset, d_keys, d_scores, d_counts, st, idx, 0);
        else rc = leann_internal_scan_finish_ex(candA, candB, cand_len, total_segs, nq, k, r->key_offset, d_keys, d_scores, d_counts, st, idx, 0);
    }
    (void)hipEventRecord(evs[ei], st);
    (void)hipStreamSynchronize(st);
    *overflowed = false;
    if (emit && rc == LEANN_OK) {
        uint32_t ov = 0;
        HIP_CHECK_RET(hipMemcpy(&ov, d_overflow, 4, hipMemcpyDeviceToHost));
        *overflowed = ov != 0;
    }
    {
        rw->last_ms[0] = rw->last_ms[1] = rw->last_ms[2] = 0.f;
        float ms = 0.f;
        for (size_t i = 0; i + 2 < ei + 1 && rc == LEANN_OK; i += 3) {
            if (hipEventElapsedTime(&ms, evs[i], evs[i + 1]) == hipSuccess) rw->last_ms[0] += ms;     // fused encode + score
            if (hipEventElapsedTime(&ms, evs[i + 1], evs[i + 2]) == hipSuccess) rw->last_ms[2] += ms; // segment top-k
        }
        if (ei >= 3 && rc == LEANN_OK && hipEventElapsedTime(&ms, evs[ei - 1], evs[ei]) == hipSuccess) rw->last_ms[2] += ms;
    }
    return rc;
}

// ================================================================================================
// Recompute-on GRAPH search (the LEANN idea proper: a graph index whose distances are recomputed from the
// encoder inputs instead of stored vectors).  The graph is built once on transiently materialised embeddings;
// afterwards the index keeps only the graph, the bf16 features and one f32 per passage (||W^T f||):
//     dist(q, i) = 1 - <f_i, W q> / ||W^T f_i||   ==   1 - <l2norm(W^T f_i), q>
// 520 B per evaluated neighbour instead of 3 072 B (h = 256, d = 768).  The traversal kernel is the same
// (search.cuh, FEAT instantiation); queries are projected once per batch (g = W q, f32 MFMA).
// ================================================================================================
__global__ void pack_feature_rows_kernel(const uint16_t *__restrict__ F, const float *__restrict__ norms, uint64_t n, uint32_t h,
                                         uint32_t hp4, uint32_t row_bytes, unsigned char *__restrict__ out) {
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; // (row, element) incl. the norm slot
    const uint32_t per = hp4 + 2;                                         // hp4 bf16 + one f32 (= 2 bf16 slots)
    if (idx >= n * per) return;
    const uint64_t row = idx / per;
    const uint32_t e = (uint32_t)(idx % per);
    unsigned char *dst = out + row * row_bytes;
    if (e < hp4) reinterpret_cast<uint16_t *>(dst)[e] = e < h ? F[row * h + e] : (uint16_t)0;
    else if (e == hp4 && row_bytes >= 2 * hp4 + 4) *reinterpret_cast<float *>(dst + 2 * (size_t)hp4) = norms[row]; // (split layout: no slot)
}
// the inline file / export form [features | f32 norm | pad] of an index whose device rows are split (GraphView::norms)
size_t leann_internal_feat_file_row_bytes(const GraphView &g) { return g.norms ? ((2 * (size_t)g.feat_h + 4 + 7) & ~(size_t)7) : g.row_bytes; }
int leann_internal_feat_rows_to_host(const leann_backend *h, size_t r0, size_t rows, unsigned char *out) {
    const size_t fb = leann_internal_feat_file_row_bytes(h->g);
    if (!h->g.norms) {
        HIP_CHECK_RET(hipMemcpy(out, reinterpret_cast<const unsigned char *>(h->g.X) + r0 * h->g.row_bytes, rows * fb, hipMemcpyDeviceToHost));
        return LEANN_OK;
    }
    // contiguous device reads, interleaved on the host (a 2-D copy of 4-byte columns is one tiny DMA per row)
    const size_t fw = 2 * (size_t)h->g.feat_h, slab = (size_t)1 << 20;
    std::vector<unsigned char> fr(std::min(slab, std::max<size_t>(rows, 1)) * fw);
    std::vector<float> nr(std::min(slab, std::max<size_t>(rows, 1)));
    for (size_t s0 = 0; s0 < rows; s0 += slab) {
        const size_t m = std::min(slab, rows - s0);
        HIP_CHECK_RET(hipMemcpy(fr.data(), reinterpret_cast<const unsigned char *>(h->g.X) + (r0 + s0) * h->g.row_bytes, m * fw, hipMemcpyDeviceToHost));
        HIP_CHECK_RET(hipMemcpy(nr.data(), h->g.norms + r0 + s0, m * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < m; i++) {
            unsigned char *dst = out + (s0 + i) * fb;
            memcpy(dst, fr.data() + i * fw, fw);
            memcpy(dst + fw, &nr[i], 4);
            memset(dst + fw + 4, 0, fb - fw - 4);
        }
    }
    return LEANN_OK;
}
__global__ void bf16_to_f32_kernel(const uint16_t *__restrict__ in, size_t n, float *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __uint_as_float((uint32_t)in[i] << 16);
}

extern "C" int leann_recompute_build_index(const leann_recompute *r, int backend, size_t graph_degree, size_t complexity,
                                           leann_backend **out) {
    if (!r || !out) { leann_set_error("leann_recompute_build_index: null argument"); return LEANN_ERR_INVALID; }
    if (!r->parts.empty()) { leann_set_error("leann_recompute_build_index: build one index per part and join them with leann_sharded_from_handles"); return LEANN_ERR_UNSUPPORTED; }
    if (r->L != 1 || r->mask) {
        leann_set_error("recompute-on graph search needs one feature vector per passage (no token pooling): the feature-space "
                        "distance <f, W q> / ||W^T f|| is only exact for bf16-exact features");
        return LEANN_ERR_UNSUPPORTED;
    }
    HIP_CHECK_RET(hipSetDevice(r->device));
    const size_t n = r->n, ld = r->ld;
    float *E = nullptr, *norms = nullptr;
    HIP_CHECK_RET(hipMalloc((void **)&E, std::max<size_t>(n * ld, 4) * 4));
    HIP_CHECK_RET(hipMalloc((void **)&norms, std::max<size_t>(n, 1) * 4));
    int rc = LEANN_OK;
    for (size_t row0 = 0; row0 < n && rc == LEANN_OK; row0 += (size_t)4 << 20) { // transient embeddings, only for construction
        const size_t rows = std::min<size_t>((size_t)4 << 20, n - row0);
        rc = launch_encode(r, row0, rows, E + row0 * ld, nullptr, nullptr, 0, nullptr, norms + row0);
    }
    if (rc == LEANN_OK) HIP_CHECK_RET(hipDeviceSynchronize());
    leann_backend *h = nullptr;
    if (rc == LEANN_OK) rc = leann_backend_build_device(backend, E, n, r->d, ld, graph_degree, complexity, r->device, r->key_offset, 0, &h);
    if (rc != LEANN_OK) { (void)hipFree(E); (void)hipFree(norms); return rc; }
    // swap the rows: features + inline norm replace the embeddings
    const uint32_t hp4 = (uint32_t)((r->h + 3) & ~(size_t)3);
    const bool split = hp4 == 256; // 512-B rows of whole lines + the norms in an array of their own (GraphView::norms)
    const uint32_t row_bytes = split ? 2 * hp4 : (2 * hp4 + 4 + 7) & ~7u;
    unsigned char *rows_b = nullptr;
    HIP_CHECK_RET(hipMalloc((void **)&rows_b, std::max<size_t>(n, 1) * row_bytes));
    HIP_CHECK_RET(hipMemset(rows_b, 0, std::max<size_t>(n, 1) * row_bytes));
    const uint64_t total = (uint64_t)n * (hp4 + 2);
    if (total)
        hipLaunchKernelGGL(pack_feature_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, nullptr, r->F, norms, (uint64_t)n,
                           (uint32_t)r->h, hp4, row_bytes, rows_b);
    HIP_CHECK_RET(hipMalloc((void **)&h->Wf32, (size_t)hp4 * r->d * 4));
    HIP_CHECK_RET(hipMemset(h->Wf32, 0, (size_t)hp4 * r->d * 4));
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3((unsigned)((r->h * r->d + 255) / 256)), dim3(256), 0, nullptr, r->Wraw, r->h * r->d, h->Wf32);
    HIP_CHECK_RET(hipGetLastError());
    HIP_CHECK_RET(hipDeviceSynchronize());
    (void)hipFree(E);
    if (split) h->g.norms = norms; // (freed with the graph)
    else (void)hipFree(norms);
    h->g.X = reinterpret_cast<const float *>(rows_b);
    h->owns_rows = true;
    h->g.feat_h = hp4;
    h->g.row_bytes = row_bytes;
    *out = h;
    return LEANN_OK;
}

// raw feature rows [n x row_bytes] of a recompute-on index (tests: the oracle walks the same bytes)
extern "C" int leann_backend_feature_rows_export(const leann_backend *h, uint32_t *feat_h, uint32_t *row_bytes, void *out) {
    if (!h || !h->g.feat_h) { leann_set_error("not a recompute-on index"); return LEANN_ERR_INVALID; }
    if (feat_h) *feat_h = h->g.feat_h;
    if (row_bytes) *row_bytes = (uint32_t)leann_internal_feat_file_row_bytes(h->g); // the inline form, whatever the device layout
    if (out) return leann_internal_feat_rows_to_host(h, 0, (size_t)h->g.n, static_cast<unsigned char *>(out));
    return LEANN_OK;
}
