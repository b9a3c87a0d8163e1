// scan.hip — exact inner-product scan + top-k selection + cross-shard merge.
//
// Replaces the arithmetic of RecomputeSearcher::search once the embeddings exist
// (src/index/recompute.rs:96-109: N dot products, stable sort descending, take k) and provides
// the exact ground truth for recall@k.  Scores are raw dot products (higher = better, §3.2 of
// SURVEY.md), ties keep the lower position (Rust sort_by is stable, N4).
//
// Kernels:
//   score_mfma_kernel   S[q][i] = sum_j x_i[j]*q[j] as ONE k-ordered fmaf chain per (q, i)
//                       (oracle: orc_dot_seqfma; |.-orc_dot_seq| <= 1e-5 is tested) — f32 MFMA
//                       (v_mfma_f32_32x32x2_f32), 128 rows x 64 queries per workgroup from LDS tiles.
//   topk_scores_kernel  per (query, segment of SEG scores): bitonic sort of u64 keys
//                       (~orderable(score) << 32 | position) in LDS, emit the k smallest.
//   topk_keys_kernel    same on lists of keys (reduction rounds).
//   merge_topk_kernel   G-way merge of per-shard lists by (dist, key) — SURVEY.md §8e.
#include "common.cuh"
#include "../../include/leann_backend.h"
#include "search.cuh"
#include "internal.h"
#include <algorithm>
#include <vector>


// ------------------------------------------------------------------------------------------------
// Scratch of the synchronous entry points (exact scan, exact filtered search, allow-bitmap compaction).  hipMalloc + hipFree cost
// 20-50 us each and hipFree synchronises the device: ten of them per call were the 0.27 ms floor of a single-query exact filtered
// search.  Blocks of at most 64 MiB are therefore kept in a small process-wide free list (at most 512 MiB, best fit per device) and
// handed out again; larger blocks are allocated and freed per call.  A block is only released after the stream that used it has
// been synchronised, so the next user — on any stream or thread — finds it idle.
// ------------------------------------------------------------------------------------------------
#include <unordered_map>
namespace {
struct PoolBlock { void *p; size_t cap; int dev; };
std::mutex g_scratch_mu;
std::vector<PoolBlock> g_scratch_free;
std::unordered_map<void *, PoolBlock> g_scratch_out;
size_t g_scratch_free_bytes = 0;
constexpr size_t SCRATCH_BLOCK_MAX = (size_t)64 << 20, SCRATCH_TOTAL_MAX = (size_t)512 << 20;
} // namespace
int leann_internal_scratch_acquire(void **out, size_t bytes) {
    *out = nullptr;
    int dev = 0;
    HIP_CHECK_RET(hipGetDevice(&dev));
    bytes = std::max<size_t>((bytes + 255) / 256 * 256, 256);
    {
        std::lock_guard<std::mutex> lk(g_scratch_mu);
        size_t best = g_scratch_free.size();
        for (size_t i = 0; i < g_scratch_free.size(); i++) {
            const PoolBlock &b = g_scratch_free[i];
            if (b.dev == dev && b.cap >= bytes && b.cap <= 4 * bytes + 65536 && (best == g_scratch_free.size() || b.cap < g_scratch_free[best].cap)) best = i;
        }
        if (best < g_scratch_free.size()) {
            PoolBlock b = g_scratch_free[best];
            g_scratch_free.erase(g_scratch_free.begin() + (long)best);
            g_scratch_free_bytes -= b.cap;
            g_scratch_out[b.p] = b;
            *out = b.p;
            return LEANN_OK;
        }
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) { // give the cached blocks back and try once more
        std::vector<PoolBlock> drop;
        {
            std::lock_guard<std::mutex> lk(g_scratch_mu);
            drop.swap(g_scratch_free);
            g_scratch_free_bytes = 0;
        }
        for (auto &b : drop) (void)hipFree(b.p);
        (void)hipGetLastError();
        e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess) {
        leann_set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return LEANN_ERR_DEVICE;
    }
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    g_scratch_out[p] = PoolBlock{p, bytes, dev};
    *out = p;
    return LEANN_OK;
}
// the caller has synchronised every stream that touched the block
void leann_internal_scratch_release(void *p) {
    if (!p) return;
    PoolBlock b{p, 0, 0};
    bool keep = false;
    {
        std::lock_guard<std::mutex> lk(g_scratch_mu);
        auto it = g_scratch_out.find(p);
        if (it != g_scratch_out.end()) {
            b = it->second;
            g_scratch_out.erase(it);
            if (b.cap <= SCRATCH_BLOCK_MAX && g_scratch_free_bytes + b.cap <= SCRATCH_TOTAL_MAX) {
                g_scratch_free.push_back(b);
                g_scratch_free_bytes += b.cap;
                keep = true;
            }
        }
    }
    if (!keep) (void)hipFree(p);
}

// ------------------------------------------------------------------------------------------------
// score_mfma_kernel: S[q][i] for a tile of 128 rows x 64 queries on the f32 matrix cores.
// v_mfma_f32_32x32x2_f32 computes D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)) — a k-ordered f32 fmaf
// chain, bit for bit (MI355X_MICROARCH.md §Matrix cores), so every score is the same single chain
// acc = fmaf(x[j], q[j], acc), j ascending, that oracle/oracle.c:orc_dot_seqfma evaluates.
// A operand = queries (i = query), B operand = rows (j = row): the 32 lanes of a half-wave then
// hold 32 consecutive rows of one query -> 128-B coalesced stores into S[q][row].
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(16))) float f32x16;

// LIST: row i of the launch is X[idx[i]] (exact search over the allowed positions of a filter, compacted ascending by
// compact_allow below); scores, slab columns and emitted keys are then indexed by i, and finalize_scan_kernel maps i -> idx[i].
// FAST (dims a multiple of the 32-wide k-tile, 16-byte aligned rows and queries): the six 16-byte loads of a k-tile are unconditional —
// rows / queries past the end are clamped to the last valid one, their scores are never stored — so they issue back to back.  The
// general path's bounds / alignment branches made hipcc wait for each load before issuing the next (same registers for address and
// data): 6 memory round trips per k-tile in front of the MFMA loop.
template <bool LIST, bool FAST>
__global__ void __launch_bounds__(256) score_mfma_kernel(const float *__restrict__ X, uint64_t n, uint32_t d,
                                                         uint32_t ld, const float *__restrict__ Q, uint32_t nq,
                                                         uint32_t ldq, uint64_t row0, uint32_t n_rows,
                                                         float *__restrict__ S /* [nq x n_rows] */, CandEmit em = CandEmit{},
                                                         const uint32_t *__restrict__ idx = nullptr) {
    constexpr int BR = 128, BQ = 64, BK = 32;
    // LDS tiles [k][row] / [k][query], unpadded, XOR-swizzled columns: col ^ sw(k), sw(k) = 32 (k & 1) xor 8 ((k >> 2) & 7).
    // Reads (k = 2s + lh, 32 consecutive columns per half wave) and the transposing writes (8 lanes = 8 k-groups of one row,
    // 8 consecutive rows per instruction) are both bank-conflict free.
    __shared__ float sX[BK][BR];
    __shared__ float sQ[BK][BQ];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const uint32_t rbase = blockIdx.x * BR, qbase = blockIdx.y * BQ;
    f32x16 acc0, acc1; // query tile 0 / 1  x  row tile `wave`
#pragma unroll
    for (int i = 0; i < 16; i++) { acc0[i] = 0.f; acc1[i] = 0.f; }
    const bool xvec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
    const bool qvec = ((ldq & 3) == 0) && ((reinterpret_cast<uintptr_t>(Q) & 15) == 0);
    // staging map: thread -> (row / query = tid / 8 (+ 32 per pass), 4 consecutive k = 4 (tid % 8)): one wave instruction reads
    // 8 rows x 128 B = whole lines (two threads per row touched 32 quarter lines per instruction and cost ~4x the address work)
    const int sr = tid >> 3, kq = (tid & 7) * 4;
    auto load4 = [&](const float *base, bool ok, bool vec, uint32_t k, uint32_t lim) -> float4 {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) {
            if (vec && k + 3 < lim) v = *reinterpret_cast<const float4 *>(base + k);
            else {
                if (k + 0 < lim) v.x = base[k + 0];
                if (k + 1 < lim) v.y = base[k + 1];
                if (k + 2 < lim) v.z = base[k + 2];
                if (k + 3 < lim) v.w = base[k + 3];
            }
        }
        return v;
    };
    float4 xv[4], qv[2]; // the next k-tile travels in registers while the current one is multiplied
    uint32_t xrow[4];    // LIST: the positions of this thread's four rows
    if constexpr (LIST && !FAST) {
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const uint32_t row = rbase + sr + 32 * p;
            xrow[p] = row < n_rows ? idx[row0 + row] : 0u;
        }
    }
    const float *xp[4], *qp[2]; // FAST: this thread's four rows and two queries at its k offset
    if constexpr (FAST) {
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const uint32_t row = min(rbase + sr + 32 * p, n_rows - 1);
            xp[p] = X + (LIST ? (size_t)idx[row0 + row] : (size_t)(row0 + row)) * ld + kq;
        }
#pragma unroll
        for (int p = 0; p < 2; p++) qp[p] = Q + (size_t)min(qbase + sr + 32 * p, nq - 1) * ldq + kq;
    }
    auto gload = [&](uint32_t k0) {
        if constexpr (FAST) {
#pragma unroll
            for (int p = 0; p < 4; p++) xv[p] = *reinterpret_cast<const float4 *>(xp[p] + k0);
#pragma unroll
            for (int p = 0; p < 2; p++) qv[p] = *reinterpret_cast<const float4 *>(qp[p] + k0);
        } else {
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const uint32_t row = rbase + sr + 32 * p;
                xv[p] = load4(X + (LIST ? (size_t)xrow[p] : (size_t)(row0 + row)) * ld, row < n_rows, xvec, k0 + kq, d);
            }
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const uint32_t qi = qbase + sr + 32 * p;
                qv[p] = load4(Q + (size_t)qi * ldq, qi < nq, qvec, k0 + kq, d);
            }
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int k = kq + e, sw = (32 * (k & 1)) ^ (8 * ((k >> 2) & 7));
#pragma unroll
            for (int p = 0; p < 4; p++) sX[k][(sr + 32 * p) ^ sw] = e == 0 ? xv[p].x : e == 1 ? xv[p].y : e == 2 ? xv[p].z : xv[p].w;
#pragma unroll
            for (int p = 0; p < 2; p++) sQ[k][(sr + 32 * p) ^ sw] = e == 0 ? qv[p].x : e == 1 ? qv[p].y : e == 2 ? qv[p].z : qv[p].w;
        }
    };
    gload(0);
    for (uint32_t k0 = 0; k0 < d; k0 += BK) {
        lstore();
        __syncthreads();
        if (k0 + BK < d) gload(k0 + BK);
#pragma unroll
        for (int s2 = 0; s2 < BK / 2; s2++) {
            const int k = 2 * s2 + lh, sw = (32 * (k & 1)) ^ (8 * ((k >> 2) & 7));
            const float b = sX[k][(wave * 32 + l31) ^ sw];
            const float a0 = sQ[k][l31 ^ sw];
            const float a1 = sQ[k][(32 + l31) ^ sw];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc1, 0, 0, 0);
        }
        __syncthreads();
    }
    // C/D layout: col = lane & 31 (row of X), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (query)
    const uint32_t row = rbase + wave * 32 + l31;
    if (em.thr) { // candidate emission (CandEmit, internal.h): only the scores that reach their query's running k-th best leave the kernel
        bool any = false;
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const uint32_t qoff = (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            any |= acc0[reg] >= em.thr[qbase + qoff];          // thr has a slot (+inf) for every query of the last, partial tile
            any |= acc1[reg] >= em.thr[qbase + 32 + qoff];
        }
        if (any && row < n_rows) {
            const uint64_t pos = em.pos0 + row;
            if (!em.allow || ((em.allow[pos >> 3] >> (pos & 7)) & 1)) {
#pragma unroll
                for (int reg = 0; reg < 16; reg++) {
                    const uint32_t qoff = (reg & 3) + 8 * (reg >> 2) + 4 * lh;
#pragma unroll
                    for (int half = 0; half < 2; half++) {
                        const uint32_t q = qbase + half * 32 + qoff;
                        const float sv = half ? acc1[reg] : acc0[reg];
                        if (sv >= em.thr[q]) {
                            const uint32_t slot = atomicAdd(&em.cnt[q], 1u);
                            if (slot < em.cap) em.list[(size_t)q * em.cap + slot] = ((uint64_t)(~f32_orderable(sv)) << 32) | (uint32_t)pos;
                        }
                    }
                }
            }
        }
        return;
    }
    if (row < n_rows) {
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const uint32_t qoff = (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            const uint32_t q0 = qbase + qoff, q1 = qbase + 32 + qoff;
            if (q0 < nq) S[(size_t)q0 * n_rows + row] = acc0[reg];
            if (q1 < nq) S[(size_t)q1 * n_rows + row] = acc1[reg];
        }
    }
}

// Merge a launch's emitted candidates into the running best-k of each query (ascending keys), publish the new k-th best score as the
// next threshold, reset the counters.  One workgroup per query slot.
__global__ void __launch_bounds__(256) fold_candidates_kernel(uint64_t *__restrict__ list, uint32_t *__restrict__ cnt, uint32_t cap,
                                                              uint32_t k, uint32_t nq, uint64_t *__restrict__ best, float *__restrict__ thr,
                                                              uint32_t *__restrict__ overflow) {
    __shared__ uint64_t keys[SEG];
    const uint32_t q = blockIdx.x;
    if (q >= nq) { // unused query slot: never emits
        if (threadIdx.x == 0) { thr[q] = __uint_as_float(0x7F800000u); cnt[q] = 0; }
        return;
    }
    uint32_t m = cnt[q];
    if (m > cap) { // the list overflowed: the caller repeats the search on the slab path
        if (threadIdx.x == 0) atomicAdd(overflow, 1u);
        m = cap;
    }
    const uint64_t *src = list + (size_t)q * cap;
    if (m) {
        // sort width: the smallest power of two that takes the running best-k and all survivors (a few dozen to a few hundred keys
        // after the first chunks) — a 512-key sort has 45 compare-exchange rounds of 1 pair per thread, the full 2048 one 66 of 4
        int wsort = 64;
        while (wsort < (int)(k + m) && wsort < SEG) wsort <<= 1;
        for (int i = threadIdx.x; i < (int)k; i += blockDim.x) keys[i] = best[(size_t)q * k + i];
        for (uint32_t base = 0; base < m; base += wsort - k) {
            for (int i = threadIdx.x; i < wsort - (int)k; i += blockDim.x) {
                const uint32_t p = base + i;
                keys[k + i] = p < m ? src[p] : ~0ull;
            }
            bitonic_sort_lds(keys, wsort);
        }
        for (int i = threadIdx.x; i < (int)k; i += blockDim.x) best[(size_t)q * k + i] = keys[i];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const uint64_t kth = best[(size_t)q * k + (k - 1)];
        thr[q] = kth == ~0ull ? __uint_as_float(0xFF800000u) : orderable_f32(~(uint32_t)(kth >> 32)); // fewer than k so far: -inf
        cnt[q] = 0;
    }
}
int leann_internal_fold_candidates(const CandEmit &em, uint32_t k, uint32_t nq, uint32_t slots, uint64_t *best, uint32_t *d_overflow,
                                   hipStream_t st) {
    hipLaunchKernelGGL(fold_candidates_kernel, dim3(slots), dim3(256), 0, st, em.list, em.cnt, em.cap, k, nq, best, const_cast<float *>(em.thr),
                       d_overflow);
    HIP_CHECK_RET(hipGetLastError());
    return LEANN_OK;
}

// scores S[q][n_rows] -> cand[q][seg][k]; key = ~orderable(score) << 32 | (row0 + row)   (position < 2^32)
// `best` (optional): per query the k best keys seen in earlier chunks, ascending; best[q][k-1] is a bound that
// every member of the final top-k must beat, so a segment without any key below it is skipped unsorted
// (after a few hundred thousand rows that is > 95 % of the segments).
__global__ void __launch_bounds__(256) topk_scores_kernel(const float *__restrict__ S, uint32_t n_rows, uint64_t row0,
                                                          const uint8_t *__restrict__ allow, uint32_t k,
                                                          uint64_t *__restrict__ cand, uint32_t cand_stride_q,
                                                          uint32_t seg_off, const uint64_t *__restrict__ best = nullptr) {
    __shared__ uint64_t keys[SEG];
    const uint32_t seg = blockIdx.x, q = blockIdx.y;
    const float *s = S + (size_t)q * n_rows;
    const uint64_t bound = best ? best[(size_t)q * k + (k - 1)] : ~0ull;
    int survivor = 0;
    for (int i = threadIdx.x; i < SEG; i += blockDim.x) {
        uint32_t row = seg * SEG + i;
        uint64_t key = ~0ull;
        if (row < n_rows) {
            uint64_t pos = row0 + row;
            bool ok = allow ? ((allow[pos >> 3] >> (pos & 7)) & 1) : true;
            if (ok) key = ((uint64_t)(~f32_orderable(s[row])) << 32) | (uint32_t)pos;
        }
        keys[i] = key;
        survivor |= (key < bound);
    }
    uint64_t *outp = cand + (size_t)q * cand_stride_q + (size_t)(seg_off + seg) * k;
    if (!__syncthreads_or(survivor)) {
        for (int i = threadIdx.x; i < (int)k; i += blockDim.x) outp[i] = ~0ull;
        return;
    }
    bitonic_sort_lds(keys, SEG);
    uint64_t *out = cand + (size_t)q * cand_stride_q + (size_t)(seg_off + seg) * k;
    for (int i = threadIdx.x; i < (int)k; i += blockDim.x) out[i] = keys[i];
}

// lists in[q][m] -> out[q][nseg][k]
__global__ void __launch_bounds__(256) topk_keys_kernel(const uint64_t *__restrict__ in, uint32_t m, uint32_t in_stride_q,
                                                        uint32_t k, uint64_t *__restrict__ out, uint32_t out_stride_q) {
    __shared__ uint64_t keys[SEG];
    const uint32_t seg = blockIdx.x, q = blockIdx.y;
    const uint64_t *src = in + (size_t)q * in_stride_q;
    for (int i = threadIdx.x; i < SEG; i += blockDim.x) {
        uint32_t p = seg * SEG + i;
        keys[i] = p < m ? src[p] : ~0ull;
    }
    bitonic_sort_lds(keys, SEG);
    uint64_t *dst = out + (size_t)q * out_stride_q + (size_t)seg * k;
    for (int i = threadIdx.x; i < (int)k; i += blockDim.x) dst[i] = keys[i];
}

__global__ void finalize_scan_kernel(const uint64_t *__restrict__ keys, uint32_t stride_q, uint32_t nq, uint32_t k,
                                     uint64_t key_offset, uint64_t *__restrict__ out_keys, float *__restrict__ out_scores,
                                     uint32_t *__restrict__ out_counts, const uint32_t *__restrict__ idx, int as_dist) {
    uint32_t q = blockIdx.x;
    if (q >= nq) return;
    uint32_t cnt = 0;
    for (uint32_t i = threadIdx.x; i < k; i += blockDim.x) {
        uint64_t key = keys[(size_t)q * stride_q + i];
        size_t o = (size_t)q * k + i;
        if (key != ~0ull) {
            const uint32_t pos = (uint32_t)(key & 0xFFFFFFFFull);
            const float sc = orderable_f32(~(uint32_t)(key >> 32));
            out_keys[o] = (idx ? idx[pos] : pos) + key_offset;
            out_scores[o] = as_dist ? 1.0f - sc : sc; // as_dist: the backends' distance 1 - <x, q> (search.cuh)
        } else {
            out_keys[o] = ~0ull;
            out_scores[o] = __uint_as_float(as_dist ? 0x7F800000u : 0xFF800000u); // +inf distance / -inf score
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (uint32_t i = 0; i < k; i++) cnt += keys[(size_t)q * stride_q + i] != ~0ull;
        out_counts[q] = cnt;
    }
}


// one launch of score_mfma_kernel (row list optional); picks the FAST instantiation when the shapes allow
static void launch_score(hipStream_t st, const float *X, size_t rows, size_t dims, size_t ld, const float *Q, size_t nq, size_t ldq, float *S,
                         const CandEmit &em, const uint32_t *idx) {
    dim3 g1((unsigned)((rows + 127) / 128), (unsigned)((nq + 63) / 64));
    const bool fast = dims % 32 == 0 && ld % 4 == 0 && ldq % 4 == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0 &&
                      (reinterpret_cast<uintptr_t>(Q) & 15) == 0 && rows > 0 && nq > 0;
#define LEANN_SCORE_LAUNCH(LIST_, FAST_)                                                                                                  \
    hipLaunchKernelGGL((score_mfma_kernel<LIST_, FAST_>), g1, dim3(256), 0, st, X, (uint64_t)rows, (uint32_t)dims, (uint32_t)ld, Q,       \
                       (uint32_t)nq, (uint32_t)ldq, (uint64_t)0, (uint32_t)rows, S, em, idx)
    if (idx) { if (fast) LEANN_SCORE_LAUNCH(true, true); else LEANN_SCORE_LAUNCH(true, false); }
    else { if (fast) LEANN_SCORE_LAUNCH(false, true); else LEANN_SCORE_LAUNCH(false, false); }
#undef LEANN_SCORE_LAUNCH
}

// S[q][row] = <X[row], Q[q]> for all rows (k-ordered f32 fmaf chains on the f32 matrix cores); also used to project
// queries into feature space (X = encoder weights) for the recompute-on graph search
int leann_internal_score(const float *X, size_t rows, size_t dims, size_t ld, const float *d_queries, size_t nq, size_t ldq, float *S,
                         hipStream_t st) {
    launch_score(st, X, rows, dims, ld, d_queries, nq, ldq, S, CandEmit{}, nullptr);
    HIP_CHECK_RET(hipGetLastError());
    return LEANN_OK;
}

__global__ void update_best_kernel(const uint64_t *__restrict__ cand, uint32_t cand_stride_q, uint32_t seg_off, uint32_t n_segs,
                                   uint32_t k, uint64_t *__restrict__ best);
// one chunk: scores of rows [0, rows) of Xbase against all queries, per-segment top-k appended to cand
int leann_internal_scan_chunk(const float *Xbase, size_t rows, size_t dims, size_t ld, const float *d_queries, size_t nq, uint32_t k,
                              const uint8_t *allow, uint64_t pos0, float *S, uint64_t *cand, size_t cand_len, size_t seg_off,
                              hipStream_t st, size_t *segs_out, hipEvent_t mid = nullptr, uint64_t *best = nullptr,
                              const uint32_t *idx = nullptr) {
    launch_score(st, Xbase, rows, dims, ld, d_queries, nq, dims, S, CandEmit{}, idx);
    if (mid) (void)hipEventRecord(mid, st);
    unsigned segs = (unsigned)((rows + SEG - 1) / SEG);
    hipLaunchKernelGGL(topk_scores_kernel, dim3(segs, (unsigned)nq), dim3(256), 0, st, S, (uint32_t)rows, pos0, allow, k, cand,
                       (uint32_t)cand_len, (uint32_t)seg_off, (const uint64_t *)best);
    if (best)
        hipLaunchKernelGGL(update_best_kernel, dim3((unsigned)nq), dim3(256), 0, st, cand, (uint32_t)cand_len, (uint32_t)seg_off, segs, k, best);
    HIP_CHECK_RET(hipGetLastError());
    *segs_out = segs;
    return LEANN_OK;
}
// merge this chunk's segment winners into the running per-query best-k (ascending keys)
__global__ void __launch_bounds__(256) update_best_kernel(const uint64_t *__restrict__ cand, uint32_t cand_stride_q, uint32_t seg_off,
                                                          uint32_t n_segs, uint32_t k, uint64_t *__restrict__ best) {
    __shared__ uint64_t keys[SEG];
    const uint32_t q = blockIdx.x;
    const uint64_t *src = cand + (size_t)q * cand_stride_q + (size_t)seg_off * k;
    const uint32_t m = n_segs * k;
    // running selection: fold the candidate list through the LDS sorter, as narrow as the list allows (see fold_candidates_kernel)
    int wsort = 64;
    while (wsort < (int)(k + m) && wsort < SEG) wsort <<= 1;
    for (int i = threadIdx.x; i < (int)k; i += blockDim.x) keys[i] = best[(size_t)q * k + i];
    for (uint32_t base = 0; base < m; base += wsort - k) {
        for (int i = threadIdx.x; i < wsort - (int)k; i += blockDim.x) {
            uint32_t p = base + i;
            keys[k + i] = p < m ? src[p] : ~0ull;
        }
        bitonic_sort_lds(keys, wsort);
    }
    for (int i = threadIdx.x; i < (int)k; i += blockDim.x) best[(size_t)q * k + i] = keys[i];
}

// segment top-k of a score slab S[nq][rows] (scores produced elsewhere, e.g. the fused recompute kernel);
// `best` [nq x k] carries the running best keys across chunks (0xFF-filled before the first chunk).
int leann_internal_topk_chunk(const float *S, size_t rows, size_t nq, uint32_t k, const uint8_t *allow, uint64_t pos0, uint64_t *cand,
                              size_t cand_len, size_t seg_off, hipStream_t st, size_t *segs_out, uint64_t *best) {
    unsigned segs = (unsigned)((rows + SEG - 1) / SEG);
    hipLaunchKernelGGL(topk_scores_kernel, dim3(segs, (unsigned)nq), dim3(256), 0, st, S, (uint32_t)rows, pos0, allow, k, cand,
                       (uint32_t)cand_len, (uint32_t)seg_off, (const uint64_t *)best);
    if (best)
        hipLaunchKernelGGL(update_best_kernel, dim3((unsigned)nq), dim3(256), 0, st, cand, (uint32_t)cand_len, (uint32_t)seg_off, segs, k, best);
    HIP_CHECK_RET(hipGetLastError());
    *segs_out = segs;
    return LEANN_OK;
}
// reduction rounds until one segment per query remains, then keys -> (position + key_offset, score)
int leann_internal_scan_finish_ex(uint64_t *candA, uint64_t *candB, size_t cand_len, size_t total_segs, size_t nq, uint32_t k, uint64_t key_offset,
                          uint64_t *d_keys, float *d_scores, uint32_t *d_counts, hipStream_t st, const uint32_t *idx, int as_dist);
int leann_internal_scan_finish(uint64_t *candA, uint64_t *candB, size_t cand_len, size_t total_segs, size_t nq, uint32_t k,
                               uint64_t key_offset, uint64_t *d_keys, float *d_scores, uint32_t *d_counts, hipStream_t st) {
    return leann_internal_scan_finish_ex(candA, candB, cand_len, total_segs, nq, k, key_offset, d_keys, d_scores, d_counts, st, nullptr, 0);
}
// idx: keys are indices into a row list (score_mfma_kernel<true>); as_dist: report 1 - score, the backends' distance
int leann_internal_scan_finish_ex(uint64_t *candA, uint64_t *candB, size_t cand_len, size_t total_segs, size_t nq, uint32_t k, uint64_t key_offset,
                          uint64_t *d_keys, float *d_scores, uint32_t *d_counts, hipStream_t st, const uint32_t *idx, int as_dist) {
    size_t m = total_segs * k;
    uint64_t *src = candA, *dst = candB;
    while (m > k) {
        unsigned segs = (unsigned)((m + SEG - 1) / SEG);
        hipLaunchKernelGGL(topk_keys_kernel, dim3(segs, (unsigned)nq), dim3(256), 0, st, src, (uint32_t)m, (uint32_t)cand_len, k, dst,
                           (uint32_t)cand_len);
        m = (size_t)segs * k;
        std::swap(src, dst);
    }
    hipLaunchKernelGGL(finalize_scan_kernel, dim3((unsigned)nq), dim3(64), 0, st, src, (uint32_t)cand_len, (uint32_t)nq, k, key_offset,
                       d_keys, d_scores, d_counts, idx, as_dist);
    HIP_CHECK_RET(hipGetLastError());
    return LEANN_OK;
}

// idx != null: exact search over the n listed positions of d_rows (ascending; no allow mask), keys = idx[i] + key_offset.
// as_dist: report the backends' distance 1 - score (ascending) instead of the raw score.
static int scan_topk_impl(const float *d_rows, size_t n, size_t dims, size_t ld, const float *d_queries, size_t nq, size_t top_k,
                          const uint8_t *d_allow_mask, uint64_t key_offset, uint64_t *d_keys, float *d_scores, uint32_t *d_counts,
                          hipStream_t st, const uint32_t *idx, int as_dist);
extern "C" int leann_scan_topk_device(const float *d_rows, size_t n, size_t dims, size_t ld, const float *d_queries,
                                      size_t nq, size_t top_k, const uint8_t *d_allow_mask, uint64_t key_offset,
                                      uint64_t *d_keys, float *d_scores, uint32_t *d_counts, void *stream) {
    if (!d_queries || !d_keys || !d_scores || !d_counts || dims == 0 || ld < dims || top_k == 0 || top_k > SEG / 2 ||
        n >= (1ull << 32)) {
        leann_set_error("leann_scan_topk_device: invalid arguments (n=%zu dims=%zu top_k=%zu)", n, dims, top_k);
        return LEANN_ERR_INVALID;
    }
    if (nq == 0) return LEANN_OK;
    return scan_topk_impl(d_rows, n, dims, ld, d_queries, nq, top_k, d_allow_mask, key_offset, d_keys, d_scores, d_counts,
                          (hipStream_t)stream, nullptr, 0);
}
static int scan_topk_impl(const float *d_rows, size_t n, size_t dims, size_t ld, const float *d_queries, size_t nq, size_t top_k,
                          const uint8_t *d_allow_mask, uint64_t key_offset, uint64_t *d_keys, float *d_scores, uint32_t *d_counts,
                          hipStream_t st, const uint32_t *idx, int as_dist) {
    const uint32_t k = (uint32_t)top_k;
    // Pass 1 (emission): a 64k-row slab + segment top-k fixes a first k-th best per query; the following launches (448k rows, then
    // everything else) emit only the scores that reach it (CandEmit).  A candidate list that overflows (adversarial order) makes the
    // call repeat on pass 2, the slab path: geometric chunks 128k, 512k, 2M, ... (slab <= 2 GiB) with threshold-pruned segment sorts.
    for (int pass = (n > ((size_t)64 << 10) && !leann_knobs().no_emit) ? 1 : 2; pass <= 2; pass++) {
        const bool emit = pass == 1;
        std::vector<std::pair<size_t, size_t>> chunks;
        {
            const size_t cap_rows = std::max<size_t>(SEG, (((size_t)1 << 29) / std::max<size_t>(nq, 1)) / SEG * SEG);
            size_t pos = 0, len = std::min(cap_rows, emit ? (size_t)64 << 10 : (size_t)128 << 10);
            while (pos < n) {
                const size_t rows = std::min(len, n - pos);
                chunks.emplace_back(pos, rows);
                pos += rows;
                if (emit) len = chunks.size() == 1 ? (size_t)448 << 10 : n;
                else len = std::min(cap_rows, len * 4);
            }
        }
        const size_t n_chunks = chunks.size();
        size_t slab_rows = SEG, total_segs = 0;
        for (size_t c = 0; c < n_chunks; c++)
            if (!emit || c == 0) { slab_rows = std::max(slab_rows, chunks[c].second); total_segs += (chunks[c].second + SEG - 1) / SEG; }
        total_segs = std::max<size_t>(total_segs, 1);
        const size_t cand_len = std::max<size_t>(total_segs * k, k); // per query
        const size_t slots = (nq + 63) / 64 * 64;                     // score_mfma_kernel reads a threshold for every query of a tile
        const uint32_t EMIT_CAP = std::max<uint32_t>(8192, 32 * k); // expected survivors per query ~ k * rows / rows_seen (x19 after 512k of 10M rows)
        float *S = nullptr;
        uint64_t *candA = nullptr, *candB = nullptr, *best = nullptr;
        unsigned char *emb = nullptr; // [slots f32 thr | slots u32 cnt | u32 overflow | pad | slots x EMIT_CAP u64]
        const size_t em_hdr = (slots * 8 + 4 + 255) / 256 * 256;
        struct Scratch { // every early return (a failed allocation, a launch error) gives the pass's scratch back
            void **p[5];
            hipStream_t st;
            ~Scratch() {
                (void)hipStreamSynchronize(st); // nothing may still be running on a block when it goes back to the pool
                for (void **q : p) { leann_internal_scratch_release(*q); *q = nullptr; }
            }
        } scratch{{(void **)&S, (void **)&candA, (void **)&candB, (void **)&best, (void **)&emb}, st};
        if (int e = leann_internal_scratch_acquire((void **)&S, sizeof(float) * nq * slab_rows)) return e;
        if (int e = leann_internal_scratch_acquire((void **)&candA, sizeof(uint64_t) * nq * cand_len)) return e;
        if (int e = leann_internal_scratch_acquire((void **)&candB, sizeof(uint64_t) * nq * cand_len)) return e;
        if (int e = leann_internal_scratch_acquire((void **)&best, sizeof(uint64_t) * nq * k)) return e;
        HIP_CHECK_RET(hipMemsetAsync(candA, 0xFF, sizeof(uint64_t) * nq * cand_len, st));
        HIP_CHECK_RET(hipMemsetAsync(best, 0xFF, sizeof(uint64_t) * nq * k, st));
        CandEmit em{};
        uint32_t *d_overflow = nullptr;
        if (emit) {
            if (int e = leann_internal_scratch_acquire((void **)&emb, em_hdr + sizeof(uint64_t) * slots * EMIT_CAP)) return e;
            HIP_CHECK_RET(hipMemsetAsync(emb, 0, em_hdr, st));
            em.thr = reinterpret_cast<const float *>(emb);
            em.cnt = reinterpret_cast<uint32_t *>(emb + slots * 4);
            d_overflow = reinterpret_cast<uint32_t *>(emb + slots * 8);
            em.list = reinterpret_cast<uint64_t *>(emb + em_hdr);
            em.cap = EMIT_CAP;
            em.allow = d_allow_mask;
        }
        size_t seg_off = 0;
        int rc = LEANN_OK;
        for (size_t c = 0; c < n_chunks && rc == LEANN_OK; c++) {
            const size_t row0 = chunks[c].first, rows = chunks[c].second;
            size_t segs = 0;
            if (emit && c > 0) {
                em.pos0 = row0;
                launch_score(st, idx ? d_rows : d_rows + row0 * ld, rows, dims, ld, d_queries, nq, dims, S, em, idx ? idx + row0 : nullptr);
            } else {
                rc = leann_internal_scan_chunk(idx ? d_rows : d_rows + row0 * ld, rows, dims, ld, d_queries, nq, k, d_allow_mask, row0, S, candA,
                                               cand_len, seg_off, st, &segs, nullptr, (n_chunks > 1 || emit) ? best : nullptr,
                                               idx ? idx + row0 : nullptr);
            }
            if (rc == LEANN_OK && emit) rc = leann_internal_fold_candidates(em, k, (uint32_t)nq, (uint32_t)slots, best, d_overflow, st);
            seg_off += segs;
        }
        if (rc == LEANN_OK) {
            if (emit) rc = leann_internal_scan_finish_ex(best, candB, k, 1, nq, k, key_offset, d_keys, d_scores, d_counts, st, idx, as_dist); // best = exact running top-k
            else rc = leann_internal_scan_finish_ex(candA, candB, cand_len, total_segs, nq, k, key_offset, d_keys, d_scores, d_counts, st, idx, as_dist);
        }
        // scratch is plain hipMalloc memory: wait for the stream before giving it back (this entry point
        // is synchronous; the stream-ordered allocator proved unreliable on this stack)
        (void)hipStreamSynchronize(st);
        uint32_t ov = 0;
        if (emit && rc == LEANN_OK && hipMemcpy(&ov, d_overflow, 4, hipMemcpyDeviceToHost) != hipSuccess) rc = LEANN_ERR_DEVICE;
        if (rc != LEANN_OK || !emit || ov == 0) return rc; // (~Scratch frees)
    }
    return LEANN_OK;
}

// ------------------------------------------------------------------------------------------------
// Exact filtered search (the selective end): a filter that allows only a small share of the rows starves the
// graph walk (1 % allowed: filtered recall@10 0.86 at ef = 256), while scanning just the allowed rows is both exact and cheaper
// (100k rows x 768 = 0.15 GFLOP per query).  compact_allow turns the bitmap into the ascending list of allowed positions
// (count -> scan of the block counts -> scatter), score_mfma_kernel<true> gathers the listed rows.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t allow_word(const uint8_t *__restrict__ allow, uint64_t n, uint64_t w) { // bits [32 w, 32 w + 32) below n
    const uint64_t nbytes = (n + 7) >> 3, b0 = w * 4;
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < 4; i++)
        if (b0 + i < nbytes) v |= (uint32_t)allow[b0 + i] << (8 * i);
    const uint64_t lo = w * 32;
    if (lo >= n) return 0;
    if (n - lo < 32) v &= (1u << (uint32_t)(n - lo)) - 1u;
    return v;
}
__global__ void __launch_bounds__(256) allow_count_kernel(const uint8_t *__restrict__ allow, uint64_t n, uint32_t *__restrict__ blk) {
    __shared__ uint32_t part[4];
    uint32_t c = __popc(allow_word(allow, n, (uint64_t)blockIdx.x * 256 + threadIdx.x));
    for (int o = 32; o; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blk[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
__global__ void __launch_bounds__(1024) allow_offsets_kernel(uint32_t *__restrict__ blk, uint32_t nb) { // in place: counts -> exclusive offsets, blk[nb] = total
    __shared__ uint32_t sh[1024];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nb; base += 1024) {
        const uint32_t i = base + threadIdx.x, v = i < nb ? blk[i] : 0u;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const uint32_t t = threadIdx.x >= (uint32_t)o ? sh[threadIdx.x - o] : 0u;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nb) blk[i] = carry + sh[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += sh[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) blk[nb] = carry;
}
__global__ void __launch_bounds__(256) allow_scatter_kernel(const uint8_t *__restrict__ allow, uint64_t n, const uint32_t *__restrict__ blk,
                                                            uint32_t *__restrict__ out) {
    __shared__ uint32_t sh[256];
    const uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t v = allow_word(allow, n, w);
    const uint32_t c = __popc(v);
    sh[threadIdx.x] = c;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const uint32_t t = threadIdx.x >= (uint32_t)o ? sh[threadIdx.x - o] : 0u;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t dst = blk[blockIdx.x] + sh[threadIdx.x] - c;
    while (v) {
        const uint32_t b = __ffs(v) - 1;
        out[dst++] = (uint32_t)(w * 32 + b);
        v &= v - 1;
    }
}
// bitmap over n positions -> *d_list (from the scratch pool: give it back with leann_internal_scratch_release once the stream that used
// it is synchronised), ascending positions, *n_list; synchronises the stream once for the count
int leann_internal_compact_allow(const uint8_t *d_allow, size_t n, uint32_t **d_list, size_t *n_list, hipStream_t st) {
    const uint32_t nb = (uint32_t)((n + 8191) / 8192);
    uint32_t *blk = nullptr, total = 0;
    *d_list = nullptr;
    *n_list = 0;
    if (int e0 = leann_internal_scratch_acquire((void **)&blk, sizeof(uint32_t) * ((size_t)nb + 1))) return e0;
    hipLaunchKernelGGL(allow_count_kernel, dim3(nb), dim3(256), 0, st, d_allow, (uint64_t)n, blk);
    hipLaunchKernelGGL(allow_offsets_kernel, dim3(1), dim3(1024), 0, st, blk, nb);
    hipError_t e = hipMemcpyAsync(&total, blk + nb, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess && total && leann_internal_scratch_acquire((void **)d_list, sizeof(uint32_t) * (size_t)total) != LEANN_OK) e = hipErrorOutOfMemory;
    if (e == hipSuccess && total) {
        hipLaunchKernelGGL(allow_scatter_kernel, dim3(nb), dim3(256), 0, st, d_allow, (uint64_t)n, blk, *d_list);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(st); // blk goes back to the pool below
    }
    if (e != hipSuccess) (void)hipStreamSynchronize(st);
    leann_internal_scratch_release(blk);
    if (e != hipSuccess) {
        leann_internal_scratch_release(*d_list);
        *d_list = nullptr;
        leann_set_error("HIP error: %s (compact_allow)", hipGetErrorString(e));
        return LEANN_ERR_DEVICE;
    }
    *n_list = total;
    return LEANN_OK;
}
__global__ void fill_empty_results_kernel(uint64_t *__restrict__ keys, float *__restrict__ dists, uint32_t *__restrict__ counts, size_t nq, size_t k) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nq * k) { keys[i] = ~0ull; dists[i] = __uint_as_float(0x7F800000u); }
    if (i < nq) counts[i] = 0;
}
// exact search over an already compacted list of m allowed positions (api.hip: registered filters)
int leann_internal_filtered_exact_list(const float *d_rows, size_t dims, size_t ld, const float *d_queries, size_t nq, size_t top_k,
                                       const uint32_t *d_list, size_t m, uint64_t key_offset, uint64_t *d_keys, float *d_dists,
                                       uint32_t *d_counts, hipStream_t st) {
    if (top_k == 0 || top_k > SEG / 2) {
        leann_set_error("exact filtered search: top_k must be in [1, %d] (top_k=%zu)", SEG / 2, top_k);
        return LEANN_ERR_INVALID;
    }
    if (m == 0) {
        const size_t cells = std::max(nq * top_k, nq);
        hipLaunchKernelGGL(fill_empty_results_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, st, d_keys, d_dists, d_counts, nq, top_k);
        HIP_CHECK_RET(hipGetLastError());
        return LEANN_OK;
    }
    return scan_topk_impl(d_rows, m, dims, ld, d_queries, nq, top_k, nullptr, key_offset, d_keys, d_dists, d_counts, st, d_list, 1);
}
// api.hip: leann_backend_search_filtered_exact_batch_device.  Rows [n x dims] (leading dimension ld), one bitmap for the batch
// (allow_stride == 0) or one per query.
int leann_internal_filtered_exact(const float *d_rows, size_t n, size_t dims, size_t ld, const float *d_queries, size_t nq, size_t top_k,
                                  const uint8_t *d_allow, size_t allow_stride, uint64_t key_offset, uint64_t *d_keys, float *d_dists,
                                  uint32_t *d_counts, hipStream_t st) {
    if (top_k == 0 || top_k > SEG / 2 || n >= (1ull << 32)) {
        leann_set_error("exact filtered search: top_k must be in [1, %d] and the index smaller than 2^32 rows (top_k=%zu n=%zu)", SEG / 2, top_k, n);
        return LEANN_ERR_INVALID;
    }
    const size_t groups = allow_stride ? nq : 1, per = allow_stride ? 1 : nq;
    for (size_t g = 0; g < groups; g++) {
        uint32_t *list = nullptr;
        size_t m = 0;
        int rc = leann_internal_compact_allow(d_allow + g * allow_stride, n, &list, &m, st);
        if (rc != LEANN_OK) return rc;
        uint64_t *ok = d_keys + g * per * top_k;
        float *od = d_dists + g * per * top_k;
        uint32_t *oc = d_counts + g * per;
        if (m == 0) {
            const size_t cells = std::max(per * top_k, per);
            hipLaunchKernelGGL(fill_empty_results_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, st, ok, od, oc, per, top_k);
            HIP_CHECK_RET(hipGetLastError());
            continue;
        }
        rc = scan_topk_impl(d_rows, m, dims, ld, d_queries + g * per * dims, per, top_k, nullptr, key_offset, ok, od, oc, st, list, 1);
        (void)hipStreamSynchronize(st);
        leann_internal_scratch_release(list);
        if (rc != LEANN_OK) return rc;
    }
    return LEANN_OK;
}

// ------------------------------------------------------------------------------------------------
// Cross-shard merge.  One wave per query; n_shards*k_in <= 12288 entries (144 KiB of LDS).
// Order: ascending (orderable(dist), key)  [descending != 0: descending score, ascending key].
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) merge_topk_kernel(const unsigned char *__restrict__ keys, const unsigned char *__restrict__ dists,
                                                        const unsigned char *__restrict__ counts, size_t kstride, size_t dstride,
                                                        size_t cstride, uint32_t n_shards,
                                                        uint32_t nq, uint32_t k_in, uint32_t k_out, int descending,
                                                        uint64_t *__restrict__ out_keys, float *__restrict__ out_dists,
                                                        uint32_t *__restrict__ out_counts) {
    // shard s's arrays start s * {k,d,c}stride BYTES after the base pointers: three contiguous [n_shards x nq x k_in] arrays, or one
    // packed {keys | dists | counts} block per shard as the all-gather delivers them (shard.hip).
    // rank-by-counting: total order has no duplicates across shards (keys are global positions)
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    uint32_t *sd = reinterpret_cast<uint32_t *>(sm);               // [n_shards*k_in] orderable dist
    uint64_t *sk = reinterpret_cast<uint64_t *>(sm + (((size_t)n_shards * k_in * 4 + 7) & ~(size_t)7));
    const uint32_t q = blockIdx.x, total = n_shards * k_in;
    for (uint32_t i = threadIdx.x; i < total; i += 64) {
        uint32_t s = i / k_in, j = i % k_in;
        const size_t src = (size_t)q * k_in + j;
        bool valid = j < reinterpret_cast<const uint32_t *>(counts + s * cstride)[q];
        uint32_t od = f32_orderable(reinterpret_cast<const float *>(dists + s * dstride)[src]);
        sd[i] = valid ? (descending ? ~od : od) : 0xFFFFFFFFu;
        sk[i] = valid ? reinterpret_cast<const uint64_t *>(keys + s * kstride)[src] : ~0ull;
    }
    __syncthreads();
    uint32_t nvalid = 0;
    for (uint32_t s = 0; s < n_shards; s++) nvalid += min(reinterpret_cast<const uint32_t *>(counts + s * cstride)[q], k_in);
    for (uint32_t i = threadIdx.x; i < total; i += 64) {
        uint32_t di = sd[i];
        uint64_t ki = sk[i];
        if (ki == ~0ull && di == 0xFFFFFFFFu) continue;
        uint32_t rank = 0;
        for (uint32_t j = 0; j < total; j++) {
            uint32_t dj = sd[j];
            uint64_t kj = sk[j];
            rank += (dj < di) || (dj == di && kj < ki);
        }
        if (rank < k_out) {
            out_keys[(size_t)q * k_out + rank] = ki;
            uint32_t od = descending ? ~di : di;
            out_dists[(size_t)q * k_out + rank] = orderable_f32(od);
        }
    }
    uint32_t nout = min(nvalid, k_out);
    for (uint32_t i = threadIdx.x; i < k_out; i += 64) {
        if (i >= nout) {
            out_keys[(size_t)q * k_out + i] = ~0ull;
            out_dists[(size_t)q * k_out + i] = __uint_as_float(descending ? 0xFF800000u : 0x7F800000u);
        }
    }
    if (threadIdx.x == 0) out_counts[q] = nout;
}

int leann_internal_merge_strided(const void *keys, const void *dists, const void *counts, size_t kstride, size_t dstride, size_t cstride,
                                 size_t n_shards, size_t nq, size_t k_in, size_t k_out, int descending, uint64_t *d_out_keys,
                                 float *d_out_dists, uint32_t *d_out_counts, hipStream_t st) {
    if (!keys || !dists || !counts || !d_out_keys || !d_out_dists || !d_out_counts || n_shards == 0 ||
        k_in == 0 || k_out == 0 || n_shards * k_in > 12288) {
        leann_set_error("leann_merge_topk_device: invalid arguments (shards=%zu k_in=%zu k_out=%zu)", n_shards, k_in, k_out);
        return LEANN_ERR_INVALID;
    }
    if (nq == 0) return LEANN_OK;
    size_t total = n_shards * k_in;
    size_t lds = ((total * 4 + 7) & ~(size_t)7) + total * 8;
    if (lds > 64 * 1024) // > 4096 entries (e.g. 8 shards x fetch_k = 5 x 200 of a filtered / hybrid query, searcher.rs:129-133): up to 144 KiB of the 160
        HIP_CHECK_RET(hipFuncSetAttribute((const void *)merge_topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(merge_topk_kernel, dim3((unsigned)nq), dim3(64), lds, st, (const unsigned char *)keys, (const unsigned char *)dists,
                       (const unsigned char *)counts, kstride, dstride, cstride, (uint32_t)n_shards, (uint32_t)nq, (uint32_t)k_in,
                       (uint32_t)k_out, descending, d_out_keys, d_out_dists, d_out_counts);
    HIP_CHECK_RET(hipGetLastError());
    return LEANN_OK;
}

extern "C" int leann_merge_topk_device(const uint64_t *d_keys, const float *d_dists, const uint32_t *d_counts,
                                       size_t n_shards, size_t nq, size_t k_in, size_t k_out, int descending,
                                       uint64_t *d_out_keys, float *d_out_dists, uint32_t *d_out_counts, void *stream) {
    return leann_internal_merge_strided(d_keys, d_dists, d_counts, nq * k_in * 8, nq * k_in * 4, nq * 4, n_shards, nq, k_in, k_out, descending,
                                        d_out_keys, d_out_dists, d_out_counts, (hipStream_t)stream);
}
