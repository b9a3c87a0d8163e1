// build.hip — on-GPU index construction (needed to HAVE the 1M/10M indexes).
//
// Replaces the arithmetic behind
//     hnsw::build_index / add_to_index   src/backend/hnsw.rs:96-191   (usearch add loop, out of tree)
//     diskann::build_index               src/backend/diskann.rs:70-105 (diskann-rs Vamana, out of tree)
// with batched insertion: points enter in a fixed pseudo-random permutation of their positions (see
// insertion_order below; hnsw.rs:128-130 adds row by row) in batches no larger than the graph built so
// far; every point of a batch
//   1. searches the current graph with the SAME traversal kernel as queries (search.cuh; ef =
//      `complexity`, one launch per level that has new members),
//   2. keeps at most M neighbours by the select-neighbours heuristic (HNSW Alg. 4; Vamana
//      RobustPrune with alpha = 1.2, diskann.rs:91) — a 128x128 candidate Gram matrix per point,
//      LDS-tiled, lower triangle kept in LDS,
//   3. proposes itself to each selected neighbour; proposals are radix-sorted by (target, dist)
//      so that every touched list is rewritten by exactly one workgroup: append while there is
//      room (2M on level 0), otherwise the same heuristic over list ∪ proposals.
// No atomics decide content, so a build is reproducible run to run.  The sequential restatement
// lives in oracle/oracle.c (orc_hnsw_build / orc_vamana_build); batched insertion is a different
// schedule, so graphs are compared by invariants + recall, not bit for bit.
#include "common.cuh"
#include "search.cuh"
#include "internal.h"
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <cstring>
#include <numeric>

#define NCMAX 128
#define EXPCAP 256 // expanded nodes recorded per construction search (Vamana)
#define PATHMAX 48 // of which at most this many path nodes join the prune candidates
#define TRI_ELEMS (NCMAX * (NCMAX - 1) / 2)

struct ListView {
    uint32_t *adj0; float *adjd0;  // [n x M0]
    uint32_t *adjU; float *adjdU;  // [n_upper_lists x M]
    const uint32_t *upper_off;
    uint32_t M, M0;
    // Vamana only: per-node PENDING back-edges (ids + distances, [n x P], compact, LEANN_EMPTY padded).  DiskANN lets a list grow to
    // 1.3 R before it re-runs RobustPrune; lists here hold one id per lane of a wave (<= 64), so the slack lives beside the list:
    // a back-edge that finds its target full waits here, invisible to searches, until P of them have gathered; then ONE RobustPrune
    // over list + pending + proposals rewrites the list.  Every touched list being full, the strict rule pruned per proposal:
    // ~600k prunes per 16k-point batch, each gathering 64+ rows (0.4 MB at 1536-d) — 67 % of a 10M x 1536 build.
    uint32_t *pend; float *pendd;
    uint32_t P;
    uint32_t two_stage; // Vamana: occlude_list's two-stage RobustPrune (prune_core)
};
__device__ __forceinline__ void list_ptr(const ListView &lv, uint32_t node, uint32_t level, uint32_t **ids, float **ds,
                                         uint32_t *cap) {
    if (level == 0) {
        *ids = lv.adj0 + (size_t)node * lv.M0;
        *ds = lv.adjd0 + (size_t)node * lv.M0;
        *cap = lv.M0;
    } else {
        size_t o = ((size_t)lv.upper_off[node] + (level - 1)) * lv.M;
        *ids = lv.adjU + o;
        *ds = lv.adjdU + o;
        *cap = lv.M;
    }
}

// ------------------------------------------------------------------------------------------------
// prune_core: candidates (ids c_id, dists-to-p c_d, ascending by (dist, id)) -> up to `limit` kept.
// alpha == 0: HNSW rule (drop c if dist(c, kept) < dist(c, p)); alpha > 0: Vamana rule
// (drop c if alpha * dist(c, kept) <= dist(c, p)).   256 threads.  Returns count in every thread;
// kept candidate positions in s_sel[0..count).
// ------------------------------------------------------------------------------------------------
template <int TS> // per-thread tile TS x TS; the 16 x 16 thread grid covers 16*TS candidates
__device__ __forceinline__ void gram_lower(const float *__restrict__ X, uint32_t ld, const uint32_t *c_id, uint32_t nc,
                                           float *tri, float *stage_) {
    float *stage = static_cast<float *>(__builtin_assume_aligned(stage_, 16));
    constexpr int KC = 32, LDW = NCMAX + 4; // rows of the staging tile stay 16-byte aligned: the 2 x TS operands of a k step are 16-B LDS reads
    // The blocks that touch the lower triangle of the nc x nc matrix — (ty, tx) with tx <= ty < ceil(nc / TS) — are handed to the FIRST
    // threads of the workgroup in triangular order, so the multiply loop runs in ceil(ntiles / 64) waves (1 for the ~70 candidates of a
    // full Vamana list) instead of in every wave that owns a row of a 16 x 16 thread grid (3 of 4 there, a handful of lanes each).
    const int tid = threadIdx.x;
    const int tdim = min(16, (int)((nc + TS - 1) / TS)), ntiles = tdim * (tdim + 1) / 2;
    const bool active = tid < ntiles;
    int ty = 0, tx = 0;
    if (active) {
        ty = (int)((sqrtf(8.f * (float)tid + 1.f) - 1.f) * 0.5f);
        while (ty * (ty + 1) / 2 > tid) ty--;
        while ((ty + 1) * (ty + 2) / 2 <= tid) ty++;
        tx = tid - ty * (ty + 1) / 2;
    }
    float acc[TS][TS];
#pragma unroll
    for (int i = 0; i < TS; i++)
#pragma unroll
        for (int j = 0; j < TS; j++) acc[i][j] = 0.f;
    const int srow = tid >> 1, shalf = tid & 1;
    const float *rowp = (srow < (int)nc) ? X + (size_t)c_id[srow] * ld : nullptr;
    for (uint32_t k0 = 0; k0 < ld; k0 += KC) {
        float4 v[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            uint32_t j = k0 + shalf * 16 + e * 4;
            v[e] = (rowp && j < ld) ? *reinterpret_cast<const float4 *>(rowp + j) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads(); // previous chunk fully consumed
        if (srow < 16 * TS) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                int kk = shalf * 16 + e * 4;
                stage[(kk + 0) * LDW + srow] = v[e].x;
                stage[(kk + 1) * LDW + srow] = v[e].y;
                stage[(kk + 2) * LDW + srow] = v[e].z;
                stage[(kk + 3) * LDW + srow] = v[e].w;
            }
        }
        __syncthreads();
        if (active) {
#pragma unroll 4
            for (int kk = 0; kk < KC; kk++) {
                float a[TS], b[TS];
#pragma unroll
                for (int i = 0; i < TS; i++) a[i] = stage[kk * LDW + ty * TS + i];
#pragma unroll
                for (int j = 0; j < TS; j++) b[j] = stage[kk * LDW + tx * TS + j];
#pragma unroll
                for (int i = 0; i < TS; i++)
#pragma unroll
                    for (int j = 0; j < TS; j++) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
            }
        }
    }
    __syncthreads(); // stage dead; tri may be written
    if (active) {
#pragma unroll
        for (int i = 0; i < TS; i++)
#pragma unroll
            for (int j = 0; j < TS; j++) {
                int r = ty * TS + i, c = tx * TS + j;
                if (c < r && r < (int)nc) tri[r * (r - 1) / 2 + c] = 1.0f - acc[i][j];
            }
    }
    __syncthreads();
}

__device__ uint32_t prune_core(const float *__restrict__ X, uint32_t ld, const uint32_t *c_id, const float *c_d,
                               uint32_t nc, uint32_t limit, float alpha, float *tri /* TRI_ELEMS floats, LDS */,
                               uint32_t *s_sel /* [64] LDS */, uint32_t *s_cnt /* LDS */, bool stage1) {
    float *stage = static_cast<float *>(__builtin_assume_aligned(tri, 16)); // [KC][LDW] aliased: dead before tri is written
    const int tid = threadIdx.x;
    if (nc <= 32) gram_lower<2>(X, ld, c_id, nc, tri, stage);       // work ~ nc^2: small lists use small tiles
    else if (nc <= 64) gram_lower<4>(X, ld, c_id, nc, tri, stage);
    else gram_lower<8>(X, ld, c_id, nc, tri, stage);
    __shared__ uint8_t s_taken[NCMAX];
    if (tid < 64) { // wave 0: sequential walk over candidates, lanes = kept slots
        uint32_t ns = 0;
        int my = -1;
        // Vamana, two-stage form (DiskANN's occlude_list): the FIRST walk over the whole pool keeps a candidate only if no kept
        // one is at least as close to it as the point itself (alpha = 1: the diverse core, which reaches the far end of the pool before
        // the list is full); only the slots still free after it are filled by the relaxed rule alpha * d(c, kept) <= d(c, p) -> drop.
        // With the one-stage rule of the paper (Alg. 2) at alpha = 1.2 almost nothing is occluded on data of high intrinsic dimension, a
        // list is simply the R nearest of the pool, and at R = 32 the graph stops being navigable as the corpus grows (recall@10 at
        // L = 128, 256-d: 0.976 at 1M, 0.905 at 5M, 0.78 at 10M; scripts/exp/vamana_scale.py, profiles/r03_vamana_scale.md).
        const bool two_stage = alpha > 1.0f && stage1;
        if (two_stage)
            for (uint32_t i = tid; i < nc; i += 64) s_taken[i] = 0;
        for (uint32_t i = 0; i < nc; i++) {
            const float di = c_d[i];
            bool bad = false;
            if (tid < (int)ns) {
                float gdist = tri[i * (i - 1) / 2 + my];
                bad = (alpha == 0.f) ? (gdist < di) : ((two_stage ? 1.0f : alpha) * gdist <= di);
            }
            if (!__any(bad)) {
                if (tid == (int)ns) my = (int)i;
                if (two_stage && tid == 0) s_taken[i] = 1;
                ns++;
                if (ns == limit) break;
            }
        }
        if (two_stage && ns < limit) {
            for (uint32_t i = 0; i < nc; i++) {
                if (s_taken[i]) continue; // (uniform: every lane reads the same byte)
                const float di = c_d[i];
                bool bad = false;
                if (tid < (int)ns) {
                    const uint32_t hi = max(i, (uint32_t)my), lo = min(i, (uint32_t)my);
                    bad = alpha * tri[hi * (hi - 1) / 2 + lo] <= di;
                }
                if (!__any(bad)) {
                    if (tid == (int)ns) my = (int)i;
                    ns++;
                    if (ns == limit) break;
                }
            }
        }
        if (tid < (int)ns) s_sel[tid] = (uint32_t)my;
        if (tid == 0) *s_cnt = ns;
    }
    __syncthreads();
    return *s_cnt;
}

// ------------------------------------------------------------------------------------------------
// select_kernel: one workgroup per new point of the batch (at one level).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) select_kernel(const float *__restrict__ X, uint32_t ld, ListView lv,
                                                     const uint32_t *__restrict__ q_rows, uint32_t nq, uint32_t level,
                                                     const uint64_t *__restrict__ cand_keys, const float *__restrict__ cand_d,
                                                     const uint32_t *__restrict__ cand_cnt, uint32_t efc, uint32_t msel,
                                                     float alpha, uint64_t *__restrict__ prop_key, uint32_t *__restrict__ prop_src,
                                                     const uint64_t *__restrict__ exp_keys, const uint32_t *__restrict__ exp_cnt,
                                                     uint32_t exp_cap) {
    __shared__ __attribute__((aligned(16))) float tri[TRI_ELEMS];
    __shared__ uint32_t c_id[NCMAX];
    __shared__ float c_d[NCMAX];
    __shared__ uint32_t s_sel[64];
    __shared__ uint32_t s_cnt;
    __shared__ uint64_t pkey[EXPCAP];
    const uint32_t qi = blockIdx.x;
    if (qi >= nq) return;
    const uint32_t q = q_rows[qi];
    uint32_t nc = min(min(cand_cnt[qi], efc), (uint32_t)NCMAX);
    for (uint32_t i = threadIdx.x; i < NCMAX; i += 256) {
        c_id[i] = i < nc ? (uint32_t)cand_keys[(size_t)qi * efc + i] : 0u;
        c_d[i] = i < nc ? cand_d[(size_t)qi * efc + i] : 0.f;
    }
    if (threadIdx.x == 0) s_cnt = NCMAX;
    __syncthreads();
    // a point that is already linked (second Vamana pass) finds itself: drop it from its own candidates
    for (uint32_t i = threadIdx.x; i < nc; i += 256)
        if (c_id[i] == q) s_cnt = i;
    __syncthreads();
    if (s_cnt < nc) {
        const uint32_t self = s_cnt;
        uint32_t mv_id[NCMAX / 256 + 1];
        float mv_d[NCMAX / 256 + 1];
        int t = 0;
        for (uint32_t i = self + threadIdx.x; i + 1 < nc; i += 256, t++) { mv_id[t] = c_id[i + 1]; mv_d[t] = c_d[i + 1]; }
        __syncthreads();
        t = 0;
        for (uint32_t i = self + threadIdx.x; i + 1 < nc; i += 256, t++) { c_id[i] = mv_id[t]; c_d[i] = mv_d[t]; }
        nc--;
        __syncthreads();
    }
    if (exp_keys && nc > 0) {
        // Vamana: prune over the visited set V = final beam  ∪  nodes expanded on the way from the medoid
        // (DiskANN Alg. 2/3).  Path nodes are the expanded entries farther than the beam's worst entry; they
        // supply the long-range edges that keep a single-level graph navigable.
        const uint64_t wmax = ((uint64_t)f32_orderable(c_d[nc - 1]) << 32) | c_id[nc - 1];
        const uint32_t np = min(exp_cnt[qi], min(exp_cap, (uint32_t)EXPCAP));
        for (uint32_t i = threadIdx.x; i < EXPCAP; i += 256) {
            uint64_t key = i < np ? exp_keys[(size_t)qi * exp_cap + i] : ~0ull;
            pkey[i] = (key != ~0ull && key > wmax) ? key : ~0ull;
        }
        for (int size = 2; size <= EXPCAP; size <<= 1)
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                __syncthreads();
                if (threadIdx.x < EXPCAP / 2) {
                    int i = threadIdx.x;
                    int lo = 2 * i - (i & (stride - 1)), hi = lo + stride;
                    bool up = ((lo & size) == 0);
                    uint64_t a = pkey[lo], b2 = pkey[hi];
                    if ((a > b2) == up) { pkey[lo] = b2; pkey[hi] = a; }
                }
            }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t n_path = 0;
            while (n_path < EXPCAP && pkey[n_path] != ~0ull) n_path++;
            s_cnt = n_path;
        }
        __syncthreads();
        const uint32_t n_path = s_cnt;
        const uint32_t take = min(n_path, (uint32_t)PATHMAX);
        const uint32_t keep = min(nc, (uint32_t)NCMAX - take);
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < take; t += 256) { // even subsample keeps near and far path nodes
            const uint64_t key = pkey[(uint64_t)t * n_path / take];
            c_id[keep + t] = (uint32_t)key;
            c_d[keep + t] = orderable_f32((uint32_t)(key >> 32));
        }
        nc = keep + take;
        __syncthreads();
    }
    uint32_t ns = prune_core(X, ld, c_id, c_d, nc, msel, alpha, tri, s_sel, &s_cnt, lv.two_stage != 0);
    uint32_t *ids; float *ds; uint32_t cap;
    list_ptr(lv, q, level, &ids, &ds, &cap);
    for (uint32_t j = threadIdx.x; j < msel; j += 256) {
        uint64_t pk = ~0ull;
        if (j < ns) {
            uint32_t c = s_sel[j];
            ids[j] = c_id[c];
            ds[j] = c_d[c];
            pk = ((uint64_t)c_id[c] << 32) | f32_orderable(c_d[c]);
        }
        prop_key[(size_t)qi * msel + j] = pk; // (target, dist) ; padding sorts to the end
        prop_src[(size_t)qi * msel + j] = q;
    }
    for (uint32_t j = ns + threadIdx.x; j < cap; j += 256) ids[j] = LEANN_EMPTY; // (a re-linked point: nothing of its old list stays behind the new one)
}

// heads of equal-target runs in the sorted proposal array
__global__ void segment_heads_kernel(const uint64_t *__restrict__ keys, uint32_t num, uint32_t *__restrict__ seg_start,
                                     uint32_t *__restrict__ nseg) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= num) return;
    uint32_t t = (uint32_t)(keys[i] >> 32);
    if (t == LEANN_EMPTY) return;
    if (i == 0 || (uint32_t)(keys[i - 1] >> 32) != t) seg_start[atomicAdd(nseg, 1u)] = i;
}

// ------------------------------------------------------------------------------------------------
// reverse_merge_kernel: one workgroup per touched list (target node at `level`).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) reverse_merge_kernel(const float *__restrict__ X, uint32_t ld, ListView lv,
                                                            uint32_t level, const uint64_t *__restrict__ keys,
                                                            const uint32_t *__restrict__ srcs, uint32_t num,
                                                            const uint32_t *__restrict__ seg_start,
                                                            const uint32_t *__restrict__ nseg_p, float alpha, uint32_t flush_nodes) {
    // flush_nodes != 0: final pass over nodes [0, flush_nodes) — fold whatever is still pending into its list (no proposals)
    __shared__ __attribute__((aligned(16))) float tri[TRI_ELEMS];
    __shared__ uint64_t skey[NCMAX];
    __shared__ uint32_t c_id[NCMAX];
    __shared__ float c_d[NCMAX];
    __shared__ uint32_t s_sel[64];
    __shared__ uint32_t s_cnt, s_k, s_len, s_pl;
    const uint32_t nseg = flush_nodes ? flush_nodes : *nseg_p;
    const uint32_t P = level == 0 ? lv.P : 0u;
    for (uint32_t seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
        const uint32_t start = flush_nodes ? 0u : seg_start[seg];
        const uint32_t t = flush_nodes ? seg : (uint32_t)(keys[start] >> 32);
        uint32_t *ids; float *ds; uint32_t cap;
        list_ptr(lv, t, level, &ids, &ds, &cap);
        uint32_t *pid = P ? lv.pend + (size_t)t * P : nullptr;
        float *pdd = P ? lv.pendd + (size_t)t * P : nullptr;
        if (threadIdx.x == 0) { s_k = 0; s_len = 0; s_pl = 0; }
        __syncthreads();
        // proposals of this run (sorted by dist): count up to NCMAX; existing list / pending lengths
        if (!flush_nodes && threadIdx.x < NCMAX) {
            uint32_t i = start + threadIdx.x;
            if (i < num && (uint32_t)(keys[i] >> 32) == t) atomicAdd(&s_k, 1u); // run is contiguous
        }
        if (threadIdx.x < cap && ids[threadIdx.x] != LEANN_EMPTY) atomicAdd(&s_len, 1u); // lists are compact
        if (threadIdx.x >= 64 && threadIdx.x < 64 + P && pid[threadIdx.x - 64] != LEANN_EMPTY) atomicAdd(&s_pl, 1u);
        __syncthreads();
        const uint32_t len = s_len, pl = s_pl;
        uint32_t k = s_k;
        if (flush_nodes && pl == 0) { __syncthreads(); continue; }
        const uint32_t room = cap - len;
        if (!flush_nodes && k <= room + (P - pl)) {
            // room in the list (visible at once, closest proposals first), then in the pending area: plain appends in (dist, src) order
            for (uint32_t j = threadIdx.x; j < k; j += 256) {
                const uint32_t src = srcs[start + j];
                const float dj = orderable_f32((uint32_t)keys[start + j]);
                if (j < room) { ids[len + j] = src; ds[len + j] = dj; }
                else { pid[pl + j - room] = src; pdd[pl + j - room] = dj; }
            }
            __syncthreads();
            continue;
        }
        if (len + pl + k > NCMAX) k = NCMAX - len - pl; // keep the closest proposals only
        const uint32_t nc = len + pl + k;
        for (uint32_t i = threadIdx.x; i < NCMAX; i += 256) {
            uint64_t key = ~0ull;
            if (i < len) key = ((uint64_t)f32_orderable(ds[i]) << 32) | ids[i];
            else if (i < len + pl) key = ((uint64_t)f32_orderable(pdd[i - len]) << 32) | pid[i - len];
            else if (i < nc) key = ((uint64_t)(uint32_t)keys[start + i - len - pl] << 32) | srcs[start + i - len - pl];
            skey[i] = key;
        }
        // bitonic sort of NCMAX keys by (dist, id)
        for (int size = 2; size <= NCMAX; size <<= 1)
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                __syncthreads();
                if (threadIdx.x < NCMAX / 2) {
                    int i = threadIdx.x;
                    int lo = 2 * i - (i & (stride - 1)), hi = lo + stride;
                    bool up = ((lo & size) == 0);
                    uint64_t a = skey[lo], b = skey[hi];
                    if ((a > b) == up) { skey[lo] = b; skey[hi] = a; }
                }
            }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < NCMAX; i += 256) {
            c_id[i] = i < nc ? (uint32_t)skey[i] : 0u;
            c_d[i] = i < nc ? orderable_f32((uint32_t)(skey[i] >> 32)) : 0.f;
        }
        __syncthreads();
        uint32_t ns = prune_core(X, ld, c_id, c_d, nc, cap, alpha, tri, s_sel, &s_cnt, lv.two_stage != 0);
        for (uint32_t j = threadIdx.x; j < cap; j += 256) {
            if (j < ns) {
                uint32_t c = s_sel[j];
                ids[j] = c_id[c];
                ds[j] = c_d[c];
            } else {
                ids[j] = LEANN_EMPTY;
                ds[j] = 0.f;
            }
        }
        for (uint32_t j = threadIdx.x; j < P; j += 256) { pid[j] = LEANN_EMPTY; pdd[j] = 0.f; }
        __syncthreads();
    }
}

__global__ void fill_u32_kernel(uint32_t *p, size_t n, uint32_t v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void col_partial_kernel(const float *__restrict__ X, size_t n, uint32_t d, uint32_t ld, uint32_t S,
                                   double *__restrict__ part /* [S x ld] */) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x, sl = blockIdx.y;
    if (j >= ld) return;
    double s = 0.0;
    if (j < d)
        for (size_t i = sl; i < n; i += S) s += X[i * ld + j];
    part[(size_t)sl * ld + j] = s;
}
__global__ void col_mean_kernel(const double *__restrict__ part, size_t n, uint32_t ld, uint32_t S, float *__restrict__ mean) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ld) return;
    double s = 0.0;
    for (uint32_t sl = 0; sl < S; sl++) s += part[(size_t)sl * ld + j];
    mean[j] = (float)(s / (double)n);
}

// Stored link distances of an existing graph (the index file does not carry them): one wave per node walks its level-0
// list and its upper lists and recomputes dist(node, neighbour) with the canonical wave dot — bit-identical to the value
// the construction search produced when the link was made (fmaf is symmetric in its factors), so an append continues
// exactly from the state a one-shot build would have had.
__global__ void __launch_bounds__(256) link_dist_kernel(const float *__restrict__ X, uint32_t ld, ListView lv,
                                                        const uint8_t *__restrict__ levels, uint32_t n_nodes) {
    const uint32_t node = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (node >= n_nodes) return;
    const float *xi = X + (size_t)node * ld;
    const int T = (int)((ld + 255) / 256);
    const uint32_t top = levels[node];
    for (uint32_t level = 0; level <= top; level++) {
        uint32_t *ids; float *ds; uint32_t cap;
        list_ptr(lv, node, level, &ids, &ds, &cap);
        for (uint32_t j = 0; j < cap; j++) {
            const uint32_t e = ids[j];
            if (e == LEANN_EMPTY) break; // lists are compact
            const float *xe = X + (size_t)e * ld;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int t = 0; t < T; t++) fma4(acc, row_load4(xi, ld, t, lane), row_load4(xe, ld, t, lane));
            const float d = 1.0f - wave_tree_sum(lane4_sum(acc));
            if (lane == 0) ds[j] = d;
        }
    }
}

// ------------------------------------------------------------------------------------------------
struct Builder {
    leann_backend *h;
    ListView lv{};
    float *adjd0 = nullptr, *adjdU = nullptr;
    uint32_t *pend = nullptr; float *pendd = nullptr; // Vamana: pending back-edges (ListView)
    std::vector<uint8_t> levels;     // host copy
    std::vector<uint32_t> upper_off; // host copy
    uint32_t *d_order = nullptr;     // insertion order (device copy)
    std::vector<uint32_t> order;     // ... host copy: order[j] = position inserted j-th
    std::vector<uint32_t> h_order_first; // only used for Vamana (medoid first)
    // per-batch scratch
    size_t bmax = 0;
    uint64_t *cand_keys = nullptr; float *cand_d = nullptr; uint32_t *cand_cnt = nullptr;
    uint64_t *candU_keys = nullptr; float *candU_d = nullptr; uint32_t *candU_cnt = nullptr; uint32_t *d_rowsU = nullptr;
    uint64_t *prop_key = nullptr, *prop_key2 = nullptr; uint32_t *prop_src = nullptr, *prop_src2 = nullptr;
    uint32_t *seg_start = nullptr, *nseg = nullptr;
    uint64_t *exp_keys = nullptr; uint32_t *exp_cnt = nullptr; // Vamana only
    void *cub_tmp = nullptr; size_t cub_bytes = 0;
    hipStream_t st = nullptr;
    size_t batch_fraction = 8; // LEANN_BUILD_BATCH_FRACTION, read once per build
    float alpha_now = 1.2f;    // Vamana: RobustPrune's alpha of the pass under way
    Builder() = default;
    Builder(const Builder &) = delete;
    ~Builder() { // every exit of build_on_device, including the error returns, releases the scratch and the stream
        if (st) (void)hipStreamSynchronize(st);
        void *ps[] = {cand_keys, cand_d, cand_cnt, candU_keys, candU_d, candU_cnt, d_rowsU, prop_key, prop_key2,
                      prop_src, prop_src2, seg_start, nseg, cub_tmp, d_order, adjd0, adjdU, exp_keys, exp_cnt, pend, pendd};
        for (void *p : ps) (void)hipFree(p);
        if (st) (void)hipStreamDestroy(st);
    }
};
// construction knobs are read once per build (a build takes seconds to minutes; the search path reads no environment at all)
static int env_int(const char *name, int dflt, int lo, int hi) {
    const char *e = getenv(name);
    if (!e || !*e) return dflt;
    const int v = atoi(e);
    return v >= lo && v <= hi ? v : dflt;
}
struct DevTmp { // small scoped device allocation
    void *p = nullptr;
    ~DevTmp() { (void)hipFree(p); }
};

#define BCHECK(expr)                                                                           \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            leann_set_error("build: %s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return LEANN_ERR_DEVICE;                                                           \
        }                                                                                      \
    } while (0)

static int link_level(Builder &b, const uint32_t *d_rows, uint32_t nq, uint32_t level, const uint64_t *ck, const float *cd,
                      const uint32_t *cc, const uint64_t *ek = nullptr, const uint32_t *ec = nullptr) {
    leann_backend *h = b.h;
    const uint32_t efc = h->efc, msel = h->g.M; // M new links per point on every level (Malkov Alg. 1)
    const float alpha = h->kind == LEANN_BACKEND_DISKANN ? b.alpha_now : 0.f;
    hipLaunchKernelGGL(select_kernel, dim3(nq), dim3(256), 0, b.st, h->g.X, h->g.ld, b.lv, d_rows, nq, level, ck, cd, cc,
                       efc, msel, alpha, b.prop_key, b.prop_src, ek, ec, (uint32_t)EXPCAP);
    const uint32_t num = nq * msel;
    size_t tmp = b.cub_bytes;
    BCHECK(hipcub::DeviceRadixSort::SortPairs(b.cub_tmp, tmp, b.prop_key, b.prop_key2, b.prop_src, b.prop_src2, (int)num, 0, 64, b.st));
    BCHECK(hipMemsetAsync(b.nseg, 0, 4, b.st));
    hipLaunchKernelGGL(segment_heads_kernel, dim3((num + 255) / 256), dim3(256), 0, b.st, b.prop_key2, num, b.seg_start, b.nseg);
    uint32_t grid = std::min<uint32_t>(num, 256 * 16);
    hipLaunchKernelGGL(reverse_merge_kernel, dim3(grid), dim3(256), 0, b.st, h->g.X, h->g.ld, b.lv, level, b.prop_key2,
                       b.prop_src2, num, b.seg_start, b.nseg, alpha, 0u);
    BCHECK(hipGetLastError());
    return LEANN_OK;
}

static int builder_alloc_scratch(Builder &b, size_t bmax) {
    leann_backend *h = b.h;
    const size_t efc = h->efc, msel = h->g.M;
    b.bmax = bmax;
    const size_t bu = bmax / 16 + 64; // upper-level members per batch (expected bmax/31)
    BCHECK(hipMalloc((void **)&b.cand_keys, bmax * efc * 8));
    BCHECK(hipMalloc((void **)&b.cand_d, bmax * efc * 4));
    BCHECK(hipMalloc((void **)&b.cand_cnt, bmax * 4));
    BCHECK(hipMalloc((void **)&b.candU_keys, bu * efc * 8 * 16));
    BCHECK(hipMalloc((void **)&b.candU_d, bu * efc * 4 * 16));
    BCHECK(hipMalloc((void **)&b.candU_cnt, bu * 4 * 16));
    BCHECK(hipMalloc((void **)&b.d_rowsU, bu * 4 * 16));
    BCHECK(hipMalloc((void **)&b.prop_key, bmax * msel * 8));
    BCHECK(hipMalloc((void **)&b.prop_key2, bmax * msel * 8));
    BCHECK(hipMalloc((void **)&b.prop_src, bmax * msel * 4));
    BCHECK(hipMalloc((void **)&b.prop_src2, bmax * msel * 4));
    BCHECK(hipMalloc((void **)&b.seg_start, bmax * msel * 4));
    BCHECK(hipMalloc((void **)&b.nseg, 16));
    if (h->kind == LEANN_BACKEND_DISKANN) {
        BCHECK(hipMalloc((void **)&b.exp_keys, bmax * EXPCAP * 8));
        BCHECK(hipMalloc((void **)&b.exp_cnt, bmax * 4));
    }
    size_t tmp = 0;
    BCHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, b.prop_key, b.prop_key2, b.prop_src, b.prop_src2,
                                              (int)(bmax * msel), 0, 64, b.st));
    b.cub_bytes = tmp;
    BCHECK(hipMalloc(&b.cub_tmp, tmp));
    return LEANN_OK;
}

// Insert order[s0 .. n) into the graph that already holds order[0 .. s0).
static int builder_insert_range(Builder &b, const std::vector<uint32_t> &order_host_upper /* unused */, size_t s0, size_t n,
                                bool refine = false) {
    (void)order_host_upper;
    leann_backend *h = b.h;
    const uint32_t efc = h->efc;
    const bool hnsw = h->kind == LEANN_BACKEND_HNSW || h->nav_levels; // (leveled: entry layers above the base graph)
    size_t s = s0;
    const size_t bu_cap = (b.bmax / 16 + 64) * 16;
    while (s < n) {
        // A batch's points do not see each other, so a batch must stay a small fraction of the graph it is inserted into: at most
        // 1/8 (LEANN_BUILD_BATCH_FRACTION) of the points already linked.  With batches as large as the graph itself — the first
        // rule here — two thirds of a 50k-row index went in without seeing their batch mates and recall@10 at ef = 32 fell 1.6
        // points below the sequential builder's (tests/test_gpu_builder_quality.py); from ~130k rows on the cap is bmax anyway.
        const size_t frac = b.batch_fraction;
        size_t B = std::min<size_t>(std::min<size_t>(b.bmax, refine ? b.bmax : std::max<size_t>(1, s / frac)), n - s);
        const uint32_t Lmax = h->g.max_level;
        // ---- phase 1: searches (graph does not contain any point of the batch yet) ----------------
        struct LevelJob { uint32_t level, nq; size_t off; };
        std::vector<LevelJob> jobs;
        size_t offU = 0;
        if (hnsw && Lmax > 0) {
            for (uint32_t l = Lmax; l >= 1; --l) {
                std::vector<uint32_t> rows;
                for (size_t i = s; i < s + B; i++)
                    if (b.levels[b.order[i]] >= l) rows.push_back(b.order[i]);
                if (rows.empty()) continue;
                if (offU + rows.size() > bu_cap) {
                    leann_set_error("build: upper-level scratch exhausted");
                    return LEANN_ERR_DEVICE;
                }
                BCHECK(hipMemcpyAsync(b.d_rowsU + offU, rows.data(), rows.size() * 4, hipMemcpyHostToDevice, b.st));
                BCHECK(hipStreamSynchronize(b.st)); // rows is a stack temporary
                SearchArgs a{};
                a.q_rows = b.d_rowsU + offU;
                a.nq = (uint32_t)rows.size();
                a.k = efc;
                a.ef = efc;
                a.target_level = l;
                a.out_keys = b.candU_keys + offU * efc;
                a.out_dists = b.candU_d + offU * efc;
                a.out_counts = b.candU_cnt + offU;
                int rc = leann_internal_launch_search(h, a, b.st);
                if (rc) return rc;
                jobs.push_back({l, (uint32_t)rows.size(), offU});
                offU += rows.size();
            }
        }
        {
            SearchArgs a{};
            a.q_rows = b.d_order + s;
            a.nq = (uint32_t)B;
            a.k = efc;
            a.ef = efc;
            a.target_level = 0;
            a.out_keys = b.cand_keys;
            a.out_dists = b.cand_d;
            a.out_counts = b.cand_cnt;
            a.out_expanded = b.exp_keys; // null for HNSW
            a.out_nexp = b.exp_cnt;
            a.exp_cap = EXPCAP;
            int rc = leann_internal_launch_search(h, a, b.st);
            if (rc) return rc;
        }
        // ---- phase 2: select + link, level by level ---------------------------------------------
        for (auto &j : jobs) {
            int rc = link_level(b, b.d_rowsU + j.off, j.nq, j.level, b.candU_keys + j.off * efc, b.candU_d + j.off * efc,
                                b.candU_cnt + j.off);
            if (rc) return rc;
        }
        int rc = link_level(b, b.d_order + s, (uint32_t)B, 0, b.cand_keys, b.cand_d, b.cand_cnt, b.exp_keys, b.exp_cnt);
        if (rc) return rc;
        // ---- entry point / top level (sequential semantics of hnsw.rs:128-130 within the batch) --
        if (hnsw)
            for (size_t i = s; i < s + B; i++)
                if (b.levels[b.order[i]] > h->g.max_level) { h->g.max_level = b.levels[b.order[i]]; h->g.entry = b.order[i]; }
        s += B;
    }
    BCHECK(hipStreamSynchronize(b.st));
    return LEANN_OK;
}

// Insertion order.  A batch is inserted against the graph of the batches before it (its points do not see each other), so every
// batch must be a representative sample of the rows still to come: inserting in storage order breaks on data stored topic by topic
// (a 10M-row corpus laid out cluster-major and built in id order: recall@10 0.20 at ef = 56, against 0.957 for the same rows stored
// in hashed order).  The rows [lo, n) are therefore inserted in a fixed pseudo-random permutation (positions sorted by a hash of
// the position): 0.957 for both layouts.  (A golden-ratio stride permutation was tried first: fine on the cluster-major layout,
// 0.947 on the hashed one.)  Rows [0, lo) (an existing index) keep their places.  `pin_first`: that position is inserted first
// (Vamana medoid).  Returns order[lo].
static uint32_t insertion_order(std::vector<uint32_t> &order, size_t lo, size_t n, const uint32_t *pin_first = nullptr) {
    order.resize(std::max<size_t>(n, 1));
    for (size_t i = 0; i < lo; i++) order[i] = (uint32_t)i;
    const uint64_t m = n - lo;
    if (m == 0) return 0;
    size_t w = lo;
    if (pin_first) order[w++] = *pin_first;
    // pseudo-random permutation: positions sorted by a hash of the position (ties impossible: the position is part of the key)
    std::vector<uint64_t> keyed(m);
    for (uint64_t j = 0; j < m; j++) keyed[j] = (mix64((lo + j) ^ 0x4F52444552ull) & 0xFFFFFFFF00000000ull) | (uint32_t)(lo + j);
    std::sort(keyed.begin(), keyed.end());
    for (uint64_t j = 0; j < m; j++) {
        const uint32_t pos = (uint32_t)keyed[j];
        if (pin_first && pos == *pin_first) continue;
        order[w++] = pos;
    }
    return order[lo];
}

static int build_on_device(leann_backend *h, size_t n_existing, size_t bmax_hint) {
    // h->g.X / n / d / ld / M / M0 / kind / efc / alpha are set; adjacency arrays allocated & initialised
    // for nodes < n_existing (append) or empty.
    Builder b;
    b.h = h;
    b.batch_fraction = (size_t)env_int("LEANN_BUILD_BATCH_FRACTION", 8, 1, 1 << 20);
    const size_t n = h->g.n;
    BCHECK(hipStreamCreateWithFlags(&b.st, hipStreamNonBlocking));
    b.lv.adj0 = const_cast<uint32_t *>(h->g.adj0);
    b.lv.adjU = const_cast<uint32_t *>(h->g.adjU);
    b.lv.upper_off = h->g.upper_off;
    b.lv.M = h->g.M;
    b.lv.M0 = h->g.M0;
    const size_t nu = std::max<size_t>(h->n_upper_lists, 1);
    BCHECK(hipMalloc((void **)&b.adjd0, std::max<size_t>(n, 1) * h->g.M0 * 4));
    BCHECK(hipMalloc((void **)&b.adjdU, nu * h->g.M * 4));
    BCHECK(hipMemset(b.adjd0, 0, std::max<size_t>(n, 1) * h->g.M0 * 4));
    BCHECK(hipMemset(b.adjdU, 0, nu * h->g.M * 4));
    b.lv.adjd0 = b.adjd0;
    b.lv.adjdU = b.adjdU;
    if (h->kind == LEANN_BACKEND_DISKANN) { // slack of DiskANN's 1.3 R, kept beside the 64-slot lists (see ListView)
        // 10M x 1536, R = 64, recall@10 at L = 72 over 2 000 queries (scripts/exp/vamana_slack.sh): strict (0 pending) 185.8 s / 0.9600;
        // 4 pending 82.2 s / 0.9596; 8: 65.3 s / 0.9531; 16: 56.1 s / 0.9547.  Back-edges that wait are invisible to the construction
        // searches of later points, which costs about half a point of recall from 8 on; 4 keeps the strict rule's recall at 2.3x its speed.
        b.lv.two_stage = (uint32_t)env_int("LEANN_VAMANA_TWO_STAGE", 1, 0, 1);
        const uint32_t slack = (uint32_t)env_int("LEANN_VAMANA_PENDING", 4, 0, 32);
        b.lv.P = slack;
        if (slack) {
            BCHECK(hipMalloc((void **)&b.pend, std::max<size_t>(n, 1) * slack * 4));
            BCHECK(hipMalloc((void **)&b.pendd, std::max<size_t>(n, 1) * slack * 4));
            BCHECK(hipMemset(b.pend, 0xFF, std::max<size_t>(n, 1) * slack * 4));
            BCHECK(hipMemset(b.pendd, 0, std::max<size_t>(n, 1) * slack * 4));
            b.lv.pend = b.pend;
            b.lv.pendd = b.pendd;
        }
    }
    b.levels.resize(std::max<size_t>(n, 1));
    BCHECK(hipMemcpy(b.levels.data(), h->d_levels, n, hipMemcpyDeviceToHost));
    BCHECK(hipMalloc((void **)&b.d_order, std::max<size_t>(n, 1) * 4));
    int rc = LEANN_OK;
    size_t s0 = n_existing;
    if (n_existing == 0 && n > 0) {
        uint32_t first = 0;
        if (h->kind == LEANN_BACKEND_DISKANN && !h->nav_levels) {
            // medoid: closest row to the mean direction, by the index metric (ties -> lower id)
            DevTmp t_mean, t_mk, t_ms, t_mc, t_part;
            const uint32_t S = 1024;
            BCHECK(hipMalloc(&t_mean.p, h->g.ld * 4));
            BCHECK(hipMalloc(&t_mk.p, 8)); BCHECK(hipMalloc(&t_ms.p, 4)); BCHECK(hipMalloc(&t_mc.p, 4));
            BCHECK(hipMalloc(&t_part.p, (size_t)S * h->g.ld * 8));
            float *mean = (float *)t_mean.p; uint64_t *mk = (uint64_t *)t_mk.p; float *ms = (float *)t_ms.p; uint32_t *mc = (uint32_t *)t_mc.p;
            double *part = (double *)t_part.p;
            hipLaunchKernelGGL(col_partial_kernel, dim3((h->g.ld + 63) / 64, S), dim3(64), 0, b.st, h->g.X, n, h->g.d, h->g.ld, S, part);
            hipLaunchKernelGGL(col_mean_kernel, dim3((h->g.ld + 63) / 64), dim3(64), 0, b.st, part, n, h->g.ld, S, mean);
            BCHECK(hipStreamSynchronize(b.st));
            rc = leann_scan_topk_device(h->g.X, n, h->g.d, h->g.ld, mean, 1, 1, nullptr, 0, mk, ms, mc, b.st);
            if (rc) return rc;
            uint64_t key = 0;
            BCHECK(hipMemcpyAsync(&key, mk, 8, hipMemcpyDeviceToHost, b.st));
            BCHECK(hipStreamSynchronize(b.st));
            first = (uint32_t)key;
        }
        const bool leveled = h->kind == LEANN_BACKEND_HNSW || h->nav_levels;
        if (leveled) first = insertion_order(b.order, 0, n);
        else insertion_order(b.order, 0, n, &first);
        h->g.entry = first;
        h->g.max_level = leveled ? b.levels[first] : 0;
        s0 = 1;
    } else {
        insertion_order(b.order, n_existing, n);
        // append: the existing links' stored distances (read by reverse_merge_kernel when a list is pruned)
        hipLaunchKernelGGL(link_dist_kernel, dim3((unsigned)((n_existing + 3) / 4)), dim3(256), 0, b.st, h->g.X, h->g.ld, b.lv, h->d_levels,
                           (uint32_t)n_existing);
        BCHECK(hipGetLastError());
    }
    BCHECK(hipMemcpyAsync(b.d_order, b.order.data(), n * 4, hipMemcpyHostToDevice, b.st));
    size_t bmax = bmax_hint ? bmax_hint : 16384;
    rc = builder_alloc_scratch(b, bmax);
    const int passes = env_int("LEANN_VAMANA_PASSES", 1, 1, 3);
    b.alpha_now = h->alpha;
    if (passes > 1 && h->kind == LEANN_BACKEND_DISKANN) { // experiment knob: alpha of every pass but the last (DiskANN: 1.0), in 1/100
        b.alpha_now = (float)env_int("LEANN_VAMANA_ALPHA1_PCT", (int)(h->alpha * 100.f + 0.5f), 100, 400) / 100.f;
    }
    if (rc == LEANN_OK) rc = builder_insert_range(b, {}, s0, n);
    // DiskANN builds in two passes over the points (the second one re-links every point against the finished graph).  Optional here
    // (LEANN_VAMANA_PASSES=2), off by default — 10M x 1536: at R = 64 it doubles the build (82 -> 173 s) and buys nothing (recall@10 0.95
    // needs L = 76 instead of 72); at R = 32, where one pass is not enough for 10M clustered rows (recall 0.60 at L = 128), it lifts the
    // recall to 0.81 — still not a usable index, which is why the 10M benchmarks use R = 64.
    for (int p = 1; rc == LEANN_OK && p < passes && h->kind == LEANN_BACKEND_DISKANN && n_existing == 0 && n > 1; p++) {
        if (p == passes - 1) b.alpha_now = h->alpha;
        rc = builder_insert_range(b, {}, 0, n, true);
    }
    b.alpha_now = h->alpha;
    if (rc == LEANN_OK && b.lv.P && n) { // fold what is still pending into the lists (DiskANN's final trim)
        const float alpha = h->alpha;
        hipLaunchKernelGGL(reverse_merge_kernel, dim3(256 * 16), dim3(256), 0, b.st, h->g.X, h->g.ld, b.lv, 0u, (const uint64_t *)nullptr,
                           (const uint32_t *)nullptr, 0u, (const uint32_t *)nullptr, (const uint32_t *)nullptr, alpha, (uint32_t)n);
        BCHECK(hipGetLastError());
        BCHECK(hipStreamSynchronize(b.st));
    }
    return rc; // ~Builder: stream synchronised, scratch freed
}

// levels / upper_off on host (orc_level twin: common.cuh:node_level), uploaded once
static int alloc_graph_arrays(leann_backend *h, uint64_t level_seed) {
    const size_t n = h->g.n, nn = std::max<size_t>(n, 1);
    std::vector<uint8_t> levels(nn, 0);
    std::vector<uint32_t> uo(nn, 0);
    uint64_t nu = 0;
    for (size_t i = 0; i < n; i++) {
        levels[i] = (h->kind == LEANN_BACKEND_HNSW || h->nav_levels) ? (uint8_t)node_level(level_seed, i, h->g.M) : 0;
        uo[i] = (uint32_t)nu;
        nu += levels[i];
    }
    h->n_upper_lists = nu;
    uint32_t *adj0 = nullptr, *adjU = nullptr, *duo = nullptr;
    BCHECK(hipMalloc((void **)&adj0, nn * h->g.M0 * 4));
    BCHECK(hipMalloc((void **)&adjU, std::max<uint64_t>(nu, 1) * h->g.M * 4));
    BCHECK(hipMalloc((void **)&duo, nn * 4));
    BCHECK(hipMalloc((void **)&h->d_levels, nn));
    BCHECK(hipMemset(adj0, 0xFF, nn * h->g.M0 * 4));
    BCHECK(hipMemset(adjU, 0xFF, std::max<uint64_t>(nu, 1) * h->g.M * 4));
    BCHECK(hipMemcpy(duo, uo.data(), nn * 4, hipMemcpyHostToDevice));
    BCHECK(hipMemcpy(h->d_levels, levels.data(), nn, hipMemcpyHostToDevice));
    h->g.adj0 = adj0;
    h->g.adjU = adjU;
    h->g.upper_off = duo;
    return LEANN_OK;
}

static constexpr uint64_t LEVEL_SEED = 0x5EED0003ull;

static int build_device_impl(int backend, const float *d_vectors, size_t n, size_t dims, size_t ld, size_t graph_degree, size_t complexity,
                             int device, uint64_t key_offset, int take_copy, leann_backend **out);
extern "C" int leann_backend_build_device(int backend, const float *d_vectors, size_t n, size_t dims, size_t ld,
                                          size_t graph_degree, size_t complexity, int device, uint64_t key_offset,
                                          int take_copy, leann_backend **out) {
    try { // host-side bookkeeping of the builder allocates (insertion order, level tables): nothing may be thrown across the C ABI
        return build_device_impl(backend, d_vectors, n, dims, ld, graph_degree, complexity, device, key_offset, take_copy, out);
    } catch (const std::exception &e) {
        leann_set_error("build: %s", e.what());
        return LEANN_ERR_DEVICE;
    }
}
static int build_device_impl(int backend, const float *d_vectors, size_t n, size_t dims, size_t ld, size_t graph_degree, size_t complexity,
                             int device, uint64_t key_offset, int take_copy, leann_backend **out) {
    if (!out || (n && !d_vectors) || dims == 0 || dims > 4096 || ld < dims || (ld & 3) || n >= (1ull << 31)) {
        leann_set_error("leann_backend_build_device: invalid arguments (n=%zu dims=%zu ld=%zu)", n, dims, ld);
        return LEANN_ERR_INVALID;
    }
    if (backend != LEANN_BACKEND_HNSW && backend != LEANN_BACKEND_DISKANN) {
        leann_set_error("Unknown backend: %d", backend);
        return LEANN_ERR_INVALID;
    }
    const size_t maxdeg = backend == LEANN_BACKEND_HNSW ? 32 : 64;
    if (graph_degree < 2 || graph_degree > maxdeg || complexity < 1) {
        leann_set_error("build: graph_degree must be in [2, %zu] and complexity >= 1 (got %zu, %zu)", maxdeg, graph_degree, complexity);
        return LEANN_ERR_INVALID;
    }
    int ndev = 0;
    leann_device_count(&ndev);
    if (device < 0 || device >= ndev) {
        leann_set_error("HIP device %d not available (%d visible). This library has no CPU fallback.", device, ndev);
        return LEANN_ERR_DEVICE;
    }
    BCHECK(hipSetDevice(device));
    leann_backend *h = new leann_backend();
    h->kind = backend;
    h->device = device;
    h->key_offset = key_offset;
    h->efc = (uint32_t)std::max<size_t>(complexity, graph_degree);
    h->alpha = 1.2f; // diskann.rs:91
    h->g.n = n;
    h->g.d = (uint32_t)dims;
    h->g.ld = (uint32_t)ld;
    h->g.M = (uint32_t)graph_degree;
    h->g.M0 = backend == LEANN_BACKEND_HNSW ? (uint32_t)(2 * graph_degree) : (uint32_t)graph_degree;
    h->nav_levels = backend == LEANN_BACKEND_DISKANN && env_int("LEANN_VAMANA_NAV", 0, 0, 1) != 0;
    if (take_copy) {
        float *cp = nullptr;
        if (hipMalloc((void **)&cp, std::max<size_t>(n * ld, 4) * 4) != hipSuccess ||
            (n && hipMemcpy(cp, d_vectors, n * ld * 4, hipMemcpyDeviceToDevice) != hipSuccess)) {
            leann_set_error("build: copying %zu rows failed: %s", n, hipGetErrorString(hipGetLastError()));
            (void)hipFree(cp);
            h->owns_rows = false;
            leann_backend_close(h);
            return LEANN_ERR_DEVICE;
        }
        h->g.X = cp;
        h->owns_rows = true;
    } else {
        h->g.X = d_vectors;
        h->owns_rows = false;
    }
    int rc = alloc_graph_arrays(h, LEVEL_SEED);
    if (rc == LEANN_OK && n) rc = build_on_device(h, 0, 0);
    if (rc) { leann_backend_close(h); return rc; }
    *out = h;
    return LEANN_OK;
}

// BackendBuilder::build — src/backend/mod.rs:55-79
extern "C" int leann_backend_build(int backend, const float *vectors, size_t n, size_t dims, size_t graph_degree,
                                   size_t complexity, const char *index_path_stem) {
    if (!index_path_stem || (n && !vectors) || dims == 0) {
        leann_set_error("leann_backend_build: invalid arguments");
        return LEANN_ERR_INVALID;
    }
    int ndev = 0;
    leann_device_count(&ndev);
    if (ndev < 1) {
        leann_set_error("no HIP device visible. This library has no CPU fallback.");
        return LEANN_ERR_DEVICE;
    }
    BCHECK(hipSetDevice(0));
    const size_t ld = (dims + 3) & ~(size_t)3;
    float *dX = nullptr;
    BCHECK(hipMalloc((void **)&dX, std::max<size_t>(n * ld, 4) * 4));
    if (n) {
        if (ld == dims) BCHECK(hipMemcpy(dX, vectors, n * dims * 4, hipMemcpyHostToDevice));
        else {
            BCHECK(hipMemset(dX, 0, n * ld * 4));
            BCHECK(hipMemcpy2D(dX, ld * 4, vectors, dims * 4, dims * 4, n, hipMemcpyHostToDevice));
        }
    }
    leann_backend *h = nullptr;
    int rc = leann_backend_build_device(backend, dX, n, dims, ld, graph_degree, complexity, 0, 0, 0, &h);
    if (rc) { (void)hipFree(dX); return rc; }
    h->owns_rows = true; // dX now belongs to the handle
    rc = leann_backend_save(h, index_path_stem);
    leann_backend_close(h);
    return rc;
}

// BackendBuilder::add_to_index — src/backend/mod.rs:82-100, hnsw.rs:142-191
extern "C" int leann_backend_add(int backend, const float *vectors, size_t n, size_t dims, size_t start_id,
                                 const char *index_path_stem) {
    if (backend == LEANN_BACKEND_DISKANN) { // mod.rs:93-98
        leann_set_error("DiskANN backend does not support incremental updates. Use --force to rebuild the entire index.");
        return LEANN_ERR_UNSUPPORTED;
    }
    if (backend != LEANN_BACKEND_HNSW || !index_path_stem || (n && !vectors)) {
        leann_set_error("leann_backend_add: invalid arguments");
        return LEANN_ERR_INVALID;
    }
    leann_backend *old = nullptr;
    int rc = leann_backend_open(index_path_stem, backend, dims, "0", &old);
    if (rc) return rc;
    const size_t n_old = old->g.n;
    if (old->g.feat_h) { // a recompute-on index holds encoder inputs, not vectors: appending needs the passages' features, not their embeddings
        leann_set_error("add_to_index: this index stores no vectors (recompute-on); rebuild it from the encoder inputs (leann_recompute_build_index)");
        leann_backend_close(old);
        return LEANN_ERR_UNSUPPORTED;
    }
    if (start_id != n_old) { // keys are positions (hnsw.rs:178-179): appended ids must continue the sequence
        leann_set_error("add_to_index: start_id %zu does not continue the index (%zu vectors)", start_id, n_old);
        leann_backend_close(old);
        return LEANN_ERR_INVALID;
    }
    // Append: the new rows continue the batched insertion from the existing graph (hnsw.rs:142-191 adds to the loaded index).
    // Levels are a hash of the position, so the first n_old nodes keep their upper-list offsets; the stored link distances
    // the builder prunes with are recomputed (link_dist_kernel).  The same algorithm as a one-shot build of the concatenation on a
    // different batch schedule (the appended rows are permuted among themselves only).
    const size_t ld = old->g.ld, nt = n_old + n;
    float *dX = nullptr;
    if (hipMalloc((void **)&dX, std::max<size_t>(nt * ld, 4) * 4) != hipSuccess ||
        (n_old && hipMemcpy(dX, old->g.X, n_old * ld * 4, hipMemcpyDeviceToDevice) != hipSuccess) ||
        (n && (hipMemset(dX + n_old * ld, 0, n * ld * 4) != hipSuccess ||
               hipMemcpy2D(dX + n_old * ld, ld * 4, vectors, dims * 4, dims * 4, n, hipMemcpyHostToDevice) != hipSuccess))) {
        leann_set_error("add_to_index: staging %zu rows on the device failed: %s", nt, hipGetErrorString(hipGetLastError()));
        (void)hipFree(dX);
        leann_backend_close(old);
        return LEANN_ERR_DEVICE;
    }
    leann_backend *h = new leann_backend();
    h->kind = backend;
    h->device = old->device;
    h->key_offset = old->key_offset;
    h->efc = old->efc;
    h->alpha = old->alpha;
    h->g = old->g;
    h->g.X = dX;
    h->g.n = nt;
    h->g.adj0 = nullptr; h->g.adjU = nullptr; h->g.upper_off = nullptr;
    h->owns_rows = true;
    rc = alloc_graph_arrays(h, LEVEL_SEED);
    bool same_levels = true; // an index that did not come from this builder (leann_backend_from_arrays + save) has its own level table
    if (rc == LEANN_OK && n_old) {
        std::vector<uint8_t> la(n_old), lb(n_old);
        if (hipMemcpy(la.data(), old->d_levels, n_old, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(lb.data(), h->d_levels, n_old, hipMemcpyDeviceToHost) != hipSuccess) {
            leann_set_error("add_to_index: reading the level tables failed");
            rc = LEANN_ERR_DEVICE;
        } else {
            same_levels = la == lb;
        }
    }
    if (rc == LEANN_OK && !same_levels) { // foreign level table: rebuild over the concatenation (the pre-append behaviour)
        const size_t M = old->g.M, efc = old->efc;
        leann_backend_close(old);
        h->g.X = nullptr;
        h->owns_rows = false;
        leann_backend_close(h);
        rc = leann_backend_build_device(backend, dX, nt, dims, ld, M, efc, 0, 0, 0, &h);
        if (rc) { (void)hipFree(dX); return rc; }
        h->owns_rows = true;
        rc = leann_backend_save(h, index_path_stem);
        leann_backend_close(h);
        return rc;
    }
    if (rc == LEANN_OK && n_old) {
        if (hipMemcpy(const_cast<uint32_t *>(h->g.adj0), old->g.adj0, n_old * h->g.M0 * 4, hipMemcpyDeviceToDevice) != hipSuccess ||
            (old->n_upper_lists && hipMemcpy(const_cast<uint32_t *>(h->g.adjU), old->g.adjU, old->n_upper_lists * h->g.M * 4,
                                             hipMemcpyDeviceToDevice) != hipSuccess)) {
            leann_set_error("add_to_index: copying the existing graph failed");
            rc = LEANN_ERR_DEVICE;
        }
    }
    const size_t old_lists = old->n_upper_lists;
    leann_backend_close(old);
    if (rc == LEANN_OK && old_lists > h->n_upper_lists) { leann_set_error("add_to_index: inconsistent level table"); rc = LEANN_ERR_INVALID; }
    if (rc == LEANN_OK && n) rc = build_on_device(h, n_old, 0);
    if (rc) { leann_backend_close(h); return rc; }
    h->owns_rows = true;
    rc = leann_backend_save(h, index_path_stem);
    leann_backend_close(h);
    return rc;
}
