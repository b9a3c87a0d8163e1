// hybrid.hip — the hybrid leg of IndexSearcher::search_with_options for BATCHES of queries (BASELINE configs[4]: DiskANN + hybrid BM25
// rerank).
//
// Reference, per query (src/index/searcher.rs:146-169, src/index/bm25.rs:135-170):
//     vector_results = backend.search(query, fetch_k = 5 * top_k)              (idx, DISTANCE)          searcher.rs:129-143
//     bm25_scores    = Bm25Scorer::score_query(text)        -> Vec<f32> over ALL N passages              :153
//     bm25_top       = Bm25Scorer::search(text, fetch_k)    -> positives, score desc (stable: index asc) :154, bm25.rs:109-122
//     BM25-only hits are appended with vector score 0.0                                                   :160-165
//     hybrid_rerank: norm_v = (v - min_v) / max(max_v - min_v, 1e-6) over the merged list,
//                    norm_b = (b - min_b) / max(max_b - min_b, 1e-6) with min / max over ALL N scores,
//                    alpha * norm_v + (1 - alpha) * norm_b, stable sort descending                        bm25.rs:135-170
// The single-query form runs on the host (host/leann_host.hpp, op for op).  A batch of 16 384 queries would spend longer in a host
// rerank than in the traversal, so the same arithmetic — every f32 operation in the same order, IEEE division, no contraction — runs
// here, one workgroup per query, on the lists the traversal kernel left in HBM.  The BM25 side stays what the host's persistent
// Bm25Scorer produces (string / hash work is host work), handed over SPARSE: a BM25 score vector is zero except for
// the passages that share a term with the query, so a query brings its positives (position, score), sorted as Bm25Scorer::search
// sorts them; every other passage scores 0.0 — which is also where min_b comes from whenever fewer than N passages are positive.
// Score polarity: compat_polarity != 0 blends the backend's distances as the reference does (larger distance = larger
// vector term); 0 = corrected, 1 - dist.  Checked bit for bit against oracle/searcher_oracle.py (tests/test_gpu_hybrid.py, bench.py --hybrid).
#include "common.cuh"
#include "../../include/leann_backend.h"
#include "internal.h"

#define HYB_MAX_FETCH 256
#define HYB_MAX_MERGED (2 * HYB_MAX_FETCH)

__global__ void __launch_bounds__(256) hybrid_rerank_kernel(const uint64_t *__restrict__ keys, const float *__restrict__ dists,
                                                            const uint32_t *__restrict__ counts, uint32_t fetch_k,
                                                            const uint32_t *__restrict__ bm_pos, const float *__restrict__ bm_score,
                                                            const uint32_t *__restrict__ bm_count, uint32_t bm_stride, uint64_t n_docs,
                                                            float alpha, int compat, uint32_t top_k, uint64_t *__restrict__ out_keys,
                                                            float *__restrict__ out_scores, uint32_t *__restrict__ out_counts) {
    __shared__ uint64_t m_key[HYB_MAX_MERGED];
    __shared__ float m_v[HYB_MAX_MERGED];
    __shared__ float m_score[HYB_MAX_MERGED];
    __shared__ uint64_t s_sort[HYB_MAX_MERGED];
    __shared__ uint32_t s_inj[HYB_MAX_FETCH];
    __shared__ uint32_t s_minv, s_maxv, s_minb, s_maxb, s_m;
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    const uint32_t n = min(counts[q], fetch_k);
    const uint32_t P = min(bm_count[q], bm_stride);
    const uint32_t top = min(P, fetch_k); // bm25_top = the first fetch_k positives (searcher.rs:154)
    const uint32_t *bp = bm_pos + (size_t)q * bm_stride;
    const float *bs = bm_score + (size_t)q * bm_stride;
    if (tid == 0) { s_minv = 0xFFFFFFFFu; s_maxv = 0u; s_minb = 0xFFFFFFFFu; s_maxb = 0u; }
    for (uint32_t i = tid; i < n; i += 256) {
        const float d = dists[(size_t)q * fetch_k + i];
        m_key[i] = keys[(size_t)q * fetch_k + i];
        m_v[i] = compat ? d : 1.0f - d;
    }
    __syncthreads();
    // BM25-only hits: positives of bm25_top that the backend did not return, appended in bm25_top order with vector score 0.0
    for (uint32_t t = tid; t < top; t += 256) {
        const uint64_t pos = bp[t];
        uint32_t found = 0;
        for (uint32_t i = 0; i < n; i++) found |= (m_key[i] == pos);
        s_inj[t] = found ? 0u : 1u;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t m = n;
        for (uint32_t t = 0; t < top; t++)
            if (s_inj[t]) { m_key[m] = bp[t]; m_v[m] = 0.0f; m++; }
        s_m = m;
    }
    __syncthreads();
    const uint32_t m = s_m;
    // f32::max / f32::min folds (bm25.rs:140-147, :152-154) — order-independent for non-NaN values
    for (uint32_t i = tid; i < m; i += 256) {
        const uint32_t o = f32_orderable(m_v[i]);
        atomicMin(&s_minv, o);
        atomicMax(&s_maxv, o);
    }
    for (uint32_t t = tid; t < P; t += 256) {
        const uint32_t o = f32_orderable(bs[t]);
        atomicMin(&s_minb, o);
        atomicMax(&s_maxb, o);
    }
    if (tid == 0 && (uint64_t)P < n_docs) { // every passage without a term of the query scores 0.0
        const uint32_t z = f32_orderable(0.0f);
        atomicMin(&s_minb, z);
        atomicMax(&s_maxb, z);
    }
    __syncthreads();
    const float min_v = orderable_f32(s_minv), max_v = orderable_f32(s_maxv);
    const float min_b = orderable_f32(s_minb), max_b = orderable_f32(s_maxb);
    const float v_range = fmaxf(max_v - min_v, 1e-6f), b_range = fmaxf(max_b - min_b, 1e-6f);
    const float one_minus_alpha = 1.0f - alpha;
    for (uint32_t i = tid; i < HYB_MAX_MERGED; i += 256) {
        uint64_t sk = ~0ull;
        if (i < m) {
            const uint64_t key = m_key[i];
            float bm = 0.0f; // bm25_scores[idx], 0.0 beyond the vector (bm25.rs:158)
            for (uint32_t t = 0; t < P; t++)
                if ((uint64_t)bp[t] == key) bm = bs[t];
            const float norm_vec = (m_v[i] - min_v) / v_range;
            const float norm_b = (bm - min_b) / b_range;
            const float t1 = alpha * norm_vec, t2 = one_minus_alpha * norm_b;
            const float sc = t1 + t2;
            m_score[i] = sc;
            sk = ((uint64_t)(~f32_orderable(sc)) << 32) | i; // ascending = score descending, ties in list order: Rust's stable sort_by
        }
        s_sort[i] = sk;
    }
    int npow = 2;
    while (npow < (int)m) npow <<= 1;
    bitonic_sort_lds(s_sort, npow); // (entries >= m are ~0 and npow <= HYB_MAX_MERGED)
    const uint32_t nout = min(m, top_k);
    for (uint32_t j = tid; j < top_k; j += 256) {
        if (j < nout) {
            const uint32_t i = (uint32_t)s_sort[j];
            out_keys[(size_t)q * top_k + j] = m_key[i];
            out_scores[(size_t)q * top_k + j] = m_score[i];
        } else {
            out_keys[(size_t)q * top_k + j] = ~0ull;
            out_scores[(size_t)q * top_k + j] = -INFINITY;
        }
    }
    if (tid == 0) out_counts[q] = nout;
}

extern "C" int leann_hybrid_rerank_device(const uint64_t *d_keys, const float *d_dists, const uint32_t *d_counts, size_t nq, size_t fetch_k,
                                          const uint32_t *d_bm25_pos, const float *d_bm25_score, const uint32_t *d_bm25_count,
                                          size_t bm25_stride, size_t n_docs, float alpha, int compat_polarity, size_t top_k,
                                          uint64_t *d_out_keys, float *d_out_scores, uint32_t *d_out_counts, void *stream) {
    if (!d_keys || !d_dists || !d_counts || !d_bm25_count || (bm25_stride && (!d_bm25_pos || !d_bm25_score)) || !d_out_keys || !d_out_scores ||
        !d_out_counts || top_k == 0 || fetch_k == 0) {
        leann_set_error("leann_hybrid_rerank_device: null/zero argument");
        return LEANN_ERR_INVALID;
    }
    if (fetch_k > HYB_MAX_FETCH || top_k > HYB_MAX_MERGED) {
        leann_set_error("leann_hybrid_rerank_device: fetch_k %zu > %d (the reference fetches 5 * top_k, searcher.rs:129-133)", fetch_k, HYB_MAX_FETCH);
        return LEANN_ERR_INVALID;
    }
    if (!(alpha >= 0.0f && alpha <= 1.0f)) {
        leann_set_error("leann_hybrid_rerank_device: alpha %g outside [0, 1]", (double)alpha);
        return LEANN_ERR_INVALID;
    }
    if (nq == 0) return LEANN_OK;
    hipLaunchKernelGGL(hybrid_rerank_kernel, dim3((unsigned)nq), dim3(256), 0, (hipStream_t)stream, d_keys, d_dists, d_counts, (uint32_t)fetch_k,
                       d_bm25_pos, d_bm25_score, d_bm25_count, (uint32_t)bm25_stride, (uint64_t)n_docs, alpha, compat_polarity, (uint32_t)top_k,
                       d_out_keys, d_out_scores, d_out_counts);
    HIP_CHECK_RET(hipGetLastError());
    return LEANN_OK;
}
