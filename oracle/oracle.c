/*
 * oracle.c — CPU restatement of the leann-rs ANN search hot path (see oracle.h header).
 * TEST INFRASTRUCTURE ONLY: never linked into, loaded by or called from the product path.
 * Build: gcc -O2 -mavx2 -mfma -ffp-contract=off (no fast-math) — see oracle/Makefile.
 */
#define _GNU_SOURCE
#include "oracle.h"
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ============================================================================================
 * Deterministic synthetic data — integer hashing + exactly-rounded f32 ops only, so that the
 * HIP generator (csrc/gen.hip) emits identical bits.  SURVEY.md §8d.
 * ========================================================================================== */
uint64_t orc_mix64(uint64_t x) { /* splitmix64 finaliser */
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
uint64_t orc_hash3(uint64_t seed, uint64_t a, uint64_t b) {
    return orc_mix64(orc_mix64(seed ^ (a * 0xD1342543DE82EF95ull)) ^ (b * 0xA24BAED4963EE407ull));
}
/* Irwin-Hall(4) of 16-bit uniforms, centred and scaled to unit variance. */
float orc_gauss(uint64_t seed, uint64_t a, uint64_t b) {
    uint64_t h = orc_hash3(seed, a, b);
    int32_t s = (int32_t)((h & 0xFFFF) + ((h >> 16) & 0xFFFF) + ((h >> 32) & 0xFFFF) + (h >> 48));
    return (float)(s - 131070) * 2.6428996e-05f; /* 1/sqrt(4*(65536^2-1)/12) */
}

#define TAG_P 0x50524F4A00000000ull /* projection */
#define TAG_C 0x43454E5400000000ull /* centres    */
#define TAG_A 0x4153534700000000ull /* assignment */
#define TAG_N 0x4E4F495300000000ull /* noise      */

void orc_gen_rows(uint64_t seed, uint32_t d, uint32_t r, uint32_t n_clusters, float sigma,
                  uint32_t stream, uint64_t i0, uint64_t n, float *out) {
    float *P = NULL, *z = NULL;
    if (r) {
        P = (float *)malloc((size_t)r * d * sizeof(float));
        z = (float *)malloc((size_t)r * sizeof(float));
        for (uint32_t k = 0; k < r; k++)
            for (uint32_t j = 0; j < d; j++) P[(size_t)k * d + j] = orc_gauss(seed ^ TAG_P, k, j);
    }
    uint64_t nseed = seed ^ TAG_N ^ ((uint64_t)stream * 0x9E3779B97F4A7C15ull);
    for (uint64_t ii = 0; ii < n; ii++) {
        uint64_t i = i0 + ii;
        float *x = out + ii * d;
        if (!r) {
            for (uint32_t j = 0; j < d; j++) x[j] = orc_gauss(nseed, i, j);
        } else {
            uint64_t c = orc_hash3(seed ^ TAG_A, stream, i) % n_clusters;
            for (uint32_t k = 0; k < r; k++)
                z[k] = fmaf(sigma, orc_gauss(nseed, i, k), orc_gauss(seed ^ TAG_C, c, k));
            for (uint32_t j = 0; j < d; j++) x[j] = 0.0f;
            for (uint32_t k = 0; k < r; k++) { /* per element: fmaf chain in increasing k */
                const float *Pk = P + (size_t)k * d;
                float zk = z[k];
                for (uint32_t j = 0; j < d; j++) x[j] = fmaf(Pk[j], zk, x[j]);
            }
        }
        float nrm = sqrtf(orc_dot_canon(x, x, d));
        if (nrm < 1e-12f) nrm = 1e-12f;
        for (uint32_t j = 0; j < d; j++) x[j] = x[j] / nrm;
    }
    free(P);
    free(z);
}

/* ============================================================================================
 * Dot products
 * ========================================================================================== */
float orc_dot_canon_ref(const float *a, const float *b, uint32_t d) {
    float acc[256];
    for (int i = 0; i < 256; i++) acc[i] = 0.0f;
    for (uint32_t j = 0; j < d; j++) acc[j & 255] = fmaf(a[j], b[j], acc[j & 255]);
    for (int w = 128; w >= 1; w >>= 1)
        for (int i = 0; i < w; i++) acc[i] = acc[2 * i] + acc[2 * i + 1];
    return acc[0];
}
/* AVX2/FMA form of the same definition: the 256 strided accumulators live in two banks of 16 ymm
 * registers; the adjacent-pair tree is three hadd levels inside each group of four registers
 * (levels 1-3), then 128-bit hadds (levels 4-8).  Bit-identical to orc_dot_canon_ref (tested). */
#include <immintrin.h>
float orc_dot_canon(const float *a, const float *b, uint32_t d) {
    float acc[256] __attribute__((aligned(32)));
    const uint32_t full = d & ~255u;
    for (int h = 0; h < 2; h++) {
        __m256 v[16];
        for (int i = 0; i < 16; i++) v[i] = _mm256_setzero_ps();
        for (uint32_t t = 0; t < full; t += 256) {
            const float *pa = a + t + 128 * h, *pb = b + t + 128 * h;
            for (int i = 0; i < 16; i++)
                v[i] = _mm256_fmadd_ps(_mm256_loadu_ps(pa + 8 * i), _mm256_loadu_ps(pb + 8 * i), v[i]);
        }
        for (int i = 0; i < 16; i++) _mm256_store_ps(acc + 128 * h + 8 * i, v[i]);
    }
    for (uint32_t j = full; j < d; j++) acc[j - full] = fmaf(a[j], b[j], acc[j - full]);
    __m128 s[8];
    for (int g = 0; g < 8; g++) {
        __m256 A = _mm256_load_ps(acc + 32 * g), B = _mm256_load_ps(acc + 32 * g + 8);
        __m256 C = _mm256_load_ps(acc + 32 * g + 16), D = _mm256_load_ps(acc + 32 * g + 24);
        __m256 r = _mm256_hadd_ps(_mm256_hadd_ps(A, B), _mm256_hadd_ps(C, D));
        s[g] = _mm_add_ps(_mm256_castps256_ps128(r), _mm256_extractf128_ps(r, 1));
    }
    __m128 t0 = _mm_hadd_ps(s[0], s[1]), t1 = _mm_hadd_ps(s[2], s[3]);
    __m128 t2 = _mm_hadd_ps(s[4], s[5]), t3 = _mm_hadd_ps(s[6], s[7]);
    __m128 u0 = _mm_hadd_ps(t0, t1), u1 = _mm_hadd_ps(t2, t3);
    __m128 w = _mm_hadd_ps(u0, u1);
    w = _mm_hadd_ps(w, w);
    w = _mm_hadd_ps(w, w);
    return _mm_cvtss_f32(w);
}
/* Plain 4-accumulator AVX2 dot (what a SIMD library such as usearch's would do).  NOT bit-compatible
 * with the GPU; used only when timing the CPU baseline (orc_set_fast_dot(1)), never for parity. */
float orc_dot_fast(const float *a, const float *b, uint32_t d) {
    __m256 s0 = _mm256_setzero_ps(), s1 = s0, s2 = s0, s3 = s0;
    uint32_t j = 0;
    for (; j + 32 <= d; j += 32) {
        s0 = _mm256_fmadd_ps(_mm256_loadu_ps(a + j), _mm256_loadu_ps(b + j), s0);
        s1 = _mm256_fmadd_ps(_mm256_loadu_ps(a + j + 8), _mm256_loadu_ps(b + j + 8), s1);
        s2 = _mm256_fmadd_ps(_mm256_loadu_ps(a + j + 16), _mm256_loadu_ps(b + j + 16), s2);
        s3 = _mm256_fmadd_ps(_mm256_loadu_ps(a + j + 24), _mm256_loadu_ps(b + j + 24), s3);
    }
    s0 = _mm256_add_ps(_mm256_add_ps(s0, s1), _mm256_add_ps(s2, s3));
    __m128 x = _mm_add_ps(_mm256_castps256_ps128(s0), _mm256_extractf128_ps(s0, 1));
    x = _mm_hadd_ps(x, x);
    x = _mm_hadd_ps(x, x);
    float r = _mm_cvtss_f32(x);
    for (; j < d; j++) r += a[j] * b[j];
    return r;
}
static int g_fast_dot = 0;
void orc_set_fast_dot(int on) { g_fast_dot = on; }
/* src/index/recompute.rs:137-139 — a.iter().zip(b).map(|(x,y)| x*y).sum(): product rounded,
 * then added left to right starting from 0.0 (f32 Sum), stops at the shorter length. */
float orc_dot_seq(const float *a, const float *b, uint32_t d) {
    float s = 0.0f;
    for (uint32_t j = 0; j < d; j++) {
        float p = a[j] * b[j];
        s = s + p;
    }
    return s;
}
float orc_dot_seqfma(const float *a, const float *b, uint32_t d) {
    float s = 0.0f;
    for (uint32_t j = 0; j < d; j++) s = fmaf(a[j], b[j], s);
    return s;
}

uint32_t orc_f32_orderable(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
float orc_orderable_f32(uint32_t u) {
    u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline uint64_t mk_key(float dist, uint32_t id) {
    return ((uint64_t)orc_f32_orderable(dist) << 32) | id;
}
static inline uint32_t key_id(uint64_t k) { return (uint32_t)k; }
static inline float key_dist(uint64_t k) { return orc_orderable_f32((uint32_t)(k >> 32)); }

/* ============================================================================================
 * Recompute / brute-force scan — src/index/recompute.rs:96-109
 * ========================================================================================== */
typedef struct {
    uint64_t idx;
    float score;
} scored_t;
/* stable merge sort, descending by score; comparator mirrors
 * b.1.partial_cmp(&a.1).unwrap_or(Equal): element y goes before x only if y.score > x.score */
static void msort_desc(scored_t *v, scored_t *tmp, uint64_t n) {
    if (n < 2) return;
    uint64_t h = n / 2;
    msort_desc(v, tmp, h);
    msort_desc(v + h, tmp, n - h);
    uint64_t i = 0, j = h, o = 0;
    while (i < h && j < n) {
        if (v[j].score > v[i].score) tmp[o++] = v[j++];
        else tmp[o++] = v[i++];
    }
    while (i < h) tmp[o++] = v[i++];
    while (j < n) tmp[o++] = v[j++];
    memcpy(v, tmp, n * sizeof(scored_t));
}
void orc_scan_topk(const float *X, uint64_t n, uint32_t d, const float *q, uint32_t k, int mode,
                   const uint8_t *allow_mask, uint64_t *keys, float *scores, uint32_t *n_out) {
    scored_t *v = (scored_t *)malloc((n ? n : 1) * sizeof(scored_t));
    scored_t *tmp = (scored_t *)malloc((n ? n : 1) * sizeof(scored_t));
    uint64_t m = 0;
    for (uint64_t i = 0; i < n; i++) {
        if (allow_mask && !((allow_mask[i >> 3] >> (i & 7)) & 1)) continue; /* recompute.rs:66-71 */
        const float *x = X + i * d;
        float s = mode == 0 ? orc_dot_seq(q, x, d) : mode == 1 ? orc_dot_seqfma(q, x, d) : orc_dot_canon(q, x, d);
        v[m].idx = i;
        v[m].score = s;
        m++;
    }
    msort_desc(v, tmp, m);
    uint32_t out = 0;
    for (uint64_t i = 0; i < m && out < k; i++, out++) {
        keys[out] = v[i].idx;
        scores[out] = v[i].score;
    }
    *n_out = out;
    free(v);
    free(tmp);
}

void orc_l2_normalize(float *x, uint32_t d) { /* candle.rs:218-225 */
    float ss = 0.0f;
    for (uint32_t j = 0; j < d; j++) ss = ss + x[j] * x[j];
    float nrm = sqrtf(ss);
    if (nrm < 1e-12f) nrm = 1e-12f;
    for (uint32_t j = 0; j < d; j++) x[j] = x[j] / nrm;
}

/* ============================================================================================
 * Recompute "embedding provider" restatement: dense layer + L2 normalise — the tail of
 * src/embedding/candle.rs:165 (forward), :218-225 (l2_normalize), with bf16 inputs and f32
 * accumulation (csrc/recompute.hip).  GPU accumulates on MFMA in a different order: parity for this
 * floating-point stage is tolerance based (1e-5 on scores, tests/test_gpu_recompute.py).
 * ========================================================================================== */
uint16_t orc_bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
static inline float bf16_f32(uint16_t b) {
    uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
#define TAG_F 0x4645415400000000ull
void orc_synth_features(uint64_t seed, uint32_t h, uint32_t r_int, uint32_t n_clusters, float sigma, uint32_t stream,
                        uint64_t i0, uint64_t n, uint16_t *out) {
    uint64_t nseed = seed ^ TAG_N ^ ((uint64_t)stream * 0x9E3779B97F4A7C15ull);
    float *A = NULL, *z = NULL;
    if (r_int) {
        A = (float *)malloc((size_t)h * r_int * sizeof(float));
        z = (float *)malloc((size_t)r_int * sizeof(float));
        for (uint32_t k = 0; k < h; k++)
            for (uint32_t m = 0; m < r_int; m++) A[(size_t)k * r_int + m] = orc_gauss(seed ^ TAG_F, k, m);
    }
    for (uint64_t ii = 0; ii < n; ii++) {
        uint64_t i = i0 + ii, c = orc_hash3(seed ^ TAG_A, stream, i) % n_clusters;
        if (!r_int) {
            for (uint32_t k = 0; k < h; k++)
                out[ii * h + k] = orc_bf16_rne(fmaf(sigma, orc_gauss(nseed, i, k), orc_gauss(seed ^ TAG_C, c, k)));
            continue;
        }
        for (uint32_t m = 0; m < r_int; m++) z[m] = fmaf(sigma, orc_gauss(nseed, i, m), orc_gauss(seed ^ TAG_C, c, m));
        for (uint32_t k = 0; k < h; k++) {
            float acc = 0.0f;
            for (uint32_t m = 0; m < r_int; m++) acc = fmaf(A[(size_t)k * r_int + m], z[m], acc);
            out[ii * h + k] = orc_bf16_rne(acc);
        }
    }
    free(A);
    free(z);
}
void orc_synth_weights(uint64_t seed, uint32_t h, uint32_t d, uint16_t *out) {
    for (uint32_t k = 0; k < h; k++)
        for (uint32_t j = 0; j < d; j++) out[(size_t)k * d + j] = orc_bf16_rne(orc_gauss(seed ^ TAG_P, k, j));
}
void orc_recompute_encode(const uint16_t *F, uint64_t n, uint32_t h, const uint16_t *W, uint32_t d, float *out) {
    for (uint64_t i = 0; i < n; i++) {
        float *x = out + i * d;
        for (uint32_t j = 0; j < d; j++) x[j] = 0.0f;
        for (uint32_t k = 0; k < h; k++) {
            float fk = bf16_f32(F[i * h + k]);
            const uint16_t *wk = W + (size_t)k * d;
            for (uint32_t j = 0; j < d; j++) x[j] = fmaf(bf16_f32(wk[j]), fk, x[j]);
        }
        orc_l2_normalize(x, d);
    }
}

/* Token-level provider: dense per token, masked mean over the L tokens of a passage
 * (candle.rs:191-216: sum(output * mask) / clamp(sum(mask), 1e-9)), then l2_normalize (:218-225). */
void orc_recompute_encode_pooled(const uint16_t *F, const uint8_t *mask, uint64_t n, uint32_t L, uint32_t h,
                                 const uint16_t *W, uint32_t d, float *out) {
    float *e = (float *)malloc((size_t)d * sizeof(float));
    for (uint64_t i = 0; i < n; i++) {
        float *x = out + i * d;
        float cnt = 0.0f;
        for (uint32_t j = 0; j < d; j++) x[j] = 0.0f;
        for (uint32_t t = 0; t < L; t++) {
            const uint16_t *f = F + (i * L + t) * h;
            float m = (!mask || mask[i * L + t]) ? 1.0f : 0.0f;
            for (uint32_t j = 0; j < d; j++) e[j] = 0.0f;
            for (uint32_t k = 0; k < h; k++) {
                float fk = bf16_f32(f[k]);
                const uint16_t *wk = W + (size_t)k * d;
                for (uint32_t j = 0; j < d; j++) e[j] = fmaf(bf16_f32(wk[j]), fk, e[j]);
            }
            for (uint32_t j = 0; j < d; j++) x[j] = x[j] + e[j] * m;
            cnt = cnt + m;
        }
        if (cnt < 1e-9f) cnt = 1e-9f;
        for (uint32_t j = 0; j < d; j++) x[j] = x[j] / cnt;
        orc_l2_normalize(x, d);
    }
    free(e);
}

/* ============================================================================================
 * Graph index
 * ========================================================================================== */
struct orc_graph {
    const float *X;
    uint64_t n;
    uint32_t d, ld, M, M0, max_level, entry;
    uint8_t *levels;
    uint32_t *upper_off;
    uint32_t *adj0;
    uint32_t *adjU;
    uint64_t n_upper_lists;
    int owns;
    /* recompute-on mode: rows are [feat_h bf16 features][f32 ||W^T f||][pad], row_bytes apart; queries are g = W q */
    const unsigned char *Xb;
    uint32_t feat_h, row_bytes;
};

uint32_t orc_level(uint64_t seed, uint64_t i, uint32_t M) {
    /* P(level >= l) = M^-l  (mL = 1/ln M, Malkov §4): compare a 64-bit uniform against 2^64/M^l */
    uint64_t u = orc_hash3(seed, i, 0x4C45564Cull);
    uint64_t thr = 0xFFFFFFFFFFFFFFFFull;
    uint32_t l = 0;
    while (l < 15) {
        thr /= M;
        if (u >= thr) break;
        l++;
    }
    return l;
}

static inline const uint32_t *nbrs(const orc_graph *g, uint32_t node, uint32_t level, uint32_t *cap) {
    if (level == 0) {
        *cap = g->M0;
        return g->adj0 + (size_t)node * g->M0;
    }
    *cap = g->M;
    return g->adjU + ((size_t)g->upper_off[node] + (level - 1)) * g->M;
}
static inline void prefetch_row(const orc_graph *g, uint32_t id) {
    const char *p = g->feat_h ? (const char *)(g->Xb + (size_t)id * g->row_bytes) : (const char *)(g->X + (size_t)id * g->ld);
    __builtin_prefetch(p, 0, 0);
    __builtin_prefetch(p + 64, 0, 0);
    __builtin_prefetch(p + 128, 0, 0);
    __builtin_prefetch(p + 192, 0, 0);
}
static inline float bf16_f32(uint16_t b);
static float gdist_feat(const orc_graph *g, const float *q, uint32_t id) {
    const unsigned char *row = g->Xb + (size_t)id * g->row_bytes;
    const uint16_t *f = (const uint16_t *)row;
    float x[1024], nrm;
    for (uint32_t j = 0; j < g->feat_h; j++) x[j] = bf16_f32(f[j]);
    memcpy(&nrm, row + 2 * (size_t)g->feat_h, 4);
    return 1.0f - orc_dot_canon(q, x, g->feat_h) / nrm; /* csrc/search.cuh: wave_dist_rows_feat */
}
static inline float gdist(const orc_graph *g, const float *q, uint32_t id) {
    if (g->feat_h) return gdist_feat(g, q, id);
    if (g_fast_dot) return 1.0f - orc_dot_fast(q, g->X + (size_t)id * g->ld, g->d);
    return 1.0f - orc_dot_canon(q, g->X + (size_t)id * g->ld, g->d);
}

typedef struct {
    uint32_t *stamp;
    uint32_t epoch;
    uint64_t *C;
    uint32_t nC, capC; /* min-heap of candidates */
    uint64_t *W;
    uint32_t nW, capW; /* max-heap (algo 0) / sorted list with flag (algo 1) of results */
    uint64_t *exp;
    uint32_t nexp, capexp; /* expanded nodes of the last layer (Vamana build) */
    uint64_t stats[3];
    /* filtered search: every allowed key evaluated on the final layer (its seed included) */
    const uint8_t *allow;
    int collect;
    uint64_t *F;
    uint32_t nF, capF;
} ctx_t;

static ctx_t *ctx_new(uint64_t n, uint32_t ef) {
    ctx_t *c = (ctx_t *)calloc(1, sizeof(ctx_t));
    c->stamp = (uint32_t *)calloc(n ? n : 1, sizeof(uint32_t));
    c->capC = 1024;
    c->C = (uint64_t *)malloc(c->capC * 8);
    c->capW = ef + 2;
    c->W = (uint64_t *)malloc((size_t)c->capW * 8);
    c->capexp = 1024;
    c->exp = (uint64_t *)malloc(c->capexp * 8);
    return c;
}
static void ctx_free(ctx_t *c) {
    free(c->stamp);
    free(c->C);
    free(c->W);
    free(c->exp);
    free(c->F);
    free(c);
}
static void filt_push(ctx_t *c, uint64_t k) {
    uint32_t id = key_id(k);
    if (!((c->allow[id >> 3] >> (id & 7)) & 1)) return;
    if (c->nF == c->capF) {
        c->capF = c->capF ? c->capF * 2 : 1024;
        c->F = (uint64_t *)realloc(c->F, (size_t)c->capF * 8);
    }
    c->F[c->nF++] = k;
}
static void ctx_new_epoch(ctx_t *c, uint64_t n) {
    if (++c->epoch == 0) {
        memset(c->stamp, 0, n * 4);
        c->epoch = 1;
    }
}
static void minheap_push(ctx_t *c, uint64_t k) {
    if (c->nC == c->capC) {
        c->capC *= 2;
        c->C = (uint64_t *)realloc(c->C, (size_t)c->capC * 8);
    }
    uint32_t i = c->nC++;
    while (i) {
        uint32_t p = (i - 1) / 2;
        if (c->C[p] <= k) break;
        c->C[i] = c->C[p];
        i = p;
    }
    c->C[i] = k;
}
static uint64_t minheap_pop(ctx_t *c) {
    uint64_t top = c->C[0], last = c->C[--c->nC];
    uint32_t i = 0, n = c->nC;
    for (;;) {
        uint32_t l = 2 * i + 1, r = l + 1, m;
        if (l >= n) break;
        m = (r < n && c->C[r] < c->C[l]) ? r : l;
        if (c->C[m] >= last) break;
        c->C[i] = c->C[m];
        i = m;
    }
    if (n) c->C[i] = last;
    return top;
}
static void maxheap_push(ctx_t *c, uint64_t k) {
    uint32_t i = c->nW++;
    while (i) {
        uint32_t p = (i - 1) / 2;
        if (c->W[p] >= k) break;
        c->W[i] = c->W[p];
        i = p;
    }
    c->W[i] = k;
}
static void maxheap_pop(ctx_t *c) {
    uint64_t last = c->W[--c->nW];
    uint32_t i = 0, n = c->nW;
    for (;;) {
        uint32_t l = 2 * i + 1, r = l + 1, m;
        if (l >= n) break;
        m = (r < n && c->W[r] > c->W[l]) ? r : l;
        if (c->W[m] <= last) break;
        c->W[i] = c->W[m];
        i = m;
    }
    if (n) c->W[i] = last;
}
static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}
static void exp_push(ctx_t *c, uint64_t k) {
    if (c->nexp == c->capexp) {
        c->capexp *= 2;
        c->exp = (uint64_t *)realloc(c->exp, (size_t)c->capexp * 8);
    }
    c->exp[c->nexp++] = k;
}

/* SEARCH-LAYER (Malkov & Yashunin Alg. 2), entry keys in `ep` (dist already known).
 * On return c->W[0..nW) is sorted ascending by (dist, id). */
static void search_layer_heap(const orc_graph *g, const float *q, const uint64_t *ep, uint32_t nep,
                              uint32_t ef, uint32_t level, ctx_t *c) {
    ctx_new_epoch(c, g->n);
    c->nC = c->nW = c->nexp = 0;
    if (c->capW < ef + 2) {
        c->capW = ef + 2;
        c->W = (uint64_t *)realloc(c->W, (size_t)c->capW * 8);
    }
    for (uint32_t i = 0; i < nep; i++) {
        c->stamp[key_id(ep[i])] = c->epoch;
        if (c->collect) filt_push(c, ep[i]);
        minheap_push(c, ep[i]);
        maxheap_push(c, ep[i]);
        if (c->nW > ef) maxheap_pop(c);
    }
    while (c->nC) {
        uint64_t cur = minheap_pop(c);
        if (c->nW == ef && cur > c->W[0]) break; /* closest candidate is farther than worst result */
        if (level == 0) c->stats[1]++; else c->stats[2]++;
        exp_push(c, cur);
        uint32_t cap;
        const uint32_t *nb = nbrs(g, key_id(cur), level, &cap);
        for (uint32_t t = 0; t < cap; t++) {
            uint32_t e = nb[t];
            if (e == ORC_EMPTY) continue;
            if (t + 1 < cap && nb[t + 1] != ORC_EMPTY) prefetch_row(g, nb[t + 1]); /* as CPU HNSW libraries do */
            if (c->stamp[e] == c->epoch) continue;
            c->stamp[e] = c->epoch;
            uint64_t k = mk_key(gdist(g, q, e), e);
            c->stats[0]++;
            if (c->collect) filt_push(c, k);
            if (c->nW < ef || k < c->W[0]) {
                minheap_push(c, k);
                maxheap_push(c, k);
                if (c->nW > ef) maxheap_pop(c);
            }
        }
    }
    qsort(c->W, c->nW, 8, cmp_u64);
}

/* GreedySearch (DiskANN/Vamana Alg. 1): one list of at most L entries sorted by (dist, id),
 * repeatedly expand the closest not-yet-expanded entry.  Keys carry no flag here; a parallel
 * byte array marks expansion. */
static void search_layer_list(const orc_graph *g, const float *q, const uint64_t *ep, uint32_t nep,
                              uint32_t L, uint32_t level, ctx_t *c) {
    ctx_new_epoch(c, g->n);
    c->nW = c->nexp = 0;
    if (c->capW < L + 2) {
        c->capW = L + 2;
        c->W = (uint64_t *)realloc(c->W, (size_t)c->capW * 8);
    }
    uint8_t *done = (uint8_t *)calloc(L + 2, 1);
    for (uint32_t i = 0; i < nep; i++) {
        c->stamp[key_id(ep[i])] = c->epoch;
        if (c->collect) filt_push(c, ep[i]);
        uint32_t p = c->nW;
        while (p && c->W[p - 1] > ep[i]) { c->W[p] = c->W[p - 1]; p--; }
        c->W[p] = ep[i];
        if (c->nW < L) c->nW++;
    }
    for (;;) {
        uint32_t p = 0;
        while (p < c->nW && done[p]) p++;
        if (p == c->nW) break;
        done[p] = 1;
        uint64_t cur = c->W[p];
        if (level == 0) c->stats[1]++; else c->stats[2]++;
        exp_push(c, cur);
        uint32_t cap;
        const uint32_t *nb = nbrs(g, key_id(cur), level, &cap);
        for (uint32_t t = 0; t < cap; t++) {
            uint32_t e = nb[t];
            if (e == ORC_EMPTY) continue;
            if (t + 1 < cap && nb[t + 1] != ORC_EMPTY) prefetch_row(g, nb[t + 1]);
            if (c->stamp[e] == c->epoch) continue;
            c->stamp[e] = c->epoch;
            uint64_t k = mk_key(gdist(g, q, e), e);
            c->stats[0]++;
            if (c->collect) filt_push(c, k);
            if (c->nW == L && k > c->W[L - 1]) continue;
            uint32_t pos = c->nW < L ? c->nW : L - 1; /* slot that falls off / new tail */
            while (pos && c->W[pos - 1] > k) {
                c->W[pos] = c->W[pos - 1];
                done[pos] = done[pos - 1];
                pos--;
            }
            c->W[pos] = k;
            done[pos] = 0;
            if (c->nW < L) c->nW++;
        }
    }
    free(done);
}

static void search_layer(const orc_graph *g, const float *q, const uint64_t *ep, uint32_t nep,
                         uint32_t ef, uint32_t level, int algo, ctx_t *c) {
    if (algo == 0) search_layer_heap(g, q, ep, nep, ef, level, c);
    else search_layer_list(g, q, ep, nep, ef, level, c);
}

/* K-NN-SEARCH (Malkov Alg. 5): greedy (ef=1) descent through the upper levels, ef-beam on level 0.
 * For a single-level graph (Vamana) this is GreedySearch from the medoid. */
static void graph_search_ctx(const orc_graph *g, const float *q, uint32_t k, uint32_t ef, int algo,
                             ctx_t *c, uint64_t *keys, float *dists, uint32_t *n_out) {
    *n_out = 0;
    if (g->n == 0 || k == 0) return;
    if (ef < k) ef = k; /* diskann.rs:54 beam = max(complexity, top_k) */
    uint64_t best = mk_key(gdist(g, q, g->entry), g->entry);
    c->stats[0]++;
    for (uint32_t lv = g->max_level; lv >= 1; lv--) {
        search_layer(g, q, &best, 1, 1, lv, algo, c);
        best = c->W[0];
    }
    c->collect = c->allow != NULL;
    c->nF = 0;
    search_layer(g, q, &best, 1, ef, 0, algo, c);
    c->collect = 0;
    /* Filtered search (replaces the over-fetch + post-filter of searcher.rs:129-133,:190-194):
     * the traversal is the unfiltered one; the answer is the k best ALLOWED keys among everything whose distance
     * the final layer evaluated (its seed included) - a superset of the beam, at no extra traversal cost. */
    const uint64_t *src = c->W;
    uint32_t ns = c->nW;
    if (c->allow) {
        qsort(c->F, c->nF, 8, cmp_u64);
        src = c->F;
        ns = c->nF;
    }
    uint32_t m = ns < k ? ns : k;
    for (uint32_t i = 0; i < m; i++) {
        keys[i] = key_id(src[i]);
        dists[i] = key_dist(src[i]);
    }
    *n_out = m;
}

int orc_graph_search(const orc_graph *g, const float *q, uint32_t k, uint32_t ef, int algo,
                     uint64_t *keys, float *dists, uint32_t *n_out, uint64_t *stats) {
    ctx_t *c = ctx_new(g->n, ef > k ? ef : k);
    graph_search_ctx(g, q, k, ef, algo, c, keys, dists, n_out);
    if (stats) memcpy(stats, c->stats, sizeof(c->stats));
    ctx_free(c);
    return 0;
}

int orc_graph_search_filtered(const orc_graph *g, const float *q, uint32_t k, uint32_t ef, int algo,
                              const uint8_t *allow, uint64_t *keys, float *dists, uint32_t *n_out, uint64_t *stats) {
    ctx_t *c = ctx_new(g->n, ef > k ? ef : k);
    c->allow = allow;
    graph_search_ctx(g, q, k, ef, algo, c, keys, dists, n_out);
    if (stats) memcpy(stats, c->stats, sizeof(c->stats));
    ctx_free(c);
    return 0;
}

typedef struct {
    const orc_graph *g;
    const float *Q;
    uint64_t nq, lo, hi;
    uint32_t k, ef;
    int algo;
    uint64_t *keys;
    float *dists;
    uint32_t *counts;
    uint64_t *stats;
    const uint8_t *allow; /* optional: bitmap(s) of allowed positions, allow_stride bytes apart per query (0 = shared) */
    uint64_t allow_stride;
} batch_job;
static void *batch_worker(void *p) {
    batch_job *j = (batch_job *)p;
    ctx_t *c = ctx_new(j->g->n, j->ef > j->k ? j->ef : j->k);
    for (uint64_t i = j->lo; i < j->hi; i++) {
        c->stats[0] = c->stats[1] = c->stats[2] = 0;
        c->allow = j->allow ? j->allow + i * j->allow_stride : NULL;
        graph_search_ctx(j->g, j->Q + i * j->g->d, j->k, j->ef, j->algo, c, j->keys + i * j->k,
                         j->dists + i * j->k, j->counts + i);
        if (j->stats) memcpy(j->stats + i * 3, c->stats, sizeof(c->stats));
    }
    ctx_free(c);
    return NULL;
}
int orc_graph_search_filtered_batch(const orc_graph *g, const float *Q, uint64_t nq, uint32_t k, uint32_t ef,
                                    int algo, uint32_t nthreads, const uint8_t *allow, uint64_t allow_stride,
                                    uint64_t *keys, float *dists, uint32_t *counts, uint64_t *stats);
int orc_graph_search_batch(const orc_graph *g, const float *Q, uint64_t nq, uint32_t k, uint32_t ef,
                           int algo, uint32_t nthreads, uint64_t *keys, float *dists,
                           uint32_t *counts, uint64_t *stats) {
    return orc_graph_search_filtered_batch(g, Q, nq, k, ef, algo, nthreads, NULL, 0, keys, dists, counts, stats);
}
int orc_graph_search_filtered_batch(const orc_graph *g, const float *Q, uint64_t nq, uint32_t k, uint32_t ef,
                                    int algo, uint32_t nthreads, const uint8_t *allow, uint64_t allow_stride,
                                    uint64_t *keys, float *dists, uint32_t *counts, uint64_t *stats) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > nq) nthreads = nq ? (uint32_t)nq : 1;
    pthread_t *th = (pthread_t *)malloc(nthreads * sizeof(pthread_t));
    batch_job *jobs = (batch_job *)malloc(nthreads * sizeof(batch_job));
    for (uint32_t t = 0; t < nthreads; t++) {
        batch_job j = {g, Q, nq, nq * t / nthreads, nq * (t + 1) / nthreads, k, ef, algo, keys, dists, counts, stats, allow, allow_stride};
        jobs[t] = j;
        if (nthreads == 1) batch_worker(&jobs[t]);
        else pthread_create(&th[t], NULL, batch_worker, &jobs[t]);
    }
    if (nthreads > 1)
        for (uint32_t t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    free(th);
    free(jobs);
    return 0;
}

/* ---- construction --------------------------------------------------------------------------- */
static orc_graph *graph_alloc(const float *X, uint64_t n, uint32_t d, uint32_t M, uint32_t M0,
                              const uint8_t *levels_in) {
    orc_graph *g = (orc_graph *)calloc(1, sizeof(orc_graph));
    g->X = X;
    g->n = n;
    g->d = d;
    g->ld = d;
    g->M = M;
    g->M0 = M0;
    g->owns = 1;
    g->levels = (uint8_t *)calloc(n ? n : 1, 1);
    g->upper_off = (uint32_t *)calloc(n ? n : 1, 4);
    uint64_t nu = 0;
    for (uint64_t i = 0; i < n; i++) {
        g->levels[i] = levels_in ? levels_in[i] : 0;
        g->upper_off[i] = (uint32_t)nu;
        nu += g->levels[i];
    }
    g->n_upper_lists = nu;
    g->adj0 = (uint32_t *)malloc((n ? n : 1) * (size_t)M0 * 4);
    memset(g->adj0, 0xFF, (n ? n : 1) * (size_t)M0 * 4);
    g->adjU = (uint32_t *)malloc((nu ? nu : 1) * (size_t)M * 4);
    memset(g->adjU, 0xFF, (nu ? nu : 1) * (size_t)M * 4);
    return g;
}
static inline uint32_t *nbrs_mut(orc_graph *g, uint32_t node, uint32_t level, uint32_t *cap) {
    return (uint32_t *)nbrs(g, node, level, cap);
}
static uint32_t list_len(const uint32_t *l, uint32_t cap) {
    uint32_t n = 0;
    while (n < cap && l[n] != ORC_EMPTY) n++;
    return n;
}
/* SELECT-NEIGHBORS-HEURISTIC (Malkov Alg. 4, no extendCandidates / keepPruned) generalised by
 * alpha (alpha = 1: HNSW; Vamana RobustPrune Alg. 2 uses alpha >= 1):
 * walk candidates by ascending (dist to p, id); keep c unless some kept r has
 * alpha * dist(c, r) <= dist(c, p)   [Vamana]   /   dist(c, r) < dist(c, p)   [HNSW, alpha == 0 flag] */
/* RobustPrune in two forms.  The paper's Alg. 2 walks the pool once with alpha.  DiskANN's implementation (occlude_list) walks it
 * with alpha = 1.0 first — the diverse core, drawn from the WHOLE pool — and only fills the slots still free with the relaxed alpha.
 * On data of high intrinsic dimension the one-stage rule at alpha = 1.2 occludes almost nothing, a list becomes the R nearest of the
 * pool, and at R = 32 the graph stops being navigable as the corpus grows (scripts/exp/vamana_scale.py / vamana_seq_oracle.py:
 * profiles/r03_vamana_scale.md).  Which form diskann-rs 0.3.4 uses cannot be checked offline; the builders here default to the
 * two-stage form (orc_set_vamana_two_stage(0) = the paper's). */
static int g_two_stage = 1;
void orc_set_vamana_two_stage(int on) { g_two_stage = on; }
static uint32_t select_two_stage(const orc_graph *g, const uint64_t *cands, uint32_t nc, uint32_t lim, float alpha, uint32_t *out) {
    uint8_t *taken = (uint8_t *)calloc(nc ? nc : 1, 1);
    uint32_t ns = 0;
    for (int stage = 0; stage < 2 && ns < lim; stage++) {
        const float a = stage == 0 ? 1.0f : alpha;
        for (uint32_t i = 0; i < nc && ns < lim; i++) {
            if (taken[i]) continue;
            uint32_t cid = key_id(cands[i]);
            float dcp = key_dist(cands[i]);
            const float *xc = g->X + (size_t)cid * g->ld;
            int good = 1;
            for (uint32_t s = 0; s < ns && good; s++)
                if (a * gdist(g, xc, out[s]) <= dcp) good = 0;
            if (good) { out[ns++] = cid; taken[i] = 1; }
        }
    }
    free(taken);
    return ns;
}
static uint32_t select_heuristic(const orc_graph *g, const uint64_t *cands, uint32_t nc, uint32_t lim,
                                 float alpha, uint32_t *out) {
    if (alpha > 1.0f && g_two_stage) return select_two_stage(g, cands, nc, lim, alpha, out);
    uint32_t ns = 0;
    for (uint32_t i = 0; i < nc && ns < lim; i++) {
        uint32_t cid = key_id(cands[i]);
        float dcp = key_dist(cands[i]);
        int good = 1;
        const float *xc = g->X + (size_t)cid * g->ld;
        for (uint32_t s = 0; s < ns; s++) {
            float dcr = gdist(g, xc, out[s]);
            if (alpha == 0.0f ? (dcr < dcp) : (alpha * dcr <= dcp)) {
                good = 0;
                break;
            }
        }
        if (good) out[ns++] = cid;
    }
    return ns;
}
static void link_back(orc_graph *g, uint32_t s, uint32_t q, uint32_t level, float alpha) {
    uint32_t cap;
    uint32_t *l = nbrs_mut(g, s, level, &cap);
    uint32_t len = list_len(l, cap);
    for (uint32_t i = 0; i < len; i++)
        if (l[i] == q) return;
    if (len < cap) {
        l[len] = q;
        return;
    }
    uint64_t *cands = (uint64_t *)malloc((size_t)(cap + 1) * 8);
    const float *xs = g->X + (size_t)s * g->ld;
    for (uint32_t i = 0; i < cap; i++) cands[i] = mk_key(gdist(g, xs, l[i]), l[i]);
    cands[cap] = mk_key(gdist(g, xs, q), q);
    qsort(cands, cap + 1, 8, cmp_u64);
    uint32_t *sel = (uint32_t *)malloc((size_t)cap * 4);
    uint32_t ns = select_heuristic(g, cands, cap + 1, cap, alpha, sel);
    for (uint32_t i = 0; i < cap; i++) l[i] = i < ns ? sel[i] : ORC_EMPTY;
    free(cands);
    free(sel);
}

orc_graph *orc_hnsw_build(const float *X, uint64_t n, uint32_t d, uint32_t M, uint32_t efc,
                          uint64_t level_seed) {
    uint8_t *lv = (uint8_t *)malloc(n ? n : 1);
    for (uint64_t i = 0; i < n; i++) lv[i] = (uint8_t)orc_level(level_seed, i, M);
    orc_graph *g = graph_alloc(X, n, d, M, 2 * M, lv);
    free(lv);
    if (n == 0) return g;
    ctx_t *c = ctx_new(n, efc);
    uint32_t *sel = (uint32_t *)malloc((size_t)(2 * M + 1) * 4);
    g->entry = 0;
    g->max_level = g->levels[0];
    for (uint64_t qi = 1; qi < n; qi++) { /* hnsw.rs:128-130: sequential add(i, v) */
        uint32_t q = (uint32_t)qi, l = g->levels[q];
        const float *xq = X + qi * d;
        uint64_t best = mk_key(gdist(g, xq, g->entry), g->entry);
        for (uint32_t lc = g->max_level; lc > l; lc--) {
            search_layer_heap(g, xq, &best, 1, 1, lc, c);
            best = c->W[0];
        }
        for (int lc = (int)(l < g->max_level ? l : g->max_level); lc >= 0; lc--) {
            search_layer_heap(g, xq, &best, 1, efc, (uint32_t)lc, c);
            uint32_t ns = select_heuristic(g, c->W, c->nW, M, 0.0f, sel);
            uint32_t cap;
            uint32_t *lq = nbrs_mut(g, q, (uint32_t)lc, &cap);
            for (uint32_t i = 0; i < ns; i++) lq[i] = sel[i];
            best = c->W[0];
            for (uint32_t i = 0; i < ns; i++) link_back(g, sel[i], q, (uint32_t)lc, 0.0f);
        }
        if (l > g->max_level) {
            g->max_level = l;
            g->entry = q;
        }
    }
    free(sel);
    ctx_free(c);
    return g;
}

orc_graph *orc_vamana_build(const float *X, uint64_t n, uint32_t d, uint32_t R, uint32_t L,
                            float alpha, uint64_t seed) {
    orc_graph *g = graph_alloc(X, n, d, R, R, NULL);
    if (n == 0) return g;
    /* random start graph: min(R, n-1) distinct out-neighbours per node */
    for (uint64_t i = 0; i < n; i++) {
        uint32_t *l = g->adj0 + i * R, cnt = 0, want = (n - 1 < R) ? (uint32_t)(n - 1) : R;
        for (uint64_t t = 0; cnt < want; t++) {
            uint32_t e = (uint32_t)(orc_hash3(seed, i, t) % n);
            int dup = (e == i);
            for (uint32_t s = 0; s < cnt && !dup; s++) dup = (l[s] == e);
            if (!dup) l[cnt++] = e;
        }
    }
    /* medoid: closest point to the (normalised direction of the) mean, by the index metric */
    double *mean = (double *)calloc(d, sizeof(double));
    for (uint64_t i = 0; i < n; i++)
        for (uint32_t j = 0; j < d; j++) mean[j] += X[i * d + j];
    float *mf = (float *)malloc(d * sizeof(float));
    for (uint32_t j = 0; j < d; j++) mf[j] = (float)(mean[j] / (double)n);
    uint64_t bestk = ~0ull;
    for (uint64_t i = 0; i < n; i++) {
        uint64_t k = mk_key(gdist(g, mf, (uint32_t)i), (uint32_t)i);
        if (k < bestk) bestk = k;
    }
    g->entry = key_id(bestk);
    free(mean);
    free(mf);

    ctx_t *c = ctx_new(n, L);
    uint64_t *pool = (uint64_t *)malloc(((size_t)n < 65536 ? 65536 : n) * 8);
    uint32_t *sel = (uint32_t *)malloc((size_t)(R + 1) * 4);
    uint64_t *order = (uint64_t *)malloc(n * 8);
    for (int pass = 0; pass < 2; pass++) {
        float a = pass == 0 ? 1.0f : alpha;
        for (uint64_t i = 0; i < n; i++) order[i] = (orc_hash3(seed ^ 0x5045524Dull, (uint64_t)pass, i) & ~0xFFFFFFFFull) | i;
        qsort(order, n, 8, cmp_u64);
        for (uint64_t oi = 0; oi < n; oi++) {
            uint32_t p = (uint32_t)order[oi];
            const float *xp = X + (size_t)p * d;
            uint64_t ep = mk_key(gdist(g, xp, g->entry), g->entry);
            search_layer_list(g, xp, &ep, 1, L, 0, c);
            /* pool = expanded nodes ∪ N_out(p), minus p, deduplicated */
            uint32_t np = 0;
            for (uint32_t t = 0; t < c->nexp; t++)
                if (key_id(c->exp[t]) != p) pool[np++] = c->exp[t];
            uint32_t *lp = g->adj0 + (size_t)p * R;
            uint32_t len = list_len(lp, R);
            for (uint32_t t = 0; t < len; t++) pool[np++] = mk_key(gdist(g, xp, lp[t]), lp[t]);
            qsort(pool, np, 8, cmp_u64);
            uint32_t u = 0;
            for (uint32_t t = 0; t < np; t++)
                if (t == 0 || pool[t] != pool[t - 1]) pool[u++] = pool[t];
            uint32_t ns = select_heuristic(g, pool, u, R, a, sel);
            for (uint32_t t = 0; t < R; t++) lp[t] = t < ns ? sel[t] : ORC_EMPTY;
            for (uint32_t t = 0; t < ns; t++) link_back(g, sel[t], p, 0, a);
        }
    }
    free(order);
    free(sel);
    free(pool);
    ctx_free(c);
    return g;
}

orc_graph *orc_graph_from_arrays(const float *X, uint64_t n, uint32_t d, uint32_t ld, uint32_t M,
                                 uint32_t M0, uint32_t max_level, uint32_t entry,
                                 const uint8_t *levels, const uint32_t *upper_off,
                                 const uint32_t *adj0, const uint32_t *adjU, uint64_t n_upper_lists) {
    orc_graph *g = (orc_graph *)calloc(1, sizeof(orc_graph));
    g->X = X;
    g->n = n;
    g->d = d;
    g->ld = ld;
    g->M = M;
    g->M0 = M0;
    g->max_level = max_level;
    g->entry = entry;
    g->levels = (uint8_t *)levels;
    g->upper_off = (uint32_t *)upper_off;
    g->adj0 = (uint32_t *)adj0;
    g->adjU = (uint32_t *)adjU;
    g->n_upper_lists = n_upper_lists;
    g->owns = 0;
    return g;
}
/* recompute-on: attach feature rows (borrowed); queries passed to orc_graph_search* are then g = W q (feat_h f32) */
void orc_graph_set_features(orc_graph *g, const unsigned char *rows, uint32_t feat_h, uint32_t row_bytes) {
    g->Xb = rows;
    g->feat_h = feat_h;
    g->row_bytes = row_bytes;
    g->d = feat_h; /* query stride of the batch API */
}
/* g[k] = sum_j W[k][j] * q[j], one k-ordered fmaf chain per k (f32 MFMA order of csrc/scan.hip:score_mfma_kernel) */
void orc_project_query(const uint16_t *W, uint32_t h, uint32_t hp4, uint32_t d, const float *q, float *gq) {
    for (uint32_t k = 0; k < hp4; k++) {
        float s = 0.0f;
        if (k < h)
            for (uint32_t j = 0; j < d; j++) s = fmaf(bf16_f32(W[(size_t)k * d + j]), q[j], s);
        gq[k] = s;
    }
}
void orc_graph_info(const orc_graph *g, uint64_t *out) {
    out[0] = g->n;
    out[1] = g->d;
    out[2] = g->ld;
    out[3] = g->M;
    out[4] = g->M0;
    out[5] = g->max_level;
    out[6] = g->entry;
    out[7] = g->n_upper_lists;
}
void orc_graph_export(const orc_graph *g, uint8_t *levels, uint32_t *upper_off, uint32_t *adj0,
                      uint32_t *adjU) {
    memcpy(levels, g->levels, g->n);
    memcpy(upper_off, g->upper_off, g->n * 4);
    memcpy(adj0, g->adj0, g->n * (size_t)g->M0 * 4);
    if (g->n_upper_lists) memcpy(adjU, g->adjU, g->n_upper_lists * (size_t)g->M * 4);
}
void orc_graph_free(orc_graph *g) {
    if (!g) return;
    if (g->owns) {
        free(g->levels);
        free(g->upper_off);
        free(g->adj0);
        free(g->adjU);
    }
    free(g);
}

/* ============================================================================================
 * Cross-shard top-k merge (new): ascending by (dist, key), deterministic.
 * ========================================================================================== */
void orc_merge_topk(const uint64_t *keys, const float *dists, const uint32_t *counts,
                    uint32_t n_shards, uint32_t k_in, uint32_t k_out, uint64_t *out_keys,
                    float *out_dists, uint32_t *out_n) {
    uint32_t *pos = (uint32_t *)calloc(n_shards, 4), o = 0;
    while (o < k_out) {
        int best = -1;
        for (uint32_t s = 0; s < n_shards; s++) {
            if (pos[s] >= counts[s]) continue;
            if (best < 0) { best = (int)s; continue; }
            float db = dists[(size_t)best * k_in + pos[best]], ds = dists[(size_t)s * k_in + pos[s]];
            uint32_t ub = orc_f32_orderable(db), us = orc_f32_orderable(ds);
            uint64_t kb = keys[(size_t)best * k_in + pos[best]], ks = keys[(size_t)s * k_in + pos[s]];
            if (us < ub || (us == ub && ks < kb)) best = (int)s;
        }
        if (best < 0) break;
        out_keys[o] = keys[(size_t)best * k_in + pos[best]];
        out_dists[o] = dists[(size_t)best * k_in + pos[best]];
        pos[best]++;
        o++;
    }
    *out_n = o;
    free(pos);
}

/* ============================================================================================
 * hybrid_rerank — src/index/bm25.rs:135-170, f32 op for op.
 * ========================================================================================== */
typedef struct {
    uint64_t idx;
    float score;
} hr_t;
void orc_hybrid_rerank(const uint64_t *idx, const float *vscore, uint32_t n, const float *bm25,
                       uint64_t n_bm25, float alpha, uint64_t *out_idx, float *out_score) {
    float max_v = -INFINITY, min_v = INFINITY; /* bm25.rs:141-148 fold(f32::max / f32::min) */
    for (uint32_t i = 0; i < n; i++) {
        max_v = fmaxf(max_v, vscore[i]);
        min_v = fminf(min_v, vscore[i]);
    }
    float vrange = fmaxf(max_v - min_v, 1e-6f); /* :149 */
    float max_b = -INFINITY, min_b = INFINITY; /* :152-153 */
    for (uint64_t i = 0; i < n_bm25; i++) {
        max_b = fmaxf(max_b, bm25[i]);
        min_b = fminf(min_b, bm25[i]);
    }
    float brange = fmaxf(max_b - min_b, 1e-6f); /* :154 */
    scored_t *v = (scored_t *)malloc((n ? n : 1) * sizeof(scored_t));
    scored_t *tmp = (scored_t *)malloc((n ? n : 1) * sizeof(scored_t));
    for (uint32_t i = 0; i < n; i++) {
        float norm_vec = (vscore[i] - min_v) / vrange;                      /* :159 */
        float b = idx[i] < n_bm25 ? bm25[idx[i]] : 0.0f;                    /* :160 */
        float norm_b = (b - min_b) / brange;                                /* :161 */
        float t1 = alpha * norm_vec, t2 = (1.0f - alpha) * norm_b;          /* :163 */
        v[i].idx = idx[i];
        v[i].score = t1 + t2;
    }
    msort_desc(v, tmp, n); /* :168 stable sort desc */
    for (uint32_t i = 0; i < n; i++) {
        out_idx[i] = v[i].idx;
        out_score[i] = v[i].score;
    }
    free(v);
    free(tmp);
}
