// search.cuh — on-GPU graph traversal for HNSW (level descent + ef beam) and Vamana (single level,
// medoid entry).  Replaces the arithmetic behind
//     HnswSearcher::search     src/backend/hnsw.rs:79-88   (usearch::Index::search, out of tree)
//     DiskAnnSearcher::search  src/backend/diskann.rs:47-62 (diskann_rs search_with_dists, out of tree)
// Restated algorithm: oracle/oracle.c (search_layer_heap / search_layer_list / graph_search_ctx).
//
// Mapping: one workgroup (NW waves of 64) per query.
//   * query vector: registers of every wave (lane l holds elements 256t+4l..+3), loaded once;
//   * result/candidate beam W: sorted u64 keys in LDS, double buffered, merged by rank;
//   * visited set: open-addressing hash table in LDS; a query that outgrows it migrates, mid-search, to a pooled
//     table in HBM (generation-tagged slots touched only by atomics) and continues;
//   * per hop: wave 0 reads the adjacency list (one neighbour id per lane) and filters it through
//     the visited table; the not-yet-seen rows are dealt round-robin to the NW waves, each wave
//     streams a whole row per instruction group (64 lanes x 16 B = 1 KiB coalesced), R rows in
//     flight, and reduces with the canonical wave tree (common.cuh);
//   * merge: one work item per old / new entry ranks itself against the new keys with pipelined 16-byte LDS scans
//     and scatters into the other W buffer; a wave-level min picks the next candidate;
//   * filtered search (allow-bitmap): same traversal; a second sorted list R in LDS keeps the k best
//     ALLOWED keys among everything evaluated on the target level (top-k of a set: order independent, so again
//     bit-identical to the oracle's collect-sort-truncate);
//   * exactly one candidate is expanded per step, in the same order as the sequential algorithm,
//     so ids AND distances are bit-identical to the oracle.
#pragma once
#include "common.cuh"

struct GraphView {
    const float *X;            // [n x ld] rows, 16-byte aligned, zero padded
    const uint32_t *adj0;      // [n x M0] level-0 neighbours, LEANN_EMPTY padded
    const uint32_t *adjU;      // [n_upper_lists x M] upper-level neighbour lists
    const uint32_t *upper_off; // [n] first upper list of a node (list of level l at upper_off+l-1)
    uint64_t n;
    uint32_t d, ld, M, M0, max_level, entry;
    // recompute-on graph search (no stored vectors): X points at feature rows instead —
    // [feat_h bf16 features][f32 ||W^T f||][pad], `row_bytes` apart; the query side is g = W q (feat_h f32) and
    // dist = 1 - <f, g> / ||W^T f||  ==  1 - <l2norm(W^T f), q>.  feat_h == 0: plain f32 rows.
    uint32_t feat_h, row_bytes;
    // Rows of exactly 256 features live SPLIT on the device: X = [n x 512 B] of features (row_bytes = 512: every row is four whole,
    // aligned 128-B lines) and `norms` = [n] f32.  With the norm inline a 520-B row touches 5.06 lines on average — 20 % of the
    // fabric traffic was padding (PMC 1.20 x the algorithmic bytes, profiles/r02_other_kernels.md).  Files, exports and the oracle keep
    // the inline form (interleaved on the way out); null: inline norm behind the features.
    const float *norms;
};

// Pools of visited tables in HBM for the rare query whose LDS table fills up.  Pool 1: GPOOL_TABLES tables of
// 2^GPOOL_BITS u64 slots {generation:32 | id:32}.  Never cleared: a slot whose generation differs
// from the owner's current one is empty.  Only atomics touch it, so hand-over between workgroups
// on different XCDs needs no fence.  A query that fills even its pool-1 table (> 49 152 nodes visited on one level:
// beams of ~1 000 and more) moves on to pool 2: a few tables sized by the host to hold EVERY node of the index at 75 % load
// (api.hip:ensure_gpool), so a search can never run out of visited-set space; the `aborted` exit below is a defensive path
// (reported as stat 3 and turned into LEANN_ERR_OVERFLOW by the host entry points; reachable only through the debug knobs).
#define GPOOL_TABLES 1024u
#define GPOOL_BITS 16u

struct SearchArgs {
    const float *queries;     // [nq x ldq] (ignored when q_rows != nullptr)
    const uint32_t *q_rows;   // optional: query i is base row q_rows[i] (index construction)
    uint32_t ldq, nq, k, ef, target_level;
    uint32_t hash_bits;       // LDS visited table = 1 << hash_bits slots
    uint64_t key_offset;
    uint64_t *out_keys;       // [nq x k]
    float *out_dists;         // [nq x k]
    uint32_t *out_counts;     // [nq]
    uint32_t *out_stats;      // [nq x 4] evals, hops0, hopsU, visited-set level (0 LDS, 1 / 2 HBM pool, 3 = overflowed: empty result)  (optional)
    uint64_t *out_expanded;   // optional [nq x exp_cap]: (orderable(dist) << 32 | id) of every node expanded on the
    uint32_t *out_nexp;       //          target level, in expansion order (Vamana's visited set V), and its count
    uint32_t exp_cap;
    unsigned long long *gpool; // [gpool_tables << gpool_bits]
    uint32_t *gpool_lock;     // [gpool_tables] 0 = free
    uint32_t *gpool_ctr;      // [0] acquire ticket (pool 1), [1] generation counter, [2] acquire ticket (pool 2)
    uint32_t gpool_bits, gpool_tables;
    unsigned long long *gpool2; // second-level pool (null: the index is small enough for pool 1 to hold all of it)
    uint32_t *gpool2_lock;
    uint32_t gpool2_bits, gpool2_tables;
    const uint8_t *allow;     // filtered kernels only: bitmap over local positions (bit i of byte i >> 3) ...
    uint64_t allow_stride;    // ... of query i at allow + i * allow_stride (0: one bitmap shared by the batch)
};

// The traversal kernels all take (GraphView g, SearchArgs a) by value.  Two dozen of `a`'s fields are needed only by rare paths (visited
// set migration, expansion trace) or once at the end (outputs); read through the by-value struct they are loaded at kernel entry and
// stay live — in scalar registers the hop loop is short of (it spilled 70 of them into vector lanes) — to the last line.  Read from
// the kernel-argument segment at the point of use they cost one scalar load there and nothing in between.
static_assert(alignof(SearchArgs) == 8 && alignof(GraphView) == 8, "lazy_search_args assumes `a` sits at the 8-byte aligned offset behind `g`");
__device__ __forceinline__ const __attribute__((address_space(4))) SearchArgs *lazy_search_args() {
    typedef const __attribute__((address_space(4))) char *kptr;
    return (const __attribute__((address_space(4))) SearchArgs *)((kptr)__builtin_amdgcn_kernarg_segment_ptr() +
                                                                  ((sizeof(GraphView) + 7) & ~(size_t)7));
}

// A value every lane read from the same LDS word: tell the compiler it is uniform, so that it and everything derived from it (beam
// size, buffer pointers, loop bounds) live in scalar registers instead of one vector register each.
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

__device__ __forceinline__ uint32_t vis_hash(uint32_t id, uint32_t bits) {
    return (id * 0x9E3779B1u) >> (32 - bits);
}

__device__ __forceinline__ bool vis_insert_lds(uint32_t *tab, uint32_t bits, uint32_t id) {
    const uint32_t mask = (1u << bits) - 1u;
    uint32_t h = vis_hash(id, bits);
    for (;;) {
        uint32_t old = atomicCAS(&tab[h], LEANN_EMPTY, id);
        if (old == LEANN_EMPTY) return true;
        if (old == id) return false;
        h = (h + 1) & mask;
    }
}
__device__ __forceinline__ bool vis_insert_hbm(unsigned long long *tab, uint32_t bits, uint32_t gen, uint32_t id) {
    const uint32_t mask = (1u << bits) - 1u;
    uint32_t h = vis_hash(id, bits);
    const unsigned long long mine = ((unsigned long long)gen << 32) | id;
    for (;;) {
        unsigned long long cur = __hip_atomic_load(&tab[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(cur >> 32) != gen) { // stale generation == empty
            unsigned long long prev = atomicCAS(&tab[h], cur, mine);
            if (prev == cur) return true;
            continue; // somebody of this workgroup took the slot: look at it again
        }
        if ((uint32_t)cur == id) return false;
        h = (h + 1) & mask;
    }
}

// ---- distance of up to R rows per wave, all loads issued before the first use -----------------
template <int T, int R>
__device__ __forceinline__ void wave_dist_rows(const float4 (&q)[T], const float *__restrict__ X, uint32_t ld,
                                               const uint32_t (&ids)[R], int nrows, int lane, float (&out)[R]) {
    float4 v[R][T];
#pragma unroll
    for (int r = 0; r < R; r++) {
        if (r < nrows) {
            const float *row = X + (size_t)ids[r] * ld;
#pragma unroll
            for (int t = 0; t < T; t++) v[r][t] = row_load4(row, ld, t, lane);
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        if (r < nrows) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int t = 0; t < T; t++) fma4(acc, q[t], v[r][t]);
            out[r] = 1.0f - wave_tree_sum(lane4_sum(acc));
        }
    }
}

// Same for bf16 feature rows with the inline norm (recompute-on mode).  Lane l owns elements 256t+4l..+3 again
// (one 8-byte load per chunk), so the canonical accumulation order is unchanged; only the operands differ.
template <int T, int R>
__device__ __forceinline__ void wave_dist_rows_feat(const float4 (&q)[T], const char *__restrict__ Xb, uint32_t row_bytes,
                                                    uint32_t h, const uint32_t (&ids)[R], int nrows, int lane, float (&out)[R],
                                                    const float *__restrict__ norms = nullptr) {
    uint2 v[R][T];
    float nrm[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        if (r < nrows) {
            const char *row = Xb + (size_t)ids[r] * row_bytes;
#pragma unroll
            for (int t = 0; t < T; t++) {
                const uint32_t j = 256u * t + 4u * lane;
                v[r][t] = j < h ? *reinterpret_cast<const uint2 *>(row + 2 * (size_t)j) : make_uint2(0u, 0u);
            }
            nrm[r] = norms ? norms[ids[r]] : *reinterpret_cast<const float *>(row + 2 * (size_t)h);
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        if (r < nrows) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int t = 0; t < T; t++) {
                float4 x;
                x.x = __uint_as_float(v[r][t].x << 16);
                x.y = __uint_as_float(v[r][t].x & 0xFFFF0000u);
                x.z = __uint_as_float(v[r][t].y << 16);
                x.w = __uint_as_float(v[r][t].y & 0xFFFF0000u);
                fma4(acc, q[t], x);
            }
            out[r] = 1.0f - wave_tree_sum(lane4_sum(acc)) / nrm[r];
        }
    }
}

// Recompute-on rows of exactly 256 bf16 features (+ inline norm), FOUR passages per wave instruction: the 16 lanes of a DPP row share a
// passage, lane m of them owns elements 8m..8m+7 and 128+8m..128+8m+7 (two 16-byte loads; each instruction reads a contiguous 256-B
// half of four rows), and one pass of the DPP ladder reduces four rows at once — against one row per 8-byte wave load and one
// six-step ladder per row above.  G such groups in flight per wave.  The sum is the SAME balanced tree over the 64 four-element
// partials (partial v = elements 4v..4v+3, lane m holds partials 2m, 2m+1, 32+2m, 32+2m+1): level 1 and level 6 are additions inside
// the lane, levels 2-5 run on the row's DPP ladder — every pairing is wave_tree_sum's, so the distances are bit-identical.
// Results (and the norm load) live in lane 15 of each DPP row.
typedef uint32_t u32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
__device__ __forceinline__ float feat8_partial_pair(const float (&q)[8], const u32x4_a8 v) {
    float4 lo, hi;
    lo.x = fmaf(q[0], __uint_as_float(v.x << 16), 0.f);
    lo.y = fmaf(q[1], __uint_as_float(v.x & 0xFFFF0000u), 0.f);
    lo.z = fmaf(q[2], __uint_as_float(v.y << 16), 0.f);
    lo.w = fmaf(q[3], __uint_as_float(v.y & 0xFFFF0000u), 0.f);
    hi.x = fmaf(q[4], __uint_as_float(v.z << 16), 0.f);
    hi.y = fmaf(q[5], __uint_as_float(v.z & 0xFFFF0000u), 0.f);
    hi.z = fmaf(q[6], __uint_as_float(v.w << 16), 0.f);
    hi.w = fmaf(q[7], __uint_as_float(v.w & 0xFFFF0000u), 0.f);
    return lane4_sum(lo) + lane4_sum(hi); // level 1: partials 2m and 2m + 1
}
template <int G>
__device__ __forceinline__ void group_dist_rows_feat256(const float (&qa)[8], const float (&qb)[8], const char *__restrict__ Xb,
                                                        uint32_t row_bytes, const uint32_t (&id)[G], const bool (&valid)[G], int lane,
                                                        float (&out)[G], const float *__restrict__ norms) {
    const int m = lane & 15;
    u32x4_a8 va[G], vb[G];
    float nrm[G];
#pragma unroll
    for (int g = 0; g < G; g++) {
        va[g] = vb[g] = u32x4_a8{0u, 0u, 0u, 0u};
        nrm[g] = 1.f;
        if (valid[g]) {
            const char *row = Xb + (size_t)id[g] * row_bytes;
            va[g] = *reinterpret_cast<const u32x4_a8 *>(row + 16 * m);
            vb[g] = *reinterpret_cast<const u32x4_a8 *>(row + 256 + 16 * m);
            if (m == 15) nrm[g] = norms ? norms[id[g]] : *reinterpret_cast<const float *>(row + 512);
        }
    }
#pragma unroll
    for (int g = 0; g < G; g++) {
        float sa = feat8_partial_pair(qa, va[g]), sb = feat8_partial_pair(qb, vb[g]);
        sa = dpp_pair_add<0xB1, 0xf>(sa);  // level 2: lane ^ 1
        sb = dpp_pair_add<0xB1, 0xf>(sb);
        sa = dpp_pair_add<0x4E, 0xf>(sa);  // level 3: lane ^ 2
        sb = dpp_pair_add<0x4E, 0xf>(sb);
        sa = dpp_pair_add<0x114, 0xf>(sa); // level 4: row_shr:4
        sb = dpp_pair_add<0x114, 0xf>(sb);
        sa = dpp_pair_add<0x118, 0xf>(sa); // level 5: row_shr:8 -> lanes 12..15 of the row: partials 0..31 / 32..63
        sb = dpp_pair_add<0x118, 0xf>(sb);
        out[g] = 1.0f - (sa + sb) / nrm[g]; // level 6
    }
}

// neighbour list of `node` on level lv: level 0 lists are [n x M0], upper lists [n_upper_lists x M] at upper_off[node] + lv - 1
__device__ __forceinline__ const uint32_t *adj_list(const uint32_t *__restrict__ base, const uint32_t *__restrict__ upper_off, uint32_t deg,
                                                    int lv, uint32_t node) {
    const size_t list = lv == 0 ? (size_t)node : (size_t)upper_off[node] + (uint32_t)(lv - 1);
    return base + list * deg;
}

// LDS carve-up (dynamic): [W0 | W1 | s_key x2 | s_new x2 | misc | (filtered: R0 | R1 | s_keyR x2) | table]
// s_key / s_new / s_keyR exist once per hop parity: wave 0 prepares hop h + 1 (adjacency list through the visited table) while the
// other waves still merge hop h.
struct SearchLds {
    uint64_t *s_key;
    uint32_t *s_new;
    uint32_t *misc;  // [1],[2]=next selection (by hop parity) [3]=table full [4]=pool slot [5]=generation
                     // [6]=1 if the target level's seed is allowed (filtered) [8],[9]=n_new (by hop parity)
};
// kf = result-list length of a filtered search (0: unfiltered)
// nbuf = 2 for the latency form of the hop loop (NW > 4: s_key / s_new / s_keyR per hop parity), 1 for the throughput form
__host__ __device__ inline size_t search_lds_bytes(uint32_t ef, uint32_t maxdeg, uint32_t hash_bits, uint32_t kf = 0, uint32_t nbuf = 2) {
    size_t efp = (ef + 1) & ~1u;
    size_t b = 2 * efp * 8 + nbuf * ((size_t)maxdeg + 8) * 8 + nbuf * (size_t)maxdeg * 4 + 16 * 4; // s_key carries 8 sentinel slots
    b = (b + 15) & ~(size_t)15;
    if (kf) b += 2 * (size_t)((kf + 1) & ~1u) * 8 + nbuf * ((size_t)maxdeg + 8) * 8;
    return b + ((size_t)1 << hash_bits) * 4;
}

// G16 (recompute-on rows of 256 features only): phase C evaluates four rows per wave instruction (group_dist_rows_feat256), R = groups
// in flight per wave.
template <int T, int R, int NW, bool FEAT, bool FILT = false, bool G16 = false>
__device__ void beam_search_one(const GraphView &g, const SearchArgs &a, uint32_t qi, unsigned char *smem) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const auto *const za = lazy_search_args(); // arguments of the rare paths and of the epilogue: read where they are used
    const uint32_t ef = a.ef;
    const uint32_t maxdeg = g.M0 > g.M ? g.M0 : g.M;
    const uint32_t efp = (ef + 1) & ~1u;
    SearchLds s;
    uint64_t *const W0 = reinterpret_cast<uint64_t *>(smem); // buffer b lives at W0 + b*efp (no pointer array: stays out of scratch)
    constexpr uint32_t NBUF = NW > 4 ? 2u : 1u; // latency form of the hop loop: per-parity buffers (search_lds_bytes)
    const uint32_t kstride = maxdeg + 8;
    s.s_key = W0 + 2 * efp;
    s.s_new = reinterpret_cast<uint32_t *>(s.s_key + NBUF * kstride);
    s.misc = s.s_new + NBUF * maxdeg;
    size_t off = 2 * (size_t)efp * 8 + NBUF * ((size_t)maxdeg + 8) * 8 + NBUF * (size_t)maxdeg * 4 + 64;
    off = (off + 15) & ~(size_t)15;
    // filtered: R list (k best allowed keys so far, double buffered) + the new keys with disallowed ones blanked
    const uint32_t kf = FILT ? a.k : 0u, kfp = (kf + 1) & ~1u;
    uint64_t *const R0 = reinterpret_cast<uint64_t *>(smem + off);
    uint64_t *const s_keyR = R0 + 2 * kfp;
    const uint8_t *const allow = FILT ? a.allow + (size_t)qi * a.allow_stride : nullptr;
    uint32_t rsize = 0;
    int rcur = 0;
    if (FILT) off += 2 * (size_t)kfp * 8 + NBUF * ((size_t)maxdeg + 8) * 8;
    uint32_t *table = reinterpret_cast<uint32_t *>(smem + off);
    const uint32_t hbits = a.hash_bits, hsize = 1u << hbits;
    uint32_t vis_limit = hsize - (hsize >> 2); // 75 % load
    uint32_t hbm = 0;                 // 0: visited set in LDS; 1 / 2: in a pool-1 / pool-2 table in HBM (after an overflow)
    unsigned long long *gtab = nullptr;
    uint32_t gslot = 0, gen = 0, gbits = 0;

    // ---- query into registers -----------------------------------------------------------------
    float4 q[T];
    float qa[8], qb[8]; // G16: elements 8m..8m+7 and 128+8m..+7 of the query, m = lane within the DPP row
    {
        const float *qv;
        uint32_t dq = FEAT ? g.feat_h : g.d;
        if (!FEAT && a.q_rows) qv = g.X + (size_t)a.q_rows[qi] * g.ld;
        else qv = a.queries + (size_t)qi * a.ldq;
#pragma unroll
        for (int t = 0; t < T; t++)
            if (!G16) q[t] = vec_load4_guard(qv, dq, t, lane);
        if (G16) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                qa[i] = qv[8 * (lane & 15) + i];
                qb[i] = qv[128 + 8 * (lane & 15) + i];
            }
        }
    }

    const uint32_t *const adj0_p = g.adj0, *const adjU_p = g.adjU, *const upoff_p = g.upper_off;
    uint32_t n_evals = 1, hops0 = 0, hopsU = 0; // meaningful in wave 0 / lane 0 only
    uint32_t n_vis = 0;
    uint64_t best;
    {
        uint32_t ids[1] = {g.entry};
        float dd[1];
        if (G16) {
            const bool valid[1] = {lane < 16};
            group_dist_rows_feat256<1>(qa, qb, reinterpret_cast<const char *>(g.X), g.row_bytes, ids, valid, lane, dd, g.norms);
            dd[0] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dd[0]), 15));
        } else if (FEAT) wave_dist_rows_feat<T, 1>(q, reinterpret_cast<const char *>(g.X), g.row_bytes, g.feat_h, ids, 1, lane, dd, g.norms);
        else wave_dist_rows<T, 1>(q, g.X, g.ld, ids, 1, lane, dd);
        best = make_key(dd[0], g.entry); // every wave computes the same value
    }

#ifdef LEANN_STAMPS
    uint64_t stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // [7]: hops whose next candidate came from the OLD beam (not from the hop's new keys)
#endif
    uint32_t wsize = 0, level_hops = 0;
    int cur = 0;
    bool aborted = false;
    for (int lv = (int)g.max_level; lv >= (int)a.target_level; --lv) {
        const uint32_t ef_l = (lv == (int)a.target_level) ? ef : 1u;
        const uint32_t deg = lv == 0 ? g.M0 : g.M;
        if (!hbm)
            for (uint32_t i = tid; i < hsize; i += NW * 64) table[i] = LEANN_EMPTY;
        __syncthreads();
        if (tid == 0) {
            W0[0] = best;
            if (hbm) {
                gen = atomicAdd(&za->gpool_ctr[1], 1u) + 1u;
                s.misc[5] = gen;
                vis_insert_hbm(gtab, gbits, gen, key_id(best));
            } else {
                table[vis_hash(key_id(best), hbits)] = key_id(best);
            }
            s.misc[1] = LEANN_EMPTY;
            s.misc[2] = LEANN_EMPTY;
            s.misc[3] = 0;
            if (FILT && lv == (int)a.target_level) {
                const uint32_t e = key_id(best);
                const uint32_t ok = (allow[e >> 3] >> (e & 7)) & 1u;
                if (ok) R0[0] = best;
                s.misc[6] = ok;
            }
        }
        __syncthreads();
        if (hbm) gen = uni(s.misc[5]);
        const bool filt_level = FILT && lv == (int)a.target_level;
        if (filt_level) { rsize = uni(s.misc[6]); rcur = 0; }
        cur = 0;
        wsize = 1;
        n_vis = 1;
        uint32_t sel = 0;
        uint32_t hop = 0;
        // Two forms of the hop loop (identical results: exactly one candidate is expanded per step, in the sequential order).
        // Throughput form (4 waves per query, every CU full): phase B — the adjacency list through the visited table, wave 0 — at the
        // head of the hop, three barriers; its list was prefetched by phase E of the hop before.  Latency form (16 waves per query,
        // batches <= 512): wave 0 runs phase B of hop h + 1 right behind phase E, UNDER the merge of hop h — two barriers per hop, single
        // query 0.365 -> 0.32 ms.  With the chip saturated the adjacency round trip outlasts the merge either way and the leaner wave 0
        // of the throughput form is faster (recompute-on search: 3.05 against 2.98 M queries/s; hnsw10m equal; scripts/exp/ab.sh).
        constexpr bool EARLY_B = NW > 4;
        if constexpr (!EARLY_B) {
            // Adjacency list of the NEXT node to expand, one neighbour id per lane of wave 0 (lists hold at most 64 ids).  The load is
            // issued as soon as the next candidate is known — right after the new distances exist, BEFORE the merge (phase E below) —
            // so its HBM round trip runs under the merge and the barrier instead of at the head of the next hop.
            uint32_t e_pref = LEANN_EMPTY;
            // (level-dependent pieces of the list address as VALUES: selecting between the two struct members by address makes hipcc keep
            // the by-value GraphView in scratch memory)
            const uint32_t *const adj_base = lv == 0 ? adj0_p : adjU_p;
            if (wave == 0) e_pref = (uint32_t)lane < deg ? adj_list(adj_base, upoff_p, deg, lv, key_id(best))[lane] : LEANN_EMPTY;
            while (sel != LEANN_EMPTY) {
                uint64_t *Wc = W0 + cur * efp, *Wn = W0 + (cur ^ 1) * efp;
    #ifdef LEANN_STAMPS // diagnostic build only (scripts/stamps.sh): where does a hop spend its cycles?
                const uint64_t st0 = __builtin_amdgcn_s_memtime();
    #endif
                // ---- phase B: adjacency list through the visited table (wave 0) ------------------
                if (wave == 0) {
                    const uint32_t node = key_id(Wc[sel]);
                    if (za->out_expanded && lv == (int)a.target_level && lane == 0 && hop < za->exp_cap)
                        za->out_expanded[(size_t)qi * za->exp_cap + hop] = ((Wc[sel] >> 32) << 32) | node;
                    uint32_t n_new = 0;
                    bool ovf = (n_vis + deg > vis_limit);
                    if (!ovf) {
                        const uint32_t e = e_pref; // this node's list (a hop redone after a table migration reads the same register again)
                        bool isnew = false;
                        if (e != LEANN_EMPTY) isnew = hbm ? vis_insert_hbm(gtab, gbits, gen, e) : vis_insert_lds(table, hbits, e);
                        unsigned long long m = __ballot(isnew);
                        uint32_t pos = __popcll(m & ((1ull << lane) - 1ull));
                        if (isnew) s.s_new[pos] = e;
                        n_new = __popcll(m);
                    }
                    n_vis += n_new;
                    n_evals += n_new;
                    if (!ovf) { if (lv == 0) hops0++; else hopsU++; }
                    if (lane < 8) s.s_key[n_new + lane] = ~0ull; // sentinels: the merge scans keys 8 at a time
                    if (filt_level && lane < 8) s_keyR[n_new + lane] = ~0ull;
                    if (lane == 0) {
                        s.misc[0] = n_new;
                        if (ovf) s.misc[3] = 1;
                    }
                }
    #ifdef LEANN_STAMPS
                const uint64_t stB = __builtin_amdgcn_s_memtime();
    #endif
                __syncthreads(); // B1
    #ifdef LEANN_STAMPS
                const uint64_t stB1 = __builtin_amdgcn_s_memtime();
    #endif
                const uint32_t table_full = uni(s.misc[3]);
                if (table_full) {
                    // Table full: move the visited set one level up (LDS -> pool 1 -> pool 2) and redo this hop.
                    const uint32_t next = hbm + 1;
                    if (next > 2 || (next == 2 && !za->gpool2)) { aborted = true; break; } // defensive: see the pool comment above
                    __syncthreads(); // every thread has read the flag before it is cleared
                    if (tid == 0) {
                        uint32_t *locks = next == 1 ? za->gpool_lock : za->gpool2_lock;
                        const uint32_t ntab = next == 1 ? za->gpool_tables : za->gpool2_tables;
                        uint32_t t = atomicAdd(&za->gpool_ctr[next == 1 ? 0 : 2], 1u), slot;
                        for (uint32_t i = 0;; i++) { // holders never wait for anything, so a table always comes free
                            slot = (t + i) % ntab;
                            if (atomicCAS(&locks[slot], 0u, 1u) == 0u) break;
                            if ((i % ntab) == ntab - 1) __builtin_amdgcn_s_sleep(32);
                        }
                        s.misc[4] = slot;
                        if (!hbm) s.misc[5] = atomicAdd(&za->gpool_ctr[1], 1u) + 1u; // pool 1 -> 2 keeps the generation
                        s.misc[3] = 0;
                    }
                    __syncthreads();
                    const uint32_t nslot = uni(s.misc[4]);
                    const uint32_t nbits = next == 1 ? za->gpool_bits : za->gpool2_bits;
                    unsigned long long *ntab_p = (next == 1 ? za->gpool : za->gpool2) + ((size_t)nslot << nbits);
                    if (!hbm) {
                        gen = uni(s.misc[5]);
                        for (uint32_t i = tid; i < hsize; i += NW * 64) {
                            uint32_t e = table[i];
                            if (e != LEANN_EMPTY) vis_insert_hbm(ntab_p, nbits, gen, e);
                        }
                    } else {
                        for (uint32_t i = tid; i < (1u << gbits); i += NW * 64) {
                            const unsigned long long cur = __hip_atomic_load(&gtab[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if ((uint32_t)(cur >> 32) == gen) vis_insert_hbm(ntab_p, nbits, gen, (uint32_t)cur);
                        }
                        __syncthreads(); // every probe of the old table has returned
                        if (tid == 0) atomicExch(&za->gpool_lock[gslot], 0u);
                    }
                    gtab = ntab_p;
                    gbits = nbits;
                    gslot = nslot;
                    hbm = next;
                    vis_limit = (1u << gbits) - (1u << (gbits - 2));
                    __syncthreads();
                    continue;
                }
                const uint32_t n_new = uni(s.misc[0]);
                // ---- phase C: stream the new rows, R in flight per wave ------------------------------
                if constexpr (G16) {
                    for (uint32_t jb = 0; jb < n_new; jb += NW * 4 * R) {
                        uint32_t ids[R], jj[R], abyte[R];
                        bool valid[R];
                        float dd[R];
    #pragma unroll
                        for (int r = 0; r < R; r++) {
                            jj[r] = jb + (uint32_t)(r * NW + wave) * 4u + (uint32_t)(lane >> 4);
                            valid[r] = jj[r] < n_new;
                            ids[r] = valid[r] ? s.s_new[jj[r]] : 0u;
                            abyte[r] = (filt_level && valid[r] && (lane & 15) == 15) ? allow[ids[r] >> 3] : 0u;
                        }
                        group_dist_rows_feat256<R>(qa, qb, reinterpret_cast<const char *>(g.X), g.row_bytes, ids, valid, lane, dd, g.norms);
                        if ((lane & 15) == 15) {
    #pragma unroll
                            for (int r = 0; r < R; r++)
                                if (valid[r]) {
                                    const uint64_t key = make_key(dd[r], ids[r]);
                                    s.s_key[jj[r]] = key;
                                    if (filt_level) s_keyR[jj[r]] = ((abyte[r] >> (ids[r] & 7u)) & 1u) ? key : ~0ull;
                                }
                        }
                    }
                } else
                for (uint32_t j0 = wave; j0 < n_new; j0 += NW * R) {
                    uint32_t ids[R];
                    float dd[R];
                    int nrows = 0;
    #pragma unroll
                    for (int r = 0; r < R; r++) {
                        uint32_t j = j0 + r * NW;
                        if (j < n_new) { ids[r] = s.s_new[j]; nrows = r + 1; }
                    }
                    uint32_t abyte = 0, abit = 0;
                    if (filt_level && lane < R && j0 + lane * NW < n_new) { // lane r fetches row r's allow bit alongside the rows
                        const uint32_t e = s.s_new[j0 + lane * NW];
                        abyte = allow[e >> 3];
                        abit = e & 7u;
                    }
                    if (FEAT) wave_dist_rows_feat<T, R>(q, reinterpret_cast<const char *>(g.X), g.row_bytes, g.feat_h, ids, nrows, lane, dd, g.norms);
                    else wave_dist_rows<T, R>(q, g.X, g.ld, ids, nrows, lane, dd);
                    const unsigned long long amask = FILT ? __ballot((abyte >> abit) & 1u) : 0ull;
                    if (lane == 0) {
    #pragma unroll
                        for (int r = 0; r < R; r++)
                            if (r < nrows) {
                                const uint64_t key = make_key(dd[r], ids[r]);
                                s.s_key[j0 + r * NW] = key;
                                if (filt_level) s_keyR[j0 + r * NW] = ((amask >> r) & 1ull) ? key : ~0ull;
                            }
                    }
                }
    #ifdef LEANN_STAMPS
                const uint64_t stC = __builtin_amdgcn_s_memtime();
    #endif
                __syncthreads(); // B2
    #ifdef LEANN_STAMPS
                const uint64_t stB2 = __builtin_amdgcn_s_memtime();
    #endif
                // ---- phase E (wave 0): the next candidate, known before the merge -------------------------------------
                // The merged beam's first unexpanded entry is the smaller of (a) the first unexpanded entry of the OLD beam other than
                // the one just expanded and (b) the smallest new key; its index in the merged beam is the number of keys below it.
                // Same entry, same index as a scan of the merged list would give — but available one merge earlier, so wave 0 issues
                // the next hop's adjacency load now and the other waves merge meanwhile.
                uint32_t *next_slot = &s.misc[1 + (hop & 1)];
                const uint32_t n_pad = (n_new + 7) & ~7u;
                if (wave == 0) {
                    uint32_t u = LEANN_EMPTY;
                    for (uint32_t base = 0; base < wsize && u == LEANN_EMPTY; base += 64) {
                        const uint32_t i = base + lane;
                        const unsigned long long m = __ballot(i < wsize && i != sel && !(Wc[i] & 1ull));
                        if (m) u = base + (uint32_t)__ffsll((long long)m) - 1u;
                    }
                    const uint64_t c_old = u != LEANN_EMPTY ? Wc[u] : ~0ull;
                    const uint64_t kj = (uint32_t)lane < n_new ? s.s_key[lane] : ~0ull; // n_new <= 64: one key per lane
                    const uint32_t hi = wave_min_u32((uint32_t)(kj >> 32));
                    const uint32_t lo = wave_min_u32((uint32_t)(kj >> 32) == hi ? (uint32_t)kj : 0xFFFFFFFFu);
                    const uint64_t c_new = n_new ? (((uint64_t)hi << 32) | lo) : ~0ull;
                    uint32_t next = LEANN_EMPTY, cnode = 0;
                    if (c_old != ~0ull || c_new != ~0ull) {
                        uint32_t rank;
                        if ((c_old >> 1) < (c_new >> 1)) {
                            rank = u + (uint32_t)__popcll(__ballot((kj >> 1) < (c_old >> 1)));
                            cnode = key_id(c_old);
    #ifdef LEANN_STAMPS
                            stamp[7] += 1;
    #endif
                        } else {
                            rank = 0;
                            for (uint32_t base = 0; base < wsize; base += 64) {
                                const uint32_t i = base + lane;
                                rank += (uint32_t)__popcll(__ballot(i < wsize && (Wc[i] >> 1) < (c_new >> 1)));
                            }
                            cnode = key_id(c_new);
                        }
                        if (rank < ef_l) next = rank;
                    }
                    if (lane == 0) *next_slot = next;
                    if (next != LEANN_EMPTY) // in flight across the merge and barrier B3
                        e_pref = (uint32_t)lane < deg ? adj_list(adj_base, upoff_p, deg, lv, cnode)[lane] : LEANN_EMPTY;
                } else {
                    // ---- phase D (waves 1 .. NW-1): merge by rank into the other buffer ---------------------------------
                    // One work item per old entry (rank = index + #new keys below it) and per new key (rank = #old below it, by
                    // binary search, + #new below it); the scan over the new keys reads 16 B per LDS instruction, 8 keys per
                    // unrolled step, so the loads pipeline instead of paying one LDS latency per key.
                    const ulonglong2 *kp = reinterpret_cast<const ulonglong2 *>(s.s_key);
                    for (uint32_t it = tid - 64; it < wsize + n_new; it += (NW - 1) * 64) {
                        const bool isW = it < wsize;
                        uint64_t k = isW ? Wc[it] : s.s_key[it - wsize];
                        if (isW && it == sel) k |= 1ull;
                        const uint64_t kk = k >> 1;
                        uint32_t cnt = 0;
    #pragma unroll 4
                        for (uint32_t j = 0; j < n_pad; j += 2) {
                            const ulonglong2 v = kp[j >> 1];
                            cnt += ((v.x >> 1) < kk) + ((v.y >> 1) < kk);
                        }
                        uint32_t rank = isW ? it + cnt : cnt;
                        if (!isW) {
                            uint32_t lo = 0, hi = wsize;
                            while (lo < hi) {
                                uint32_t mid = (lo + hi) >> 1;
                                if ((Wc[mid] >> 1) < kk) lo = mid + 1; else hi = mid;
                            }
                            rank += lo;
                        }
                        if (rank < ef_l) Wn[rank] = k;
                    }
                }
                if (filt_level) {
                    // R <- kf best of R ∪ {allowed new keys}: the same merge by rank, on the blanked copy of the new keys.
                    // Skipped (uniformly) when no allowed new key beats the current kf-th.
                    uint64_t *Rc = R0 + rcur * kfp, *Rn = R0 + (rcur ^ 1) * kfp;
                    const uint64_t thr = rsize == kf ? Rc[kf - 1] : ~0ull;
                    uint32_t n_in = 0;
                    for (uint32_t j = 0; j < n_new; j += 64) n_in += __popcll(__ballot(j + lane < n_new && s_keyR[j + lane] < thr));
                    if (n_in) {
                        const ulonglong2 *kr = reinterpret_cast<const ulonglong2 *>(s_keyR);
                        for (uint32_t it = tid; it < rsize + n_new; it += NW * 64) {
                            const bool isR = it < rsize;
                            const uint64_t k = isR ? Rc[it] : s_keyR[it - rsize];
                            if (k >= thr && !isR) continue; // disallowed, or cannot enter a full list
                            uint32_t cnt = 0;
    #pragma unroll 4
                            for (uint32_t j = 0; j < n_pad; j += 2) {
                                const ulonglong2 v = kr[j >> 1];
                                cnt += (v.x < k) + (v.y < k);
                            }
                            uint32_t rank = isR ? it + cnt : cnt;
                            if (!isR) {
                                uint32_t lo = 0, hi = rsize;
                                while (lo < hi) {
                                    uint32_t mid = (lo + hi) >> 1;
                                    if (Rc[mid] < k) lo = mid + 1; else hi = mid;
                                }
                                rank += lo;
                            }
                            if (rank < kf) Rn[rank] = k;
                        }
                        rsize = min(rsize + n_in, kf);
                        rcur ^= 1;
                    }
                }
    #ifdef LEANN_STAMPS
                const uint64_t stD = __builtin_amdgcn_s_memtime();
    #endif
                __syncthreads(); // B3
    #ifdef LEANN_STAMPS
                {
                    const uint64_t stE = __builtin_amdgcn_s_memtime();
                    stamp[0] += stB - st0; stamp[1] += stB1 - stB; stamp[2] += stC - stB1; stamp[3] += stB2 - stC;
                    stamp[4] += stD - stB2; stamp[5] += stE - stD; stamp[6] += 1;
                }
    #endif
                sel = uni(*next_slot);
                wsize = min(wsize + n_new, ef_l);
                cur ^= 1;
                hop++;
            }
        } else {
            // Per hop, two workgroup barriers:
            //   C  every wave: distances of the hop's unseen neighbours (rows in flight, wave-tree reductions)            | barrier
            //   E  wave 0: the NEXT candidate, known before the merge — the merged beam's first unexpanded entry is the smaller of the old
            //      beam's first unexpanded entry other than the one just expanded and the smallest new key; its index is the number of
            //      keys below it — then that node's adjacency list (one id per lane, lists hold <= 64) through the visited table:
            //      phase B of the next hop, into the other parity's buffers
            //   D  waves 1 .. NW-1, meanwhile: merge by rank into the other beam buffer                                    | barrier
            // The adjacency round trip of hop h + 1 thus runs under the merge of hop h.  Exactly one candidate is expanded per step, in
            // the sequential order: same entry, same index as a scan of the merged list would give.
            const uint32_t *const adj_base = lv == 0 ? adj0_p : adjU_p; // (as VALUES: an address-select between struct members puts the by-value GraphView in scratch)
            uint32_t e_pref = LEANN_EMPTY;
            auto phase_b = [&](uint32_t p, uint64_t ckey, uint32_t hidx) __attribute__((always_inline)) { // wave 0
                uint64_t *skey = s.s_key + p * kstride;
                uint32_t *snew = s.s_new + p * maxdeg;
                const uint32_t node = key_id(ckey);
                if (za->out_expanded && lv == (int)a.target_level && lane == 0 && hidx < za->exp_cap)
                    za->out_expanded[(size_t)qi * za->exp_cap + hidx] = ((ckey >> 32) << 32) | node;
                uint32_t n_new = 0;
                const bool ovf = (n_vis + deg > vis_limit);
                if (!ovf) {
                    const uint32_t e = e_pref; // (a hop redone after a table migration reads the same register again)
                    bool isnew = false;
                    if (e != LEANN_EMPTY) isnew = hbm ? vis_insert_hbm(gtab, gbits, gen, e) : vis_insert_lds(table, hbits, e);
                    const unsigned long long m = __ballot(isnew);
                    const uint32_t pos = __popcll(m & ((1ull << lane) - 1ull));
                    if (isnew) snew[pos] = e;
                    n_new = __popcll(m);
                }
                n_vis += n_new;
                n_evals += n_new;
                if (!ovf) { if (lv == 0) hops0++; else hopsU++; }
                if (lane < 8) skey[n_new + lane] = ~0ull; // sentinels: the merge scans keys 8 at a time
                if (filt_level && lane < 8) (s_keyR + p * kstride)[n_new + lane] = ~0ull;
                if (lane == 0) {
                    s.misc[8 + p] = n_new;
                    if (ovf) s.misc[3] = 1;
                }
            };
            if (wave == 0) {
                e_pref = (uint32_t)lane < deg ? adj_list(adj_base, upoff_p, deg, lv, key_id(best))[lane] : LEANN_EMPTY;
                phase_b(0u, best, 0u);
            }
            __syncthreads();
            uint32_t par = 0;
            for (;;) {
                uint64_t *Wc = W0 + cur * efp, *Wn = W0 + (cur ^ 1) * efp;
#ifdef LEANN_STAMPS // diagnostic build only (scripts/stamps.sh): where does a hop spend its cycles?
                const uint64_t st0 = __builtin_amdgcn_s_memtime();
#endif
                if (uni(s.misc[3])) {
                    // Table full: move the visited set one level up (LDS -> pool 1 -> pool 2), then wave 0 prepares this hop again.
                    const uint32_t next = hbm + 1;
                    if (next > 2 || (next == 2 && !za->gpool2)) { aborted = true; break; } // defensive: see the pool comment above
                    __syncthreads(); // every thread has read the flag before it is cleared
                    if (tid == 0) {
                        uint32_t *locks = next == 1 ? za->gpool_lock : za->gpool2_lock;
                        const uint32_t ntab = next == 1 ? za->gpool_tables : za->gpool2_tables;
                        uint32_t t = atomicAdd(&za->gpool_ctr[next == 1 ? 0 : 2], 1u), slot;
                        for (uint32_t i = 0;; i++) { // holders never wait for anything, so a table always comes free
                            slot = (t + i) % ntab;
                            if (atomicCAS(&locks[slot], 0u, 1u) == 0u) break;
                            if ((i % ntab) == ntab - 1) __builtin_amdgcn_s_sleep(32);
                        }
                        s.misc[4] = slot;
                        if (!hbm) s.misc[5] = atomicAdd(&za->gpool_ctr[1], 1u) + 1u; // pool 1 -> 2 keeps the generation
                        s.misc[3] = 0;
                    }
                    __syncthreads();
                    const uint32_t nslot = uni(s.misc[4]);
                    const uint32_t nbits = next == 1 ? za->gpool_bits : za->gpool2_bits;
                    unsigned long long *ntab_p = (next == 1 ? za->gpool : za->gpool2) + ((size_t)nslot << nbits);
                    if (!hbm) {
                        gen = uni(s.misc[5]);
                        for (uint32_t i = tid; i < hsize; i += NW * 64) {
                            uint32_t e = table[i];
                            if (e != LEANN_EMPTY) vis_insert_hbm(ntab_p, nbits, gen, e);
                        }
                    } else {
                        for (uint32_t i = tid; i < (1u << gbits); i += NW * 64) {
                            const unsigned long long cur_e = __hip_atomic_load(&gtab[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if ((uint32_t)(cur_e >> 32) == gen) vis_insert_hbm(ntab_p, nbits, gen, (uint32_t)cur_e);
                        }
                        __syncthreads(); // every probe of the old table has returned
                        if (tid == 0) atomicExch(&za->gpool_lock[gslot], 0u);
                    }
                    gtab = ntab_p;
                    gbits = nbits;
                    gslot = nslot;
                    hbm = next;
                    vis_limit = (1u << gbits) - (1u << (gbits - 2));
                    __syncthreads();
                    if (wave == 0) phase_b(par, Wc[sel], hop);
                    __syncthreads();
                    continue;
                }
                const uint32_t n_new = uni(s.misc[8 + par]);
                uint64_t *const skey = s.s_key + par * kstride;
                const uint32_t *const snew = s.s_new + par * maxdeg;
                uint64_t *const skeyR = s_keyR + par * kstride;
                // ---- phase C: stream the new rows, R in flight per wave ------------------------------
                if constexpr (G16) {
                    for (uint32_t jb = 0; jb < n_new; jb += NW * 4 * R) {
                        uint32_t ids[R], jj[R], abyte[R];
                        bool valid[R];
                        float dd[R];
#pragma unroll
                        for (int r = 0; r < R; r++) {
                            jj[r] = jb + (uint32_t)(r * NW + wave) * 4u + (uint32_t)(lane >> 4);
                            valid[r] = jj[r] < n_new;
                            ids[r] = valid[r] ? snew[jj[r]] : 0u;
                            abyte[r] = (filt_level && valid[r] && (lane & 15) == 15) ? allow[ids[r] >> 3] : 0u;
                        }
#ifndef LEANN_NO_ADJ_TOUCH
                        // adjacency lists of the rows being evaluated (see the row-per-wave form below): lanes 12, 13 of a row touch
                        uint32_t touch = 0;
                        if (lv == 0 && valid[0] && (lane & 14) == 12 && (uint32_t)(lane & 1) * 32u < deg)
                            touch = adj0_p[(size_t)ids[0] * deg + (uint32_t)(lane & 1) * 32u];
#endif
                        group_dist_rows_feat256<R>(qa, qb, reinterpret_cast<const char *>(g.X), g.row_bytes, ids, valid, lane, dd, g.norms);
#ifndef LEANN_NO_ADJ_TOUCH
                        asm volatile("" ::"v"(touch));
#endif
                        if ((lane & 15) == 15) {
#pragma unroll
                            for (int r = 0; r < R; r++)
                                if (valid[r]) {
                                    const uint64_t key = make_key(dd[r], ids[r]);
                                    skey[jj[r]] = key;
                                    if (filt_level) skeyR[jj[r]] = ((abyte[r] >> (ids[r] & 7u)) & 1u) ? key : ~0ull;
                                }
                        }
                    }
                } else
                for (uint32_t j0 = wave; j0 < n_new; j0 += NW * R) {
                    uint32_t ids[R];
                    float dd[R];
                    int nrows = 0;
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        uint32_t j = j0 + r * NW;
                        if (j < n_new) { ids[r] = snew[j]; nrows = r + 1; }
                    }
                    uint32_t abyte = 0, abit = 0;
                    if (filt_level && lane < R && j0 + lane * NW < n_new) { // lane r fetches row r's allow bit alongside the rows
                        const uint32_t e = snew[j0 + lane * NW];
                        abyte = allow[e >> 3];
                        abit = e & 7u;
                    }
#ifndef LEANN_NO_ADJ_TOUCH
                    // Latency form only: touch the adjacency lists of the rows being evaluated (two 128-B lines each, one lane per
                    // line).  The next candidates come from these rows, so when phase E picks one its list is an L2 hit instead of
                    // an HBM round trip at the head of the dependent chain; the chip is idle in this mode, the +8 % traffic is free.
                    // (An ordinary load issued BEFORE the row loads and consumed behind them: the rows' wait covers it, and the
                    // compiler keeps its register reserved until then.)
                    uint32_t touch = 0;
                    if (lv == 0 && (uint32_t)(lane >> 1) < (uint32_t)nrows && (uint32_t)(lane & 1) * 32u < deg)
                        touch = adj0_p[(size_t)snew[j0 + (uint32_t)(lane >> 1) * NW] * deg + (uint32_t)(lane & 1) * 32u];
#endif
                    if (FEAT) wave_dist_rows_feat<T, R>(q, reinterpret_cast<const char *>(g.X), g.row_bytes, g.feat_h, ids, nrows, lane, dd, g.norms);
                    else wave_dist_rows<T, R>(q, g.X, g.ld, ids, nrows, lane, dd);
#ifndef LEANN_NO_ADJ_TOUCH
                    asm volatile("" ::"v"(touch));
#endif
                    const unsigned long long amask = FILT ? __ballot((abyte >> abit) & 1u) : 0ull;
                    if (lane == 0) {
#pragma unroll
                        for (int r = 0; r < R; r++)
                            if (r < nrows) {
                                const uint64_t key = make_key(dd[r], ids[r]);
                                skey[j0 + r * NW] = key;
                                if (filt_level) skeyR[j0 + r * NW] = ((amask >> r) & 1ull) ? key : ~0ull;
                            }
                    }
                }
#ifdef LEANN_STAMPS
                const uint64_t stC = __builtin_amdgcn_s_memtime();
#endif
                __syncthreads(); // B2
#ifdef LEANN_STAMPS
                const uint64_t stB2 = __builtin_amdgcn_s_memtime();
#endif
                uint32_t *next_slot = &s.misc[1 + (hop & 1)];
                const uint32_t n_pad = (n_new + 7) & ~7u;
                if (wave == 0) {
                    // ---- phase E: next candidate; then phase B of the next hop ---------------------------------------------
                    uint32_t u = LEANN_EMPTY;
                    for (uint32_t base = 0; base < wsize && u == LEANN_EMPTY; base += 64) {
                        const uint32_t i = base + lane;
                        const unsigned long long m = __ballot(i < wsize && i != sel && !(Wc[i] & 1ull));
                        if (m) u = base + (uint32_t)__ffsll((long long)m) - 1u;
                    }
                    const uint64_t c_old = u != LEANN_EMPTY ? Wc[u] : ~0ull;
                    const uint64_t kj = (uint32_t)lane < n_new ? skey[lane] : ~0ull; // n_new <= 64: one key per lane
                    const uint32_t hi = wave_min_u32((uint32_t)(kj >> 32));
                    const uint32_t lo = wave_min_u32((uint32_t)(kj >> 32) == hi ? (uint32_t)kj : 0xFFFFFFFFu);
                    const uint64_t c_new = n_new ? (((uint64_t)hi << 32) | lo) : ~0ull;
                    uint32_t next = LEANN_EMPTY;
                    uint64_t ckey = 0;
                    if (c_old != ~0ull || c_new != ~0ull) {
                        uint32_t rank;
                        if ((c_old >> 1) < (c_new >> 1)) {
                            rank = u + (uint32_t)__popcll(__ballot((kj >> 1) < (c_old >> 1)));
                            ckey = c_old;
#ifdef LEANN_STAMPS
                            stamp[7] += 1;
#endif
                        } else {
                            rank = 0;
                            for (uint32_t base = 0; base < wsize; base += 64) {
                                const uint32_t i = base + lane;
                                rank += (uint32_t)__popcll(__ballot(i < wsize && (Wc[i] >> 1) < (c_new >> 1)));
                            }
                            ckey = c_new;
                        }
                        if (rank < ef_l) next = rank;
                    }
                    if (lane == 0) *next_slot = next;
                    if (next != LEANN_EMPTY) {
                        e_pref = (uint32_t)lane < deg ? adj_list(adj_base, upoff_p, deg, lv, key_id(ckey))[lane] : LEANN_EMPTY;
                        phase_b(par ^ 1u, ckey, hop + 1); // waits for the list while the other waves merge
                    }
                } else {
                    // ---- phase D (waves 1 .. NW-1): merge by rank into the other buffer ---------------------------------
                    // One work item per old entry (rank = index + #new keys below it) and per new key (rank = #old below it, by
                    // binary search, + #new below it); the scan over the new keys reads 16 B per LDS instruction, 8 keys per
                    // unrolled step, so the loads pipeline instead of paying one LDS latency per key.
                    const ulonglong2 *kp = reinterpret_cast<const ulonglong2 *>(skey);
                    for (uint32_t it = tid - 64; it < wsize + n_new; it += (NW - 1) * 64) {
                        const bool isW = it < wsize;
                        uint64_t k = isW ? Wc[it] : skey[it - wsize];
                        if (isW && it == sel) k |= 1ull;
                        const uint64_t kk = k >> 1;
                        uint32_t cnt = 0;
#pragma unroll 4
                        for (uint32_t j = 0; j < n_pad; j += 2) {
                            const ulonglong2 v = kp[j >> 1];
                            cnt += ((v.x >> 1) < kk) + ((v.y >> 1) < kk);
                        }
                        uint32_t rank = isW ? it + cnt : cnt;
                        if (!isW) {
                            uint32_t lo = 0, hi = wsize;
                            while (lo < hi) {
                                uint32_t mid = (lo + hi) >> 1;
                                if ((Wc[mid] >> 1) < kk) lo = mid + 1; else hi = mid;
                            }
                            rank += lo;
                        }
                        if (rank < ef_l) Wn[rank] = k;
                    }
                }
                if (filt_level) {
                    // R <- kf best of R ∪ {allowed new keys}: the same merge by rank, on the blanked copy of the new keys.
                    // Skipped (uniformly) when no allowed new key beats the current kf-th.
                    uint64_t *Rc = R0 + rcur * kfp, *Rn = R0 + (rcur ^ 1) * kfp;
                    const uint64_t thr = rsize == kf ? Rc[kf - 1] : ~0ull;
                    uint32_t n_in = 0;
                    for (uint32_t j = 0; j < n_new; j += 64) n_in += __popcll(__ballot(j + lane < n_new && skeyR[j + lane] < thr));
                    if (n_in) {
                        const ulonglong2 *kr = reinterpret_cast<const ulonglong2 *>(skeyR);
                        for (uint32_t it = tid; it < rsize + n_new; it += NW * 64) {
                            const bool isR = it < rsize;
                            const uint64_t k = isR ? Rc[it] : skeyR[it - rsize];
                            if (k >= thr && !isR) continue; // disallowed, or cannot enter a full list
                            uint32_t cnt = 0;
#pragma unroll 4
                            for (uint32_t j = 0; j < n_pad; j += 2) {
                                const ulonglong2 v = kr[j >> 1];
                                cnt += (v.x < k) + (v.y < k);
                            }
                            uint32_t rank = isR ? it + cnt : cnt;
                            if (!isR) {
                                uint32_t lo = 0, hi = rsize;
                                while (lo < hi) {
                                    uint32_t mid = (lo + hi) >> 1;
                                    if (Rc[mid] < k) lo = mid + 1; else hi = mid;
                                }
                                rank += lo;
                            }
                            if (rank < kf) Rn[rank] = k;
                        }
                        rsize = min(rsize + n_in, kf);
                        rcur ^= 1;
                    }
                }
#ifdef LEANN_STAMPS
                const uint64_t stD = __builtin_amdgcn_s_memtime();
#endif
                __syncthreads(); // B3: the merged beam and the next hop's unseen neighbours are in place
#ifdef LEANN_STAMPS
                {
                    const uint64_t stE = __builtin_amdgcn_s_memtime();
                    stamp[0] += 0; stamp[1] += 0; stamp[2] += stC - st0; stamp[3] += stB2 - stC;
                    stamp[4] += stD - stB2; stamp[5] += stE - stD; stamp[6] += 1;
                }
#endif
                sel = uni(*next_slot);
                wsize = min(wsize + n_new, ef_l);
                cur ^= 1;
                hop++;
                par ^= 1u;
                if (sel == LEANN_EMPTY) break;
            }
        }
        level_hops = hop;
        if (aborted) break;
        best = W0[cur * efp] & ~1ull;
        __syncthreads(); // everyone has read `best` before the next level rewrites W[0]
    }

    // ---- results ------------------------------------------------------------------------------
    __syncthreads();
    if (hbm && tid == 0) atomicExch(&(hbm == 1 ? za->gpool_lock : za->gpool2_lock)[gslot], 0u); // every probe of this workgroup has returned
    if (aborted) { // defensive (see the pool comment): the last table level filled up; empty result + stat 3 -> LEANN_ERR_OVERFLOW
        for (uint32_t t = tid; t < za->k; t += NW * 64) {
            za->out_keys[(size_t)qi * za->k + t] = 0xFFFFFFFFFFFFFFFFull;
            za->out_dists[(size_t)qi * za->k + t] = __uint_as_float(0x7F800000u);
        }
        if (tid == 0) {
            za->out_counts[qi] = 0;
            if (za->out_stats) {
                za->out_stats[(size_t)qi * 4 + 0] = n_evals;
                za->out_stats[(size_t)qi * 4 + 1] = hops0;
                za->out_stats[(size_t)qi * 4 + 2] = hopsU;
                za->out_stats[(size_t)qi * 4 + 3] = 3u;
            }
        }
        return;
    }
    const uint32_t nout = FILT ? rsize : min(wsize, za->k);
    uint64_t *Wc = FILT ? R0 + rcur * kfp : W0 + cur * efp;
    for (uint32_t t = tid; t < za->k; t += NW * 64) {
        size_t o = (size_t)qi * za->k + t;
        if (t < nout) {
            uint64_t k = Wc[t];
            za->out_keys[o] = (uint64_t)key_id(k) + za->key_offset;
            za->out_dists[o] = key_dist(k);
        } else {
            za->out_keys[o] = 0xFFFFFFFFFFFFFFFFull;
            za->out_dists[o] = __uint_as_float(0x7F800000u);
        }
    }
#ifdef LEANN_STAMPS
    if (tid == 0 && za->out_expanded == nullptr && za->exp_cap == 0xFEED) { // stamps go to a buffer of their own (passed via out_nexp)
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(za->out_nexp) + (size_t)qi * 8;
        for (int i = 0; i < 8; i++) dst[i] = stamp[i];
    }
#endif
    if (tid == 0) {
        za->out_counts[qi] = nout;
        if (za->out_nexp && za->exp_cap != 0xFEED) za->out_nexp[qi] = min(level_hops, za->exp_cap);
        if (za->out_stats) {
            za->out_stats[(size_t)qi * 4 + 0] = n_evals;
            za->out_stats[(size_t)qi * 4 + 1] = hops0;
            za->out_stats[(size_t)qi * 4 + 2] = hopsU;
            za->out_stats[(size_t)qi * 4 + 3] = hbm; // 0 LDS table only, 1 / 2 the visited set moved to pool 1 / 2
        }
    }
    __syncthreads();
}

// One workgroup per query.  BUILD only names the instantiation used by index construction (queries are
// base rows, k = ef) so that profiles keep query launches and construction launches apart.
// Narrow rows (<= 512 floats) are short of workgroups per CU, not of HBM (api.hip gives them the 16 KiB visited table): the compiler is
// asked for 6 workgroups per CU at 257..512 floats (80 registers instead of 83, no spill; 384-d 1.84 -> 1.95 M queries/s) and for 8 at
// <= 256 floats (64 registers, one spilled; 128-d 3.10 -> 3.64 M, 256-d 2.67 -> 2.80 M; scripts/exp/dims_sweep.py through scripts/variant.sh).
#ifndef LEANN_T2_OCC
#define LEANN_T2_OCC 6
#endif
#ifndef LEANN_T1_OCC
#define LEANN_T1_OCC 8
#endif
template <int T, int R, int NW, bool BUILD>
__global__ void __launch_bounds__(NW * 64, (NW == 4 && T == 2) ? LEANN_T2_OCC : (NW == 4 && T == 1) ? LEANN_T1_OCC : 1) beam_search_kernel(GraphView g, SearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t qi = blockIdx.x;
    if (qi >= a.nq) return;
    beam_search_one<T, R, NW, false>(g, a, qi, smem);
}
// filtered search (allow-bitmap): answers come from the R list
template <int T, int R, int NW>
__global__ void __launch_bounds__(NW * 64) beam_search_filtered_kernel(GraphView g, SearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t qi = blockIdx.x;
    if (qi >= a.nq) return;
    beam_search_one<T, R, NW, false, true>(g, a, qi, smem);
}
template <int T, int R, int NW>
__global__ void __launch_bounds__(NW * 64) beam_search_feat_filtered_kernel(GraphView g, SearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t qi = blockIdx.x;
    if (qi >= a.nq) return;
    beam_search_one<T, R, NW, true, true>(g, a, qi, smem);
}
// Throughput instantiation of the recompute-on search (4 waves per query, 520-B rows): the memory system delivers random 520-B rows at
// no more than ~3.9 TB/s however many are in flight (scripts/micro/gather_bw.hip: 4.8 TB/s of 128-B lines; 3 KiB rows reach 6.4), and the
// kernel gets closest to that with MANY queries per CU rather than many rows per wave: 8 workgroups per CU (the 16 KiB visited tables
// allow 8) at 5 rows in flight per wave — a typical hop has ~24 unseen neighbours = 6 per wave — measured 3.14 M queries/s against 2.91 M
// at 6 workgroups x 8 rows and 3.08 M at 7 x 6 (recompute10m_graph, ef = 56; scripts/exp/feat_occupancy.sh).
#ifndef LEANN_FEAT_OCC
#define LEANN_FEAT_OCC 8
#endif
#ifndef LEANN_FEAT_R1
#define LEANN_FEAT_R1 5
#endif
// recompute-on rows of exactly 256 features: four rows per wave instruction, G groups in flight per wave (G16 above).  Throughput
// shape = 7 workgroups per CU x 1 group (72 registers, no spills; 16 rows per workgroup and round): 4.5 M queries/s at ef = 52 =
// 7.9 G random rows/s, against the 10 G rows/s scripts/micro/gather_bw.hip measures as the memory system's ceiling for this access
// shape.  6 workgroups x 2 groups (a typical hop's ~26 unseen neighbours in ONE round) is 2 % behind, 8 workgroups (64 registers) spill:
// 3.7-3.9 M, 6 x 3 groups 3.5 M (scripts/exp/feat256_shape.sh).
#ifndef LEANN_FEAT_G
#define LEANN_FEAT_G 1
#endif
#ifndef LEANN_FEAT256_OCC
#define LEANN_FEAT256_OCC 7
#endif
template <int G, int NW>
__global__ void __launch_bounds__(NW * 64, NW == 4 ? LEANN_FEAT256_OCC : 1) beam_search_feat256_kernel(GraphView g, SearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t qi = blockIdx.x;
    if (qi >= a.nq) return;
    beam_search_one<1, G, NW, true, false, true>(g, a, qi, smem);
}
template <int G, int NW>
__global__ void __launch_bounds__(NW * 64) beam_search_feat256_filtered_kernel(GraphView g, SearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t qi = blockIdx.x;
    if (qi >= a.nq) return;
    beam_search_one<1, G, NW, true, true, true>(g, a, qi, smem);
}
// recompute-on instantiation: rows are bf16 features + inline norm, queries are W q
template <int T, int R, int NW>
__global__ void __launch_bounds__(NW * 64, (NW == 4 && T == 1) ? LEANN_FEAT_OCC : 1) beam_search_feat_kernel(GraphView g, SearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t qi = blockIdx.x;
    if (qi >= a.nq) return;
    beam_search_one<T, R, NW, true>(g, a, qi, smem);
}
