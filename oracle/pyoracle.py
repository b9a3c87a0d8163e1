"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY (see oracle/oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)
f32p = C.POINTER(C.c_float)


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE], stdout=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.orc_gauss.restype = C.c_float
        L.orc_gauss.argtypes = [C.c_uint64] * 3
        L.orc_hash3.restype = C.c_uint64
        L.orc_hash3.argtypes = [C.c_uint64] * 3
        L.orc_gen_rows.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float,
                                   C.c_uint32, C.c_uint64, C.c_uint64, f32p]
        L.orc_set_fast_dot.argtypes = [C.c_int]
        L.orc_set_vamana_two_stage.argtypes = [C.c_int]
        for n in ("orc_dot_canon", "orc_dot_canon_ref", "orc_dot_seq", "orc_dot_seqfma", "orc_dot_fast"):
            f = getattr(L, n)
            f.restype = C.c_float
            f.argtypes = [f32p, f32p, C.c_uint32]
        L.orc_scan_topk.argtypes = [f32p, C.c_uint64, C.c_uint32, f32p, C.c_uint32, C.c_int, u8p,
                                    u64p, f32p, u32p]
        L.orc_level.restype = C.c_uint32
        L.orc_level.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]
        L.orc_hnsw_build.restype = C.c_void_p
        L.orc_hnsw_build.argtypes = [f32p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64]
        L.orc_vamana_build.restype = C.c_void_p
        L.orc_vamana_build.argtypes = [f32p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                       C.c_float, C.c_uint64]
        L.orc_graph_from_arrays.restype = C.c_void_p
        L.orc_graph_from_arrays.argtypes = [f32p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                            C.c_uint32, C.c_uint32, C.c_uint32, u8p, u32p, u32p,
                                            u32p, C.c_uint64]
        L.orc_graph_info.argtypes = [C.c_void_p, u64p]
        L.orc_graph_set_features.argtypes = [C.c_void_p, u8p, C.c_uint32, C.c_uint32]
        L.orc_project_query.argtypes = [C.POINTER(C.c_uint16), C.c_uint32, C.c_uint32, C.c_uint32, f32p, f32p]
        L.orc_graph_export.argtypes = [C.c_void_p, u8p, u32p, u32p, u32p]
        L.orc_graph_free.argtypes = [C.c_void_p]
        L.orc_graph_search.argtypes = [C.c_void_p, f32p, C.c_uint32, C.c_uint32, C.c_int, u64p,
                                       f32p, u32p, u64p]
        L.orc_graph_search_batch.argtypes = [C.c_void_p, f32p, C.c_uint64, C.c_uint32, C.c_uint32,
                                             C.c_int, C.c_uint32, u64p, f32p, u32p, u64p]
        L.orc_graph_search_filtered_batch.argtypes = [C.c_void_p, f32p, C.c_uint64, C.c_uint32, C.c_uint32,
                                                      C.c_int, C.c_uint32, u8p, C.c_uint64, u64p, f32p, u32p, u64p]
        L.orc_merge_topk.argtypes = [u64p, f32p, u32p, C.c_uint32, C.c_uint32, C.c_uint32, u64p,
                                     f32p, u32p]
        L.orc_hybrid_rerank.argtypes = [u64p, f32p, C.c_uint32, f32p, C.c_uint64, C.c_float, u64p,
                                        f32p]
        L.orc_l2_normalize.argtypes = [f32p, C.c_uint32]
        u16p = C.POINTER(C.c_uint16)
        L.orc_synth_features.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_uint32, C.c_uint64,
                                         C.c_uint64, u16p]
        L.orc_synth_weights.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, u16p]
        L.orc_recompute_encode.argtypes = [u16p, C.c_uint64, C.c_uint32, u16p, C.c_uint32, f32p]
        L.orc_recompute_encode_pooled.argtypes = [u16p, u8p, C.c_uint64, C.c_uint32, C.c_uint32, u16p, C.c_uint32, f32p]
        _LIB = L
    return _LIB


# Default synthetic-set parameters; shared with leann-rs_amd/synth.py by value.
SEED_CORPUS = 0x5EED0001
SEED_LEVELS = 0x5EED0003


def gen_rows(seed, d, r, n_clusters, sigma, stream, i0, n):
    out = np.empty((n, d), np.float32)
    lib().orc_gen_rows(seed, d, r, n_clusters, sigma, stream, i0, n, _p(out, f32p))
    return out


def dot(a, b, kind="canon"):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return float(getattr(lib(), "orc_dot_" + kind)(_p(a, f32p), _p(b, f32p), a.shape[0]))


def scan_topk(X, q, k, mode=0, allow_mask=None):
    X = np.ascontiguousarray(X, np.float32)
    q = np.ascontiguousarray(q, np.float32)
    keys = np.zeros(k, np.uint64)
    scores = np.zeros(k, np.float32)
    n = C.c_uint32(0)
    lib().orc_scan_topk(_p(X, f32p), X.shape[0], X.shape[1], _p(q, f32p), k, mode,
                        _p(allow_mask, u8p), _p(keys, u64p), _p(scores, f32p), C.byref(n))
    return keys[: n.value], scores[: n.value]


class Graph:
    """Flat graph (levels, upper_off, adj0, adjU) over borrowed f32 rows."""

    def __init__(self, handle, X, keep=()):
        self.h = handle
        self.X = X
        self._keep = keep
        info = np.zeros(8, np.uint64)
        lib().orc_graph_info(self.h, _p(info, u64p))
        (self.n, self.d, self.ld, self.M, self.M0, self.max_level, self.entry,
         self.n_upper_lists) = [int(v) for v in info]

    @classmethod
    def build_hnsw(cls, X, M=32, efc=64, level_seed=SEED_LEVELS):
        X = np.ascontiguousarray(X, np.float32)
        h = lib().orc_hnsw_build(_p(X, f32p), X.shape[0], X.shape[1], M, efc, level_seed)
        return cls(h, X)

    @classmethod
    def build_vamana(cls, X, R=32, L=64, alpha=1.2, seed=SEED_LEVELS, two_stage=True):
        """two_stage: RobustPrune as DiskANN implements it (occlude_list: alpha 1.0 over the whole pool, then the relaxed alpha for the
        free slots) — what the GPU builder does; False = the paper's one-stage Alg. 2 (the round-2 pins were made with it)."""
        X = np.ascontiguousarray(X, np.float32)
        lib().orc_set_vamana_two_stage(1 if two_stage else 0)
        try:
            h = lib().orc_vamana_build(_p(X, f32p), X.shape[0], X.shape[1], R, L, alpha, seed)
        finally:
            lib().orc_set_vamana_two_stage(1)
        return cls(h, X)

    @classmethod
    def from_arrays(cls, X, M, M0, max_level, entry, levels, upper_off, adj0, adjU):
        X = np.ascontiguousarray(X, np.float32)
        levels = np.ascontiguousarray(levels, np.uint8)
        upper_off = np.ascontiguousarray(upper_off, np.uint32)
        adj0 = np.ascontiguousarray(adj0, np.uint32)
        adjU = np.ascontiguousarray(adjU, np.uint32)
        if adjU.size == 0:
            adjU = np.full(max(M, 1), 0xFFFFFFFF, np.uint32)
            nul = 0
        else:
            nul = adjU.size // M
        h = lib().orc_graph_from_arrays(_p(X, f32p), X.shape[0], X.shape[1], X.shape[1], M, M0,
                                        max_level, entry, _p(levels, u8p), _p(upper_off, u32p),
                                        _p(adj0, u32p), _p(adjU, u32p), nul)
        return cls(h, X, keep=(levels, upper_off, adj0, adjU))

    def set_features(self, rows_bytes, feat_h, row_bytes):
        """recompute-on mode: rows_bytes = uint8 [n, row_bytes]; queries must then be projected (project_queries)"""
        self._feat = np.ascontiguousarray(rows_bytes, np.uint8)
        lib().orc_graph_set_features(self.h, _p(self._feat, u8p), feat_h, row_bytes)
        self.d = feat_h

    def export(self):
        levels = np.zeros(self.n, np.uint8)
        upper_off = np.zeros(self.n, np.uint32)
        adj0 = np.zeros((self.n, self.M0), np.uint32)
        adjU = np.zeros((max(self.n_upper_lists, 1), self.M), np.uint32)
        lib().orc_graph_export(self.h, _p(levels, u8p), _p(upper_off, u32p), _p(adj0, u32p),
                               _p(adjU, u32p))
        return levels, upper_off, adj0, adjU[: self.n_upper_lists]

    def search(self, q, k, ef, algo=0):
        q = np.ascontiguousarray(q, np.float32)
        keys = np.zeros(k, np.uint64)
        dists = np.zeros(k, np.float32)
        n = C.c_uint32(0)
        stats = np.zeros(3, np.uint64)
        lib().orc_graph_search(self.h, _p(q, f32p), k, ef, algo, _p(keys, u64p), _p(dists, f32p),
                               C.byref(n), _p(stats, u64p))
        return keys[: n.value], dists[: n.value], stats

    def search_batch(self, Q, k, ef, algo=0, nthreads=1):
        Q = np.ascontiguousarray(Q, np.float32)
        nq = Q.shape[0]
        keys = np.full((nq, k), 0xFFFFFFFFFFFFFFFF, np.uint64)
        dists = np.full((nq, k), np.inf, np.float32)
        counts = np.zeros(nq, np.uint32)
        stats = np.zeros((nq, 3), np.uint64)
        lib().orc_graph_search_batch(self.h, _p(Q, f32p), nq, k, ef, algo, nthreads, _p(keys, u64p),
                                     _p(dists, f32p), _p(counts, u32p), _p(stats, u64p))
        return keys, dists, counts, stats

    def search_filtered_batch(self, Q, k, ef, allow, algo=0, nthreads=1):
        """allow: uint8 bitmap [ceil(n/8)] shared by all queries, or [nq, stride] one per query."""
        Q = np.ascontiguousarray(Q, np.float32)
        allow = np.ascontiguousarray(allow, np.uint8)
        nq = Q.shape[0]
        stride = 0 if allow.ndim == 1 else allow.shape[1]
        keys = np.full((nq, k), 0xFFFFFFFFFFFFFFFF, np.uint64)
        dists = np.full((nq, k), np.inf, np.float32)
        counts = np.zeros(nq, np.uint32)
        stats = np.zeros((nq, 3), np.uint64)
        lib().orc_graph_search_filtered_batch(self.h, _p(Q, f32p), nq, k, ef, algo, nthreads, _p(allow, u8p), stride,
                                              _p(keys, u64p), _p(dists, f32p), _p(counts, u32p), _p(stats, u64p))
        return keys, dists, counts, stats

    def __del__(self):
        try:
            lib().orc_graph_free(self.h)
        except Exception:
            pass


def synth_features(seed, h, n_clusters, sigma, stream, i0, n, r_int=0):
    out = np.empty((n, h), np.uint16)
    lib().orc_synth_features(seed, h, r_int, n_clusters, sigma, stream, i0, n, out.ctypes.data_as(C.POINTER(C.c_uint16)))
    return out


def synth_weights(seed, h, d):
    out = np.empty((h, d), np.uint16)
    lib().orc_synth_weights(seed, h, d, out.ctypes.data_as(C.POINTER(C.c_uint16)))
    return out


def recompute_encode(F, W):
    F = np.ascontiguousarray(F, np.uint16)
    W = np.ascontiguousarray(W, np.uint16)
    out = np.empty((F.shape[0], W.shape[1]), np.float32)
    u16p = C.POINTER(C.c_uint16)
    lib().orc_recompute_encode(F.ctypes.data_as(u16p), F.shape[0], F.shape[1], W.ctypes.data_as(u16p), W.shape[1],
                               _p(out, f32p))
    return out


def recompute_encode_pooled(F, mask, W, L):
    """F: [n*L, h] token features, mask: [n, L] or None"""
    F = np.ascontiguousarray(F, np.uint16)
    W = np.ascontiguousarray(W, np.uint16)
    n = F.shape[0] // L
    out = np.empty((n, W.shape[1]), np.float32)
    u16p = C.POINTER(C.c_uint16)
    m = np.ascontiguousarray(mask, np.uint8) if mask is not None else None
    lib().orc_recompute_encode_pooled(F.ctypes.data_as(u16p), _p(m, u8p), n, L, F.shape[1], W.ctypes.data_as(u16p),
                                      W.shape[1], _p(out, f32p))
    return out


def project_queries(W, Q, hp4):
    W = np.ascontiguousarray(W, np.uint16)
    Q = np.ascontiguousarray(Q, np.float32)
    G = np.zeros((Q.shape[0], hp4), np.float32)
    for i in range(Q.shape[0]):
        lib().orc_project_query(W.ctypes.data_as(C.POINTER(C.c_uint16)), W.shape[0], hp4, W.shape[1], _p(Q[i], f32p), _p(G[i], f32p))
    return G


def merge_topk(keys, dists, counts, k_out):
    keys = np.ascontiguousarray(keys, np.uint64)
    dists = np.ascontiguousarray(dists, np.float32)
    counts = np.ascontiguousarray(counts, np.uint32)
    S, k_in = keys.shape
    ok = np.zeros(k_out, np.uint64)
    od = np.zeros(k_out, np.float32)
    n = C.c_uint32(0)
    lib().orc_merge_topk(_p(keys, u64p), _p(dists, f32p), _p(counts, u32p), S, k_in, k_out,
                         _p(ok, u64p), _p(od, f32p), C.byref(n))
    return ok[: n.value], od[: n.value]


def hybrid_rerank(vector_results, bm25_scores, alpha):
    """src/index/bm25.rs:135-170"""
    idx = np.array([i for i, _ in vector_results], np.uint64)
    vs = np.array([s for _, s in vector_results], np.float32)
    b = np.ascontiguousarray(bm25_scores, np.float32)
    oi = np.zeros(len(idx), np.uint64)
    os_ = np.zeros(len(idx), np.float32)
    lib().orc_hybrid_rerank(_p(idx, u64p), _p(vs, f32p), len(idx), _p(b, f32p), len(b), alpha,
                            _p(oi, u64p), _p(os_, f32p))
    return [(int(i), float(s)) for i, s in zip(oi, os_)]


def exact_topk(X, Q, k):
    """Ground truth for recall: exact IP top-k in float64 (numpy), ties -> lower id."""
    S = Q.astype(np.float64) @ X.astype(np.float64).T
    idx = np.argsort(-S, axis=1, kind="stable")[:, :k]
    return idx
