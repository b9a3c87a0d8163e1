"""Host-side mirror of leann-rs's backend layer (src/backend/{mod,traits,hnsw,diskann}.rs) over the
C ABI.  Same names, argument meaning and error behaviour; the arithmetic runs in HIP kernels."""
import ctypes as C
import enum
import os

import numpy as np

from . import _native as N
from ._native import LeannError, f32p, u32p, u64p, u8p


class BackendType(enum.IntEnum):
    """enum BackendType { Hnsw, DiskAnn } — src/backend/mod.rs:15-19"""
    Hnsw = 0
    DiskAnn = 1

    @classmethod
    def from_name(cls, name):
        # src/index/searcher.rs:95-99
        if name == "hnsw":
            return cls.Hnsw
        if name == "diskann":
            return cls.DiskAnn
        raise LeannError(1, f"Unknown backend: {name}")

    def load_searcher(self, index_path, dimensions, device=0):
        """BackendType::load_searcher — src/backend/mod.rs:23-45"""
        return BackendSearcher.load(self, index_path, dimensions, device)


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


class BackendSearcher:
    """trait BackendSearcher — src/backend/traits.rs:11-30 (HnswSearcher / DiskAnnSearcher)."""

    def __init__(self, handle, backend_type):
        self._h = handle
        self.backend_type = BackendType(backend_type)

    # -- constructors -------------------------------------------------------------------------
    @classmethod
    def load(cls, backend_type, index_path, dimensions, device=0):
        """HnswSearcher::load (hnsw.rs:18-75) / DiskAnnSearcher::load (diskann.rs:21-43).  `device`: an ordinal, or a list / range
        ("0,1,2,3", "0-7") for a sharded index behind the same handle."""
        h = C.c_void_p()
        N.check(N.lib().leann_backend_open(os.fsencode(str(index_path)), int(backend_type), dimensions,
                                           str(device).encode(), C.byref(h)))
        return cls(h, backend_type)

    @classmethod
    def from_arrays(cls, backend_type, vectors, M, M0, max_level, entry, levels, upper_off, adj0, adjU,
                    device=0, key_offset=0):
        vectors = np.ascontiguousarray(vectors, np.float32)
        levels = np.ascontiguousarray(levels, np.uint8)
        upper_off = np.ascontiguousarray(upper_off, np.uint32)
        adj0 = np.ascontiguousarray(adj0, np.uint32)
        adjU = np.ascontiguousarray(adjU, np.uint32)
        nul = adjU.size // M if adjU.size else 0
        h = C.c_void_p()
        n, d = vectors.shape
        N.check(N.lib().leann_backend_from_arrays(
            int(backend_type), _p(vectors, f32p), n, d, M, M0, max_level, entry, _p(levels, u8p),
            _p(upper_off, u32p), _p(adj0, u32p), _p(adjU, u32p) if nul else None, nul, device,
            key_offset, C.byref(h)))
        return cls(h, backend_type)

    @classmethod
    def build_device(cls, backend_type, d_vectors_ptr, n, dims, ld, graph_degree, complexity, device=0,
                     key_offset=0, take_copy=False):
        """Index straight from rows already in HBM (additive API)."""
        h = C.c_void_p()
        N.check(N.lib().leann_backend_build_device(int(backend_type), d_vectors_ptr, n, dims, ld,
                                                   graph_degree, complexity, device, key_offset,
                                                   1 if take_copy else 0, C.byref(h)))
        return cls(h, backend_type)

    # -- trait surface ------------------------------------------------------------------------
    def search(self, query, top_k, complexity):
        """fn search(&self, query, top_k, complexity) -> (Vec<u64>, Vec<f32>) — traits.rs:16-21"""
        q = np.ascontiguousarray(query, np.float32).reshape(-1)
        if q.shape[0] != self.dims():
            raise LeannError(1, f"query has {q.shape[0]} dimensions, index has {self.dims()}")
        keys = np.zeros(max(top_k, 1), np.uint64)
        dists = np.zeros(max(top_k, 1), np.float32)
        n = C.c_size_t(0)
        N.check(N.lib().leann_backend_search(self._h, _p(q, f32p), top_k, complexity, _p(keys, u64p),
                                             _p(dists, f32p), C.byref(n)))
        return keys[: n.value].copy(), dists[: n.value].copy()

    def search_batch(self, queries, top_k, complexity):
        Q = np.ascontiguousarray(queries, np.float32)
        if Q.ndim != 2 or Q.shape[1] != self.dims():
            raise LeannError(1, f"queries must be [nq x {self.dims()}]")
        nq = Q.shape[0]
        keys = np.full((nq, top_k), np.iinfo(np.uint64).max, np.uint64)
        dists = np.full((nq, top_k), np.inf, np.float32)
        counts = np.zeros(nq, np.uint32)
        N.check(N.lib().leann_backend_search_batch(self._h, _p(Q, f32p), nq, top_k, complexity,
                                                   _p(keys, u64p), _p(dists, f32p), _p(counts, u32p)))
        return keys, dists, counts

    def search_filtered(self, query, top_k, complexity, allow):
        """search restricted to the positions whose bit is set in `allow` (uint8 bitmap, ceil(len/8) bytes);
        the filter runs inside the traversal (replaces searcher.rs:129-133 over-fetch)."""
        q = np.ascontiguousarray(query, np.float32).reshape(-1)
        if q.shape[0] != self.dims():
            raise LeannError(1, f"query has {q.shape[0]} dimensions, index has {self.dims()}")
        allow = np.ascontiguousarray(allow, np.uint8).reshape(-1)
        if allow.shape[0] < (self.len() + 7) // 8:
            raise LeannError(1, f"allow-bitmap has {allow.shape[0]} bytes, index needs {(self.len() + 7) // 8}")
        keys = np.zeros(max(top_k, 1), np.uint64)
        dists = np.zeros(max(top_k, 1), np.float32)
        n = C.c_size_t(0)
        N.check(N.lib().leann_backend_search_filtered(self._h, _p(q, f32p), top_k, complexity, _p(allow, u8p),
                                                      _p(keys, u64p), _p(dists, f32p), C.byref(n)))
        return keys[: n.value].copy(), dists[: n.value].copy()

    def search_filtered_batch(self, queries, top_k, complexity, allow):
        """allow: [ceil(len/8)] shared bitmap or [nq, stride] one bitmap per query"""
        Q = np.ascontiguousarray(queries, np.float32)
        if Q.ndim != 2 or Q.shape[1] != self.dims():
            raise LeannError(1, f"queries must be [nq x {self.dims()}]")
        allow = np.ascontiguousarray(allow, np.uint8)
        stride = 0 if allow.ndim == 1 else allow.shape[1]
        if allow.shape[-1] < (self.len() + 7) // 8 or (allow.ndim == 2 and allow.shape[0] != Q.shape[0]):
            raise LeannError(1, "allow-bitmap shape does not match the index / the batch")
        nq = Q.shape[0]
        keys = np.full((nq, top_k), np.iinfo(np.uint64).max, np.uint64)
        dists = np.full((nq, top_k), np.inf, np.float32)
        counts = np.zeros(nq, np.uint32)
        N.check(N.lib().leann_backend_search_filtered_batch(self._h, _p(Q, f32p), nq, top_k, complexity, _p(allow, u8p),
                                                            stride, _p(keys, u64p), _p(dists, f32p), _p(counts, u32p)))
        return keys, dists, counts

    def search_filtered_exact_batch(self, queries, top_k, allow):
        """exact answer for a selective filter: the allowed rows are compacted and scanned (no graph).  allow as in
        search_filtered_batch."""
        Q = np.ascontiguousarray(queries, np.float32)
        if Q.ndim != 2 or Q.shape[1] != self.dims():
            raise LeannError(1, f"queries must be [nq x {self.dims()}]")
        allow = np.ascontiguousarray(allow, np.uint8)
        stride = 0 if allow.ndim == 1 else allow.shape[1]
        if allow.shape[-1] < (self.len() + 7) // 8 or (allow.ndim == 2 and allow.shape[0] != Q.shape[0]):
            raise LeannError(1, "allow-bitmap shape does not match the index / the batch")
        nq = Q.shape[0]
        keys = np.full((nq, top_k), np.iinfo(np.uint64).max, np.uint64)
        dists = np.full((nq, top_k), np.inf, np.float32)
        counts = np.zeros(nq, np.uint32)
        N.check(N.lib().leann_backend_search_filtered_exact_batch(self._h, _p(Q, f32p), nq, top_k, _p(allow, u8p), stride,
                                                                  _p(keys, u64p), _p(dists, f32p), _p(counts, u32p)))
        return keys, dists, counts

    def search_filtered_exact_batch_device(self, d_queries, nq, top_k, d_allow, allow_stride, d_keys, d_dists, d_counts,
                                           stream=None):
        N.check(N.lib().leann_backend_search_filtered_exact_batch_device(self._h, d_queries, nq, top_k, d_allow, allow_stride,
                                                                         d_keys, d_dists, d_counts, stream))

    def register_filter(self, allow):
        """upload + compact an allow-bitmap once; returns a Filter for search_filter_batch (close() it when done)"""
        allow = np.ascontiguousarray(allow, np.uint8)
        if allow.ndim != 1 or allow.shape[0] < (self.len() + 7) // 8:
            raise LeannError(1, "allow-bitmap shape does not match the index")
        h = C.c_void_p()
        N.check(N.lib().leann_backend_filter_create(self._h, _p(allow, u8p), C.byref(h)))
        return Filter(h)

    def search_filter_batch(self, queries, top_k, complexity, flt, mode="auto"):
        """nq queries under a registered filter; mode: "walk" | "exact" | "auto" (the library chooses by the filter's selectivity)"""
        Q = np.ascontiguousarray(queries, np.float32)
        if Q.ndim != 2 or Q.shape[1] != self.dims():
            raise LeannError(1, f"queries must be [nq x {self.dims()}]")
        nq = Q.shape[0]
        keys = np.full((nq, top_k), np.iinfo(np.uint64).max, np.uint64)
        dists = np.full((nq, top_k), np.inf, np.float32)
        counts = np.zeros(nq, np.uint32)
        N.check(N.lib().leann_backend_search_filter_batch(self._h, _p(Q, f32p), nq, top_k, complexity, flt._h,
                                                          {"walk": 0, "exact": 1, "auto": 2}[mode], _p(keys, u64p), _p(dists, f32p),
                                                          _p(counts, u32p)))
        return keys, dists, counts

    def search_batch_device(self, d_queries, nq, top_k, complexity, d_keys, d_dists, d_counts,
                            d_stats=None, stream=None):
        N.check(N.lib().leann_backend_search_batch_device(self._h, d_queries, nq, top_k, complexity,
                                                          d_keys, d_dists, d_counts, d_stats, stream))

    def search_filtered_batch_device(self, d_queries, nq, top_k, complexity, d_allow, allow_stride, d_keys, d_dists,
                                     d_counts, d_stats=None, stream=None):
        N.check(N.lib().leann_backend_search_filtered_batch_device(self._h, d_queries, nq, top_k, complexity, d_allow,
                                                                   allow_stride, d_keys, d_dists, d_counts, d_stats, stream))

    def set_coalescing(self, wait_us=200, max_batch=4096):
        """gather concurrent single-query search() callers into one batched launch ((0, 0) disables)"""
        N.check(N.lib().leann_backend_set_coalescing(self._h, wait_us, max_batch))

    def coalescing_stats(self):
        a, b = C.c_uint64(0), C.c_uint64(0)
        N.check(N.lib().leann_backend_coalescing_stats(self._h, C.byref(a), C.byref(b)))
        return {"launches": a.value, "queries": b.value}

    def len(self):
        return int(N.lib().leann_backend_len(self._h))

    __len__ = len

    def is_empty(self):
        return self.len() == 0

    def dims(self):
        return int(N.lib().leann_backend_dims(self._h))

    # -- extras -------------------------------------------------------------------------------
    def stats(self, reset=False):
        s = N.SearchStats()
        N.check(N.lib().leann_backend_stats(self._h, C.byref(s), 1 if reset else 0))
        return {k: int(getattr(s, k)) for k, _ in N.SearchStats._fields_}

    def graph_info(self):
        info = np.zeros(8, np.uint64)
        N.check(N.lib().leann_backend_graph_info(self._h, _p(info, u64p)))
        names = ("n", "dims", "ld", "M", "M0", "max_level", "entry", "n_upper_lists")
        return {k: int(v) for k, v in zip(names, info)}

    def graph_export(self, with_vectors=False):
        gi = self.graph_info()
        n = gi["n"]
        levels = np.zeros(n, np.uint8)
        upper_off = np.zeros(n, np.uint32)
        adj0 = np.zeros((n, gi["M0"]), np.uint32)
        adjU = np.zeros((max(gi["n_upper_lists"], 1), gi["M"]), np.uint32)
        X = np.zeros((n, gi["dims"]), np.float32) if with_vectors else None
        N.check(N.lib().leann_backend_graph_export(self._h, _p(levels, u8p), _p(upper_off, u32p),
                                                   _p(adj0, u32p), _p(adjU, u32p), _p(X, f32p)))
        return dict(gi, levels=levels, upper_off=upper_off, adj0=adj0, adjU=adjU[: gi["n_upper_lists"]],
                    vectors=X)

    def n_shards(self):
        """0 for a plain handle; G for a composite (sharded) one"""
        return int(N.lib().leann_backend_shard_count(self._h))

    def shard(self, g):
        """the sub-index of shard g as an ordinary searcher (borrowed: valid while this handle lives; never closed by itself)"""
        h = C.c_void_p()
        N.check(N.lib().leann_backend_shard(self._h, g, C.byref(h)))
        s = BackendSearcher(h, self.backend_type)
        s._borrowed = True
        return s

    def device_rows_ptr(self):
        return N.lib().leann_backend_device_rows(self._h)

    def save(self, index_path):
        N.check(N.lib().leann_backend_save(self._h, os.fsencode(str(index_path))))

    def close(self):
        if self._h:
            if not getattr(self, "_borrowed", False):
                N.lib().leann_backend_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HnswSearcher(BackendSearcher):
    """src/backend/hnsw.rs:12-14"""

    @classmethod
    def load(cls, index_path, dimensions, device=0):
        return BackendSearcher.load.__func__(cls, BackendType.Hnsw, index_path, dimensions, device)


class DiskAnnSearcher(BackendSearcher):
    """src/backend/diskann.rs:15-17"""

    @classmethod
    def load(cls, index_path, dimensions, device=0):
        return BackendSearcher.load.__func__(cls, BackendType.DiskAnn, index_path, dimensions, device)


class BackendBuilder:
    """struct BackendBuilder — src/backend/traits.rs:6-8, impl src/backend/mod.rs:48-101"""

    def __init__(self, backend_type):
        self.backend_type = BackendType(backend_type)

    def build(self, embeddings, ids, index_path, dimensions, graph_degree, complexity):
        """mod.rs:55-79 -> hnsw.rs:96-139 / diskann.rs:70-105.  `ids` is unused, as in the reference."""
        X = np.ascontiguousarray(embeddings, np.float32)
        if X.ndim != 2 or X.shape[1] != dimensions:
            raise LeannError(1, f"embeddings must be [n x {dimensions}]")
        N.check(N.lib().leann_backend_build(int(self.backend_type), _p(X, f32p), X.shape[0], dimensions,
                                            graph_degree, complexity, os.fsencode(str(index_path))))

    def add_to_index(self, embeddings, index_path, dimensions, start_id):
        """mod.rs:82-100 -> hnsw.rs:142-191; DiskANN refuses (mod.rs:93-98)."""
        X = np.ascontiguousarray(embeddings, np.float32)
        N.check(N.lib().leann_backend_add(int(self.backend_type), _p(X, f32p), X.shape[0], dimensions,
                                          start_id, os.fsencode(str(index_path))))



class Filter:
    """a metadata filter registered on the device (leann_backend_filter_create)"""

    def __init__(self, handle):
        self._h = handle

    def count(self):
        return int(N.lib().leann_backend_filter_count(self._h))

    def close(self):
        if self._h:
            N.lib().leann_backend_filter_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedIndex:
    """leann_sharded (include/leann_backend.h "sharded indexes", csrc/shard.hip): the corpus partitioned into contiguous position
    ranges, one sub-index per shard; a search = per-shard traversal + gather of the per-shard lists + merge kernel.  One process
    over several devices (open / build_device / from_searchers) or one process per GPU over RCCL (attach)."""

    def __init__(self, handle, keep=()):
        self._h = handle
        self._keep = keep  # objects whose device memory the shards borrow

    @classmethod
    def open(cls, backend_type, index_path, dimensions, device_spec):
        h = C.c_void_p()
        N.check(N.lib().leann_sharded_open(os.fsencode(str(index_path)), int(backend_type), dimensions, str(device_spec).encode(), C.byref(h)))
        return cls(h)

    @classmethod
    def build_device(cls, backend_type, d_ptrs, rows, dims, ld, graph_degree, complexity, devices, keep=()):
        G = len(d_ptrs)
        ptrs = (C.c_void_p * G)(*[C.c_void_p(p) for p in d_ptrs])
        rws = (C.c_size_t * G)(*rows)
        devs = (C.c_int * G)(*devices)
        h = C.c_void_p()
        N.check(N.lib().leann_sharded_build_device(int(backend_type), ptrs, rws, G, dims, ld, graph_degree, complexity, devs, C.byref(h)))
        return cls(h, keep)

    @classmethod
    def from_searchers(cls, searchers, take_ownership=False):
        G = len(searchers)
        hs = (C.c_void_p * G)(*[s._h for s in searchers])
        h = C.c_void_p()
        N.check(N.lib().leann_sharded_from_handles(hs, G, 1 if take_ownership else 0, C.byref(h)))
        if take_ownership:
            for s in searchers:
                s._h = None
        return cls(h, tuple(searchers))

    @staticmethod
    def rccl_unique_id():
        """128 opaque bytes made by rank 0; the host distributes them to the other ranks by its own means"""
        buf = C.create_string_buffer(128)
        N.check(N.lib().leann_rccl_get_unique_id(buf))
        return buf.raw

    @classmethod
    def attach(cls, searcher, unique_id, world, rank, total_rows=0):
        """join this rank's shard to the RCCL group (collective: every rank calls it)"""
        h = C.c_void_p()
        N.check(N.lib().leann_sharded_attach(searcher._h, C.c_char_p(bytes(unique_id)), world, rank, total_rows, C.byref(h)))
        return cls(h, (searcher,))

    def search_batch_device(self, d_queries, nq, top_k, complexity, d_keys, d_dists, d_counts, d_stats=None, stream=None):
        N.check(N.lib().leann_sharded_search_batch_device(self._h, d_queries, nq, top_k, complexity, d_keys, d_dists, d_counts, d_stats, stream))

    def search_batch_device_async(self, d_queries, nq, top_k, complexity, d_keys, d_dists, d_counts, d_stats=None, stream=None):
        """traversal queued behind `stream`, exchange + merge on the handle's own stream; returns a ticket for wait()"""
        t = C.c_uint64(0)
        N.check(N.lib().leann_sharded_search_batch_device_async(self._h, d_queries, nq, top_k, complexity, d_keys, d_dists, d_counts, d_stats,
                                                                stream, C.byref(t)))
        return t.value

    def wait(self, ticket, stream=None):
        N.check(N.lib().leann_sharded_wait(self._h, ticket, stream))

    def as_backend(self, backend_type=BackendType.Hnsw):
        """the same group as an ordinary BackendSearcher handle (closing it closes the group)"""
        h = C.c_void_p()
        N.check(N.lib().leann_sharded_as_backend(self._h, C.byref(h)))
        keep, self._h = self._keep, None
        s = BackendSearcher(h, backend_type)
        s._keep = keep
        return s

    def len(self):
        return int(N.lib().leann_sharded_len(self._h))

    def n_shards(self):
        return int(N.lib().leann_sharded_shards(self._h))

    def close(self):
        if self._h:
            N.lib().leann_sharded_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
