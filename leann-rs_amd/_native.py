"""ctypes binding of csrc/libleann_hip.so (the C ABI declared in include/leann_backend.h).

There is deliberately no fallback: if the HIP library is missing or fails to load, importing the
product path raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LEANN_LIB") or os.path.join(_HERE, "csrc", "libleann_hip.so")  # LEANN_LIB: diagnostic builds

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)
f32p = C.POINTER(C.c_float)
vp = C.c_void_p


class LeannError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


class SearchStats(C.Structure):
    _fields_ = [("n_queries", C.c_uint64), ("n_dist_evals", C.c_uint64), ("n_hops_base", C.c_uint64),
                ("n_hops_upper", C.c_uint64), ("n_table_overflow", C.c_uint64),
                ("algorithmic_bytes", C.c_uint64)]


# name -> (restype, argtypes); every symbol of include/leann_backend.h
SIGNATURES = {
    "leann_last_error": (C.c_char_p, []),
    "leann_version": (C.c_char_p, []),
    "leann_debug_reload_env": (None, []),
    "leann_backend_shard_count": (C.c_size_t, [vp]),
    "leann_backend_shard": (C.c_int, [vp, C.c_size_t, C.POINTER(vp)]),
    "leann_hybrid_rerank_device": (C.c_int, [vp, vp, vp, C.c_size_t, C.c_size_t, vp, vp, vp, C.c_size_t, C.c_size_t, C.c_float, C.c_int,
                                            C.c_size_t, vp, vp, vp, vp]),
    "leann_backend_open": (C.c_int, [C.c_char_p, C.c_int, C.c_size_t, C.c_char_p, C.POINTER(vp)]),
    "leann_backend_search": (C.c_int, [vp, f32p, C.c_size_t, C.c_size_t, u64p, f32p, C.POINTER(C.c_size_t)]),
    "leann_backend_search_batch": (C.c_int, [vp, f32p, C.c_size_t, C.c_size_t, C.c_size_t, u64p, f32p, u32p]),
    "leann_backend_search_filtered": (C.c_int, [vp, f32p, C.c_size_t, C.c_size_t, u8p, u64p, f32p, C.POINTER(C.c_size_t)]),
    "leann_backend_search_filtered_batch": (C.c_int, [vp, f32p, C.c_size_t, C.c_size_t, C.c_size_t, u8p, C.c_size_t,
                                                     u64p, f32p, u32p]),
    "leann_backend_set_coalescing": (C.c_int, [vp, C.c_uint32, C.c_uint32]),
    "leann_backend_coalescing_stats": (C.c_int, [vp, u64p, u64p]),
    "leann_backend_len": (C.c_size_t, [vp]),
    "leann_backend_dims": (C.c_size_t, [vp]),
    "leann_backend_close": (None, [vp]),
    "leann_backend_build": (C.c_int, [C.c_int, f32p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_char_p]),
    "leann_backend_add": (C.c_int, [C.c_int, f32p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_char_p]),
    "leann_backend_stats": (C.c_int, [vp, C.POINTER(SearchStats), C.c_int]),
    "leann_backend_build_device": (C.c_int, [C.c_int, vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t,
                                            C.c_size_t, C.c_int, C.c_uint64, C.c_int, C.POINTER(vp)]),
    "leann_backend_from_arrays": (C.c_int, [C.c_int, f32p, C.c_size_t, C.c_size_t, C.c_uint32, C.c_uint32,
                                           C.c_uint32, C.c_uint32, u8p, u32p, u32p, u32p, C.c_size_t,
                                           C.c_int, C.c_uint64, C.POINTER(vp)]),
    "leann_backend_graph_info": (C.c_int, [vp, u64p]),
    "leann_backend_graph_export": (C.c_int, [vp, u8p, u32p, u32p, u32p, f32p]),
    "leann_backend_save": (C.c_int, [vp, C.c_char_p]),
    "leann_backend_search_batch_device": (C.c_int, [vp, vp, C.c_size_t, C.c_size_t, C.c_size_t, vp, vp, vp,
                                                   vp, vp]),
    "leann_backend_search_filtered_exact_batch": (C.c_int, [vp, f32p, C.c_size_t, C.c_size_t, u8p, C.c_size_t, u64p, f32p, u32p]),
    "leann_backend_filter_create": (C.c_int, [vp, u8p, C.POINTER(vp)]),
    "leann_backend_filter_count": (C.c_size_t, [vp]),
    "leann_backend_filter_free": (None, [vp]),
    "leann_backend_search_filter_batch": (C.c_int, [vp, f32p, C.c_size_t, C.c_size_t, C.c_size_t, vp, C.c_int, u64p, f32p, u32p]),
    "leann_backend_search_filtered_exact_batch_device": (C.c_int, [vp, vp, C.c_size_t, C.c_size_t, vp, C.c_size_t, vp, vp, vp, vp]),
    "leann_backend_search_filtered_batch_device": (C.c_int, [vp, vp, C.c_size_t, C.c_size_t, C.c_size_t, vp, C.c_size_t,
                                                            vp, vp, vp, vp, vp]),
    "leann_backend_device_rows": (vp, [vp]),
    "leann_synth_rows_device": (C.c_int, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.c_float, C.c_uint32, C.c_uint64, C.c_uint64, vp, vp]),
    "leann_scan_topk_device": (C.c_int, [vp, C.c_size_t, C.c_size_t, C.c_size_t, vp, C.c_size_t, C.c_size_t,
                                        vp, C.c_uint64, vp, vp, vp, vp]),
    "leann_recompute_create": (C.c_int, [vp, C.c_size_t, C.c_size_t, vp, C.c_size_t, C.c_int, C.c_uint64, C.POINTER(vp)]),
    "leann_recompute_create_pooled": (C.c_int, [vp, vp, C.c_size_t, C.c_size_t, C.c_size_t, vp, C.c_size_t, C.c_int, C.c_uint64,
                                               C.POINTER(vp)]),
    "leann_recompute_create_sharded": (C.c_int, [C.POINTER(vp), C.c_size_t, C.POINTER(vp)]),
    "leann_recompute_create_host": (C.c_int, [C.POINTER(C.c_uint16), C.c_size_t, C.c_size_t, C.POINTER(C.c_uint16), C.c_size_t, C.c_int,
                                             C.c_uint64, C.POINTER(vp)]),
    "leann_recompute_search_batch": (C.c_int, [vp, f32p, C.c_size_t, C.c_size_t, u8p, u64p, f32p, u32p]),
    "leann_recompute_search_batch_device": (C.c_int, [vp, vp, C.c_size_t, C.c_size_t, vp, vp, vp, vp, vp]),
    "leann_recompute_encode_device": (C.c_int, [vp, C.c_uint64, C.c_uint64, vp, vp]),
    "leann_recompute_len": (C.c_size_t, [vp]),
    "leann_recompute_last_timing": (C.c_int, [vp, f32p]),
    "leann_recompute_close": (None, [vp]),
    "leann_recompute_build_index": (C.c_int, [vp, C.c_int, C.c_size_t, C.c_size_t, C.POINTER(vp)]),
    "leann_backend_feature_rows_export": (C.c_int, [vp, u32p, u32p, vp]),
    "leann_synth_features_device": (C.c_int, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_uint32, C.c_uint64,
                                             C.c_uint64, vp, vp]),
    "leann_synth_weights_device": (C.c_int, [C.c_uint64, C.c_uint32, C.c_uint32, vp, vp]),
    "leann_merge_topk_device": (C.c_int, [vp, vp, vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t,
                                         C.c_int, vp, vp, vp, vp]),
    "leann_sharded_open": (C.c_int, [C.c_char_p, C.c_int, C.c_size_t, C.c_char_p, C.POINTER(vp)]),
    "leann_sharded_build_device": (C.c_int, [C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t), C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t,
                                            C.c_size_t, C.POINTER(C.c_int), C.POINTER(vp)]),
    "leann_sharded_from_handles": (C.c_int, [C.POINTER(vp), C.c_size_t, C.c_int, C.POINTER(vp)]),
    "leann_sharded_as_backend": (C.c_int, [vp, C.POINTER(vp)]),
    "leann_rccl_get_unique_id": (C.c_int, [vp]),
    "leann_sharded_attach": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_size_t, C.POINTER(vp)]),
    "leann_sharded_search_batch_device": (C.c_int, [vp, vp, C.c_size_t, C.c_size_t, C.c_size_t, vp, vp, vp, vp, vp]),
    "leann_sharded_search_batch_device_async": (C.c_int, [vp, vp, C.c_size_t, C.c_size_t, C.c_size_t, vp, vp, vp, vp, vp, u64p]),
    "leann_sharded_wait": (C.c_int, [vp, C.c_uint64, vp]),
    "leann_sharded_len": (C.c_size_t, [vp]),
    "leann_sharded_shards": (C.c_size_t, [vp]),
    "leann_sharded_close": (None, [vp]),
    "leann_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "leann_device_malloc": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(vp)]),
    "leann_device_free": (C.c_int, [vp]),
    "leann_device_upload": (C.c_int, [vp, vp, C.c_size_t]),
    "leann_device_download": (C.c_int, [vp, vp, C.c_size_t]),
    "leann_device_sync": (C.c_int, [C.c_int]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make` (hipcc --offload-arch=gfx950). "
                "leann-rs_amd has no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise LeannError(rc, lib().leann_last_error().decode("utf-8", "replace"))


def device_count():
    n = C.c_int(0)
    lib().leann_device_count(C.byref(n))
    return n.value
