"""CPU suite: the C-ABI library loads and exports every symbol include/leann_backend.h declares; error
paths that need no GPU behave like the reference's; no compute calls."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "leann_backend.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(leann_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(la):
    L = la.lib()
    syms = _declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(L, s), f"libleann_hip.so does not export {s}"
    assert set(syms) == set(la._native.SIGNATURES), "ctypes table and header disagree"


def test_product_library_does_not_link_the_oracle(la):
    out = os.popen(f"ldd '{la.LIB_PATH}'").read()
    assert "oracle" not in out
    # and no product source mentions it
    for dp, _, files in os.walk(os.path.join(ROOT, "leann-rs_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".h", ".cpp")):
                src = open(os.path.join(dp, f), errors="replace").read()
                assert "pyoracle" not in src and "liboracle" not in src and "import oracle" not in src, f


def test_open_missing_index_message(la, tmp_path):
    stem = str(tmp_path / "documents.leann")
    with pytest.raises(la.LeannError) as e:
        la.HnswSearcher.load(stem, 128)
    assert e.value.code == 2 and "Index file not found" in str(e.value) and "documents.index" in str(e.value)
    with pytest.raises(la.LeannError) as e:
        la.DiskAnnSearcher.load(stem, 128)
    assert "DiskANN index not found" in str(e.value) and "documents.diskann" in str(e.value)
    (tmp_path / "documents.index").write_bytes(b"IxFl" + b"\0" * 200)  # compat.rs:27-29
    with pytest.raises(la.LeannError) as e:
        la.HnswSearcher.load(stem, 128)
    assert e.value.code == 3 and "Python LEANN (FAISS format)" in str(e.value)
    with pytest.raises(la.LeannError, match="Unknown backend"):
        la.BackendType.from_name("faiss")
    assert la.BackendType.from_name("hnsw") == la.BackendType.Hnsw
    assert la.BackendType.from_name("diskann") == la.BackendType.DiskAnn


def test_no_cpu_fallback_without_gpu(la):
    """On a box without a GPU the product path must fail loudly, never compute on the CPU."""
    if la.device_count() > 0:
        pytest.skip("GPU present")
    import numpy as np
    X = np.zeros((4, 8), np.float32)
    with pytest.raises(la.LeannError) as e:
        la.BackendBuilder(la.BackendType.Hnsw).build(X, [], "/tmp/nonexistent/documents.leann", 8, 4, 8)
    assert e.value.code == 4 and "no CPU fallback" in str(e.value)
    with pytest.raises(la.LeannError) as e:
        la.BackendSearcher.from_arrays(0, X, 2, 4, 0, 0, np.zeros(4, np.uint8), np.zeros(4, np.uint32),
                                       np.full((4, 4), 0xFFFFFFFF, np.uint32), np.zeros((0, 2), np.uint32))
    assert e.value.code == 4


def test_diskann_add_is_refused(la, tmp_path):
    import numpy as np
    with pytest.raises(la.LeannError) as e:  # mod.rs:93-98
        la.BackendBuilder(la.BackendType.DiskAnn).add_to_index(np.zeros((1, 8), np.float32), str(tmp_path / "d.leann"), 8, 0)
    assert e.value.code == 5 and "does not support incremental updates" in str(e.value)


def test_fused_kernel_asm_reads_are_not_consumed_early():
    """fused_fstat_kernel counts its LDS fragment reads by hand (inline asm); hipcc must not touch their destination
    registers before the asm wait that retires them.  scripts/check_fstat_asm.py compiles recompute.hip to ISA and scans."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "check_fstat_asm.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "3 fused_fstat_kernel instantiations scanned, 0 early reads" in r.stdout
    assert "0 vector instructions reading an accumulator within 4 MFMAs" in r.stdout  # MFMA -> VALU hazard distance (no hardware interlock)
    assert "0 readers of an in-flight asm feature load / AGPR shuffles" in r.stdout   # no v_accvgpr moves around the unit loop


def test_new_entry_points_reject_bad_arguments_without_a_gpu(la):
    """filtered search / recompute host twins: argument errors come back as LEANN_ERR_INVALID (1) with a message, or as
    LEANN_ERR_DEVICE (4, "no CPU fallback") when only the missing GPU stands in the way — never a crash, never CPU compute."""
    import numpy as np
    L = la.lib()
    n_out = C.c_size_t(0)
    q = np.zeros(8, np.float32)
    keys, dists, cnt = np.zeros(4, np.uint64), np.zeros(4, np.float32), np.zeros(1, np.uint32)
    f32p, u64p, u32p, u8p, u16p = (C.POINTER(t) for t in (C.c_float, C.c_uint64, C.c_uint32, C.c_uint8, C.c_uint16))
    bm = np.zeros(4, np.uint8)
    rc = L.leann_backend_search_filtered(None, q.ctypes.data_as(f32p), 4, 16, bm.ctypes.data_as(u8p), keys.ctypes.data_as(u64p),
                                         dists.ctypes.data_as(f32p), C.byref(n_out))
    assert rc == 1 and b"null" in L.leann_last_error()
    rc = L.leann_backend_search_filtered_batch_device(None, None, 1, 4, 16, None, 0, None, None, None, None, None)
    assert rc == 1
    rc = L.leann_backend_search_filtered_exact_batch(None, q.ctypes.data_as(f32p), 1, 4, bm.ctypes.data_as(u8p), 0, keys.ctypes.data_as(u64p),
                                                     dists.ctypes.data_as(f32p), cnt.ctypes.data_as(u32p))
    assert rc == 1 and b"null" in L.leann_last_error()
    rc = L.leann_backend_search_filtered_exact_batch_device(None, None, 1, 4, None, 0, None, None, None, None)
    assert rc == 1
    flt = C.c_void_p()
    rc = L.leann_backend_filter_create(None, bm.ctypes.data_as(u8p), C.byref(flt))
    assert rc == 1 and b"null" in L.leann_last_error() and not flt.value
    rc = L.leann_backend_search_filter_batch(None, q.ctypes.data_as(f32p), 1, 4, 16, None, 2, keys.ctypes.data_as(u64p),
                                             dists.ctypes.data_as(f32p), cnt.ctypes.data_as(u32p))
    assert rc == 1
    assert L.leann_backend_filter_count(None) == 0
    L.leann_backend_filter_free(None)
    rc = L.leann_recompute_search_batch(None, q.ctypes.data_as(f32p), 1, 4, None, keys.ctypes.data_as(u64p), dists.ctypes.data_as(f32p),
                                        cnt.ctypes.data_as(u32p))
    assert rc == 1 and b"null" in L.leann_last_error()
    out = C.c_void_p()
    F, W = np.zeros((4, 16), np.uint16), np.zeros((16, 8), np.uint16)
    rc = L.leann_recompute_create_host(F.ctypes.data_as(u16p), 4, 0, W.ctypes.data_as(u16p), 8, 0, 0, C.byref(out))
    assert rc == 1  # h == 0
    if la.device_count() == 0:
        rc = L.leann_recompute_create_host(F.ctypes.data_as(u16p), 4, 16, W.ctypes.data_as(u16p), 8, 0, 0, C.byref(out))
        assert rc == 4 and b"no CPU fallback" in L.leann_last_error()


def test_round3_entry_points_reject_bad_arguments_without_a_gpu(la):
    """hybrid rerank, sharded recompute, shard accessors, the knob reload hook: argument errors are LEANN_ERR_INVALID with a message;
    nothing computes on the CPU."""
    L = la.lib()
    assert L.leann_hybrid_rerank_device(None, None, None, 4, 50, None, None, None, 64, 1000, 0.7, 1, 10, None, None, None, None) == 1
    assert b"null" in L.leann_last_error()
    out = C.c_void_p()
    assert L.leann_recompute_create_sharded(None, 2, C.byref(out)) == 1 and not out.value
    assert L.leann_backend_shard_count(None) == 0
    assert L.leann_backend_shard(None, 0, C.byref(out)) == 1 and b"null" in L.leann_last_error()
    os.environ["LEANN_HNSW_REFERENCE_EF"] = "1"
    try:
        L.leann_debug_reload_env()  # reads the environment again; no device needed
    finally:
        del os.environ["LEANN_HNSW_REFERENCE_EF"]
        L.leann_debug_reload_env()
