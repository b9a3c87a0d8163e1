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
    assert "2 fused_fstat_kernel instantiations scanned, 0 early reads" in r.stdout
