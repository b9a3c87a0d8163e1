"""CPU suite: leann_backend_open never trusts an index file: the header is checked against the file length before
anything is allocated, every graph array is validated against n before upload, nothing aborts the process.  A well-formed file
gets as far as the device (LEANN_ERR_DEVICE on a box without a GPU) — that is the control.  No GPU needed."""
import os
import struct

import numpy as np
import pytest

from util import synth, write_gx1


@pytest.fixture(scope="module")
def good(po):
    X = synth(po, 300, 32)
    G = po.Graph.build_hnsw(X, M=4, efc=16)
    lv, uo, a0, aU = G.export()
    return dict(X=X, M=4, M0=8, max_level=G.max_level, entry=G.entry, levels=lv, upper_off=uo, adj0=a0, adjU=aU)


def _open(la, tmp_path, kind=0, dims=32):
    return la.BackendSearcher.load(kind, str(tmp_path / "documents.leann"), dims)


def test_well_formed_file_reaches_the_device(la, good, tmp_path):
    write_gx1(tmp_path / "documents.index", 0, **good)
    if la.device_count() > 0:
        s = _open(la, tmp_path)
        assert s.len() == 300
        s.close()
    else:
        with pytest.raises(la.LeannError) as e:
            _open(la, tmp_path)
        assert e.value.code == 4 and "no CPU fallback" in str(e.value)  # validation passed; only the GPU is missing


def test_truncated_and_padded_files(la, good, tmp_path):
    write_gx1(tmp_path / "documents.index", 0, **good)
    raw = (tmp_path / "documents.index").read_bytes()
    for cut in (len(raw) - 1, len(raw) // 2, 200, 128, 100, 9):
        (tmp_path / "documents.index").write_bytes(raw[:cut])
        with pytest.raises(la.LeannError) as e:
            _open(la, tmp_path)
        assert e.value.code == 3, cut
    (tmp_path / "documents.index").write_bytes(raw + b"\0" * 16)
    with pytest.raises(la.LeannError, match="file length does not match the header"):
        _open(la, tmp_path)


@pytest.mark.parametrize("field,value,needle", [
    ("n", 2 ** 40, "out of range"),          # would have sized std::vectors from the header (std::bad_alloc across the C boundary)
    ("n", 10 ** 6, "file length"),
    ("M0", 4096, "out of range"),
    ("M", 0, "out of range"),
    ("d", 1 << 20, "out of range"),
    ("n_upper_lists", 2 ** 50, "out of range"),
    ("entry", 300, "out of range"),
    ("max_level", 200, "out of range"),
    ("version", 7, "unsupported version"),
    ("kind", 1, "holds a DiskANN graph"),
])
def test_header_fields_are_bounded(la, good, tmp_path, field, value, needle):
    write_gx1(tmp_path / "documents.index", 0, **good)
    raw = bytearray((tmp_path / "documents.index").read_bytes())
    off = {"version": (8, "<I"), "kind": (12, "<I"), "n": (16, "<Q"), "d": (24, "<I"), "M": (28, "<I"), "M0": (32, "<I"),
           "max_level": (36, "<I"), "entry": (40, "<I"), "n_upper_lists": (56, "<Q")}[field]
    struct.pack_into(off[1], raw, off[0], value)
    (tmp_path / "documents.index").write_bytes(bytes(raw))
    with pytest.raises(la.LeannError) as e:
        _open(la, tmp_path)
    assert e.value.code == 3 and needle in str(e.value), str(e.value)


def test_graph_arrays_are_validated(la, good, tmp_path):
    n = 300

    def expect(msg, **over):
        g = dict(good)
        g.update({k: v.copy() if hasattr(v, "copy") else v for k, v in over.items()})
        write_gx1(tmp_path / "documents.index", 0, **g)
        with pytest.raises(la.LeannError) as e:
            _open(la, tmp_path)
        assert e.value.code == 3 and msg in str(e.value), str(e.value)

    a0 = good["adj0"].copy(); a0[17, 3] = n  # neighbour id >= n: an out-of-bounds row read in the traversal kernel
    expect("level-0 neighbour id >= n", adj0=a0)
    a0 = good["adj0"].copy(); a0[0, 0] = 0xFFFFFFFE
    expect("level-0 neighbour id >= n", adj0=a0)
    top = int(np.argmax(good["levels"]))
    assert good["levels"][top] >= 1
    aU = good["adjU"].copy(); aU[good["upper_off"][top], 0] = n + 5
    expect("upper-level neighbour id >= n", adjU=aU)
    low = int(np.argmin(good["levels"]))  # a level-0-only node named on level 1: its "list" would be somebody else's
    aU = good["adjU"].copy(); aU[good["upper_off"][top], 0] = low
    expect("does not exist on that level", adjU=aU)
    uo = good["upper_off"].copy(); uo[top] = len(good["adjU"])
    expect("runs past the upper lists", upper_off=uo)
    lv = good["levels"].copy(); lv[good["entry"]] = 0
    expect("entry point does not reach max_level", levels=lv)
    lv = good["levels"].copy(); lv[5] = 99
    expect("node level > 15", levels=lv)


def test_from_arrays_validates_too(la, good):
    a0 = good["adj0"].copy(); a0[1, 1] = 300
    with pytest.raises(la.LeannError) as e:
        la.BackendSearcher.from_arrays(la.BackendType.Hnsw, good["X"], 4, 8, good["max_level"], good["entry"], good["levels"],
                                       good["upper_off"], a0, good["adjU"])
    assert e.value.code == 3


def test_bit_flips_never_crash(la, good, tmp_path):
    """every single-bit flip of the header, and a sample of the payload, either loads (GPU box), is refused, or stops at the device check"""
    write_gx1(tmp_path / "documents.index", 0, **good)
    raw = (tmp_path / "documents.index").read_bytes()
    rng = np.random.default_rng(7)
    positions = list(range(8, 68)) + [int(x) for x in rng.integers(128, len(raw) - 300 * 32 * 4, 200)]
    for pos in positions:
        b = bytearray(raw)
        b[pos] ^= 1 << int(rng.integers(0, 8))
        (tmp_path / "documents.index").write_bytes(bytes(b))
        try:
            s = _open(la, tmp_path)
            s.close()
        except la.LeannError as e:
            assert e.code in (3, 4), (pos, str(e))


def test_foreign_file_without_embeddings_keeps_the_reference_message(la, tmp_path):
    (tmp_path / "documents.index").write_bytes(b"usearch" + bytes(range(200)))
    with pytest.raises(la.LeannError) as e:
        _open(la, tmp_path)
    assert e.value.code == 3 and "incompatible format" in str(e.value) and "documents.embeddings" in str(e.value)
    # embeddings of the wrong size are refused, not mis-read
    (tmp_path / "documents.embeddings").write_bytes(b"\0" * (32 * 4 * 10 + 3))
    with pytest.raises(la.LeannError) as e:
        _open(la, tmp_path)
    assert e.value.code == 3 and "not a whole number" in str(e.value)


def test_sharded_open_argument_and_file_errors_need_no_gpu(la, good, tmp_path):
    """device lists are parsed, and the row source of a sharded open is located and checked, before any device is touched"""
    stem = str(tmp_path / "documents.leann")
    for spec in ("0,x", "3-1", "0,,1", "-2", "0-9999"):
        with pytest.raises(la.LeannError) as e:
            la.BackendSearcher.load(0, stem, 32, device=spec)
        assert e.value.code == 1 and "device_spec" in str(e.value), spec
    with pytest.raises(la.LeannError) as e:  # nothing to partition
        la.BackendSearcher.load(0, stem, 32, device="0,1")
    assert e.value.code == 2 and "documents.embeddings" in str(e.value)
    (tmp_path / "documents.index").write_bytes(b"usearch" + bytes(200))  # a foreign file holds no rows we can read
    with pytest.raises(la.LeannError) as e:
        la.BackendSearcher.load(0, stem, 32, device="0,1")
    assert e.value.code == 3 and "holds no rows to partition" in str(e.value)
    (tmp_path / "documents.embeddings").write_bytes(b"\0" * (32 * 4 * 7 + 1))  # not a whole number of rows: ignored as a row source
    with pytest.raises(la.LeannError) as e:
        la.BackendSearcher.load(0, stem, 32, device="0,1")
    assert e.value.code == 3
    write_gx1(tmp_path / "documents.index", 0, **good)  # our own file is a row source; on a box without GPUs it stops at the device check
    (tmp_path / "documents.embeddings").unlink()
    if la.device_count() == 0:
        with pytest.raises(la.LeannError) as e:
            la.BackendSearcher.load(0, stem, 32, device="0,0")
        assert e.value.code == 4 and "no CPU fallback" in str(e.value)
    else:
        s = la.BackendSearcher.load(0, stem, 32, device="0,0")
        assert s.len() == 300
        s.close()


def test_sharded_entry_points_reject_bad_arguments(la):
    import ctypes as C
    L = la.lib()
    h = C.c_void_p()
    assert L.leann_sharded_from_handles(None, 0, 0, C.byref(h)) == 1
    assert L.leann_sharded_open(None, 0, 32, b"0,1", C.byref(h)) == 1
    assert L.leann_sharded_attach(None, None, 2, 0, 0, C.byref(h)) == 1
    assert L.leann_sharded_search_batch_device(None, None, 1, 1, 1, None, None, None, None, None) == 1
    assert L.leann_sharded_wait(None, 0, None) == 1
    assert L.leann_sharded_len(None) == 0 and L.leann_sharded_shards(None) == 0
    L.leann_sharded_close(None)
