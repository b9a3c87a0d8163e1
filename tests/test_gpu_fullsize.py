"""-m gpu: BASELINE.json's full size (10M x 768, HNSW M=32) through size-independent properties, plus oracle parity
on a sample with the graph copied back from HBM.  ~60 s on one MI355X.  LEANN_FULLSIZE_ROWS overrides the size."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 0x5EED0001
ROWS = int(os.environ.get("LEANN_FULLSIZE_ROWS", "10000000"))
D, M, EFC, K = 768, 32, 128, 10


@pytest.fixture(scope="module")
def full(la, gpu):
    L, chk = la.lib(), la._native.check
    X = la.DeviceArray((ROWS, D), np.float32)
    chk(L.leann_synth_rows_device(SEED, D, D, 64, 4096, 1.0, 0, 0, ROWS, X.ptr, None))
    nq = 4096
    Q = la.DeviceArray((nq, D), np.float32)
    chk(L.leann_synth_rows_device(SEED, D, D, 64, 4096, 1.0, 1, 0, nq, Q.ptr, None))
    la.sync()
    s = la.BackendSearcher.build_device(la.BackendType.Hnsw, X.ptr, ROWS, D, D, M, EFC)
    yield la, L, chk, X, Q, nq, s
    s.close()


def _search(la, s, qptr, nq, k, ef):
    dk, dd, dc = la.DeviceArray((nq, k), np.uint64), la.DeviceArray((nq, k), np.float32), la.DeviceArray(nq, np.uint32)
    st = la.DeviceArray((nq, 4), np.uint32)
    s.search_batch_device(qptr, nq, k, ef, dk.ptr, dd.ptr, dc.ptr, st.ptr, None)
    la.sync()
    return dk.to_host(), dd.to_host(), dc.to_host(), st.to_host()


def test_fullsize_properties(full):
    la, L, chk, X, Q, nq, s = full
    assert s.len() == ROWS
    keys, dists, counts, st = _search(la, s, Q.ptr, nq, K, 96)
    assert (counts == K).all()
    assert (keys < ROWS).all()
    assert (np.diff(dists, axis=1) >= 0).all()                                   # best first (ascending distance)
    assert all(len(set(r.tolist())) == K for r in keys[:512])                   # no duplicate ids
    assert (dists > -1e-4).all() and (dists < 2.0001).all()                      # 1 - cos on unit vectors
    # determinism / idempotence: same launch again, and the batch split in two halves
    k2, d2, _, _ = _search(la, s, Q.ptr, nq, K, 96)
    assert (k2 == keys).all() and (d2 == dists).all()
    h = nq // 2
    ka, da, _, _ = _search(la, s, Q.ptr, h, K, 96)
    kb, db, _, _ = _search(la, s, Q.ptr + h * D * 4, nq - h, K, 96)
    assert (np.concatenate([ka, kb]) == keys).all() and (np.concatenate([da, db]) == dists).all()
    # prefix property: with the same beam, top-5 is the prefix of top-10
    k5, d5, _, _ = _search(la, s, Q.ptr, 256, 5, 96)
    assert (k5 == keys[:256, :5]).all()
    # recall against the exact scan on the same vectors
    ngt = 1000
    gk, gs, gc = la.DeviceArray((ngt, K), np.uint64), la.DeviceArray((ngt, K), np.float32), la.DeviceArray(ngt, np.uint32)
    chk(L.leann_scan_topk_device(X.ptr, ROWS, D, D, Q.ptr, ngt, K, None, 0, gk.ptr, gs.ptr, gc.ptr, None))
    truth, tscore = gk.to_host(), gs.to_host()
    rec = np.mean([len(set(keys[i].tolist()) & set(truth[i].tolist())) / K for i in range(ngt)])
    assert rec >= 0.95, rec
    # exact scores and ANN distances agree where the ids agree: dist = 1 - score within 1e-5
    for i in range(50):
        common = {int(k): j for j, k in enumerate(truth[i])}
        for j, kk in enumerate(keys[i]):
            if int(kk) in common:
                assert abs((1.0 - tscore[i][common[int(kk)]]) - dists[i][j]) <= 1e-5
    # a stored row finds itself first (distance ~ 0)
    rows = la.DeviceArray((64, D), np.float32)
    chk(L.leann_synth_rows_device(SEED, D, D, 64, 4096, 1.0, 0, 123456, 64, rows.ptr, None))
    ks, ds_, _, _ = _search(la, s, rows.ptr, 64, 1, 96)
    hit = ks[:, 0] == np.arange(123456, 123456 + 64)
    assert hit.mean() >= 0.9 and np.abs(ds_[hit]).max() < 1e-5


def test_fullsize_oracle_parity_on_sample(full, po):
    la, L, chk, X, Q, nq, s = full
    g = s.graph_export(with_vectors=True)
    G = po.Graph.from_arrays(g["vectors"], g["M"], g["M0"], g["max_level"], g["entry"], g["levels"], g["upper_off"],
                             g["adj0"], g["adjU"])
    n = 512
    Qh = Q.to_host()[:n]
    ok, od, oc, ost = G.search_batch(Qh, K, 128, 0, 16)
    gk, gd, gc, gst = _search(la, s, Q.ptr, n, K, 128)
    assert (gk == ok).all() and (gd.view(np.uint32) == od.view(np.uint32)).all() and (gc == oc).all()
    assert (gst[:, 0] == ost[:, 0]).all() and (gst[:, 1] == ost[:, 1]).all() and (gst[:, 2] == ost[:, 2]).all()


def test_fullsize_exact_filter_equals_masked_scan(full):
    """Exact filtered search at full size (allowed rows compacted into a list and gathered) against the masked exact scan over all
    rows (leann_scan_topk_device with the same allow mask): different code paths, the same fmaf chains -> the same keys, and
    distance == 1 - score bit for bit; every key allowed; the filtered walk never beats it."""
    la, L, chk, X, Q, nq, s = full
    rng = np.random.default_rng(3)
    n_q = 256
    for sel in (0.01, 0.0005):
        allowed = rng.random(ROWS) < sel
        bm = np.packbits(allowed, bitorder="little")
        dB = la.DeviceArray.from_host(bm)
        ek, ed, ec = la.DeviceArray((n_q, K), np.uint64), la.DeviceArray((n_q, K), np.float32), la.DeviceArray(n_q, np.uint32)
        s.search_filtered_exact_batch_device(Q.ptr, n_q, K, dB.ptr, 0, ek.ptr, ed.ptr, ec.ptr)
        mk, ms, mc = la.DeviceArray((n_q, K), np.uint64), la.DeviceArray((n_q, K), np.float32), la.DeviceArray(n_q, np.uint32)
        chk(L.leann_scan_topk_device(X.ptr, ROWS, D, D, Q.ptr, n_q, K, dB.ptr, 0, mk.ptr, ms.ptr, mc.ptr, None))
        la.sync()
        ek, ed, ec, mk, ms, mc = ek.to_host(), ed.to_host(), ec.to_host(), mk.to_host(), ms.to_host(), mc.to_host()
        assert (ec == K).all() and (mc == K).all()
        assert (ek == mk).all()
        assert (ed.view(np.uint32) == (np.float32(1.0) - ms).view(np.uint32)).all()
        assert allowed[ek.astype(np.int64)].all()
        wk, wd, wc = la.DeviceArray((n_q, K), np.uint64), la.DeviceArray((n_q, K), np.float32), la.DeviceArray(n_q, np.uint32)
        s.search_filtered_batch_device(Q.ptr, n_q, K, 128, dB.ptr, 0, wk.ptr, wd.ptr, wc.ptr)
        la.sync()
        wd, wc = wd.to_host(), wc.to_host()
        for i in range(n_q):  # the walk's j-th best allowed distance is never better than the exact j-th best
            c = int(wc[i])
            assert (wd[i, :c] >= ed[i, :c] - 1e-6).all()


def test_fullsize_recompute_properties(la, po, gpu, monkeypatch):
    """BASELINE configs[2] at full size (10M passages x 256 bf16 features, W[256 x 768], 64 queries): the emission path equals the
    score-slab path bit for bit, the winners' scores match the oracle's literal recompute (embed, then dot) within 1e-5, prefixes
    are consistent, and the early filter (compacted row list) equals the masked pass over everything."""
    L, chk = la.lib(), la._native.check
    n, h, d, nq, k = ROWS, 256, 768, 64, K
    dF, dW = la.DeviceArray((n, h), np.uint16), la.DeviceArray((h, d), np.uint16)
    chk(L.leann_synth_features_device(SEED, h, 64, 4096, 1.0, 0, 0, n, dF.ptr, None))
    chk(L.leann_synth_weights_device(SEED, h, d, dW.ptr, None))
    W = po.synth_weights(SEED, h, d)
    Q = po.recompute_encode(po.synth_features(SEED, h, 4096, 1.0, 1, 0, nq, r_int=64), W)
    dQ = la.DeviceArray.from_host(Q)
    r = C.c_void_p()
    chk(L.leann_recompute_create(dF.ptr, n, h, dW.ptr, d, 0, 0, C.byref(r)))

    def search(kk, dM=None):
        dk, ds, dc = la.DeviceArray((nq, kk), np.uint64), la.DeviceArray((nq, kk), np.float32), la.DeviceArray(nq, np.uint32)
        chk(L.leann_recompute_search_batch_device(r, dQ.ptr, nq, kk, dM.ptr if dM is not None else None, dk.ptr, ds.ptr, dc.ptr, None))
        la.sync()
        return dk.to_host(), ds.to_host(), dc.to_host()

    gk, gs, gc = search(k)
    assert (gc == k).all() and (gk < n).all() and (np.diff(gs, axis=1) <= 0).all()
    assert all(len(set(row.tolist())) == k for row in gk)
    k5, s5, _ = search(5)
    assert (k5 == gk[:, :5]).all() and (s5.view(np.uint32) == gs[:, :5].view(np.uint32)).all()   # prefix property
    monkeypatch.setenv("LEANN_DEBUG_NO_EMIT", "1")                                                 # same kernel, slab + segment top-k
    la.lib().leann_debug_reload_env()
    sk, ss, _ = search(k)
    monkeypatch.delenv("LEANN_DEBUG_NO_EMIT")
    la.lib().leann_debug_reload_env()
    assert (sk == gk).all() and (ss.view(np.uint32) == gs.view(np.uint32)).all()
    # oracle on the winners (rows fetched back from HBM): score = <l2norm(W^T f), q>  (recompute.rs:96-103)
    for i in (0, 17, 63):
        rows = np.empty((k, h), np.uint16)
        for j, pos in enumerate(gk[i]):
            chk(L.leann_device_download(rows[j].ctypes.data, dF.ptr + int(pos) * h * 2, h * 2))
        E = po.recompute_encode(rows, W)
        assert np.abs(E @ Q[i] - gs[i]).max() <= 1e-5
    # early filter: 1 % allowed through the compacted row list == the masked pass
    rng = np.random.default_rng(5)
    allowed = rng.random(n) < 0.01
    dM = la.DeviceArray.from_host(np.packbits(allowed, bitorder="little"))
    lk, ls, lc = search(k, dM)
    monkeypatch.setenv("LEANN_RECOMPUTE_NO_LIST", "1")
    la.lib().leann_debug_reload_env()
    fk, fs, fc = search(k, dM)
    monkeypatch.delenv("LEANN_RECOMPUTE_NO_LIST")
    la.lib().leann_debug_reload_env()
    assert (lk == fk).all() and (ls.view(np.uint32) == fs.view(np.uint32)).all() and (lc == fc).all()
    assert allowed[lk.astype(np.int64)].all()
    L.leann_recompute_close(r)


def test_fullsize_sharded_handle_properties(full):
    """BASELINE configs[3]'s shape on one device: the 10M rows as two contiguous 5M shards behind ONE handle (csrc/shard.hip), searched
    through the ordinary backend entry points — size-independent properties + recall + exact-score consistency."""
    la, L, chk, X, Q, nq, s = full
    half = (ROWS // 2) & ~63
    sh = la.ShardedIndex.build_device(la.BackendType.Hnsw, [X.ptr, X.ptr + half * D * 4], [half, ROWS - half], D, D, M, EFC, [0, 0])
    assert sh.len() == ROWS and sh.n_shards() == 2
    h = sh.as_backend()
    nqs = 2048

    def _search(la, h, qptr, n_, k, ef):  # (per-query counters of a composite handle are [shards x nq x 4]: not asked for here)
        dk, dd, dc = la.DeviceArray((n_, k), np.uint64), la.DeviceArray((n_, k), np.float32), la.DeviceArray(n_, np.uint32)
        h.search_batch_device(qptr, n_, k, ef, dk.ptr, dd.ptr, dc.ptr, None, None)
        la.sync()
        return dk.to_host(), dd.to_host(), dc.to_host(), None
    keys, dists, counts, _ = _search(la, h, Q.ptr, nqs, K, 96)
    assert (counts == K).all() and (keys < ROWS).all() and (np.diff(dists, axis=1) >= 0).all()
    assert all(len(set(r.tolist())) == K for r in keys[:512])
    assert ((keys >= half).any(axis=1) & (keys < half).any(axis=1)).mean() > 0.5  # answers really come from both shards
    k2, d2, _, _ = _search(la, h, Q.ptr, nqs, K, 96)                              # idempotent
    assert (k2 == keys).all() and (d2 == dists).all()
    ka, da, _, _ = _search(la, h, Q.ptr, 1000, K, 96)                             # batch split
    kb, db, _, _ = _search(la, h, Q.ptr + 1000 * D * 4, nqs - 1000, K, 96)
    assert (np.concatenate([ka, kb]) == keys).all() and (np.concatenate([da, db]) == dists).all()
    ngt = 1000
    gk, gs, gc = la.DeviceArray((ngt, K), np.uint64), la.DeviceArray((ngt, K), np.float32), la.DeviceArray(ngt, np.uint32)
    chk(L.leann_scan_topk_device(X.ptr, ROWS, D, D, Q.ptr, ngt, K, None, 0, gk.ptr, gs.ptr, gc.ptr, None))
    truth, tscore = gk.to_host(), gs.to_host()
    rec = np.mean([len(set(keys[i].tolist()) & set(truth[i].tolist())) / K for i in range(ngt)])
    assert rec >= 0.95, rec
    for i in range(50):
        common = {int(k): j for j, k in enumerate(truth[i])}
        for j, kk in enumerate(keys[i]):
            if int(kk) in common:
                assert abs((1.0 - tscore[i][common[int(kk)]]) - dists[i][j]) <= 1e-5
    h.close()
