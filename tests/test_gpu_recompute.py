"""-m gpu: recompute search (no stored vectors) — bf16-MFMA encode + normalise, f32-MFMA scoring, top-k —
against the oracle restatement of src/index/recompute.rs:86-109 with the dense + L2-normalise provider
tail (src/embedding/candle.rs:165,218-225).  Floating-point stage: tolerance 1e-5 on embeddings and
scores (north_star: "cosine scores within 1e-5"); ids must agree wherever score gaps exceed 2e-5."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 0x5EED0001


def _mk(la, po, n, h, d, nq):
    L, chk = la.lib(), la._native.check
    F = po.synth_features(SEED, h, 64, 1.0, 0, 0, n)
    W = po.synth_weights(SEED, h, d)
    Fq = po.synth_features(SEED, h, 64, 1.0, 1, 0, nq)
    Q = po.recompute_encode(Fq, W)  # queries = embeddings of query-side features
    dF, dW = la.DeviceArray.from_host(F), la.DeviceArray.from_host(W)
    # device generators are bit-identical to the oracle's
    gF, gW = la.DeviceArray((n, h), np.uint16), la.DeviceArray((h, d), np.uint16)
    chk(L.leann_synth_features_device(SEED, h, 0, 64, 1.0, 0, 0, n, gF.ptr, None))
    chk(L.leann_synth_weights_device(SEED, h, d, gW.ptr, None))
    la.sync()
    assert (gF.to_host() == F).all() and (gW.to_host() == W).all()
    chk(L.leann_synth_features_device(SEED, h, 16, 64, 1.0, 1, 77, min(n, 500), gF.ptr, None))  # lifted (intrinsic dim 16) variant
    la.sync()
    assert (gF.to_host()[:min(n, 500)] == po.synth_features(SEED, h, 64, 1.0, 1, 77, min(n, 500), r_int=16)).all()
    r = C.c_void_p()
    chk(L.leann_recompute_create(dF.ptr, n, h, dW.ptr, d, 0, 0, C.byref(r)))
    return L, chk, F, W, Q, r, (dF, dW)


@pytest.mark.parametrize("n,h,d", [(3000, 256, 768), (1000, 64, 128), (777, 100, 200), (5000, 256, 384)])
def test_encode_matches_oracle(la, po, gpu, n, h, d):
    L, chk, F, W, Q, r, keep = _mk(la, po, n, h, d, 4)
    ld = (d + 3) // 4 * 4
    dE = la.DeviceArray((n, ld), np.float32)
    chk(L.leann_recompute_encode_device(r, 0, n, dE.ptr, None))
    la.sync()
    E = dE.to_host()[:, :d]
    ref = po.recompute_encode(F, W)
    assert np.abs(E - ref).max() <= 1e-6
    assert np.abs(np.linalg.norm(E, axis=1) - 1).max() <= 1e-5
    L.leann_recompute_close(r)


@pytest.mark.parametrize("n,h,d,nq,k", [(6000, 256, 768, 70, 10), (2500, 64, 128, 5, 3), (4100, 128, 256, 64, 16),
                                        (4500, 256, 384, 33, 5), (5000, 256, 512, 64, 12), (4200, 256, 600, 3, 10), (300, 256, 768, 2, 10),
                                        (5000, 256, 768, 200, 10), (4800, 256, 768, 300, 4)])
def test_recompute_search_matches_restatement(la, po, gpu, n, h, d, nq, k):
    L, chk, F, W, Q, r, keep = _mk(la, po, n, h, d, nq)
    dQ = la.DeviceArray.from_host(Q)
    dk, ds, dc = la.DeviceArray((nq, k), np.uint64), la.DeviceArray((nq, k), np.float32), la.DeviceArray(nq, np.uint32)
    chk(L.leann_recompute_search_batch_device(r, dQ.ptr, nq, k, None, dk.ptr, ds.ptr, dc.ptr, None))
    la.sync()
    gk, gs, gc = dk.to_host(), ds.to_host(), dc.to_host()
    assert (gc == k).all()
    E = po.recompute_encode(F, W)  # the reference materialises every embedding (recompute.rs:86-93) ...
    for i in range(nq):
        k0, s0 = po.scan_topk(E, Q[i], k + 5, mode=0)  # ... then dot_product + stable sort desc + take (:96-109)
        assert np.abs(gs[i] - s0[:k]).max() <= 1e-5
        assert (np.diff(gs[i]) <= 0).all()
        for j in range(k):
            if gk[i, j] != k0[j]:  # only allowed across a near-tie
                assert abs(s0[j] - s0[list(k0).index(gk[i, j])]) <= 2e-5 if gk[i, j] in k0 else False
    L.leann_recompute_close(r)


def test_recompute_allow_mask_and_offset(la, po, gpu):
    n, h, d, nq, k = 3000, 64, 128, 6, 8
    L, chk, F, W, Q, r0, keep = _mk(la, po, n, h, d, nq)
    L.leann_recompute_close(r0)
    r = C.c_void_p()
    chk(L.leann_recompute_create(keep[0].ptr, n, h, keep[1].ptr, d, 0, 1000000, C.byref(r)))  # key_offset
    mask = np.zeros((n + 7) // 8, np.uint8)
    for i in range(0, n, 3):
        mask[i >> 3] |= 1 << (i & 7)
    dM, dQ = la.DeviceArray.from_host(mask), la.DeviceArray.from_host(Q)
    dk, ds, dc = la.DeviceArray((nq, k), np.uint64), la.DeviceArray((nq, k), np.float32), la.DeviceArray(nq, np.uint32)
    chk(L.leann_recompute_search_batch_device(r, dQ.ptr, nq, k, dM.ptr, dk.ptr, ds.ptr, dc.ptr, None))
    la.sync()
    gk = dk.to_host()
    assert (gk >= 1000000).all() and ((gk - 1000000) % 3 == 0).all()
    E = po.recompute_encode(F, W)
    for i in range(nq):
        k0, s0 = po.scan_topk(E, Q[i], k, mode=0, allow_mask=mask)
        assert np.abs(ds.to_host()[i] - s0).max() <= 1e-5
    L.leann_recompute_close(r)


@pytest.mark.parametrize("L", [1, 2, 4, 8])
def test_masked_mean_pooling_provider(la, po, gpu, L):
    """Token-level provider: dense per token -> masked mean over L tokens (candle.rs:191-216, count clamp 1e-9)
    -> l2_normalize; search scores against the oracle's literal order."""
    n, h, d, nq, k = 1500, 128, 384, 20, 10
    Lc, chk = la.lib(), la._native.check
    F = po.synth_features(SEED, h, 64, 1.0, 0, 0, n * L)          # token rows
    W = po.synth_weights(SEED, h, d)
    rng = np.random.default_rng(L)
    mask = (rng.random((n, L)) < 0.7).astype(np.uint8)
    mask[:5] = 0                                                   # fully padded passages: count clamps to 1e-9 -> zero vector
    mask[5:10] = 1
    E = po.recompute_encode_pooled(F, mask, W, L)
    assert np.abs(E[:5]).max() == 0.0
    Q = po.recompute_encode(po.synth_features(SEED, h, 64, 1.0, 1, 0, nq), W)
    dF, dW, dM = la.DeviceArray.from_host(F), la.DeviceArray.from_host(W), la.DeviceArray.from_host(mask)
    r = C.c_void_p()
    chk(Lc.leann_recompute_create_pooled(dF.ptr, dM.ptr, n, L, h, dW.ptr, d, 0, 0, C.byref(r)))
    dE = la.DeviceArray((n, d), np.float32)
    chk(Lc.leann_recompute_encode_device(r, 0, n, dE.ptr, None))
    la.sync()
    assert np.abs(dE.to_host() - E).max() <= 2e-6
    dQ = la.DeviceArray.from_host(Q)
    dk, ds, dc = la.DeviceArray((nq, k), np.uint64), la.DeviceArray((nq, k), np.float32), la.DeviceArray(nq, np.uint32)
    chk(Lc.leann_recompute_search_batch_device(r, dQ.ptr, nq, k, None, dk.ptr, ds.ptr, dc.ptr, None))
    la.sync()
    gk, gs = dk.to_host(), ds.to_host()
    for i in range(nq):
        k0, s0 = po.scan_topk(E, Q[i], k, mode=0)
        assert np.abs(gs[i] - s0).max() <= 1e-5
        assert len(set(gk[i].tolist()) & set(k0.tolist())) >= k - 1
    Lc.leann_recompute_close(r)


@pytest.mark.parametrize("nq,h", [(200, 256), (700, 256), (700, 128), (130, 128)])
def test_recompute_on_graph_search(la, po, gpu, nq, h):
    """Graph index with NO stored vectors: distances recomputed from bf16 features, dist = 1 - <f, W q> / ||W^T f||.
    (a) GPU traversal == oracle traversal over the same bytes, bit for bit; (b) same neighbours as the stored-vector
    index built from the materialised embeddings (scores within 1e-5); (c) recall vs exact search.
    Batches of <= 512 queries run the 16-waves-per-query form, larger ones the 4-wave form; rows of exactly 256 features are read
    four per wave instruction (search.cuh: group_dist_rows_feat256), other widths one per wave load — all four against the oracle."""
    from util import recall_at_k
    n, d, k = 20000, 768, 10
    Lc, chk = la.lib(), la._native.check
    F = po.synth_features(SEED, h, 64, 1.0, 0, 0, n)
    W = po.synth_weights(SEED, h, d)
    Q = po.recompute_encode(po.synth_features(SEED, h, 64, 1.0, 1, 0, nq), W)
    dF, dW = la.DeviceArray.from_host(F), la.DeviceArray.from_host(W)
    r = C.c_void_p()
    chk(Lc.leann_recompute_create(dF.ptr, n, h, dW.ptr, d, 0, 0, C.byref(r)))
    hb = C.c_void_p()
    chk(Lc.leann_recompute_build_index(r, 0, 16, 64, C.byref(hb)))
    s = la.BackendSearcher(hb, la.BackendType.Hnsw)
    fh, rb = C.c_uint32(0), C.c_uint32(0)
    chk(Lc.leann_backend_feature_rows_export(hb, C.byref(fh), C.byref(rb), None))
    assert fh.value == h and rb.value == 2 * h + 8
    rows = np.zeros((n, rb.value), np.uint8)
    chk(Lc.leann_backend_feature_rows_export(hb, None, None, rows.ctypes.data))
    assert (rows[:, : 2 * h].view(np.uint16) == F).all()
    s.stats(reset=True)
    gk, gd, gc = s.search_batch(Q, k, 64)
    st = s.stats()
    assert st["algorithmic_bytes"] < st["n_dist_evals"] * 600  # 2 h + 8 B per evaluated neighbour, not 3 072
    # (a) oracle over the same graph + feature bytes + projected queries
    g = s.graph_export()
    Gr = po.Graph.from_arrays(np.zeros((n, 1), np.float32), 16, 32, g["max_level"], g["entry"], g["levels"], g["upper_off"],
                              g["adj0"], g["adjU"])
    Gr.set_features(rows, fh.value, rb.value)
    ok, od, oc, ost = Gr.search_batch(po.project_queries(W, Q, fh.value), k, 64, 0, 8)
    assert (gk == ok).all() and (gd.view(np.uint32) == od.view(np.uint32)).all()
    assert st["n_dist_evals"] == int(ost[:, 0].sum())
    # (b) the stored-vector twin: materialised embeddings, same construction -> same graph, same neighbours
    E = po.recompute_encode(F, W)
    dE = la.DeviceArray((n, d), np.float32)
    chk(Lc.leann_recompute_encode_device(r, 0, n, dE.ptr, None))
    la.sync()
    s2 = la.BackendSearcher.build_device(la.BackendType.Hnsw, dE.ptr, n, d, d, 16, 64)
    g2 = s2.graph_export()
    assert (g2["adj0"] == g["adj0"]).all()
    k2, d2, _ = s2.search_batch(Q, k, 64)
    assert np.abs(d2 - gd).max() <= 1e-5
    assert (k2 == gk).mean() >= 0.99  # identical except across float near-ties
    # (c) recall against exact search over the true embeddings
    assert recall_at_k(gk, po.exact_topk(E, Q, k)) >= 0.93
    s2.close()
    s.close()
    Lc.leann_recompute_close(r)


def test_recompute_on_graph_index_round_trips_through_its_file(la, po, gpu, tmp_path):
    """The LEANN state proper — graph kept, vectors dropped (src/index/meta.rs:38-42 is_pruned, src/cli/prune.rs:17-79) — saved and
    reopened: format v2 holds the graph, the bf16 feature rows and the encoder weights; the reopened handle answers bit for bit like
    the in-memory one and holds the same bytes."""
    n, h, d, nq, k = 12000, 256, 384, 64, 10
    Lc, chk = la.lib(), la._native.check
    F = po.synth_features(SEED, h, 64, 1.0, 0, 0, n)
    W = po.synth_weights(SEED, h, d)
    Q = po.recompute_encode(po.synth_features(SEED, h, 64, 1.0, 1, 0, nq), W)
    dF, dW = la.DeviceArray.from_host(F), la.DeviceArray.from_host(W)
    r = C.c_void_p()
    chk(Lc.leann_recompute_create(dF.ptr, n, h, dW.ptr, d, 0, 0, C.byref(r)))
    for backend, deg in ((0, 16), (1, 32)):
        hb = C.c_void_p()
        chk(Lc.leann_recompute_build_index(r, backend, deg, 64, C.byref(hb)))
        s = la.BackendSearcher(hb, backend)
        stem = str(tmp_path / f"b{backend}" / "documents.leann")
        (tmp_path / f"b{backend}").mkdir()
        s.save(stem)
        fname = tmp_path / f"b{backend}" / ("documents.diskann" if backend else "documents.index")
        rows_b = n * 520
        g = s.graph_export()
        assert fname.stat().st_size == 128 + n + 4 * n + 4 * n * g["M0"] + 4 * g["n_upper_lists"] * g["M"] + rows_b + 4 * h * d
        s2 = la.BackendSearcher.load(backend, stem, d)
        g2 = s2.graph_export()
        for key in ("n", "dims", "M", "M0", "max_level", "entry", "n_upper_lists"):
            assert g[key] == g2[key], key
        for key in ("levels", "upper_off", "adj0", "adjU"):
            assert (g[key] == g2[key]).all(), key
        rows1, rows2 = np.zeros((n, 520), np.uint8), np.zeros((n, 520), np.uint8)
        chk(Lc.leann_backend_feature_rows_export(s._h, None, None, rows1.ctypes.data))
        chk(Lc.leann_backend_feature_rows_export(s2._h, None, None, rows2.ctypes.data))
        assert (rows1 == rows2).all()
        for ef in (10, 64, 200):
            k1, d1, c1 = s.search_batch(Q, k, ef)
            k2, d2, c2 = s2.search_batch(Q, k, ef)
            assert (k1 == k2).all() and (d1.view(np.uint32) == d2.view(np.uint32)).all() and (c1 == c2).all()
        with pytest.raises(la.LeannError, match="holds no vectors"):
            s2.graph_export(with_vectors=True)
        with pytest.raises(la.LeannError):  # wrong backend kind for this file
            la.BackendSearcher.load(1 - backend, str(tmp_path / f"b{backend}" / "documents.leann"), d)
        s2.close()
        s.close()
    Lc.leann_recompute_close(r)


def _search(la, r, dQ, nq, k, dM=None):
    L, chk = la.lib(), la._native.check
    dk, ds, dc = la.DeviceArray((nq, k), np.uint64), la.DeviceArray((nq, k), np.float32), la.DeviceArray(nq, np.uint32)
    chk(L.leann_recompute_search_batch_device(r, dQ.ptr, nq, k, dM.ptr if dM is not None else None, dk.ptr, ds.ptr, dc.ptr, None))
    la.sync()
    return dk.to_host(), ds.to_host(), dc.to_host()


def test_multi_chunk_candidate_emission(la, po, gpu, monkeypatch):
    """> 128k passages: chunks after the first emit their few survivors straight from the fused kernel (no score slab).
    Must equal the slab path bit for bit (same kernel arithmetic), agree with the general kernel within 1e-5, and honour
    the allow mask; spot-checked against the oracle restatement on the winners."""
    n, h, d, nq, k = 700000, 256, 768, 150, 10
    L, chk = la.lib(), la._native.check
    dF, dW = la.DeviceArray((n, h), np.uint16), la.DeviceArray((h, d), np.uint16)
    chk(L.leann_synth_features_device(SEED, h, 64, 4096, 1.0, 0, 0, n, dF.ptr, None))
    chk(L.leann_synth_weights_device(SEED, h, d, dW.ptr, None))
    W = po.synth_weights(SEED, h, d)
    Q = po.recompute_encode(po.synth_features(SEED, h, 4096, 1.0, 1, 0, nq, r_int=64), W)
    dQ = la.DeviceArray.from_host(Q)
    r = C.c_void_p()
    chk(L.leann_recompute_create(dF.ptr, n, h, dW.ptr, d, 0, 5000, C.byref(r)))
    gk, gs, gc = _search(la, r, dQ, nq, k)
    assert (gc == k).all() and (np.diff(gs, axis=1) <= 0).all()
    monkeypatch.setenv("LEANN_DEBUG_NO_EMIT", "1")           # same fused kernel, score slab + segment top-k
    la.lib().leann_debug_reload_env()
    sk, ss, sc = _search(la, r, dQ, nq, k)
    assert (gk == sk).all() and (gs.view(np.uint32) == ss.view(np.uint32)).all()
    monkeypatch.setenv("LEANN_DEBUG_FUSED_V1", "1")          # general kernel (different accumulation order)
    la.lib().leann_debug_reload_env()
    vk, vs, vc = _search(la, r, dQ, nq, k)
    assert np.abs(vs - gs).max() <= 1e-5 and (vk == gk).mean() > 0.98
    monkeypatch.delenv("LEANN_DEBUG_NO_EMIT")
    la.lib().leann_debug_reload_env()
    monkeypatch.delenv("LEANN_DEBUG_FUSED_V1")
    la.lib().leann_debug_reload_env()
    # a handle without the fragment-major feature copy (row-major loads): the same arithmetic, bit for bit
    monkeypatch.setenv("LEANN_RECOMPUTE_NO_TILED", "1")
    la.lib().leann_debug_reload_env()
    r2 = C.c_void_p()
    chk(L.leann_recompute_create(dF.ptr, n, h, dW.ptr, d, 0, 5000, C.byref(r2)))
    rk, rs, rc = _search(la, r2, dQ, nq, k)
    assert (rk == gk).all() and (rs.view(np.uint32) == gs.view(np.uint32)).all()
    L.leann_recompute_close(r2)
    monkeypatch.delenv("LEANN_RECOMPUTE_NO_TILED")
    la.lib().leann_debug_reload_env()
    # oracle on the winners: score = <l2norm(W^T f), q>  (recompute.rs:96-103)
    F = dF.to_host()
    for i in (0, 33, 69, 149):
        pos = (gk[i] - 5000).astype(np.int64)
        E = po.recompute_encode(F[pos], W)
        assert np.abs(E @ Q[i] - gs[i]).max() <= 1e-5
    # allow mask (every 5th position) through the emission path
    mask = np.zeros((n + 7) // 8, np.uint8)
    idx = np.arange(0, n, 5)
    np.bitwise_or.at(mask, idx >> 3, (1 << (idx & 7)).astype(np.uint8))
    dM = la.DeviceArray.from_host(mask)
    mk, ms, mc = _search(la, r, dQ, nq, k, dM)
    assert ((mk - 5000) % 5 == 0).all() and (mc == k).all()
    monkeypatch.setenv("LEANN_DEBUG_NO_EMIT", "1")
    la.lib().leann_debug_reload_env()
    nk, ns, nc = _search(la, r, dQ, nq, k, dM)
    assert (mk == nk).all() and (ms.view(np.uint32) == ns.view(np.uint32)).all()
    monkeypatch.delenv("LEANN_DEBUG_NO_EMIT")
    la.lib().leann_debug_reload_env()
    # The early filter (recompute.rs:62-79): a mask that allows at most half of the passages is compacted and only the allowed rows
    # are embedded (fused_fstat_kernel<16, false, true>).  Per-passage arithmetic is unchanged, so the answer equals the masked
    # pass over everything (LEANN_RECOMPUTE_NO_LIST=1) bit for bit — 20 % (emission over a 140k-row list), 0.3 % (slab only),
    # 60 % (stays on the masked pass), an empty mask and a mask with fewer allowed rows than k.
    rng = np.random.default_rng(11)
    for sel in (0.2, 0.003, 0.6, 0.0, 5e-6):
        allowed = rng.random(n) < sel
        if sel == 5e-6:
            allowed[:] = False
            allowed[[7, 123456, 699999]] = True
        mask = np.packbits(allowed, bitorder="little")
        dM = la.DeviceArray.from_host(mask)
        lk, ls, lc = _search(la, r, dQ, nq, k, dM)
        monkeypatch.setenv("LEANN_RECOMPUTE_NO_LIST", "1")
        la.lib().leann_debug_reload_env()
        fk, fs, fc = _search(la, r, dQ, nq, k, dM)
        monkeypatch.delenv("LEANN_RECOMPUTE_NO_LIST")
        la.lib().leann_debug_reload_env()
        assert (lc == fc).all() and (lk == fk).all() and (ls.view(np.uint32) == fs.view(np.uint32)).all(), sel
        assert (lc == min(k, int(allowed.sum()))).all()
        live = lk != np.iinfo(np.uint64).max
        assert allowed[(lk[live] - 5000).astype(np.int64)].all()
    L.leann_recompute_close(r)


def test_candidate_list_overflow_falls_back(la, po, gpu, monkeypatch):
    """Adversarial order: the score rises with the position, so every row of the later chunks beats the running k-th best and
    the per-query candidate lists overflow; the search must notice and repeat on the slab path (same answer)."""
    n, h, d, nq, k = 300000, 256, 768, 3, 10
    L, chk = la.lib(), la._native.check
    rng = np.random.default_rng(1)
    W = po.synth_weights(SEED, h, d)
    u, v = rng.standard_normal(h).astype(np.float32), rng.standard_normal(h).astype(np.float32)
    t = (np.arange(n, dtype=np.float32) / n)[:, None]
    Ff = (1 - t) * u[None, :] + t * v[None, :]
    F = ((Ff.view(np.uint32) + 0x8000) >> 16).astype(np.uint16)            # bf16 (round half up is fine here)
    qf = ((v.view(np.uint32) + 0x8000) >> 16).astype(np.uint16)[None, :]
    Q = np.repeat(po.recompute_encode(qf, W), nq, axis=0)                  # queries = embedding of v: closest to the LAST rows
    dF, dW, dQ = la.DeviceArray.from_host(F), la.DeviceArray.from_host(W), la.DeviceArray.from_host(Q)
    r = C.c_void_p()
    chk(L.leann_recompute_create(dF.ptr, n, h, dW.ptr, d, 0, 0, C.byref(r)))
    gk, gs, gc = _search(la, r, dQ, nq, k)
    monkeypatch.setenv("LEANN_DEBUG_NO_EMIT", "1")
    la.lib().leann_debug_reload_env()
    sk, ss, sc = _search(la, r, dQ, nq, k)
    assert (gk == sk).all() and (gs.view(np.uint32) == ss.view(np.uint32)).all()
    assert gk.min() > n - 2000                                             # the winners are at the far end
    L.leann_recompute_close(r)


def test_host_pointer_twins(la, po, gpu):
    """leann_recompute_create_host / leann_recompute_search_batch (SURVEY 8b recompute boundary: host pointers) == device API"""
    n, h, d, nq, k = 5000, 256, 768, 9, 7
    L, chk, F, W, Q, r, keep = _mk(la, po, n, h, d, nq)
    dQ = la.DeviceArray.from_host(Q)
    gk, gs, gc = _search(la, r, dQ, nq, k)
    u16p, f32p, u64p, u32p, u8p = (C.POINTER(t) for t in (C.c_uint16, C.c_float, C.c_uint64, C.c_uint32, C.c_uint8))
    rh = C.c_void_p()
    chk(L.leann_recompute_create_host(F.ctypes.data_as(u16p), n, h, W.ctypes.data_as(u16p), d, 0, 0, C.byref(rh)))
    hk, hs, hc = np.zeros((nq, k), np.uint64), np.zeros((nq, k), np.float32), np.zeros(nq, np.uint32)
    chk(L.leann_recompute_search_batch(rh, Q.ctypes.data_as(f32p), nq, k, None, hk.ctypes.data_as(u64p), hs.ctypes.data_as(f32p),
                                       hc.ctypes.data_as(u32p)))
    assert (hk == gk).all() and (hs.view(np.uint32) == gs.view(np.uint32)).all() and (hc == gc).all()
    mask = np.zeros((n + 7) // 8, np.uint8)
    mask[::2] = 0x55
    chk(L.leann_recompute_search_batch(rh, Q.ctypes.data_as(f32p), nq, k, mask.ctypes.data_as(u8p), hk.ctypes.data_as(u64p),
                                       hs.ctypes.data_as(f32p), hc.ctypes.data_as(u32p)))
    assert (((hk >> 3) % 2 == 0) & (hk % 2 == 0)).all()
    L.leann_recompute_close(rh)
    L.leann_recompute_close(r)
