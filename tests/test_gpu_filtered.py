"""-m gpu: filtered traversal (allow-bitmap inside the kernel) vs the oracle, bit-exact.
The reference has no such path (it over-fetches 5*top_k and post-filters, searcher.rs:129-133,:190-194), so the pinned
behaviour is the restatement in oracle/oracle.c:graph_search_ctx — "parity unpinned" against the reference itself."""
import ctypes as C

import numpy as np
import pytest

from util import SEED, recall_at_k, synth

pytestmark = pytest.mark.gpu


def _bitmap(rng, n, sel, nq=None):
    shape = (n,) if nq is None else (nq, n)
    allowed = rng.random(shape) < sel
    return np.packbits(allowed, axis=-1, bitorder="little"), allowed


def _same(G, s, Q, k, ef, bm, algo=0):
    ok, od, oc, ost = G.search_filtered_batch(Q, k, ef, bm, algo, nthreads=8)
    s.stats(reset=True)
    gk, gd, gc = s.search_filtered_batch(Q, k, ef, bm)
    st = s.stats()
    assert (gc == oc).all()
    assert (gk == ok).all(), f"ids differ in {(gk != ok).any(axis=1).sum()} of {len(Q)} queries"
    assert (gd.view(np.uint32) == od.view(np.uint32)).all()
    assert st["n_dist_evals"] == int(ost[:, 0].sum()) and st["n_hops_base"] == int(ost[:, 1].sum())
    return gk, gd, gc


@pytest.mark.parametrize("n,d,M,ef,nq", [(4000, 128, 16, 64, 64), (3000, 768, 32, 100, 600), (1500, 1536, 8, 32, 40)])
def test_filtered_hnsw_matches_oracle(la, po, gpu, n, d, M, ef, nq):
    rng = np.random.default_rng(n)
    X = synth(po, n, d)
    Q = synth(po, nq, d, stream=1)
    G = po.Graph.build_hnsw(X, M=M, efc=64)
    lv, uo, a0, aU = G.export()
    s = la.BackendSearcher.from_arrays(la.BackendType.Hnsw, X, M, 2 * M, G.max_level, G.entry, lv, uo, a0, aU)
    for sel in (0.5, 0.1, 0.01):
        bm, allowed = _bitmap(rng, n, sel)
        for k, e in ((10, ef), (1, 1), (ef, ef), (7, 3)):
            gk, gd, gc = _same(G, s, Q, k, e, bm)
            for i in range(len(Q)):
                ids = gk[i, : gc[i]].astype(np.int64)
                assert allowed[ids].all()          # nothing disallowed comes back
                assert (np.diff(gd[i, : gc[i]]) >= 0).all()
    # one bitmap per query
    bmq, allowed_q = _bitmap(rng, n, 0.2, nq=len(Q))
    gk, gd, gc = _same(G, s, Q, 10, ef, bmq)
    for i in range(len(Q)):
        assert allowed_q[i, gk[i, : gc[i]].astype(np.int64)].all()
    # all-ones == unfiltered; all-zeros == empty
    ones = np.full((n + 7) // 8, 0xFF, np.uint8)
    uk, ud, uc = s.search_batch(Q, 10, ef)
    gk, gd, gc = s.search_filtered_batch(Q, 10, ef, ones)
    assert (gk == uk).all() and (gd.view(np.uint32) == ud.view(np.uint32)).all() and (gc == uc).all()
    gk, gd, gc = s.search_filtered_batch(Q, 10, ef, np.zeros((n + 7) // 8, np.uint8))
    assert (gc == 0).all() and (gk == np.iinfo(np.uint64).max).all()
    # single-query entry point
    bm, allowed = _bitmap(rng, n, 0.3)
    k1, d1 = s.search_filtered(Q[5], 10, ef, bm)
    ok, od, oc, _ = G.search_filtered_batch(Q[5:6], 10, ef, bm)
    assert (k1 == ok[0, : oc[0]]).all() and (d1.view(np.uint32) == od[0, : oc[0]].view(np.uint32)).all()
    with pytest.raises(la.LeannError):
        s.search_filtered(Q[5], 10, ef, bm[:-2])
    s.close()


def test_filtered_vamana_and_hbm_table(la, po, gpu, monkeypatch):
    n, d, R = 5000, 128, 24
    rng = np.random.default_rng(7)
    X = synth(po, n, d, r=0)
    Q = synth(po, 100, d, stream=1, r=0)
    G = po.Graph.build_vamana(X, R=R, L=48, alpha=1.2)
    lv, uo, a0, aU = G.export()
    s = la.BackendSearcher.from_arrays(la.BackendType.DiskAnn, X, R, R, 0, G.entry, lv, uo, a0, aU)
    bm, _ = _bitmap(rng, n, 0.1)
    _same(G, s, Q, 10, 64, bm, algo=1)
    monkeypatch.setenv("LEANN_DEBUG_HASH_BITS", "8")  # every query migrates to the HBM visited table mid-search
    la.lib().leann_debug_reload_env()
    _same(G, s, Q, 10, 32, bm, algo=1)
    assert s.stats()["n_table_overflow"] == len(Q)
    s.close()


def test_in_traversal_filter_beats_post_filter(la, po, gpu):
    """Quality: at 5 % selectivity the in-kernel filter (pool = every evaluated node) finds far more of the true filtered
    top-10 than the reference's 5x over-fetch + post-filter, for the same traversal."""
    n, d, M, ef, k = 30000, 128, 16, 64, 10
    rng = np.random.default_rng(11)
    X = synth(po, n, d)
    Q = synth(po, 200, d, stream=1)
    dX = la.DeviceArray.from_host(X)
    s = la.BackendSearcher.build_device(la.BackendType.Hnsw, dX.ptr, n, d, d, M, 100)
    bm, allowed = _bitmap(rng, n, 0.05)
    idx = np.nonzero(allowed)[0]
    sc = Q @ X[idx].T
    truth = idx[np.argsort(-sc, axis=1, kind="stable")[:, :k]]
    fk, _, fc = s.search_filtered_batch(Q, k, ef, bm)
    pk, _, pc = s.search_batch(Q, 5 * k, ef)  # searcher.rs:129-133
    post = np.full((len(Q), k), -1, np.int64)
    for i in range(len(Q)):
        keep = [int(x) for x in pk[i, : pc[i]] if allowed[int(x)]][:k]
        post[i, : len(keep)] = keep
    r_in = recall_at_k(np.where(np.arange(k)[None, :] < fc[:, None], fk.astype(np.int64), -1), truth)
    r_post = recall_at_k(post, truth)
    assert r_in > r_post + 0.2, (r_in, r_post)
    assert r_in > 0.7, r_in
    s.close()


@pytest.mark.parametrize("nq", [64, 600])
def test_filtered_recompute_on_graph(la, po, gpu, nq):
    """feature-row (no stored vectors) instantiation of the filtered kernel vs the oracle over the same bytes
    (64 queries: 16 waves per query; 600: the 4-wave throughput form)"""
    n, h, d, M, k = 6000, 256, 768, 16, 10
    rng = np.random.default_rng(3)
    Lc, chk = la.lib(), la._native.check
    F = po.synth_features(SEED, h, 64, 1.0, 0, 0, n)
    W = po.synth_weights(SEED, h, d)
    Q = po.recompute_encode(po.synth_features(SEED, h, 64, 1.0, 1, 0, nq), W)
    dF, dW = la.DeviceArray.from_host(F), la.DeviceArray.from_host(W)
    r = C.c_void_p()
    chk(Lc.leann_recompute_create(dF.ptr, n, h, dW.ptr, d, 0, 0, C.byref(r)))
    hb = C.c_void_p()
    chk(Lc.leann_recompute_build_index(r, 0, M, 64, C.byref(hb)))
    s = la.BackendSearcher(hb, la.BackendType.Hnsw)
    fh, rb = C.c_uint32(0), C.c_uint32(0)
    chk(Lc.leann_backend_feature_rows_export(hb, C.byref(fh), C.byref(rb), None))
    rows = np.zeros((n, rb.value), np.uint8)
    chk(Lc.leann_backend_feature_rows_export(hb, None, None, rows.ctypes.data))
    g = s.graph_export()
    Gr = po.Graph.from_arrays(np.zeros((n, 1), np.float32), M, 2 * M, g["max_level"], g["entry"], g["levels"], g["upper_off"],
                              g["adj0"], g["adjU"])
    Gr.set_features(rows, fh.value, rb.value)
    PQ = po.project_queries(W, Q, fh.value)
    for sel in (0.3, 0.02):
        bm, allowed = _bitmap(rng, n, sel)
        ok, od, oc, ost = Gr.search_filtered_batch(PQ, k, 64, bm, 0, 8)
        gk, gd, gc = s.search_filtered_batch(Q, k, 64, bm)
        assert (gc == oc).all() and (gk == ok).all() and (gd.view(np.uint32) == od.view(np.uint32)).all()
        for i in range(len(Q)):
            assert allowed[gk[i, : gc[i]].astype(np.int64)].all()
    s.close()
    Lc.leann_recompute_close(r)


# ---- the selective end: exact filtered search (compacted allowed rows + f32 MFMA scan) --------------------------------------

def _exact_oracle(po, X, Q, k, bm):
    """oracle/oracle.c:orc_scan_topk with the sequential fmaf dot (mode 1) and the early filter of recompute.rs:66-71;
    distance = 1 - score as the backends report it."""
    n_q = len(Q)
    keys = np.full((n_q, k), np.iinfo(np.uint64).max, np.uint64)
    dists = np.full((n_q, k), np.inf, np.float32)
    counts = np.zeros(n_q, np.uint32)
    for i in range(n_q):
        b = bm if bm.ndim == 1 else bm[i]
        kk, ss = po.scan_topk(X, Q[i], k, mode=1, allow_mask=b)
        keys[i, : len(kk)], dists[i, : len(kk)], counts[i] = kk, np.float32(1.0) - ss, len(kk)
    return keys, dists, counts


@pytest.mark.parametrize("n,d,nq", [(20000, 96, 70), (5000, 50, 9), (150000, 64, 33)])
def test_filtered_exact_matches_oracle(la, po, gpu, n, d, nq):
    rng = np.random.default_rng(n + d)
    X = synth(po, n, d)
    Q = synth(po, nq, d, stream=1)
    ld = (d + 3) // 4 * 4                    # device rows are 16-byte aligned, zero padded
    Xp = np.zeros((n, ld), np.float32)
    Xp[:, :d] = X
    dX = la.DeviceArray.from_host(Xp)
    off = 1000
    s = la.BackendSearcher.build_device(la.BackendType.Hnsw, dX.ptr, n, d, ld, 8, 32, key_offset=off)
    for sel in (0.6, 0.05, 0.004):           # 150k x 0.6 = 90k listed rows: the emission pass of the scan runs over a row list
        bm, allowed = _bitmap(rng, n, sel)
        for k in (10, 1, 100):
            ok, od, oc = _exact_oracle(po, X, Q, k, bm)
            gk, gd, gc = s.search_filtered_exact_batch(Q, k, bm)
            assert (gc == oc).all()
            live = ok != np.iinfo(np.uint64).max
            assert (gk[live] == ok[live] + np.uint64(off)).all() and (gk[~live] == np.iinfo(np.uint64).max).all()
            assert (gd.view(np.uint32) == od.view(np.uint32)).all()   # same fmaf chain, same 1 - s: bit for bit
    # one bitmap per query; a query with an empty filter; a filter with fewer allowed rows than k
    bmq, allowed_q = _bitmap(rng, n, 0.01, nq=nq)
    bmq[1] = 0
    bmq[2] = 0
    bmq[2, 5] = 0b00010010  # positions 41 and 44
    ok, od, oc = _exact_oracle(po, X, Q, 10, bmq)
    gk, gd, gc = s.search_filtered_exact_batch(Q, 10, bmq)
    assert gc[1] == 0 and gc[2] == 2 and set(gk[2, :2].tolist()) == {41 + off, 44 + off}
    assert (gc == oc).all() and (gd.view(np.uint32) == od.view(np.uint32)).all()
    live = ok != np.iinfo(np.uint64).max
    assert (gk[live] == ok[live] + np.uint64(off)).all()
    # device-pointer twin
    bm, _ = _bitmap(rng, n, 0.02)
    hk, hd, hc = s.search_filtered_exact_batch(Q, 10, bm)
    dQ, dB = la.DeviceArray.from_host(Q), la.DeviceArray.from_host(bm)
    dk, dd, dc = la.DeviceArray((nq, 10), np.uint64), la.DeviceArray((nq, 10), np.float32), la.DeviceArray(nq, np.uint32)
    s.search_filtered_exact_batch_device(dQ.ptr, nq, 10, dB.ptr, 0, dk.ptr, dd.ptr, dc.ptr)
    la.sync()
    assert (dk.to_host() == hk).all() and (dd.to_host().view(np.uint32) == hd.view(np.uint32)).all() and (dc.to_host() == hc).all()
    # it beats the walk where the walk starves: recall of the exact path is 1 by construction, the traversal's is not
    bm, allowed = _bitmap(rng, n, 0.004)
    ek, _, _ = s.search_filtered_exact_batch(Q, 10, bm)
    wk, _, _ = s.search_filtered_batch(Q, 10, 64, bm)
    assert recall_at_k(wk, ek) <= 1.0
    with pytest.raises(la.LeannError):
        s.search_filtered_exact_batch(Q, 10, bm[:-3])
    with pytest.raises(la.LeannError):
        s.search_filtered_exact_batch(Q, 5000, bm)
    s.close()


def test_registered_filter_matches_the_per_call_paths(la, po, gpu):
    """leann_backend_filter_create uploads + compacts a bitmap once; searching under it must equal the per-call entry points bit
    for bit in both modes, and mode "auto" must pick the exact scan for selective filters and the walk for broad ones."""
    n, d, nq, k, ef = 30000, 64, 40, 10, 48
    rng = np.random.default_rng(9)
    X = synth(po, n, d)
    Q = synth(po, nq, d, stream=1)
    dX = la.DeviceArray.from_host(X)
    s = la.BackendSearcher.build_device(la.BackendType.Hnsw, dX.ptr, n, d, d, 8, 32, key_offset=77)
    for sel, auto_is_exact in ((0.5, False), (0.01, True), (0.0, True)):
        bm, allowed = _bitmap(rng, n, sel)
        f = s.register_filter(bm)
        assert f.count() == int(allowed.sum())
        wk, wd, wc = s.search_filter_batch(Q, k, ef, f, "walk")
        rk, rd, rc = s.search_filtered_batch(Q, k, ef, bm)
        assert (wk == rk).all() and (wd.view(np.uint32) == rd.view(np.uint32)).all() and (wc == rc).all()
        ek, ed, ec = s.search_filter_batch(Q, k, ef, f, "exact")
        xk, xd, xc = s.search_filtered_exact_batch(Q, k, bm)
        assert (ek == xk).all() and (ed.view(np.uint32) == xd.view(np.uint32)).all() and (ec == xc).all()
        ak, ad, ac = s.search_filter_batch(Q, k, ef, f, "auto")
        want = (ek, ed, ec) if auto_is_exact or allowed.sum() <= 65536 else (wk, wd, wc)
        assert (ak == want[0]).all() and (ad.view(np.uint32) == want[1].view(np.uint32)).all() and (ac == want[2]).all()
        # a large batch under a broad filter walks (<= 1.5 % rule), under a selective one still scans
        Qb = np.repeat(Q, 3, axis=0)[:100]
        bk, bd, bc = s.search_filter_batch(Qb, k, ef, f, "auto")
        if sel == 0.5:
            ck, cd, cc = s.search_filtered_batch(Qb, k, ef, bm)
        else:
            ck, cd, cc = s.search_filtered_exact_batch(Qb, k, bm)
        assert (bk == ck).all() and (bd.view(np.uint32) == cd.view(np.uint32)).all() and (bc == cc).all()
        f.close()
    # a filter made for another index is refused
    s2 = la.BackendSearcher.build_device(la.BackendType.Hnsw, dX.ptr, n - 8, d, d, 8, 32)
    f2 = s2.register_filter(np.zeros((n + 7) // 8, np.uint8))
    with pytest.raises(la.LeannError):
        s.search_filter_batch(Q, k, ef, f2, "auto")
    f2.close()
    s2.close()
    s.close()
