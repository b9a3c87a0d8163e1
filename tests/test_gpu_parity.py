"""-m gpu: HIP path vs the CPU oracle on the same seeded inputs, through the C ABI.
Bar: ids bit-exact, distances bit-exact (the canonical wave-order dot is shared), stats identical."""
import ctypes as C

import numpy as np
import pytest

from util import SEED, recall_at_k, synth

pytestmark = pytest.mark.gpu


def _gpu_search(la, s, Q, k, ef):
    keys, dists, counts = s.search_batch(Q, k, ef)
    return keys, dists, counts


def _assert_same(po, G, s, Q, k, ef, algo=0):
    ok, od, oc, ost = G.search_batch(Q, k, ef, algo, nthreads=8)
    s.stats(reset=True)
    gk, gd, gc = s.search_batch(Q, k, ef)
    st = s.stats()
    assert (gc == oc).all()
    assert (gk == ok).all(), f"ids differ in {(gk != ok).any(axis=1).sum()} of {len(Q)} queries"
    assert (gd.view(np.uint32) == od.view(np.uint32)).all()
    assert st["n_dist_evals"] == int(ost[:, 0].sum())
    assert st["n_hops_base"] == int(ost[:, 1].sum())
    assert st["n_hops_upper"] == int(ost[:, 2].sum())
    return gk


@pytest.mark.parametrize("d,r", [(128, 0), (128, 32), (768, 64), (1536, 64), (100, 16), (260, 8), (3072, 64), (4096, 16)])
def test_synth_rows_bit_exact(la, po, gpu, d, r):
    n = 300
    ld = (d + 3) // 4 * 4
    for stream, i0 in ((0, 0), (1, 12345)):
        ref = po.gen_rows(SEED, d, r, 97, 0.7, stream, i0, n)
        buf = la.DeviceArray((n, ld), np.float32)
        la._native.check(la.lib().leann_synth_rows_device(SEED, d, ld, r, 97, 0.7, stream, i0, n, buf.ptr, None))
        la.sync()
        got = buf.to_host()
        assert (got[:, :d].view(np.uint32) == ref.view(np.uint32)).all()
        assert (got[:, d:] == 0).all()


@pytest.mark.parametrize("n,d,M,ef", [(3000, 128, 16, 64), (2000, 768, 32, 128), (1500, 1536, 8, 32), (800, 100, 4, 10),
                                      (1200, 3072, 8, 48), (700, 4096, 8, 32), (900, 2500, 8, 40)])  # 3 072: text-embedding-3-large (embedding/models.rs:113)
def test_hnsw_search_matches_oracle(la, po, gpu, n, d, M, ef):
    X = synth(po, n, d)
    Q = synth(po, 64, d, stream=1)
    G = po.Graph.build_hnsw(X, M=M, efc=64)
    lv, uo, a0, aU = G.export()
    s = la.BackendSearcher.from_arrays(la.BackendType.Hnsw, X, M, 2 * M, G.max_level, G.entry, lv, uo, a0, aU)
    assert s.len() == n and not s.is_empty()
    for k, e in ((10, ef), (1, 1), (5, 3), (ef, ef)):
        _assert_same(po, G, s, Q, k, e)
    # single-query trait call == batch row
    k1, d1 = s.search(Q[3], 10, ef)
    ok, od, _ = G.search(Q[3], 10, ef)
    assert (k1 == ok).all() and (d1 == od).all()
    s.close()


def test_hand_built_graph_traced_visit_order(la, gpu):
    """the same fixture through the HIP kernel: ids, distances and the visit counters of the independent trace (no oracle involved)"""
    from util import traced_graph
    fx, X, levels, upper_off, adj0, adjU = traced_graph()
    s = la.BackendSearcher.from_arrays(la.BackendType.Hnsw, X, fx["M"], fx["M0"], fx["max_level"], fx["entry"], levels, upper_off, adj0, adjU)
    for c in fx["cases"]:
        q = np.array(c["query"], np.float32)
        s.stats(reset=True)
        keys, dists = s.search(q, c["k"], c["ef"])
        st = s.stats()
        assert keys.tolist() == c["ids"] and dists.tolist() == [float(x) for x in c["dists"]]
        assert st["n_dist_evals"] == c["n_evals"] and st["n_hops_base"] == len(c["expanded_base"]) and st["n_hops_upper"] == len(c["expanded_upper"])
    s.close()
    v = la.BackendSearcher.from_arrays(la.BackendType.DiskAnn, X, fx["M0"], fx["M0"], 0, 0, np.zeros(fx["n"], np.uint8),
                                       np.zeros(fx["n"], np.uint32), adj0, np.zeros((0, fx["M0"]), np.uint32))
    for c in fx["vamana_cases"]:  # GreedySearch (DiskANN Alg. 1) traced independently
        v.stats(reset=True)
        keys, dists = v.search(np.array(c["query"], np.float32), c["k"], c["L"])
        st = v.stats()
        assert keys.tolist() == c["ids"] and dists.tolist() == [float(x) for x in c["dists"]]
        assert st["n_dist_evals"] == c["n_evals"] and st["n_hops_base"] == len(c["expanded"])
    v.close()


def test_vamana_search_matches_oracle(la, po, gpu):
    n, d, R = 2500, 128, 24
    X = synth(po, n, d)
    Q = synth(po, 50, d, stream=1)
    G = po.Graph.build_vamana(X, R=R, L=48, alpha=1.2)
    lv, uo, a0, aU = G.export()
    s = la.BackendSearcher.from_arrays(la.BackendType.DiskAnn, X, R, R, 0, G.entry, lv, uo, a0, aU)
    for k, L in ((10, 64), (10, 5), (3, 128)):
        _assert_same(po, G, s, Q, k, L, algo=1)
    s.close()


def test_reference_exact_ef_mode(la, po, gpu, monkeypatch):
    """the reference's HNSW searcher runs at expansion_search = 64 whatever --complexity says (hnsw.rs:49, :83
    `_complexity`).  LEANN_HNSW_REFERENCE_EF=1 when the handle is made latches that into an HNSW handle: every complexity gives the
    oracle's ef = max(64, top_k) answer; a DiskANN handle still honours complexity (diskann.rs:54); a handle made without the
    variable honours it too."""
    n, d, M = 4000, 128, 16
    X = synth(po, n, d, r=0)
    Q = synth(po, 48, d, stream=1, r=0)
    G = po.Graph.build_hnsw(X, M=M, efc=64)
    lv, uo, a0, aU = G.export()
    plain = la.BackendSearcher.from_arrays(la.BackendType.Hnsw, X, M, 2 * M, G.max_level, G.entry, lv, uo, a0, aU)
    monkeypatch.setenv("LEANN_HNSW_REFERENCE_EF", "1")
    la.lib().leann_debug_reload_env()
    ref = la.BackendSearcher.from_arrays(la.BackendType.Hnsw, X, M, 2 * M, G.max_level, G.entry, lv, uo, a0, aU)
    V = po.Graph.build_vamana(X, R=16, L=48, alpha=1.2)
    vl, vu, v0, vU = V.export()
    vam = la.BackendSearcher.from_arrays(la.BackendType.DiskAnn, X, 16, 16, 0, V.entry, vl, vu, v0, vU)
    monkeypatch.delenv("LEANN_HNSW_REFERENCE_EF")
    la.lib().leann_debug_reload_env()  # the mode is a property of the handle from here on
    o64 = G.search_batch(Q, 10, 64, 0, 4)
    for complexity in (1, 16, 64, 300):
        gk, gd, _ = ref.search_batch(Q, 10, complexity)
        assert (gk == o64[0]).all() and (gd == o64[1]).all()
        k1, d1 = ref.search(Q[7], 10, complexity)
        assert (k1 == o64[0][7]).all()
    o100 = G.search_batch(Q, 100, 100, 0, 4)  # top_k above 64 widens the beam (usearch: expansion = max(expansion_search, wanted))
    gk, gd, _ = ref.search_batch(Q, 100, 5)
    assert (gk == o100[0]).all()
    o16 = G.search_batch(Q, 10, 16, 0, 4)
    gk, _, _ = plain.search_batch(Q, 10, 16)
    assert (gk == o16[0]).all() and not (o16[0] == o64[0]).all()
    v16 = V.search_batch(Q, 10, 16, 1, 4)
    gk, _, _ = vam.search_batch(Q, 10, 16)
    assert (gk == v16[0]).all()
    for s in (plain, ref, vam):
        s.close()


def test_duplicates_and_ties(la, po, gpu):
    """Adversarial: every vector appears 4 times -> exact distance ties must resolve to the lower id."""
    base = synth(po, 500, 128)
    X = np.concatenate([base, base, base, base])
    Q = np.concatenate([base[:20], synth(po, 20, 128, stream=1)])
    G = po.Graph.build_hnsw(X, M=8, efc=32)
    lv, uo, a0, aU = G.export()
    s = la.BackendSearcher.from_arrays(la.BackendType.Hnsw, X, 8, 16, G.max_level, G.entry, lv, uo, a0, aU)
    _assert_same(po, G, s, Q, 8, 40)
    s.close()


def test_tiny_and_short_results(la, po, gpu):
    X = synth(po, 5, 128)
    Q = synth(po, 3, 128, stream=1)
    G = po.Graph.build_hnsw(X, M=4, efc=8)
    lv, uo, a0, aU = G.export()
    s = la.BackendSearcher.from_arrays(la.BackendType.Hnsw, X, 4, 8, G.max_level, G.entry, lv, uo, a0, aU)
    gk, gd, gc = s.search_batch(Q, 10, 64)  # top_k > n: short result, padded
    assert (gc == 5).all()
    assert (gk[:, 5:] == np.iinfo(np.uint64).max).all() and np.isinf(gd[:, 5:]).all()
    _assert_same(po, G, s, Q, 10, 64)
    s.close()


def test_visited_table_overflow_moves_to_hbm_pool(la, po, gpu, monkeypatch):
    """A query whose LDS visited table fills up migrates to a pooled HBM table mid-search; results
    and counters must not change.  LEANN_DEBUG_HASH_BITS forces a 256-slot LDS table."""
    n, d, M = 6000, 128, 32
    X = synth(po, n, d, r=0)  # i.i.d.: searches wander
    Q = synth(po, 300, d, stream=1, r=0)
    G = po.Graph.build_hnsw(X, M=M, efc=32)
    lv, uo, a0, aU = G.export()
    s = la.BackendSearcher.from_arrays(la.BackendType.Hnsw, X, M, 2 * M, G.max_level, G.entry, lv, uo, a0, aU)
    monkeypatch.setenv("LEANN_DEBUG_HASH_BITS", "8")
    la.lib().leann_debug_reload_env()
    _assert_same(po, G, s, Q, 10, 16)
    assert s.stats()["n_table_overflow"] == len(Q)
    monkeypatch.setenv("LEANN_DEBUG_HASH_BITS", "11")
    la.lib().leann_debug_reload_env()
    _assert_same(po, G, s, Q, 10, 64)
    assert 0 < s.stats()["n_table_overflow"] <= len(Q)
    monkeypatch.delenv("LEANN_DEBUG_HASH_BITS")
    la.lib().leann_debug_reload_env()
    _assert_same(po, G, s, Q, 10, 64)
    assert s.stats()["n_table_overflow"] == 0
    s.close()


def test_gpu_built_hnsw(la, po, gpu, tmp_path):
    n, d, M = 20000, 128, 16
    X = synth(po, n, d)
    Q = synth(po, 200, d, stream=1)
    stem = str(tmp_path / "documents.leann")
    la.BackendBuilder(la.BackendType.Hnsw).build(X, [], stem, d, M, 64)
    s = la.HnswSearcher.load(stem, d)
    g = s.graph_export(with_vectors=True)
    assert g["n"] == n and g["M"] == M and g["M0"] == 2 * M
    assert (g["vectors"] == X).all()
    a0 = g["adj0"]
    valid = a0 != 0xFFFFFFFF
    assert (a0[valid] < n).all()
    deg = valid.sum(1)
    assert deg.min() >= 1 and deg.max() <= 2 * M
    assert (valid[:, :-1] >= valid[:, 1:]).all()  # lists compact
    rows = np.repeat(np.arange(n), 2 * M).reshape(n, 2 * M)
    assert not (a0 == rows).any()  # no self loops
    srt = np.sort(a0, axis=1)
    assert not ((srt[:, 1:] == srt[:, :-1]) & (srt[:, 1:] != 0xFFFFFFFF)).any()  # no duplicates
    # levels follow the shared hash
    assert (g["levels"] == np.array([po.lib().orc_level(0x5EED0003, i, M) for i in range(n)], np.uint8)).all()
    # the oracle searching the GPU-built graph agrees bit for bit with the GPU searching it
    G = po.Graph.from_arrays(X, M, 2 * M, g["max_level"], g["entry"], g["levels"], g["upper_off"], a0, g["adjU"])
    gk = _assert_same(po, G, s, Q, 10, 64)
    truth = po.exact_topk(X, Q, 10)
    assert recall_at_k(gk, truth) >= 0.95
    s.close()


def test_gpu_built_vamana(la, po, gpu, tmp_path):
    n, d, R = 10000, 128, 32
    X = synth(po, n, d)
    Q = synth(po, 100, d, stream=1)
    stem = str(tmp_path / "documents.leann")
    la.BackendBuilder(la.BackendType.DiskAnn).build(X, [], stem, d, R, 64)
    s = la.DiskAnnSearcher.load(stem, d)
    g = s.graph_export()
    assert g["max_level"] == 0 and g["M0"] == R
    G = po.Graph.from_arrays(X, R, R, 0, g["entry"], g["levels"], g["upper_off"], g["adj0"], g["adjU"])
    gk = _assert_same(po, G, s, Q, 10, 64, algo=1)
    assert recall_at_k(gk, po.exact_topk(X, Q, 10)) >= 0.9
    with pytest.raises(la.LeannError, match="does not support incremental"):
        la.BackendBuilder(la.BackendType.DiskAnn).add_to_index(X[:10], stem, d, n)
    s.close()


@pytest.mark.parametrize("knobs", [{"LEANN_VAMANA_TWO_STAGE": "0"}, {"LEANN_VAMANA_NAV": "1"}, {"LEANN_VAMANA_PASSES": "2", "LEANN_VAMANA_ALPHA1_PCT": "100"},
                                   {"LEANN_VAMANA_PENDING": "0"}])
def test_vamana_builder_knobs_produce_searchable_graphs(la, po, gpu, monkeypatch, tmp_path, knobs):
    """the construction knobs behind profiles/r03_vamana_scale.md (one-stage prune, entry layers, DiskANN's two-pass schedule, strict
    back-edges): each builds a valid graph — the kernel's walk == the oracle's on the exported arrays, levels included — saves, reloads
    and finds its neighbours."""
    for kname, v in knobs.items():
        monkeypatch.setenv(kname, v)  # (read once per build)
    n, d, R = 6000, 96, 16
    X = synth(po, n, d)
    Q = synth(po, 64, d, stream=1)
    dX = la.DeviceArray.from_host(X)
    s = la.BackendSearcher.build_device(la.BackendType.DiskAnn, dX.ptr, n, d, d, R, 48)
    g = s.graph_export()
    assert (g["max_level"] > 0) == ("LEANN_VAMANA_NAV" in knobs) and g["M0"] == R
    G = po.Graph.from_arrays(X, g["M"], g["M0"], g["max_level"], g["entry"], g["levels"], g["upper_off"], g["adj0"], g["adjU"])
    gk = _assert_same(po, G, s, Q, 10, 48, algo=1)
    assert recall_at_k(gk, po.exact_topk(X, Q, 10)) >= 0.9
    stem = str(tmp_path / "documents.leann")
    s.save(stem)
    s2 = la.DiskAnnSearcher.load(stem, d)
    k2, d2, _ = s2.search_batch(Q, 10, 48)
    assert (k2 == gk).all()
    s.close(); s2.close()


def test_scan_topk_matches_recompute_restatement(la, po, gpu):
    n, d, nq, k = 5000, 768, 9, 10
    X = synth(po, n, d)
    Q = synth(po, nq, d, stream=1)
    dX, dQ = la.DeviceArray.from_host(X), la.DeviceArray.from_host(Q)
    dk, ds, dc = la.DeviceArray((nq, k), np.uint64), la.DeviceArray((nq, k), np.float32), la.DeviceArray(nq, np.uint32)
    la._native.check(la.lib().leann_scan_topk_device(dX.ptr, n, d, d, dQ.ptr, nq, k, None, 0, dk.ptr, ds.ptr, dc.ptr, None))
    la.sync()
    gk, gs, gc = dk.to_host(), ds.to_host(), dc.to_host()
    assert (gc == k).all()
    for i in range(nq):
        k1, s1 = po.scan_topk(X, Q[i], k, mode=1)  # k-ordered fmaf chain: bit-exact
        assert (gk[i] == k1).all() and (gs[i].view(np.uint32) == s1.view(np.uint32)).all()
        k0, s0 = po.scan_topk(X, Q[i], k, mode=0)  # literal recompute.rs:137-139: within 1e-5
        assert np.abs(gs[i] - s0).max() <= 1e-5
        assert set(gk[i].tolist()) == set(k0.tolist()) or np.abs(np.sort(s0) - np.sort(gs[i])).max() <= 1e-5


def test_scan_topk_allow_mask_and_ties(la, po, gpu):
    base = synth(po, 300, 128)
    X = np.concatenate([base, base])  # ties: the lower position must win (stable sort, N4)
    Q = base[:4].copy()
    n, d, nq, k = X.shape[0], 128, 4, 6
    mask = np.zeros((n + 7) // 8, np.uint8)
    allowed = np.arange(n) % 3 != 0
    for i in np.nonzero(allowed)[0]:
        mask[i >> 3] |= 1 << (i & 7)
    dX, dQ, dM = la.DeviceArray.from_host(X), la.DeviceArray.from_host(Q), la.DeviceArray.from_host(mask)
    dk, ds, dc = la.DeviceArray((nq, k), np.uint64), la.DeviceArray((nq, k), np.float32), la.DeviceArray(nq, np.uint32)
    la._native.check(la.lib().leann_scan_topk_device(dX.ptr, n, d, d, dQ.ptr, nq, k, dM.ptr, 0, dk.ptr, ds.ptr, dc.ptr, None))
    la.sync()
    gk, gs = dk.to_host(), ds.to_host()
    for i in range(nq):
        k1, s1 = po.scan_topk(X, Q[i], k, mode=1, allow_mask=mask)
        assert (gk[i] == k1).all() and (gs[i] == s1).all()


def test_scan_topk_candidate_emission_and_overflow(la, po, gpu, monkeypatch):
    """> 64k rows: the launches after the first slab emit only scores that reach the running k-th best (no score slab, no segment
    sorts).  Bit-exact against the oracle's k-ordered dot (duplicated rows: ties -> lower position), equal to the slab path, with an
    allow mask, and with an adversarial order (scores rise with the position -> the lists overflow -> the call repeats on slabs)."""
    n, d, nq, k = 200000, 128, 70, 10
    base = synth(po, n // 2, d)
    X = np.concatenate([base, base])
    Q = synth(po, nq, d, stream=1)
    mask = np.zeros((n + 7) // 8, np.uint8)
    idx = np.arange(1, n, 3)
    np.bitwise_or.at(mask, idx >> 3, (1 << (idx & 7)).astype(np.uint8))
    dX, dQ, dM = la.DeviceArray.from_host(X), la.DeviceArray.from_host(Q), la.DeviceArray.from_host(mask)

    def run(m=None, rows=dX):
        dk, ds, dc = la.DeviceArray((nq, k), np.uint64), la.DeviceArray((nq, k), np.float32), la.DeviceArray(nq, np.uint32)
        la._native.check(la.lib().leann_scan_topk_device(rows.ptr, n, d, d, dQ.ptr, nq, k, m.ptr if m is not None else None, 7, dk.ptr, ds.ptr,
                                                         dc.ptr, None))
        la.sync()
        return dk.to_host(), ds.to_host(), dc.to_host()
    gk, gs, gc = run()
    mk, ms, mc = run(dM)
    for i in (0, 31, 32, 69):
        k1, s1 = po.scan_topk(X, Q[i], k, mode=1)
        assert (gk[i] - 7 == k1).all() and (gs[i].view(np.uint32) == s1.view(np.uint32)).all()
        k2, s2 = po.scan_topk(X, Q[i], k, mode=1, allow_mask=mask)
        assert (mk[i] - 7 == k2).all() and (ms[i].view(np.uint32) == s2.view(np.uint32)).all()
    # adversarial order: rows sorted by their score against query 0, ascending
    order = np.argsort(X @ Q[0], kind="stable")
    dXs = la.DeviceArray.from_host(X[order])
    ak, as_, ac = run(rows=dXs)
    monkeypatch.setenv("LEANN_DEBUG_NO_EMIT", "1")
    la.lib().leann_debug_reload_env()
    sk, ss, sc = run()
    assert (sk == gk).all() and (ss.view(np.uint32) == gs.view(np.uint32)).all()
    bk, bs, bc = run(rows=dXs)
    assert (ak == bk).all() and (as_.view(np.uint32) == bs.view(np.uint32)).all()
    assert (ak[0] - 7 >= n - 40).all()  # query 0's winners sit at the very end
    monkeypatch.delenv("LEANN_DEBUG_NO_EMIT")
    la.lib().leann_debug_reload_env()
    # a deep list (k = 300): the candidate lists are sized with k
    dk, ds, dc = la.DeviceArray((4, 300), np.uint64), la.DeviceArray((4, 300), np.float32), la.DeviceArray(4, np.uint32)
    la._native.check(la.lib().leann_scan_topk_device(dX.ptr, n, d, d, dQ.ptr, 4, 300, None, 0, dk.ptr, ds.ptr, dc.ptr, None))
    la.sync()
    k1, s1 = po.scan_topk(X, Q[2], 300, mode=1)
    assert (dk.to_host()[2] == k1).all() and (ds.to_host()[2].view(np.uint32) == s1.view(np.uint32)).all()


def test_merge_topk_matches_oracle(la, po, gpu):
    rng = np.random.default_rng(5)
    S, nq, k_in, k_out = 8, 33, 10, 10
    keys = rng.permutation(S * nq * k_in).astype(np.uint64).reshape(S, nq, k_in)
    dists = rng.integers(0, 40, (S, nq, k_in)).astype(np.float32) / 8  # many ties
    for s_ in range(S):  # valid inputs: each list ascending by (dist, key)
        for q in range(nq):
            o = np.lexsort((keys[s_, q], dists[s_, q]))
            keys[s_, q], dists[s_, q] = keys[s_, q][o], dists[s_, q][o]
    counts = rng.integers(0, k_in + 1, (S, nq)).astype(np.uint32)
    dk, dd, dc = la.DeviceArray.from_host(keys), la.DeviceArray.from_host(dists), la.DeviceArray.from_host(counts)
    ok, od, oc = la.DeviceArray((nq, k_out), np.uint64), la.DeviceArray((nq, k_out), np.float32), la.DeviceArray(nq, np.uint32)
    la._native.check(la.lib().leann_merge_topk_device(dk.ptr, dd.ptr, dc.ptr, S, nq, k_in, k_out, 0, ok.ptr, od.ptr, oc.ptr, None))
    la.sync()
    gk, gd, gc = ok.to_host(), od.to_host(), oc.to_host()
    for q in range(nq):
        rk, rd = po.merge_topk(keys[:, q], dists[:, q], counts[:, q], k_out)
        m = len(rk)
        assert int(gc[q]) == m
        assert (gk[q, :m] == rk).all() and (gd[q, :m] == rd).all()
        assert (gk[q, m:] == np.iinfo(np.uint64).max).all()


def test_open_errors(la, gpu, tmp_path):
    stem = str(tmp_path / "documents.leann")
    with pytest.raises(la.LeannError, match="Index file not found"):
        la.HnswSearcher.load(stem, 128)
    with pytest.raises(la.LeannError, match="DiskANN index not found"):
        la.DiskAnnSearcher.load(stem, 128)
    (tmp_path / "documents.index").write_bytes(b"IxHN" + b"\0" * 64)
    with pytest.raises(la.LeannError, match="Python LEANN"):
        la.HnswSearcher.load(stem, 128)
    (tmp_path / "documents.index").write_bytes(b"garbage!" * 32)
    with pytest.raises(la.LeannError, match="incompatible format"):
        la.HnswSearcher.load(stem, 128)


def test_regression_pins_gpu(la, po, gpu):
    """The HIP generator, level hash and traversal reproduce tests/golden/pins_v1.npz bit for bit."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pins_v1.npz"))
    for name, (d, r, C_, sig, stream, i0) in {"c768": (768, 64, 4096, 1.0, 0, 0), "q768": (768, 64, 4096, 1.0, 1, 7),
                                               "iid128": (128, 0, 1, 0.0, 0, 3), "c1536": (1536, 64, 4096, 1.0, 0, 10 ** 7)}.items():
        buf = la.DeviceArray((3, d), np.float32)
        la._native.check(la.lib().leann_synth_rows_device(SEED, d, d, r, C_, sig, stream, i0, 3, buf.ptr, None))
        la.sync()
        assert (buf.to_host().view(np.uint32)[:, :16] == z["gen_" + name]).all(), name
    X = po.gen_rows(SEED, 96, 32, 64, 1.0, 0, 0, 2000)
    Q = po.gen_rows(SEED, 96, 32, 64, 1.0, 1, 0, 16)
    G = po.Graph.build_hnsw(X, M=8, efc=32)
    lv, uo, a0, aU = G.export()
    s = la.BackendSearcher.from_arrays(la.BackendType.Hnsw, X, 8, 16, G.max_level, G.entry, lv, uo, a0, aU)
    s.stats(reset=True)
    k, dd, c = s.search_batch(Q, 5, 24)
    assert (k == z["hnsw_keys"]).all() and (dd.view(np.uint32) == z["hnsw_dists"]).all()
    st = s.stats()
    assert st["n_dist_evals"] == int(z["hnsw_stats"][:, 0].sum()) and st["n_hops_base"] == int(z["hnsw_stats"][:, 1].sum())
    s.close()


def test_regression_pins_v2_gpu(la, po, gpu):
    """The HIP path reproduces tests/golden/pins_v2.npz: Vamana GreedySearch and the filtered search bit for bit (ids, f32 distance bits,
    counters), the synthetic recompute inputs bit for bit, embeddings within 1e-6, the cross-shard merge bit for bit."""
    import ctypes as C
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pins_v2.npz"))
    X = po.gen_rows(SEED, 96, 32, 64, 1.0, 0, 0, 2000)
    Q = po.gen_rows(SEED, 96, 32, 64, 1.0, 1, 0, 16)
    V = po.Graph.build_vamana(X, R=12, L=32, alpha=1.2, two_stage=False)  # the pins predate the two-stage prune
    lv, uo, a0, aU = V.export()
    assert int(a0.astype(np.uint64).sum()) == int(z["vam_graph"][0]) and V.entry == int(z["vam_graph"][1])
    s = la.BackendSearcher.from_arrays(la.BackendType.DiskAnn, X, 12, 12, 0, V.entry, lv, uo, a0, aU)
    for L_ in (8, 40):
        s.stats(reset=True)
        k, dd, c = s.search_batch(Q, 6, L_)
        st = s.stats()
        assert (k == z[f"vam_keys_L{L_}"]).all() and (dd.view(np.uint32) == z[f"vam_dists_L{L_}"]).all()
        assert st["n_dist_evals"] == int(z[f"vam_stats_L{L_}"][:, 0].sum()) and st["n_hops_base"] == int(z[f"vam_stats_L{L_}"][:, 1].sum())
    s.close()
    G = po.Graph.build_hnsw(X, M=8, efc=32)
    lv, uo, a0, aU = G.export()
    s = la.BackendSearcher.from_arrays(la.BackendType.Hnsw, X, 8, 16, G.max_level, G.entry, lv, uo, a0, aU)
    allow = np.packbits((np.arange(2000) % 3) == 0, bitorder="little")
    k, dd, c = s.search_filtered_batch(Q, 5, 24, allow)
    assert (k == z["filt_keys"]).all() and (dd.view(np.uint32) == z["filt_dists"]).all() and (c == z["filt_counts"]).all()
    s.close()
    L, chk = la.lib(), la._native.check
    gF, gW = la.DeviceArray((40, 64), np.uint16), la.DeviceArray((64, 96), np.uint16)
    chk(L.leann_synth_features_device(SEED, 64, 0, 16, 1.0, 0, 5, 40, gF.ptr, None))
    chk(L.leann_synth_weights_device(SEED, 64, 96, gW.ptr, None))
    la.sync()
    assert (gF.to_host()[:4] == z["rc_features"]).all() and int(gW.to_host().astype(np.uint64).sum()) == int(z["rc_weights_crc"][0])
    r = C.c_void_p()
    chk(L.leann_recompute_create(gF.ptr, 40, 64, gW.ptr, 96, 0, 0, C.byref(r)))
    dE = la.DeviceArray((40, 96), np.float32)
    chk(L.leann_recompute_encode_device(r, 0, 40, dE.ptr, None))
    la.sync()
    assert np.abs(dE.to_host()[:8] - z["rc_embed"].view(np.float32)).max() <= 1e-6
    L.leann_recompute_close(r)
    mk = np.array([[[5, 9, 40]], [[7, 8, 41]], [[1, 2, 3]]], np.uint64)
    md = np.array([[[0.1, 0.3, 0.5]], [[0.1, 0.2, 0.9]], [[0.4, 0.45, 0.0]]], np.float32)
    mc = np.array([[3], [3], [2]], np.uint32)
    dk, dd_, dc = la.DeviceArray.from_host(mk), la.DeviceArray.from_host(md), la.DeviceArray.from_host(mc)
    ok_, od_, oc_ = la.DeviceArray((1, 5), np.uint64), la.DeviceArray((1, 5), np.float32), la.DeviceArray(1, np.uint32)
    chk(L.leann_merge_topk_device(dk.ptr, dd_.ptr, dc.ptr, 3, 1, 3, 5, 0, ok_.ptr, od_.ptr, oc_.ptr, None))
    la.sync()
    assert (ok_.to_host()[0] == z["merge_keys"]).all() and (od_.to_host()[0].view(np.uint32) == z["merge_dists"]).all()


def test_traced_three_level_graph_with_ties_gpu(la, gpu):
    """tests/golden/traced_graph_400.json through the HIP kernel (no oracle involved): ids, distances and the visit counters of the
    independent pure-Python trace — three levels, duplicated vectors (ties -> lower id), k > ef, single queries (16-wave latency form)
    and one batch (same form below 512 queries), HNSW and GreedySearch."""
    from util import traced_graph
    fx, X, levels, upper_off, adj0, adjU = traced_graph("traced_graph_400.json")
    s = la.BackendSearcher.from_arrays(la.BackendType.Hnsw, X, fx["M"], fx["M0"], fx["max_level"], fx["entry"], levels, upper_off, adj0, adjU)
    for c in fx["cases"]:
        q = np.array(c["query"], np.float32)
        s.stats(reset=True)
        keys, dists = s.search(q, c["k"], c["ef"])
        st = s.stats()
        assert keys.tolist() == c["ids"] and dists.tolist() == [float(x) for x in c["dists"]], (c["k"], c["ef"])
        assert st["n_dist_evals"] == c["n_evals"] and st["n_hops_base"] == len(c["expanded_base"]) and st["n_hops_upper"] == c["hops_upper"]
    # the throughput form of the hop loop (4 waves per query): replicate one case 600 times so that the batch exceeds 512 queries
    c = fx["cases"][0]
    Q = np.tile(np.array(c["query"], np.float32), (600, 1))
    k, dd, cnt = s.search_batch(Q, c["k"], c["ef"])
    assert (k == np.array(c["ids"], np.uint64)).all() and (dd == np.array(c["dists"], np.float32)).all()
    s.close()
    v = la.BackendSearcher.from_arrays(la.BackendType.DiskAnn, X, fx["M0"], fx["M0"], 0, fx["entry"], np.zeros(fx["n"], np.uint8),
                                       np.zeros(fx["n"], np.uint32), adj0, np.zeros((0, fx["M0"]), np.uint32))
    for c in fx["vamana_cases"]:
        v.stats(reset=True)
        keys, dists = v.search(np.array(c["query"], np.float32), c["k"], c["L"])
        st = v.stats()
        assert keys.tolist() == c["ids"] and dists.tolist() == [float(x) for x in c["dists"]]
        assert st["n_dist_evals"] == c["n_evals"] and st["n_hops_base"] == len(c["expanded"])
    v.close()
