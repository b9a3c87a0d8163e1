"""-m gpu: the batched hybrid leg (csrc/hybrid.hip: BM25-only injection + hybrid_rerank of src/index/searcher.rs:146-169 and
src/index/bm25.rs:135-170, one workgroup per query) against oracle/searcher_oracle.py — ids and f32 scores bit for bit, both polarities
(SURVEY.md N1), with the edge cases the reference's own tests name (bm25.rs:283-329: empty vector list, no BM25 match) plus short
lists, ties, an all-positive BM25 vector and a Vamana walk feeding the rerank end to end."""
import numpy as np
import pytest

from util import synth

pytestmark = pytest.mark.gpu
f32 = np.float32
U64MAX = np.iinfo(np.uint64).max


def _oracle(so, keys, dists, pos, sc, n_docs, alpha, compat, top_k, fetch_k):
    """searcher_oracle.search_with_options' hybrid branch on a dense BM25 vector rebuilt from the sparse positives"""
    vr = [(int(k), f32(d) if compat else f32(f32(1.0) - f32(d))) for k, d in zip(keys, dists)]
    dense = np.zeros(n_docs, np.float32)
    dense[pos] = sc
    order = sorted(range(len(pos)), key=lambda t: (-float(sc[t]), int(pos[t])))  # Bm25Scorer::search: score desc, stable by index
    have = {i for i, _ in vr}
    for t in order[:fetch_k]:
        if int(pos[t]) not in have:
            vr.append((int(pos[t]), f32(0.0)))
    return so.hybrid_rerank(vr, dense, alpha)[:top_k]


def _run(la, keys, dists, counts, pos, sc, pcnt, n_docs, alpha, compat, top_k):
    nq, fetch_k = keys.shape
    dk, dd, dc = la.DeviceArray.from_host(keys), la.DeviceArray.from_host(dists), la.DeviceArray.from_host(counts)
    dp, ds, dn = la.DeviceArray.from_host(pos), la.DeviceArray.from_host(sc), la.DeviceArray.from_host(pcnt)
    ok, os_, oc = la.DeviceArray((nq, top_k), np.uint64), la.DeviceArray((nq, top_k), np.float32), la.DeviceArray(nq, np.uint32)
    la._native.check(la.lib().leann_hybrid_rerank_device(dk.ptr, dd.ptr, dc.ptr, nq, fetch_k, dp.ptr, ds.ptr, dn.ptr, pos.shape[1], n_docs,
                                                         alpha, 1 if compat else 0, top_k, ok.ptr, os_.ptr, oc.ptr, None))
    la.sync()
    return ok.to_host(), os_.to_host(), oc.to_host()


def _sparse_bm25(rng, nq, n_docs, stride, ann_keys, ann_counts, max_pos):
    """per query: `cnt` positives, some of them ANN hits, quantised scores (ties), sorted as Bm25Scorer::search sorts"""
    pos = np.full((nq, stride), 0xFFFFFFFF, np.uint32)
    sc = np.zeros((nq, stride), np.float32)
    cnt = np.zeros(nq, np.uint32)
    for q in range(nq):
        c = int(rng.integers(0, max_pos + 1))
        chosen = set()
        if ann_counts[q] and c:
            for k in rng.choice(ann_keys[q, :ann_counts[q]], size=min(c // 3, int(ann_counts[q])), replace=False):
                chosen.add(int(k))
        while len(chosen) < c:
            chosen.add(int(rng.integers(0, n_docs)))
        p = np.array(sorted(chosen), np.uint32)
        s = (rng.integers(1, 40, size=len(p)) * 0.25).astype(np.float32)  # few distinct values: ties exercise the stable order
        o = sorted(range(len(p)), key=lambda t: (-float(s[t]), int(p[t])))
        pos[q, :len(p)], sc[q, :len(p)], cnt[q] = p[o], s[o], len(p)
    return pos, sc, cnt


@pytest.mark.parametrize("compat", [True, False])
@pytest.mark.parametrize("top_k,alpha", [(10, 0.7), (5, 0.3), (20, 1.0), (3, 0.0)])
def test_hybrid_rerank_matches_the_oracle(la, po, gpu, compat, top_k, alpha):
    import searcher_oracle as so
    rng = np.random.default_rng(1000 + top_k)
    nq, n_docs, fetch_k, stride = 200, 5000, 5 * top_k, 96
    keys = np.full((nq, fetch_k), U64MAX, np.uint64)
    dists = np.full((nq, fetch_k), np.inf, np.float32)
    counts = np.zeros(nq, np.uint32)
    for q in range(nq):
        c = fetch_k if q % 5 else int(rng.integers(0, fetch_k))  # short results (searcher.rs:139-143 zips whatever came back), empty ones
        keys[q, :c] = rng.choice(n_docs, size=c, replace=False)
        d = np.sort(rng.uniform(0.02, 1.3, size=c)).astype(np.float32)
        if q % 7 == 0 and c > 4:
            d[1:4] = d[1]  # equal distances
        dists[q, :c], counts[q] = d, c
    pos, sc, pcnt = _sparse_bm25(rng, nq, n_docs, stride, keys, counts, stride)
    pcnt[3] = 0  # no BM25 match at all (bm25.rs:311-329)
    gk, gs, gc = _run(la, keys, dists, counts, pos, sc, pcnt, n_docs, alpha, compat, top_k)
    for q in range(nq):
        exp = _oracle(so, keys[q, :counts[q]], dists[q, :counts[q]], pos[q, :pcnt[q]], sc[q, :pcnt[q]], n_docs, alpha, compat, top_k, fetch_k)
        assert gc[q] == len(exp), q
        assert [int(x) for x in gk[q, :gc[q]]] == [i for i, _ in exp], q
        assert (gs[q, :gc[q]].view(np.uint32) == np.array([s for _, s in exp], np.float32).view(np.uint32)).all(), q
        assert (gk[q, gc[q]:] == U64MAX).all()


def test_every_passage_positive_moves_min_b_off_zero(la, po, gpu):
    """when all n_docs passages are BM25-positive the minimum of the score vector is the smallest positive, not 0.0 (bm25.rs:152-154)"""
    import searcher_oracle as so
    rng = np.random.default_rng(5)
    n_docs, fetch_k, top_k = 40, 25, 5
    keys = rng.choice(n_docs, size=(1, fetch_k), replace=False).astype(np.uint64)
    dists = np.sort(rng.uniform(0.1, 0.9, size=(1, fetch_k)).astype(np.float32), axis=1)
    counts = np.array([fetch_k], np.uint32)
    s = rng.uniform(2.0, 9.0, size=n_docs).astype(np.float32)
    o = sorted(range(n_docs), key=lambda t: (-float(s[t]), t))
    pos, sc, pcnt = np.array([o], np.uint32), s[o][None, :].copy(), np.array([n_docs], np.uint32)
    gk, gs, gc = _run(la, keys, dists, counts, pos, sc, pcnt, n_docs, 0.6, True, top_k)
    exp = _oracle(so, keys[0], dists[0], pos[0], sc[0], n_docs, 0.6, True, top_k, fetch_k)
    assert [int(x) for x in gk[0]] == [i for i, _ in exp]
    assert (gs[0].view(np.uint32) == np.array([x for _, x in exp], np.float32).view(np.uint32)).all()


def test_vamana_walk_then_hybrid_rerank_end_to_end(la, po, gpu):
    """configs[4] in small: Vamana R = 32 at d = 1536 searched with fetch_k = 5 k on the device, reranked on the device, against the
    oracle walking the same graph and searcher_oracle doing the rerank."""
    import searcher_oracle as so
    n, d, R, k, L = 6000, 1536, 32, 10, 64
    X = synth(po, n, d)
    Q = synth(po, 64, d, stream=1)
    dX = la.DeviceArray.from_host(X)
    s = la.BackendSearcher.build_device(la.BackendType.DiskAnn, dX.ptr, n, d, d, R, 96)
    g = s.graph_export()
    G = po.Graph.from_arrays(X, g["M"], g["M0"], g["max_level"], g["entry"], g["levels"], g["upper_off"], g["adj0"], g["adjU"])
    fetch_k = 5 * k
    ok, od, oc, _ = G.search_batch(Q, fetch_k, L, 1, 4)
    gk, gd, gc = s.search_batch(Q, fetch_k, L)
    assert (gk == ok).all() and (gd == od).all()
    rng = np.random.default_rng(9)
    pos, sc, pcnt = _sparse_bm25(rng, len(Q), n, 64, gk, gc, 64)
    for compat in (True, False):
        rk, rs, rc = _run(la, gk, gd, gc, pos, sc, pcnt, n, 0.7, compat, k)
        for q in range(len(Q)):
            exp = _oracle(so, ok[q, :oc[q]], od[q, :oc[q]], pos[q, :pcnt[q]], sc[q, :pcnt[q]], n, 0.7, compat, k, fetch_k)
            assert [int(x) for x in rk[q, :rc[q]]] == [i for i, _ in exp]
            assert (rs[q, :rc[q]].view(np.uint32) == np.array([x for _, x in exp], np.float32).view(np.uint32)).all()
    s.close()


def test_bad_arguments(la, gpu):
    z = la.DeviceArray(16, np.uint64)
    with pytest.raises(la.LeannError, match="fetch_k"):
        la._native.check(la.lib().leann_hybrid_rerank_device(z.ptr, z.ptr, z.ptr, 1, 300, z.ptr, z.ptr, z.ptr, 4, 10, 0.7, 1, 10, z.ptr, z.ptr, z.ptr, None))
    with pytest.raises(la.LeannError, match="alpha"):
        la._native.check(la.lib().leann_hybrid_rerank_device(z.ptr, z.ptr, z.ptr, 1, 50, z.ptr, z.ptr, z.ptr, 4, 10, 1.5, 1, 10, z.ptr, z.ptr, z.ptr, None))
