"""CPU suite (-m "not gpu"): the oracle against the committed golden vectors and the reference's own
test assertions; no GPU needed."""
import json
import os

import numpy as np
import pytest

from util import recall_at_k, synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
G = json.load(open(os.path.join(GOLD, "reference_formulas.json")))
f32 = np.float32


def _ulps(a, b):
    return abs(int(f32(a).view(np.int32)) - int(f32(b).view(np.int32)))


# ---- dot_product: src/index/recompute.rs:137-139 on the vectors of benches/benchmarks.rs:28-31,43-45 ----
@pytest.mark.parametrize("dims", [1536, 768])
def test_dot_seq_golden(po, dims):
    a = (np.arange(dims, dtype=f32) / f32(1000.0)).astype(f32)
    assert po.dot(a, a, "seq") == G[f"dot_seq_{dims}"]  # bit exact
    exact = float((a.astype(np.float64) ** 2).sum())
    for kind in ("canon", "canon_ref", "seqfma", "fast"):
        assert abs(po.dot(a, a, kind) - exact) <= 1e-5 * exact


def test_dot_canon_matches_definition_bitwise(po):
    rng = np.random.default_rng(7)
    for d in (1, 3, 4, 5, 100, 128, 255, 256, 257, 300, 768, 1000, 1536, 2048):
        for scale in (1e-3, 1.0, 1e3):
            a = (rng.standard_normal(d) * scale).astype(f32)
            b = rng.standard_normal(d).astype(f32)
            assert f32(po.dot(a, b, "canon")).view(np.uint32) == f32(po.dot(a, b, "canon_ref")).view(np.uint32)
            # numpy restatement of the definition: 256 strided fmaf chains + adjacent-pair tree
            acc = np.zeros(256, np.float64)  # fmaf == round(a*b+c): emulate with float64 (exact product) then round
            accf = np.zeros(256, f32)
            for j in range(d):
                accf[j & 255] = f32(np.float64(a[j]) * np.float64(b[j]) + np.float64(accf[j & 255]))
            v = accf
            while len(v) > 1:
                v = (v[0::2] + v[1::2]).astype(f32)
            if d <= 1536 and scale == 1.0:  # float64 emulation of fmaf is exact unless the sum needs > 53 bits
                assert abs(float(v[0]) - po.dot(a, b, "canon")) <= 1e-6 * max(1.0, abs(float(v[0])))


def test_orderable_roundtrip_and_order(po):
    L = po.lib()
    L.orc_f32_orderable.restype = np.ctypeslib.ctypes.c_uint32
    L.orc_f32_orderable.argtypes = [np.ctypeslib.ctypes.c_float]
    vals = np.array([-np.inf, -3.5, -1e-30, -0.0, 0.0, 1e-30, 0.25, 1.0, 2.0, np.inf], f32)
    keys = [L.orc_f32_orderable(float(v)) for v in vals]
    assert keys == sorted(keys)


# ---- recompute scan: src/index/recompute.rs:96-109 ----
def test_scan_topk_stable_desc_and_mask(po):
    base = synth(po, 200, 64, r=0)
    X = np.concatenate([base, base])  # every score appears twice: the lower position must come first
    q = base[5]
    keys, scores = po.scan_topk(X, q, 8, mode=0)
    assert keys[0] == 5 and keys[1] == 205 and scores[0] == scores[1]
    assert (np.diff(scores) <= 0).all()
    for a, b in zip(keys[0::2], keys[1::2]):
        assert b == a + 200
    mask = np.zeros((400 + 7) // 8, np.uint8)
    for i in range(400):
        if i % 2:
            mask[i >> 3] |= 1 << (i & 7)
    keys2, _ = po.scan_topk(X, q, 4, mode=0, allow_mask=mask)
    assert all(int(k) % 2 == 1 for k in keys2) and keys2[0] == 5
    k0, s0 = po.scan_topk(X, q, 500, mode=0)  # top_k > n: short result
    assert len(k0) == 400


@pytest.mark.parametrize("n", [1000, 10000])
def test_exact_top10_golden(po, n):
    """SURVEY.md §8c golden (1): exact float64 top-10 of the seeded 128-d sets vs the literal f32 restatement."""
    z = np.load(os.path.join(GOLD, f"exact_top10_{n}x128.npz"))
    rng = np.random.default_rng(1234 + n)
    X = rng.standard_normal((n, 128)).astype(f32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    Q = rng.standard_normal((16, 128)).astype(f32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    for i in range(16):
        for mode in (0, 1, 2):
            keys, scores = po.scan_topk(X, Q[i], 10, mode=mode)
            assert np.abs(scores - z["score"][i]).max() <= 1e-5
            gaps_ok = np.abs(np.diff(z["score"][i])).min() > 2e-6
            if gaps_ok:
                assert (keys == z["idx"][i].astype(np.uint64)).all()


# ---- hybrid_rerank: src/index/bm25.rs:135-170 and its tests :283-329 ----
def test_hybrid_rerank_golden(po):
    for c in G["hybrid_rerank"]:
        out = po.hybrid_rerank([tuple(v) for v in c["vr"]], c["bm"], c["alpha"])
        assert [i for i, _ in out] == [i for i, _ in c["out"]]
        assert all(f32(a[1]) == f32(b[1]) for a, b in zip(out, c["out"]))


def test_hybrid_rerank_reference_assertions(po):
    out = po.hybrid_rerank([(0, 0.9), (1, 0.8), (2, 0.7)], [0.5, 0.9, 0.3], 0.5)  # bm25.rs:283-299
    assert len(out) == 3 and all(0.0 <= s <= 1.0 for _, s in out)
    assert po.hybrid_rerank([(0, 0.9), (1, 0.5)], [0.1, 0.9], 1.0)[0][0] == 0  # :302-314
    assert po.hybrid_rerank([(0, 0.9), (1, 0.5)], [0.1, 0.9], 0.0)[0][0] == 1  # :317-329


# ---- BM25: src/index/bm25.rs:33-122 and its tests :176-280 ----
def test_tokenize(po):
    import bm25_oracle as bo
    for text, toks in G["tokenize"].items():
        assert bo.tokenize(text) == toks
    t = bo.tokenize("Hello, World! This is a test.")
    assert "hello" in t and "world" in t and "test" in t and "a" not in t  # bm25.rs:176-184
    assert bo.tokenize("") == []
    t = bo.tokenize("test123 456abc")
    assert "test123" in t and "456abc" in t


def test_bm25_golden_and_reference_assertions(po):
    import bm25_oracle as bo
    for name, c in G["bm25"].items():
        s = bo.Bm25Scorer.build(c["docs"]).score_query(c["query"])
        assert all(_ulps(a, b) <= 1 for a, b in zip(s, c["scores"])), name
    sc = bo.Bm25Scorer.build(["rust rust rust programming", "rust programming"]).score_query("rust")
    assert sc[0] > sc[1]  # :219-227
    sc = bo.Bm25Scorer.build(["common rare", "common", "common"]).score_query("rare")
    assert sc[0] > 0 and sc[1] == 0 and sc[2] == 0  # :230-244
    assert bo.Bm25Scorer.build(["hello world"]).score_query("")[0] == 0  # :247-253
    assert bo.Bm25Scorer.build(["hello world"]).search("xyz", 5) == []  # :256-262
    r = bo.Bm25Scorer.build(["apple banana", "apple cherry", "banana cherry", "apple apple apple"]).search("apple", 2)
    assert len(r) == 2 and r[0][0] == 3  # :265-280
    r = bo.Bm25Scorer.build(["the quick brown fox jumps over the lazy dog", "a quick brown dog outpaces a swift fox",
                             "the dog chases the fox around the yard"]).search("quick fox", 3)
    assert 0 < len(r) <= 3  # :200-216


# ---- graph search restatement ----
def test_hnsw_two_formulations_agree_and_recall(po):
    X = synth(po, 4000, 96)
    Q = synth(po, 100, 96, stream=1)
    Gr = po.Graph.build_hnsw(X, M=12, efc=48)
    truth = po.exact_topk(X, Q, 10)
    for ef in (1, 10, 40, 100):
        k0, d0, c0, s0 = Gr.search_batch(Q, 10, ef, 0, 4)
        k1, d1, c1, s1 = Gr.search_batch(Q, 10, ef, 1, 2)
        assert (k0 == k1).all() and (d0.view(np.uint32) == d1.view(np.uint32)).all() and (s0 == s1).all()
    assert recall_at_k(k0, truth) >= 0.95
    assert (np.diff(d0, axis=1) >= 0).all()  # best first
    # the stored vector is its own nearest neighbour (dist 1 - <x,x> ~ 0)
    keys, dists, _ = Gr.search(X[123], 1, 32)
    assert keys[0] == 123 and abs(dists[0]) < 1e-6


def test_filtered_search_restatement(po):
    """Filtered search: same walk, answer = best allowed keys among everything evaluated on level 0.
    Properties: both formulations agree; all-ones bitmap == unfiltered; only allowed ids, ascending; at least as good as
    post-filtering the beam; no reference behaviour to pin (searcher.rs post-filters) -> parity unpinned."""
    n, d, k, ef = 4000, 128, 10, 48
    rng = np.random.default_rng(5)
    X = synth(po, n, d)
    Q = synth(po, 50, d, stream=1)
    G = po.Graph.build_hnsw(X, M=16, efc=64)
    ones = np.full((n + 7) // 8, 0xFF, np.uint8)
    uk, ud, uc, ust = G.search_batch(Q, k, ef, 0)
    fk, fd, fc, fst = G.search_filtered_batch(Q, k, ef, ones, 0)
    assert (fk == uk).all() and (fd == ud).all() and (fc == uc).all() and (fst == ust).all()
    allowed = rng.random(n) < 0.1
    bm = np.packbits(allowed, bitorder="little")
    k0, d0, c0, s0 = G.search_filtered_batch(Q, k, ef, bm, 0)
    k1, d1, c1, s1 = G.search_filtered_batch(Q, k, ef, bm, 1, nthreads=4)
    assert (k0 == k1).all() and (d0 == d1).all() and (c0 == c1).all() and (s0 == s1).all()
    assert (s0 == ust).all()                                   # the walk itself is the unfiltered one
    bk, bd, bc, _ = G.search_batch(Q, ef, ef, 0)               # the whole beam
    truth = np.argsort(-(Q @ X[allowed].T), axis=1, kind="stable")[:, :k]
    truth = np.nonzero(allowed)[0][truth]
    hits = 0
    for i in range(len(Q)):
        ids = k0[i, : c0[i]].astype(np.int64)
        assert allowed[ids].all() and (np.diff(d0[i, : c0[i]]) >= 0).all()
        beam_ok = [j for j in range(bc[i]) if allowed[int(bk[i, j])]][:k]
        assert c0[i] >= len(beam_ok)
        for t, j in enumerate(beam_ok):                        # never worse than post-filtering the beam
            assert d0[i, t] <= bd[i, j]
        hits += len(set(ids.tolist()) & set(truth[i].tolist()))
    assert hits / (len(Q) * k) > 0.6
    # one bitmap per query
    per = np.packbits(rng.random((len(Q), n)) < 0.2, axis=1, bitorder="little")
    pk, pd, pc, _ = G.search_filtered_batch(Q, k, ef, per, 0)
    for i in (0, 17, 49):
        sk, sd, sc, _ = G.search_filtered_batch(Q[i : i + 1], k, ef, per[i], 0)
        assert (sk[0] == pk[i]).all() and sc[0] == pc[i]


def test_hand_built_graph_traced_visit_order(po):
    """SURVEY 8c fixture (4): 64-node hand-built two-level graph over integer vectors; ids, distances, expansion and evaluation
    counts traced by an independent pure-Python textbook HNSW (tests/golden/make_traced_graph.py) - both oracle formulations."""
    from util import traced_graph
    fx, X, levels, upper_off, adj0, adjU = traced_graph()
    G = po.Graph.from_arrays(X, fx["M"], fx["M0"], fx["max_level"], fx["entry"], levels, upper_off, adj0, adjU)
    for c in fx["cases"]:
        q = np.array(c["query"], np.float32)
        for algo in (0, 1):
            keys, dists, st = G.search(q, c["k"], c["ef"], algo)
            assert keys.tolist() == c["ids"] and dists.tolist() == [float(x) for x in c["dists"]]
            assert int(st[0]) == c["n_evals"] and int(st[1]) == len(c["expanded_base"]) and int(st[2]) == len(c["expanded_upper"])
    # the level-0 graph alone as a Vamana index: GreedySearch (DiskANN Alg. 1) traced independently
    V = po.Graph.from_arrays(X, fx["M0"], fx["M0"], 0, 0, np.zeros(fx["n"], np.uint8), np.zeros(fx["n"], np.uint32), adj0,
                             np.zeros((0, fx["M0"]), np.uint32))
    for c in fx["vamana_cases"]:
        q = np.array(c["query"], np.float32)
        for algo in (0, 1):
            keys, dists, st = V.search(q, c["k"], c["L"], algo)
            assert keys.tolist() == c["ids"] and dists.tolist() == [float(x) for x in c["dists"]]
            assert int(st[0]) == c["n_evals"] and int(st[1]) == len(c["expanded"])


def test_config0_plumbing_10k_x_128(po):
    """BASELINE configs[0]: 10k x 128 random f32 vectors, HNSW ef=64 (CPU, plumbing)."""
    X = synth(po, 10000, 128, r=0)
    Q = synth(po, 20, 128, stream=1, r=0)
    Gr = po.Graph.build_hnsw(X, M=32, efc=64)
    keys, dists, counts, stats = Gr.search_batch(Q, 5, 64, 0, 8)
    assert (counts == 5).all() and (keys < 10000).all() and (np.diff(dists, axis=1) >= 0).all()
    lv, uo, a0, aU = Gr.export()
    G2 = po.Graph.from_arrays(X, 32, 64, Gr.max_level, Gr.entry, lv, uo, a0, aU)
    k2, d2, _, _ = G2.search_batch(Q, 5, 64, 0, 1)
    assert (k2 == keys).all() and (d2 == dists).all()


def test_vamana_restatement(po):
    X = synth(po, 3000, 64)
    Q = synth(po, 100, 64, stream=1)
    Gr = po.Graph.build_vamana(X, R=24, L=48, alpha=1.2)
    assert Gr.max_level == 0 and Gr.M0 == 24
    k1, d1, _, _ = Gr.search_batch(Q, 10, 64, 1, 4)
    k0, d0, _, _ = Gr.search_batch(Q, 10, 64, 0, 4)
    assert (k0 == k1).all()
    assert recall_at_k(k1, po.exact_topk(X, Q, 10)) >= 0.9
    kk, dd, _ = Gr.search(Q[0], 10, 4)  # beam = max(complexity, top_k): diskann.rs:54
    assert len(kk) == 10


def test_levels_follow_geometric_law(po):
    L = po.lib()
    lv = np.array([L.orc_level(3, i, 32) for i in range(200000)])
    c = np.bincount(lv)
    assert abs(c[1] / c[0] - 1 / 32) < 0.004 and c[0] > 190000


def test_merge_topk_oracle(po):
    rng = np.random.default_rng(0)
    S, k = 4, 6
    keys = rng.permutation(S * k).astype(np.uint64).reshape(S, k)
    dists = rng.integers(0, 5, (S, k)).astype(f32)
    for s in range(S):
        o = np.lexsort((keys[s], dists[s]))
        keys[s], dists[s] = keys[s][o], dists[s][o]
    counts = np.array([6, 0, 3, 6], np.uint32)
    mk, md = po.merge_topk(keys, dists, counts, 8)
    allp = sorted((float(dists[s, j]), int(keys[s, j])) for s in range(S) for j in range(counts[s]))
    assert [(float(d), int(kk)) for d, kk in zip(md, mk)] == allp[:8]


def test_oracle_under_address_and_ub_sanitizers():
    """Sanitizers run on the CPU build only (GPU ASan is unavailable on this pool)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "selfcheck"])
    out = subprocess.run([os.path.join(root, "oracle", "selfcheck")], capture_output=True, text=True,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert out.returncode == 0 and "selfcheck OK" in out.stdout, out.stderr[-2000:]


def test_regression_pins_oracle(po):
    """tests/golden/pins_v1.npz: bits this implementation produced when GPU == oracle was first established."""
    z = np.load(os.path.join(GOLD, "pins_v1.npz"))
    SEED = 0x5EED0001
    for name, (d, r, C_, sig, stream, i0) in {"c768": (768, 64, 4096, 1.0, 0, 0), "q768": (768, 64, 4096, 1.0, 1, 7),
                                               "iid128": (128, 0, 1, 0.0, 0, 3), "c1536": (1536, 64, 4096, 1.0, 0, 10 ** 7)}.items():
        assert (po.gen_rows(SEED, d, r, C_, sig, stream, i0, 3).view(np.uint32)[:, :16] == z["gen_" + name]).all(), name
    X = po.gen_rows(SEED, 96, 32, 64, 1.0, 0, 0, 2000)
    Q = po.gen_rows(SEED, 96, 32, 64, 1.0, 1, 0, 16)
    assert (np.array([po.dot(Q[i], X[i], "canon") for i in range(16)], f32).view(np.uint32) == z["dot_canon"]).all()
    assert (np.array([po.lib().orc_level(0x5EED0003, i, 32) for i in range(4096)], np.uint8) == z["levels_M32"]).all()
    Gr = po.Graph.build_hnsw(X, M=8, efc=32)
    k, dd, c, st = Gr.search_batch(Q, 5, 24, 0, 1)
    assert (k == z["hnsw_keys"]).all() and (dd.view(np.uint32) == z["hnsw_dists"]).all() and (st == z["hnsw_stats"]).all()


def test_regression_pins_v2_oracle(po):
    """tests/golden/pins_v2.npz (round 2): Vamana search, filtered search, recompute provider, cross-shard merge — the oracle must
    keep reproducing the bits it produced when the HIP path was validated against it (tests/golden/make_pins_v2.py regenerates)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_pins_v2", os.path.join(GOLD, "make_pins_v2.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    now, z = m.build(), np.load(os.path.join(GOLD, "pins_v2.npz"))
    assert set(now) == set(z.files)
    for k in z.files:
        assert np.array_equal(now[k], z[k]), k


def test_traced_three_level_graph_with_ties(po):
    """tests/golden/traced_graph_400.json: 400 nodes on three levels, exact integer arithmetic, 20 duplicated vectors (ties broken by
    id), 52 HNSW cases (k > ef included) and 39 GreedySearch cases traced by the independent pure-Python implementation
    (tests/golden/make_traced_graph_v2.py) — ids, distances, expansion counts and evaluation counts, both oracle formulations."""
    from util import traced_graph
    fx, X, levels, upper_off, adj0, adjU = traced_graph("traced_graph_400.json")
    G = po.Graph.from_arrays(X, fx["M"], fx["M0"], fx["max_level"], fx["entry"], levels, upper_off, adj0, adjU)
    for c in fx["cases"]:
        q = np.array(c["query"], np.float32)
        for algo in (0, 1):
            keys, dists, st = G.search(q, c["k"], max(c["ef"], c["k"]), algo)
            assert keys.tolist() == c["ids"] and dists.tolist() == [float(x) for x in c["dists"]], (c["k"], c["ef"], algo)
            assert int(st[0]) == c["n_evals"] and int(st[1]) == len(c["expanded_base"]) and int(st[2]) == c["hops_upper"]
    V = po.Graph.from_arrays(X, fx["M0"], fx["M0"], 0, fx["entry"], np.zeros(fx["n"], np.uint8), np.zeros(fx["n"], np.uint32), adj0,
                             np.zeros((0, fx["M0"]), np.uint32))
    for c in fx["vamana_cases"]:
        q = np.array(c["query"], np.float32)
        for algo in (0, 1):
            keys, dists, st = V.search(q, c["k"], c["L"], algo)
            assert keys.tolist() == c["ids"] and dists.tolist() == [float(x) for x in c["dists"]]
            assert int(st[0]) == c["n_evals"] and int(st[1]) == len(c["expanded"])
