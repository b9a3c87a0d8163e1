"""-m gpu: BASELINE configs[4] — "10M x 1536-d, DiskANN backend, hybrid BM25 rerank" — as parity tests.

(i)  Vamana R = 64 at d = 1536 (the T = 6 instantiation of the traversal kernel): graph built on the GPU, walked by the GPU and
     by the oracle (GreedySearch, oracle.c) — ids, f32 distance bits and visit counters identical at L in {10, 72, 128}.
     Reference: DiskAnnSearcher::search src/backend/diskann.rs:47-62 (beam = max(complexity, top_k) :54), alpha 1.2 :91.
(ii) IndexSearcher::search_with_options (src/index/searcher.rs:123-210) over a DiskANN index through the C++ host (`leann search`):
     same ids and f32 scores as the Python restatement (oracle/searcher_oracle.py) fed by the oracle's walk of the same graph —
     plain, filter (5x over-fetch + post-filter), hybrid (fetch_k = 5k, BM25 injection with 0.0, rerank on distances: N1), both.
     The corpus is the committed fixture tests/golden/searcher_cases.json."""
import json
import os
import subprocess

import numpy as np
import pytest

from util import SEED, recall_at_k

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "leann-rs_amd", "host", "leann")
f32 = np.float32


def _device_rows(la, n, d, stream, i0=0, clusters=512):
    buf = la.DeviceArray((n, d), np.float32)
    la._native.check(la.lib().leann_synth_rows_device(SEED, d, d, 64, clusters, 1.0, stream, i0, n, buf.ptr, None))
    la.sync()
    return buf


def test_vamana_r64_d1536_matches_oracle(la, po, gpu):
    n, d, R = 50_000, 1536, 64
    dX = _device_rows(la, n, d, 0)
    Q = _device_rows(la, 128, d, 1).to_host()
    s = la.BackendSearcher.build_device(la.BackendType.DiskAnn, dX.ptr, n, d, d, R, 128)
    g = s.graph_export()
    assert g["max_level"] == 0 and g["M0"] == R and g["n"] == n
    X = dX.to_host()
    G = po.Graph.from_arrays(X, R, R, 0, g["entry"], g["levels"], g["upper_off"], g["adj0"], g["adjU"])
    truth = po.exact_topk(X, Q, 10)
    for L in (10, 72, 128):
        ok, od, oc, ost = G.search_batch(Q, 10, L, 1, nthreads=8)
        s.stats(reset=True)
        gk, gd, gc = s.search_batch(Q, 10, L)
        st = s.stats()
        assert (gc == oc).all() and (gk == ok).all()
        assert (gd.view(np.uint32) == od.view(np.uint32)).all()
        assert st["n_dist_evals"] == int(ost[:, 0].sum()) and st["n_hops_base"] == int(ost[:, 1].sum()) and st["n_hops_upper"] == 0
        assert st["algorithmic_bytes"] == int(ost[:, 0].sum()) * d * 4 + int(ost[:, 1].sum()) * R * 4
        if L >= 72:
            assert recall_at_k(gk, truth) >= 0.95
    # the two-heap formulation (Malkov Alg. 2) and GreedySearch give the same answer on this graph too
    ok0, od0, _, _ = G.search_batch(Q, 10, 72, 0, nthreads=8)
    gk, gd, _ = s.search_batch(Q, 10, 72)
    assert (gk == ok0).all() and (gd.view(np.uint32) == od0.view(np.uint32)).all()
    # single-query trait call (16 waves per query) == batch row
    k1, d1 = s.search(Q[5], 10, 72)
    assert (k1 == gk[5]).all() and (d1 == gd[5]).all()
    s.close()


@pytest.mark.parametrize("backend,d", [("hnsw", 3072), ("diskann", 3072), ("hnsw", 4096)])
def test_wide_embeddings_built_saved_reopened(la, po, gpu, tmp_path, backend, d):
    """3 072-d is text-embedding-3-large (src/embedding/models.rs:113): the T = 12 instantiation of the traversal kernel (4 096: T = 16),
    through the whole boundary — built on the GPU, saved, reopened with load_searcher's arguments, walked by the GPU and by the oracle."""
    n = 6000
    bt = la.BackendType.Hnsw if backend == "hnsw" else la.BackendType.DiskAnn
    dX = _device_rows(la, n, d, 0)
    Q = _device_rows(la, 64, d, 1).to_host()
    X = dX.to_host()
    stem = str(tmp_path / "documents.leann")
    la.BackendBuilder(bt).build(X, [], stem, d, 16, 64)
    s = la.BackendSearcher.load(bt, stem, d)
    g = s.graph_export()
    M0 = g["M0"]
    G = po.Graph.from_arrays(X, g["M"], M0, g["max_level"], g["entry"], g["levels"], g["upper_off"], g["adj0"], g["adjU"])
    algo = 0 if backend == "hnsw" else 1
    for k, ef in ((10, 64), (1, 1), (5, 20)):
        ok, od, oc, ost = G.search_batch(Q, k, ef, algo, nthreads=8)
        s.stats(reset=True)
        gk, gd, gc = s.search_batch(Q, k, ef)
        assert (gc == oc).all() and (gk == ok).all() and (gd.view(np.uint32) == od.view(np.uint32)).all()
        assert s.stats()["n_dist_evals"] == int(ost[:, 0].sum())
    gk, _, _ = s.search_batch(Q, 10, 64)
    assert recall_at_k(gk, po.exact_topk(X, Q, 10)) >= 0.95
    k1, d1 = s.search(Q[7], 10, 64)  # single-query trait call: 16 waves per query
    assert (k1 == gk[7]).all()
    s.close()


# ---- (ii) DiskANN + hybrid through IndexSearcher ------------------------------------------------------------------------
FX = json.load(open(os.path.join(ROOT, "tests", "golden", "searcher_cases.json")))
DIMS = 1536
QUERIES = [  # (query text, top_k, complexity, extra CLI flags, oracle kwargs)
    ("diskann vamana robust prune alpha for graphs", 5, 64, ["--auto-hybrid", "false"], {}),
    ("gpu kernel wavefront lds bandwidth numbers", 4, 32, ["--auto-hybrid", "false", "-f", "lang=rust"], dict(filter_text="lang=rust")),
    ("hybrid rerank normalise blend of scores", 5, 64, ["--hybrid"], dict(hybrid=True, hybrid_alpha=0.7)),
    ("bm25 ranking", 6, 64, [], dict(hybrid=True, hybrid_alpha=0.7)),  # <= 3 words: auto hybrid (search.rs:147-148)
    ("vector database embedding search engines", 5, 48, ["--hybrid", "--hybrid-alpha", "0.25"], dict(hybrid=True, hybrid_alpha=0.25)),
    ("python asyncio event loop coroutine tasks", 3, 64, ["--hybrid", "-f", "source:*.py"], dict(hybrid=True, hybrid_alpha=0.7, filter_text="source:*.py")),
    ("rust ownership borrow checker lifetimes explained", 8, 100, ["--auto-hybrid", "false", "-f", "lines<40"], dict(filter_text="lines<40")),
    # corrected polarity (`--compat-polarity false`): the ANN hits enter hybrid_rerank as 1 - dist
    ("hybrid rerank normalise blend of scores", 5, 64, ["--hybrid", "--compat-polarity", "false"], dict(hybrid=True, hybrid_alpha=0.7, compat_polarity=False)),
    ("diskann vamana", 4, 64, ["--compat-polarity", "false", "--hybrid-alpha", "0.4"], dict(hybrid=True, hybrid_alpha=0.4, compat_polarity=False)),
]


def _run(*args):
    return subprocess.run([EXE, *args], capture_output=True, text=True)


@pytest.fixture(scope="module")
def diskann_dir(tmp_path_factory, gpu):
    d = tmp_path_factory.mktemp("config5")
    (d / "docs.jsonl").write_text("\n".join(json.dumps(x) for x in FX["corpus"]))
    r = _run("build", "--index-dir", str(d / "idx"), "--passages-jsonl", str(d / "docs.jsonl"), "--dimensions", str(DIMS),
             "--backend-name", "diskann", "--graph-degree", "16", "--complexity", "64", "--recompute")  # (documents.embeddings is read below)
    assert r.returncode == 0, r.stderr
    return d / "idx"


def test_diskann_hybrid_through_index_searcher_matches_oracle(la, po, diskann_dir):
    import searcher_oracle as so
    meta = json.loads((diskann_dir / "documents.leann.meta.json").read_text())
    assert meta["backend_name"] == "diskann" and meta["dimensions"] == DIMS
    assert (diskann_dir / "documents.diskann").exists() and not (diskann_dir / "documents.index").exists()  # diskann.rs:22
    stem = str(diskann_dir / "documents.leann")
    s = la.DiskAnnSearcher.load(stem, DIMS)
    g = s.graph_export(with_vectors=True)
    emb = np.fromfile(diskann_dir / "documents.embeddings", np.float32).reshape(-1, DIMS)
    assert (g["vectors"] == emb).all()  # the index holds exactly the embeddings the builder wrote (embeddings.rs layout)
    G = po.Graph.from_arrays(g["vectors"], g["M"], g["M0"], 0, g["entry"], g["levels"], g["upper_off"], g["adj0"], g["adjU"])
    docs = FX["corpus"]
    id_map = [x["id"] for x in docs]
    passages = {x["id"]: x for x in docs}

    def backend_search(q, fetch_k, complexity):  # DiskAnnSearcher::search on the CPU: the oracle's GreedySearch, beam = max(complexity, k)
        keys, dists, _ = G.search(q, fetch_k, max(complexity, fetch_k), 1)
        return keys, dists

    rng = np.random.default_rng(5)
    for qi, (text, top_k, cx, flags, kw) in enumerate(QUERIES):
        # the query embedding is handed to both sides as data (every real provider is a network service): a stored row + noise
        q = emb[(qi * 11 + 3) % len(emb)] + 0.05 * rng.standard_normal(DIMS).astype(f32)
        q = (q / np.linalg.norm(q)).astype(f32)
        qf = diskann_dir / f"q{qi}.f32"
        q.tofile(qf)
        r = _run("search", text, "-i", str(diskann_dir), "--top-k", str(top_k), "--complexity", str(cx), "--format", "json",
                 "--query-vector-file", str(qf), *flags)
        assert r.returncode == 0, r.stderr
        got = [(x["id"], f32(x["score"])) for x in json.loads(r.stdout)]
        want = so.search_with_options(backend_search, id_map, passages, q, top_k, cx, query_text=text, **kw)
        assert got == [(i, f32(sc)) for i, sc in want], (text, got, want)
        assert len(got) > 0
        # the GPU backend call alone == the oracle's backend call (what the equality above rests on)
        fk = top_k * 5 if kw else top_k
        gk, gd = s.search(q, fk, cx)
        ok, od = backend_search(q, fk, cx)
        assert (gk == ok).all() and (gd.view(np.uint32) == od.view(np.uint32)).all()
    s.close()


def test_hybrid_scores_are_blends_of_distances(diskann_dir):
    """N1 kept on purpose: in hybrid mode the blended score grows with the DISTANCE (bm25.rs:159 on searcher.rs:139-143's pairs)"""
    q = "diskann vamana"
    r = _run("search", q, "-i", str(diskann_dir), "--top-k", "5", "--format", "json", "--embedding-mode", "synthetic")
    assert r.returncode == 0, r.stderr
    sc = [x["score"] for x in json.loads(r.stdout)]
    assert sc == sorted(sc, reverse=True) and 0.0 <= min(sc) and max(sc) <= 1.0 + 1e-6
