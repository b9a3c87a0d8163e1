"""-m gpu: an index DIRECTORY as stock leann-rs leaves it is searchable, and a pruned
directory with a recompute-on graph is walked, not scanned (item 4).

The fixture directory is written here, byte layout as the reference writes it:
    documents.leann.meta.json   serde field names of IndexMeta                       src/index/meta.rs:9-43
    documents.ids.txt           one id per line                                        src/index/searcher.rs:83-87
    documents.passages.jsonl / documents.passages.idx.json                             src/index/passages.rs:48-49,120-158
    documents.embeddings        raw LE f32 [n x dims]                                  src/index/embeddings.rs:21-153
    documents.index             usearch's file — NOT readable here; a dummy with a foreign magic stands in for it
                                (no usearch build exists offline; synthetic bytes, said so on purpose)
leann_backend_open then rebuilds the graph on the GPU from documents.embeddings and caches it as documents.gpu.index."""
import json
import os
import subprocess
import time

import numpy as np
import pytest

from test_cpu_searcher import write_reference_layout
from util import recall_at_k, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "leann-rs_amd", "host", "leann")


def _stock_dir(po, d, n, dims, backend="hnsw"):
    docs = [dict(id=f"doc-{i}", text=f"chunk {i} of the corpus", metadata=dict(source=f"f{i % 7}.md", lines=i)) for i in range(n)]
    stem = write_reference_layout(str(d), docs)
    X = synth(po, n, dims)
    X.tofile(os.path.join(d, "documents.embeddings"))
    meta = dict(version="1.0", backend_name=backend, embedding_model="nomic-embed-text", embedding_mode="ollama", dimensions=dims,
                passage_count=n, is_recompute=True, is_pruned=False)  # meta.rs: optional kwargs/options omitted when None
    json.dump(meta, open(os.path.join(d, "documents.leann.meta.json"), "w"), indent=2)
    ann = "documents.diskann" if backend == "diskann" else "documents.index"
    open(os.path.join(d, ann), "wb").write(b"usearch\x02\x17\x00" + np.random.default_rng(1).bytes(4096))
    return stem, X, docs


@pytest.mark.parametrize("backend", ["hnsw", "diskann"])
def test_stock_directory_opens_by_rebuilding_from_embeddings(la, po, gpu, tmp_path, backend):
    n, dims = 20000, 96
    stem, X, docs = _stock_dir(po, tmp_path / "idx", n, dims, backend)
    kind = la.BackendType.from_name(backend)
    foreign = (tmp_path / "idx" / ("documents.diskann" if backend == "diskann" else "documents.index")).read_bytes()
    s = la.BackendSearcher.load(kind, stem, dims)
    assert s.len() == n and s.dims() == dims
    Q = synth(po, 100, dims, stream=1)
    gk, gd, gc = s.search_batch(Q, 10, 64)
    assert recall_at_k(gk, po.exact_topk(X, Q, 10)) >= 0.95
    g = s.graph_export(with_vectors=True)
    assert (g["vectors"] == X).all()
    # GPU == oracle on the rebuilt graph
    G = po.Graph.from_arrays(X, g["M"], g["M0"], g["max_level"], g["entry"], g["levels"], g["upper_off"], g["adj0"], g["adjU"])
    ok, od, oc, _ = G.search_batch(Q, 10, 64, 1 if backend == "diskann" else 0, nthreads=8)
    assert (gk == ok).all() and (gd.view(np.uint32) == od.view(np.uint32)).all()
    s.close()
    side = tmp_path / "idx" / ("documents.gpu.diskann" if backend == "diskann" else "documents.gpu.index")
    assert side.exists() and side.read_bytes()[:8] == b"LEANNGX1"
    assert (tmp_path / "idx" / ("documents.diskann" if backend == "diskann" else "documents.index")).read_bytes() == foreign  # untouched
    # second open: served from the sidecar (same graph, no rebuild)
    mtime = side.stat().st_mtime_ns
    s2 = la.BackendSearcher.load(kind, stem, dims)
    k2, d2, _ = s2.search_batch(Q, 10, 64)
    assert (k2 == gk).all() and (d2 == gd).all() and side.stat().st_mtime_ns == mtime
    s2.close()
    # embeddings replaced (a rebuild by stock leann): the stale sidecar is ignored and rewritten
    time.sleep(1.1)
    X2 = synth(po, n // 2, dims, i0=777)
    X2.tofile(tmp_path / "idx" / "documents.embeddings")
    s3 = la.BackendSearcher.load(kind, stem, dims)
    assert s3.len() == n // 2
    s3.close()


def test_stock_directory_through_the_cli(la, po, gpu, tmp_path):
    n, dims = 3000, 64
    stem, X, docs = _stock_dir(po, tmp_path / "idx", n, dims)
    q = X[123] + 0.01
    (q / np.linalg.norm(q)).astype(np.float32).tofile(tmp_path / "q.f32")
    env = dict(os.environ, LEANN_LOG="info")
    r = subprocess.run([EXE, "search", "some longer query text here", "-i", str(tmp_path / "idx"), "--top-k", "3", "--format", "json",
                        "--query-vector-file", str(tmp_path / "q.f32")], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    res = json.loads(r.stdout)
    assert res[0]["id"] == "doc-123" and res[0]["metadata"]["lines"] == 123 and res[0]["text"] == "chunk 123 of the corpus"
    assert "rebuilding the graph on the GPU" in r.stderr
    r2 = subprocess.run([EXE, "search", "some longer query text here", "-i", str(tmp_path / "idx"), "--top-k", "3", "--format", "json",
                         "--query-vector-file", str(tmp_path / "q.f32")], capture_output=True, text=True, env=env)
    assert json.loads(r2.stdout) == res and "using the cached GPU graph" in r2.stderr and "rebuilding" not in r2.stderr


TOPICS = ["rust ownership borrow checker lifetimes", "python asyncio event loop coroutine", "vector database embedding search",
          "graph traversal beam hnsw neighbours", "gpu kernel wavefront lds bandwidth", "bm25 ranking term frequency"]


def test_pruned_directory_with_recompute_graph_is_walked(la, gpu, tmp_path):
    """`leann build --recompute-graph`: graph + 520-B feature rows, no vectors, no .embeddings; meta says is_pruned.  `leann search`
    walks the graph (distances recomputed on the device) and agrees with the stored-vector twin of the same provider."""
    docs = [dict(id=str(i + 1), text=f"passage {i} about {TOPICS[i % 6]} number {i * 7919 % 1000}", metadata=dict(lines=i)) for i in range(1500)]
    (tmp_path / "docs.jsonl").write_text("\n".join(json.dumps(x) for x in docs))

    def run(*a):
        return subprocess.run([EXE, *a], capture_output=True, text=True)
    r = run("build", "--index-dir", str(tmp_path / "rg"), "--passages-jsonl", str(tmp_path / "docs.jsonl"), "--dimensions", "384",
            "--graph-degree", "16", "--recompute-graph")
    assert r.returncode == 0, r.stderr
    r = run("build", "--index-dir", str(tmp_path / "full"), "--passages-jsonl", str(tmp_path / "docs.jsonl"), "--dimensions", "384",
            "--graph-degree", "16", "--embedding-mode", "synthetic-linear")
    assert r.returncode == 0, r.stderr
    files = set(os.listdir(tmp_path / "rg"))
    assert "documents.index" in files and "documents.embeddings" not in files
    meta = json.loads((tmp_path / "rg" / "documents.leann.meta.json").read_text())
    assert meta["is_pruned"] and meta["is_recompute"] and meta["embedding_mode"] == "synthetic-linear"
    sz_rg, sz_full = os.path.getsize(tmp_path / "rg" / "documents.index"), os.path.getsize(tmp_path / "full" / "documents.index")
    assert sz_rg < 0.6 * sz_full  # 520 B instead of 1 536 B per passage (+ the graph and, once, 256 x 384 f32 weights = 393 KB)
    for q in ("gpu kernel wavefront lds bandwidth for the win", "what about rust ownership and the borrow checker"):
        a = json.loads(run("search", q, "-i", str(tmp_path / "rg"), "--top-k", "6", "--format", "json", "--complexity", "128").stdout)
        b = json.loads(run("search", q, "-i", str(tmp_path / "full"), "--top-k", "6", "--format", "json", "--complexity", "128").stdout)
        assert len(a) == 6 and [x["score"] for x in a] == sorted(x["score"] for x in a)  # distances, ascending: a graph walk, not the raw-dot scan
        assert np.allclose([x["score"] for x in a], [x["score"] for x in b], atol=1e-5)
        assert len({x["id"] for x in a} & {x["id"] for x in b}) >= 5
    # hybrid and filters work on the pruned graph too (the reference's pruned path supports neither hybrid nor ANN)
    h = json.loads(run("search", "bm25 ranking frequency", "-i", str(tmp_path / "rg"), "--top-k", "5", "--format", "json").stdout)
    assert len(h) == 5 and all("bm25" in x["text"] for x in h[:3])
    f = json.loads(run("search", "vector database embedding search and more", "-i", str(tmp_path / "rg"), "--top-k", "4", "--format", "json",
                       "-f", "lines<700").stdout)
    assert len(f) == 4 and all(x["metadata"]["lines"] < 700 for x in f)
