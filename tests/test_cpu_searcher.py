"""CPU suite: IndexSearcher::search_with_options (src/index/searcher.rs:123-210) after the backend call — the C++ host's
assemble_results and the Python restatement (oracle/searcher_oracle.py) against tests/golden/searcher_cases.json: recorded backend
outputs -> expected (id, f32 score) lists.  Bit-exact scores.  The index directory is written here in the reference's layout
(passages.rs:48-49,120-158; searcher.rs:83), so the C++ readers are exercised on files they did not write.  No GPU."""
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "leann-rs_amd", "host")
FX = json.load(open(os.path.join(ROOT, "tests", "golden", "searcher_cases.json")))
f32 = np.float32


def write_reference_layout(d, docs, with_ids=True):
    """documents.passages.jsonl + .passages.idx.json (+ documents.ids.txt) exactly as the reference's PassageStoreWriter does"""
    os.makedirs(d, exist_ok=True)
    offsets, pos = {}, 0
    with open(os.path.join(d, "documents.passages.jsonl"), "wb") as f:
        for p in docs:
            line = (json.dumps(dict(id=p["id"], text=p["text"], metadata=p["metadata"])) + "\n").encode()
            offsets[p["id"]] = pos
            f.write(line)
            pos += len(line)
    json.dump(offsets, open(os.path.join(d, "documents.passages.idx.json"), "w"))
    if with_ids:
        open(os.path.join(d, "documents.ids.txt"), "w").write("".join(p["id"] + "\n" for p in docs))
    return os.path.join(d, "documents.leann")


def test_python_restatement_reproduces_the_fixture(po):
    import searcher_oracle as so
    docs = FX["corpus"]
    id_map = [d["id"] for d in docs]
    passages = {d["id"]: d for d in docs}
    for c in FX["cases"]:
        rec = c["backend"]
        res = so.search_with_options(lambda q, fk, cx: ([k for k, _ in rec], [f32(d) for _, d in rec]), id_map, passages, None,
                                     c["top_k"], 64, filter_text=c.get("filter"), hybrid=c.get("hybrid", False),
                                     hybrid_alpha=c.get("alpha", 0.7), query_text=c.get("query_text"),
                                     compat_polarity=c.get("compat_polarity", True))
        assert [(i, f32(s)) for i, s in res] == [(i, f32(s)) for i, s in c["expect"]], c["name"]


def test_hybrid_rerank_restatements_agree(po):
    """searcher_oracle.hybrid_rerank (numpy f32) == oracle.c:orc_hybrid_rerank on the fixture's hybrid cases"""
    import bm25_oracle as bo
    import searcher_oracle as so
    docs = FX["corpus"]
    scorer = bo.Bm25Scorer.build([d["text"] for d in docs])
    for c in FX["cases"]:
        if not c.get("hybrid") or not c["backend"]:
            continue
        vr = [(k, f32(d)) for k, d in c["backend"] if k < len(docs)]
        bm = scorer.score_query(c["query_text"])
        a = so.hybrid_rerank(vr, bm, c["alpha"])
        b = po.hybrid_rerank([(k, float(d)) for k, d in vr], bm, c["alpha"])
        assert [(i, f32(s)) for i, s in a] == [(i, f32(s)) for i, s in b], c["name"]


def test_sparse_form_of_hybrid_rerank_equals_the_dense_one(po):
    """hybrid_rerank_sparse / hybrid_leg_sparse (used where N is 10M: bench.py --hybrid, tests/test_gpu_hybrid.py) == the literal
    dense restatement, f32 bit for bit, incl. no positives, all passages positive, ties and ANN hits that are BM25 positives"""
    import searcher_oracle as so
    rng = np.random.default_rng(77)
    for trial in range(200):
        n_docs = int(rng.integers(5, 400))
        nv = int(rng.integers(0, min(n_docs, 60) + 1))
        keys = rng.choice(n_docs, size=nv, replace=False)
        dists = np.sort(rng.uniform(0.0, 1.4, size=nv)).astype(np.float32)
        npos = n_docs if trial % 9 == 0 else int(rng.integers(0, min(n_docs, 70) + 1))
        pidx = rng.choice(n_docs, size=npos, replace=False)
        psc = (rng.integers(1, 30, size=npos) * 0.25).astype(np.float32)
        dense = np.zeros(n_docs, np.float32)
        dense[pidx] = psc
        alpha = float(rng.choice([0.0, 0.3, 0.7, 1.0]))
        vr = [(int(k), f32(d)) for k, d in zip(keys, dists)]
        a = so.hybrid_rerank(vr, dense, alpha)
        b = so.hybrid_rerank_sparse(vr, {int(i): f32(s) for i, s in zip(pidx, psc)}, n_docs, alpha)
        assert [(i, f32(s).tobytes()) for i, s in a] == [(i, f32(s).tobytes()) for i, s in b], trial
        # the whole leg against bm25_oracle's own ordering rule (score desc, stable by index)
        order = sorted(range(npos), key=lambda t: (-float(psc[t]), int(pidx[t])))
        pos_sorted = [(int(pidx[t]), f32(psc[t])) for t in order]
        fetch_k = 50
        for compat in (True, False):
            got = so.hybrid_leg_sparse(keys, dists, pos_sorted, n_docs, alpha, 10, fetch_k, compat)
            vr2 = [(int(k), f32(d) if compat else f32(f32(1.0) - f32(d))) for k, d in zip(keys, dists)]
            have = {i for i, _ in vr2}
            top = [i for i in sorted(range(n_docs), key=lambda i: (-float(dense[i]), i)) if dense[i] > 0][:fetch_k]  # Bm25Scorer::search
            vr2 += [(i, f32(0.0)) for i in top if i not in have]
            exp = so.hybrid_rerank(vr2, dense, alpha)[:10]
            assert [(i, f32(s).tobytes()) for i, s in got] == [(i, f32(s).tobytes()) for i, s in exp], (trial, compat)


def test_semantics_the_fixture_pins():
    by = {c["name"]: c for c in FX["cases"]}
    # keys beyond the id map become their decimal string (searcher.rs:180-184); no such passage -> skipped (:203-205)
    assert len(by["plain_key_beyond_id_map"]["expect"]) == 3
    # post-filter on a 5x over-fetch can starve (searcher.rs:129-133,:190-194)
    assert len(by["filter_starved"]["expect"]) < by["filter_starved"]["top_k"]
    # BM25-only hits enter with vector score 0.0 (:160-165): with an empty backend answer every blended score is (1-alpha)*norm_bm25
    assert all(abs(s - 0.3) < 1e-6 or s < 0.3 for _, s in by["hybrid_empty_backend"]["expect"])
    # polarity quirk N1: the WORST distance of the backend list gets norm_vec = 1 -> with no BM25 match the order is reversed
    c = by["hybrid_no_bm25_match"]
    worst_first = [str(k + 1) for k, _ in sorted(c["backend"], key=lambda t: -t[1])][:3]
    assert [i for i, _ in c["expect"]] == worst_first
    # ... and the corrected mode (`--compat-polarity false`, 1 - dist into the blend) puts the BEST distance first on the same kind of input
    c = by["hybrid_corrected_no_bm25_match"]
    best_first = [str(k + 1) for k, _ in sorted(c["backend"], key=lambda t: t[1])][:3]
    assert [i for i, _ in c["expect"]] == best_first
    assert by["plain_corrected_polarity_is_a_noop"]["expect"] == [[str(k + 1), d] for k, d in by["plain_corrected_polarity_is_a_noop"]["backend"]]


def test_cpp_assemble_results_matches_the_fixture(tmp_path):
    exe = os.path.join(HOST, "host_selftest")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", ROOT, "leann-rs_amd/host/host_selftest"])
    stem = write_reference_layout(str(tmp_path / "idx"), FX["corpus"])
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_formulas.json")))
    cases = dict(tokenize=gold["tokenize"], bm25=gold["bm25"], hybrid_rerank=gold["hybrid_rerank"],
                 searcher=dict(index_path=stem, cases=FX["cases"]))
    p = tmp_path / "cases.json"
    p.write_text(json.dumps(cases))
    out = json.loads(subprocess.check_output([exe, str(p)]))
    for c, got in zip(FX["cases"], out["searcher"]):
        assert [(i, f32(s)) for i, s in got] == [(i, f32(s)) for i, s in c["expect"]], c["name"]


def test_host_cpp_is_clean_under_asan_and_ubsan(tmp_path):
    """the same fixtures through the AddressSanitizer + UBSan build of the host C++ (JSON parser, passage store, tokenizer, BM25,
    hybrid_rerank, metadata filters, assemble_results): identical output, no sanitizer report.  (CPU only; VERDICT r1 aux note.)"""
    exe = os.path.join(HOST, "host_selftest_asan")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", ROOT, "leann-rs_amd/host/host_selftest_asan"])
    stem = write_reference_layout(str(tmp_path / "idx"), FX["corpus"])
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_formulas.json")))
    filters = [dict(filter=f, metadata=m) for f, m in (("source:*.rs", {"source": "a.rs"}), ("lines>=10,lang=rust", {"lines": 12, "lang": "rust"}),
                                                       ("t in [a,b,c]", {"t": "b"}), ("x.y~zz", {"x": {"y": "azzb"}}), ("bad filter", {}), ("q?", {}))]
    cases = dict(tokenize=gold["tokenize"], bm25=gold["bm25"], hybrid_rerank=gold["hybrid_rerank"], filters=filters,
                 synthetic_embed="hello sanitizer world", searcher=dict(index_path=stem, cases=FX["cases"]))
    p = tmp_path / "cases.json"
    p.write_text(json.dumps(cases))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([exe, str(p)], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    out = json.loads(r.stdout)
    for c, got in zip(FX["cases"], out["searcher"]):
        assert [(i, f32(s)) for i, s in got] == [(i, f32(s)) for i, s in c["expect"]], c["name"]
    assert out["filters"][:4] == [True, True, True, True]
