"""CPU suite: the C++ host mirror of the reference's index layer (leann-rs_amd/host) — BM25, tokeniser,
hybrid_rerank, metadata filter, JSON, CLI surface — against tests/golden, the oracle and the
reference's own test assertions (src/index/bm25.rs:176-329, src/index/filter.rs:445-551,
tests/integration_test.rs:13-53).  No GPU."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "leann-rs_amd", "host")
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_formulas.json")))
f32 = np.float32

FILTER_CASES = [  # (filter, metadata, expected) — filter.rs tests :451-551 plus operator coverage
    ("source:*.rs", {"source": "main.rs", "type": "code", "lines": 100}, True),
    ("type=code", {"source": "main.rs", "type": "code", "lines": 100}, True),
    ("lines>50", {"source": "main.rs", "type": "code", "lines": 100}, True),
    ("type in [code,text,doc]", {"type": "code", "lang": "rust"}, True),
    ("type in [text,doc]", {"type": "code", "lang": "rust"}, False),
    ("type not_in [text,doc]", {"type": "code"}, True),
    ("type not_in [code,text]", {"type": "code"}, False),
    ("type=code,lines>50", {"type": "code", "lines": 100}, True),
    ("type=code AND lines>50", {"type": "code", "lines": 100}, True),
    ("type=code,lines>200", {"type": "code", "lines": 100}, False),
    ("type=code OR type=text", {"type": "code"}, True),
    ("type=text OR type=doc", {"type": "code"}, False),
    ("source~main", {"source": "/path/to/main.rs"}, True),
    ("source:*main*", {"source": "/path/to/main.rs"}, True),
    ("source?", {"source": "main.rs"}, True),
    ("missing?", {"source": "main.rs"}, False),
    ("source^/path", {"source": "/path/to/main.rs"}, True),
    ("source$.py", {"source": "/path/to/main.rs"}, False),
    ("lines>=100", {"lines": 100}, True),
    ("lines<100", {"lines": 100}, False),
    ("lines<=100.5", {"lines": 100}, True),
    ("type!=code", {"type": "code"}, False),
    ("type!=code", {}, True),
    ("meta.lang=rust", {"meta": {"lang": "rust"}}, True),
    ("flag=true", {"flag": True}, True),
]


@pytest.fixture(scope="module")
def selftest(tmp_path_factory):
    exe = os.path.join(HOST, "host_selftest")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", ROOT, "leann-rs_amd/host/host_selftest"])
    cases = dict(GOLD)
    cases["filters"] = [dict(filter=f, metadata=m) for f, m, _ in FILTER_CASES]
    cases["synthetic_embed"] = "hello vector world"
    p = tmp_path_factory.mktemp("host") / "cases.json"
    p.write_text(json.dumps(cases))
    return json.loads(subprocess.check_output([exe, str(p)]))


def test_tokenize_matches_golden(selftest):
    assert selftest["tokenize"] == GOLD["tokenize"]


def test_bm25_matches_golden_and_oracle(selftest, po):
    import bm25_oracle as bo
    for name, c in GOLD["bm25"].items():
        got = [f32(x) for x in selftest["bm25"][name]["scores"]]
        for a, b in zip(got, c["scores"]):
            assert abs(int(f32(a).view(np.int32)) - int(f32(b).view(np.int32))) <= 1, name
        orc = bo.Bm25Scorer.build(c["docs"])
        assert selftest["bm25"][name]["top2"] == [i for i, _ in orc.search(c["query"], 2)]
    assert selftest["bm25"]["apple"]["top2"][0] == 3  # bm25.rs:265-280


def test_hybrid_rerank_matches_golden_and_oracle(selftest, po):
    for got, c in zip(selftest["hybrid_rerank"], GOLD["hybrid_rerank"]):
        assert [g[0] for g in got] == [o[0] for o in c["out"]]
        assert all(f32(g[1]) == f32(o[1]) for g, o in zip(got, c["out"]))
        orc = po.hybrid_rerank([tuple(v) for v in c["vr"]], c["bm"], c["alpha"])
        assert [(g[0], f32(g[1])) for g in got] == [(i, f32(s)) for i, s in orc]


def test_metadata_filter_cases(selftest):
    for (flt, meta, expect), got in zip(FILTER_CASES, selftest["filters"]):
        assert got is expect, (flt, meta)


def test_synthetic_embedding_is_unit_norm(selftest):
    v = np.array(selftest["synthetic_embed"], f32)
    assert len(v) == 16 and abs(float(np.linalg.norm(v)) - 1.0) < 1e-5


def _cli(*args, cwd=None):
    exe = os.path.join(HOST, "leann")
    return subprocess.run([exe, *args], capture_output=True, text=True, cwd=cwd)


def test_cli_surface_like_reference_integration_tests():
    r = _cli("--help")  # tests/integration_test.rs:13-24 (subset kept: the search path + build)
    assert r.returncode == 0 and "search" in r.stdout and "build" in r.stdout
    r = _cli("--version")  # :27-32
    assert r.returncode == 0 and "leann" in r.stdout
    r = _cli("search", "--help")  # :46-53 and src/cli/search.rs:10-71
    assert r.returncode == 0
    for flag in ("--top-k", "--filter", "--hybrid", "--index", "--complexity", "--show-metadata", "--auto-hybrid",
                 "--expand", "--hybrid-alpha", "--format", "--embedding-api-key", "--embedding-api-base",
                 "--embedding-host", "--query-prompt-template"):
        assert flag in r.stdout, flag


def test_cli_errors(tmp_path):
    r = _cli("search", "hello", "-i", "no-such-index", cwd=tmp_path)
    assert r.returncode == 1 and "Index 'no-such-index' not found" in r.stderr  # locate.rs:32-35
    r = _cli("search")
    assert r.returncode == 1 and "<QUERY>" in r.stderr
    r = _cli("search", "q", "--format", "xml")
    assert r.returncode == 1 and "possible values: text, json" in r.stderr
    d = tmp_path / "idx"
    d.mkdir()
    (d / "documents.leann.meta.json").write_text(json.dumps(dict(
        version="1.0", backend_name="hnsw", embedding_model="nomic-embed-text", embedding_mode="ollama",
        dimensions=768, passage_count=0)))
    r = _cli("search", "some longer query text here", "-i", str(d))
    assert r.returncode == 1  # no passages / network provider: must fail, never silently succeed
