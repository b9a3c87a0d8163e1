"""Generates tests/golden/searcher_cases.json: a small corpus + recorded backend outputs + the results
IndexSearcher::search_with_options (src/index/searcher.rs:123-210) must produce from them, computed by the Python
restatement oracle/searcher_oracle.py (f32 step by step, libm logf).  Run from the repo root:
    python tests/golden/make_searcher_cases.py
The expectations pin: fetch_k = 5 * top_k, BM25-only hits injected with vector score 0.0, hybrid_rerank on DISTANCES (polarity
quirk N1), post-filter, ids beyond the id map mapped to their decimal string (and skipped when no such passage exists)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import searcher_oracle as so  # noqa: E402

TOPICS = ["rust ownership borrow checker lifetimes", "python asyncio event loop coroutine", "vector database embedding search",
          "graph traversal beam hnsw neighbours", "gpu kernel wavefront lds bandwidth", "bm25 ranking term frequency",
          "diskann vamana robust prune alpha", "hybrid rerank normalise blend"]


def corpus(n=64):
    docs = []
    for i in range(n):
        t = TOPICS[i % len(TOPICS)]
        extra = " ".join(TOPICS[(i * 5 + 3) % len(TOPICS)].split()[: i % 3])
        docs.append(dict(id=str(i + 1), text=f"passage {i} about {t} {extra} number {i * 7919 % 1000}".replace("  ", " "),
                         metadata=dict(source=f"file{i % 10}.{'rs' if i % 2 else 'py'}", lines=i, lang="rust" if i % 2 else "python")))
    return docs


def backend(rng, n, count, extra=()):
    """a plausible backend answer: distinct positions, ascending f32 distances in [0.05, 1.2]"""
    keys = [int(x) for x in rng.choice(n, size=count, replace=False)] + list(extra)
    d = np.sort(rng.uniform(0.05, 1.2, size=len(keys))).astype(np.float32)
    return [[k, float(x)] for k, x in zip(keys, d)]


def main():
    rng = np.random.default_rng(0x5EED0005)
    docs = corpus()
    n = len(docs)
    id_map = [d["id"] for d in docs]
    passages = {d["id"]: d for d in docs}
    cases = [
        dict(name="plain", top_k=5, backend=backend(rng, n, 5)),
        dict(name="plain_short_result", top_k=8, backend=backend(rng, n, 3)),
        dict(name="plain_key_beyond_id_map", top_k=4, backend=backend(rng, n, 3, extra=[n + 7])),
        dict(name="filter_post", top_k=4, filter="lang=rust", backend=backend(rng, n, 20)),
        dict(name="filter_starved", top_k=5, filter="lines<6", backend=backend(rng, n, 25)),
        dict(name="hybrid_default_alpha", top_k=5, hybrid=True, alpha=0.7, query_text="diskann vamana prune", backend=backend(rng, n, 25)),
        dict(name="hybrid_alpha_03", top_k=6, hybrid=True, alpha=0.3, query_text="gpu kernel bandwidth", backend=backend(rng, n, 30)),
        dict(name="hybrid_no_bm25_match", top_k=3, hybrid=True, alpha=0.7, query_text="zzzz qqqq", backend=backend(rng, n, 15)),
        dict(name="hybrid_empty_backend", top_k=3, hybrid=True, alpha=0.7, query_text="hybrid rerank blend", backend=[]),
        dict(name="hybrid_and_filter", top_k=4, hybrid=True, alpha=0.7, filter="source:*.py", query_text="asyncio event loop",
             backend=backend(rng, n, 20)),
        dict(name="hybrid_equal_distances", top_k=4, hybrid=True, alpha=0.7, query_text="bm25 ranking",
             backend=[[k, 0.5] for k in (9, 3, 40, 17, 22)]),
        # corrected polarity (`--compat-polarity false`): ANN hits enter the blend as 1 - dist
        dict(name="hybrid_corrected_polarity", top_k=5, hybrid=True, alpha=0.7, compat_polarity=False, query_text="diskann vamana prune",
             backend=backend(rng, n, 25)),
        dict(name="hybrid_corrected_no_bm25_match", top_k=3, hybrid=True, alpha=0.7, compat_polarity=False, query_text="zzzz qqqq",
             backend=backend(rng, n, 15)),
        dict(name="plain_corrected_polarity_is_a_noop", top_k=5, compat_polarity=False, backend=backend(rng, n, 5)),
    ]
    for c in cases:
        rec = c["backend"]

        def bsearch(q, fetch_k, complexity, rec=rec):
            return [k for k, _ in rec], [np.float32(d) for _, d in rec]  # the recorded list (<= fetch_k entries by construction)
        assert len(rec) <= c["top_k"] * (5 if (c.get("filter") or c.get("hybrid")) else 1)
        res = so.search_with_options(bsearch, id_map, passages, None, c["top_k"], 64, filter_text=c.get("filter"),
                                     hybrid=c.get("hybrid", False), hybrid_alpha=c.get("alpha", 0.7), query_text=c.get("query_text"),
                                     compat_polarity=c.get("compat_polarity", True))
        c["expect"] = [[i, float(s)] for i, s in res]
    out = dict(corpus=docs, cases=cases)
    with open(os.path.join(ROOT, "tests", "golden", "searcher_cases.json"), "w") as f:
        json.dump(out, f, indent=1)
    for c in cases:
        print(c["name"], c["expect"])


if __name__ == "__main__":
    main()
