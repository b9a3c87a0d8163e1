"""Generates the committed golden fixtures (tests/golden/*.json / *.npz).

The reference (Rust) cannot be built or imported here, so the vectors are computed
INDEPENDENTLY of oracle/oracle.c — in numpy float32 / float64 and plain Python — from the formulas of
the reference source that IS in tree:
    dot_product       src/index/recompute.rs:137-139  on the vectors of benches/benchmarks.rs:28-31,43-45
    stable sort+take  src/index/recompute.rs:106-109
    hybrid_rerank     src/index/bm25.rs:135-170       on the cases of its tests (:283-329)
    BM25              src/index/bm25.rs:33-122        on the corpora of its tests (:200-280)
    exact IP top-10   ground truth for the 1k x 128 / 10k x 128 plumbing sets
Run:  python tests/golden/make_golden.py     (needs only numpy; writes next to itself)
"""
import json
import math
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
f32 = np.float32


def seq_dot(a, b):  # recompute.rs:137-139: products rounded to f32, summed left to right
    s = f32(0.0)
    for x, y in zip(a, b):
        s = f32(s + f32(f32(x) * f32(y)))
    return s


def tokenize(text):  # bm25.rs:127-132
    return [t.lower() for t in re.findall(r"[a-zA-Z0-9]+", text) if len(t) > 1]


def bm25_scores(docs, query, k1=f32(1.2), b=f32(0.75)):  # bm25.rs:33-106, f32 op for op
    toks = [tokenize(d) for d in docs]
    n = len(docs)
    lens = [len(t) for t in toks]
    avg = f32(sum(lens)) / f32(n) if n else f32(1.0)
    df = {}
    for t in toks:
        for w in set(t):
            df[w] = df.get(w, 0) + 1
    scores = [f32(0.0)] * n
    for qt in tokenize(query):
        d_f = f32(df.get(qt, 0))
        if d_f == 0:
            continue
        idf = f32(math.log(f32(f32(f32(f32(n) - d_f) + f32(0.5)) / f32(d_f + f32(0.5))) + f32(1.0)))
        # NOTE: Rust f32::ln is correctly rounded from the f32 argument in practice; math.log(double)
        # then rounding to f32 agrees except in rare half-ulp cases — the test allows 1 ulp on BM25.
        for i, t in enumerate(toks):
            tf = f32(t.count(qt))
            if tf == 0:
                continue
            norm = f32(f32(f32(1.0) - b) + f32(b * f32(f32(lens[i]) / avg)))
            sc = f32(f32(idf * f32(tf * f32(k1 + f32(1.0)))) / f32(tf + f32(k1 * norm)))
            scores[i] = f32(scores[i] + sc)
    return [float(s) for s in scores]


def hybrid_rerank(vr, bm, alpha):  # bm25.rs:135-170
    alpha = f32(alpha)
    vs = [f32(s) for _, s in vr]
    mx, mn = max(vs), min(vs)
    vrange = max(f32(mx - mn), f32(1e-6))
    bmf = [f32(x) for x in bm]
    bmx, bmn = max(bmf), min(bmf)
    brange = max(f32(bmx - bmn), f32(1e-6))
    out = []
    for (idx, s) in vr:
        nv = f32(f32(f32(s) - mn) / vrange)
        bb = bmf[idx] if idx < len(bmf) else f32(0.0)
        nb = f32(f32(bb - bmn) / brange)
        out.append((idx, f32(f32(alpha * nv) + f32(f32(f32(1.0) - alpha) * nb))))
    out = sorted(out, key=lambda t: -t[1])  # Python's sort is stable, like Rust's sort_by
    return [(int(i), float(s)) for i, s in out]


def main():
    g = {}
    # (5) dot_product on benches/benchmarks.rs vectors a_i = b_i = i/1000
    for dims in (1536, 768):
        a = [f32(f32(i) / f32(1000.0)) for i in range(dims)]
        g[f"dot_seq_{dims}"] = float(seq_dot(a, a))
    # (2) hybrid_rerank cases of bm25.rs:283-329 (+ a tie / out-of-range case)
    cases = [
        dict(vr=[(0, 0.9), (1, 0.8), (2, 0.7)], bm=[0.5, 0.9, 0.3], alpha=0.5),
        dict(vr=[(0, 0.9), (1, 0.5)], bm=[0.1, 0.9], alpha=1.0),
        dict(vr=[(0, 0.9), (1, 0.5)], bm=[0.1, 0.9], alpha=0.0),
        dict(vr=[(3, 0.25), (1, 0.25), (7, 0.0), (0, 0.75)], bm=[0.0, 2.5, 0.0, 2.5], alpha=0.7),
        dict(vr=[(2, 0.4)], bm=[1.0, 2.0, 3.0], alpha=0.7),
    ]
    g["hybrid_rerank"] = [dict(c, out=hybrid_rerank(c["vr"], c["bm"], c["alpha"])) for c in cases]
    # (3) BM25 on the corpora of bm25.rs:200-280
    corp = {
        "fox": (["the quick brown fox jumps over the lazy dog", "a quick brown dog outpaces a swift fox",
                 "the dog chases the fox around the yard"], "quick fox"),
        "rust": (["rust rust rust programming", "rust programming"], "rust"),
        "rare": (["common rare", "common", "common"], "rare"),
        "apple": (["apple banana", "apple cherry", "banana cherry", "apple apple apple"], "apple"),
        "empty": (["hello world"], ""),
        "nomatch": (["hello world"], "xyz"),
    }
    g["bm25"] = {k: dict(docs=d, query=q, scores=bm25_scores(d, q)) for k, (d, q) in corp.items()}
    g["tokenize"] = {"Hello, World! This is a test.": tokenize("Hello, World! This is a test."),
                     "test123 456abc": tokenize("test123 456abc"), "": tokenize("")}
    json.dump(g, open(os.path.join(HERE, "reference_formulas.json"), "w"), indent=1)

    # (1) exact IP top-10 (float64) of seeded i.i.d. sets; vectors regenerated from the numpy seed in the test
    for n in (1000, 10000):
        rng = np.random.default_rng(1234 + n)
        X = rng.standard_normal((n, 128)).astype(np.float32)
        X /= np.linalg.norm(X, axis=1, keepdims=True)
        Q = rng.standard_normal((16, 128)).astype(np.float32)
        Q /= np.linalg.norm(Q, axis=1, keepdims=True)
        S = Q.astype(np.float64) @ X.astype(np.float64).T
        idx = np.argsort(-S, axis=1, kind="stable")[:, :10]
        np.savez_compressed(os.path.join(HERE, f"exact_top10_{n}x128.npz"), idx=idx.astype(np.int32),
                            score=np.take_along_axis(S, idx, 1).astype(np.float32))


if __name__ == "__main__":
    main()
