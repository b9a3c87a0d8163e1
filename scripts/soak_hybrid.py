"""Randomised parity soak of the batched hybrid leg (csrc/hybrid.hip) — not part of the test suite: random (queries, top_k, fetch_k = 5 k,
positives per query, corpus size, alpha, polarity, short / empty backend lists, ties) configurations, device rerank vs
oracle/searcher_oracle.py, ids and f32 score bits.  Usage (GPU box): python scripts/soak_hybrid.py [n_configs] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import leann_rs_amd as la
import searcher_oracle as so
from test_gpu_hybrid import _run, _sparse_bm25

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2468)
U64MAX = np.iinfo(np.uint64).max
bad = 0
for c in range(n_cfg):
    top_k = int(rng.integers(1, 52)); fetch_k = 5 * top_k if rng.random() < 0.8 else int(rng.integers(top_k, 257))
    fetch_k = min(fetch_k, 256); top_k = min(top_k, fetch_k)
    nq = int(rng.choice([1, 3, 64, 300])); n_docs = int(rng.choice([60, 500, 5000, 2_000_000])); stride = int(rng.choice([1, 8, 64, 200]))
    alpha = float(rng.choice([0.0, 0.25, 0.7, 1.0])); compat = bool(rng.integers(0, 2))
    keys = np.full((nq, fetch_k), U64MAX, np.uint64); dists = np.full((nq, fetch_k), np.inf, np.float32); counts = np.zeros(nq, np.uint32)
    for q in range(nq):
        cmax = min(fetch_k, n_docs)
        cq = cmax if rng.random() < 0.6 else int(rng.integers(0, cmax + 1))
        keys[q, :cq] = rng.choice(n_docs, size=cq, replace=False)
        dq = np.sort(rng.uniform(0.0, 1.5, size=cq)).astype(np.float32)
        if cq > 3 and rng.random() < 0.3:
            dq[1:3] = dq[1]
        dists[q, :cq], counts[q] = dq, cq
    pos, sc, pcnt = _sparse_bm25(rng, nq, n_docs, stride, keys, counts, min(stride, n_docs))
    gk, gs, gc = _run(la, keys, dists, counts, pos, sc, pcnt, n_docs, alpha, compat, top_k)
    ok = True
    for q in range(nq):
        # (the sparse form of the restatement: equal to the dense one by tests/test_cpu_searcher.py, and no 2M-element Python folds)
        exp = so.hybrid_leg_sparse(keys[q, :counts[q]], dists[q, :counts[q]], list(zip(pos[q, :pcnt[q]].tolist(), sc[q, :pcnt[q]])), n_docs, alpha,
                                   top_k, fetch_k, compat)
        ok &= gc[q] == len(exp) and [int(x) for x in gk[q, :gc[q]]] == [i for i, _ in exp] and \
            (gs[q, :gc[q]].view(np.uint32) == np.array([s for _, s in exp], np.float32).view(np.uint32)).all()
    bad += 0 if ok else 1
    print(f"cfg {c}: nq={nq} top_k={top_k} fetch_k={fetch_k} n_docs={n_docs} positives<= {stride} alpha={alpha} compat={compat}: {'ok' if ok else 'MISMATCH'}", flush=True)
print(f"{n_cfg} configurations, {bad} mismatching")
sys.exit(1 if bad else 0)
