"""IndexSearcher::search_with_options restated — TEST INFRASTRUCTURE ONLY (see oracle/oracle.h).

Follows src/index/searcher.rs:123-210 step by step, in numpy float32 scalars:
    fetch_k = 5 * top_k when a filter or hybrid is on            searcher.rs:129-133
    backend.search(query, fetch_k, complexity)                     :136       (injected: the oracle's graph walk, or fixture data)
    zip to (idx, dist) — the backend DISTANCE is the score (N1)    :139-143
    hybrid: BM25 over all texts, bm25_top = search(text, fetch_k), BM25-only hits appended with vector score 0.0   :146-165
            hybrid_rerank(vector_results, bm25_scores, alpha)      :167       (bm25.rs:135-170, restated below)
    walk the list: stop at top_k, idx -> id_map[idx] or str(idx) (:180-184), passage lookup, filter AFTER the fetch (:190-194),
    a passage that cannot be loaded is skipped with a warning (:203-205)
hybrid_rerank's polarity quirk is reproduced as is: distances enter the blend as if larger were better.

The metadata filter is a small subset of src/index/filter.rs (conditions `field=value`, `field:glob*`, `!=`, `<`, `<=`, `>`, `>=`,
joined by " AND " or commas) — enough for the fixtures; the filter language itself is out of scope.
"""
import numpy as np

import bm25_oracle as bo

f32 = np.float32


def hybrid_rerank(vector_results, bm25_scores, alpha):
    """bm25.rs:135-170 in f32: min-max normalise both, alpha blend, stable sort descending."""
    alpha = f32(alpha)
    max_v, min_v = f32(-np.inf), f32(np.inf)
    for _, s in vector_results:  # f32::max / f32::min folds (:140-147)
        max_v = max(max_v, f32(s))
        min_v = min(min_v, f32(s))
    v_range = max(f32(max_v - min_v), f32(1e-6))
    max_b, min_b = f32(-np.inf), f32(np.inf)
    for b in bm25_scores:  # over ALL N scores (:152-154)
        max_b = max(max_b, f32(b))
        min_b = min(min_b, f32(b))
    b_range = max(f32(max_b - min_b), f32(1e-6))
    out = []
    with np.errstate(invalid="ignore", over="ignore"):
        for idx, s in vector_results:
            norm_vec = f32(f32(f32(s) - min_v) / v_range)
            bm = f32(bm25_scores[idx]) if idx < len(bm25_scores) else f32(0.0)
            norm_b = f32(f32(bm - min_b) / b_range)
            out.append((idx, f32(f32(alpha * norm_vec) + f32(f32(f32(1.0) - alpha) * norm_b))))
    out.sort(key=lambda t: -float(t[1]))  # stable, like sort_by(partial_cmp) descending (:168)
    return out


def hybrid_rerank_sparse(vector_results, positives, n_docs, alpha):
    """hybrid_rerank when the BM25 score vector is given by its non-zero entries: positives = {idx: f32 score}, every other of the
    n_docs passages scores 0.0.  By definition equal to hybrid_rerank(vector_results, dense, alpha) with dense[idx] = score
    (tests/test_cpu_searcher.py checks the two against each other); exists so that a 10M-passage check does not fold 10M zeros in Python."""
    alpha = f32(alpha)
    max_v, min_v = f32(-np.inf), f32(np.inf)
    for _, s in vector_results:
        max_v = max(max_v, f32(s))
        min_v = min(min_v, f32(s))
    v_range = max(f32(max_v - min_v), f32(1e-6))
    max_b, min_b = f32(-np.inf), f32(np.inf)
    for b in positives.values():
        max_b = max(max_b, f32(b))
        min_b = min(min_b, f32(b))
    if len(positives) < n_docs:  # at least one passage scores 0.0
        max_b = max(max_b, f32(0.0))
        min_b = min(min_b, f32(0.0))
    b_range = max(f32(max_b - min_b), f32(1e-6))
    out = []
    with np.errstate(invalid="ignore", over="ignore"):
        for idx, s in vector_results:
            norm_vec = f32(f32(f32(s) - min_v) / v_range)
            bm = f32(positives.get(idx, 0.0)) if idx < n_docs else f32(0.0)
            norm_b = f32(f32(bm - min_b) / b_range)
            out.append((idx, f32(f32(alpha * norm_vec) + f32(f32(f32(1.0) - alpha) * norm_b))))
    out.sort(key=lambda t: -float(t[1]))
    return out


def hybrid_leg_sparse(keys, dists, positives_sorted, n_docs, alpha, top_k, fetch_k, compat_polarity=True):
    """searcher.rs:146-169 on a backend answer (keys, dists) and BM25 positives [(idx, score)] sorted as Bm25Scorer::search sorts
    them (score descending, stable): polarity, injection of BM25-only hits of bm25_top with 0.0, rerank, first top_k."""
    vr = [(int(k), f32(d) if compat_polarity else f32(f32(1.0) - f32(d))) for k, d in zip(keys, dists)]
    have = {i for i, _ in vr}
    for idx, _ in positives_sorted[:fetch_k]:
        if int(idx) not in have:
            vr.append((int(idx), f32(0.0)))
    return hybrid_rerank_sparse(vr, {int(i): f32(s) for i, s in positives_sorted}, n_docs, alpha)[:top_k]


def _parse_value(s):
    try:
        return int(s)
    except ValueError:
        pass
    try:
        return float(s)
    except ValueError:
        pass
    if s == "true":
        return True
    if s == "false":
        return False
    return s


def _single(cond):
    cond = cond.strip()
    for sym, op in (("!=", "ne"), (">=", "ge"), ("<=", "le"), (">", "gt"), ("<", "lt")):
        if sym in cond:
            field, val = cond.split(sym, 1)
            return field, op, _parse_value(val)
    sep = "=" if "=" in cond else ":"
    field, val = cond.split(sep, 1)
    if "*" in val:
        if val.startswith("*") and val.endswith("*") and len(val) > 2:
            return field, "contains", val[1:-1]
        if val.startswith("*"):
            return field, "endswith", val[1:]
        if val.endswith("*"):
            return field, "startswith", val[:-1]
    return field, "eq", _parse_value(val)


def parse_filter(text):
    """subset of MetadataFilter::parse (filter.rs:52-137): AND of single conditions"""
    parts = text.split(" AND ") if " AND " in text else text.split(",")
    conds = [_single(p) for p in parts if p.strip()]

    def num(x):
        return isinstance(x, (int, float)) and not isinstance(x, bool)

    def matches(meta):
        for field, op, val in conds:
            cur = meta
            for part in field.split("."):
                cur = cur.get(part) if isinstance(cur, dict) else None
                if cur is None:
                    break
            if op == "ne":
                if cur is not None and _eq(cur, val):
                    return False
                continue
            if cur is None:
                return False
            if op == "eq":
                ok = _eq(cur, val)
            elif op in ("gt", "ge", "lt", "le"):
                if num(cur) and num(val):
                    c = (cur > val) - (cur < val)
                elif isinstance(cur, str) and isinstance(val, str):
                    c = (cur > val) - (cur < val)
                else:
                    c = 0
                ok = {"gt": c > 0, "ge": c >= 0, "lt": c < 0, "le": c <= 0}[op]
            else:
                ok = isinstance(cur, str) and {"contains": val in cur, "startswith": cur.startswith(val),
                                               "endswith": cur.endswith(val)}[op]
            if not ok:
                return False
        return True

    def _eq(a, b):
        if isinstance(a, bool) or isinstance(b, bool):
            return isinstance(a, bool) and isinstance(b, bool) and a == b
        if num(a) and num(b):
            return abs(float(a) - float(b)) < np.finfo(np.float64).eps
        return isinstance(a, str) and isinstance(b, str) and a == b

    return matches


def search_with_options(backend_search, id_map, passages, query_embedding, top_k, complexity, filter_text=None, hybrid=False,
                        hybrid_alpha=0.7, query_text=None, compat_polarity=True):
    """backend_search(query, fetch_k, complexity) -> (keys, dists), best first, possibly fewer than fetch_k.
    passages: {id: {"text": ..., "metadata": {...}}}.  Returns [(id, f32 score)].
    compat_polarity=True is the reference as written (N1: the backend's DISTANCES enter hybrid_rerank as if larger were better);
    False is the corrected pair SURVEY.md N1 asks for: the ANN hits enter the blend as similarities 1 - dist (f32), BM25-only hits
    still with 0.0 as at searcher.rs:160-165.  Only the hybrid branch is affected."""
    matches = parse_filter(filter_text) if filter_text else None
    fetch_k = top_k * 5 if (matches is not None or hybrid) else top_k
    keys, dists = backend_search(query_embedding, fetch_k, complexity)
    vector_results = [(int(k), f32(d)) for k, d in zip(keys, dists)]
    if hybrid and query_text is not None:
        if not compat_polarity:
            vector_results = [(i, f32(f32(1.0) - d)) for i, d in vector_results]
        all_texts = [passages[i]["text"] if i in passages else "" for i in id_map]  # get_all_texts :213-224
        scorer = bo.Bm25Scorer.build(all_texts)
        bm25_scores = scorer.score_query(query_text)
        bm25_top = scorer.search(query_text, fetch_k)
        have = {i for i, _ in vector_results}
        for idx, _ in bm25_top:
            if idx not in have:
                vector_results.append((idx, f32(0.0)))
        vector_results = hybrid_rerank(vector_results, bm25_scores, hybrid_alpha)
    results = []
    for idx, score in vector_results:
        if len(results) >= top_k:
            break
        pid = id_map[idx] if idx < len(id_map) else str(idx)
        p = passages.get(pid)
        if p is None:
            continue
        if matches is not None and not matches(p.get("metadata", {})):
            continue
        results.append((pid, f32(score)))
    return results
